"""Row N2: the DEVICE generator draws the reference's DISTRIBUTIONS.

Every bit-exact test of the device-resident loop replays the oracle's draws (tests/test_replay_gpu.py), so
the randomness of the loop bench.py times - `DevRng` in csrc/kernels.hip: Dirichlet(alpha) root noise at root
expansions (backup kernels) and at re-rootings (`k_prune`), one uniform symmetry id per non-terminal leaf -
needs a test of its own.  The reference draws them from std::gamma_distribution<float>(alpha, 1) normalised
over the E root edges (MCTS.h:113-132,352-358) and std::uniform_int_distribution (BatchedMCTS.h:33-40,
148-154); `host_rng.h` restates both and is pinned to libstdc++ (tests/golden/rng_std.npz).  Here >= 2e5
noise rows and symmetry ids are pulled out of the device loop and compared with

  * the Dirichlet moments: per-component mean 1/E and variance (E-1)/(E^2 (E alpha + 1)) within 4 sigma
    (sigma of the variance estimate from the sample's own fourth moment) - for alpha >= 0.1; at alpha 0.03 the
    reference's float32 draws and its 1e-8 in the normalisation make many rows sum to less than one, and the
    reference's distribution is then what the host generator draws, not the textbook's,
  * the pinned HOST generator, every alpha: two-sample Kolmogorov-Smirnov of the first and last marginal and of
    the row sums, and the marginals' means,
  * chi-square uniformity of the symmetry ids ({0,1}; Othello {0,2,6,7}),
  * independence: no correlation between tree i and tree i+1 of one call, nor between consecutive calls;

for E in 1..7 (Connect4; roots with full columns) and up to 33 edges (Othello), alpha in {0.03, 0.3, 1.0},
root-expansion and re-rooting paths.  The same checks REJECT a sample drawn with a wrong alpha < 1 boost
(gamma(alpha+1) left unboosted), a biased symmetry draw and a stream shared by neighbouring trees.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import scenarios as S

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "alphazero-al_amd")


@pytest.fixture(scope="module")
def env():
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401  (before the engine library: one HIP runtime per process)
    import __graft_entry__ as ge
    ge.build()
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    import torch
    from src import MCTS_cpp, fused, hash_eval
    return dict(torch=torch, W=MCTS_cpp, F=fused, H=hash_eval)


# ---------------------------------------------------------------------------------------- the checks

def host_dirichlet(L, alpha, E, rows, seed):
    """Rows of the pinned host generator: E gamma(alpha, 1) draws each, normalised as MCTS.h:360-362."""
    out = np.zeros(rows * E, np.float32)
    L.az_rng_gamma_selftest.argtypes = [C.c_uint32, C.c_float, C.c_int, C.c_void_p]
    assert L.az_rng_gamma_selftest(seed, alpha, rows * E, out.ctypes.data) == 0
    g = out.reshape(rows, E)
    inv = np.float32(1.0) / (g.sum(1, dtype=np.float32) + np.float32(1e-8))
    return g * inv[:, None]


def dirichlet_problems(x, alpha, host=None, sigmas=4.0):
    """Reasons why the rows `x` (n, E) do not look like the reference's root noise for this alpha; empty = passes.
    `host`: rows of the pinned host generator for the same (alpha, E).  The reference normalises E float32 gamma
    draws by 1 / (sum + 1e-8) (MCTS.h:360-362): for alpha << 1 a draw is G(alpha + 1) * u^(1/alpha) and most of them
    are tiny, so rows that sum to LESS than one (or to zero) are part of the reference's distribution - the
    textbook Dirichlet moments are only asserted where that artefact is negligible (alpha >= 0.1); the comparison
    with the host generator's rows covers every alpha."""
    from scipy import stats
    n, E = x.shape
    bad = []
    if (x < 0).any() or (x > 1.0 + 1e-6).any():
        bad.append("component outside [0, 1]")
    x = x.astype(np.float64)
    s = x.sum(1)
    if (s > 1.0 + 1e-5).any():
        bad.append("a row sums to more than one")
    if alpha >= 0.1:
        # (for E <= 2 a few percent of the rows of alpha 0.3 still sum to less than one - g < 1e-4 against the 1e-8 in
        # the normalisation - which moves the moments by less than their 4 sigma; the row sums are compared with the
        # host generator's below)
        if E > 1:
            mean, var = 1.0 / E, (E - 1.0) / (E * E * (E * alpha + 1.0))
            m, v = x.mean(0), x.var(0)
            se_m = np.sqrt(v / n)
            if (np.abs(m - mean) > sigmas * se_m + 2e-4).any():
                bad.append("component mean off: %s vs %.5f" % (np.round(m, 5), mean))
            m4 = ((x - m) ** 4).mean(0)
            se_v = np.sqrt(np.maximum(m4 - v * v, 1e-30) / n)
            if (np.abs(v - var) > sigmas * se_v + 2e-4 * var).any():
                bad.append("component variance off: %s vs %.6f (se %s)" % (np.round(v, 6), var, np.round(se_v, 6)))
    if host is not None:
        h = host.astype(np.float64)
        for comp in sorted({0, E - 1}):
            p = stats.ks_2samp(x[:, comp], h[:, comp]).pvalue
            if p < 1e-4:
                bad.append("marginal %d differs from the host generator's (KS p = %.2e)" % (comp, p))
            dm = abs(x[:, comp].mean() - h[:, comp].mean())
            se = np.sqrt(x[:, comp].var() / n + h[:, comp].var() / h.shape[0])
            if dm > sigmas * se + 1e-7:
                bad.append("mean of component %d: %.5f vs host %.5f" % (comp, x[:, comp].mean(), h[:, comp].mean()))
        # row sums: rounding noise around one is not a property of the generator - compared are the rows that fall short
        # of one by more than 1e-4 (the 1e-8 of the normalisation against tiny draws): their share, and at small alpha,
        # where they are many, their distribution
        hs = h.sum(1)
        short, hshort = s < 1.0 - 1e-4, hs < 1.0 - 1e-4
        fa, fb = short.mean(), hshort.mean()
        se = np.sqrt(fa * (1 - fa) / n + fb * (1 - fb) / h.shape[0])
        if abs(fa - fb) > sigmas * se + 1e-4:
            bad.append("share of rows short of one: %.5f vs host %.5f" % (fa, fb))
        if short.sum() > 500 and hshort.sum() > 500:
            p = stats.ks_2samp(s[short], hs[hshort]).pvalue
            if p < 1e-4:
                bad.append("sums of the short rows differ from the host generator's (KS p = %.2e)" % p)
    return bad


def corr_problems(a, b, what, sigmas=4.5):
    a = a.astype(np.float64) - a.mean()
    b = b.astype(np.float64) - b.mean()
    den = np.sqrt((a * a).sum() * (b * b).sum())
    if den == 0:
        return ["%s: constant sample" % what]
    r = float((a * b).sum() / den)
    return ["%s: correlation %.5f over %d pairs" % (what, r, a.size)] if abs(r) > sigmas / np.sqrt(a.size) else []


def uniform_problems(ids, choices):
    from scipy import stats
    cnt = np.array([(ids == c).sum() for c in choices], np.float64)
    if cnt.sum() != ids.size:
        return ["ids outside %s" % (choices,)]
    p = stats.chisquare(cnt).pvalue
    return ["symmetry ids not uniform over %s: %s (chi-square p = %.2e)" % (choices, cnt.astype(int), p)] if p < 1e-4 else []


# ---------------------------------------------------------------------------------------- device draws

def c4_roots_with_edges(E, n):
    """n non-terminal Connect4 roots with exactly E open columns: the 7 - E full columns carry the pattern
    s(col) * [1, 1, -1, -1, 1, 1] (no four in a row in any direction, three stones each), which ones are
    full varies from root to root; X to move."""
    boards = np.zeros((n, 6, 7), np.int8)
    pat = np.array([1, 1, -1, -1, 1, 1], np.int8)
    rng = np.random.default_rng(100 + E)
    checked = {}
    for i in range(n):
        full = tuple(sorted(rng.permutation(7)[:7 - E].tolist()))
        for c in full:
            boards[i, :, c] = pat * (1 if c % 2 == 0 else -1)
        if full not in checked:
            checked[full] = S.np_winner(boards[i]) == 0 and len(S.np_valid(boards[i])) == E
        assert checked[full]
    return boards, np.ones(n, np.int32)


def c4_edge_rows(st, boards):
    """Per-action noise of the root statistics -> {edge count: (tree indices, rows over the open columns)}."""
    noise = st["noise"]
    open_ = boards[:, 0, :] == 0
    out = {}
    cnt = open_.sum(1)
    for E in np.unique(cnt):
        if E == 0:
            continue
        idx = np.nonzero(cnt == E)[0]
        out[int(E)] = (idx, np.stack([noise[i, open_[i]] for i in idx]))
    return out


def ot_edge_rows(st, edges_of_tree):
    noise = st["noise"]
    out = {}
    for i, ed in enumerate(edges_of_tree):
        if ed:
            out.setdefault(len(ed), []).append((i, noise[i, ed]))
    return {E: (np.array([i for i, _ in lst]), np.stack([r for _, r in lst])) for E, lst in out.items()}


def tiled(arrs, times):
    return tuple(np.concatenate([a] * times) for a in arrs)


def make_search(env, game, B, alpha, seed, n_playout):
    w = env["W"].BatchedMCTS(B, 1.4, 5 * n_playout, alpha, n_playout, game_name=game, noise_epsilon=0.25,
                             fpu_reduction=0.2, use_symmetry=True)
    w.seed(seed)
    net = env["H"].HashEvaluator("cuda") if game == "Connect4" else env["H"].OthelloHashEvaluator("cuda")
    return w, net


def leaf_syms(env, w, K):
    torch, F = env["torch"], env["F"]
    out = torch.zeros(w.batch_size * K, dtype=torch.int32, device="cuda")
    flags = torch.zeros(w.batch_size * K, dtype=torch.uint8, device="cuda")
    F.check(F.lib().az_mcts_dev_leaf_syms(w._fused.h, K, out.data_ptr(), F._stream()))
    F.check(F.lib().az_mcts_dev_leaves(w._fused.h, K, None, None, None, flags.data_ptr(), F._stream()))
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(w.batch_size, K), (flags.cpu().numpy().reshape(w.batch_size, K) & 1) != 0


B_C4 = 16384


@pytest.mark.parametrize("alpha", [0.03, 0.3, 1.0])
def test_connect4_root_noise_is_dirichlet_on_both_paths(env, alpha):
    """Root-expansion path (first simulation of a search: backup kernels) and re-rooting path (`k_prune`), every
    edge count 1..7, 16384 rows each: 2 x 7 x 16384 = 2.3e5 rows per alpha."""
    import gc
    L = env["F"].lib()
    torch, F = env["torch"], env["F"]
    problems, rows_seen = [], 0
    for E in range(1, 8):
        boards, turns = c4_roots_with_edges(E, B_C4)
        w, net = make_search(env, "Connect4", B_C4, alpha, seed=11 * E, n_playout=24)
        w.batch_playout(net, boards, turns, vl_batch=4, fused=True)
        rows = c4_edge_rows(w.get_root_stats(), boards)
        assert list(rows) == [E]
        x = rows[E][1]
        rows_seen += x.shape[0]
        host = host_dirichlet(L, alpha, E, 60000, seed=1234 + E)
        problems += ["expansion E=%d: %s" % (E, p) for p in dirichlet_problems(x, alpha, host)]
        if E > 1:
            problems += corr_problems(x[:-1, 0], x[1:, 0], "expansion E=%d, neighbouring trees" % E)
        # re-rooting: the most visited child becomes the root and gets fresh noise over ITS edges
        acts = np.argmax(w.get_visits_count(), 1).astype(np.int32)
        a = torch.from_numpy(acts).cuda()
        F.check(L.az_mcts_dev_prune_roots(w._fused.h, a.data_ptr(), F._stream()))
        torch.cuda.synchronize()
        nb = boards.copy()
        for i in range(B_C4):
            S.np_drop(nb[i], int(acts[i]), 1)
        st2 = w.get_root_stats()
        grown = st2["prior"].sum(1) > 0                               # the child had been expanded by the search
        for E2, (idx, x2) in c4_edge_rows(st2, nb).items():
            keep = grown[idx]
            idx, x2 = idx[keep], x2[keep]
            if x2.shape[0] < 4000:
                continue
            rows_seen += x2.shape[0]
            host2 = host if E2 == E else host_dirichlet(L, alpha, E2, 60000, seed=99 + E2)
            problems += ["re-rooting E=%d: %s" % (E2, p) for p in dirichlet_problems(x2, alpha, host2)]
            if E2 == E and E > 1:                                      # the new row owes nothing to the old one
                problems += corr_problems(x[idx, 0], x2[:, 0], "E=%d, expansion vs re-rooting noise of one tree" % E)
        del w
        gc.collect()
    assert rows_seen >= 200000, rows_seen
    assert not problems, "\n".join(problems)


def test_connect4_consecutive_calls_and_symmetry_ids(env):
    """Symmetry ids of >= 1e5 non-terminal leaves are uniform over {0, 1}; the draws of neighbouring trees, of the
    K leaves of one tree and of consecutive calls are uncorrelated; a later call of one engine does not repeat an
    earlier call's noise.  (2048 distinct openings, each in 8 trees: trees on the SAME root must differ too.)"""
    rng = np.random.default_rng(5)
    boards, turns = tiled(S.random_openings(rng, B_C4 // 8, 10), 8)
    w, net = make_search(env, "Connect4", B_C4, 0.3, seed=3, n_playout=9)        # 1 + 2 x 4: the last call is a K = 4 batch
    w.batch_playout(net, boards, turns, vl_batch=4, fused=True)
    sym1, term1 = leaf_syms(env, w, 4)
    n1 = w.get_root_stats()["noise"].copy()
    w.batch_playout(net, boards, turns, vl_batch=4, fused=True)                   # trees kept: two more K = 4 calls
    sym2, term2 = leaf_syms(env, w, 4)
    problems = []
    ids = np.concatenate([sym1[~term1], sym2[~term2]])
    assert ids.size >= 100000
    problems += uniform_problems(ids, (0, 1))
    assert (sym1[term1] == 0).all()
    both = ~term1[:-1, 0] & ~term1[1:, 0]
    problems += corr_problems(sym1[:-1, 0][both], sym1[1:, 0][both], "symmetry ids of neighbouring trees")
    both = ~term1[:, 0] & ~term1[:, 1]
    problems += corr_problems(sym1[:, 0][both], sym1[:, 1][both], "symmetry ids of two leaves of one tree")
    both = ~term1[:, 0] & ~term2[:, 0]
    problems += corr_problems(sym1[:, 0][both], sym2[:, 0][both], "symmetry ids of consecutive calls")
    half = B_C4 // 8                                                              # tree i and tree i + 2048 search the same root
    problems += corr_problems(n1[:half].max(1), n1[half:2 * half].max(1), "noise of two trees on the same root")
    # a second engine, same seed, same roots: identical draws; its NEXT search after a reset: unrelated ones
    w2, _ = make_search(env, "Connect4", B_C4, 0.3, seed=3, n_playout=9)
    w2.batch_playout(net, boards, turns, vl_batch=4, fused=True)
    assert np.array_equal(w2.get_root_stats()["noise"], n1)
    for i in range(64):
        w2.reset_env(i)
    w2.batch_playout(net, boards, turns, vl_batch=4, fused=True)
    assert not np.array_equal(w2.get_root_stats()["noise"][:64], n1[:64]), "a later call must not repeat the first call's noise"
    assert not problems, "\n".join(problems)


def test_othello_noise_up_to_33_edges_and_symmetry_subset(env):
    """Othello: roots with 1..~20 legal moves from random play (1024 distinct openings, each in 40 trees; every
    edge count with >= 1200 rows is checked), the widest root the openings hold in 8192 trees of its own, and
    symmetry ids uniform over the subset {0, 2, 6, 7} (Othello.h:363-367)."""
    import gc
    L = env["F"].lib()
    rng = np.random.default_rng(17)
    n0, rep = 1024, 40
    b0, t0 = S.ot_openings(rng, n0, 30)
    e0 = []
    for i in range(n0):
        mv = S.ot_moves(b0[i], int(t0[i]))
        e0.append(mv if mv else [S.OT_PASS])
    boards, turns = tiled((b0, t0), rep)
    edges = e0 * rep
    B = n0 * rep
    problems, checked = [], 0
    for alpha in (0.3, 0.03, 1.0):
        w, net = make_search(env, "Othello", B, alpha, seed=29, n_playout=5)
        w.batch_playout(net, boards, turns, vl_batch=4, fused=True)
        for E, (idx, x) in sorted(ot_edge_rows(w.get_root_stats(), edges).items()):
            if x.shape[0] < 1200:
                continue
            host = host_dirichlet(L, alpha, E, 40000, seed=500 + E)
            problems += ["Othello alpha=%.2f E=%d: %s" % (alpha, E, p) for p in dirichlet_problems(x, alpha, host)]
            checked += x.shape[0]
        if alpha == 0.3:
            sym, term = leaf_syms(env, w, 4)
            ids = sym[~term]
            assert ids.size >= 50000
            problems += uniform_problems(ids, (0, 2, 6, 7))
            both = ~term[:-1, 0] & ~term[1:, 0]
            problems += corr_problems((sym[:-1, 0][both] == 0) * 1.0, (sym[1:, 0][both] == 0) * 1.0, "Othello ids of neighbouring trees")
        del w
        gc.collect()
    assert checked >= 100000, checked
    # the widest root the openings hold, alone in 8192 trees: up to 33 lanes of one wavefront draw one row
    Emax = max(len(e) for e in e0)
    i0 = next(i for i in range(n0) if len(e0[i]) == Emax)
    assert Emax >= 14
    wide_b, wide_t = np.repeat(b0[i0:i0 + 1], 8192, 0), np.repeat(t0[i0:i0 + 1], 8192, 0)
    w, net = make_search(env, "Othello", 8192, 0.3, seed=31, n_playout=1)
    w.batch_playout(net, wide_b, wide_t, vl_batch=1, fused=True)
    x = ot_edge_rows(w.get_root_stats(), [e0[i0]] * 8192)[Emax][1]
    problems += ["Othello E=%d (one root, 8192 trees): %s" % (Emax, p)
                 for p in dirichlet_problems(x, 0.3, host_dirichlet(L, 0.3, Emax, 20000, seed=7))]
    problems += corr_problems(x[:-1, 0], x[1:, 0], "Othello E=%d, neighbouring trees on one root" % Emax)
    assert not problems, "\n".join(problems)


def test_the_checks_reject_wrong_generators(env):
    """Power of the checks above, on host-made samples: an alpha < 1 boost left out (gamma(alpha + 1) rows), a
    wrong alpha, a biased symmetry draw and streams shared by neighbouring trees must all be caught."""
    L = env["F"].lib()
    good = host_dirichlet(L, 0.3, 7, 40000, seed=1)
    ref = host_dirichlet(L, 0.3, 7, 40000, seed=2)
    assert not dirichlet_problems(good, 0.3, ref)
    unboosted = host_dirichlet(L, 1.3, 7, 40000, seed=3)            # Marsaglia-Tsang at alpha + 1 without the u^(1/alpha) factor
    assert dirichlet_problems(unboosted, 0.3, ref)
    assert dirichlet_problems(host_dirichlet(L, 0.25, 7, 40000, seed=4), 0.3, ref)
    assert dirichlet_problems(host_dirichlet(L, 0.03, 5, 40000, seed=5), 0.3, None)
    rng = np.random.default_rng(0)
    assert uniform_problems((rng.random(200000) < 0.51).astype(np.int32), (0, 1))
    assert not uniform_problems((rng.random(200000) < 0.5).astype(np.int32), (0, 1))
    shared = np.repeat(good[:20000, 0], 2)                          # trees 2i and 2i+1 on one stream
    assert corr_problems(shared[:-1], shared[1:], "shared stream")
    assert not corr_problems(good[:-1, 0], good[1:, 0], "independent rows")
