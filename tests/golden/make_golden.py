#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference, compiled into oracle/_ref.

Run in the build container only (needs /root/reference):

    make -C oracle ref && OMP_NUM_THREADS=1 python tests/golden/make_golden.py

OMP_NUM_THREADS=1 is required: the reference seeds one mt19937 per OpenMP thread
(BatchedMCTS.h:68-84), so seeded streams are only machine-independent with one thread.

Nothing from the reference is copied: its compiled extensions are imported from
oracle/_ref/native, its unmodified Python (player.py, game.py, MCTS_cpp.py, Network.py) from
/root/reference through a namespace-package overlay.  The committed
outputs are data: seeded inputs and the reference's outputs for them, plus the tensors of
the checkpoint the reference ships (params/Connect4/001/current/model.pt, loaded with
weights_only=True) so the GPU box can evaluate the same network.
"""
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("AZ_REFERENCE", "/root/reference")
assert os.environ.get("OMP_NUM_THREADS") == "1", "run with OMP_NUM_THREADS=1"

sys.path[:0] = [os.path.join(ROOT, "oracle", "_ref", "native"), REF, os.path.join(ROOT, "tests")]
numba = types.ModuleType("numba")
numba.njit = lambda *a, **k: (lambda f: f)
sys.modules["numba"] = numba

import scenarios as S                      # noqa: E402
from src import mcts_cpp                   # noqa: E402  (compiled reference)
from src.env_cpp.connect4 import Env       # noqa: E402  (compiled reference)
from src.MCTS_cpp import BatchedMCTS       # noqa: E402  (reference python, unmodified)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------ rng_std
def gen_rng():
    exe = "/tmp/std_rng_dump"
    subprocess.check_call(["g++", "-std=c++20", "-O3", "-march=native",
                           os.path.join(HERE, "std_rng_dump.cpp"), "-o", exe])
    raw = subprocess.check_output([exe])
    per = 16 + 1024 + 1024 + 4 * 200 * 7 + 1
    a = np.frombuffer(raw, np.uint32).reshape(3, per)
    save("rng_std",
         seeds=np.array([0, 1234, 20071], np.uint32),
         alphas=np.array([0.3, 0.03, 1.0, 2.5], np.float32),
         mt=a[:, :16].copy(),
         bits=a[:, 16:1040].astype(np.int32),
         ranged=a[:, 1040:2064].astype(np.int32),
         gamma=a[:, 2064:2064 + 5600].copy().view(np.float32).reshape(3, 4, 200, 7),
         tail=a[:, -1].copy())


# ------------------------------------------------------------------ G1 game logic
def gen_g1():
    rng = np.random.default_rng(101)
    rec = {k: [] for k in ("board", "turn", "winner", "full", "done", "mask", "state",
                           "mirror", "mirror_state", "action", "game")}
    for g in range(160):
        e = Env()
        bias = rng.integers(0, 3)      # 0 uniform, 1 prefer few columns (fast wins), 2 fill up
        while True:
            valid = e.valid_move()
            rec["board"].append(np.asarray(e.board, np.float32).astype(np.int8))
            rec["turn"].append(e.turn)
            rec["winner"].append(e.winPlayer())
            rec["full"].append(e.check_full())
            rec["done"].append(e.done())
            rec["mask"].append(np.array(e.valid_mask(), np.uint8))
            rec["state"].append(e.current_state()[0].astype(np.int8))
            m = e.apply_symmetry(1)
            rec["mirror"].append(np.asarray(m.board, np.float32).astype(np.int8))
            rec["mirror_state"].append(m.current_state()[0].astype(np.int8))
            rec["game"].append(g)
            if e.done():
                rec["action"].append(-1)
                break
            if bias == 1:
                pref = [c for c in valid if c in (2, 3, 4)] or valid
                a = int(rng.choice(pref))
            elif bias == 2:
                a = int(valid[(len(rec["game"]) * 3) % len(valid)])
            else:
                a = int(rng.choice(valid))
            rec["action"].append(a)
            e.step(a)
    out = {k: np.array(v) for k, v in rec.items()}
    # board setter / constructor: turn inferred from parity (env_common.h:55-70), including a
    # board with a floating piece (sync_from_board stops at the first gap, Connect4.h:109-121)
    odd = np.zeros((4, 6, 7), np.float32)
    odd[0, 5, 3] = 1
    odd[1, 5, 3] = 1; odd[1, 5, 4] = -1
    odd[2, 5, 0] = 1; odd[2, 3, 0] = -1                      # floating piece above a gap
    odd[3, 5, :] = [1, -1, 1, -1, 1, -1, 1]
    setter = []
    for b in odd:
        e = Env(b)
        setter.append(dict(turn=e.turn, board=np.asarray(e.board), mask=e.valid_mask(),
                           winner=e.winPlayer()))
    save("g1_game_logic", **{k: v.astype(np.int8) if v.dtype != np.bool_ else v.astype(np.uint8)
                             for k, v in out.items()},
         setter_in=odd.astype(np.int8),
         setter_turn=np.array([s["turn"] for s in setter], np.int8),
         setter_board=np.array([s["board"] for s in setter]).astype(np.int8),
         setter_mask=np.array([s["mask"] for s in setter], np.uint8))


# ------------------------------------------------------------------ G2-G5 search scenarios
def gen_search():
    for name in S.SEARCH_SCENARIOS:
        r = S.run_search_scenario(mcts_cpp.BatchedMCTS_Connect4, name)
        save(name, **r)


# ------------------------------------------------------------------ G2 hand-built single calls
def gen_g2():
    """One search_batch / backprop_batch round trip on hand-built positions; raw outputs."""
    b = np.zeros((6, 6, 7), np.int8)
    t = np.ones(6, np.int32)
    # 0: empty board; 1: P1 has three in a row on the bottom (win available at col 3)
    b[1, 5, 0:3] = 1; b[1, 4, 0:3] = -1; t[1] = 1
    # 2: P2 to move, fresh tree (root turn quirk)
    b[2, 5, 3] = 1; t[2] = -1
    # 3: root already won by P1 (terminal root), P2 "to move"
    b[3, 5, 0:4] = 1; b[3, 4, 0:3] = -1; t[3] = -1
    # 4: full board draw pattern
    pat = np.array([[1, 1, -1, -1, 1, 1, -1]] * 2 + [[-1, -1, 1, 1, -1, -1, 1]] * 2 +
                   [[1, 1, -1, -1, 1, 1, -1]] * 2, np.int8)
    b[4] = pat; t[4] = 1
    # 5: one empty cell left (col 6 top), move fills the board
    b[5] = pat; b[5, 0, 6] = 0; t[5] = -1
    m = mcts_cpp.BatchedMCTS_Connect4(6)
    S.apply_cfg(m, S.DET_CFG)
    outs = {}
    for it in range(6):
        res = m.search_batch(b, t)
        for j, nm in enumerate(("lb", "td", "t1", "t2", "it", "lt", "vm")):
            outs[f"s{it}_{nm}"] = res[j]
        lb, td, t1, t2, itm, lt, vm = res
        pr, wdl, ml = S.hash_eval(lb, lt)
        d, p1, p2 = S.rel_to_abs(wdl, lt)
        nt = ~itm.astype(bool)
        probs = np.where(nt[:, None], pr * vm, 0).astype(np.float32)
        m.backprop_batch(probs, np.where(nt, d, td), np.where(nt, p1, t1), np.where(nt, p2, t2),
                         np.where(nt, ml, 0).astype(np.float32), itm)
        outs[f"s{it}_counts"] = np.array(m.get_all_counts(), np.int32)
        outs[f"s{it}_stats"] = m.get_all_root_stats()
    # VL round with K=3 then remove_all_vl twice (idempotence), stats must be unchanged by it
    res = m.search_batch_vl(3, b, t)
    for j, nm in enumerate(("lb", "td", "t1", "t2", "it", "lt", "sy", "vm")):
        outs[f"vl_{nm}"] = res[j]
    m.remove_all_vl(3); m.remove_all_vl(3)
    outs["vl_removed_stats"] = m.get_all_root_stats()
    res2 = m.search_batch(b, t)     # same leaves as a clean tree would give
    outs["after_remove_lb"] = res2[0]; outs["after_remove_it"] = res2[4]
    save("g2_single_calls", boards=b, turns=t, **outs)


# ------------------------------------------------------------------ G6 wrapper
def gen_g6():
    rng = np.random.default_rng(66)
    boards, turns = S.random_openings(rng, 24, 8)
    out = {}
    for tag, cache, K, seed in (("nocache_k4", 0, 4, 5), ("cache_k4", 4096, 4, 5),
                                ("nocache_k1", 0, 1, 9), ("cache_k1", 64, 1, 9)):
        w = BatchedMCTS(24, c_init=1.4, c_base=250, alpha=0.3, n_playout=50, game_name="Connect4",
                        cache_size=cache, noise_epsilon=0.25, fpu_reduction=0.2, use_symmetry=True,
                        mlh_slope=0.1, mlh_cap=0.2)
        w.seed(seed)
        pv = S.HashPV()
        b, t = boards.copy(), turns.copy()
        cs, ss = [], []
        for ply in range(3):
            w.batch_playout(pv, b, t, vl_batch=K)
            c = w.get_visits_count()
            cs.append(c.astype(np.int32)); ss.append(w.mcts.get_all_root_stats())
            acts = np.argmax(c, 1).astype(np.int32)
            w.prune_roots(acts)
            for i in range(24):
                if not S.np_done(b[i]) and b[i][0, acts[i]] == 0:
                    S.np_drop(b[i], int(acts[i]), int(t[i])); t[i] = -t[i]
        out[f"{tag}_counts"] = np.stack(cs); out[f"{tag}_stats"] = np.stack(ss)
        out[f"{tag}_calls"] = np.array(pv.calls, np.int32)
        if cache:
            out[f"{tag}_cache_len"] = np.array([len(w.cache)], np.int32)
    save("g6_wrapper", boards=boards, turns=turns, **out)


# ------------------------------------------------------------------ G9 rollout search
def gen_rollout():
    rng = np.random.default_rng(77)
    boards, turns = S.random_openings(rng, 12, 12)
    w = BatchedMCTS(12, c_init=4, c_base=500, alpha=0, n_playout=120, game_name="Connect4",
                    noise_epsilon=0.0, fpu_reduction=0.0, use_symmetry=False)   # player.py:84-88
    w.seed(3)
    w.rollout_playout(boards, turns)
    save("g9_rollout", boards=boards, turns=turns, counts=w.get_visits_count().astype(np.int32),
         stats=w.mcts.get_all_root_stats())
    # root noise drawn between the playout moves (alpha > 0), two searches on the same trees, and Othello
    out = {}
    rng = np.random.default_rng(78)
    boards, turns = S.random_openings(rng, 24, 20)
    w = BatchedMCTS(24, c_init=4, c_base=500, alpha=0.3, n_playout=90, game_name="Connect4",
                    noise_epsilon=0.25, fpu_reduction=0.2, use_symmetry=False)
    w.seed(11)
    w.rollout_playout(boards, turns)
    out.update(c4_boards=boards, c4_turns=turns, c4_counts1=w.get_visits_count().astype(np.int32),
               c4_stats1=w.mcts.get_all_root_stats())
    w.rollout_playout(boards, turns)
    out.update(c4_counts2=w.get_visits_count().astype(np.int32), c4_stats2=w.mcts.get_all_root_stats())
    b, t = S.ot_openings(rng, 16, 44, 2)
    w = BatchedMCTS(16, c_init=4, c_base=300, alpha=0.3, n_playout=60, game_name="Othello",
                    noise_epsilon=0.25, fpu_reduction=0.2, use_symmetry=False, score_utility_factor=0.15, score_scale=8.0)
    w.seed(5)
    w.rollout_playout(b, t)
    out.update(ot_boards=b, ot_turns=t, ot_counts=w.get_visits_count().astype(np.int32), ot_stats=w.mcts.get_all_root_stats())
    save("g9_rollout_more", **out)


# ------------------------------------------------------------------ G7 network
def gen_g7():
    import torch
    from src.environments.Connect4.Network import CNN
    torch.manual_seed(0)
    net = CNN(lr=1e-3, device="cpu")
    net.eval()
    sd = torch.load(os.path.join(REF, "params/Connect4/001/current/model.pt"),
                    weights_only=True, map_location="cpu")
    rng = np.random.default_rng(7)
    boards, turns = S.random_openings(rng, 256, 30)
    planes = np.stack([(boards == turns[:, None, None]), (boards == -turns[:, None, None]),
                       np.ones_like(boards) * turns[:, None, None]], 1).astype(np.float32)
    masks = (boards[:, 0, :] == 0)
    p0, w0, m0 = net.predict(planes, masks)          # random init: zero-init heads
    net.load_state_dict(sd, strict=True)
    p1, w1, m1 = net.predict(planes, masks)
    with torch.no_grad():
        lp, lv, st = net(torch.from_numpy(planes), action_mask=torch.from_numpy(masks))
    save("g7_network", boards=boards, turns=turns, masks=masks.astype(np.uint8),
         init_probs=p0, init_wdl=w0, init_ml=m0,
         ckpt_probs=p1, ckpt_wdl=w1, ckpt_ml=m1,
         ckpt_logp=lp.numpy(), ckpt_logv=lv.numpy(), ckpt_steps=st.numpy())
    save("g7_checkpoint_weights", **{k: v.numpy() for k, v in sd.items()})


# ------------------------------------------------------------------ G8 self-play harness
def gen_g8():
    from src.game import Game
    from src.player import AlphaZeroPlayer
    pv = S.HashPV()
    np.random.seed(11)
    player = AlphaZeroPlayer(pv, n_envs=8, c_init=1.4, c_base=160, n_playout=32, alpha=0.3,
                             is_selfplay=1, noise_epsilon=0.25, fpu_reduction=0.2,
                             use_symmetry=True, mlh_slope=0.1, mlh_cap=0.2, vl_batch=4)
    player.mcts.seed(21)
    game = Game(Env())
    data = game.batch_self_play(player, 8, temperature=1.0, temp_decay_moves=6, temp_endgame=0,
                                td_steps=2)
    out = {}
    for i, (winner, play) in enumerate(data):
        out[f"g{i}_winner"] = np.array([winner], np.int32)
        for j, nm in enumerate(("state", "prob", "z", "steps", "aux", "root_wdl", "mask", "fut")):
            out[f"g{i}_{nm}"] = np.array([np.asarray(tup[j]) for tup in play])
    save("g8_selfplay", **out)


# ------------------------------------------------------------------ G10 self-play harness without native randomness
def gen_g10():
    """The same harness as G8 with Dirichlet noise and symmetry off (the epsilon prior scaling
    stays on): every random draw is numpy's, so a device-resident self-play driver that
    samples moves with the reference's numpy procedure must reproduce these tuples bit for
    bit."""
    from src.game import Game
    from src.player import AlphaZeroPlayer
    pv = S.HashPV()
    np.random.seed(5)
    player = AlphaZeroPlayer(pv, n_envs=16, c_init=1.4, c_base=240, n_playout=48, alpha=0.0,
                             is_selfplay=1, noise_epsilon=0.25, fpu_reduction=0.2,
                             use_symmetry=False, mlh_slope=0.1, mlh_cap=0.2, vl_batch=4)
    player.mcts.seed(3)
    data = Game(Env()).batch_self_play(player, 16, temperature=1.0, temp_decay_moves=8, temp_endgame=0,
                                       td_steps=2)
    out = {}
    for i, (winner, play) in enumerate(data):
        out[f"g{i}_winner"] = np.array([winner], np.int32)
        for j, nm in enumerate(("state", "prob", "z", "steps", "aux", "root_wdl", "mask", "fut")):
            out[f"g{i}_{nm}"] = np.array([np.asarray(tup[j]) for tup in play])
    # what client.py:366-368,386-388 would POST for these games
    import pickle
    payload = pickle.dumps({'__az__': True, 'data': [play for _, play in data]}, protocol=pickle.HIGHEST_PROTOCOL)
    out["upload_payload"] = np.frombuffer(payload, dtype=np.uint8)
    out["upload_numpy_version"] = np.array([int(x) for x in np.__version__.split(".")[:2]], np.int32)
    out["upload_python_version"] = np.array(sys.version_info[:2], np.int32)
    save("g10_selfplay_numpy_rng", **out)


def gen_g14():
    """G10's harness with the noise-epsilon decay on (AlphaZeroPlayer.noise_steps, game.py:87-91): the
    epsilon that scales the root priors falls from 0.25 to 0.05 over the first 6 plies.  alpha = 0, so
    no Dirichlet draw is consumed and every random number is numpy's."""
    from src.game import Game
    from src.player import AlphaZeroPlayer
    pv = S.HashPV()
    np.random.seed(29)
    player = AlphaZeroPlayer(pv, n_envs=12, c_init=1.4, c_base=240, n_playout=48, alpha=0.0,
                             is_selfplay=1, noise_epsilon=0.25, fpu_reduction=0.2,
                             use_symmetry=False, mlh_slope=0.1, mlh_cap=0.2, vl_batch=4,
                             noise_steps=6, noise_eps_min=0.05)
    player.mcts.seed(3)
    data = Game(Env()).batch_self_play(player, 12, temperature=1.0, temp_decay_moves=8, temp_endgame=0,
                                       td_steps=2)
    out = {}
    for i, (winner, play) in enumerate(data):
        out[f"g{i}_winner"] = np.array([winner], np.int32)
        for j, nm in enumerate(("state", "prob", "z", "steps", "aux", "root_wdl", "mask", "fut")):
            out[f"g{i}_{nm}"] = np.array([np.asarray(tup[j]) for tup in play])
    save("g14_selfplay_noise_decay", **out)


# ------------------------------------------------------------------ G11: the harness without virtual loss and without td targets
def gen_g11():
    """As G10 with the other branches of the harness: vl_batch=1 (the plain search loop,
    MCTS_cpp.py:110-209), td_steps=0 (7-tuples, game.py:136-141), no temperature switch
    (temp_decay_moves=0: every move sampled), value_decay < 1 and fpu_reduction 0.4."""
    from src.game import Game
    from src.player import AlphaZeroPlayer
    pv = S.HashPV()
    np.random.seed(17)
    player = AlphaZeroPlayer(pv, n_envs=8, c_init=1.25, c_base=500, n_playout=40, alpha=0.0,
                             is_selfplay=1, noise_epsilon=0.0, fpu_reduction=0.4,
                             use_symmetry=False, mlh_slope=0.0, mlh_cap=0.2, vl_batch=1, value_decay=0.98)
    player.mcts.seed(1)
    data = Game(Env()).batch_self_play(player, 8, temperature=0.8, temp_decay_moves=0, temp_endgame=0, td_steps=0)
    out = {}
    for i, (winner, play) in enumerate(data):
        out[f"g{i}_winner"] = np.array([winner], np.int32)
        for j, nm in enumerate(("state", "prob", "z", "steps", "aux", "root_wdl", "mask")):
            out[f"g{i}_{nm}"] = np.array([np.asarray(tup[j]) for tup in play])
    save("g11_selfplay_plain_search", **out)


# ------------------------------------------------------------------ G12: the self-play harness on Othello
def gen_g12():
    """Game.batch_self_play + AlphaZeroPlayer on Othello (pass action, terminal disc difference
    as auxiliary target, score utility in the search): no native randomness, numpy sampling."""
    from src.game import Game
    from src.player import AlphaZeroPlayer
    from src.env_cpp.othello import Env as OEnv
    pv = S.OthelloHashPV()
    np.random.seed(23)
    player = AlphaZeroPlayer(pv, n_envs=8, c_init=1.4, c_base=160, n_playout=32, alpha=0.0,
                             is_selfplay=1, noise_epsilon=0.25, fpu_reduction=0.2,
                             use_symmetry=False, game_name='Othello', score_utility_factor=0.15, score_scale=8.0,
                             vl_batch=4)
    player.mcts.seed(4)
    data = Game(OEnv()).batch_self_play(player, 8, temperature=1.0, temp_decay_moves=10, temp_endgame=0, td_steps=2)
    out = {}
    for i, (winner, play) in enumerate(data):
        out[f"g{i}_winner"] = np.array([winner], np.int32)
        for j, nm in enumerate(("state", "prob", "z", "steps", "aux", "root_wdl", "mask", "fut")):
            out[f"g{i}_{nm}"] = np.array([np.asarray(tup[j]) for tup in play])
    save("g12_selfplay_othello", **out)


# ------------------------------------------------------------------ Gomoku Env (surface only: no search binding in the reference)
def gen_g13():
    """The reference's Othello network (src/environments/Othello/Network.py) at a small width, with
    non-trivial BatchNorm statistics and non-zero output layers: weights, inputs, and what forward
    and predict return on the CPU.  No Othello checkpoint ships with the reference."""
    import torch
    from src.environments.Othello.Network import CNN
    from src.env_cpp.othello import Env as OEnv
    torch.manual_seed(13)
    net = CNN(lr=1e-3, h_dim=32, num_res_blocks=2, device="cpu")
    rng = np.random.default_rng(13)
    planes, masks = [], []
    for g in range(24):                                   # positions of random games, every 3rd ply
        e = OEnv()
        ply = 0
        while not e.done():
            if ply % 3 == g % 3:
                planes.append(e.current_state()[0].astype(np.float32)); masks.append(np.asarray(e.valid_mask()).astype(bool))
            e.step(int(rng.choice(e.valid_move())))
            ply += 1
    planes, masks = np.stack(planes)[:400], np.stack(masks)[:400]
    with torch.no_grad():
        for m in net.modules():                           # zero-initialised outputs would hide the heads
            if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)) and float(m.weight.abs().sum()) == 0.0:
                m.weight.normal_(0.0, 0.1); m.bias.normal_(0.0, 0.1)
    net.train()
    with torch.no_grad():
        for _ in range(3):                                # moves the running statistics off (0, 1)
            net(torch.from_numpy(planes[:128] * 1.0), action_mask=torch.from_numpy(masks[:128]))
    net.eval()
    with torch.no_grad():
        lp, lv, aux = net(torch.from_numpy(planes), action_mask=torch.from_numpy(masks))
    p, w, u = net.predict(planes, masks)
    save("g13_othello_network", planes=planes.astype(np.int8), masks=masks.astype(np.uint8), logp=lp.numpy(), logv=lv.numpy(),
         aux=aux.numpy(), probs=p, wdl=w, utility=u)
    save("g13_othello_weights", **{k: v.numpy() for k, v in net.state_dict().items()})


def gen_gomoku():
    from src.env_cpp.gomoku import Env as GEnv
    rng = np.random.default_rng(77)
    out = {}
    for gi, (size, need) in enumerate(((15, 5), (9, 5), (7, 4), (5, 3), (6, 6), (3, 3))):
        for rep in range(3):
            e = GEnv(size, need)
            acts, boards, turns, winners, dones, states = [], [], [], [], [], []
            while not e.done():
                a = int(rng.choice(e.valid_move()))
                e.step(a)
                acts.append(a); boards.append(np.asarray(e.board).astype(np.int8)); turns.append(e.turn)
                winners.append(e.winPlayer()); dones.append(e.done()); states.append(e.current_state()[0].astype(np.int8))
            k = f"s{gi}r{rep}"
            out[k + "_cfg"] = np.array([size, need], np.int32)
            out[k + "_actions"] = np.array(acts, np.int32)
            out[k + "_boards"] = np.array(boards)
            out[k + "_turns"] = np.array(turns, np.int32)
            out[k + "_winners"] = np.array(winners, np.int32)
            out[k + "_dones"] = np.array(dones, np.uint8)
            out[k + "_states"] = np.array(states)
            # a mid-game position under the 8 symmetries, and the action map
            mid = GEnv(size, need)
            for a in acts[: max(1, len(acts) // 2)]:
                mid.step(a)
            out[k + "_sym_boards"] = np.array([np.asarray(mid.apply_symmetry(sid).board).astype(np.int8) for sid in range(8)])
            out[k + "_sym_actions"] = np.array([[mid.inverse_symmetry_action(sid, a) for a in range(size * size)]
                                                for sid in range(8)], np.int32)
            # position imported through the board setter: side to move and result are re-derived
            imp = GEnv(size, need)
            imp.board = boards[-1].astype(np.float32)
            out[k + "_import"] = np.array([imp.turn, imp.winPlayer(), int(imp.done()), int(imp.check_full())], np.int32)
    save("g1_gomoku_logic", **out)


# ------------------------------------------------------------------ Othello (config 4)
def gen_othello():
    from src.env_cpp.othello import Env as OEnv
    for name in S.OTHELLO_SCENARIOS:
        cfg, boards, turns, n, K, plies, seed = S.othello_scenario_inputs(name)
        r = S.run_othello_scenario(mcts_cpp.BatchedMCTS_Othello, name, inputs=(boards, turns))
        save(name, in_boards=boards, in_turns=turns, **r)
    # game logic of env_cpp.othello.Env: random games with forced passes, all 8 symmetries
    rng = np.random.default_rng(202)
    rec = {k: [] for k in ("board", "turn", "winner", "full", "done", "mask", "state", "action", "game", "syms")}
    for g in range(40):
        e = OEnv()
        while True:
            rec["board"].append(np.asarray(e.board, np.float32).astype(np.int8))
            rec["turn"].append(e.turn); rec["winner"].append(e.winPlayer()); rec["full"].append(e.check_full())
            rec["done"].append(e.done()); rec["mask"].append(np.array(e.valid_mask(), np.uint8))
            rec["state"].append(e.current_state()[0].astype(np.int8)); rec["game"].append(g)
            rec["syms"].append(np.stack([np.asarray(e.apply_symmetry(sid).board).astype(np.int8) for sid in range(8)]))
            if e.done():
                rec["action"].append(-1)
                break
            a = int(rng.choice(e.valid_move()))
            rec["action"].append(a)
            e.step(a)
    inv = np.array([[OEnv.inverse_symmetry_action(sid, a) for a in range(65)] for sid in range(8)], np.int32)
    save("g1_othello_logic", inv_action=inv, **{k: np.array(v) for k, v in rec.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["rng", "g1", "g2", "search", "g6", "rollout", "g7", "g8", "g10", "g11", "g12", "g13", "g14", "gomoku", "othello"]
    fns = dict(rng=gen_rng, g1=gen_g1, g2=gen_g2, search=gen_search, g6=gen_g6,
               rollout=gen_rollout, g7=gen_g7, g8=gen_g8, g10=gen_g10, g11=gen_g11, g12=gen_g12, g13=gen_g13, g14=gen_g14, gomoku=gen_gomoku, othello=gen_othello)
    for w in which:
        fns[w]()
