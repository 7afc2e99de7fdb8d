// kernels.hip - gfx950 kernels of the batched PUCT search, instantiated per game (games.h).
//
// Work decomposition: ONE TREE PER LANE GROUP of G::LANES lanes (Connect4: 8 lanes, 8 trees
// per 64-wide wavefront; Othello: 64 lanes, one tree per wavefront), one wavefront per
// workgroup (no LDS, no barriers: all cross-lane traffic is group-wide shuffles).  Lane e of a
// group owns edge e of whatever node the group is looking at, so a PUCT step is
//   1 coalesced load of E adjacent 32-byte child records (tree_layout.h)
//   an E-term ordered prior sum + log2(LANES)-step (score, index) max over the group
//   <= 2 dword stores by the winning lane (in-flight count, lazily set flags).
// Trees are independent (reference: one MCTS object per env under `omp parallel for`,
// BatchedMCTS.h:107-332), so there is no inter-group or inter-workgroup communication at all.
// The K virtual-loss descents of one tree are sequential, exactly as in the reference
// (BatchedMCTS.h:249-285); groups of a wave do not wait for each other between descents.
//
// Memory-ordering rule used throughout: a record that is re-read later in the same kernel is
// always re-read by the SAME LANE that wrote it (lane = index in its sibling block for
// selection, lane = depth mod LANES for backup), so plain program order is sufficient.
//
// Arithmetic: compiled with -ffp-contract=off; IEEE fp32 divide and sqrt.  The expression
// order of MCTS.h:140-234,329-402 and MCTSNode.h:116-133 is kept, and the three fused
// multiply-adds the compiled reference contains (FPU value, root prior/noise mix, value
// decay - see oracle/mcts_impl.inc) are explicit fmaf().  No MFMA: this is index/bit work.
#include "kernels.h"

#include <cstdlib>

#include "games.h"

namespace az {
namespace {

constexpr int WAVE = 64;

// ------------------------------------------------------------------ small device helpers

__device__ __forceinline__ float mean_q(int n, float w1, float w2, bool turn_p1)
{
    // MCTSNode.h:116-125: uniform WDL (q = 0) without visits; inv = 1/N then multiply
    if (n == 0) return 0.0f;
    const float inv = 1.0f / static_cast<float>(n);
    const float p1 = w1 * inv, p2 = w2 * inv;
    return turn_p1 ? (p1 - p2) : (p2 - p1);
}

__device__ __forceinline__ float mean_m(int n, float msum)
{
    return n == 0 ? 0.0f : msum / static_cast<float>(n);   // MCTSNode.h:131-133
}

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

// counter-based generator of the dev_* entry points
struct DevRng {
    uint64_t s;
    __device__ DevRng(uint64_t seed, uint64_t call, uint64_t a, uint64_t b)
        : s(mix64(seed ^ mix64(call + 0x9E3779B97F4A7C15ull * (a + 1)) ^ (b << 32))) {}
    __device__ uint32_t next() { s += 0x9E3779B97F4A7C15ull; return static_cast<uint32_t>(mix64(s) >> 32); }
    __device__ float uniform() { return (static_cast<float>(next() >> 8) + 0.5f) * (1.0f / 16777216.0f); }
    __device__ float normal()
    {
        const float u1 = uniform(), u2 = uniform();
        return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530718f * u2);
    }
    __device__ float gamma(float alpha)   // Marsaglia-Tsang, boosted for alpha < 1
    {
        const float a = alpha < 1.0f ? alpha + 1.0f : alpha;
        const float d = a - 1.0f / 3.0f, c = 1.0f / sqrtf(9.0f * d);
        float v = 1.0f, x, u;
        for (int it = 0; it < 64; ++it) {
            x = normal();
            v = 1.0f + c * x;
            if (v <= 0.0f) continue;
            v = v * v * v;
            u = uniform();
            if (u < 1.0f - 0.0331f * x * x * x * x) break;
            if (logf(u) < 0.5f * x * x + d * (1.0f - v + logf(v))) break;
        }
        float g = d * v;
        if (alpha < 1.0f) g *= powf(uniform(), 1.0f / alpha);
        return g;
    }
};

// Work counters: CNT_STRIPES copies of the CNT_N counters, one 64-byte line each; a workgroup
// adds to the copy picked by its index, the host sums the copies.  A single copy made a thousand
// wavefronts queue on three L2 atomics per launch.
__device__ __forceinline__ void wave_add_counter(unsigned long long *counters, int which, unsigned v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
    if ((threadIdx.x & 63) == 0 && v)
        atomicAdd(&counters[(blockIdx.x % CNT_STRIPES) * CNT_N + which], static_cast<unsigned long long>(v));
}

// sum over the wavefront on the DPP network (no LDS crossbar trips), the total in every lane's copy of lane 63
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
    auto mv = [](unsigned x, auto ctrl, auto rmask) {
        return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), decltype(ctrl)::value, decltype(rmask)::value, 0xf, false));
    };
    v += mv(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});     // quad_perm [1,0,3,2]
    v += mv(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});     // quad_perm [2,3,0,1]
    v += mv(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});    // row_half_mirror
    v += mv(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});    // row_mirror
    v += mv(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});    // row_bcast:15 into rows 1 and 3
    v += mv(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});    // row_bcast:31 into rows 2 and 3
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

template <int L>
__device__ __forceinline__ HotRec group_bcast(const HotRec &c, int src)
{
    HotRec r;
    r.n_visits   = __shfl(c.n_visits, src, L);
    r.n_inflight = __shfl(c.n_inflight, src, L);
    r.w_p1       = __shfl(c.w_p1, src, L);
    r.w_p2       = __shfl(c.w_p2, src, L);
    r.m_sum      = __shfl(c.m_sum, src, L);
    r.prior      = __shfl(c.prior, src, L);
    r.child_off  = __shfl(c.child_off, src, L);
    r.meta       = static_cast<uint32_t>(__shfl(static_cast<int>(c.meta), src, L));
    return r;
}

// ---- one tree per wavefront (Othello: 64 lanes, up to 33 edges): the group-wide exchanges without the LDS crossbar.
// A __shfl with a runtime lane is a ds_bpermute (address VGPR, LDS round trip, wait); in a loop bounded by the edge
// count that is one dependent round trip per edge.  With the whole wavefront as the group the source lane is
// wave-uniform, so v_readlane_b32 (a scalar result, no LDS) does it, and reductions run on the DPP network.

// sum of `term` over lanes 0..E-1 IN LANE ORDER (the reference adds in edge order: MCTS.h:145-151,343-345); lanes
// >= E must hold +0 (adding it changes nothing), so the common case is a fixed unrolled chain of 40 readlanes
__device__ __forceinline__ float wave_ordered_sum(float term, int E)
{
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 40; ++i) s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(term), i));
    for (int i = 40; i < E; ++i) s += __shfl(term, i, WAVE);               // positions with more than 40 legal moves (imported roots)
    return s;
}
// maximum of an unsigned key over the wavefront (0 = the identity of lanes switched off)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    auto mv = [](unsigned x, auto ctrl, auto rmask) {
        return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), decltype(ctrl)::value, decltype(rmask)::value, 0xf, false));
    };
    auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
    v = mx(v, mv(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{}));     // quad_perm [1,0,3,2]
    v = mx(v, mv(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{}));     // quad_perm [2,3,0,1]
    v = mx(v, mv(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{}));    // row_half_mirror
    v = mx(v, mv(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{}));    // row_mirror
    v = mx(v, mv(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}));    // row_bcast:15 into rows 1 and 3
    v = mx(v, mv(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{}));    // row_bcast:31 into rows 2 and 3
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}
// strict '>' over ascending edges == highest score, lowest lane on ties; NaN and -inf never win (MCTS.h:172,226-231):
// an order-preserving integer key of the score (0 for lanes that cannot win), its wave-wide maximum, the lowest lane
// that holds it.  Returns -1 when no lane can win.
__device__ __forceinline__ int wave_argmax(float score, bool can_win)
{
    const uint32_t b = __float_as_uint(score);
    uint32_t key = (b & 0x80000000u) ? ~b : (b | 0x80000000u);             // monotone in the float order; -inf -> 0x007fffff
    if (!can_win || !(score > -INFINITY)) key = 0u;                         // NaN fails the comparison too
    const unsigned mx = wave_max_u32(key);
    if (mx == 0u) return -1;
    const unsigned long long hit = __ballot(key == mx);
    return static_cast<int>(__builtin_ctzll(hit));
}
template <>
__device__ __forceinline__ HotRec group_bcast<WAVE>(const HotRec &c, int src)
{
    const int u = __builtin_amdgcn_readfirstlane(src);                      // the winner's lane is the same in every lane
    HotRec r;
    r.n_visits   = __builtin_amdgcn_readlane(c.n_visits, u);
    r.n_inflight = __builtin_amdgcn_readlane(c.n_inflight, u);
    r.w_p1       = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.w_p1), u));
    r.w_p2       = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.w_p2), u));
    r.m_sum      = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.m_sum), u));
    r.prior      = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.prior), u));
    r.child_off  = __builtin_amdgcn_readlane(c.child_off, u);
    r.meta       = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(c.meta), u));
    return r;
}

// first record of tree t's arena: the half it lives in now (tree_layout.h: two halves per tree, a
// re-rooting copies the kept subtree into the other one)
__device__ __forceinline__ size_t tree_base(const TreeArena &ar, int t)
{
    return (static_cast<size_t>(t) * 2 + ar.half[t]) * static_cast<size_t>(ar.S);
}

__device__ __forceinline__ HotRec empty_rec()
{
    HotRec c;
    c.n_visits = 0; c.n_inflight = 0; c.w_p1 = 0.f; c.w_p2 = 0.f; c.m_sum = 0.f;
    c.prior = 0.f; c.child_off = -1; c.meta = 0u;
    return c;
}

// ------------------------------------------------------------------ import (host entry points)

template <class G>
__global__ void __launch_bounds__(256) k_import(const int8_t *boards, const int32_t *turns, RootState rs, int B)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    GameState s;
    s.turn = turns[t];
    G::import_cells(boards + static_cast<size_t>(t) * G::CELLS, s);
    rs.bb0[t] = s.bb0; rs.bb1[t] = s.bb1; rs.turn[t] = s.turn; rs.aux[t] = s.aux;
}

// device-resident roots: the game's small integer is derived as an import would derive it
template <class G>
__global__ void __launch_bounds__(256) k_set_roots(const uint64_t *bb0, const uint64_t *bb1, const int32_t *turns,
                                                   RootState rs, int B)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    const uint64_t a = bb0[t], b = bb1[t];
    rs.bb0[t] = a; rs.bb1[t] = b; rs.turn[t] = turns[t];
    rs.aux[t] = G::root_aux(a, b);
}

// ------------------------------------------------------------------ selection

// MCTS.h:242-322 (VL=false) / 443-545 (VL=true) for K consecutive descents of every tree,
// with compute_fpu (140-156) and select_edge (163-234) evaluated across the group's lanes.
template <class G, bool VL>
__global__ void __launch_bounds__(WAVE) k_select(TreeArena ar, RootState rs, LeafBuf lf, SearchParams p, int K,
                                                 int tpw, unsigned long long *counters, uint64_t *bump, long long *zero)
{
    constexpr int L = G::LANES;
    const int lane = threadIdx.x;
    // Device generator: one new call number per iteration.  Selection draws nothing, and every
    // kernel that does (gather: symmetry ids; backup: root noise - distinct streams of one call
    // number) runs after it on the stream, so the bump rides here instead of in launches of its own.
    if (bump != nullptr && blockIdx.x == 0 && lane == 0) *bump += 1;
    if (zero != nullptr && blockIdx.x == 0 && lane == 0) *zero = 0;   // the live-leaf count of this iteration
    const int sub = lane % L;
    const int grp = lane / L;
    const int tree = blockIdx.x * tpw + grp;
    const bool live = grp < tpw && tree < ar.B;
    const int t = live ? tree : 0;
    const float tree_ne = p.noise_eps_tree != nullptr ? p.noise_eps_tree[t] : p.noise_eps;   // root-noise epsilon of this tree

    HotRec *hot = ar.hot + tree_base(ar, t);
    const ColdRec *cold = ar.cold + tree_base(ar, t);
    const int root = ar.root[t];
    HotRec rootrec = hot[root];
    int root_infl = rootrec.n_inflight;
    GameState rstate;
    rstate.bb0 = rs.bb0[t]; rstate.bb1 = rs.bb1[t]; rstate.turn = rs.turn[t]; rstate.aux = rs.aux[t];

    // state of the descent in progress (uniform across the group)
    int k = 0;
    bool done = !live;
    int cur = root, cur_lane = 0, depth = 0;
    HotRec R = rootrec;
    GameState st = rstate;
    size_t flat = static_cast<size_t>(t) * K;
    int32_t *path = lf.path + flat * G::MAX_PATH;
    if (!done && sub == 0) path[0] = root;

    unsigned n_levels = 0, n_terminal = 0;

    for (;;) {
        if (!done) {
            const uint32_t meta = R.meta;
            const int E = static_cast<int>((meta & META_NEDGE_MASK) >> META_NEDGE_SHIFT);
            bool stop = !(meta & META_EXPANDED) || (meta & META_TERMINAL) || E == 0;   // MCTS.h:250-258
            int best = -1;
            HotRec c = empty_rec();
            if (!stop) {
                const bool has = sub < E;
                const bool is_root = cur == root;
                const float ne = tree_ne;
                float noise = 0.0f;
                if (has) {
                    c = hot[R.child_off + sub];
                    if (is_root && ne > 0.0f) noise = cold[R.child_off + sub].noise;
                }
                const bool exists = has && (c.meta & META_EXISTS);
                const bool real = exists && c.n_visits > 0;

                // compute_fpu: prior mass of children with real visits, summed in edge order
                const float pq = mean_q(R.n_visits, R.w_p1, R.w_p2, (meta & META_TURN_P1) != 0);
                const float seen_term = real ? c.prior : 0.0f;
                float seen = 0.0f;
                if (L <= 8) {
#pragma unroll
                    for (int i = 0; i < L - 1; ++i) seen += __shfl(seen_term, i, L);   // lanes >= E hold 0
                } else if (L == WAVE) {
                    seen = wave_ordered_sum(seen_term, E);
                } else {
                    for (int i = 0; i < E; ++i) seen += __shfl(seen_term, i, L);
                }
                const float scale = (1.0f + pq) / 2.0f;
                const float eff = p.fpu_reduction * scale;
                float fpu = fmaf(-eff, sqrtf(seen), pq);
                fpu = (-1.0f < fpu) ? fpu : -1.0f;

                // select_edge
                const int pn_i = R.n_visits + R.n_inflight;
                const float parent_n = static_cast<float>(pn_i);
                const float parent_m = mean_m(R.n_visits, R.m_sum);
                const float c_puct = (pn_i >= 0 && pn_i < p.tab_n)
                    ? p.cpuct_tab[pn_i]
                    : p.c_init + logf((parent_n + p.c_base + 1.0f) / p.c_base);
                float eff_prior = c.prior;
                if (is_root && ne > 0.0f) eff_prior = fmaf(c.prior, 1.0f - ne, ne * noise);

                float q, child_q = 0.0f, child_m = 0.0f;
                int child_total = 0;
                if (real) {
                    child_total = c.n_visits + c.n_inflight;
                    child_q = mean_q(c.n_visits, c.w_p1, c.w_p2, (c.meta & META_TURN_P1) != 0);
                    child_m = mean_m(c.n_visits, c.m_sum);
                    q = -child_q;
                } else if (exists && c.n_inflight > 0) {
                    q = fpu;
                    child_total = c.n_inflight;
                } else {
                    q = fpu;
                }
                const float u = c_puct * eff_prior * sqrtf(parent_n) /
                                (1.0f + static_cast<float>(child_total));
                const float m_util = real ? G::aux_utility(child_m, parent_m, child_q, p) : 0.0f;
                const float score = q + u + m_util;

                // strict '>' over ascending edges == max score, lowest index on ties; NaN and
                // -inf can never win (MCTS.h:172,226-231)
                if (L == WAVE) {
                    best = wave_argmax(score, has);
                } else {
                    float s = (has && score == score) ? score : -INFINITY;
                    int si = sub;
#pragma unroll
                    for (int o = L / 2; o > 0; o >>= 1) {
                        const float os = __shfl_xor(s, o, L);
                        const int oi = __shfl_xor(si, o, L);
                        if (os > s || (os == s && oi < si)) { s = os; si = oi; }
                    }
                    best = (s > -INFINITY) ? si : -1;
                }
                if (best < 0) stop = true;
            }

            if (!stop) {
                ++n_levels;
                if (VL && depth == 0) root_infl += p.vl_count;      // MCTS.h:470-475
                const int action = L == WAVE
                    ? __builtin_amdgcn_readlane(static_cast<int>(c.meta & META_ACTION_MASK), __builtin_amdgcn_readfirstlane(best))
                    : __shfl(static_cast<int>(c.meta & META_ACTION_MASK), best, L);
                G::step(st, action);
                const int res = G::result(st);
                const int child_slot = R.child_off + best;
                if (sub == best) {
                    uint32_t nm = c.meta;
                    if (!(nm & META_EXISTS))                          // lazy child, MCTS.h:268-275
                        nm = (nm & ~META_TURN_P1) | META_EXISTS | (st.turn == 1 ? META_TURN_P1 : 0u);
                    if (res >= 0)                                     // MCTS.h:279-288
                        nm = (nm & ~META_RESULT_MASK) | META_TERMINAL |
                             (static_cast<uint32_t>(res) << META_RESULT_SHIFT);
                    if (VL) {                                         // MCTS.h:492
                        c.n_inflight += p.vl_count;
                        hot[child_slot].n_inflight = c.n_inflight;
                    }
                    if (nm != c.meta) { c.meta = nm; hot[child_slot].meta = nm; }
                }
                R = group_bcast<L>(c, best);
                cur = child_slot;
                cur_lane = best;
                ++depth;
                if (sub == 0) path[depth] = cur;
            } else {
                // leaf reached: MCTS.h:291-321 / 512-544
                uint32_t lm = R.meta;
                bool term = (lm & META_TERMINAL) != 0;
                int code = static_cast<int>((lm & META_RESULT_MASK) >> META_RESULT_SHIFT);
                if (!term) {
                    const int res = G::result(st);
                    if (res >= 0) {
                        term = true; code = res;
                        lm = (lm & ~META_RESULT_MASK) | META_TERMINAL |
                             (static_cast<uint32_t>(res) << META_RESULT_SHIFT);
                        if (sub == cur_lane) hot[cur].meta = lm;
                        if (depth == 0) rootrec.meta = lm;
                    }
                }
                if (term) ++n_terminal;
                uint8_t fl = static_cast<uint8_t>((term ? LEAF_TERMINAL : 0) | (code << LEAF_RESULT_SHIFT));
                if (VL && depth > 0) fl |= LEAF_VL_APPLIED;
                if (depth == 0 && !(lm & META_EXPANDED)) fl |= LEAF_ROOT_UNEXPANDED;
                if (lm & META_EXPANDED) fl |= LEAF_EXPANDED;
                const int nv = term ? 0 : G::num_valid(st);
                if (sub == 0) lf.slot[flat] = cur;
                if (sub == 1) lf.bb0[flat] = st.bb0;
                if (sub == 2) lf.bb1[flat] = st.bb1;
                if (sub == 3) lf.turn[flat] = st.turn;
                if (sub == 4) lf.flags[flat] = fl;
                if (sub == 5) lf.path_len[flat] = depth + 1;
                if (sub == 6) lf.aux[flat] = st.aux;
                if (sub == 7) lf.nvalid[flat] = static_cast<uint8_t>(nv);

                ++k;
                if (k == K) {
                    done = true;
                } else {
                    ++flat;
                    path += G::MAX_PATH;
                    cur = root; cur_lane = 0; depth = 0;
                    R = rootrec; R.n_inflight = root_infl;
                    st = rstate;
                    if (sub == 0) path[0] = root;
                }
            }
        }
        if (__all(done)) break;
    }

    if (live && VL && sub == 0 && root_infl != rootrec.n_inflight) hot[root].n_inflight = root_infl;

    wave_add_counter(counters, CNT_LEVELS, sub == 0 ? n_levels : 0u);
    wave_add_counter(counters, CNT_TERMINAL, sub == 0 ? n_terminal : 0u);
    wave_add_counter(counters, CNT_SIMS, (live && sub == 0) ? static_cast<unsigned>(K) : 0u);
}

// ------------------------------------------------------------------ selection, Connect4-shaped groups of 8 lanes
//
// k_select run by ONE wavefront per SIMD issues every instruction - vector or scalar - at 4 cycles, so
// a level costs (instructions x 4) cycles plus its waits, and a launch ends when the wavefront with the
// deepest trees is done.  k_select's level body is ~400 instructions and its leaf path another ~430;
// the eight trees of a wavefront are out of step, so nearly every trip through its loop pays BOTH.
// This kernel computes the same search (bit-identical: tests) with the instruction stream cut down:
//   * a trip = one level, then - only for groups that just arrived at a leaf - a short emit; no trips
//     spent on leaves alone (levels instead of levels + K trips per tree);
//   * cross-lane traffic inside a group is DPP on the vector ALU (quad_perm / row_half_mirror compose
//     every 8-lane exchange): the ordered 7-term prior sum and the (score, lowest index) arg-max no
//     longer take five dependent trips through the LDS crossbar; the arg-max runs on an
//     order-preserving integer key and ends in one ballot;
//   * what the winner's lane holds is fetched in ONE batch of seven independent ds_bpermute; the new
//     node's flags are then computed by every lane alike instead of being computed by one and re-sent;
//   * predicated single-lane stores (a branch each) are gathered into one block per level and one per
//     leaf; the first 16 path entries ride in two registers per lane (lane j keeps depths j, j+8) and
//     leave with two unconditional stores per leaf (entries past the path's end are ignored downstream);
// (Touching the grandchildren's blocks while a level's arithmetic runs was tried and was slower: the loads
// mostly hit L2 already and the touches only add instructions.)
constexpr int DPP_QP0 = 0x00, DPP_QP1 = 0x55, DPP_QP2 = 0xAA, DPP_QP3 = 0xFF;     // quad_perm broadcasts of lane 0..3
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_QREV = 0x1B;                  // quad_perm [1,0,3,2] [2,3,0,1] [3,2,1,0]
constexpr int DPP_HALF_MIRROR = 0x141;                                            // lane i <- lane 7 - i in each 8

template <int CTRL, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, 0xf, BANK_MASK, false));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t x)
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(x), static_cast<int>(x), CTRL, 0xf, 0xf, false));
}

// ((((((0 + t0) + t1) + t2) + t3) + t4) + t5) + t6 of the group's lanes 0..6, in every lane of the group:
// the order of `for (i < 7) sum += shfl(t, i)`, i.e. of the reference's loop over edges.  Lanes 0-3 add
// their quad's four terms, the running sum crosses to lanes 4-7 mirrored, they add theirs, and the total
// crosses back into the banks of lanes 0-3.
__device__ __forceinline__ float group8_ordered_sum7(float t)
{
    float a = 0.0f + dpp_f<DPP_QP0>(t, t);
    a = a + dpp_f<DPP_QP1>(t, t);
    a = a + dpp_f<DPP_QP2>(t, t);
    a = a + dpp_f<DPP_QP3>(t, t);
    const float x = dpp_f<DPP_HALF_MIRROR>(a, a);
    float s = x + dpp_f<DPP_QP0>(t, t);
    s = s + dpp_f<DPP_QP1>(t, t);
    s = s + dpp_f<DPP_QP2>(t, t);
    return dpp_f<DPP_HALF_MIRROR, 0x5>(s, s);          // banks 0 and 2 (lanes 0-3 of each group) take lanes 7-4's total
}

// Lane of the group's largest score, the lowest one on ties; -1 if no lane is `valid` or every valid
// score is -inf (MCTS.h:172,226-231: strict '>' over ascending edges, from -inf).
__device__ __forceinline__ int group8_argmax(float score, bool valid, int lane)
{
    const float s = valid ? score + 0.0f : -INFINITY;  // -0 -> +0: the two compare equal in the reference
    const uint32_t b = __float_as_uint(s);
    const uint32_t key = b ^ (static_cast<uint32_t>(static_cast<int32_t>(b) >> 31) | 0x80000000u);   // order-preserving
    uint32_t m = max(key, dpp_u<DPP_XOR1>(key));
    m = max(m, dpp_u<DPP_XOR2>(m));
    m = max(m, dpp_u<DPP_QREV>(dpp_u<DPP_HALF_MIRROR>(m)));                                         // lane ^ 4
    const unsigned long long bal = __ballot(key == m && m != 0x007FFFFFu);                          // 0x007FFFFF = key(-inf)
    const uint32_t g = static_cast<uint32_t>(bal >> (lane & 56)) & 0xffu;
    return g ? __ffs(g) - 1 : -1;
}

template <bool VL>
__global__ void __launch_bounds__(WAVE) k_select8(TreeArena ar, RootState rs, LeafBuf lf, SearchParams p, int K,
                                                  int tpw, unsigned long long *counters, uint64_t *bump, long long *zero)
{
    using G = Connect4Dev;
    constexpr int L = 8;
    const int lane = threadIdx.x;
    if (bump != nullptr && blockIdx.x == 0 && lane == 0) *bump += 1;      // see k_select
    if (zero != nullptr && blockIdx.x == 0 && lane == 0) *zero = 0;
    const int sub = lane % L;
    const int grp = lane / L;
    const int tree = blockIdx.x * tpw + grp;
    const bool live = grp < tpw && tree < ar.B;
    const int t = live ? tree : 0;
    const float ne = p.noise_eps_tree != nullptr ? p.noise_eps_tree[t] : p.noise_eps;
    const bool root_mix = ne > 0.0f;

    HotRec *hot = ar.hot + tree_base(ar, t);
    const ColdRec *cold = ar.cold + tree_base(ar, t);
    const int root = ar.root[t];
    HotRec rootrec = hot[root];
    int root_infl = rootrec.n_inflight;
    GameState rstate;
    rstate.bb0 = rs.bb0[t]; rstate.bb1 = rs.bb1[t]; rstate.turn = rs.turn[t]; rstate.aux = rs.aux[t];

    int k = 0;
    bool done = !live;
    int cur = root, cur_lane = 0, depth = 0;
    HotRec R = rootrec;
    // the root's own means (MCTSNode.h:116-133): selection changes neither visits nor sums, so once per launch
    const float root_q = mean_q(rootrec.n_visits, rootrec.w_p1, rootrec.w_p2, (rootrec.meta & META_TURN_P1) != 0);
    const float root_m = mean_m(rootrec.n_visits, rootrec.m_sum);
    float Rq = root_q, Rm = root_m;
    GameState st = rstate;
    size_t flat = static_cast<size_t>(t) * K;
    int path0 = root, path1 = 0;                      // this lane's path entries: depths sub and sub + 8
    unsigned n_levels = 0, n_terminal = 0;

    auto is_leaf = [](uint32_t meta) {                // MCTS.h:250-258
        return !(meta & META_EXPANDED) || (meta & META_TERMINAL) || (meta & META_NEDGE_MASK) == 0;
    };
    // MCTS.h:291-321 / 512-544: what a finished descent leaves behind, then the next descent starts at the root
    auto emit = [&]() {
        uint32_t lm = R.meta;
        bool term = (lm & META_TERMINAL) != 0;
        int code = static_cast<int>((lm & META_RESULT_MASK) >> META_RESULT_SHIFT);
        if (depth == 0 && !term) {                    // a node entered by a move carries its result already (MCTS.h:279-288)
            const int res = G::result(st);
            if (res >= 0) {
                term = true; code = res;
                lm = (lm & ~META_RESULT_MASK) | META_TERMINAL | (static_cast<uint32_t>(res) << META_RESULT_SHIFT);
                if (sub == cur_lane) hot[cur].meta = lm;
                rootrec.meta = lm;
            }
        }
        if (term) ++n_terminal;
        uint8_t fl = static_cast<uint8_t>((term ? LEAF_TERMINAL : 0) | (code << LEAF_RESULT_SHIFT));
        if (VL && depth > 0) fl |= LEAF_VL_APPLIED;
        if (depth == 0 && !(lm & META_EXPANDED)) fl |= LEAF_ROOT_UNEXPANDED;
        if (lm & META_EXPANDED) fl |= LEAF_EXPANDED;
        constexpr uint64_t TOP = 0x0000810204081020ull;               // the top cell of every column
        const int nv = term ? 0 : 7 - static_cast<int>(__builtin_popcountll((st.bb0 | st.bb1) & TOP));
        int32_t *path = lf.path + flat * G::MAX_PATH;
        if (sub <= depth) path[sub] = path0;                           // entries past depth are ignored downstream: not written
        if (sub + 8 <= depth) path[sub + 8] = path1;                   // (they were 2 MB of stores per launch that nobody reads)
        if (sub == 0) {
            lf.slot[flat] = cur; lf.bb0[flat] = st.bb0; lf.bb1[flat] = st.bb1; lf.turn[flat] = st.turn;
            lf.flags[flat] = fl; lf.path_len[flat] = depth + 1; lf.aux[flat] = st.aux;
            lf.nvalid[flat] = static_cast<uint8_t>(nv);
        }
        ++k;
        if (k == K) {
            done = true;
        } else {
            ++flat;
            cur = root; cur_lane = 0; depth = 0;
            R = rootrec; R.n_inflight = root_infl;
            Rq = root_q; Rm = root_m;
            st = rstate;
            path0 = root;                                              // depth 0 in lane 0; the others are overwritten on the way
        }
    };

    while (!done && is_leaf(R.meta)) emit();          // a root that is a leaf ends all K descents where they start

    for (;;) {
        if (!done) {
            // ---- one level from the inner node R (MCTS.h:140-234 across the lanes)
            const uint32_t meta = R.meta;
            const int E = static_cast<int>((meta & META_NEDGE_MASK) >> META_NEDGE_SHIFT);
            const bool has = sub < E;
            const bool is_root = cur == root;
            HotRec c = hot[R.child_off + (has ? sub : 0)];
            float noise = 0.0f;
            if (is_root && root_mix && has) noise = cold[R.child_off + sub].noise;
            if (!has) { c.meta = 0u; c.prior = 0.0f; c.child_off = -1; }
            const bool exists = (c.meta & META_EXISTS) != 0;
            const bool real = exists && c.n_visits > 0;

            const float pq = Rq;                      // mean_q / mean_m of R: computed when R was a candidate one level up
            const float seen = group8_ordered_sum7(real ? c.prior : 0.0f);
            const float scale = (1.0f + pq) / 2.0f;
            const float eff = p.fpu_reduction * scale;
            float fpu = fmaf(-eff, sqrtf(seen), pq);
            fpu = (-1.0f < fpu) ? fpu : -1.0f;

            const int pn_i = R.n_visits + R.n_inflight;
            const float parent_n = static_cast<float>(pn_i);
            const float parent_m = Rm;
            // both table entries by one unconditional pair of loads (index 0 outside the table, then the formulas)
            const bool in_tab = static_cast<unsigned>(pn_i) < static_cast<unsigned>(p.tab_n);
            const float *tab = p.cpuct_tab + (in_tab ? pn_i : 0);
            float c_puct = tab[0], sqrt_pn = tab[p.tab_n];
            if (!in_tab) {
                c_puct = p.c_init + logf((parent_n + p.c_base + 1.0f) / p.c_base);
                sqrt_pn = sqrtf(parent_n);
            }
            float eff_prior = c.prior;
            if (is_root && root_mix) eff_prior = fmaf(c.prior, 1.0f - ne, ne * noise);

            float q = fpu, child_q = 0.0f, child_m = 0.0f;
            int child_total = (exists && c.n_inflight > 0) ? c.n_inflight : 0;
            if (real) {
                child_total = c.n_visits + c.n_inflight;
                child_q = mean_q(c.n_visits, c.w_p1, c.w_p2, (c.meta & META_TURN_P1) != 0);
                child_m = mean_m(c.n_visits, c.m_sum);
                q = -child_q;
            }
            const float u = c_puct * eff_prior * sqrt_pn / (1.0f + static_cast<float>(child_total));
            const float m_util = real ? G::aux_utility(child_m, parent_m, child_q, p) : 0.0f;
            const float score = q + u + m_util;
            const int best = group8_argmax(score, has && score == score, lane);

            if (best >= 0) {
                ++n_levels;
                if (VL && depth == 0) root_infl += p.vl_count;                 // MCTS.h:470-475
                // the winner's record, one batch of independent exchanges
                const int src = (lane & 56) + best;
                const uint32_t bmeta = static_cast<uint32_t>(__shfl(static_cast<int>(c.meta), src));
                const int b_off = __shfl(c.child_off, src);
                const int b_n = __shfl(c.n_visits, src);
                const int b_infl = __shfl(c.n_inflight, src);
                const float b_w1 = __shfl(c.w_p1, src);
                const float b_w2 = __shfl(c.w_p2, src);
                const float b_ms = __shfl(c.m_sum, src);
                Rq = __shfl(child_q, src);             // 0 without real visits, as mean_q / mean_m of such a node are
                Rm = __shfl(child_m, src);
                G::step(st, static_cast<int>(bmeta & META_ACTION_MASK));
                const int res = G::result(st);
                uint32_t nm = bmeta;
                if (!(nm & META_EXISTS))                                       // lazy child, MCTS.h:268-275
                    nm = (nm & ~META_TURN_P1) | META_EXISTS | (st.turn == 1 ? META_TURN_P1 : 0u);
                if (res >= 0)                                                  // MCTS.h:279-288
                    nm = (nm & ~META_RESULT_MASK) | META_TERMINAL | (static_cast<uint32_t>(res) << META_RESULT_SHIFT);
                const int n_infl = VL ? b_infl + p.vl_count : b_infl;          // MCTS.h:492
                const int child_slot = R.child_off + best;
                if (sub == best) {
                    if (VL) hot[child_slot].n_inflight = n_infl;
                    if (nm != bmeta) hot[child_slot].meta = nm;
                }
                R.n_visits = b_n; R.n_inflight = n_infl; R.w_p1 = b_w1; R.w_p2 = b_w2; R.m_sum = b_ms;
                R.child_off = b_off; R.meta = nm;
                cur = child_slot;
                cur_lane = best;
                ++depth;
                if (depth < 8) { if (sub == depth) path0 = cur; }
                else if (depth < 16) { if (sub == depth - 8) path1 = cur; }
                else if (sub == 0) lf.path[flat * G::MAX_PATH + depth] = cur;
            }
            // arrived at a leaf - or no edge can be chosen (all scores NaN / -inf): the node itself is the leaf
            if (best < 0 || is_leaf(R.meta)) emit();
        }
        if (__all(done)) break;
    }
    if (live && VL && sub == 0 && root_infl != rootrec.n_inflight) hot[root].n_inflight = root_infl;

    wave_add_counter(counters, CNT_LEVELS, sub == 0 ? n_levels : 0u);
    wave_add_counter(counters, CNT_TERMINAL, sub == 0 ? n_terminal : 0u);
    wave_add_counter(counters, CNT_SIMS, (live && sub == 0) ? static_cast<unsigned>(K) : 0u);
}

// The K <= 4 virtual-loss descents of a tree SIDE BY SIDE in one wavefront: descent j lives in its own
// group of 8 lanes (a tree takes 32 lanes, a wavefront holds two trees) and starts j steps after
// descent 0.  Why they may run one level apart: descent j + 1 meets
// descent j only through what j leaves on a node when it ARRIVES there (in-flight visits, the EXISTS /
// TERMINAL bits), and j arrives one step before j + 1 reads that node among its parent's children; two
// descents of a tree are never on the same level in the same step, so they never write the same record
// in the same step.  Here the descents of a step execute as lanes of the SAME instructions, so a step
// costs one level's instructions whatever K is, and a tree needs (deepest descent + K - 1) steps instead
// of the sum of its descents' depths - which is what a launch waits for: its deepest trees.  A store of
// step s is read by another lane of the same wavefront in step s + 1: vector memory operations of one
// wavefront reach its CU's L1 in program order (wavefront scope needs no cache action in the AMDGPU
// memory model); the fences below only keep the compiler from moving them.  Results are bit-identical
// to k_select / k_select8 (tests).  Four wavefronts per SIMD at 8192 trees.
__global__ void __launch_bounds__(WAVE) k_select8x4(TreeArena ar, RootState rs, LeafBuf lf, SearchParams p, int K,
                                                    unsigned long long *counters, uint64_t *bump, long long *zero)
{
    using G = Connect4Dev;
    const int lane = threadIdx.x;
    if (bump != nullptr && blockIdx.x == 0 && lane == 0) *bump += 1;      // see k_select
    if (zero != nullptr && blockIdx.x == 0 && lane == 0) *zero = 0;
    const int sub = lane & 7;
    const int j = (lane >> 3) & 3;                    // which descent of its tree this group runs
    const int tree = blockIdx.x * 2 + (lane >> 5);
    const bool live = tree < ar.B && j < K;
    const int t = tree < ar.B ? tree : 0;
    const float ne = p.noise_eps_tree != nullptr ? p.noise_eps_tree[t] : p.noise_eps;
    const bool root_mix = ne > 0.0f;

    HotRec *hot = ar.hot + tree_base(ar, t);
    const ColdRec *cold = ar.cold + tree_base(ar, t);
    const int root = ar.root[t];
    const HotRec rootrec = hot[root];
    GameState st;
    st.bb0 = rs.bb0[t]; st.bb1 = rs.bb1[t]; st.turn = rs.turn[t]; st.aux = rs.aux[t];

    bool done = !live;
    bool passed_root = false;                         // this descent left the root with a chosen edge (MCTS.h:470-475)
    int cur = root, cur_lane = 0, depth = 0;
    HotRec R = rootrec;
    float Rq = mean_q(rootrec.n_visits, rootrec.w_p1, rootrec.w_p2, (rootrec.meta & META_TURN_P1) != 0);   // see k_select8
    float Rm = mean_m(rootrec.n_visits, rootrec.m_sum);
    const size_t flat = static_cast<size_t>(t) * K + (j < K ? j : 0);
    int path0 = root, path1 = 0;                      // this lane's path entries: depths sub and sub + 8
    unsigned n_levels = 0, n_terminal = 0;
    HotRec cpre = rootrec;                            // lane's record of R's children block, when have_pre
    bool have_pre = false;

    auto is_leaf = [](uint32_t meta) {                // MCTS.h:250-258
        return !(meta & META_EXPANDED) || (meta & META_TERMINAL) || (meta & META_NEDGE_MASK) == 0;
    };
    auto emit = [&]() {                               // MCTS.h:512-544
        uint32_t lm = R.meta;
        bool term = (lm & META_TERMINAL) != 0;
        int code = static_cast<int>((lm & META_RESULT_MASK) >> META_RESULT_SHIFT);
        if (depth == 0 && !term) {                    // first-time terminal test of a root (MCTS.h:299-319); every descent
            const int res = G::result(st);            // of the tree finds the same, the first one records it
            if (res >= 0) {
                term = true; code = res;
                lm = (lm & ~META_RESULT_MASK) | META_TERMINAL | (static_cast<uint32_t>(res) << META_RESULT_SHIFT);
                if (j == 0 && sub == cur_lane) hot[cur].meta = lm;
            }
        }
        if (term) ++n_terminal;
        uint8_t fl = static_cast<uint8_t>((term ? LEAF_TERMINAL : 0) | (code << LEAF_RESULT_SHIFT));
        if (depth > 0) fl |= LEAF_VL_APPLIED;
        if (depth == 0 && !(lm & META_EXPANDED)) fl |= LEAF_ROOT_UNEXPANDED;
        if (lm & META_EXPANDED) fl |= LEAF_EXPANDED;
        constexpr uint64_t TOP = 0x0000810204081020ull;               // the top cell of every column
        const int nv = term ? 0 : 7 - static_cast<int>(__builtin_popcountll((st.bb0 | st.bb1) & TOP));
        int32_t *path = lf.path + flat * G::MAX_PATH;
        if (sub <= depth) path[sub] = path0;                           // entries past depth are ignored downstream: not written
        if (sub + 8 <= depth) path[sub + 8] = path1;                   // (they were 2 MB of stores per launch that nobody reads)
        if (sub == 0) {
            lf.slot[flat] = cur; lf.bb0[flat] = st.bb0; lf.bb1[flat] = st.bb1; lf.turn[flat] = st.turn;
            lf.flags[flat] = fl; lf.path_len[flat] = depth + 1; lf.aux[flat] = st.aux;
            lf.nvalid[flat] = static_cast<uint8_t>(nv);
        }
    };
    // A group runs ONE descent, and what it found stays in its registers once it is done: the leaf is written
    // out after the loop, once per wavefront, instead of in every step in which some group arrives somewhere.

    if (!done && is_leaf(R.meta)) done = true;        // a root that is a leaf: every descent ends where it starts

    for (int step = 0;; ++step) {
        // descents of this tree that have left the root already, below this one (their in-flight visits are on it)
        const unsigned long long pb = __ballot(passed_root);
        const unsigned tree_groups = static_cast<unsigned>(pb >> (lane & 32)) & 0x01010101u;
        const bool act = !done && step >= j;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (act) {
            if (depth == 0)
                R.n_inflight = rootrec.n_inflight + p.vl_count * static_cast<int>(__builtin_popcount(tree_groups & ((1u << (8 * j)) - 1u)));
            const uint32_t meta = R.meta;
            const int E = static_cast<int>((meta & META_NEDGE_MASK) >> META_NEDGE_SHIFT);
            const bool has = sub < E;
            const bool is_root = depth == 0;
            HotRec c = cpre;                          // requested while the previous step finished (below) ...
            if (!have_pre) c = hot[R.child_off + (has ? sub : 0)];    // ... except at the root
            float noise = 0.0f;
            if (is_root && root_mix && has) noise = cold[R.child_off + sub].noise;
            if (!has) { c.meta = 0u; c.prior = 0.0f; c.child_off = -1; }
            const bool exists = (c.meta & META_EXISTS) != 0;
            const bool real = exists && c.n_visits > 0;

            const float pq = Rq;                      // mean_q / mean_m of R: computed when R was a candidate one level up
            const float seen = group8_ordered_sum7(real ? c.prior : 0.0f);
            const float scale = (1.0f + pq) / 2.0f;
            const float eff = p.fpu_reduction * scale;
            float fpu = fmaf(-eff, sqrtf(seen), pq);
            fpu = (-1.0f < fpu) ? fpu : -1.0f;

            const int pn_i = R.n_visits + R.n_inflight;
            const float parent_n = static_cast<float>(pn_i);
            const float parent_m = Rm;
            // both table entries by one unconditional pair of loads (index 0 outside the table, then the formulas)
            const bool in_tab = static_cast<unsigned>(pn_i) < static_cast<unsigned>(p.tab_n);
            const float *tab = p.cpuct_tab + (in_tab ? pn_i : 0);
            float c_puct = tab[0], sqrt_pn = tab[p.tab_n];
            if (!in_tab) {
                c_puct = p.c_init + logf((parent_n + p.c_base + 1.0f) / p.c_base);
                sqrt_pn = sqrtf(parent_n);
            }
            float eff_prior = c.prior;
            if (is_root && root_mix) eff_prior = fmaf(c.prior, 1.0f - ne, ne * noise);

            float q = fpu, child_q = 0.0f, child_m = 0.0f;
            int child_total = (exists && c.n_inflight > 0) ? c.n_inflight : 0;
            if (real) {
                child_total = c.n_visits + c.n_inflight;
                child_q = mean_q(c.n_visits, c.w_p1, c.w_p2, (c.meta & META_TURN_P1) != 0);
                child_m = mean_m(c.n_visits, c.m_sum);
                q = -child_q;
            }
            const float u = c_puct * eff_prior * sqrt_pn / (1.0f + static_cast<float>(child_total));
            const float m_util = real ? G::aux_utility(child_m, parent_m, child_q, p) : 0.0f;
            const float score = q + u + m_util;
            const int best = group8_argmax(score, has && score == score, lane);

            if (best >= 0) {
                ++n_levels;
                if (depth == 0) passed_root = true;                            // MCTS.h:470-475
                const int src = (lane & 56) + best;
                const uint32_t bmeta = static_cast<uint32_t>(__shfl(static_cast<int>(c.meta), src));
                const int b_off = __shfl(c.child_off, src);
                const int b_n = __shfl(c.n_visits, src);
                const int b_infl = __shfl(c.n_inflight, src);
                const float b_w1 = __shfl(c.w_p1, src);
                const float b_w2 = __shfl(c.w_p2, src);
                const float b_ms = __shfl(c.m_sum, src);
                Rq = __shfl(child_q, src);             // 0 without real visits, as mean_q / mean_m of such a node are
                Rm = __shfl(child_m, src);
                // What the tree's next descent must find on this node when it scores it one step from now goes
                // out first: the in-flight visits (MCTS.h:492) and, for a lazy child, EXISTS + the side to move
                // (MCTS.h:268-275; a Connect4 move always hands the turn over).  Then the node's own children are
                // requested - an expanded node is never terminal, so the descent does go on there - and the move,
                // the four-in-a-row test and the path bookkeeping below run while that load is in flight.
                const int n_infl = b_infl + p.vl_count;
                const int child_slot = R.child_off + best;
                uint32_t nm = bmeta;
                if (!(nm & META_EXISTS)) nm = (nm & ~META_TURN_P1) | META_EXISTS | (st.turn == -1 ? META_TURN_P1 : 0u);
                if (sub == best) {
                    hot[child_slot].n_inflight = n_infl;
                    if (nm != bmeta) hot[child_slot].meta = nm;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // compiler only: the stores stay above the load
                have_pre = (bmeta & META_EXPANDED) != 0 && (bmeta & META_NEDGE_MASK) != 0;
                if (have_pre) {
                    const int e_next = static_cast<int>((bmeta & META_NEDGE_MASK) >> META_NEDGE_SHIFT);
                    cpre = hot[b_off + (sub < e_next ? sub : 0)];
                }
                G::step(st, static_cast<int>(bmeta & META_ACTION_MASK));
                const int res = G::result(st);
                if (res >= 0) {                                                // MCTS.h:279-288
                    const uint32_t tm = (nm & ~META_RESULT_MASK) | META_TERMINAL | (static_cast<uint32_t>(res) << META_RESULT_SHIFT);
                    if (sub == best && tm != nm) hot[child_slot].meta = tm;
                    nm = tm;
                }
                R.n_visits = b_n; R.n_inflight = n_infl; R.w_p1 = b_w1; R.w_p2 = b_w2; R.m_sum = b_ms;
                R.child_off = b_off; R.meta = nm;
                cur = child_slot;
                cur_lane = best;
                ++depth;
                if (depth < 8) { if (sub == depth) path0 = cur; }
                else if (depth < 16) { if (sub == depth - 8) path1 = cur; }
                else if (sub == 0) lf.path[flat * G::MAX_PATH + depth] = cur;
            }
            if (best < 0 || is_leaf(R.meta)) done = true;                      // the node itself is the leaf (MCTS.h:250-258)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (__all(done)) break;
    }
    if (live) emit();

    // in-flight visits of the root: one per descent that left it (MCTS.h:470-475)
    {
        const unsigned long long pb = __ballot(passed_root);
        const unsigned tree_groups = static_cast<unsigned>(pb >> (lane & 32)) & 0x01010101u;
        const int add = p.vl_count * static_cast<int>(__builtin_popcount(tree_groups));
        if (tree < ar.B && j == 0 && sub == 0 && add != 0) hot[root].n_inflight = rootrec.n_inflight + add;
    }

    // the three work counters in one reduction: levels of a group < 2^12, one terminal flag and one simulation each
    // (8 groups per wavefront: 8 x C4_MAX_PATH levels in 16 bits, <= 8 terminal leaves and simulations in 8 bits each;
    // readlane(63) needs every lane of the wavefront here: no path of this kernel may return before this point)
    static_assert(8 * C4_MAX_PATH < 65536, "the level counter of a wavefront is a 16-bit field");
    {
        const unsigned packed = sub == 0 ? (n_levels | (n_terminal << 16) | ((live ? 1u : 0u) << 24)) : 0u;
        const unsigned tot = wave_sum_u32(packed);
        if (lane == 0) {
            unsigned long long *c = counters + (blockIdx.x % CNT_STRIPES) * CNT_N;
            if (tot & 0xffffu) atomicAdd(&c[CNT_LEVELS], static_cast<unsigned long long>(tot & 0xffffu));
            if ((tot >> 16) & 0xffu) atomicAdd(&c[CNT_TERMINAL], static_cast<unsigned long long>((tot >> 16) & 0xffu));
            if (tot >> 24) atomicAdd(&c[CNT_SIMS], static_cast<unsigned long long>(tot >> 24));
        }
    }
}

// ------------------------------------------------------------------ virtual-loss removal

// MCTS.h:561-581: every node of a recorded path (root included) loses vl_count in-flight
// visits once; clearing the flag makes the call idempotent.  Lane j handles depth j mod LANES.
template <class G>
__device__ __forceinline__ void remove_vl_of_tree(HotRec *hot, LeafBuf lf, size_t flat0, int K, int vl, int sub)
{
    for (int k = 0; k < K; ++k) {
        const size_t flat = flat0 + k;
        const uint8_t fl = lf.flags[flat];
        if (fl & LEAF_VL_APPLIED) {
            const int len = lf.path_len[flat];
            const int32_t *path = lf.path + flat * G::MAX_PATH;
            for (int j = sub; j < len; j += G::LANES) hot[path[j]].n_inflight -= vl;
            if (sub == 0) lf.flags[flat] = fl & static_cast<uint8_t>(~LEAF_VL_APPLIED);
        }
    }
}

template <class G>
__global__ void __launch_bounds__(WAVE) k_remove_vl(TreeArena ar, LeafBuf lf, SearchParams p, int K, int strideK)
{
    constexpr int L = G::LANES;
    const int sub = threadIdx.x % L;
    const int tree = blockIdx.x * (WAVE / L) + threadIdx.x / L;
    if (tree >= ar.B) return;
    remove_vl_of_tree<G>(ar.hot + tree_base(ar, tree), lf, static_cast<size_t>(tree) * strideK, K,
                         p.vl_count, sub);
}

// ------------------------------------------------------------------ expansion + backup

// BatchedMCTS.h:176-199 (VL=false) / 296-332 (VL=true): per tree remove_all_vl, then for
// k = 0..K-1 expand_leaf (MCTS.h:329-375) and propagate (MCTS.h:381-402).
// FUSED=true takes the evaluator's raw outputs (relative WDL) and the leaf's own flags, i.e.
// it also does what MCTS_cpp.py:275-297 does between the two native calls.
template <class G, bool VL, bool FUSED>
__global__ void __launch_bounds__(WAVE) k_backprop(TreeArena ar, LeafBuf lf, SearchParams p, int K, int tpw,
                                                   EvalIn in, unsigned long long *counters, int *err)
{
    constexpr int L = G::LANES;
    constexpr int A = G::ACTIONS;
    const int lane = threadIdx.x;
    const int sub = lane % L;
    const int grp = lane / L;
    const int tree = blockIdx.x * tpw + grp;
    const bool live = grp < tpw && tree < ar.B;
    const int t = live ? tree : 0;
    HotRec *hot = ar.hot + tree_base(ar, t);
    ColdRec *cold = ar.cold + tree_base(ar, t);
    const size_t flat0 = static_cast<size_t>(t) * K;
    unsigned n_exp = 0, n_dup = 0, n_backup = 0;

    if (live) {
        if (VL) remove_vl_of_tree<G>(hot, lf, flat0, K, p.vl_count, sub);

        int used = ar.used[t];
        const int used0 = used;
        for (int k = 0; k < K; ++k) {
            const size_t flat = flat0 + k;
            const int len = lf.path_len[flat];
            if (len <= 0) continue;                                 // MCTS.h:409,599
            const int leaf = lf.slot[flat];
            GameState ls;
            ls.bb0 = lf.bb0[flat]; ls.bb1 = lf.bb1[flat]; ls.turn = lf.turn[flat]; ls.aux = lf.aux[flat];
            const uint8_t lflags = lf.flags[flat];
            const int32_t *path = lf.path + flat * G::MAX_PATH;
            const int owner = (len - 1) % L;

            bool term;
            float wd, w1, w2, ml;
            if (FUSED) {
                term = (lflags & LEAF_TERMINAL) != 0;
                if (term) {                                         // MCTS_cpp.py:275-282
                    const int code = (lflags >> LEAF_RESULT_SHIFT) & 3;
                    wd = code == 0 ? 1.0f : 0.0f; w1 = code == 1 ? 1.0f : 0.0f; w2 = code == 2 ? 1.0f : 0.0f;
                    ml = 0.0f;
                } else {                                            // MCTS_cpp.py:23-30
                    const float *r = in.wdl_rel + flat * 3;
                    wd = r[0];
                    w1 = (ls.turn == 1) ? r[1] : r[2];
                    w2 = (ls.turn == 1) ? r[2] : r[1];
                    ml = in.moves_left[flat];
                }
            } else {
                term = in.is_term[flat] != 0;
                wd = in.d[flat]; w1 = in.p1w[flat]; w2 = in.p2w[flat];
                ml = in.moves_left[flat];
            }
            if (term) ml = G::terminal_aux(ls, p);                  // MCTS.h:412,608

            if (!term) {
                // is_expanded must be the CURRENT value (an earlier k of this call may have
                // expanded the same leaf, MCTS.h:601-607): read by the lane that writes it.
                uint32_t lm = 0;
                if (sub == owner) lm = hot[leaf].meta;
                lm = static_cast<uint32_t>(__shfl(static_cast<int>(lm), owner, L));
                if (VL && (lm & META_EXPANDED)) {
                    ++n_dup;
                } else {
                    // expand_leaf: legal moves in edge order
                    const int nv = G::num_valid(ls);
                    const int my_action = sub < nv ? G::nth_valid(ls, sub) : -1;
                    const int s = in.sym ? in.sym[flat] : lf.sym[flat];
                    float my_pol = 0.0f;
                    if (my_action >= 0) my_pol = in.policy[flat * A + G::policy_index(s, my_action)];
                    float psum = 0.0f;
                    if (L <= 8) {
#pragma unroll
                        for (int i = 0; i < L - 1; ++i) psum += __shfl(my_pol, i, L);
                    } else {
                        if (L == WAVE) psum = wave_ordered_sum(my_pol, nv); else for (int i = 0; i < nv; ++i) psum += __shfl(my_pol, i, L);
                    }
                    const float prior = my_pol / (psum + 1e-8f);    // MCTS.h:370
                    if (static_cast<int64_t>(used) + nv > ar.S) {
                        if (sub == 0) atomicOr(err, ERR_ARENA_OVERFLOW);
                    } else {
                        const bool root_leaf = (len == 1);          // leaf.parent == -1, MCTS.h:349
                        float noise = 0.0f;
                        if (root_leaf && p.alpha > 0.0f && sub < nv) {
                            if (in.root_noise) {
                                noise = in.root_noise[static_cast<size_t>(t) * A + sub];
                            } else {
                                DevRng g(p.seed, *p.call_ptr, static_cast<uint64_t>(t), static_cast<uint64_t>(sub) + 16);
                                noise = g.gamma(p.alpha);
                            }
                        }
                        if (root_leaf && p.alpha > 0.0f && !in.root_noise) {
                            float sum = 0.0f;
                            if (L == WAVE) sum = wave_ordered_sum(noise, nv); else for (int i = 0; i < nv; ++i) sum += __shfl(noise, i, L);
                            noise = noise * (1.0f / (sum + 1e-8f));
                        }
                        if (sub < nv) {
                            HotRec h = empty_rec();
                            h.prior = prior; h.meta = static_cast<uint32_t>(my_action);
                            hot[used + sub] = h;
                            if (root_leaf) {                        // cold record: see k_backprop_spread
                                ColdRec cr;
                                cr.w_draw = 0.f; cr.noise = noise; cr.parent = leaf; cr.reserved = 0;
                                cold[used + sub] = cr;
                            }
                        }
                        if (sub == owner) {
                            lm = (lm & ~META_NEDGE_MASK) | META_EXPANDED |
                                 (static_cast<uint32_t>(nv) << META_NEDGE_SHIFT);
                            hot[leaf].child_off = used;
                            hot[leaf].meta = lm;
                        }
                        used += nv;
                        ++n_exp;
                    }
                }
            }

            // propagate, lane j <-> depth j (root = 0): the node `dist` levels above the leaf
            // receives the auxiliary value after `dist` per-ply steps (+1 or sign flip) and the
            // value decayed dist times
            for (int j = sub; j < len; j += L) {
                const int dist = len - 1 - j;
                float a = wd, b = w1, c = w2, mm = ml;
                const float g = p.value_decay;
                const float cst = (1.0f - g) * (1.0f / 3.0f);
                for (int i = 0; i < dist; ++i) {
                    if (G::AUX_PLUS_ONE) mm += 1.0f;
                    if (G::AUX_NEGATE) mm = -mm;
                    if (g < 1.0f) { a = fmaf(a, g, cst); b = fmaf(b, g, cst); c = fmaf(c, g, cst); }
                }
                const int slot = path[j];
                HotRec h = hot[slot];
                const float dr = h.n_visits != 0 ? cold[slot].w_draw : 0.0f;   // first backup through a node: zero
                h.n_visits += 1; h.w_p1 += b; h.w_p2 += c; h.m_sum += mm;
                hot[slot].n_visits = h.n_visits;
                hot[slot].w_p1 = h.w_p1; hot[slot].w_p2 = h.w_p2; hot[slot].m_sum = h.m_sum;
                cold[slot].w_draw = dr + a;
                ++n_backup;
            }
        }
        if (sub == 0 && used != used0) ar.used[t] = used;
    }
    wave_add_counter(counters, CNT_EXPANSIONS, sub == 0 ? n_exp : 0u);
    wave_add_counter(counters, CNT_DUP, sub == 0 ? n_dup : 0u);
    wave_add_counter(counters, CNT_BACKUP, n_backup);
}

// The same call for K <= KMAX with the dependent memory round trips taken out.  k_backprop walks
// k = 0..K-1 and, for each, loads the leaf, then its record, then its path, then the path's
// records, updates and stores them - about four dependent HBM round trips per k, sixteen per
// launch, which is what the kernel's 50 us were made of (90 % of its wave cycles in s_waitcnt).
// Here everything any k needs is fetched up front in two rounds (leaf descriptors and path slots;
// then the records), the K updates happen in registers in the reference's order - a node that
// several paths share (the root always, duplicates of a leaf) is found by comparing slots at the
// lane that owns its depth and is accumulated once per k in ascending k, the order of the
// reference's sequential read-modify-writes, so the float sums are bit-identical - and every
// distinct node is written back once.  Depths beyond the first LANES levels (rare) take the
// sequential route of k_backprop.
template <class G, bool VL, bool FUSED, int KMAX>
__global__ void __launch_bounds__(WAVE) k_backprop_batched(TreeArena ar, LeafBuf lf, SearchParams p, int K, int tpw,
                                                           EvalIn in, unsigned long long *counters, int *err)
{
    constexpr int L = G::LANES;
    constexpr int A = G::ACTIONS;
    const int lane = threadIdx.x;
    const int sub = lane % L;
    const int grp = lane / L;
    const int tree = blockIdx.x * tpw + grp;
    const bool live = grp < tpw && tree < ar.B;
    const int t = live ? tree : 0;
    HotRec *hot = ar.hot + tree_base(ar, t);
    ColdRec *cold = ar.cold + tree_base(ar, t);
    const size_t flat0 = static_cast<size_t>(t) * K;
    unsigned n_exp = 0, n_dup = 0, n_backup = 0;

    if (live) {
        // ---- round 1: what selection left behind for every k, and this lane's path slot
        int len[KMAX], leaf[KMAX], pslot[KMAX], sym[KMAX];
        uint8_t lflags[KMAX];
        GameState ls[KMAX];
        float ev_d[KMAX], ev_w[KMAX], ev_l[KMAX], ev_ml[KMAX];
        int used = ar.used[t];
        const int used0 = used;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const bool on = k < K;
            const size_t flat = flat0 + (on ? k : 0);
            len[k] = on ? lf.path_len[flat] : 0;
            leaf[k] = lf.slot[flat];
            lflags[k] = lf.flags[flat];
            ls[k].bb0 = lf.bb0[flat]; ls[k].bb1 = lf.bb1[flat]; ls[k].turn = lf.turn[flat]; ls[k].aux = lf.aux[flat];
            sym[k] = in.sym ? in.sym[flat] : lf.sym[flat];
            pslot[k] = lf.path[flat * G::MAX_PATH + sub];            // garbage past the path's end: masked by len
            if (FUSED) {
                ev_d[k] = in.wdl_rel[flat * 3]; ev_w[k] = in.wdl_rel[flat * 3 + 1]; ev_l[k] = in.wdl_rel[flat * 3 + 2];
            } else {
                ev_d[k] = in.d[flat]; ev_w[k] = in.p1w[flat]; ev_l[k] = in.p2w[flat];
            }
            ev_ml[k] = in.moves_left[flat];
        }
        // ---- round 2: the records (this lane's node of every path, every leaf's flags) and the
        // policy entry of this lane's move
        bool mine[KMAX];
        HotRec rec[KMAX];
        float cdraw[KMAX], my_pol[KMAX];
        uint32_t leaf_meta[KMAX];
        int nv[KMAX], my_action[KMAX];
        bool is_term_host[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            mine[k] = sub < len[k];
            const int slot = mine[k] ? pslot[k] : 0;
            rec[k] = hot[slot];
            cdraw[k] = (mine[k] && rec[k].n_visits != 0) ? cold[slot].w_draw : 0.0f;   // first backup through a node: zero
            leaf_meta[k] = len[k] > 0 ? hot[leaf[k]].meta : 0u;
            nv[k] = G::num_valid(ls[k]);
            my_action[k] = sub < nv[k] ? G::nth_valid(ls[k], sub) : -1;
            const size_t flat = flat0 + (k < K ? k : 0);
            my_pol[k] = my_action[k] >= 0 ? in.policy[flat * A + G::policy_index(sym[k], my_action[k])] : 0.0f;
            is_term_host[k] = FUSED ? false : in.is_term[flat] != 0;
        }

        // ---- virtual loss comes off every node of every recorded path (MCTS.h:561-581)
        int first[KMAX];                                            // earliest k whose node at this depth is the same record
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            first[k] = k;
#pragma unroll
            for (int q = KMAX - 1; q >= 0; --q)
                if (q < k && mine[q] && mine[k] && pslot[q] == pslot[k]) first[k] = q;
        }
        if (VL) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (len[k] > 0 && (lflags[k] & LEAF_VL_APPLIED)) {
#pragma unroll
                    for (int q = 0; q < KMAX; ++q)
                        if (mine[k] && first[k] == q) rec[q].n_inflight -= p.vl_count;
                    for (int j = sub + L; j < len[k]; j += L) hot[lf.path[(flat0 + k) * G::MAX_PATH + j]].n_inflight -= p.vl_count;
                    if (sub == 0) lf.flags[flat0 + k] = lflags[k] & static_cast<uint8_t>(~LEAF_VL_APPLIED);
                }
            }
        }

        // ---- k = 0..K-1 in order: expansion (MCTS.h:329-375), then the backup of that leaf
        bool expanded_here[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            expanded_here[k] = false;
            if (len[k] <= 0) continue;                              // MCTS.h:409,599
            const size_t flat = flat0 + k;
            const int owner = (len[k] - 1) % L;
            bool term;
            float wd, w1, w2, ml;
            if (FUSED) {
                term = (lflags[k] & LEAF_TERMINAL) != 0;
                if (term) {                                         // MCTS_cpp.py:275-282
                    const int code = (lflags[k] >> LEAF_RESULT_SHIFT) & 3;
                    wd = code == 0 ? 1.0f : 0.0f; w1 = code == 1 ? 1.0f : 0.0f; w2 = code == 2 ? 1.0f : 0.0f;
                    ml = 0.0f;
                } else {                                            // MCTS_cpp.py:23-30
                    wd = ev_d[k];
                    w1 = (ls[k].turn == 1) ? ev_w[k] : ev_l[k];
                    w2 = (ls[k].turn == 1) ? ev_l[k] : ev_w[k];
                    ml = ev_ml[k];
                }
            } else {
                term = is_term_host[k];
                wd = ev_d[k]; w1 = ev_w[k]; w2 = ev_l[k];
                ml = ev_ml[k];
            }
            if (term) ml = G::terminal_aux(ls[k], p);               // MCTS.h:412,608

            if (!term) {
                // already expanded: before this call, or by an earlier k of it (MCTS.h:601-607)
                bool was = (leaf_meta[k] & META_EXPANDED) != 0;
#pragma unroll
                for (int q = 0; q < KMAX; ++q)
                    if (q < k && expanded_here[q] && leaf[q] == leaf[k]) was = true;
                if (VL && was) {
                    ++n_dup;
                } else {
                    float psum = 0.0f;
                    if (L <= 8) {
#pragma unroll
                        for (int i = 0; i < L - 1; ++i) psum += __shfl(my_pol[k], i, L);
                    } else {
                        if (L == WAVE) psum = wave_ordered_sum(my_pol[k], nv[k]); else for (int i = 0; i < nv[k]; ++i) psum += __shfl(my_pol[k], i, L);
                    }
                    const float prior = my_pol[k] / (psum + 1e-8f);  // MCTS.h:370
                    if (static_cast<int64_t>(used) + nv[k] > ar.S) {
                        if (sub == 0) atomicOr(err, ERR_ARENA_OVERFLOW);
                    } else {
                        const bool root_leaf = (len[k] == 1);       // leaf.parent == -1, MCTS.h:349
                        float noise = 0.0f;
                        if (root_leaf && p.alpha > 0.0f && sub < nv[k]) {
                            if (in.root_noise) {
                                noise = in.root_noise[static_cast<size_t>(t) * A + sub];
                            } else {
                                DevRng g(p.seed, *p.call_ptr, static_cast<uint64_t>(t), static_cast<uint64_t>(sub) + 16);
                                noise = g.gamma(p.alpha);
                            }
                        }
                        if (root_leaf && p.alpha > 0.0f && !in.root_noise) {
                            float sum = 0.0f;
                            if (L == WAVE) sum = wave_ordered_sum(noise, nv[k]); else for (int i = 0; i < nv[k]; ++i) sum += __shfl(noise, i, L);
                            noise = noise * (1.0f / (sum + 1e-8f));
                        }
                        if (sub < nv[k]) {
                            HotRec h = empty_rec();
                            h.prior = prior; h.meta = static_cast<uint32_t>(my_action[k]);
                            hot[used + sub] = h;
                            if (root_leaf) {                        // cold record: see k_backprop_spread
                                ColdRec cr;
                                cr.w_draw = 0.f; cr.noise = noise; cr.parent = leaf[k]; cr.reserved = 0;
                                cold[used + sub] = cr;
                            }
                        }
                        if (sub == owner) {
                            const uint32_t lm = (leaf_meta[k] & ~META_NEDGE_MASK) | META_EXPANDED |
                                                (static_cast<uint32_t>(nv[k]) << META_NEDGE_SHIFT);
                            hot[leaf[k]].child_off = used;
                            hot[leaf[k]].meta = lm;
                        }
                        used += nv[k];
                        expanded_here[k] = true;
                        ++n_exp;
                    }
                }
            }

            // propagate (MCTS.h:381-402): lane j <-> depth j; the node `dist` levels above the leaf
            // receives the auxiliary value after `dist` per-ply steps and the value decayed dist times
            const float g = p.value_decay;
            const float cst = (1.0f - g) * (1.0f / 3.0f);
            if (mine[k]) {
                const int dist = len[k] - 1 - sub;
                float a = wd, b = w1, c = w2, mm = ml;
                for (int i = 0; i < dist; ++i) {
                    if (G::AUX_PLUS_ONE) mm += 1.0f;
                    if (G::AUX_NEGATE) mm = -mm;
                    if (g < 1.0f) { a = fmaf(a, g, cst); b = fmaf(b, g, cst); c = fmaf(c, g, cst); }
                }
#pragma unroll
                for (int q = 0; q < KMAX; ++q)
                    if (first[k] == q) {
                        rec[q].n_visits += 1; rec[q].w_p1 += b; rec[q].w_p2 += c; rec[q].m_sum += mm;
                        cdraw[q] += a;
                    }
                ++n_backup;
            }
            for (int j = sub + L; j < len[k]; j += L) {             // deeper than the first LANES levels
                const int dist = len[k] - 1 - j;
                float a = wd, b = w1, c = w2, mm = ml;
                for (int i = 0; i < dist; ++i) {
                    if (G::AUX_PLUS_ONE) mm += 1.0f;
                    if (G::AUX_NEGATE) mm = -mm;
                    if (g < 1.0f) { a = fmaf(a, g, cst); b = fmaf(b, g, cst); c = fmaf(c, g, cst); }
                }
                const int slot = lf.path[flat * G::MAX_PATH + j];
                HotRec h = hot[slot];
                const float dr = h.n_visits != 0 ? cold[slot].w_draw : 0.0f;   // first backup through a node: zero
                h.n_visits += 1; h.w_p1 += b; h.w_p2 += c; h.m_sum += mm;
                hot[slot].n_visits = h.n_visits;
                hot[slot].w_p1 = h.w_p1; hot[slot].w_p2 = h.w_p2; hot[slot].m_sum = h.m_sum;
                cold[slot].w_draw = dr + a;
                ++n_backup;
            }
        }
        // ---- every distinct node goes back once (statistics only: child_off / meta were written above)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (mine[k] && first[k] == k) {
                HotRec *d = hot + pslot[k];
                d->n_visits = rec[k].n_visits; d->n_inflight = rec[k].n_inflight;
                d->w_p1 = rec[k].w_p1; d->w_p2 = rec[k].w_p2; d->m_sum = rec[k].m_sum;
                cold[pslot[k]].w_draw = cdraw[k];
            }
        }
        if (sub == 0 && used != used0) ar.used[t] = used;
    }
    // the three work counters in one reduction: a wavefront backs up < 2^16 nodes, expands and skips <= 32 leaves.
    // The fields must not carry into each other, and readlane(63) needs every lane of the wavefront here:
    // no path of this kernel may return before this point.
    static_assert((WAVE / G::LANES) * KMAX < 256, "expansion / duplicate counters of a wavefront are 8-bit fields");
    static_assert((WAVE / G::LANES) * KMAX * G::MAX_PATH < 65536, "the backup-node counter of a wavefront is a 16-bit field");
    {
        const unsigned tot = wave_sum_u32(n_backup | (sub == 0 ? (n_exp << 16) | (n_dup << 24) : 0u));
        if (lane == 0) {
            unsigned long long *c = counters + (blockIdx.x % CNT_STRIPES) * CNT_N;
            if ((tot >> 16) & 0xffu) atomicAdd(&c[CNT_EXPANSIONS], static_cast<unsigned long long>((tot >> 16) & 0xffu));
            if (tot >> 24) atomicAdd(&c[CNT_DUP], static_cast<unsigned long long>(tot >> 24));
            if (tot & 0xffffu) atomicAdd(&c[CNT_BACKUP], static_cast<unsigned long long>(tot & 0xffffu));
        }
    }
}

// The virtual-loss call with the K leaves of a tree SPREAD over K groups of lanes (KG = K rounded up to a power of
// two; lane = tree_in_wave * KG * LANES + k * LANES + depth): four times the wavefronts of k_backprop_batched at
// K = 4, each with a quarter of the instruction stream, so a SIMD has four wavefronts to switch between while
// records are in flight instead of one.  What k_backprop_batched kept in KMAX-unrolled registers is exchanged
// between the groups of a tree (one ds_bpermute per value and partner):
//   * a node that several paths share sits at the same depth of each of them, i.e. at the same `sub` of several
//     groups: the first group that holds it owns it, takes the record and adds the contributions of the groups
//     k' >= k in ascending k' - the order of the reference's sequential read-modify-writes (MCTS.h:381-402), so
//     the float sums are bit-identical - and writes it back once;
//   * the allocation of child blocks is a serial scan over k (MCTS.h:329-375: a leaf selected twice is expanded by
//     its first k only; a block that does not fit raises the overflow flag and is skipped); every lane runs the
//     K-step scan on the exchanged (leaf, edge count, candidate) triples and keeps its own group's outcome.
// Levels past the first LANES of a path (rare) are read-modify-written in memory, group after group.
// Expansion no longer writes the 16-byte cold record of a child that is not the root's: its noise is read for the
// root's children only (re-rooting writes it, k_prune) and w_draw only once n_visits != 0 - the first backup
// through a node takes w_draw as zero instead of loading it.
template <class G, bool FUSED, int KG, int WPB>
__global__ void __launch_bounds__(WAVE * WPB) k_backprop_spread(TreeArena ar, LeafBuf lf, SearchParams p, int K,
                                                                EvalIn in, unsigned long long *counters, int *err)
{
    constexpr int L = G::LANES;
    constexpr int A = G::ACTIONS;
    constexpr int TPW = WAVE / (L * KG);
    static_assert(L * KG <= WAVE && L <= 8, "groups of a tree share a wavefront");
    const int lane = threadIdx.x % WAVE;
    const int sub = lane % L;
    const int k = (lane / L) % KG;
    const int tw = lane / (L * KG);
    const int base = tw * (L * KG) + sub;                           // lane of group 0 at this depth
    const int tree = (blockIdx.x * WPB + threadIdx.x / WAVE) * TPW + tw;
    const bool live = tree < ar.B;
    const int t = live ? tree : 0;
    HotRec *hot = ar.hot + tree_base(ar, t);
    ColdRec *cold = ar.cold + tree_base(ar, t);
    const bool on = live && k < K;
    const size_t flat = static_cast<size_t>(t) * K + (on ? k : 0);
    unsigned n_exp = 0, n_dup = 0, n_backup = 0;

    // ---- round 1: what selection left behind for this group's leaf, and this lane's path slot
    const int len = on ? lf.path_len[flat] : 0;
    const int leaf = lf.slot[flat];
    const uint8_t lflags = lf.flags[flat];
    GameState ls;
    ls.bb0 = lf.bb0[flat]; ls.bb1 = lf.bb1[flat]; ls.turn = lf.turn[flat]; ls.aux = lf.aux[flat];
    const int sym = in.sym ? in.sym[flat] : lf.sym[flat];
    const int pslot_raw = lf.path[flat * G::MAX_PATH + sub];
    float ev_d, ev_w, ev_l;
    if (FUSED) { ev_d = in.wdl_rel[flat * 3]; ev_w = in.wdl_rel[flat * 3 + 1]; ev_l = in.wdl_rel[flat * 3 + 2]; }
    else       { ev_d = in.d[flat]; ev_w = in.p1w[flat]; ev_l = in.p2w[flat]; }
    const float ev_ml = in.moves_left[flat];
    const int used0 = ar.used[t];
    const bool is_term_host = FUSED ? false : in.is_term[flat] != 0;

    // ---- round 2: this lane's node of the path, the leaf's flags, the policy entry of this lane's move
    const bool mine = sub < len;
    const int pslot = mine ? pslot_raw : 0;
    HotRec rec = hot[pslot];
    float cdraw = (mine && rec.n_visits != 0) ? cold[pslot].w_draw : 0.0f;
    const uint32_t leaf_meta = len > 0 ? hot[leaf].meta : 0u;
    const int nv = G::num_valid(ls);
    const int my_action = sub < nv ? G::nth_valid(ls, sub) : -1;
    const float my_pol = my_action >= 0 ? in.policy[flat * A + G::policy_index(sym, my_action)] : 0.0f;

    // ---- the leaf's value as the tree takes it (MCTS_cpp.py:23-30, 275-282; MCTS.h:412,608)
    bool term;
    float wd, w1, w2, ml;
    if (FUSED) {
        term = (lflags & LEAF_TERMINAL) != 0;
        if (term) {
            const int code = (lflags >> LEAF_RESULT_SHIFT) & 3;
            wd = code == 0 ? 1.0f : 0.0f; w1 = code == 1 ? 1.0f : 0.0f; w2 = code == 2 ? 1.0f : 0.0f;
            ml = 0.0f;
        } else {
            wd = ev_d;
            w1 = (ls.turn == 1) ? ev_w : ev_l;
            w2 = (ls.turn == 1) ? ev_l : ev_w;
            ml = ev_ml;
        }
    } else {
        term = is_term_host;
        wd = ev_d; w1 = ev_w; w2 = ev_l; ml = ev_ml;
    }
    if (term) ml = G::terminal_aux(ls, p);
    const bool vl_on = len > 0 && (lflags & LEAF_VL_APPLIED) != 0;
    const bool cand = len > 0 && !term && !(leaf_meta & META_EXPANDED);   // would be expanded if no earlier k took the leaf
    const bool dup0 = len > 0 && !term && (leaf_meta & META_EXPANDED);    // expanded before this call: a duplicate

    // this lane's contribution to its node: the node `dist` levels above the leaf receives the auxiliary value
    // after `dist` per-ply steps and the value decayed dist times (MCTS.h:381-402)
    const float g = p.value_decay;
    const float cst = (1.0f - g) * (1.0f / 3.0f);
    float ca = wd, cb = w1, cc = w2, cm = ml;
    {
        const int dist = len - 1 - sub;
        for (int i = 0; i < dist; ++i) {
            if (G::AUX_PLUS_ONE) cm += 1.0f;
            if (G::AUX_NEGATE) cm = -cm;
            if (g < 1.0f) { ca = fmaf(ca, g, cst); cb = fmaf(cb, g, cst); cc = fmaf(cc, g, cst); }
        }
    }

    // ---- exchange between the groups of the tree
    const int word = (mine ? 1 : 0) | (vl_on ? 2 : 0) | (cand ? 4 : 0) | (nv << 8);
    int q_slot[KG], q_word[KG], q_leaf[KG];
    float q_a[KG], q_b[KG], q_c[KG], q_m[KG];
#pragma unroll
    for (int q = 0; q < KG; ++q) {
        const int src = base + q * L;
        q_slot[q] = __shfl(pslot, src, WAVE);
        q_word[q] = __shfl(word, src, WAVE);
        q_leaf[q] = __shfl(leaf, src, WAVE);
        q_a[q] = __shfl(ca, src, WAVE); q_b[q] = __shfl(cb, src, WAVE);
        q_c[q] = __shfl(cc, src, WAVE); q_m[q] = __shfl(cm, src, WAVE);
    }

    // ---- allocation scan over k (every lane, same result within a tree)
    int used = used0;
    bool exp_me = false, dup_me = dup0, ovf_me = false;
    int off_me = 0;
    {
        bool exp_q[KG];
#pragma unroll
        for (int q = 0; q < KG; ++q) {
            exp_q[q] = false;
            const bool c_q = (q_word[q] & 4) != 0;
            const int nv_q = (q_word[q] >> 8) & 0xff;
            bool was = false;
#pragma unroll
            for (int r = 0; r < KG; ++r)
                if (r < q && exp_q[r] && q_leaf[r] == q_leaf[q]) was = true;
            if (c_q && was) { if (q == k) dup_me = true; }
            else if (c_q) {
                if (static_cast<int64_t>(used) + nv_q > ar.S) { if (q == k) ovf_me = true; }
                else {
                    exp_q[q] = true;
                    if (q == k) { exp_me = true; off_me = used; }
                    used += nv_q;
                }
            }
        }
    }

    if (on) {
        // ---- virtual loss comes off every node of every recorded path (MCTS.h:561-581)
        if (vl_on) {
            for (int j = sub + L; j < len; j += L) atomicSub(&hot[lf.path[flat * G::MAX_PATH + j]].n_inflight, p.vl_count);
            if (sub == 0) lf.flags[flat] = lflags & static_cast<uint8_t>(~LEAF_VL_APPLIED);
        }

        // ---- expansion of this group's leaf (MCTS.h:329-375)
        if (ovf_me && sub == 0) atomicOr(err, ERR_ARENA_OVERFLOW);
        if (dup_me && sub == 0) ++n_dup;
    }
    {
        // (the shuffles of the group's ordered sums need every lane of the group: outside the branches)
        float psum = 0.0f;
#pragma unroll
        for (int i = 0; i < L - 1; ++i) psum += __shfl(my_pol, i, L);
        const float prior = my_pol / (psum + 1e-8f);                // MCTS.h:370
        const bool root_leaf = (len == 1);                          // leaf.parent == -1, MCTS.h:349
        float noise = 0.0f;
        if (on && exp_me && root_leaf && p.alpha > 0.0f && sub < nv) {
            if (in.root_noise) {
                noise = in.root_noise[static_cast<size_t>(t) * A + sub];
            } else {
                DevRng rg(p.seed, *p.call_ptr, static_cast<uint64_t>(t), static_cast<uint64_t>(sub) + 16);
                noise = rg.gamma(p.alpha);
            }
        }
        if (p.alpha > 0.0f && !in.root_noise) {
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < L - 1; ++i) sum += __shfl(noise, i, L);   // lanes >= nv hold 0: the same sum as over nv terms
            if (on && exp_me && root_leaf) noise = noise * (1.0f / (sum + 1e-8f));
        }
        if (on && exp_me) {
            if (sub < nv) {
                HotRec h = empty_rec();
                h.prior = prior; h.meta = static_cast<uint32_t>(my_action);
                hot[off_me + sub] = h;
                if (root_leaf) {
                    ColdRec cr;
                    cr.w_draw = 0.f; cr.noise = noise; cr.parent = leaf; cr.reserved = 0;
                    cold[off_me + sub] = cr;
                }
            }
            if (sub == (len - 1) % L) {
                hot[leaf].child_off = off_me;
                hot[leaf].meta = (leaf_meta & ~META_NEDGE_MASK) | META_EXPANDED | (static_cast<uint32_t>(nv) << META_NEDGE_SHIFT);
            }
            if (sub == 0) ++n_exp;
        }
    }
    if (on) {
        // ---- statistics: the owner of a node adds the contributions in ascending k and writes the node back once
        bool own = mine;                                            // the first group that holds this node owns it
        int infl = 0;
#pragma unroll
        for (int q = 0; q < KG; ++q) {
            const bool same = (q_word[q] & 1) && q_slot[q] == pslot;
            if (q < k && same) own = false;
            if (same && (q_word[q] & 2)) infl += p.vl_count;
        }
        if (mine) ++n_backup;
        if (own) {
#pragma unroll
            for (int q = 0; q < KG; ++q) {
                const bool same = (q_word[q] & 1) && q_slot[q] == pslot;
                if (q >= k && same) {
                    rec.n_visits += 1; rec.w_p1 += q_b[q]; rec.w_p2 += q_c[q]; rec.m_sum += q_m[q];
                    cdraw += q_a[q];
                }
            }
            HotRec *d = hot + pslot;
            d->n_visits = rec.n_visits; d->n_inflight = rec.n_inflight - infl;
            d->w_p1 = rec.w_p1; d->w_p2 = rec.w_p2; d->m_sum = rec.m_sum;
            cold[pslot].w_draw = cdraw;
        }
        if (sub == 0 && k == 0 && used != used0) ar.used[t] = used;
    }
    // ---- levels past the first LANES of a path: in memory, group after group (two groups may share such a node)
    if (__any(on && len > L)) {
#pragma unroll 1
        for (int q = 0; q < KG; ++q) {
            if (on && q == k) {
                for (int j = sub + L; j < len; j += L) {
                    const int dist = len - 1 - j;
                    float a = wd, b = w1, c = w2, mm = ml;
                    for (int i = 0; i < dist; ++i) {
                        if (G::AUX_PLUS_ONE) mm += 1.0f;
                        if (G::AUX_NEGATE) mm = -mm;
                        if (g < 1.0f) { a = fmaf(a, g, cst); b = fmaf(b, g, cst); c = fmaf(c, g, cst); }
                    }
                    const int slot = lf.path[flat * G::MAX_PATH + j];
                    HotRec h = hot[slot];
                    const float dr = h.n_visits != 0 ? cold[slot].w_draw : 0.0f;
                    h.n_visits += 1; h.w_p1 += b; h.w_p2 += c; h.m_sum += mm;
                    hot[slot].n_visits = h.n_visits;
                    hot[slot].w_p1 = h.w_p1; hot[slot].w_p2 = h.w_p2; hot[slot].m_sum = h.m_sum;
                    cold[slot].w_draw = dr + a;
                    ++n_backup;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_s_waitcnt(0);
        }
    }
    static_assert((WAVE / G::LANES) < 256, "expansion / duplicate counters of a wavefront are 8-bit fields");
    static_assert((WAVE / G::LANES) * G::MAX_PATH < 65536, "the backup-node counter of a wavefront is a 16-bit field");
    {
        const unsigned tot = wave_sum_u32(n_backup | (n_exp << 16) | (n_dup << 24));
        if (lane == 0) {
            unsigned long long *c = counters + (blockIdx.x % CNT_STRIPES) * CNT_N;
            if ((tot >> 16) & 0xffu) atomicAdd(&c[CNT_EXPANSIONS], static_cast<unsigned long long>((tot >> 16) & 0xffu));
            if (tot >> 24) atomicAdd(&c[CNT_DUP], static_cast<unsigned long long>(tot >> 24));
            if (tot & 0xffffu) atomicAdd(&c[CNT_BACKUP], static_cast<unsigned long long>(tot & 0xffffu));
        }
    }
}

// ------------------------------------------------------------------ leaf gather

// BatchedMCTS.h:141-169 / 254-283 (symmetry, grid export, valid mask) and MCTS_cpp.py:15-20
// (relative feature planes).  One thread per (leaf, cell); cells of a leaf are contiguous so
// every plane row is a coalesced store.
template <class G>
__global__ void __launch_bounds__(256) k_export(LeafBuf lf, SearchParams p, int n_leaves, int gen_sym,
                                                int8_t *boards, uint8_t *valid_mask, float *features)
{
    constexpr int CELLS = G::CELLS, A = G::ACTIONS;
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t leaf = gid / CELLS;
    const int cell = static_cast<int>(gid - leaf * CELLS);
    if (leaf >= n_leaves) return;
    GameState s;
    s.bb0 = lf.bb0[leaf]; s.bb1 = lf.bb1[leaf]; s.turn = lf.turn[leaf]; s.aux = lf.aux[leaf];
    const bool term = (lf.flags[leaf] & LEAF_TERMINAL) != 0;
    int sym;
    if (gen_sym) {
        sym = 0;
        if (!term && p.use_symmetry) {
            DevRng g(p.seed, *p.call_ptr, static_cast<uint64_t>(leaf), 7);
            sym = G::sym_of_choice(static_cast<int>((static_cast<uint64_t>(g.next()) * G::SYM_CHOICES) >> 32));
        }
        if (cell == 0) lf.sym[leaf] = sym;
    } else {
        sym = lf.sym[leaf];
    }
    const int v = G::cell_value(s, sym, cell);
    if (boards) boards[leaf * CELLS + cell] = static_cast<int8_t>(v);
    if (features) {
        float *f = features + leaf * (3 * CELLS) + cell;
        f[0] = (v == s.turn) ? 1.0f : 0.0f;
        f[CELLS] = (v == -s.turn) ? 1.0f : 0.0f;
        f[2 * CELLS] = static_cast<float>(s.turn);
    }
    if (valid_mask) {
        for (int a = cell; a < A; a += CELLS)
            valid_mask[leaf * A + a] = (!term && G::valid_in_frame(s, sym, a)) ? 1 : 0;
    }
}

// What the native loop needs between a selection and its evaluator when the evaluator reads leaf POSITIONS
// (az_nn_model_forward_positions) instead of feature planes: the symmetry id every non-terminal leaf is shown
// under (BatchedMCTS.h:148-154), its action mask in that frame, and the compact list of the leaves to
// evaluate (everything but terminal leaves, MCTS_cpp.py:275-297) - k_export's ids and mask and
// k_live_leaves' list in one pass of one thread per leaf, with no feature tensor written or read.
template <class G>
__global__ void __launch_bounds__(1024) k_leaf_prep(LeafBuf lf, SearchParams p, int n_leaves, int gen_sym,
                                                    uint8_t *valid_mask, int32_t *idx, long long *count, int *err)
{
    constexpr int A = G::ACTIONS;
    __shared__ int s_wave[16];
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t leaf = static_cast<int64_t>(blockIdx.x) * blockDim.x + tid;
    bool livel = false;
    if (leaf < n_leaves) {
        GameState s;
        s.bb0 = lf.bb0[leaf]; s.bb1 = lf.bb1[leaf]; s.turn = lf.turn[leaf]; s.aux = lf.aux[leaf];
        const bool term = (lf.flags[leaf] & LEAF_TERMINAL) != 0;
        livel = !term;
        int sym;
        if (gen_sym) {
            sym = 0;
            if (!term && p.use_symmetry) {                         // the draw of k_export, leaf for leaf
                DevRng g(p.seed, *p.call_ptr, static_cast<uint64_t>(leaf), 7);
                sym = G::sym_of_choice(static_cast<int>((static_cast<uint64_t>(g.next()) * G::SYM_CHOICES) >> 32));
            }
            lf.sym[leaf] = sym;
        } else {
            sym = lf.sym[leaf];
        }
        if (valid_mask != nullptr)
            for (int a = 0; a < A; ++a) valid_mask[leaf * A + a] = (!term && G::valid_in_frame(s, sym, a)) ? 1 : 0;
    }
    if (idx == nullptr) return;                                    // uniform: the table's lookup builds the list
    const unsigned long long m = __ballot(livel);
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int c = s_wave[w];
        if (w < wave) before += c;
        total += c;
    }
    if (tid == 0)
        s_base = total ? static_cast<long long>(atomicAdd(reinterpret_cast<unsigned long long *>(count),
                                                          static_cast<unsigned long long>(total))) : 0;
    __syncthreads();
    if (livel) {
        const long long pos = s_base + before + __popcll(m & ((1ull << lane) - 1));
        if (pos >= 0 && pos < n_leaves) idx[pos] = static_cast<int32_t>(leaf);
        else atomicOr(err, ERR_LIST_OVERFLOW);
    }
}

// ------------------------------------------------------------------ tree maintenance

__device__ __forceinline__ void write_fresh_root(HotRec *hot, ColdRec *cold)
{
    HotRec h = empty_rec();                                            // MCTS.h:77-82
    h.meta = META_TURN_P1 | META_EXISTS;
    hot[0] = h;
    ColdRec c;
    c.w_draw = 0.f; c.noise = 0.f; c.parent = -1; c.reserved = 0;
    cold[0] = c;
}

__global__ void __launch_bounds__(256) k_init_trees(TreeArena ar)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ar.B) return;
    write_fresh_root(ar.hot + tree_base(ar, t), ar.cold + tree_base(ar, t));
    ar.root[t] = 0;
    ar.used[t] = 1;
}

__global__ void __launch_bounds__(256) k_reset_masked(TreeArena ar, const uint8_t *mask)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ar.B || !mask[t]) return;
    write_fresh_root(ar.hot + tree_base(ar, t), ar.cold + tree_base(ar, t));
    ar.root[t] = 0;
    ar.used[t] = 1;
}

// group-local ballot: bit i set <=> lane i of this lane group votes true
template <int L>
__device__ __forceinline__ unsigned long long group_ballot(bool pred, int lane)
{
    const unsigned long long bal = __ballot(pred);
    constexpr unsigned long long mask = L >= 64 ? ~0ull : ((1ull << (L & 63)) - 1ull);
    return (bal >> (lane - lane % L)) & mask;
}

// MCTS.h:90-132: re-root at the child reached by `action` if the reference would have allocated it,
// else reset.  The reference keeps the whole old tree in its pools until the next reset
// (MCTS.h:100-101: only `root_idx` moves); here the KEPT SUBTREE IS COPIED into the tree's other
// arena half, breadth first, and the tree continues there: what a tree occupies is what is
// reachable from its root (node numbering is not observable through the API; the order of records
// inside a sibling block - the edge order - is kept).  One wavefront per tree: a pass takes the
// records appended by the previous pass (64 at a time), a wave-wide prefix sum of their edge counts
// places their child blocks behind what is there, every lane copies its node's block.  Stores of a
// pass are read by other lanes of the same wavefront in the next one (wavefront-scope order, as in
// k_select8x4; the fence also drains the stores).  noise_req[t] = number of root edges that need
// fresh Dirichlet noise (host generator); with dev_noise the noise is written here, from the device
// generator or from `replay_noise`.  max_live: running maximum of the records a tree occupies now.
template <class G>
__global__ void __launch_bounds__(WAVE) k_prune(TreeArena ar, SearchParams p, const int32_t *actions,
                                                int32_t *noise_req, int dev_noise, const float *replay_noise,
                                                int *max_live, int *err, int compact_above)
{
    const int lane = threadIdx.x;
    const int t = blockIdx.x;
    const int S = static_cast<int>(ar.S);
    const int h = ar.half[t];
    HotRec *hot = ar.hot + (static_cast<size_t>(t) * 2 + h) * ar.S;
    ColdRec *cold = ar.cold + (static_cast<size_t>(t) * 2 + h) * ar.S;
    HotRec *nhot = ar.hot + (static_cast<size_t>(t) * 2 + (1 - h)) * ar.S;
    ColdRec *ncold = ar.cold + (static_cast<size_t>(t) * 2 + (1 - h)) * ar.S;
    const int root = ar.root[t];
    const HotRec R = hot[root];
    const int action = actions[t];
    const int E = (R.meta & META_EXPANDED) ? static_cast<int>((R.meta & META_NEDGE_MASK) >> META_NEDGE_SHIFT) : 0;
    HotRec c = empty_rec();
    bool match = false;
    if (lane < E) {
        c = hot[R.child_off + lane];
        match = static_cast<int>(c.meta & META_ACTION_MASK) == action && (c.meta & META_EXISTS);
    }
    const unsigned long long mb = __ballot(match);
    if (!mb) {                                                         // MCTS.h:107: reset()
        if (lane == 0) {
            write_fresh_root(hot, cold);
            ar.root[t] = 0;
            ar.used[t] = 1;
            if (!dev_noise) noise_req[t] = 0;
            atomicMax(max_live, 1);
        }
        return;
    }
    const int e = __ffsll(mb) - 1;
    const int old_root = R.child_off + e;
    // A tree that still has room for the plies to come stays where it is - the root moves to the chosen
    // record, as in the reference (MCTS.h:100-101) - and only a tree past `compact_above` records pays
    // for the copy: every few plies instead of every ply.
    if (ar.used[t] <= compact_above) {
        const uint32_t nm = static_cast<uint32_t>(__shfl(static_cast<int>(c.meta), e, WAVE));
        const int noff0 = __shfl(c.child_off, e, WAVE);
        const int nE0 = (nm & META_EXPANDED) ? static_cast<int>((nm & META_NEDGE_MASK) >> META_NEDGE_SHIFT) : 0;
        if (lane == 0) {
            ar.root[t] = old_root;
            cold[old_root].parent = -1;
            atomicMax(max_live, ar.used[t]);
        }
        const int want0 = (p.alpha > 0.0f) ? nE0 : 0;                   // apply_root_noise, MCTS.h:113-132
        if (want0 == 0 && lane < nE0) cold[noff0 + lane].noise = 0.0f;  // expansion writes the noise of the root's children only
        if (!dev_noise) {
            if (lane == 0) noise_req[t] = want0;
        } else if (want0 > 0 && replay_noise != nullptr) {
            if (lane < want0) cold[noff0 + lane].noise = replay_noise[static_cast<size_t>(t) * G::ACTIONS + lane];
        } else if (want0 > 0) {
            float g = 0.0f;
            if (lane < want0) {
                DevRng rng(p.seed, *p.call_ptr, static_cast<uint64_t>(t), static_cast<uint64_t>(lane) + 128);
                g = rng.gamma(p.alpha);
            }
            const float sum = wave_ordered_sum(g, want0);   // lanes >= want0 hold 0
            if (lane < want0) cold[noff0 + lane].noise = g * (1.0f / (sum + 1e-8f));
        }
        return;
    }
    if (lane == 0) {
        nhot[0] = hot[old_root];
        ColdRec cr = cold[old_root];
        cr.parent = -1;                                                // MCTS.h:100-101
        ncold[0] = cr;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    int lo = 0, hi = 1, used = 1;
    bool overflow = false;
    while (lo < hi && !overflow) {
        for (int q0 = lo; q0 < hi; q0 += WAVE) {
            const int q = q0 + lane;
            const bool valid = q < hi;
            HotRec rec = empty_rec();
            if (valid) rec = nhot[q];                                  // still points at its OLD child block
            const int nE = (valid && (rec.meta & META_EXPANDED)) ? static_cast<int>((rec.meta & META_NEDGE_MASK) >> META_NEDGE_SHIFT) : 0;
            int incl = nE;
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                const int v = __shfl_up(incl, o, WAVE);
                if (lane >= o) incl += v;
            }
            const int total = __shfl(incl, WAVE - 1, WAVE);
            if (used + total > S) { overflow = true; break; }         // cannot happen: the subtree fitted in one half before
            if (nE > 0) {
                const int dst = used + incl - nE;
                nhot[q].child_off = dst;
                for (int j = 0; j < nE; ++j) {
                    nhot[dst + j] = hot[rec.child_off + j];
                    ColdRec y = cold[rec.child_off + j];
                    y.parent = q;
                    ncold[dst + j] = y;
                }
            }
            used += total;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        lo = hi; hi = used;
    }
    if (overflow && lane == 0) atomicOr(err, ERR_ARENA_OVERFLOW);
    const HotRec NR = nhot[0];
    const int nE = (NR.meta & META_EXPANDED) ? static_cast<int>((NR.meta & META_NEDGE_MASK) >> META_NEDGE_SHIFT) : 0;
    const int noff = NR.child_off;
    if (lane == 0) {
        ar.half[t] = static_cast<uint8_t>(1 - h);
        ar.root[t] = 0;
        ar.used[t] = used;
        atomicMax(max_live, used);
    }
    const int want = (p.alpha > 0.0f) ? nE : 0;                        // apply_root_noise, MCTS.h:113-132
    if (want == 0 && lane < nE) ncold[noff + lane].noise = 0.0f;
    if (!dev_noise) {
        if (lane == 0) noise_req[t] = want;
    } else if (want > 0 && replay_noise != nullptr) {
        // recorded draws instead of the generator (az_mcts_dev_replay): row t, edge order
        if (lane < want) ncold[noff + lane].noise = replay_noise[static_cast<size_t>(t) * G::ACTIONS + lane];
    } else if (want > 0) {
        float g = 0.0f;
        if (lane < want) {
            DevRng rng(p.seed, *p.call_ptr, static_cast<uint64_t>(t), static_cast<uint64_t>(lane) + 128);
            g = rng.gamma(p.alpha);
        }
        const float sum = wave_ordered_sum(g, want);   // lanes >= want hold 0
        if (lane < want) ncold[noff + lane].noise = g * (1.0f / (sum + 1e-8f));
    }
}

template <class G>
__global__ void __launch_bounds__(WAVE) k_apply_noise(TreeArena ar, const int32_t *noise_req, const float *noise)
{
    constexpr int L = G::LANES;
    const int sub = threadIdx.x % L;
    const int tree = blockIdx.x * (WAVE / L) + threadIdx.x / L;
    if (tree >= ar.B) return;
    const int nv = noise_req[tree];
    if (nv <= 0 || sub >= nv) return;
    const HotRec *hot = ar.hot + tree_base(ar, tree);
    ColdRec *cold = ar.cold + tree_base(ar, tree);
    const int off = hot[ar.root[tree]].child_off;
    cold[off + sub].noise = noise[static_cast<size_t>(tree) * G::ACTIONS + sub];
}

// ------------------------------------------------------------------ root queries

// get_counts (MCTS.h:617-630) and get_root_stats (MCTS.h:637-673).  Lane e scatters the values
// of edge e into the slot of its action; the slots of actions that have no edge are zeroed by
// other writes to DISJOINT addresses (presence comes from a group ballot), so no two lanes
// ever store to the same word.
template <class G, bool STATS>
__global__ void __launch_bounds__(WAVE) k_root_query(TreeArena ar, int32_t *counts, float *stats)
{
    constexpr int L = G::LANES, A = G::ACTIONS;
    const int lane = threadIdx.x;
    const int sub = lane % L;
    const int tree = blockIdx.x * (WAVE / L) + lane / L;
    const bool live = tree < ar.B;
    const int t = live ? tree : 0;
    const HotRec *hot = ar.hot + tree_base(ar, t);
    const ColdRec *cold = ar.cold + tree_base(ar, t);
    const int root = ar.root[t];
    const HotRec R = hot[root];
    const bool expanded = (R.meta & META_EXPANDED) != 0;
    const int E = expanded ? static_cast<int>((R.meta & META_NEDGE_MASK) >> META_NEDGE_SHIFT) : 0;
    HotRec c = empty_rec();
    ColdRec cc;
    cc.w_draw = 0.f; cc.noise = 0.f; cc.parent = -1; cc.reserved = 0;
    const bool has = live && sub < E;
    if (has) {
        c = hot[R.child_off + sub];
        if (STATS) cc = cold[R.child_off + sub];
    }
    const int my_action = has ? static_cast<int>(c.meta & META_ACTION_MASK) : -1;
    // which actions own an edge: squares/columns 0..63 in a mask, action 64 (pass) separately
    const unsigned long long low = group_ballot<L>(has && my_action < 64, lane);
    unsigned long long present = 0;
    for (unsigned long long m = low; m; m &= m - 1) {
        const int e = __ffsll(m) - 1;
        present |= 1ull << __shfl(my_action, e, L);
    }
    const bool pass_present = group_ballot<L>(has && my_action == 64, lane) != 0;
    if (!live) return;
    const bool exists = has && (c.meta & META_EXISTS);

    if (!STATS) {
        int32_t *o = counts + static_cast<size_t>(t) * A;
        if (has) o[my_action] = exists ? c.n_visits : 0;
        for (int a = sub; a < A; a += L) {
            const bool pres = a < 64 ? ((present >> a) & 1ull) : pass_present;
            if (!pres) o[a] = 0;
        }
        return;
    }
    float *o = stats + static_cast<size_t>(t) * G::STATS;
    const float u3 = 1.f / 3;
    if (sub == 0) {
        const ColdRec rc = cold[root];
        const bool hv = R.n_visits != 0;
        const float inv = hv ? 1.0f / static_cast<float>(R.n_visits) : 0.0f;
        const float d = hv ? rc.w_draw * inv : u3;
        const float p1 = hv ? R.w_p1 * inv : u3;
        const float p2 = hv ? R.w_p2 * inv : u3;
        o[0] = static_cast<float>(R.n_visits);
        o[1] = (R.meta & META_TURN_P1) ? (p1 - p2) : (p2 - p1);
        o[2] = mean_m(R.n_visits, R.m_sum);
        o[3] = d; o[4] = p1; o[5] = p2;
    }
    if (has) {
        float *slot = o + 6 + my_action * 8;
        const bool hv = exists && c.n_visits != 0;
        const float inv = hv ? 1.0f / static_cast<float>(c.n_visits) : 0.0f;
        const float d = exists ? (hv ? cc.w_draw * inv : u3) : 0.0f;
        const float p1 = exists ? (hv ? c.w_p1 * inv : u3) : 0.0f;
        const float p2 = exists ? (hv ? c.w_p2 * inv : u3) : 0.0f;
        float mm = exists ? mean_m(c.n_visits, c.m_sum) : 0.0f;
        if (G::AUX_NEGATE && exists) mm = -mm;                        // MCTS.h:662-664
        slot[0] = exists ? static_cast<float>(c.n_visits) : 0.0f;
        slot[1] = exists ? ((c.meta & META_TURN_P1) ? (p1 - p2) : (p2 - p1)) : 0.0f;
        slot[2] = c.prior;
        slot[3] = cc.noise;
        slot[4] = mm;
        slot[5] = d; slot[6] = p1; slot[7] = p2;
    }
    for (int a = sub; a < A; a += L) {
        const bool pres = a < 64 ? ((present >> a) & 1ull) : pass_present;
        if (!pres) {
            float *slot = o + 6 + a * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) slot[q] = 0.0f;
        }
    }
}

// ------------------------------------------------------------------ random-playout evaluator

// RolloutEvaluator::evaluate_single (RolloutEvaluator.h:23-48) for the leaves of one plain
// selection: uniform policy (all ones), value = result of a uniformly random playout from
// the leaf, auxiliary value 0.  One thread per tree; moves come from the device generator.
template <class G>
__global__ void __launch_bounds__(256) k_rollout(LeafBuf lf, SearchParams p, int B, float *policy, float *d,
                                                 float *p1w, float *p2w, float *ml, uint8_t *is_term)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    const uint8_t fl = lf.flags[t];
    const bool term = (fl & LEAF_TERMINAL) != 0;
    int code = (fl >> LEAF_RESULT_SHIFT) & 3;
    if (!term) {
        GameState s;
        s.bb0 = lf.bb0[t]; s.bb1 = lf.bb1[t]; s.turn = lf.turn[t]; s.aux = lf.aux[t];
        DevRng g(p.seed, *p.call_ptr, static_cast<uint64_t>(t), 3);
        int res = G::result(s);
        for (int ply = 0; res < 0 && ply < 4 * G::CELLS; ++ply) {
            const int nv = G::num_valid(s);
            if (nv <= 0) break;
            const int pick = static_cast<int>((static_cast<uint64_t>(g.next()) * static_cast<uint64_t>(nv)) >> 32);
            G::step(s, G::nth_valid(s, pick));
            res = G::result(s);
        }
        code = res < 0 ? 0 : res;
    }
    is_term[t] = term ? 1 : 0;
    d[t] = code == 0 ? 1.0f : 0.0f;
    p1w[t] = code == 1 ? 1.0f : 0.0f;
    p2w[t] = code == 2 ? 1.0f : 0.0f;
    ml[t] = 0.0f;
    for (int a = 0; a < G::ACTIONS; ++a) policy[static_cast<size_t>(t) * G::ACTIONS + a] = term ? 0.0f : 1.0f;
}

// ------------------------------------------------------------------ batched game step

// step + result (Connect4.h:159-203 / Othello.h:206-258) on HBM-resident positions.  `aux` is the
// game's small integer carried from ply to ply (Othello: consecutive passes - the GAME remembers
// them although a tree forgets them at every import); nullptr: derived as an import derives it.
template <class G>
__global__ void __launch_bounds__(256) k_game_step(uint64_t *bb0, uint64_t *bb1, int32_t *turns, int32_t *aux,
                                                   const int32_t *actions, uint8_t *done, int32_t *winner,
                                                   int64_t n, int reset_finished)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = actions[i];
    if (a < 0 || a >= G::ACTIONS) { done[i] = 0; winner[i] = 0; return; }
    GameState s;
    s.bb0 = bb0[i]; s.bb1 = bb1[i]; s.turn = turns[i];
    s.aux = aux != nullptr ? aux[i] : G::root_aux(s.bb0, s.bb1);
    G::step(s, a);
    const int res = G::result(s);
    const bool fin = res >= 0;
    done[i] = fin ? 1 : 0;
    winner[i] = res == 1 ? 1 : (res == 2 ? -1 : 0);
    if (fin && reset_finished) G::start(s);
    bb0[i] = s.bb0; bb1[i] = s.bb1; turns[i] = s.turn;
    if (aux != nullptr) aux[i] = s.aux;
}

// legal actions of HBM-resident positions, one byte per action (env_common.h `valid_mask()`)
template <class G>
__global__ void __launch_bounds__(256) k_game_valid_mask(const uint64_t *bb0, const uint64_t *bb1, const int32_t *turns,
                                                         const int32_t *aux, uint8_t *mask, int64_t n)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t i = gid / G::ACTIONS;
    const int a = static_cast<int>(gid - i * G::ACTIONS);
    if (i >= n) return;
    GameState s;
    s.bb0 = bb0[i]; s.bb1 = bb1[i]; s.turn = turns[i];
    s.aux = aux != nullptr ? aux[i] : G::root_aux(s.bb0, s.bb1);
    mask[gid] = G::valid_in_frame(s, 0, a) ? 1 : 0;
}

__global__ void k_bump_call(uint64_t *ctr) { *ctr += 1; }

inline unsigned grid_for(int B, int trees_per_wg) { return static_cast<unsigned>((B + trees_per_wg - 1) / trees_per_wg); }

}  // namespace

// ------------------------------------------------------------------ launchers

// Trees per wavefront for Connect4's two heavy kernels (AZ_TREES_PER_WAVE, default 8 = all lanes
// busy).  Measured on MI355X at 8192 trees, K=4 (hash evaluator): 8 -> 75.6 us per selection
// launch, 4 -> 96.4, 2 -> 142.4, 1 -> 236.5: the kernels are bound by instruction issue and
// dependent-instruction latency, not by memory latency, so spreading the trees over more
// wavefronts only multiplies the instruction count.
int trees_per_wave(int lanes)
{
    static const int v = [] {
        const char *e = getenv("AZ_TREES_PER_WAVE");
        int t = e ? atoi(e) : 8;
        return (t == 1 || t == 2 || t == 4 || t == 8) ? t : 8;
    }();
    const int max_tpw = WAVE / lanes;
    return v < max_tpw ? v : max_tpw;
}

#define AZ_DISPATCH(game, ...)                                                     \
    do {                                                                           \
        if ((game) == Connect4Dev::GAME_ID) { using G = Connect4Dev; __VA_ARGS__; } \
        else { using G = OthelloDev; __VA_ARGS__; }                                 \
    } while (0)

void launch_import(int game, const int8_t *boards, const int32_t *turns, RootState rs, int B, hipStream_t s)
{
    AZ_DISPATCH(game, hipLaunchKernelGGL(k_import<G>, dim3((B + 255) / 256), dim3(256), 0, s, boards, turns, rs, B));
}

void launch_set_roots(int game, const uint64_t *bb0, const uint64_t *bb1, const int32_t *turns, RootState rs,
                      int B, hipStream_t s)
{
    AZ_DISPATCH(game, hipLaunchKernelGGL(k_set_roots<G>, dim3((B + 255) / 256), dim3(256), 0, s, bb0, bb1, turns, rs, B));
}

void launch_bump_call(uint64_t *call_ctr, hipStream_t s)
{
    hipLaunchKernelGGL(k_bump_call, dim3(1), dim3(1), 0, s, call_ctr);
}

const char *launch_select(int game, TreeArena ar, RootState rs, LeafBuf lf, SearchParams p, int K, bool vl,
                          unsigned long long *counters, hipStream_t s, uint64_t *bump_call, int64_t *zero)
{
    // AZ_SELECT_VARIANT: 0 = k_select (the first kernel, every game), 1 = k_select8 (Connect4) for every launch,
    // 3 (default) = k_select8x4 for virtual-loss batches of 2..4 descents, k_select8 for the rest
    static const int variant = [] { const char *e = getenv("AZ_SELECT_VARIANT"); return e ? atoi(e) : 3; }();
    if (game == Connect4Dev::GAME_ID && variant >= 3 && vl && K >= 2 && K <= 4) {
        hipLaunchKernelGGL(k_select8x4, dim3(grid_for(ar.B, 2)), dim3(WAVE), 0, s, ar, rs, lf, p, K, counters, bump_call,
                           reinterpret_cast<long long *>(zero));
        return "k_select8x4";
    }
    if (game == Connect4Dev::GAME_ID && variant >= 1) {
        const int tpw = trees_per_wave(Connect4Dev::LANES);
        const dim3 grid(grid_for(ar.B, tpw)), block(WAVE);
        if (vl) hipLaunchKernelGGL((k_select8<true>), grid, block, 0, s, ar, rs, lf, p, K, tpw, counters, bump_call, reinterpret_cast<long long *>(zero));
        else    hipLaunchKernelGGL((k_select8<false>), grid, block, 0, s, ar, rs, lf, p, K, tpw, counters, bump_call, reinterpret_cast<long long *>(zero));
        return vl ? "k_select8<true>" : "k_select8<false>";
    }
    AZ_DISPATCH(game, {
        const int tpw = trees_per_wave(G::LANES);
        const dim3 grid(grid_for(ar.B, tpw)), block(WAVE);
        if (vl) hipLaunchKernelGGL((k_select<G, true>), grid, block, 0, s, ar, rs, lf, p, K, tpw, counters, bump_call, reinterpret_cast<long long *>(zero));
        else    hipLaunchKernelGGL((k_select<G, false>), grid, block, 0, s, ar, rs, lf, p, K, tpw, counters, bump_call, reinterpret_cast<long long *>(zero));
    });
    if (game == Connect4Dev::GAME_ID) return vl ? "k_select<Connect4Dev,true>" : "k_select<Connect4Dev,false>";
    return vl ? "k_select<OthelloDev,true>" : "k_select<OthelloDev,false>";
}

void launch_backprop(int game, TreeArena ar, LeafBuf lf, SearchParams p, int K, bool vl, bool fused,
                     EvalIn in, unsigned long long *counters, int *err, hipStream_t s)
{
    static const bool v1 = getenv("AZ_BACKPROP_V1") != nullptr && getenv("AZ_BACKPROP_V1")[0] == '1';
    static const bool no_spread = getenv("AZ_BACKPROP_SPREAD") != nullptr && getenv("AZ_BACKPROP_SPREAD")[0] == '0';
    if (game == Connect4Dev::GAME_ID && vl && K >= 2 && K <= 8 && !v1 && !no_spread) {
        using G = Connect4Dev;
        const int kg = K <= 2 ? 2 : (K <= 4 ? 4 : 8);
        constexpr int W = 1;                                        // wavefronts per workgroup (4: 19.6 us against 19.1)
        const dim3 grid(grid_for(ar.B, W * WAVE / (G::LANES * kg))), block(W * WAVE);
        if (kg == 2) {
            if (fused) hipLaunchKernelGGL((k_backprop_spread<G, true, 2, W>), grid, block, 0, s, ar, lf, p, K, in, counters, err);
            else       hipLaunchKernelGGL((k_backprop_spread<G, false, 2, W>), grid, block, 0, s, ar, lf, p, K, in, counters, err);
        } else if (kg == 4) {
            if (fused) hipLaunchKernelGGL((k_backprop_spread<G, true, 4, W>), grid, block, 0, s, ar, lf, p, K, in, counters, err);
            else       hipLaunchKernelGGL((k_backprop_spread<G, false, 4, W>), grid, block, 0, s, ar, lf, p, K, in, counters, err);
        } else {
            if (fused) hipLaunchKernelGGL((k_backprop_spread<G, true, 8, W>), grid, block, 0, s, ar, lf, p, K, in, counters, err);
            else       hipLaunchKernelGGL((k_backprop_spread<G, false, 8, W>), grid, block, 0, s, ar, lf, p, K, in, counters, err);
        }
        return;
    }
    if (K <= 4 && !v1) {
        AZ_DISPATCH(game, {
            const int tpw = trees_per_wave(G::LANES);
            const dim3 grid(grid_for(ar.B, tpw)), block(WAVE);
            if (vl && fused)        hipLaunchKernelGGL((k_backprop_batched<G, true, true, 4>), grid, block, 0, s, ar, lf, p, K, tpw, in, counters, err);
            else if (vl && !fused)  hipLaunchKernelGGL((k_backprop_batched<G, true, false, 4>), grid, block, 0, s, ar, lf, p, K, tpw, in, counters, err);
            else if (!vl && fused)  hipLaunchKernelGGL((k_backprop_batched<G, false, true, 1>), grid, block, 0, s, ar, lf, p, K, tpw, in, counters, err);
            else                    hipLaunchKernelGGL((k_backprop_batched<G, false, false, 1>), grid, block, 0, s, ar, lf, p, K, tpw, in, counters, err);
        });
        return;
    }
    AZ_DISPATCH(game, {
        const int tpw = trees_per_wave(G::LANES);
        const dim3 grid(grid_for(ar.B, tpw)), block(WAVE);
        if (vl && fused)        hipLaunchKernelGGL((k_backprop<G, true, true>), grid, block, 0, s, ar, lf, p, K, tpw, in, counters, err);
        else if (vl && !fused)  hipLaunchKernelGGL((k_backprop<G, true, false>), grid, block, 0, s, ar, lf, p, K, tpw, in, counters, err);
        else if (!vl && fused)  hipLaunchKernelGGL((k_backprop<G, false, true>), grid, block, 0, s, ar, lf, p, K, tpw, in, counters, err);
        else                    hipLaunchKernelGGL((k_backprop<G, false, false>), grid, block, 0, s, ar, lf, p, K, tpw, in, counters, err);
    });
}

void launch_remove_vl(int game, TreeArena ar, LeafBuf lf, SearchParams p, int K, int strideK, hipStream_t s)
{
    AZ_DISPATCH(game, hipLaunchKernelGGL(k_remove_vl<G>, dim3(grid_for(ar.B, WAVE / G::LANES)), dim3(WAVE), 0, s, ar,
                                         lf, p, K, strideK));
}

void launch_export(int game, LeafBuf lf, SearchParams p, int n_leaves, bool gen_sym, int8_t *boards,
                   uint8_t *valid_mask, float *features, hipStream_t s)
{
    AZ_DISPATCH(game, {
        const int64_t threads = static_cast<int64_t>(n_leaves) * G::CELLS;
        hipLaunchKernelGGL(k_export<G>, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, s, lf, p,
                           n_leaves, gen_sym ? 1 : 0, boards, valid_mask, features);
    });
}

void launch_leaf_prep(int game, LeafBuf lf, SearchParams p, int n_leaves, bool gen_sym, uint8_t *valid_mask, int32_t *idx,
                      int64_t *count, int *err, hipStream_t s)
{
    AZ_DISPATCH(game, hipLaunchKernelGGL(k_leaf_prep<G>, dim3((n_leaves + 1023) / 1024), dim3(1024), 0, s, lf, p, n_leaves,
                                         gen_sym ? 1 : 0, valid_mask, idx, reinterpret_cast<long long *>(count), err));
}

void launch_prune(int game, TreeArena ar, SearchParams p, const int32_t *actions, int32_t *noise_req,
                  bool dev_noise, hipStream_t s, const float *replay_noise, int *max_live, int *err, int compact_above)
{
    AZ_DISPATCH(game, hipLaunchKernelGGL(k_prune<G>, dim3(static_cast<unsigned>(ar.B)), dim3(WAVE), 0, s, ar, p,
                                         actions, noise_req, dev_noise ? 1 : 0, replay_noise, max_live, err, compact_above));
}

void launch_apply_noise(int game, TreeArena ar, const int32_t *noise_req, const float *noise, hipStream_t s)
{
    AZ_DISPATCH(game, hipLaunchKernelGGL(k_apply_noise<G>, dim3(grid_for(ar.B, WAVE / G::LANES)), dim3(WAVE), 0, s, ar,
                                         noise_req, noise));
}

void launch_reset_masked(TreeArena ar, const uint8_t *mask, hipStream_t s)
{
    hipLaunchKernelGGL(k_reset_masked, dim3((ar.B + 255) / 256), dim3(256), 0, s, ar, mask);
}

void launch_counts(int game, TreeArena ar, int32_t *counts, hipStream_t s)
{
    AZ_DISPATCH(game, hipLaunchKernelGGL((k_root_query<G, false>), dim3(grid_for(ar.B, WAVE / G::LANES)), dim3(WAVE), 0,
                                         s, ar, counts, static_cast<float *>(nullptr)));
}

void launch_root_stats(int game, TreeArena ar, float *stats, hipStream_t s)
{
    AZ_DISPATCH(game, hipLaunchKernelGGL((k_root_query<G, true>), dim3(grid_for(ar.B, WAVE / G::LANES)), dim3(WAVE), 0,
                                         s, ar, static_cast<int32_t *>(nullptr), stats));
}

void launch_init_trees(TreeArena ar, hipStream_t s)
{
    hipLaunchKernelGGL(k_init_trees, dim3((ar.B + 255) / 256), dim3(256), 0, s, ar);
}

void launch_rollout(int game, LeafBuf lf, SearchParams p, int B, float *policy, float *d, float *p1w, float *p2w,
                    float *ml, uint8_t *is_term, hipStream_t s)
{
    AZ_DISPATCH(game, hipLaunchKernelGGL(k_rollout<G>, dim3((B + 255) / 256), dim3(256), 0, s, lf, p, B, policy, d, p1w,
                                         p2w, ml, is_term));
}

void launch_game_step(int game, uint64_t *bb0, uint64_t *bb1, int32_t *turns, int32_t *aux, const int32_t *actions,
                      uint8_t *done, int32_t *winner, int64_t n, bool reset_finished, hipStream_t s)
{
    AZ_DISPATCH(game, hipLaunchKernelGGL(k_game_step<G>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s,
                                         bb0, bb1, turns, aux, actions, done, winner, n, reset_finished ? 1 : 0));
}

void launch_game_valid_mask(int game, const uint64_t *bb0, const uint64_t *bb1, const int32_t *turns, const int32_t *aux,
                            uint8_t *mask, int64_t n, hipStream_t s)
{
    AZ_DISPATCH(game, hipLaunchKernelGGL(k_game_valid_mask<G>, dim3(static_cast<unsigned>((n * G::ACTIONS + 255) / 256)),
                                         dim3(256), 0, s, bb0, bb1, turns, aux, mask, n));
}

}  // namespace az
