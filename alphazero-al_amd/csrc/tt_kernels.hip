// tt_kernels.hip - device transposition table of evaluator outputs, templated over the game.
//
// What the reference keeps in a Python OrderedDict (src/Cache.py:5-58, used per leaf by
// src/MCTS_cpp.py:146-189 and 298-339): key = the leaf position AS THE EVALUATOR SEES IT (the
// symmetrised board) + side to move, value = (policy[A], relative wdl[3], auxiliary value).  Here:
// a flat open-addressing table in HBM, entries of whole 64-byte lines (Connect4: one, Othello: five),
// 4-entry buckets.
//
//   lookup   one thread per leaf: hit -> the cached 11 floats go straight into the arrays the
//            backup kernel reads; miss -> the leaf's index is appended to a compact list whose
//            length stays on the device (the evaluator kernels size their work from it).
//   insert   one thread per miss, after the evaluator ran: bucket slot = same key, else empty,
//            else the entry with the oldest stamp (approximate LRU).
//
// Entries are written with no lock.  The stored key is XORed with a checksum of the value, so an
// entry torn by two concurrent writers fails the comparison and reads as a miss: a lookup never
// returns a value that was not evaluated for exactly that key.  Because the evaluator kernels
// compute every sample independently of its batch (bit-identical outputs for identical inputs),
// a search with the table visits exactly the nodes it visits without it.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "board_sym.h"
#include "games.h"
#include "kernels.h"

namespace az {
namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// Geometry of a game's table: values = policy[A], relative wdl[3], auxiliary value; an entry is the key
// (two u64, XORed with a checksum of the value), the values, a stamp, padded to whole 64-byte lines.
template <class G>
struct Tt {
    static constexpr int A = G::ACTIONS;
    static constexpr int NV = A + 4;
    static constexpr int BYTES = (16 + 4 * NV + 4 + 63) / 64 * 64;      // Connect4 64, Othello 320
    __device__ static uint8_t *entry(const TtTable &t, uint64_t i) { return reinterpret_cast<uint8_t *>(t.e) + i * BYTES; }
    __device__ static uint64_t *key(uint8_t *e) { return reinterpret_cast<uint64_t *>(e); }
    __device__ static float *val(uint8_t *e) { return reinterpret_cast<float *>(e + 16); }
    __device__ static uint32_t *stamp(uint8_t *e) { return reinterpret_cast<uint32_t *>(e + 16 + 4 * NV); }
    __device__ static uint64_t value_sum(const float *v)
    {
        uint64_t c = 0x9e3779b97f4a7c15ull;
        for (int i = 0; i < NV; ++i) c = mix64(c ^ __float_as_uint(v[i])) + i;
        return c;
    }
    // the evaluator's view of a leaf: stones of the side to move, stones of the opponent, who moves
    __device__ static void leaf_key(const LeafBuf &lf, int64_t leaf, uint64_t &k0, uint64_t &k1)
    {
        uint64_t p1 = lf.bb0[leaf], p2 = lf.bb1[leaf];
        const int sym = lf.sym[leaf];
        if (G::GAME_ID == 0) {
            if (sym) { p1 = mirror_columns(p1); p2 = mirror_columns(p2); }
        } else {
            p1 = othello_sym(p1, sym); p2 = othello_sym(p2, sym);
        }
        const bool first = lf.turn[leaf] > 0;
        if (G::GAME_ID == 0) {
            k0 = (first ? p1 : p2) | (first ? (1ull << 63) : 0ull);
            k1 = (first ? p2 : p1) | (1ull << 62);                       // never zero: an empty entry matches nothing
        } else {
            // all 64 bits are squares: the side to move goes into the ORDER of the words, and the stamp word's
            // neighbour - the values - is what tells an empty entry apart (an empty entry has k0 = k1 = 0 and
            // stamp 0; a resident one always has a non-zero checksum word, see lookup)
            k0 = first ? p1 : ~p2;
            k1 = first ? p2 : ~p1;
        }
    }
};

template <class G>
__global__ void __launch_bounds__(256) k_tt_lookup(LeafBuf lf, int n_leaves, TtTable t, const uint64_t *clock,
                                                   float *probs, float *wdl, float *ml, int32_t *miss_idx,
                                                   int64_t *miss_count, uint64_t *keys, int *err)
{
    using T = Tt<G>;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool in = i < n_leaves;
    bool miss = false, hit = false;
    if (in && !(lf.flags[i] & LEAF_TERMINAL)) {           // terminal leaves take their result from the game
        uint64_t k0, k1;
        T::leaf_key(lf, i, k0, k1);
        keys[2 * i] = k0;
        keys[2 * i + 1] = k1;
        const uint64_t bucket = (mix64(k0 ^ mix64(k1)) & t.mask) & ~3ull;
        miss = true;
        for (int j = 0; j < 4 && miss; ++j) {
            uint8_t *e = T::entry(t, bucket + j);
            if (*T::stamp(e) == 0) continue;              // never written
            const float *v = T::val(e);
            const uint64_t c = T::value_sum(v);
            if ((T::key(e)[0] ^ c) == k0 && (T::key(e)[1] ^ (c << 17 | c >> 47)) == k1) {
                miss = false;
                hit = true;
                for (int a = 0; a < T::A; ++a) probs[i * T::A + a] = v[a];
                wdl[i * 3 + 0] = v[T::A]; wdl[i * 3 + 1] = v[T::A + 1]; wdl[i * 3 + 2] = v[T::A + 2];
                ml[i] = v[T::A + 3];
                *T::stamp(e) = static_cast<uint32_t>(*clock) | 1u;      // recently used (never 0)
            }
        }
    }
    // one atomic per wavefront for the compact list, one for the statistics
    const unsigned long long mm = __ballot(miss), hm = __ballot(hit);
    const int lane = threadIdx.x & 63;
    int64_t base = 0;
    if (lane == 0) {
        if (mm) base = static_cast<int64_t>(atomicAdd(reinterpret_cast<unsigned long long *>(miss_count),
                                                      static_cast<unsigned long long>(__popcll(mm))));
        if (mm | hm) {
            atomicAdd(&t.stats[0], static_cast<unsigned long long>(__popcll(mm) + __popcll(hm)));
            if (hm) atomicAdd(&t.stats[1], static_cast<unsigned long long>(__popcll(hm)));
        }
    }
    base = __shfl(base, 0, 64);
    if (miss) {
        // the list holds n_leaves entries: a position outside it (a count that was not cleared) raises
        // the engine's error flag instead of storing anywhere
        const int64_t pos = base + __popcll(mm & ((1ull << lane) - 1));
        if (pos >= 0 && pos < n_leaves) miss_idx[pos] = static_cast<int32_t>(i);
        else atomicOr(err, ERR_LIST_OVERFLOW);
    }
}

template <class G>
__global__ void __launch_bounds__(256) k_tt_insert(TtTable t, int n_leaves, const uint64_t *clock, const int32_t *miss_idx,
                                                   const int64_t *miss_count, const uint64_t *keys, const float *probs,
                                                   const float *wdl, const float *ml)
{
    using T = Tt<G>;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= *miss_count || j >= n_leaves) return;
    const int64_t i = miss_idx[j];
    if (i < 0 || i >= n_leaves) return;
    const uint64_t k0 = keys[2 * i], k1 = keys[2 * i + 1];
    // checksum of the value as it will be stored
    uint64_t c = 0x9e3779b97f4a7c15ull;
    for (int a = 0; a < T::NV; ++a) {
        const float x = a < T::A ? probs[i * T::A + a] : (a < T::A + 3 ? wdl[i * 3 + (a - T::A)] : ml[i]);
        c = mix64(c ^ __float_as_uint(x)) + a;
    }
    const uint64_t e0 = k0 ^ c, e1 = k1 ^ (c << 17 | c >> 47);
    const uint32_t now = static_cast<uint32_t>(*clock) | 1u;
    const uint64_t bucket = (mix64(k0 ^ mix64(k1)) & t.mask) & ~3ull;
    int victim = 0;
    uint32_t oldest = 0;
    bool replaced = true;
    for (int s = 0; s < 4; ++s) {
        uint8_t *cur = T::entry(t, bucket + s);
        const uint32_t st = *T::stamp(cur);
        if (st != 0 && T::key(cur)[0] == e0 && T::key(cur)[1] == e1) { victim = s; replaced = false; break; }   // same key, same value
        const bool empty = st == 0;
        const uint32_t age = empty ? 0xffffffffu : now - st;
        if (s == 0 || age > oldest) { oldest = age; victim = s; replaced = !empty; }
    }
    uint8_t *dst = T::entry(t, bucket + victim);
    T::key(dst)[0] = e0; T::key(dst)[1] = e1;
    float *v = T::val(dst);
    for (int a = 0; a < T::A; ++a) v[a] = probs[i * T::A + a];
    v[T::A] = wdl[i * 3 + 0]; v[T::A + 1] = wdl[i * 3 + 1]; v[T::A + 2] = wdl[i * 3 + 2];
    v[T::A + 3] = ml[i];
    *T::stamp(dst) = now;
    atomicAdd(&t.stats[2], 1ull);
    if (replaced) atomicAdd(&t.stats[3], 1ull);
}

// ---- refresh: every resident key evaluated again (MCTS_cpp.py:361-377 `refresh_cache`, which the
// reference calls after a weight update so that the table keeps its keys and loses no warmth).
// Pass 1 decodes the entries [e0, e0 + n) - the key comes back by XORing with the value's checksum -
// and writes, for every resident one, the evaluator's input exactly as it was when the entry was
// made (the stored frame IS the evaluator's frame): the position (symmetry id 0 from here on), its
// action mask, its row in the compact list.  The evaluator runs on that list.  Pass 2 stores the fresh values under the same
// keys.  An entry torn by an earlier race decodes to an implausible position and is emptied.
__device__ __forceinline__ bool plausible_c4(uint64_t own, uint64_t opp)
{
    constexpr uint64_t BOTTOM = 0x0000040810204081ull;                // bit 0 of every column
    constexpr uint64_t SENT = BOTTOM << 6;                            // bit 6 of every column: never a stone
    constexpr uint64_t BOARD = (1ull << 49) - 1;
    const uint64_t occ = own | opp;
    if ((own & opp) || (occ & SENT) || (occ & ~BOARD)) return false;
    return ((occ + BOTTOM) & occ) == 0;                               // every column filled from the bottom, no gaps
}

// decode one entry's key into the position the evaluator saw (frame of the entry: symmetry id 0 from here on)
template <class G>
__device__ __forceinline__ bool decode_key(uint64_t k0, uint64_t k1, uint64_t &p1, uint64_t &p2, int &turn)
{
    if (G::GAME_ID == 0) {
        const uint64_t own = k0 & ~(1ull << 63), opp = k1 & ~(1ull << 62);
        if ((k1 >> 62) != 1ull || !plausible_c4(own, opp)) return false;
        const bool first = (k0 >> 63) != 0;
        p1 = first ? own : opp; p2 = first ? opp : own; turn = first ? 1 : -1;
        return true;
    }
    // Othello: player +1 to move stores (p1, p2), player -1 stores (~p2, ~p1); stones never overlap, so the
    // plain form has k0 & k1 == 0 and the complemented one does not (a full board is terminal: never stored)
    if ((k0 & k1) == 0) { p1 = k0; p2 = k1; turn = 1; }
    else { p1 = ~k1; p2 = ~k0; turn = -1; }
    return (p1 & p2) == 0 && (p1 | p2) != 0;
}

template <class G>
__global__ void __launch_bounds__(256) k_tt_refresh_gather(TtTable t, uint64_t e0, int n, uint64_t *bb_p1, uint64_t *bb_p2,
                                                           int32_t *turn, int32_t *sym, uint8_t *mask, int32_t *rows,
                                                           int64_t *count, uint64_t *keys)
{
    using T = Tt<G>;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    bool live = false;
    if (j < n) {
        uint8_t *ent = T::entry(t, e0 + j);
        if (*T::stamp(ent) != 0) {
            const uint64_t c = T::value_sum(T::val(ent));
            const uint64_t k0 = T::key(ent)[0] ^ c, k1 = T::key(ent)[1] ^ (c << 17 | c >> 47);
            GameState st;
            st.aux = G::root_aux(0, 0);
            if (decode_key<G>(k0, k1, st.bb0, st.bb1, st.turn)) {
                live = true;
                keys[2 * j] = k0; keys[2 * j + 1] = k1;
                bb_p1[j] = st.bb0; bb_p2[j] = st.bb1; turn[j] = st.turn; sym[j] = 0;
                st.aux = G::root_aux(st.bb0, st.bb1);
                for (int a = 0; a < G::ACTIONS; ++a) mask[j * G::ACTIONS + a] = G::valid_in_frame(st, 0, a) ? 1 : 0;
            } else {
                *T::stamp(ent) = 0;                                   // unreadable: empty it
            }
        }
    }
    const unsigned long long m = __ballot(live);
    const int lane = threadIdx.x & 63;
    long long base = 0;
    if (lane == 0 && m) base = static_cast<long long>(atomicAdd(reinterpret_cast<unsigned long long *>(count),
                                                                static_cast<unsigned long long>(__popcll(m))));
    base = __shfl(base, 0, 64);
    if (live) {
        const long long pos = base + __popcll(m & ((1ull << lane) - 1));
        if (pos >= 0 && pos < n) rows[pos] = static_cast<int32_t>(j);
    }
}

template <class G>
__global__ void __launch_bounds__(256) k_tt_refresh_store(TtTable t, uint64_t e0, int n, const int32_t *rows,
                                                          const int64_t *count, const uint64_t *keys, const float *probs,
                                                          const float *wdl, const float *ml)
{
    using T = Tt<G>;
    const int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q >= *count || q >= n) return;
    const int64_t j = rows[q];
    if (j < 0 || j >= n) return;
    uint8_t *ent = T::entry(t, e0 + j);
    float *v = T::val(ent);
    for (int a = 0; a < T::A; ++a) v[a] = probs[j * T::A + a];
    v[T::A] = wdl[j * 3 + 0]; v[T::A + 1] = wdl[j * 3 + 1]; v[T::A + 2] = wdl[j * 3 + 2];
    v[T::A + 3] = ml[j];
    const uint64_t c = T::value_sum(v);
    T::key(ent)[0] = keys[2 * j] ^ c;
    T::key(ent)[1] = keys[2 * j + 1] ^ (c << 17 | c >> 47);            // the stamp - its age - is not changed by a refresh
}

// The leaves an evaluator has to see: everything but terminal leaves, whose value comes from the
// game (the reference's wrapper calls `predict` on the non-terminal rows only, MCTS_cpp.py:275-297).
// One thread per leaf; the workgroup agrees on its share of the list through LDS, so the global
// counter sees one atomic per workgroup (a single-workgroup scan was tried: 18 us, slower than
// the contended atomics it was meant to avoid).
__global__ void __launch_bounds__(1024) k_live_leaves(LeafBuf lf, int n_leaves, int32_t *idx, int64_t *count, int *err)
{
    __shared__ int s_wave[16];
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + tid;
    const bool livel = i < n_leaves && !(lf.flags[i] & LEAF_TERMINAL);
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
        // bounded by the list's size whatever the counter held (see k_tt_lookup)
        const long long pos = s_base + before + __popcll(m & ((1ull << lane) - 1));
        if (pos >= 0 && pos < n_leaves) idx[pos] = static_cast<int32_t>(i);
        else atomicOr(err, ERR_LIST_OVERFLOW);
    }
}

}  // namespace

void launch_live_leaves(LeafBuf lf, int n_leaves, int32_t *idx, int64_t *count, int *err, hipStream_t s, bool clear_count)
{
    if (clear_count) (void)hipMemsetAsync(count, 0, sizeof(int64_t), s);
    hipLaunchKernelGGL(k_live_leaves, dim3((n_leaves + 1023) / 1024), dim3(1024), 0, s, lf, n_leaves, idx, count, err);
}

void launch_tt_refresh_gather(int game, TtTable t, uint64_t e0, int n, uint64_t *bb_p1, uint64_t *bb_p2, int32_t *turn, int32_t *sym,
                              uint8_t *mask, int32_t *rows, int64_t *count, uint64_t *keys, hipStream_t s)
{
    (void)hipMemsetAsync(count, 0, sizeof(int64_t), s);
    const dim3 grid((n + 255) / 256), block(256);
    if (game == Connect4Dev::GAME_ID)
        hipLaunchKernelGGL(k_tt_refresh_gather<Connect4Dev>, grid, block, 0, s, t, e0, n, bb_p1, bb_p2, turn, sym, mask, rows, count, keys);
    else
        hipLaunchKernelGGL(k_tt_refresh_gather<OthelloDev>, grid, block, 0, s, t, e0, n, bb_p1, bb_p2, turn, sym, mask, rows, count, keys);
}

void launch_tt_refresh_store(int game, TtTable t, uint64_t e0, int n, const int32_t *rows, const int64_t *count, const uint64_t *keys,
                             const float *probs, const float *wdl, const float *ml, hipStream_t s)
{
    const dim3 grid((n + 255) / 256), block(256);
    if (game == Connect4Dev::GAME_ID)
        hipLaunchKernelGGL(k_tt_refresh_store<Connect4Dev>, grid, block, 0, s, t, e0, n, rows, count, keys, probs, wdl, ml);
    else
        hipLaunchKernelGGL(k_tt_refresh_store<OthelloDev>, grid, block, 0, s, t, e0, n, rows, count, keys, probs, wdl, ml);
}

size_t tt_entry_bytes(int game) { return game == Connect4Dev::GAME_ID ? Tt<Connect4Dev>::BYTES : Tt<OthelloDev>::BYTES; }

void launch_tt_lookup(int game, LeafBuf lf, int n_leaves, TtTable t, const uint64_t *clock, float *probs, float *wdl, float *ml,
                      int32_t *miss_idx, int64_t *miss_count, uint64_t *keys, int *err, hipStream_t s)
{
    (void)hipMemsetAsync(miss_count, 0, sizeof(int64_t), s);
    const dim3 grid((n_leaves + 255) / 256), block(256);
    if (game == Connect4Dev::GAME_ID)
        hipLaunchKernelGGL(k_tt_lookup<Connect4Dev>, grid, block, 0, s, lf, n_leaves, t, clock, probs, wdl, ml, miss_idx, miss_count, keys, err);
    else
        hipLaunchKernelGGL(k_tt_lookup<OthelloDev>, grid, block, 0, s, lf, n_leaves, t, clock, probs, wdl, ml, miss_idx, miss_count, keys, err);
}

void launch_tt_insert(int game, int n_leaves, TtTable t, const uint64_t *clock, const int32_t *miss_idx, const int64_t *miss_count,
                      const uint64_t *keys, const float *probs, const float *wdl, const float *ml, hipStream_t s)
{
    const dim3 grid((n_leaves + 255) / 256), block(256);
    if (game == Connect4Dev::GAME_ID)
        hipLaunchKernelGGL(k_tt_insert<Connect4Dev>, grid, block, 0, s, t, n_leaves, clock, miss_idx, miss_count, keys, probs, wdl, ml);
    else
        hipLaunchKernelGGL(k_tt_insert<OthelloDev>, grid, block, 0, s, t, n_leaves, clock, miss_idx, miss_count, keys, probs, wdl, ml);
}

}  // namespace az
