// nn_conv.hip - one residual convolution block of the evaluator as ONE MFMA kernel.
//
//   y = [x +] silu( conv3x3( [GroupNorm1(x) * gamma + beta] ) + bias )          (Network.py:27-48,166-170)
//
// on token-layout activations (B, 42, C) bf16.  The reference runs this as GroupNorm ->
// convolution -> (bias) -> SiLU -> add: four kernels and five passes over a 176 MB tensor per
// block at 32768 leaves.  Here a 4-wave workgroup owns a tile of 4 samples (168 tokens = 10.5
// MFMA tiles of 16), keeps their normalised, zero-padded 8x9 images in LDS and computes the
// 3x3 convolution as an implicit GEMM with v_mfma_f32_16x16x32_bf16 in the orientation
//
//   out^T (64 channels x tokens) = W (64 x 9*C_in) . X^T (9*C_in x tokens)
//
//   A operand = weights: wave (mh, th) owns output channels 32*mh.. and keeps its 2 x (K/32)
//               fragments in registers (144 VGPRs at C_in = 64) for the whole kernel;
//   B operand = one ds_read_b128 per lane: 8 input channels of the tap's neighbour cell of
//               token (lane & 15).  Cells are 128 B; the 16-byte chunk index is XORed with
//               (cell & 7), which makes every read group of 16 consecutive cells hit 16
//               distinct bank slots (cdna guide T2) - the padded-stride layout this replaces
//               was 2-way conflicted on every read;
//   C layout  = 4 consecutive output channels of one token per lane: bias + SiLU + residual
//               happen on the accumulators and leave as 8-byte stores, no staging pass.
//
// Two token tiles share each weight fragment (4 MFMAs per 2 LDS reads).  58 KB of LDS and
// <= 256 VGPRs let two workgroups share a CU, so one workgroup's load / GroupNorm / epilogue
// VALU work overlaps the other's MFMA phase; workgroups are persistent over tiles and fetch
// the next tile's activations while the current one is multiplied.  HBM traffic per block:
// read x once, write y once (the residual is re-read from an LDS copy of the raw tile).
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "az_nn.h"

namespace {

constexpr int CELLS = 42, COLS = 7;
// Zero-padded image with a row pitch of 8 cells: the right halo of a row IS the left halo of the
// next one (both zero), so cell p = (r+1)*8 + (c+1) and p & 7 - the swizzle key - depends on the
// column only: the three rows of a 3x3 window are the same address +- 1 KiB.
constexpr int PCOLS = 8, PCELLS = 72;          // 65 cells used
constexpr int TS = 4;                          // samples per tile = wavefronts per workgroup
constexpr int COUT = 64;
constexpr int CELLB = 128;                     // bytes per image cell (C_in 32 uses half of it)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct alignas(16) V8 { uint32_t w[4]; };

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// one v_cvt_pk_bf16_f32 (round to nearest even, NaN preserving)
__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(pk_f32x2{a, b}, pk_bf16x2));
}
// wave-wide sum on the DPP network (the adds carry the lane movement as an operand modifier):
// 8 VALU instructions and a readlane, against 6 x (ds_bpermute + address + add) for xor-shuffles
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_mov<0xB1>(v);           // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);           // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);          // row_half_mirror
    v += dpp_mov<0x140>(v);          // row_mirror: every lane of a 16-lane row holds the row sum
    v += dpp_mov<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v += dpp_mov<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed-f32 arithmetic (v_pk_mul/add/fma_f32): two elements per VALU instruction
__device__ __forceinline__ f32x2 unpack2(uint32_t w) { return f32x2{bf_lo(w), bf_hi(w)}; }
// 16 bytes per lane from global memory straight into LDS at (wave-uniform) lds_off + lane * 16.
// Written as inline assembly on purpose: when the compiler knows that a global_load_lds is in
// flight it puts s_waitcnt vmcnt(0) in front of every later LDS read that might alias it, i.e.
// at the top of the MFMA phase, and the HBM latency this staging is meant to hide is paid in
// full.  The kernel orders the accesses itself (vmcnt(0) + barrier before the buffer is read).
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_off)
{
    uint32_t keep;      // M0 is compiler-reserved: saved and restored inside the statement
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_off) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const void *)p));
}
// workgroup barrier that orders LDS traffic only: __syncthreads() would also wait for this
// wave's outstanding global stores
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// az_nn_debug bit 4: per-wave cycle totals of the phases (s_memtime), read back by az_nn_conv_profile
__device__ unsigned long long g_prof[2048 * 8];

// what the stem needs to build its input tokens itself (EMBED): the evaluator's feature planes and
// the embedding tables of Network.py:226-239
struct EmbedIn {
    const float    *features;          // (rows, 3, 6, 7) relative planes
    const uint16_t *emb_own, *emb_opp; // (32,) bf16
    const uint16_t *pos;               // (42, 32) bf16
    const int32_t  *gather;            // compact sample b shows row gather[b] (NULL: b)
    // features == nullptr: the planes are built from the leaf POSITIONS instead (no feature tensor in HBM):
    // two bitboards (bit = 7 * column + height, Connect4.h:15-29), side to move, symmetry id (1 = mirrored)
    const uint64_t *bb_p1, *bb_p2;
    const int32_t  *turn, *sym;
};

template <int CIN, bool NORM, bool RESID, bool EMBED = false>
__global__ void __launch_bounds__(256, 2) k_conv_block(const uint16_t *x, const uint16_t *w, const uint16_t *bias,
                                                       const uint16_t *gamma, const uint16_t *beta, uint16_t *y,
                                                       int64_t B, float eps, int dbg, const int64_t *batch_dev, EmbedIn em)
{
    static_assert(!EMBED || (CIN == 32 && !NORM && !RESID), "the embedding is fused into the stem only");
    const int64_t rows_total = B;                                     // rows of the feature tensor a gather index may name
    if (batch_dev != nullptr && *batch_dev < B) B = *batch_dev;       // compact batch whose size only the device knows
    constexpr int K = 9 * CIN;
    constexpr int KSTEPS = K / 32;                // 18 (C_in 64) or 9 (C_in 32)
    constexpr int KPT = CIN / 32;                 // k steps per tap
    constexpr int VPC = CIN / 8;                  // 16-byte vectors per input cell
    constexpr int VPS = CELLS * VPC;              // vectors per input sample
    constexpr int PER = (VPS + 63) / 64;          // vectors per lane of a sample's wavefront
    constexpr int SB = VPS * 16;                  // bytes per raw sample
    constexpr int OVPS = CELLS * COUT / 8;        // vectors per output sample
    constexpr int OPER = (OVPS + 63) / 64;
    constexpr int OSB = OVPS * 16;                // bytes per output sample

    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t *img = smem;                                          // TS * PCELLS * CELLB, swizzled
    uint8_t *rawb = smem + TS * PCELLS * CELLB;                   // 2 x TS x SB: raw tiles, double buffered
    uint8_t *outs = rawb + 2 * TS * SB;                           // TS x OSB output tile (unless RESID: in place)
    // (no static __shared__ objects: the image must sit at LDS address 0 for the compiler to
    // fold the window's row displacements into the ds_read offset fields)
    float *s_gam = reinterpret_cast<float *>(outs + (RESID ? 0 : TS * OSB)), *s_bet = s_gam + CIN;
    uint8_t *dump = reinterpret_cast<uint8_t *>(s_gam + 2 * CIN);         // 4 x 8 x 16 B: results of dummy tokens

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mh = wave & 1, th = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- weights: this wave's A fragments, resident for the whole kernel.  MFMA row r of m tile
    // mt stands for output channel 32*mh + 8*(r>>2) + 4*mt + (r&3): with the C layout (rows
    // 4*(lane>>4)+reg) a lane then ends up with EIGHT CONSECUTIVE channels of its token - chunk
    // 4*mh + (lane>>4) - i.e. one 16-byte vector of the output row.
    bf16x8 aw[2][KSTEPS];
    float bia[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int o = mh * 32 + (l15 >> 2) * 8 + mt * 4 + (l15 & 3);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            aw[mt][s] = *reinterpret_cast<const bf16x8 *>(w + static_cast<size_t>(o) * K + s * 32 + l4 * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            bia[mt][r] = __uint_as_float(static_cast<uint32_t>(bias[mh * 32 + l4 * 8 + mt * 4 + r]) << 16);
    }
    if (NORM && tid < CIN) {
        s_gam[tid] = __uint_as_float(static_cast<uint32_t>(gamma[tid]) << 16);
        s_bet[tid] = __uint_as_float(static_cast<uint32_t>(beta[tid]) << 16);
    }
    // ---- zero the padded images (the halo is never written again) and the raw buffers (slots
    // of samples past the batch are never filled)
    {
        V8 z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0;
        constexpr int NV = (TS * PCELLS * CELLB + 2 * TS * SB) / 16;
        for (int i = tid; i < NV; i += 256) reinterpret_cast<V8 *>(smem)[i] = z;
    }
    __syncthreads();

    // Raw tiles go from HBM straight into LDS (global_load_lds_dwordx4: no registers held while
    // the previous tile is multiplied).  Wave s stages sample s; an instruction fills 1 KiB in
    // lane order, so slot j of a sample holds cell j / VPC; within a 64-channel cell the chunk
    // order is XORed with (cell & 7) through the SOURCE address.  That is also the layout of the
    // output tile, so the residual block updates the raw tile in place, and it leaves every
    // lane with one fixed channel chunk in P1.
    auto swz_in = [](int cell) { return VPC == 8 ? (cell & 7) : 0; };
    const uint32_t raw_lds = lds_addr(rawb);
    if (EMBED) {                                   // the raw buffers are free: the position table lives there
        for (int i = tid; i < CELLS * CIN / 8; i += 256)
            reinterpret_cast<V8 *>(rawb)[i] = reinterpret_cast<const V8 *>(em.pos)[i];
        __syncthreads();
    }
    auto stage_tile = [&](int64_t tile, int buf, int lane) {
        const int64_t b = tile * TS + wave;
        if (EMBED || b >= B) return;
        const uint16_t *xs = x + b * (CELLS * CIN);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int s = lane + 64 * i;
            if (s < VPS) {
                const int cell = s / VPC;
                glds16(xs + (cell * VPC + ((s % VPC) ^ swz_in(cell))) * 8, raw_lds + (buf * TS + wave) * SB + i * 1024);
            }
        }
    };

    const int64_t ntiles = (B + TS - 1) / TS;
    // EMBED: the two planes of this wave's sample at this lane's cells, one tile ahead
    float pl_own[PER], pl_opp[PER];
    auto load_planes = [&](int64_t tile, int lane) {
        const int64_t b = tile * TS + wave;
        const bool live = b < B;
        int64_t row = !live ? 0 : (em.gather != nullptr ? em.gather[b] : b);
        if (row < 0 || row >= rows_total) row = 0;                    // never dereference an index outside the rows
        if (em.features != nullptr) {
            const float *fs = em.features + row * (3 * CELLS);
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int s = lane + 64 * i;
                const bool ok = live && s < VPS;
                pl_own[i] = ok ? fs[s / VPC] : 0.0f;
                pl_opp[i] = ok ? fs[CELLS + s / VPC] : 0.0f;
            }
        } else {
            // the planes MCTS_cpp.py:15-20 builds from the (symmetrised) grid, straight from the bitboards
            const bool p1 = em.turn[row] > 0, mir = em.sym[row] != 0;
            const uint64_t own = p1 ? em.bb_p1[row] : em.bb_p2[row], opp = p1 ? em.bb_p2[row] : em.bb_p1[row];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int s = lane + 64 * i;
                const bool ok = live && s < VPS;
                const int cell = ok ? s / VPC : 0;
                const int r = cell / COLS, c = cell - r * COLS;
                const int bit = (mir ? COLS - 1 - c : c) * 7 + (5 - r);
                pl_own[i] = (ok && ((own >> bit) & 1ull)) ? 1.0f : 0.0f;
                pl_opp[i] = (ok && ((opp >> bit) & 1ull)) ? 1.0f : 0.0f;
            }
        }
    };
    if (EMBED && static_cast<int64_t>(blockIdx.x) < ntiles) load_planes(blockIdx.x, lane);
    if (static_cast<int64_t>(blockIdx.x) < ntiles) stage_tile(blockIdx.x, 0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    int par = 0;
    const bool prof = dbg & 16;
    unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, tl = prof ? __builtin_amdgcn_s_memtime() : 0;
    auto stamp = [&](int k) {
        if (prof) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tp[k] += now - tl;
            tl = now;
        }
    };
    // where this lane's vectors of its sample go in the padded, swizzled image: six dividing address chains that
    // do not depend on the tile - kept in registers across the tile loop (the rest of P1's and P3's lane
    // arithmetic is cheap and is recomputed per tile, see lane_t below)
    uint32_t img_off[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int sl = lane + 64 * i;
        const int cell = sl / VPC;
        const int r = cell / COLS, c = cell - r * COLS;
        const int p = (r + 1) * PCOLS + (c + 1);
        const int ck = (lane % VPC) ^ swz_in(lane / VPC);
        img_off[i] = static_cast<uint32_t>((wave * PCELLS + p) * CELLB + ((ck ^ (p & 7)) << 4));
    }
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
        const int64_t b0 = tile * TS;
        uint8_t *rawt = rawb + par * TS * SB;
        uint8_t *outt = RESID ? rawt : outs;
        // Opaque copy of the lane id: everything P1, P3 and the staging derive from it is
        // recomputed per tile (a few VALU ops) instead of being hoisted out of the tile loop,
        // where ~40 loop-invariant addresses would push the resident weights into scratch.
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));

        // ---- P1: GroupNorm statistics of this wave's sample (one pass, fp32), then the
        // normalised vectors go to the padded image
        {
            const int ck = (lane_t % VPC) ^ swz_in(lane_t / VPC);   // the channel chunk of every slot this lane owns
            V8 raw[PER];
            if (EMBED) {
                // tokens = pos[cell] + own * emb_own + opp * emb_opp (fp32 on the bf16 tables, rounded once:
                // the arithmetic of k_embed, nn_kernels.hip)
                const V8 eo = *reinterpret_cast<const V8 *>(em.emb_own + ck * 8), ep = *reinterpret_cast<const V8 *>(em.emb_opp + ck * 8);
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int s = lane_t + 64 * i;
                    if (s < VPS) {
                        const V8 ps = *reinterpret_cast<const V8 *>(rawb + s * 16);
                        const f32x2 own = {pl_own[i], pl_own[i]}, opp = {pl_opp[i], pl_opp[i]};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            // p + own*a + opp*o evaluated left to right, as k_embed does
                            const f32x2 v = unpack2(ps.w[q]) + own * unpack2(eo.w[q]) + opp * unpack2(ep.w[q]);
                            raw[i].w[q] = pack2(v.x, v.y);
                        }
                    } else {
                        raw[i].w[0] = raw[i].w[1] = raw[i].w[2] = raw[i].w[3] = 0;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int s = lane_t + 64 * i;
                    if (s < VPS) raw[i] = *reinterpret_cast<const V8 *>(rawt + wave * SB + s * 16);
                    else raw[i].w[0] = raw[i].w[1] = raw[i].w[2] = raw[i].w[3] = 0;
                }
            }
            f32x2 sc[4], sh[4];
            if (NORM) {
                f32x2 sum2 = {0.0f, 0.0f}, sq2 = {0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < PER; ++i) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x2 f = unpack2(raw[i].w[q]);              // vectors past the sample are zero
                        sum2 += f;
                        sq2 = __builtin_elementwise_fma(f, f, sq2);
                    }
                }
                const float sum = wave_sum(sum2.x + sum2.y), sq = wave_sum(sq2.x + sq2.y);
                const float mean = sum * (1.0f / (CELLS * CIN));
                const float var = fmaxf(sq * (1.0f / (CELLS * CIN)) - mean * mean, 0.0f);
                const float rstd = rsqrtf(var + eps);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 gq = *reinterpret_cast<const f32x2 *>(&s_gam[ck * 8 + 2 * q]);
                    const f32x2 bq = *reinterpret_cast<const f32x2 *>(&s_bet[ck * 8 + 2 * q]);
                    sc[q] = gq * f32x2{rstd, rstd};
                    sh[q] = bq - sc[q] * f32x2{mean, mean};
                }
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int s = lane_t + 64 * i;
                if (s < VPS) {
                    V8 out = raw[i];
                    if (NORM) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x2 f = __builtin_elementwise_fma(unpack2(raw[i].w[q]), sc[q], sh[q]);
                            out.w[q] = pack2(f.x, f.y);
                        }
                    }
                    *reinterpret_cast<V8 *>(img + img_off[i]) = out;
                }
            }
        }
        stamp(0);
        lds_barrier();
        stamp(1);
        // the next tile travels from HBM into the other raw buffer while this one is multiplied
        // (every wave finished reading that buffer - P3 of the previous tile - before the barrier)
        if (tile + gridDim.x < ntiles) {
            stage_tile(tile + gridDim.x, par ^ 1, lane_t);
            if (EMBED) load_planes(tile + gridDim.x, lane_t);
        }

        // ---- P2: wave (mh, th) owns samples 2*th and 2*th+1 of the tile, each as three token
        // tiles of 16 over a 6 x 8 token grid: token t sits at image cell t + 9, its 8th column is
        // the halo cell (a dummy token whose result is dropped), so a tile's cells are
        // consecutive - every B-fragment read is bank-conflict free and needs no division.
        // The loop is software pipelined by one tile: the epilogue of tile i-1 (VALU) is issued
        // inside the MFMA stream of tile i, where an MFMA leaves half of its 16 issue cycles free.
        // Token tile `it` of this wave's pair of samples starts (it / 3) * PCELLS + (it % 3) * 16 cells after
        // tile 0 - a multiple of 8 either way, so the swizzle key (cell & 7) of a lane's cell is the same in
        // every tile and its byte offset differs by a CONSTANT: the three column offsets are computed once
        // per 4-sample tile and every fragment read carries tile and row displacement in its offset field.
        uint32_t col[3][KPT];
        {
            const uint32_t pc = static_cast<uint32_t>(2 * th * PCELLS + l15 + 9 - PCOLS - 1);   // row above, dx = -1
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const uint32_t p = pc + d;
                const uint32_t o = (p * CELLB + ((l4 ^ (p & 7)) << 4)) & 0xffffu;
#pragma unroll
                for (int ks = 0; ks < KPT; ++ks) col[d][ks] = o ^ (ks << 6);
            }
        }
        auto fetch = [&](int it, int tap, bf16x8 (&xf)[KPT]) {
            const int dy = tap / 3, d = tap % 3;
            const int disp = (dy * PCOLS + (it / 3) * PCELLS + (it % 3) * 16) * CELLB;
#pragma unroll
            for (int ks = 0; ks < KPT; ++ks)
                xf[ks] = *reinterpret_cast<const bf16x8 *>(&smem[col[d][ks] + disp]);   // img = smem + 0
        };
        // Epilogue of one token tile on its accumulators, cut into 36 single-instruction steps so
        // that it can be issued INSIDE the next tile's MFMA stream.  This lane holds channels
        // 8*(4*mh+l4) .. +7 of token (lane & 15): the bias is already in; SiLU, residual, and the
        // vector goes to the output tile in LDS (dummy tokens: to a scratch slot).
        struct Epi { V8 *slot; V8 rr, o; f32x2 t, v; };
        auto epi_begin = [&](int it, Epi &e) {
            const int smp = 2 * th + it / 3, t = (it % 3) * 16 + l15;
            const int cell = (t >> 3) * COLS + (t & 7);
            // branch-free on purpose: a branch would end the scheduling region
            const uint32_t m = (t & 7) == 7 ? 0xffffffffu : 0u;
            const uint32_t off_real = static_cast<uint32_t>(outt - smem) + smp * OSB + cell * 128 + (((mh * 4 + l4) ^ (cell & 7)) << 4);
            const uint32_t off_dump = static_cast<uint32_t>(dump - smem) + (wave * 8 + (l15 >> 3) + 2 * l4) * 16;
            e.slot = reinterpret_cast<V8 *>(&smem[(off_real & ~m) | (off_dump & m)]);
            if (RESID) e.rr = *e.slot;
        };
        auto epi_step = [&](int step, const f32x4 (&acc)[2], Epi &e) {       // step 0..35, constant after unrolling
            const int q = step / 9;
            const f32x2 x = {acc[q >> 1][2 * (q & 1)], acc[q >> 1][2 * (q & 1) + 1]};
            switch (step % 9) {
            // SiLU = x / (1 + 2^(-x log2 e)) on the hardware exp2 / reciprocal (about 1 ulp each; the
            // result is rounded to bf16 right after)
            case 0: e.t = x * f32x2{-1.44269504f, -1.44269504f}; break;
            case 1: e.t.x = __builtin_amdgcn_exp2f(e.t.x); break;
            case 2: e.t.y = __builtin_amdgcn_exp2f(e.t.y); break;
            case 3: e.t += f32x2{1.0f, 1.0f}; break;
            case 4: e.t.x = __builtin_amdgcn_rcpf(e.t.x); break;
            case 5: e.t.y = __builtin_amdgcn_rcpf(e.t.y); break;
            case 6: e.v = x * e.t; break;
            case 7: if (RESID) e.v += unpack2(e.rr.w[q]); break;
            default: e.o.w[q] = pack2(e.v.x, e.v.y); break;
            }
        };
        auto epi_end = [&](Epi &e) { *e.slot = e.o; };

        // One pipelined block: the 36 (18) MFMAs of token tile `it` (B fragments read one tap
        // ahead), each followed by one or two epilogue steps of the previous tile.  The
        // scheduling barriers pin that order: a wave issues in order, so a VALU instruction
        // hides in an MFMA's free issue cycles only if it sits right behind it in the stream.
        auto block = [&](int it, f32x4 (&acc)[2], bool with_epi, const f32x4 (&pacc)[2]) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt] = f32x4{bia[mt][0], bia[mt][1], bia[mt][2], bia[mt][3]};
            Epi e;
            if (with_epi) epi_begin(it - 1, e);
            bf16x8 xa[KPT], xb[KPT];
            fetch(it, 0, xa);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                bf16x8 (&cur)[KPT] = (tap & 1) ? xb : xa;
                bf16x8 (&nxt)[KPT] = (tap & 1) ? xa : xb;
                if (tap + 1 < 9) fetch(it, tap + 1, nxt);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KPT; ++ks)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int s = tap * KPT + ks;
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[mt][s], cur[ks], acc[mt], 0, 0, 0);
                        if (with_epi) {
                            __builtin_amdgcn_sched_barrier(0);
                            constexpr int PER_SLOT = 36 / (18 * KPT);          // 1 (C_in 64) or 2 (C_in 32)
                            const int slot = (tap * KPT + ks) * 2 + mt;
#pragma unroll
                            for (int j = 0; j < PER_SLOT; ++j) epi_step(slot * PER_SLOT + j, pacc, e);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
            if (with_epi) epi_end(e);
        };
        if (!(dbg & 1)) {
            // az_nn_debug bit 5: the matrix phase at a higher issue priority than the other workgroup's load / store phases
            if ((dbg >> 5) & 3) {
                if (((dbg >> 5) & 3) == 1) __builtin_amdgcn_s_setprio(1);
                else if (((dbg >> 5) & 3) == 2) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(3);
            }
            f32x4 acc_a[2], acc_b[2];
            block(0, acc_a, false, acc_b);
            // unrolled: the tile number is a constant in every read's offset field; the accumulator sets alternate
            block(1, acc_b, true, acc_a);
            block(2, acc_a, true, acc_b);
            block(3, acc_b, true, acc_a);
            block(4, acc_a, true, acc_b);
            block(5, acc_b, true, acc_a);
            {   // drain: the last tile's epilogue on its own
                Epi e;
                epi_begin(5, e);
#pragma unroll
                for (int step = 0; step < 36; ++step) epi_step(step, acc_b, e);
                epi_end(e);
            }
            if ((dbg >> 5) & 3) __builtin_amdgcn_s_setprio(0);
        }
        // the staged tile has landed; every wave is done with img and has written its outputs
        stamp(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(3);
        lds_barrier();
        stamp(4);

        // ---- P3: the output tile leaves as whole 128-byte rows (wave = sample, 16 bytes per lane)
        if (b0 + wave < B && !(dbg & 2)) {
            uint16_t *ys = y + (b0 + wave) * (CELLS * COUT);
#pragma unroll
            for (int i = 0; i < OPER; ++i) {
                const int s = lane_t + 64 * i;
                if (s < OVPS) {
                    const int cell = s >> 3;
                    const V8 v = *reinterpret_cast<const V8 *>(outt + wave * OSB + s * 16);
                    *reinterpret_cast<V8 *>(ys + cell * COUT + (((s & 7) ^ (cell & 7)) << 3)) = v;
                }
            }
        }
        // (the next tile's P1 writes img and reads the other raw buffer; its barrier orders these
        // reads of the output tile before the staging that overwrites it)
        stamp(5);
    }
    if (prof && lane == 0 && blockIdx.x < 512)
        for (int k = 0; k < 6; ++k) g_prof[(blockIdx.x * 4 + wave) * 8 + k] = tp[k];
}

int g_dbg = 0;   // timing experiments only (az_nn_debug): 1 skips the MFMA loop, 2 skips the store

template <int CIN, bool NORM, bool RESID, bool EMBED = false>
int launch(const void *x, const void *w, const void *bias, const void *gamma, const void *beta, void *y, int64_t B,
           float eps, const int64_t *batch_dev, hipStream_t s, EmbedIn em = EmbedIn{})
{
    constexpr size_t smem = static_cast<size_t>(TS) * PCELLS * CELLB + 2 * static_cast<size_t>(TS) * CELLS * CIN * 2 +
                            (RESID ? 0 : static_cast<size_t>(TS) * CELLS * COUT * 2) + 2 * CIN * sizeof(float) + 512;
    static bool attr_set = false;
    auto kern = k_conv_block<CIN, NORM, RESID, EMBED>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(smem)) != hipSuccess)
            return 2;
        attr_set = true;
        if (getenv("AZ_NN_VERBOSE") != nullptr) {
            int per_cu = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), 256, smem);
            fprintf(stderr, "[az_nn] conv block C_in=%d: %zu B LDS, %d workgroups per CU\n", CIN, smem, per_cu);
        }
    }
    // issue priority of the matrix phase over the other workgroup's load / store phases (s_setprio 0..3; AZ_NN_CONV_PRIO)
    static const int prio = [] { const char *e = getenv("AZ_NN_CONV_PRIO"); const int v = e ? atoi(e) : 1; return v < 0 || v > 3 ? 1 : v; }();
    const int64_t ntiles = (B + TS - 1) / TS;
    static const int64_t max_grid = getenv("AZ_NN_CONV_GRID") ? atoll(getenv("AZ_NN_CONV_GRID")) : 512;   // two workgroups per CU
    const unsigned grid = static_cast<unsigned>(ntiles < max_grid ? ntiles : max_grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, static_cast<const uint16_t *>(x),
                       static_cast<const uint16_t *>(w), static_cast<const uint16_t *>(bias),
                       static_cast<const uint16_t *>(gamma), static_cast<const uint16_t *>(beta),
                       static_cast<uint16_t *>(y), B, eps, g_dbg | (((g_dbg >> 5) & 3) ? 0 : (prio << 5)), batch_dev, em);
    return 0;
}

}  // namespace

extern "C" {

int az_nn_debug(int flags) { g_dbg = flags; return 0; }

int az_nn_conv_profile(unsigned long long *out, int n)
{
    if (n > 2048 * 8) n = 2048 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 2;
}

int az_nn_conv_block(const void *x, int c_in, const void *weight_ohwi, const void *bias, const void *gamma,
                     const void *beta, int residual, void *y, int64_t batch, float eps, const int64_t *batch_dev,
                     void *stream)
{
    if (batch <= 0) return 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool norm = gamma != nullptr && beta != nullptr;
    if (c_in == 64 && norm && residual) return launch<64, true, true>(x, weight_ohwi, bias, gamma, beta, y, batch, eps, batch_dev, s);
    if (c_in == 64 && norm && !residual) return launch<64, true, false>(x, weight_ohwi, bias, gamma, beta, y, batch, eps, batch_dev, s);
    if (c_in == 32 && !norm && !residual) return launch<32, false, false>(x, weight_ohwi, bias, gamma, beta, y, batch, eps, batch_dev, s);
    return 1;
}

int az_nn_stem_embed(const float *features, const void *emb_own, const void *emb_opp, const void *pos,
                     const void *weight_ohwi, const void *bias, void *y, int64_t batch, const int32_t *gather,
                     const int64_t *batch_dev, void *stream)
{
    if (batch <= 0 || features == nullptr) return 1;
    EmbedIn em{features, static_cast<const uint16_t *>(emb_own), static_cast<const uint16_t *>(emb_opp),
               static_cast<const uint16_t *>(pos), gather, nullptr, nullptr, nullptr, nullptr};
    return launch<32, false, false, true>(nullptr, weight_ohwi, bias, nullptr, nullptr, y, batch, 0.0f, batch_dev,
                                          static_cast<hipStream_t>(stream), em);
}

int az_nn_stem_embed_positions(const az_nn_positions *positions, const void *emb_own, const void *emb_opp, const void *pos,
                               const void *weight_ohwi, const void *bias, void *y, int64_t batch, const int32_t *gather,
                               const int64_t *batch_dev, void *stream)
{
    if (batch <= 0 || positions == nullptr || !positions->bb_p1 || !positions->bb_p2 || !positions->turn || !positions->sym) return 1;
    EmbedIn em{nullptr, static_cast<const uint16_t *>(emb_own), static_cast<const uint16_t *>(emb_opp),
               static_cast<const uint16_t *>(pos), gather, positions->bb_p1, positions->bb_p2, positions->turn, positions->sym};
    return launch<32, false, false, true>(nullptr, weight_ohwi, bias, nullptr, nullptr, y, batch, 0.0f, batch_dev,
                                          static_cast<hipStream_t>(stream), em);
}

}  // extern "C"
