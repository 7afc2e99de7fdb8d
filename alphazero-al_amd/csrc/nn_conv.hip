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

#include "az_nn.h"

namespace {

constexpr int CELLS = 42, COLS = 7;
constexpr int PCOLS = 9, PCELLS = 72;          // zero-padded 8 x 9 image
constexpr int TS = 4;                          // samples per tile = wavefronts per workgroup
constexpr int TROWS = TS * CELLS;              // 168 tokens
constexpr int MT = (TROWS + 15) / 16;          // 11 token tiles, the last one half full
constexpr int COUT = 64;
constexpr int CELLB = 128;                     // bytes per image cell (C_in 32 uses half of it)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct alignas(16) V8 { uint32_t w[4]; };

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint16_t to_bf16(float a)
{
    const __hip_bfloat16 x = __float2bfloat16(a);
    return *reinterpret_cast<const uint16_t *>(&x);
}
__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    return static_cast<uint32_t>(to_bf16(a)) | (static_cast<uint32_t>(to_bf16(b)) << 16);
}
__device__ __forceinline__ void unpack8(const V8 &v, float f[8])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf_lo(v.w[i]); f[2 * i + 1] = bf_hi(v.w[i]); }
}
__device__ __forceinline__ V8 pack8(const float f[8])
{
    V8 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v.w[i] = pack2(f[2 * i], f[2 * i + 1]);
    return v;
}
// x * sigmoid(x) with the hardware exp2 / reciprocal (about 1 ulp each; the result is rounded to
// bf16 right after).  The IEEE divide costs ~10 instructions per element, and this kernel's
// elementwise work is issued by only four wavefronts per CU.
__device__ __forceinline__ float silu(float x)
{
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x));
}

// 16 bytes per lane from global memory straight into LDS at (wave-uniform) lds + lane * 16
__device__ __forceinline__ void glds16(const void *gsrc, uint8_t *lds)
{
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

template <int CIN, bool NORM, bool RESID>
__global__ void __launch_bounds__(256, 2) k_conv_block(const uint16_t *x, const uint16_t *w, const uint16_t *bias,
                                                       const uint16_t *gamma, const uint16_t *beta, uint16_t *y,
                                                       int64_t B, float eps, int dbg)
{
    constexpr int K = 9 * CIN;
    constexpr int KSTEPS = K / 32;                // 18 (C_in 64) or 9 (C_in 32)
    constexpr int KPT = CIN / 32;                 // k steps per tap
    constexpr int VPC = CIN / 8;                  // 16-byte vectors per cell
    constexpr int VPS = CELLS * VPC;              // vectors per sample
    constexpr int PER = (VPS + 63) / 64;          // vectors per lane of a sample's wavefront
    constexpr int SB = VPS * 16;                  // bytes per raw sample

    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t *img = smem;                                          // TS * PCELLS * CELLB, swizzled
    uint8_t *rawb = smem + TS * PCELLS * CELLB;                   // 2 x TS x SB: raw tiles, double buffered
    __shared__ float s_gam[NORM ? CIN : 1], s_bet[NORM ? CIN : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int mh = wave & 1, th = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- weights: this wave's A fragments (rows = its 32 output channels), resident for the kernel
    bf16x8 aw[2][KSTEPS];
    float bia[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int o = mh * 32 + mt * 16 + l15;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            aw[mt][s] = *reinterpret_cast<const bf16x8 *>(w + static_cast<size_t>(o) * K + s * 32 + l4 * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            bia[mt][r] = __uint_as_float(static_cast<uint32_t>(bias[mh * 32 + mt * 16 + l4 * 4 + r]) << 16);
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
    // order is XORed with (cell & 7) through the SOURCE address, which spreads the residual reads
    // of the epilogue over the banks and leaves every lane with one fixed channel chunk in P1.
    auto swz = [](int cell) { return VPC == 8 ? (cell & 7) : 0; };
    auto stage_tile = [&](int64_t tile, int buf, int lane) {
        const int64_t b = tile * TS + wave;
        if (b >= B) return;
        const uint16_t *xs = x + b * (CELLS * CIN);
        uint8_t *dst = rawb + (buf * TS + wave) * SB;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int s = lane + 64 * i;
            if (s < VPS) {
                const int cell = s / VPC;
glds16(xs + (cell * VPC + ((s % VPC) ^ swz(cell))) * 8, dst + i * 1024);
            }
        }
    };

    const int64_t ntiles = (B + TS - 1) / TS;
    if (static_cast<int64_t>(blockIdx.x) < ntiles) stage_tile(blockIdx.x, 0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int par = 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
        const int64_t b0 = tile * TS;
        const uint8_t *rawt = rawb + par * TS * SB;
        // Opaque copy of the lane id: everything P1 and the staging derive from it is recomputed
        // per tile (a few VALU ops) instead of being hoisted out of the tile loop, where ~40
        // loop-invariant addresses would push the resident weight fragments into scratch.
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));

        // ---- P1: GroupNorm statistics of this wave's sample (one pass, fp32), then the
        // normalised vectors go to the padded image
        {
            const int ck = (lane_t % VPC) ^ swz(lane_t / VPC);      // the channel chunk of every slot this lane owns
            V8 raw[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int s = lane_t + 64 * i;
                if (s < VPS) raw[i] = *reinterpret_cast<const V8 *>(rawt + wave * SB + s * 16);
                else raw[i].w[0] = raw[i].w[1] = raw[i].w[2] = raw[i].w[3] = 0;
            }
            float sc[8], sh[8];
            if (NORM) {
                float sum = 0.0f, sq = 0.0f;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    float f[8];
                    unpack8(raw[i], f);
#pragma unroll
                    for (int q = 0; q < 8; ++q) { sum += f[q]; sq += f[q] * f[q]; }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); sq += __shfl_xor(sq, o, 64); }
                const float mean = sum * (1.0f / (CELLS * CIN));
                const float var = fmaxf(sq * (1.0f / (CELLS * CIN)) - mean * mean, 0.0f);
                const float rstd = rsqrtf(var + eps);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    sc[q] = rstd * s_gam[ck * 8 + q];
                    sh[q] = s_bet[ck * 8 + q] - mean * sc[q];
                }
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int s = lane_t + 64 * i;
                if (s < VPS) {
                    const int cell = s / VPC;
                    const int r = cell / COLS, c = cell - r * COLS;
                    const int p = (r + 1) * PCOLS + (c + 1);
                    V8 out = raw[i];
                    if (NORM) {
                        float f[8];
                        unpack8(raw[i], f);
#pragma unroll
                        for (int q = 0; q < 8; ++q) f[q] = f[q] * sc[q] + sh[q];
                        out = pack8(f);
                    }
                    *reinterpret_cast<V8 *>(img + (wave * PCELLS + p) * CELLB + ((ck ^ (p & 7)) << 4)) = out;
                }
            }
        }
        __syncthreads();
        // the next tile travels from HBM into the other raw buffer while this one is multiplied
        if (tile + gridDim.x < ntiles) stage_tile(tile + gridDim.x, par ^ 1, lane_t);

        // ---- P2: wave (mh, th) multiplies token tiles [6*th, 6*th+6) two at a time
#pragma unroll 1
        for (int pr = 0; pr < ((dbg & 1) ? 0 : 3); ++pr) {
            const int t0 = th * 6 + pr * 2;
            int row[2], pc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = t0 + u < MT ? t0 + u : t0;           // the 12th tile does not exist
                row[u] = t * 16 + l15;
                const int rc = row[u] < TROWS ? row[u] : TROWS - 1;
                const int smp = rc / CELLS, cell = rc - smp * CELLS;
                const int r = cell / COLS, c = cell - r * COLS;
                pc[u] = smp * PCELLS + (r + 1) * PCOLS + (c + 1);
                if (t0 + u >= MT) row[u] = TROWS;                   // nothing to store
            }
            f32x4 acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
            // B fragments of one tap: cell p = centre + tap offset (PCELLS is a multiple of 8, so
            // p & 7 is the swizzle key); chunk 4*ks + l4, i.e. the second half of the cell is the
            // first with address bit 6 flipped
            auto fetch = [&](int tap, bf16x8 (&xf)[2][KPT]) {
                const int off = (tap / 3 - 1) * PCOLS + (tap % 3 - 1);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int p = pc[u] + off;
                    const uint32_t a0 = static_cast<uint32_t>(p * CELLB + ((l4 ^ (p & 7)) << 4));
#pragma unroll
                    for (int ks = 0; ks < KPT; ++ks) xf[u][ks] = *reinterpret_cast<const bf16x8 *>(img + (a0 ^ (ks << 6)));
                }
            };
            // one tap ahead: the reads of tap+1 are issued before the MFMAs of tap, and the
            // scheduling barrier keeps the compiler from hoisting all 36 reads (144 VGPRs) to the top
            bf16x8 xa[2][KPT], xb[2][KPT];
            fetch(0, xa);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                bf16x8 (&cur)[2][KPT] = (tap & 1) ? xb : xa;
                bf16x8 (&nxt)[2][KPT] = (tap & 1) ? xa : xb;
                if (tap + 1 < 9) fetch(tap + 1, nxt);
#pragma unroll
                for (int ks = 0; ks < KPT; ++ks) {
                    const int s = tap * KPT + ks;
                    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[0][s], cur[0][ks], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[1][s], cur[0][ks], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[0][s], cur[1][ks], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[1][s], cur[1][ks], acc[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // C layout: column = token (lane & 15), rows = output channels 4*(lane>>4)+reg of the m tile
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (row[u] >= TROWS || (dbg & 2)) continue;
                const int smp = row[u] / CELLS, cell = row[u] - smp * CELLS;
                if (b0 + smp >= B) continue;
                uint16_t *yt = y + (b0 * CELLS + row[u]) * COUT;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    uint32_t lo = pack2(silu(acc[u][mt][0] + bia[mt][0]), silu(acc[u][mt][1] + bia[mt][1]));
                    uint32_t hi = pack2(silu(acc[u][mt][2] + bia[mt][2]), silu(acc[u][mt][3] + bia[mt][3]));
                    if (RESID) {
                        const int chunk = mh * 4 + mt * 2 + (l4 >> 1);
                        const uint2 rr = *reinterpret_cast<const uint2 *>(
                            rawt + smp * SB + cell * 128 + ((chunk ^ swz(cell)) << 4) + (l4 & 1) * 8);
                        lo = pack2(bf_lo(lo) + bf_lo(rr.x), bf_hi(lo) + bf_hi(rr.x));
                        hi = pack2(bf_lo(hi) + bf_lo(rr.y), bf_hi(hi) + bf_hi(rr.y));
                    }
                    uint2 o; o.x = lo; o.y = hi;
                    *reinterpret_cast<uint2 *>(yt + mh * 32 + mt * 16 + l4 * 4) = o;
                }
            }
        }
        // the staged tile has landed, and every wave is done with img / this raw buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

int g_dbg = 0;   // timing experiments only (az_nn_debug): 1 skips the MFMA loop, 2 skips the store

template <int CIN, bool NORM, bool RESID>
int launch(const void *x, const void *w, const void *bias, const void *gamma, const void *beta, void *y, int64_t B,
           float eps, hipStream_t s)
{
    constexpr size_t smem = static_cast<size_t>(TS) * PCELLS * CELLB + 2 * static_cast<size_t>(TS) * CELLS * CIN * 2;
    static bool attr_set = false;
    auto kern = k_conv_block<CIN, NORM, RESID>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(smem)) != hipSuccess)
            return 2;
        attr_set = true;
    }
    const int64_t ntiles = (B + TS - 1) / TS;
    const unsigned grid = static_cast<unsigned>(ntiles < 512 ? ntiles : 512);     // two workgroups per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, static_cast<const uint16_t *>(x),
                       static_cast<const uint16_t *>(w), static_cast<const uint16_t *>(bias),
                       static_cast<const uint16_t *>(gamma), static_cast<const uint16_t *>(beta),
                       static_cast<uint16_t *>(y), B, eps, g_dbg);
    return 0;
}

}  // namespace

extern "C" {

int az_nn_debug(int flags) { g_dbg = flags; return 0; }

int az_nn_conv_block(const void *x, int c_in, const void *weight_ohwi, const void *bias, const void *gamma,
                     const void *beta, int residual, void *y, int64_t batch, float eps, void *stream)
{
    if (batch <= 0) return 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool norm = gamma != nullptr && beta != nullptr;
    if (c_in == 64 && norm && residual) return launch<64, true, true>(x, weight_ohwi, bias, gamma, beta, y, batch, eps, s);
    if (c_in == 64 && norm && !residual) return launch<64, true, false>(x, weight_ohwi, bias, gamma, beta, y, batch, eps, s);
    if (c_in == 32 && !norm && !residual) return launch<32, false, false>(x, weight_ohwi, bias, gamma, beta, y, batch, eps, s);
    return 1;
}

}  // extern "C"
