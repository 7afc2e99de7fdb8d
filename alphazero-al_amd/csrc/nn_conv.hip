// nn_conv.hip - one residual convolution block of the evaluator as ONE MFMA kernel.
//
//   y = [x +] silu( conv3x3( [GroupNorm1(x) * gamma + beta] ) + bias )          (Network.py:27-48,166-170)
//
// on token-layout activations (B, 42, C) bf16.  The reference runs this as GroupNorm ->
// convolution -> (bias) -> SiLU -> add: four kernels and five passes over a 176 MB tensor per
// block at 32768 leaves.  Here a workgroup owns a tile of 8 samples (336 tokens = 21 MFMA row
// tiles), keeps the normalised, zero-padded 8x9 images of those samples in LDS, and computes
// the 3x3 convolution as an implicit GEMM with v_mfma_f32_16x16x32_bf16:
//
//   rows  M = 336 tokens of the tile          (A operand: ds_read_b128 from the padded image,
//   cols  N = 64 output channels                one 16-byte read per lane = 8 input channels of
//   depth K = 9 taps x C_in                     the tap's neighbour cell; cell stride C_in+8
//                                               elements keeps the 16 lanes of a read group on
//                                               distinct banks)
//
// The weights never go through LDS: wave (mh, nh) of the 4-wave workgroup owns M half mh and
// the 32 output channels nh*32.., and holds its 2 x (K/32) B fragments in registers (144
// VGPRs for C_in = 64) for the whole kernel; workgroups are persistent over tiles.
// Epilogue: bias + SiLU in fp32 on the accumulators, staged through LDS so that the store is
// 16-byte coalesced rows; the residual comes from the registers that held the tile load.  The
// next tile is fetched from HBM while the current one is multiplied.  HBM traffic per block:
// read x once, write y once.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "az_nn.h"

namespace {

constexpr int CELLS = 42, COLS = 7;
constexpr int PCOLS = 9, PCELLS = 72;          // zero-padded 8 x 9 image
constexpr int TS = 8;                          // samples per tile
constexpr int TROWS = TS * CELLS;              // 336 rows = 21 row tiles of 16
constexpr int MT = TROWS / 16;                 // 21
constexpr int COUT = 64;

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

template <int CIN, bool NORM, bool RESID>
__global__ void __launch_bounds__(256) k_conv_block(const uint16_t *x, const uint16_t *w, const uint16_t *bias,
                                                    const uint16_t *gamma, const uint16_t *beta, uint16_t *y,
                                                    int64_t B, float eps, int dbg)
{
    constexpr int CSTR = CIN + 8;                 // padded cell stride (elements)
    constexpr int K = 9 * CIN;
    constexpr int KSTEPS = K / 32;                // 18 (C_in 64) or 9 (C_in 32)
    constexpr int VPC = CIN / 8;                  // 16-byte vectors per cell
    constexpr int VPS = CELLS * VPC;              // vectors per sample
    constexpr int PER = (VPS + 31) / 32;          // vectors per thread in the tile load

    extern __shared__ __align__(16) uint8_t smem[];
    uint16_t *img = reinterpret_cast<uint16_t *>(smem);                         // TS * PCELLS * CSTR
    uint16_t *stage = img + TS * PCELLS * CSTR;                                  // TROWS * COUT

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int mh = wave >> 1, nh = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- weights: this wave's B fragments, resident for the whole kernel
    bf16x8 bw[2][KSTEPS];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = nh * 32 + nt * 16 + l15;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            bw[nt][s] = *reinterpret_cast<const bf16x8 *>(w + static_cast<size_t>(n) * K + s * 32 + l4 * 8);
    }
    float bia[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bia[nt] = __uint_as_float(static_cast<uint32_t>(bias[nh * 32 + nt * 16 + l15]) << 16);

    // ---- zero the padded images once: the halo is never written again
    {
        V8 z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0;
        constexpr int NV = TS * PCELLS * CSTR / 8;
        for (int i = tid; i < NV; i += 256) reinterpret_cast<V8 *>(img)[i] = z;
    }
    __syncthreads();

    // thread <-> data mapping of the tile load AND of the final store: 32 threads per sample,
    // thread j of a sample owns 16-byte vectors j, j+32, ... of that sample
    const int smp_t = tid >> 5, j_t = tid & 31;

    auto load_tile = [&](int64_t tile, V8 (&raw)[PER]) {
        const int64_t b = tile * TS + smp_t;
        const bool live = b < B;
        const uint16_t *xs = x + (live ? b : 0) * (CELLS * CIN);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int v = j_t + 32 * i;
            if (live && v < VPS) raw[i] = *reinterpret_cast<const V8 *>(xs + v * 8);
            else raw[i].w[0] = raw[i].w[1] = raw[i].w[2] = raw[i].w[3] = 0;
        }
    };

    const int64_t ntiles = (B + TS - 1) / TS;
    V8 raw[PER], nxt[PER];
    if (static_cast<int64_t>(blockIdx.x) < ntiles) load_tile(blockIdx.x, raw);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t b0 = tile * TS;
        const bool live = b0 + smp_t < B;

        // ---- P1: normalise this thread's vectors (already in registers), write the padded images
        {
            float mean = 0.0f, rstd = 1.0f;
            if (NORM) {
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    float f[8];
                    unpack8(raw[i], f);                       // vectors past the sample are zero
#pragma unroll
                    for (int q = 0; q < 8; ++q) sum += f[q];
                }
                for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 32);
                mean = sum * (1.0f / (CELLS * CIN));
                float sq = 0.0f;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    if (j_t + 32 * i < VPS) {
                        float f[8];
                        unpack8(raw[i], f);
#pragma unroll
                        for (int q = 0; q < 8; ++q) { const float d = f[q] - mean; sq += d * d; }
                    }
                }
                for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 32);
                rstd = rsqrtf(sq * (1.0f / (CELLS * CIN)) + eps);
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int v = j_t + 32 * i;
                if (v < VPS) {
                    const int cell = v / VPC, ch = (v - cell * VPC) * 8;
                    const int r = cell / COLS, c = cell - r * COLS;
                    V8 out = raw[i];
                    if (NORM && live) {
                        float f[8], g[8], be[8];
                        unpack8(raw[i], f);
                        unpack8(*reinterpret_cast<const V8 *>(gamma + ch), g);
                        unpack8(*reinterpret_cast<const V8 *>(beta + ch), be);
#pragma unroll
                        for (int q = 0; q < 8; ++q) f[q] = (f[q] - mean) * rstd * g[q] + be[q];
                        out = pack8(f);
                    }
                    *reinterpret_cast<V8 *>(img + (smp_t * PCELLS + (r + 1) * PCOLS + (c + 1)) * CSTR + ch) = out;
                }
            }
        }
        // the next tile's activations travel from HBM while this tile is multiplied
        const int64_t next_tile = tile + gridDim.x;
        if (next_tile < ntiles) load_tile(next_tile, nxt);
        __syncthreads();

        // ---- P2: implicit GEMM on this wave's row tiles x 32 output channels, two row tiles at
        // a time (four independent accumulators keep the matrix pipe busy)
        const int mt0 = mh == 0 ? 0 : (MT + 1) / 2, mt1 = (dbg & 1) ? 0 : (mh == 0 ? (MT + 1) / 2 : MT);
        for (int mt = mt0; mt < mt1; mt += 2) {
            const bool two = mt + 1 < mt1;
            const uint16_t *centre[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int m = (two || u == 0 ? mt + u : mt) * 16 + l15;
                const int smp = m / CELLS, cell = m - smp * CELLS;
                const int r = cell / COLS, c = cell - r * COLS;
                centre[u] = img + (smp * PCELLS + (r + 1) * PCOLS + (c + 1)) * CSTR + l4 * 8;
            }
            f32x4 acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const int k0 = s * 32;
                const int tap = k0 / CIN, chb = k0 - tap * CIN;
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const int off = (dy * PCOLS + dx) * CSTR + chb;
                const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(centre[0] + off);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(centre[1] + off);
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bw[0][s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bw[1][s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bw[0][s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bw[1][s], acc[1][1], 0, 0, 0);
            }
            // C layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = (mt + u) * 16 + l4 * 4 + q;
                    stage[row * COUT + nh * 32 + l15] = to_bf16(silu(acc[u][0][q] + bia[0]));
                    stage[row * COUT + nh * 32 + 16 + l15] = to_bf16(silu(acc[u][1][q] + bia[1]));
                }
            }
        }
        __syncthreads();

        // ---- P3: residual add (from the registers that fed P1) and coalesced store
        if (live && !(dbg & 2)) {
            uint16_t *ys = y + (b0 + smp_t) * (CELLS * COUT);
            constexpr int OVPS = CELLS * COUT / 8;              // output vectors per sample
#pragma unroll
            for (int i = 0; i < (OVPS + 31) / 32; ++i) {
                const int v = j_t + 32 * i;
                if (v < OVPS) {
                    float f[8];
                    unpack8(*reinterpret_cast<const V8 *>(stage + smp_t * (CELLS * COUT) + v * 8), f);
                    if (RESID) {                                // C_in == C_out: same vector index
                        float rr[8];
                        unpack8(raw[i < PER ? i : 0], rr);
#pragma unroll
                        for (int q = 0; q < 8; ++q) f[q] += rr[q];
                    }
                    *reinterpret_cast<V8 *>(ys + v * 8) = pack8(f);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) raw[i] = nxt[i];
        // the next tile's P1 writes only `img`, which every wave finished reading before the
        // barrier above; its barrier in turn orders this P3's reads of `stage` before the next P2
    }
}

int g_dbg = 0;   // timing experiments only (az_nn_debug): 1 skips the MFMA loop, 2 skips the store

template <int CIN, bool NORM, bool RESID>
int launch(const void *x, const void *w, const void *bias, const void *gamma, const void *beta, void *y, int64_t B,
           float eps, hipStream_t s)
{
    constexpr size_t smem = (static_cast<size_t>(TS) * PCELLS * (CIN + 8) + static_cast<size_t>(TROWS) * COUT) * 2;
    static bool attr_set = false;
    auto kern = k_conv_block<CIN, NORM, RESID>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(smem)) != hipSuccess)
            return 2;
        attr_set = true;
    }
    const int64_t ntiles = (B + TS - 1) / TS;
    const unsigned grid = static_cast<unsigned>(ntiles < 256 ? ntiles : 256);
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
