// nn_othello.hip - the 3x3 convolutions of the reference's Othello network (256 channels on
// 10x10 / 8x8 maps; Othello/Network.py:22-66, 129-139) as an implicit-GEMM MFMA kernel.
//
// ~97 % of that network's ~1 GFLOP per leaf is eight such convolutions.  One layer here is
//
//   y = [silu]( post_s * conv3x3( pad( [pre_s * x + pre_b] ) ) + post_b  [+ residual] )
//
// on NHWC bf16 activations: the BatchNorm in FRONT of a convolution (residual blocks) is an
// affine applied while the sample is copied into LDS - the zero padding stays zero, which is why
// it cannot be folded into the weights - and the BatchNorm BEHIND one is an affine on the fp32
// accumulators.  Roundings follow the reference under bf16 autocast: the normalised input, the
// convolution (+ affine) output and the sum with the residual are each rounded to bf16.
//
// Mapping (v_mfma_f32_16x16x32_bf16, out^T = W . X^T as in nn_conv.hip):
//   workgroup = one sample, 4 wavefronts; wavefront w owns output channels 64w .. 64w+63
//               (4 channel tiles) for ALL token tiles of the sample (7 at 10x10, 4 at 8x8):
//               112 accumulator VGPRs, 4 MFMAs per LDS read;
//   B operand  = the sample's zero-padded image in LDS, 16-byte chunks XOR-swizzled with the
//               cell index so that the 16 tokens of a read hit distinct bank slots;
//   A operand  = weights, 1.18 MB per layer: too large for registers or LDS, they stream from
//               L2 in FRAGMENT ORDER (packed once on the host: [k step][channel tile][lane][8]),
//               so a wavefront's four fragments of a k step are one contiguous 4 KB, fetched
//               one k step ahead of their use;
//   epilogue   = accumulators -> affine -> bf16 -> LDS (token major, swizzled) -> a coalesced
//               pass that adds the residual, applies SiLU and writes whole 16-byte vectors.
// Two workgroups share a CU (73.7 KB of LDS each), so one's copy / epilogue phases overlap the
// other's MFMA phase.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "az_nn.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int COUT = 256;
constexpr int ROWB = COUT * 2;            // bytes of one output token

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float round_bf(float v) { return bf_lo(pack2(v, 0.0f) & 0xffffu); }
// v * sigmoid(v) as v_mul, v_exp, v_add, v_rcp, v_mul (the IEEE division __fdividef compiles to without fast-math is
// ten instructions per element; the result is rounded to bf16 right after)
__device__ __forceinline__ float silu(float v)
{
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}

template <int CIN, int HI, int PAD, bool PRE, bool RES, bool SILU>
__global__ void __launch_bounds__(256, 2) k_oth_conv(const uint16_t *x, const uint16_t *wp, const float *pre_s,
                                                     const float *pre_b, const float *post_s, const float *post_b,
                                                     const uint16_t *res, uint16_t *y, int64_t B, int dbg,
                                                     const int64_t *batch_dev)
{
    if (batch_dev != nullptr && *batch_dev < B) B = *batch_dev;      // compact batch whose size only the device knows
    constexpr int PW = HI + 2 * PAD;          // padded width
    constexpr int HO = PW - 2;                // output width
    constexpr int NT = HO * HO;               // output tokens
    constexpr int T = (NT + 15) / 16;         // token tiles
    constexpr int CELLB = CIN * 2;            // bytes per image cell
    constexpr int CPC = CIN / 8;              // 16-byte chunks per cell
    constexpr int KPT = CIN / 32;             // k steps per tap
    // The image is LINEAR in LDS - no XOR swizzle - so that a fragment's address is one per-lane base plus a
    // constant: byte(row, col, chunk) = row * ROWP + col * CELLP + chunk * 16.  CELLP = one chunk more than a
    // cell: consecutive cells sit an odd number of 16-byte slots apart, so the 16 lanes of a read (16 consecutive
    // tokens, the same chunk of 16 cells) cover all 64 banks.  ROWP = PW cells plus what makes a step from the
    // last token of an output row to the first of the next one (PW - HO = 2 cells further) look like one more
    // cell to the banks: (2 * CELLP + pad) % 256 == 0.  Tap (ky, kx) and k step kc are then the CONSTANT
    // ky * ROWP + kx * CELLP + kc * 64 in the read's offset field.
    constexpr int CELLP = CELLB + 16;
    constexpr int ROWP = PW * CELLP + (256 - (2 * CELLP) % 256) % 256;
    // Inside a 512-byte cell the chunks are permuted: a ds_read_b128 is serviced in four groups of 16 lanes that
    // mix two k groups (lanes 0-3, 12-15 of k group g with lanes 4-11 of k group g + 1: MI355X_MICROARCH.md,
    // LDS), so the chunks of k groups g and g + 1 of one k step must sit a multiple of 256 bytes apart for the
    // 16 tokens of such a group to land on 16 distinct slots: chunk c = 4 * kc + g lives at
    // 16 * kc + 256 * (g & 1) + 128 * (g >> 1).  (Cells of 64 bytes - the stem - stay in channel order.)
    constexpr bool PERM = CPC == 32;
    constexpr int KSTEP = PERM ? 16 : 64;     // bytes from one k step's chunk to the next one's
    constexpr int KS = 9 * KPT;
    static_assert(!RES || PAD == 1, "the residual has the output's geometry");
    static_assert(256 % CPC == 0, "a thread keeps its channel chunk over the copy loop");

    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, tl = lane & 15;

    // byte address of this lane's chunk (k group g) in the top-left cell of every token's window
    int cell0[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        int token = t * 16 + tl;
        token = token < NT ? token : NT - 1;         // the clamped tail of the last tile repeats its last token
        cell0[t] = (token / HO) * ROWP + (token % HO) * CELLP + (PERM ? ((g & 1) << 8) | ((g & 2) << 6) : g * 16);
    }
    const int wlane_off = (wave * 4) * 512 + lane * 8;           // this lane's place in a k step's 16 weight fragments

    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        // ---- copy: sample -> zero-padded image (the BatchNorm in front of the convolution rides here)
        const uint16_t *xs = x + b * (HI * HI * CIN);
        // the pre-affine of the channel chunk this thread copies (loaded per sample: 16 registers that
        // must not stay live through the MFMA phase)
        float ps[8], pb[8];
        if (PRE) {
            const int c = tid % CPC;
#pragma unroll
            for (int j = 0; j < 8; ++j) { ps[j] = pre_s[c * 8 + j]; pb[j] = pre_b[c * 8 + j]; }
        }
        // loads in batches of six, then their arithmetic and LDS stores: as a rolled loop this phase is
        // one global-load latency per iteration (18 of them per sample); all 18 at once cost 72 registers
        constexpr int NV = PW * PW * CPC, ITERS = (NV + 255) / 256, BATCH = 6;
        int ctid = tid;
        asm volatile("" : "+v"(ctid));      // per-sample recomputation of the copy addresses: hoisted out of the sample loop they pin ~60 registers
#pragma unroll
        for (int base = 0; base < ITERS; base += BATCH) {
            uint4 vals[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int v = ctid + (base + k) * 256;
                const int cell = v / CPC, c = v % CPC;
                const int iy = cell / PW - PAD, ix = cell % PW - PAD;
                vals[k] = make_uint4(0u, 0u, 0u, 0u);
                if (base + k < ITERS && v < NV && iy >= 0 && iy < HI && ix >= 0 && ix < HI)
                    vals[k] = *reinterpret_cast<const uint4 *>(xs + (iy * HI + ix) * CIN + c * 8);
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int v = ctid + (base + k) * 256;
                const int cell = v / CPC, c = v % CPC;
                const int iy = cell / PW - PAD, ix = cell % PW - PAD;
                uint4 val = vals[k];
                if (PRE && iy >= 0 && iy < HI && ix >= 0 && ix < HI) {
                    uint32_t *wv = reinterpret_cast<uint32_t *>(&val);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        wv[j] = pack2(bf_lo(wv[j]) * ps[2 * j] + pb[2 * j], bf_hi(wv[j]) * ps[2 * j + 1] + pb[2 * j + 1]);
                }
                if (base + k < ITERS && v < NV)
                    *reinterpret_cast<uint4 *>(smem + (cell / PW) * ROWP + (cell % PW) * CELLP +
                                               (PERM ? ((c >> 2) << 4) | ((c & 1) << 8) | ((c & 2) << 6) : (c << 4))) = val;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();

        // ---- implicit GEMM: 4 channel tiles x T token tiles per wavefront
        f32x4 acc[4][T];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto fetch_a = [&](bf16x8 (&a)[4], int ks) {
            if (dbg & 1) ks &= 1;       // timing experiment (AZ_OTH_DEBUG=1, results wrong): the weight stream stays in L1
            const uint16_t *wk = wp + static_cast<size_t>(ks) * (16 * 512);      // wave-uniform base, lane offset, constant
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(wk + wlane_off + i * 512);
        };
        auto multiply = [&](const bf16x8 (&a)[4], const bf16x8 (&bq)[T]) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bq[t], acc[i][t], 0, 0, 0);
        };
        // the matrix phase at a higher issue priority than the partner workgroup's copy / output phases (AZ_OTH_DEBUG bit 3: off)
        if (!(dbg & 8)) __builtin_amdgcn_s_setprio(1);
        if (dbg & 2) {
            // timing experiment (AZ_OTH_DEBUG=2, results wrong): no MFMA phase - what the copy and output phases cost alone
        } else if constexpr (KPT >= 2) {
            // Software pipeline over k steps.  Weight fragments (L2, ~1 us away) of step k+1 are
            // requested before the MFMAs of step k, into the other half of a register double
            // buffer; a token fragment (LDS) of step k+1 is requested as soon as the four MFMAs
            // that use its register in step k are issued - one whole step ahead of its use, with no
            // second buffer.  The scheduling barriers keep the compiler from sinking the loads to
            // their first use (left alone it does, and every k step waits out the L2 latency).
            // KPT is even, so the halves of the A buffer are back in place at the loop edge.
            bf16x8 a[2][4], bq[T];
            int cb[T];                                   // cell0 moved to the current tap
#pragma unroll
            for (int t = 0; t < T; ++t) cb[t] = cell0[t];
            fetch_a(a[0], 0);
#pragma unroll
            for (int t = 0; t < T; ++t) bq[t] = *reinterpret_cast<const bf16x8 *>(smem + cb[t]);
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int ntap = tap < 8 ? tap + 1 : 8;                                     // the last step re-reads itself
                const int to_next = (ntap / 3 - tap / 3) * ROWP + (ntap % 3 - tap % 3) * CELLP;   // wave-uniform
#pragma unroll
                for (int kc = 0; kc < KPT; ++kc) {
                    const int ks = tap * KPT + kc;
                    fetch_a(a[(kc + 1) & 1], ks + 1 < KS ? ks + 1 : ks);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kc & 1][i], bq[t], acc[i][t], 0, 0, 0);
                        if (kc + 1 < KPT) {
                            bq[t] = *reinterpret_cast<const bf16x8 *>(smem + cb[t] + (kc + 1) * KSTEP);    // offset field
                        } else {
                            cb[t] += to_next;                                                              // one add per tap and tile
                            bq[t] = *reinterpret_cast<const bf16x8 *>(smem + cb[t]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        } else {
            bf16x8 a[4], bq[T];
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                fetch_a(a, tap);
                const int off = (tap / 3) * ROWP + (tap % 3) * CELLP;
#pragma unroll
                for (int t = 0; t < T; ++t) bq[t] = *reinterpret_cast<const bf16x8 *>(smem + cell0[t] + off);
                multiply(a, bq);
            }
        }
        if (!(dbg & 8)) __builtin_amdgcn_s_setprio(0);
        __syncthreads();                       // every wavefront is done with the image

        // ---- accumulators -> affine -> bf16 -> token-major stage in LDS (16-byte chunks swizzled by token)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = (wave * 4 + i) * 16 + g * 4;
            const f32x4 s4 = *reinterpret_cast<const f32x4 *>(post_s + ch);
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(post_b + ch);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int token = t * 16 + tl;
                if (token < NT) {
                    const f32x4 v = acc[i][t] * s4 + b4;
                    uint2 o;
                    o.x = pack2(v[0], v[1]);
                    o.y = pack2(v[2], v[3]);
                    *reinterpret_cast<uint2 *>(smem + token * ROWB + ((((ch >> 3)) ^ (token & 15)) << 4) + ((ch & 7) << 1)) = o;
                }
            }
        }
        __syncthreads();

        // ---- coalesced pass: [+ residual] -> [SiLU] -> global
        uint16_t *ys = y + b * (NT * COUT);
        const uint16_t *rs = RES ? res + b * (NT * COUT) : nullptr;
        constexpr int OV = NT * 32, OITERS = (OV + 255) / 256, OBATCH = 7;
        int otid = tid;
        asm volatile("" : "+v"(otid));
#pragma unroll
        for (int base = 0; base < OITERS; base += OBATCH) {
            uint4 rvals[OBATCH];
            if (RES) {
#pragma unroll
                for (int k = 0; k < OBATCH; ++k) {
                    const int v = otid + (base + k) * 256;
                    rvals[k] = make_uint4(0u, 0u, 0u, 0u);
                    if (base + k < OITERS && v < OV) rvals[k] = *reinterpret_cast<const uint4 *>(rs + (v >> 5) * COUT + (v & 31) * 8);
                }
            }
#pragma unroll
            for (int k = 0; k < OBATCH; ++k) {
                const int v = otid + (base + k) * 256;
                if (base + k < OITERS && v < OV) {
                    const int token = v >> 5, cc = v & 31;
                    uint4 val = *reinterpret_cast<const uint4 *>(smem + token * ROWB + ((cc ^ (token & 15)) << 4));
                    uint32_t *wv = reinterpret_cast<uint32_t *>(&val);
                    const uint32_t *rw = reinterpret_cast<const uint32_t *>(&rvals[k]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float lo = bf_lo(wv[j]), hi = bf_hi(wv[j]);
                        if (RES) { lo = round_bf(lo + bf_lo(rw[j])); hi = round_bf(hi + bf_hi(rw[j])); }
                        if (SILU) { lo = silu(lo); hi = silu(hi); }
                        wv[j] = pack2(lo, hi);
                    }
                    *reinterpret_cast<uint4 *>(ys + token * COUT + cc * 8) = val;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                       // the stage is the next sample's image
    }
}

template <int CIN, int HI, int PAD, bool PRE, bool RES, bool SILU>
int launch(const void *x, const void *wp, const float *pre_s, const float *pre_b, const float *post_s,
           const float *post_b, const void *res, void *y, int64_t B, const int64_t *batch_dev, hipStream_t s)
{
    constexpr int PW = HI + 2 * PAD, HO = PW - 2;
    constexpr int CELLP = CIN * 2 + 16, ROWP = PW * CELLP + (256 - (2 * CELLP) % 256) % 256;   // the kernel's image layout
    constexpr int IMG = PW * ROWP, STAGE = HO * HO * ROWB;
    constexpr int SMEM = IMG > STAGE ? IMG : STAGE;
    auto kern = k_oth_conv<CIN, HI, PAD, PRE, RES, SILU>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
            return 2;
        attr_set = true;
        if (getenv("AZ_NN_VERBOSE") != nullptr) {
            int per_cu = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), 256, SMEM);
            fprintf(stderr, "[az_nn] othello conv C_in=%d H=%d pad=%d: %d B LDS, %d workgroups per CU\n", CIN, HI, PAD, SMEM, per_cu);
        }
    }
    static const int dbg = getenv("AZ_OTH_DEBUG") ? atoi(getenv("AZ_OTH_DEBUG")) : 0;
    static const int64_t max_grid = getenv("AZ_OTH_GRID") ? atoll(getenv("AZ_OTH_GRID")) : 512;   // two workgroups per CU, persistent over samples
    const unsigned grid = static_cast<unsigned>(B < max_grid ? B : max_grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), SMEM, s, static_cast<const uint16_t *>(x),
                       static_cast<const uint16_t *>(wp), pre_s, pre_b, post_s, post_b,
                       static_cast<const uint16_t *>(res), static_cast<uint16_t *>(y), B, dbg, batch_dev);
    return 0;
}

// The dual head's bottleneck (Othello/Network.py:81-83): 3x3, NO padding, 256 -> 8 channels on the
// 10x10 map -> (8, 8, 8), BatchNorm, SiLU.  One channel tile (8 real channels + 8 of zero weight), so
// the four wavefronts of a sample's workgroup split its four TOKEN tiles instead; the weights (72 KB
// in fragment order) are the same lines for every wavefront and stay in L1.  The kernel is bound by
// the copy of the sample into LDS, not by its 72 MFMAs per wavefront.
__global__ void __launch_bounds__(256, 2) k_oth_conv_narrow(const uint16_t *x, const uint16_t *wp, const float *post_s,
                                                            const float *post_b, uint16_t *y, int64_t B,
                                                            const int64_t *batch_dev)
{
    if (batch_dev != nullptr && *batch_dev < B) B = *batch_dev;
    constexpr int CIN = 256, HI = 10, PW = 10, HO = 8, NT = 64, CELLB = CIN * 2, CPC = 32, KPT = 8;
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, tl = lane & 15;
    const int token = wave * 16 + tl;
    const int cell0 = (token / HO) * PW + (token % HO);
    const f32x4 s4 = *reinterpret_cast<const f32x4 *>(post_s + g * 4);
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(post_b + g * 4);
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        const uint16_t *xs = x + b * (HI * HI * CIN);
        constexpr int NV = PW * PW * CPC, ITERS = (NV + 255) / 256;         // 3200 chunks: 12.5 per thread
        int ctid = tid;
        asm volatile("" : "+v"(ctid));
#pragma unroll
        for (int base = 0; base < ITERS; base += 7) {
            uint4 vals[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int v = ctid + (base + k) * 256;
                vals[k] = make_uint4(0u, 0u, 0u, 0u);
                if (base + k < ITERS && v < NV) vals[k] = *reinterpret_cast<const uint4 *>(xs + v * 8);
            }
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int v = ctid + (base + k) * 256;
                const int cell = v / CPC, c = v % CPC;
                if (base + k < ITERS && v < NV)
                    *reinterpret_cast<uint4 *>(smem + cell * CELLB + ((c ^ (((cell / PW) * HO + cell % PW) & 15)) << 4)) = vals[k];
            }
        }
        __syncthreads();
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int p = cell0 + (tap / 3) * PW + (tap % 3);
            const int key = (tl + (tap / 3) * HO + (tap % 3)) & 15;
            bf16x8 a[KPT], bv[KPT];
#pragma unroll
            for (int kc = 0; kc < KPT; ++kc) {
                a[kc] = *reinterpret_cast<const bf16x8 *>(wp + static_cast<size_t>(tap * KPT + kc) * 512 + lane * 8);
                bv[kc] = *reinterpret_cast<const bf16x8 *>(smem + p * CELLB + ((((kc << 2) | g) ^ key) << 4));
            }
#pragma unroll
            for (int kc = 0; kc < KPT; ++kc) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kc], bv[kc], acc, 0, 0, 0);
        }
        if (g < 2) {                                 // channels 0..7 live in the lanes of k groups 0 and 1
            const f32x4 v = acc * s4 + b4;
            uint2 o;
            o.x = pack2(silu(round_bf(v[0])), silu(round_bf(v[1])));
            o.y = pack2(silu(round_bf(v[2])), silu(round_bf(v[3])));
            *reinterpret_cast<uint2 *>(y + (b * NT + token) * 8 + g * 4) = o;
        }
        __syncthreads();                             // the image is rewritten for the next sample
    }
}

}  // namespace

extern "C" int az_nn_othello_conv(const void *x, const void *w_packed, const float *pre_scale, const float *pre_shift,
                                  const float *post_scale, const float *post_shift, const void *residual, void *y,
                                  int64_t batch, int c_in, int h_in, int pad, int apply_silu, const int64_t *batch_dev,
                                  void *stream)
{
    if (batch <= 0 || x == nullptr || w_packed == nullptr || y == nullptr || post_scale == nullptr || post_shift == nullptr)
        return 1;
    if ((pre_scale == nullptr) != (pre_shift == nullptr)) return 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool pre = pre_scale != nullptr, res = residual != nullptr;
#define AZ_OTH(CIN, HI, PAD, PRE, RES, SILU) \
    return launch<CIN, HI, PAD, PRE, RES, SILU>(x, w_packed, pre_scale, pre_shift, post_scale, post_shift, residual, y, batch, batch_dev, s)
    if (c_in == 32 && h_in == 8 && pad == 2 && !pre && !res && apply_silu) AZ_OTH(32, 8, 2, false, false, true);
    if (c_in == 256 && h_in == 10 && pad == 1 && pre && !res && apply_silu) AZ_OTH(256, 10, 1, true, false, true);
    if (c_in == 256 && h_in == 10 && pad == 1 && pre && res && apply_silu) AZ_OTH(256, 10, 1, true, true, true);
    if (c_in == 256 && h_in == 10 && pad == 1 && !pre && !res && apply_silu) AZ_OTH(256, 10, 1, false, false, true);
    if (c_in == 256 && h_in == 10 && pad == 0 && !pre && !res && apply_silu) AZ_OTH(256, 10, 0, false, false, true);
    if (c_in == 256 && h_in == 8 && pad == 1 && !pre && !res && apply_silu) AZ_OTH(256, 8, 1, false, false, true);
#undef AZ_OTH
    return 1;
}

extern "C" int az_nn_othello_conv_narrow(const void *x, const void *w_packed16, const float *post_scale16,
                                         const float *post_shift16, void *y, int64_t batch, const int64_t *batch_dev,
                                         void *stream)
{
    if (batch <= 0 || x == nullptr || w_packed16 == nullptr || y == nullptr || post_scale16 == nullptr || post_shift16 == nullptr)
        return 1;
    constexpr int SMEM = 100 * 512;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_oth_conv_narrow), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
            return 2;
        attr_set = true;
    }
    const unsigned grid = static_cast<unsigned>(batch < 768 ? batch : 768);       // three workgroups per CU
    hipLaunchKernelGGL(k_oth_conv_narrow, dim3(grid), dim3(256), SMEM, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(x), static_cast<const uint16_t *>(w_packed16), post_scale16, post_shift16,
                       static_cast<uint16_t *>(y), batch, batch_dev);
    return 0;
}
