// nn_kernels.hip - memory-bound glue of the leaf evaluator, fused for gfx950.
//
// The network itself stays a PyTorch-ROCm module (MIOpen implicit-GEMM convolutions,
// hipBLASLt projections, SDPA).  What PyTorch does badly at 1.4 M tokens x 64 channels is
// everything BETWEEN those: RMSNorm over 16/64 values launches one tiny block per row
// (1.5 ms per call in the first profile), GroupNorm/affine/SiLU/residual/transposes are 2-4
// kernels each.  These kernels do each such chain in one pass at HBM speed: 16-byte loads and
// stores (8 bf16 per lane), statistics in fp32, 8-lane shuffle reductions, no LDS except the
// per-sample head pooling.  Layout everywhere: tokens (B, 42, C) bf16 == channels-last image.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "az_nn.h"

namespace {

constexpr int CELLS = 42, ROWS = 6, COLS = 7;

struct alignas(16) V8 { uint32_t w[4]; };        // 8 bf16

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ uint32_t pack_bf16(float a, float b)
{
    const __hip_bfloat16 x = __float2bfloat16(a), y = __float2bfloat16(b);   // round to nearest even
    return static_cast<uint32_t>(*reinterpret_cast<const uint16_t *>(&x)) |
           (static_cast<uint32_t>(*reinterpret_cast<const uint16_t *>(&y)) << 16);
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
    for (int i = 0; i < 4; ++i) v.w[i] = pack_bf16(f[2 * i], f[2 * i + 1]);
    return v;
}

__device__ __forceinline__ float bf1(const uint16_t *p) { return __uint_as_float(static_cast<uint32_t>(*p) << 16); }

__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ---------------------------------------------------------------- embedding
// tokens[b, cell, :] = pos[cell, :] + own * e_own + opp * e_opp      (Network.py:226-239)
template <int E>
__global__ void __launch_bounds__(256) k_embed(const float *feat, const uint16_t *e_own, const uint16_t *e_opp,
                                               const uint16_t *pos, uint16_t *tokens, int64_t B,
                                               const int32_t *gather, const int64_t *batch_dev)
{
    constexpr int VPT = E / 8;
    const int64_t rows_total = B;
    if (batch_dev != nullptr && *batch_dev < B) B = *batch_dev;
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t tok = gid / VPT;
    const int vec = static_cast<int>(gid - tok * VPT);
    if (tok >= B * CELLS) return;
    const int64_t b = tok / CELLS;
    const int cell = static_cast<int>(tok - b * CELLS);
    int64_t src = gather != nullptr ? gather[b] : b;             // row of `feat` that sample b of the compact batch shows
    if (src < 0 || src >= rows_total) src = 0;
    const float own = feat[src * 3 * CELLS + cell], opp = feat[src * 3 * CELLS + CELLS + cell];
    float p[8], a[8], o[8], r[8];
    unpack8(*reinterpret_cast<const V8 *>(pos + cell * E + vec * 8), p);
    unpack8(*reinterpret_cast<const V8 *>(e_own + vec * 8), a);
    unpack8(*reinterpret_cast<const V8 *>(e_opp + vec * 8), o);
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = p[i] + own * a[i] + opp * o[i];
    *reinterpret_cast<V8 *>(tokens + tok * E + vec * 8) = pack8(r);
}

// ---------------------------------------------------------------- GroupNorm(1, C) + affine
// One wavefront per sample: the sample's 42*C values (C = 64: 336 vectors of 8) live in
// registers between the statistics pass and the normalise pass.
template <int C>
__global__ void __launch_bounds__(64) k_groupnorm1(const uint16_t *x, const uint16_t *gamma, const uint16_t *beta,
                                                   uint16_t *y, int64_t B, float eps)
{
    constexpr int NV = CELLS * C / 8;                 // 336
    constexpr int PER = (NV + 63) / 64;               // 6
    const int64_t b = blockIdx.x;
    if (b >= B) return;
    const int lane = threadIdx.x;
    const uint16_t *xs = x + b * (CELLS * C);
    float v[PER][8];
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int vi = lane + 64 * k;
        if (vi < NV) {
            unpack8(*reinterpret_cast<const V8 *>(xs + vi * 8), v[k]);
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += v[k][i];
        }
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum * (1.0f / (CELLS * C));
    float sq = 0.0f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int vi = lane + 64 * k;
        if (vi < NV) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[k][i] - mean; sq += d * d; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq * (1.0f / (CELLS * C)) + eps);
    uint16_t *ys = y + b * (CELLS * C);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int vi = lane + 64 * k;
        if (vi < NV) {
            const int ch = (vi * 8) % C;
            float g[8], be[8], r[8];
            unpack8(*reinterpret_cast<const V8 *>(gamma + ch), g);
            unpack8(*reinterpret_cast<const V8 *>(beta + ch), be);
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = (v[k][i] - mean) * rstd * g[i] + be[i];
            *reinterpret_cast<V8 *>(ys + vi * 8) = pack8(r);
        }
    }
}

// ---------------------------------------------------------------- y = residual + silu(x + bias)
__global__ void __launch_bounds__(256) k_silu_add(const uint16_t *x, const uint16_t *bias, int channels,
                                                  const uint16_t *res, uint16_t *y, int64_t nvec)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    float a[8], r[8];
    unpack8(reinterpret_cast<const V8 *>(x)[i], a);
    if (bias) {
        float bb[8];
        unpack8(*reinterpret_cast<const V8 *>(bias + (i * 8) % channels), bb);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __bfloat162float(__float2bfloat16(a[k] + bb[k]));   // conv output is bf16
    }
    if (res) {
        unpack8(reinterpret_cast<const V8 *>(res)[i], r);
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] += silu(a[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = silu(a[k]);
    }
    reinterpret_cast<V8 *>(y)[i] = pack8(r);
}

// ---------------------------------------------------------------- RMSNorm over C = 64
// 8 lanes per row (nn.RMSNorm: x * rsqrt(mean(x^2) + eps) * w)
__global__ void __launch_bounds__(256) k_rmsnorm64(const uint16_t *x, const uint16_t *w, uint16_t *y, int64_t rows,
                                                   float eps)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t row = gid >> 3;
    const int vec = static_cast<int>(gid & 7);
    const bool live = row < rows;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (live) unpack8(*reinterpret_cast<const V8 *>(x + row * 64 + vec * 8), a);
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += a[i] * a[i];
    ss += __shfl_xor(ss, 1, 8); ss += __shfl_xor(ss, 2, 8); ss += __shfl_xor(ss, 4, 8);
    const float r = rsqrtf(ss * (1.0f / 64.0f) + eps);
    if (!live) return;
    float g[8];
    unpack8(*reinterpret_cast<const V8 *>(w + vec * 8), g);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = a[i] * r * g[i];
    *reinterpret_cast<V8 *>(y + row * 64 + vec * 8) = pack8(a);
}

// ---------------------------------------------------------------- attention input prep
// qkvg (T, 3*64 + 4) -> q, k, v as (B, heads=4, 42, 16) contiguous with per-head RMSNorm on q
// and k (Network.py:66-71), gate -> sigmoid(gate) as (T, 4).  32 lanes per token: lanes 0-23
// one 8-vector each (part = lane/8, head = (lane%8)/2, half = lane%2), lanes 24-27 one gate.
__global__ void __launch_bounds__(256) k_qkv_prep(const uint16_t *qkvg, int ROWLEN, const uint16_t *qn_w,
                                                  const uint16_t *kn_w, uint16_t *q, uint16_t *k, uint16_t *v,
                                                  uint16_t *gate_sig, int64_t B, float eps)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t tok = gid >> 5;
    const int l = static_cast<int>(gid & 31);
    const bool live = tok < B * CELLS;
    const uint16_t *row = qkvg + tok * ROWLEN;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (live && l < 24) {
        if ((ROWLEN & 7) == 0) {                      // 16-byte aligned rows (padded projection)
            unpack8(*reinterpret_cast<const V8 *>(row + l * 8), a);
        } else {                                      // 196-wide rows are only 8-byte aligned
            const uint32_t *p = reinterpret_cast<const uint32_t *>(row + l * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const uint32_t wv = p[i]; a[2 * i] = bf_lo(wv); a[2 * i + 1] = bf_hi(wv); }
        }
    }
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += a[i] * a[i];
    ss += __shfl_xor(ss, 1, 32);                       // the two halves of a head
    if (!live) return;
    const int64_t b = tok / CELLS;
    const int t = static_cast<int>(tok - b * CELLS);
    if (l < 24) {
        const int part = l >> 3, head = (l & 7) >> 1, half = l & 1;
        if (part < 2) {
            const float r = rsqrtf(ss * (1.0f / 16.0f) + eps);
            float g[8];
            unpack8(*reinterpret_cast<const V8 *>((part == 0 ? qn_w : kn_w) + half * 8), g);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = a[i] * r * g[i];
        }
        uint16_t *dst = (part == 0 ? q : (part == 1 ? k : v)) + ((b * 4 + head) * CELLS + t) * 16 + half * 8;
        *reinterpret_cast<V8 *>(dst) = pack8(a);
    } else if (l < 28) {
        const int h = l - 24;
        const float gte = sigmoidf(bf1(row + 192 + h));
        const __hip_bfloat16 o = __float2bfloat16(gte);
        gate_sig[tok * 4 + h] = *reinterpret_cast<const uint16_t *>(&o);
    }
}

// ---------------------------------------------------------------- attention output gather
// out[tok, h*16 + d] = a[b, h, t, d] * gate_sig[tok, h]            (Network.py:80-82)
__global__ void __launch_bounds__(256) k_attn_post(const uint16_t *a, const uint16_t *gate_sig, uint16_t *out,
                                                   int64_t B)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t tok = gid >> 3;
    const int vec = static_cast<int>(gid & 7);
    if (tok >= B * CELLS) return;
    const int64_t b = tok / CELLS;
    const int t = static_cast<int>(tok - b * CELLS);
    const int head = vec >> 1, half = vec & 1;
    float f[8];
    unpack8(*reinterpret_cast<const V8 *>(a + ((b * 4 + head) * CELLS + t) * 16 + half * 8), f);
    const float g = bf1(gate_sig + tok * 4 + head);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] *= g;
    *reinterpret_cast<V8 *>(out + tok * 64 + vec * 8) = pack8(f);
}

// ---------------------------------------------------------------- head pooling
// Per sample (one wavefront): RMSNorm of the 42 tokens with the policy head's norm weight,
// row-gate score per token, softmax over the 6 rows of every column, weighted column sum
// (Network.py:107-113) -> col (B, 7, 64); and the plain token mean for the value head
// (Network.py:135) -> mean (B, 64).
__global__ void __launch_bounds__(64) k_heads_prep(const uint16_t *tok, const uint16_t *p_norm_w,
                                                   const uint16_t *p_gate_w, float p_gate_b, uint16_t *col,
                                                   uint16_t *mean, int64_t B, float eps)
{
    __shared__ float s_pn[CELLS * 64];
    __shared__ float s_score[CELLS + 6];
    const int64_t b = blockIdx.x;
    if (b >= B) return;
    const int lane = threadIdx.x;
    const int sub = lane >> 3, vec = lane & 7;             // token slot (0-7), channel vector
    const uint16_t *xs = tok + b * (CELLS * 64);
    float nw[8], gw[8];
    unpack8(*reinterpret_cast<const V8 *>(p_norm_w + vec * 8), nw);
    unpack8(*reinterpret_cast<const V8 *>(p_gate_w + vec * 8), gw);
    float msum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int t = sub + 8 * k;
        const bool live = t < CELLS;
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (live) unpack8(*reinterpret_cast<const V8 *>(xs + t * 64 + vec * 8), a);
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { ss += a[i] * a[i]; msum[i] += a[i]; }
        ss += __shfl_xor(ss, 1, 8); ss += __shfl_xor(ss, 2, 8); ss += __shfl_xor(ss, 4, 8);
        const float r = rsqrtf(ss * (1.0f / 64.0f) + eps);
        float sc = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // the reference rounds the normalised token to bf16 before the gate and the pooling
            const float pn = __bfloat162float(__float2bfloat16(a[i] * r * nw[i]));
            a[i] = pn;
            sc += pn * gw[i];
        }
        sc += __shfl_xor(sc, 1, 8); sc += __shfl_xor(sc, 2, 8); sc += __shfl_xor(sc, 4, 8);
        if (live) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s_pn[t * 64 + vec * 8 + i] = a[i];
            if (vec == 0) s_score[t] = sc + p_gate_b;
        }
    }
    __syncthreads();
    // softmax over rows, per column; then weighted sum over rows: 7*64 outputs, 7 per lane
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        float sc[ROWS], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { sc[r] = s_score[r * COLS + c]; mx = fmaxf(mx, sc[r]); }
        float den = 0.0f;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { sc[r] = __expf(sc[r] - mx); den += sc[r]; }
        const float inv = 1.0f / den;
        float acc = 0.0f;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const float w = __bfloat162float(__float2bfloat16(sc[r] * inv));
            acc += w * s_pn[(r * COLS + c) * 64 + lane];
        }
        const __hip_bfloat16 o = __float2bfloat16(acc);
        col[(b * COLS + c) * 64 + lane] = *reinterpret_cast<const uint16_t *>(&o);
    }
    // token mean: reduce the 8 token slots that share a channel vector
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float m = msum[i];
        m += __shfl_xor(m, 8, 64); m += __shfl_xor(m, 16, 64); m += __shfl_xor(m, 32, 64);
        msum[i] = m * (1.0f / CELLS);
    }
    if (sub == 0) *reinterpret_cast<V8 *>(mean + b * 64 + vec * 8) = pack8(msum);
}

inline unsigned blocks(int64_t threads, int per) { return static_cast<unsigned>((threads + per - 1) / per); }
inline const uint16_t *u16(const void *p) { return static_cast<const uint16_t *>(p); }
inline uint16_t *u16(void *p) { return static_cast<uint16_t *>(p); }

}  // namespace

extern "C" {

int az_nn_embed(const float *features, const void *emb_own, const void *emb_opp, const void *pos, void *tokens,
                int64_t batch, int embed_dim, const int32_t *gather, const int64_t *batch_dev, void *stream)
{
    if (embed_dim != 32 || batch <= 0) return 1;
    hipLaunchKernelGGL(k_embed<32>, dim3(blocks(batch * CELLS * 4, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       features, u16(emb_own), u16(emb_opp), u16(pos), u16(tokens), batch, gather, batch_dev);
    return 0;
}

int az_nn_groupnorm1(const void *x, const void *gamma, const void *beta, void *y, int64_t batch, int channels,
                     float eps, void *stream)
{
    if (channels != 64 || batch <= 0) return 1;
    hipLaunchKernelGGL(k_groupnorm1<64>, dim3(static_cast<unsigned>(batch)), dim3(64), 0,
                       static_cast<hipStream_t>(stream), u16(x), u16(gamma), u16(beta), u16(y), batch, eps);
    return 0;
}

int az_nn_silu_add(const void *x, const void *bias, int channels, const void *residual, void *y,
                   int64_t n_elements, void *stream)
{
    if (n_elements <= 0 || (n_elements & 7) || (bias && (channels <= 0 || (channels & 7)))) return 1;
    const int64_t nvec = n_elements / 8;
    hipLaunchKernelGGL(k_silu_add, dim3(blocks(nvec, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), u16(x),
                       u16(bias), channels, u16(residual), u16(y), nvec);
    return 0;
}

int az_nn_rmsnorm64(const void *x, const void *w, void *y, int64_t rows, float eps, void *stream)
{
    if (rows <= 0) return 1;
    hipLaunchKernelGGL(k_rmsnorm64, dim3(blocks(rows * 8, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), u16(x),
                       u16(w), u16(y), rows, eps);
    return 0;
}

int az_nn_qkv_prep(const void *qkvg, int row_len, const void *q_norm_w, const void *k_norm_w, void *q, void *k,
                   void *v, void *gate_sigmoid, int64_t batch, float eps, void *stream)
{
    if (batch <= 0 || (row_len != 196 && row_len != 200)) return 1;
    hipLaunchKernelGGL(k_qkv_prep, dim3(blocks(batch * CELLS * 32, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), u16(qkvg), row_len, u16(q_norm_w), u16(k_norm_w), u16(q), u16(k), u16(v),
                       u16(gate_sigmoid), batch, eps);
    return 0;
}

int az_nn_attn_post(const void *attn, const void *gate_sigmoid, void *out, int64_t batch, void *stream)
{
    if (batch <= 0) return 1;
    hipLaunchKernelGGL(k_attn_post, dim3(blocks(batch * CELLS * 8, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), u16(attn), u16(gate_sigmoid), u16(out), batch);
    return 0;
}

int az_nn_heads_prep(const void *tokens, const void *p_norm_w, const void *p_gate_w, float p_gate_b, void *col,
                     void *mean, int64_t batch, float eps, void *stream)
{
    if (batch <= 0) return 1;
    hipLaunchKernelGGL(k_heads_prep, dim3(static_cast<unsigned>(batch)), dim3(64), 0, static_cast<hipStream_t>(stream),
                       u16(tokens), u16(p_norm_w), u16(p_gate_w), p_gate_b, u16(col), u16(mean), batch, eps);
    return 0;
}

}  // extern "C"
