// nn_heads.hip - both output heads of the evaluator as ONE kernel: final tokens in, the three
// arrays the tree backup consumes out.
//
//   policy  (Network.py:96-118):  pn = RMSNorm(tokens); per column, softmax over its 6 rows of
//           row_gate(pn) pools the column's tokens; logits = out(silu(fc(col))); masked softmax
//   value / moves left (Network.py:121-141):  x = mean(tokens); x += silu(pool_fc(norm(x)));
//           h = out_norm(silu(fc(norm(x)))); wdl = softmax(value_out(h)); ml = 42*sigmoid(aux_out(h))
//
// One wavefront owns one sample at a time and walks a grid-stride list of samples; the next
// sample's tokens are in flight while the current one is reduced.  The 64x64 linears run on the
// matrix cores in the orientation out^T = W . V^T: the A operand is a weight fragment (LDS,
// staged once per workgroup in fragment order), the B operand is a 16-column matrix whose
// columns 0-6 are the seven pooled policy columns and whose column 7 is the value head's
// current vector, so policy fc and value pool_fc share one operand fetch.  A single vector on
// a 16-wide tile wastes 15/16 of that MFMA and is still ~10x cheaper than the 64 LDS reads +
// 64 FMAs per lane of a VALU matvec.  Everything between the linears (RMSNorms, SiLU, bias,
// softmaxes) happens on the accumulator registers; rounding points (bf16 after every
// normalisation / linear / activation) are those of the reference under bf16 autocast.
// HBM traffic: read tokens (5376 B per sample), write 11 floats.  Replaces az_nn_heads_prep
// + ~25 small PyTorch kernels (0.33 ms per 32768-leaf iteration in profiles/r01).
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "az_nn.h"

namespace {

constexpr int CELLS = 42, ROWS = 6, COLS = 7, C = 64;
constexpr int WPB = 4;          // wavefronts (samples in flight) per workgroup
constexpr int VS = 72;          // bf16 row stride of the B-operand buffer: 144 B keeps b128 reads conflict-free
constexpr int DUAL = 7;         // B-operand column that carries the value head's vector

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
__device__ __forceinline__ float rbf(float a) { return __uint_as_float(static_cast<uint32_t>(to_bf16(a)) << 16); }
__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    return static_cast<uint32_t>(to_bf16(a)) | (static_cast<uint32_t>(to_bf16(b)) << 16);
}
__device__ __forceinline__ void unpack8(const V8 &v, float *a)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[2 * i] = bf_lo(v.w[i]); a[2 * i + 1] = bf_hi(v.w[i]); }
}
__device__ __forceinline__ bf16x8 as_bf16x8(const V8 &v)
{
    union { V8 a; bf16x8 b; } r;
    r.a = v;
    return r.b;
}
__device__ __forceinline__ float bf1(const uint16_t *p) { return __uint_as_float(static_cast<uint32_t>(*p) << 16); }
// silu of a bf16 value, evaluated in fp32 and rounded back (what the bf16 elementwise kernel does)
__device__ __forceinline__ float silu_bf(float x) { return rbf(x / (1.0f + __expf(-x))); }
__device__ __forceinline__ float col_sum(float v)   // over the 4 lane groups that share lane & 15
{
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// LDS traffic between lanes of ONE wavefront: the LDS executes a wavefront's instructions in
// order, so only the compiler has to be stopped from moving accesses across this point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

enum { K_PFC_B, K_POUT_W, K_DPOOL_B, K_DNORM, K_DFC_B, K_DOUT_NORM, K_DVAL_B, K_N };

__device__ __forceinline__ void load_tokens(V8 (&v)[6], const uint16_t *xs, int sub, int vec)
{
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int t = sub + 8 * k;
        V8 z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0;
        v[k] = t < CELLS ? *reinterpret_cast<const V8 *>(xs + t * C + vec * 8) : z;
    }
}

__global__ void __launch_bounds__(64 * WPB) k_heads(const uint16_t *tok, az_nn_heads_weights w, const uint8_t *mask,
                                                    float *probs, float *wdl, float *moves_left, int64_t B, float eps)
{
    // A fragments (fragment f, lane l -> 16 bytes at f*64+l): policy fc 0-7, pool_fc 8-15, fc 16-23
    // as [m tile][k step]; 24-25 = rows {value_out 0-2, aux_out} x k step
    __shared__ V8 s_a[26 * 64];
    __shared__ float s_c[K_N][C];                          // per-channel constants of the epilogues, fp32
    __shared__ V8 s_pn[WPB][CELLS * C / 8];                // normalised tokens, bf16
    __shared__ uint16_t s_vec[WPB][16 * VS];               // B operand: row n = column n of V^T
    __shared__ float s_score[WPB][48];
    __shared__ float s_wt[WPB][48];
    __shared__ float s_mean[WPB][C];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int sub = lane >> 3, vec = lane & 7;

    const uint16_t *mats[3] = {static_cast<const uint16_t *>(w.p_fc_w), static_cast<const uint16_t *>(w.d_pool_w),
                               static_cast<const uint16_t *>(w.d_fc_w)};
    for (int i = threadIdx.x; i < 26 * 64; i += blockDim.x) {
        const int f = i >> 6, l = i & 63, r = l & 15, q = l >> 4;
        V8 v; v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0;
        if (f < 24) {
            const int m = (f >> 1) & 3, ks = f & 1;
            v = *reinterpret_cast<const V8 *>(mats[f >> 3] + (16 * m + r) * C + 32 * ks + 8 * q);
        } else if (r < 3) {
            v = *reinterpret_cast<const V8 *>(static_cast<const uint16_t *>(w.d_val_w) + r * C + 32 * (f & 1) + 8 * q);
        } else if (r == 3) {
            v = *reinterpret_cast<const V8 *>(static_cast<const uint16_t *>(w.d_aux_w) + 32 * (f & 1) + 8 * q);
        }
        s_a[i] = v;
    }
    if (threadIdx.x < C) {
        const int i = threadIdx.x;
        s_c[K_PFC_B][i] = bf1(static_cast<const uint16_t *>(w.p_fc_b) + i);
        s_c[K_POUT_W][i] = bf1(static_cast<const uint16_t *>(w.p_out_w) + i);
        s_c[K_DPOOL_B][i] = bf1(static_cast<const uint16_t *>(w.d_pool_b) + i);
        s_c[K_DNORM][i] = bf1(static_cast<const uint16_t *>(w.d_norm) + i);
        s_c[K_DFC_B][i] = bf1(static_cast<const uint16_t *>(w.d_fc_b) + i);
        s_c[K_DOUT_NORM][i] = bf1(static_cast<const uint16_t *>(w.d_out_norm) + i);
        s_c[K_DVAL_B][i] = i < 3 ? bf1(static_cast<const uint16_t *>(w.d_val_b) + i) : 0.0f;
    }
    for (int i = lane; i < 16 * VS; i += 64) s_vec[wave][i] = 0;         // columns 8-15 stay zero
    __syncthreads();

    float nw[8], gw[8], dpn[8];
    {
        V8 t = *reinterpret_cast<const V8 *>(static_cast<const uint16_t *>(w.p_norm) + vec * 8);
        unpack8(t, nw);
        t = *reinterpret_cast<const V8 *>(static_cast<const uint16_t *>(w.p_gate_w) + vec * 8);
        unpack8(t, gw);
        t = *reinterpret_cast<const V8 *>(static_cast<const uint16_t *>(w.d_pool_norm) + vec * 8);
        unpack8(t, dpn);
    }
    auto afrag = [&](int f) { return as_bf16x8(s_a[f * 64 + lane]); };
    auto bfrag = [&](int ks) {
        return as_bf16x8(*reinterpret_cast<const V8 *>(&s_vec[wave][l15 * VS + 32 * ks + 8 * l4]));
    };
    auto cvec = [&](int which, int m) { return *reinterpret_cast<const f32x4 *>(&s_c[which][16 * m + 4 * l4]); };
    // the wavefront's 16 accumulator values of column `l15` -> s_vec row DUAL (channel 16m+4*l4+reg)
    auto put_dual = [&](const float (&v)[4][4]) {
        if (l15 == DUAL) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                uint32_t *p = reinterpret_cast<uint32_t *>(&s_vec[wave][DUAL * VS + 16 * m + 4 * l4]);
                p[0] = pack2(v[m][0], v[m][1]);
                p[1] = pack2(v[m][2], v[m][3]);
            }
        }
    };

    const int64_t stride = static_cast<int64_t>(gridDim.x) * WPB;
    int64_t b = static_cast<int64_t>(blockIdx.x) * WPB + wave;
    V8 cur[6];
    if (b < B) load_tokens(cur, tok + b * (CELLS * C), sub, vec);
    for (; b < B; b += stride) {
        V8 nxt[6];
        load_tokens(nxt, tok + (b + stride < B ? b + stride : b) * (CELLS * C), sub, vec);

        // ---- pass over the tokens: policy RMSNorm + row-gate score, and the channel sums
        float msum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int t = sub + 8 * k;
            float a[8];
            unpack8(cur[k], a);
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { ss += a[i] * a[i]; msum[i] += a[i]; }
            ss += __shfl_xor(ss, 1, 8); ss += __shfl_xor(ss, 2, 8); ss += __shfl_xor(ss, 4, 8);
            const float r = rsqrtf(ss * (1.0f / C) + eps);
            float sc = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a[i] = rbf(a[i] * r * nw[i]);
                sc += a[i] * gw[i];
            }
            sc += __shfl_xor(sc, 1, 8); sc += __shfl_xor(sc, 2, 8); sc += __shfl_xor(sc, 4, 8);
            if (t < CELLS) {
                V8 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o.w[i] = (__float_as_uint(a[2 * i]) >> 16) | (__float_as_uint(a[2 * i + 1]) & 0xffff0000u);
                s_pn[wave][t * 8 + vec] = o;
                if (vec == 0) s_score[wave][t] = sc + w.p_gate_b;
            }
        }
        // token mean (bf16, as the reference's mean over a bf16 tensor) and its pool_norm
        float ssm = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float m = msum[i];
            m += __shfl_xor(m, 8, 64); m += __shfl_xor(m, 16, 64); m += __shfl_xor(m, 32, 64);
            msum[i] = rbf(m * (1.0f / CELLS));
            ssm += msum[i] * msum[i];
        }
        ssm += __shfl_xor(ssm, 1, 8); ssm += __shfl_xor(ssm, 2, 8); ssm += __shfl_xor(ssm, 4, 8);
        {
            const float r = rsqrtf(ssm * (1.0f / C) + eps);
            if (sub == 0) {
                uint32_t *p = reinterpret_cast<uint32_t *>(&s_vec[wave][DUAL * VS + vec * 8]);
#pragma unroll
                for (int i = 0; i < 4; ++i) p[i] = pack2(msum[2 * i] * r * dpn[2 * i], msum[2 * i + 1] * r * dpn[2 * i + 1]);
#pragma unroll
                for (int i = 0; i < 8; ++i) s_mean[wave][vec * 8 + i] = msum[i];
            }
        }
        wave_lds_sync();

        // ---- softmax over the 6 rows of each column: lane t owns token t's weight
        if (lane < CELLS) {
            const int c = lane % COLS;
            float sc[ROWS], mx = -INFINITY, den = 0.0f;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { sc[r] = s_score[wave][r * COLS + c]; mx = fmaxf(mx, sc[r]); }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) den += __expf(sc[r] - mx);
            s_wt[wave][lane] = rbf(__expf(s_score[wave][lane] - mx) / den);
        }
        wave_lds_sync();
        // ---- weighted column sums: lane = channel
        {
            const uint16_t *pn = reinterpret_cast<const uint16_t *>(s_pn[wave]);
#pragma unroll
            for (int c = 0; c < COLS; ++c) {
                float acc = 0.0f;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) acc += s_wt[wave][r * COLS + c] * bf1(pn + (r * COLS + c) * C + lane);
                s_vec[wave][c * VS + lane] = to_bf16(acc);
            }
        }
        wave_lds_sync();

        // ---- policy fc (columns 0-6) and value pool_fc (column 7) share the B operand
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 ap[4], ad[4];
        {
            const bf16x8 b0 = bfrag(0), b1 = bfrag(1);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                ap[m] = MFMA32(afrag(2 * m), b0, zero);
                ap[m] = MFMA32(afrag(2 * m + 1), b1, ap[m]);
                ad[m] = MFMA32(afrag(8 + 2 * m), b0, zero);
                ad[m] = MFMA32(afrag(8 + 2 * m + 1), b1, ad[m]);
            }
        }
        // policy: logit[c] = out . silu(fc(col_c) + b), masked softmax over the 7 columns
        {
            float part = 0.0f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 fb = cvec(K_PFC_B, m), ow = cvec(K_POUT_W, m);
#pragma unroll
                for (int r = 0; r < 4; ++r) part += silu_bf(rbf(ap[m][r] + fb[r])) * ow[r];
            }
            float logit = col_sum(part) + w.p_out_b;
            const bool live = l15 < COLS;
            if (live && mask != nullptr && mask[b * COLS + l15] == 0) logit = -1e9f;
            if (!live) logit = -INFINITY;
            float mx = logit;
            mx = fmaxf(mx, __shfl_xor(mx, 1, 16)); mx = fmaxf(mx, __shfl_xor(mx, 2, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 4, 16)); mx = fmaxf(mx, __shfl_xor(mx, 8, 16));
            const float e = live ? __expf(logit - mx) : 0.0f;
            float den = e;
            den += __shfl_xor(den, 1, 16); den += __shfl_xor(den, 2, 16);
            den += __shfl_xor(den, 4, 16); den += __shfl_xor(den, 8, 16);
            if (live && l4 == 0) probs[b * COLS + l15] = e / den;
        }
        // value head, stage 1 (column 7): g = mean + silu(pool_fc(pool_norm(mean)) + b); n2 = norm(g)
        float g[4][4];
        {
            float ss = 0.0f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 pb = cvec(K_DPOOL_B, m);
                const f32x4 mean = *reinterpret_cast<const f32x4 *>(&s_mean[wave][16 * m + 4 * l4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    g[m][r] = rbf(mean[r] + silu_bf(rbf(ad[m][r] + pb[r])));
                    ss += g[m][r] * g[m][r];
                }
            }
            const float rn = rsqrtf(col_sum(ss) * (1.0f / C) + eps);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 nwt = cvec(K_DNORM, m);
#pragma unroll
                for (int r = 0; r < 4; ++r) g[m][r] = g[m][r] * rn * nwt[r];
            }
        }
        wave_lds_sync();
        put_dual(g);
        wave_lds_sync();
        // stage 2: h = out_norm(silu(fc(n2) + b))
        {
            const bf16x8 b0 = bfrag(0), b1 = bfrag(1);
            float ss = 0.0f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 acc = MFMA32(afrag(16 + 2 * m), b0, zero);
                acc = MFMA32(afrag(16 + 2 * m + 1), b1, acc);
                const f32x4 fb = cvec(K_DFC_B, m);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    g[m][r] = silu_bf(rbf(acc[r] + fb[r]));
                    ss += g[m][r] * g[m][r];
                }
            }
            const float rn = rsqrtf(col_sum(ss) * (1.0f / C) + eps);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 nwt = cvec(K_DOUT_NORM, m);
#pragma unroll
                for (int r = 0; r < 4; ++r) g[m][r] = g[m][r] * rn * nwt[r];
            }
        }
        wave_lds_sync();
        put_dual(g);
        wave_lds_sync();
        // stage 3: rows 0-2 = value logits, row 3 = moves-left logit, all in lane (l4 0, l15 7)
        {
            f32x4 acc = MFMA32(afrag(24), bfrag(0), zero);
            acc = MFMA32(afrag(25), bfrag(1), acc);
            if (lane == DUAL) {
                const float v0 = rbf(acc[0] + s_c[K_DVAL_B][0]), v1 = rbf(acc[1] + s_c[K_DVAL_B][1]),
                            v2 = rbf(acc[2] + s_c[K_DVAL_B][2]);
                const float mx = fmaxf(v0, fmaxf(v1, v2));
                const float e0 = __expf(v0 - mx), e1 = __expf(v1 - mx), e2 = __expf(v2 - mx);
                const float inv = 1.0f / (e0 + e1 + e2);
                wdl[b * 3 + 0] = e0 * inv;
                wdl[b * 3 + 1] = e1 * inv;
                wdl[b * 3 + 2] = e2 * inv;
                moves_left[b] = w.aux_scale / (1.0f + __expf(-(acc[3] + w.d_aux_b)));
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 6; ++k) cur[k] = nxt[k];
    }
}

}  // namespace

extern "C" int az_nn_heads(const void *tokens, const az_nn_heads_weights *w, const uint8_t *mask, float *probs,
                           float *wdl, float *moves_left, int64_t batch, float eps, void *stream)
{
    if (batch <= 0 || w == nullptr || tokens == nullptr || probs == nullptr || wdl == nullptr || moves_left == nullptr) return 1;
    // 60 KB of LDS per workgroup: two workgroups (8 wavefronts) per CU, each walking its samples
    const int64_t want = (batch + WPB - 1) / WPB;
    const unsigned grid = static_cast<unsigned>(want < 512 ? want : 512);
    hipLaunchKernelGGL(k_heads, dim3(grid), dim3(64 * WPB), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(tokens), *w, mask, probs, wdl, moves_left, batch, eps);
    return 0;
}
