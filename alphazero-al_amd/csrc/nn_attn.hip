// nn_attn.hip - the evaluator's gated attention block as ONE MFMA kernel.
//
//   y = x + o_proj( sigmoid(gate) * softmax(q_norm(Q) k_norm(K)^T / sqrt(16)) V )
//   with [Q | K | V | gate] = qkvg_proj(RMSNorm(x))                     (Network.py:51-93)
//
// One wavefront owns one sample (42 tokens padded to 3 row tiles of 16) and keeps every
// intermediate in registers.  The point of the design is the ORIENTATION of each product:
// with C/D in the MFMA layout (column = lane & 15, rows = 4*(lane>>4)+reg) a result can feed
// the next MFMA with no lane movement if that product sums over its ROW index.  So
//
//   Q^T, K^T, gate^T = W . H^T      (features x tokens; A = weights, B = H^T)   v_mfma 16x16x32
//   V               = H . Wv^T      (tokens x features; A = H, B = Wv^T)        v_mfma 16x16x32
//   S^T             = K . Q^T       (keys x queries;  A = K^T regs, B = Q^T regs) 16x16x16
//   O^T             = V^T . P^T     (d x queries;     A = V regs,   B = P^T regs) 16x16x16
//   out^T           = Wo_h . O_h^T  summed over heads (A = weights, B = O^T regs) 16x16x16
//
// The per-(token, head) RMSNorm of q and k and the softmax over keys reduce over rows, i.e.
// over the 4 registers of a lane and the 4 lane groups that share a column: two xor-shuffles.
// H fragments serve both as B operand (H^T) and as A operand (H): the element sets coincide.
// The weights (25 KB + 8 KB) sit in LDS in operand-fragment order.  HBM traffic: read x,
// write y.  Replaces RMSNorm + 196-wide GEMM + split/normalise + SDPA + gate + out-projection
// (six kernels, 1.8 ms per 32768-leaf iteration in the first profile).
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "az_nn.h"

namespace {

constexpr int CELLS = 42, C = 64, HEADS = 4, HD = 16, TT = 3;     // 3 token tiles of 16

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct alignas(16) V8 { uint32_t w[4]; };
struct alignas(8) V4 { uint32_t w[2]; };

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// one v_cvt_pk_bf16_f32 (round to nearest even, NaN preserving)
__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(pk_f32x2{a, b}, pk_bf16x2));
}
__device__ __forceinline__ f32x2 unpack2(uint32_t w) { return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
__device__ __forceinline__ s16x4 to_s16x4(const f32x4 &v)
{
    union { uint32_t u[2]; s16x4 s; } r;
    r.u[0] = pack2(v[0], v[1]);
    r.u[1] = pack2(v[2], v[3]);
    return r.s;
}
__device__ __forceinline__ bf16x8 as_bf16x8(const V8 &v)
{
    union { V8 a; bf16x8 b; } r;
    r.a = v;
    return r.b;
}
__device__ __forceinline__ float bf1(const uint16_t *p) { return __uint_as_float(static_cast<uint32_t>(*p) << 16); }
// Reductions over the 4 lane groups that share lane & 15 (the rows of a 16-lane column) on the vector ALU:
// v_permlane16_swap / v_permlane32_swap (gfx950) hand every lane its partner across rows {0,1},{2,3} and
// across the wavefront's halves - two instructions per step where __shfl_xor takes a trip through the LDS
// crossbar (ds_bpermute + address + wait) each.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float col_sum(float v)   // sum over the 4 lane groups that share lane & 15
{
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float col_max(float v)
{
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

// qkvg: (196, 64) row-major [out][in]: rows 0-63 Q, 64-127 K, 128-191 V, 192-195 gate
template <int MINB>      // workgroups per CU the register budget is cut for (2: 190 VGPRs, no spills; 3, the default: 168 + 13 spilled)
__global__ void __launch_bounds__(256, MINB) k_attn_block(const uint16_t *x, const uint16_t *pre_w, const uint16_t *qkvg,
                                                    const uint16_t *qn_w, const uint16_t *kn_w, const uint16_t *o_w,
                                                    uint16_t *y, int64_t B, float eps, const int64_t *batch_dev)
{
    if (batch_dev != nullptr && *batch_dev < B) B = *batch_dev;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- weights: staged once per workgroup into LDS in exactly the order the lanes read them
    // (fragment f, lane l -> 16 or 8 contiguous bytes at f*64+l), so every operand fetch is one
    // conflict-free ds_read.  Keeping them out of the register file is what lets two
    // wavefronts share a SIMD (the kernel is bound by dependent VALU/MFMA latency, not by LDS).
    __shared__ V8 s_w32[(3 * HEADS * 2 + 2) * 64];          // wq, wk, wv [h][s], wg [s]
    __shared__ V4 s_w16[4 * HEADS * 64];                    // wo [ot][h]
    for (int i = threadIdx.x; i < (3 * HEADS * 2 + 2) * 64; i += blockDim.x) {
        const int f = i >> 6, l = i & 63, ll15 = l & 15, ll4 = l >> 4;
        V8 v; v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0;
        if (f < 3 * HEADS * 2) {
            const int part = f / (HEADS * 2), h = (f >> 1) % HEADS, sk = f & 1;
            v = *reinterpret_cast<const V8 *>(qkvg + (part * C + h * HD + ll15) * C + 32 * sk + 8 * ll4);
        } else {
            // the 4 gate rows, repeated four times over the tile's 16 rows: accumulator register r of
            // EVERY lane group is then head r of the lane's token - no lane has to ask another for it
            v = *reinterpret_cast<const V8 *>(qkvg + (3 * C + (ll15 & 3)) * C + 32 * (f & 1) + 8 * ll4);
        }
        s_w32[i] = v;
    }
    for (int i = threadIdx.x; i < 4 * HEADS * 64; i += blockDim.x) {
        const int f = i >> 6, l = i & 63, ot = f / HEADS, h = f % HEADS;
        s_w16[i] = *reinterpret_cast<const V4 *>(o_w + (ot * 16 + (l & 15)) * C + h * HD + 4 * (l >> 4));
    }
    // MINB >= 3 (three or four wavefronts per SIMD): what the two-per-SIMD kernel keeps in registers across a whole
    // sample - the sigmoid gates of every (token, head), the pre-norm weight, the q / k norm weights - lives in LDS
    // instead (36 registers: the variant then fits its budget without scratch)
    constexpr bool LEAN = MINB >= 3;
    __shared__ f32x4 s_gate[LEAN ? 4 * TT * 64 : 1];      // [wave][token tile][lane]: gate of heads 0..3
    __shared__ float s_pw[LEAN ? C : 1];                   // pre-norm weight
    __shared__ float s_qk[LEAN ? 2 * HD : 1];              // q norm weight x QSCALE, k norm weight
    if (LEAN) {
        if (threadIdx.x < C) s_pw[threadIdx.x] = bf1(pre_w + threadIdx.x);
        if (threadIdx.x < HD) {
            s_qk[threadIdx.x] = bf1(qn_w + threadIdx.x) * (0.25f * 1.44269504f);
            s_qk[HD + threadIdx.x] = bf1(kn_w + threadIdx.x);
        }
    }
    __syncthreads();
    // (the head loop below is not unrolled and indexes these by the runtime head number, so the
    // reads stay ds_read_b128 / ds_read_b64 inside the loop instead of becoming live registers)
    auto frag32 = [&](int f) { return as_bf16x8(s_w32[f * 64 + lane]); };     // part*8 + h*2 + s ; gate: 24 + s
    auto frag16 = [&](int f) {
        union { V4 v; s16x4 s; } r;
        r.v = s_w16[f * 64 + lane];
        return r.s;
    };
    f32x2 pw[2][4];                           // prenorm weight of this lane's 16 input channels
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) pw[s][j] = f32x2{bf1(pre_w + 32 * s + 8 * l4 + 2 * j), bf1(pre_w + 32 * s + 8 * l4 + 2 * j + 1)};
    // per-head norm weights of this lane's rows d = 4*l4 + r.  The 1/sqrt(16) of the scores and
    // the log2(e) of their softmax ride on q: the scores come out of the MFMA ready for exp2.
    constexpr float QSCALE = 0.25f * 1.44269504f;
    f32x2 qnw[2], knw[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        qnw[r] = f32x2{bf1(qn_w + 4 * l4 + 2 * r) * QSCALE, bf1(qn_w + 4 * l4 + 2 * r + 1) * QSCALE};
        knw[r] = f32x2{bf1(kn_w + 4 * l4 + 2 * r), bf1(kn_w + 4 * l4 + 2 * r + 1)};
    }

    // bound of the scores in log2 units (see the softmax below)
    float mq = 0.0f, mk = 0.0f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        mq = fmaxf(mq, fmaxf(fabsf(qnw[r].x), fabsf(qnw[r].y)));
        mk = fmaxf(mk, fmaxf(fabsf(knw[r].x), fabsf(knw[r].y)));
    }
    const bool bounded = 16.0f * col_max(mq) * col_max(mk) < 100.0f;

    const int64_t stride = static_cast<int64_t>(gridDim.x) * 4;
    for (int64_t b = static_cast<int64_t>(blockIdx.x) * 4 + wave; b < B; b += stride) {
        const uint16_t *xs = x + b * (CELLS * C);

        // ---- H = RMSNorm(x) * w, as MFMA fragments (token = tile*16 + lane&15, 8 channels per k-step)
        bf16x8 hf[TT][2];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int tok = tt * 16 + l15;
            f32x2 f[2][4], ss2 = {0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                V8 v; v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0;
                if (tok < CELLS) v = *reinterpret_cast<const V8 *>(xs + tok * C + 32 * s + 8 * l4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f[s][i] = unpack2(v.w[i]);
                    ss2 = __builtin_elementwise_fma(f[s][i], f[s][i], ss2);
                }
            }
            const float ss = col_sum(ss2.x + ss2.y);
            const float r = rsqrtf(ss * (1.0f / C) + eps);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                V8 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 pwv = LEAN ? *reinterpret_cast<const f32x2 *>(&s_pw[32 * s + 8 * l4 + 2 * i]) : pw[s][i];
                    const f32x2 hv = f[s][i] * f32x2{r, r} * pwv;
                    o.w[i] = pack2(hv.x, hv.y);
                }
                hf[tt][s] = as_bf16x8(o);
            }
        }

        // ---- gate logits of every token (gate tile rows 4q + h = head h: register h of every lane)
        float gate[TT][HEADS];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            f32x4 g = MFMA32(frag32(24), hf[tt][0], zero);
            g = MFMA32(frag32(25), hf[tt][1], g);
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
                gate[tt][h] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * g[h]));
            }
            if (LEAN) s_gate[(wave * TT + tt) * 64 + lane] = f32x4{gate[tt][0], gate[tt][1], gate[tt][2], gate[tt][3]};
        }

        f32x4 out[4][TT];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int qt = 0; qt < TT; ++qt) out[ot][qt] = zero;

        // ---- one head at a time: projection, attention, contribution to the output projection.
        // Not unrolled: the four heads share the code and, more importantly, the registers.
#pragma unroll 1
        for (int h = 0; h < HEADS; ++h) {
            s16x4 qb[TT], kb[TT], vb[TT];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                f32x4 q = MFMA32(frag32(h * 2), hf[tt][0], zero);
                q = MFMA32(frag32(h * 2 + 1), hf[tt][1], q);
                f32x4 k = MFMA32(frag32(8 + h * 2), hf[tt][0], zero);
                k = MFMA32(frag32(8 + h * 2 + 1), hf[tt][1], k);
                f32x4 v = MFMA32(hf[tt][0], frag32(16 + h * 2), zero);
                v = MFMA32(hf[tt][1], frag32(16 + h * 2 + 1), v);
                // per-(token, head) RMSNorm over d: rows of the column this lane sits in
                f32x2 q2[2] = {{q[0], q[1]}, {q[2], q[3]}}, k2[2] = {{k[0], k[1]}, {k[2], k[3]}};
                const f32x2 qq = __builtin_elementwise_fma(q2[1], q2[1], q2[0] * q2[0]);
                const f32x2 kk = __builtin_elementwise_fma(k2[1], k2[1], k2[0] * k2[0]);
                const float qs = col_sum(qq.x + qq.y), ks = col_sum(kk.x + kk.y);
                const float qr = rsqrtf(qs * (1.0f / HD) + eps), kr = rsqrtf(ks * (1.0f / HD) + eps);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const f32x2 qw = LEAN ? *reinterpret_cast<const f32x2 *>(&s_qk[4 * l4 + 2 * r]) : qnw[r];
                    const f32x2 kw = LEAN ? *reinterpret_cast<const f32x2 *>(&s_qk[HD + 4 * l4 + 2 * r]) : knw[r];
                    q2[r] = q2[r] * f32x2{qr, qr} * qw; k2[r] = k2[r] * f32x2{kr, kr} * kw;
                }
                qb[tt] = to_s16x4(f32x4{q2[0].x, q2[0].y, q2[1].x, q2[1].y});
                kb[tt] = to_s16x4(f32x4{k2[0].x, k2[0].y, k2[1].x, k2[1].y});
                vb[tt] = to_s16x4(v);
            }
#pragma unroll
            for (int qt = 0; qt < TT; ++qt) {
                // S^T tile rows = keys, column = query lane&15
                f32x4 st[TT];
                float den;
                if (bounded) {
                    // |score| <= 16 max|q_norm w| max|k_norm w| (q and k are RMS-normalised) is far from
                    // fp32's exp2 range: no running maximum.  The six padding keys have k = 0, i.e.
                    // score 0 and weight exp2(0) = 1 exactly, and their V rows are 0: they add nothing
                    // to the product and exactly 6 to the denominator.
                    f32x2 den2 = {0.0f, 0.0f};
#pragma unroll
                    for (int kt = 0; kt < TT; ++kt) {
                        st[kt] = MFMA16(kb[kt], qb[qt], zero);      // already in log2 units (QSCALE)
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            const f32x2 e = {__builtin_amdgcn_exp2f(st[kt][r]), __builtin_amdgcn_exp2f(st[kt][r + 1])};
                            st[kt][r] = e.x;
                            st[kt][r + 1] = e.y;
                            den2 += e;
                        }
                    }
                    den = col_sum(den2.x + den2.y) - static_cast<float>(TT * 16 - CELLS);
                } else {
                    float m = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < TT; ++kt) {
                        st[kt] = MFMA16(kb[kt], qb[qt], zero);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (kt == TT - 1 && kt * 16 + 4 * l4 + r >= CELLS) st[kt][r] = -INFINITY;   // padding keys
                            m = fmaxf(m, st[kt][r]);
                        }
                    }
                    m = col_max(m);
                    f32x2 den2 = {0.0f, 0.0f};
                    const f32x2 nm = {-m, -m};
#pragma unroll
                    for (int kt = 0; kt < TT; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            const f32x2 d = f32x2{st[kt][r], st[kt][r + 1]} + nm;
                            const f32x2 e = {__builtin_amdgcn_exp2f(d.x), __builtin_amdgcn_exp2f(d.y)};
                            st[kt][r] = e.x;
                            st[kt][r + 1] = e.y;
                            den2 += e;
                        }
                    den = col_sum(den2.x + den2.y);
                }
                // normalise after the product: O^T = (V^T . E^T) / den, one scale per output element
                const float gq = LEAN ? reinterpret_cast<const float *>(&s_gate[(wave * TT + qt) * 64 + lane])[h]
                                      : (h == 0 ? gate[qt][0] : (h == 1 ? gate[qt][1] : (h == 2 ? gate[qt][2] : gate[qt][3])));
                const float scale = __builtin_amdgcn_rcpf(den) * gq;
                f32x4 o = zero;                                  // O^T rows = d, column = query
#pragma unroll
                for (int kt = 0; kt < TT; ++kt) o = MFMA16(vb[kt], to_s16x4(st[kt]), o);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] *= scale;
                const s16x4 ob = to_s16x4(o);
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) out[ot][qt] = MFMA16(frag16(ot * HEADS + h), ob, out[ot][qt]);
            }
        }

        // ---- y = out + x : lane holds 4 consecutive output channels of token qt*16 + lane&15
        uint16_t *ys = y + b * (CELLS * C);
#pragma unroll
        for (int qt = 0; qt < TT; ++qt) {
            const int tok = qt * 16 + l15;
            if (tok < CELLS) {
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) {
                    const int ch = ot * 16 + 4 * l4;
                    const V4 xr = *reinterpret_cast<const V4 *>(xs + tok * C + ch);
                    V4 o;
                    o.w[0] = pack2(out[ot][qt][0] + bf_lo(xr.w[0]), out[ot][qt][1] + bf_hi(xr.w[0]));
                    o.w[1] = pack2(out[ot][qt][2] + bf_lo(xr.w[1]), out[ot][qt][3] + bf_hi(xr.w[1]));
                    *reinterpret_cast<V4 *>(ys + tok * C + ch) = o;
                }
            }
        }
    }
}

}  // namespace

extern "C" {

int az_nn_attn_block(const void *x, const void *prenorm_w, const void *qkvg_w, const void *q_norm_w,
                     const void *k_norm_w, const void *o_w, void *y, int64_t batch, float eps, const int64_t *batch_dev,
                     void *stream)
{
    if (batch <= 0) return 1;
    const int64_t wgs = (batch + 3) / 4;
    const unsigned grid = static_cast<unsigned>(wgs < 1024 ? wgs : 1024);
    // three wavefronts per SIMD (168 registers, the per-sample constants and the gates in LDS: 169 us against 179 at two
    // per SIMD with everything in 190 registers; AZ_ATTN_OCC=2 / 4 select the other register budgets)
    static const int occ = [] { const char *e = getenv("AZ_ATTN_OCC"); const int v = e ? atoi(e) : 3; return (v == 2 || v == 4) ? v : 3; }();
    auto go = [&](auto kern, unsigned g) {
        hipLaunchKernelGGL(kern, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const uint16_t *>(x), static_cast<const uint16_t *>(prenorm_w),
                           static_cast<const uint16_t *>(qkvg_w), static_cast<const uint16_t *>(q_norm_w),
                           static_cast<const uint16_t *>(k_norm_w), static_cast<const uint16_t *>(o_w),
                           static_cast<uint16_t *>(y), batch, eps, batch_dev);
    };
    (void)grid;
    const unsigned cap = 256u * static_cast<unsigned>(occ) * 2u;       // two rounds of resident workgroups
    const unsigned g = static_cast<unsigned>(wgs < cap ? wgs : cap);
    if (occ == 3) go(k_attn_block<3>, g);
    else if (occ == 4) go(k_attn_block<4>, g);
    else go(k_attn_block<2>, g);
    return 0;
}

}  // extern "C"
