// nn_heads.hip - both output heads of the evaluator as ONE kernel: final tokens in, the three
// arrays the tree backup consumes out.
//
//   policy  (Network.py:96-118):  pn = RMSNorm(tokens); per column, softmax over its 6 rows of
//           row_gate(pn) pools the column's tokens; logits = out(silu(fc(col))); masked softmax
//   value / moves left (Network.py:121-141):  x = mean(tokens); x += silu(pool_fc(norm(x)));
//           h = out_norm(silu(fc(norm(x)))); wdl = softmax(value_out(h)); ml = 42*sigmoid(aux_out(h))
//
// One wavefront walks a grid-stride list of sample PAIRS; the next sample's tokens are in flight
// while the current one is reduced.  The 64x64 linears run on the matrix cores in the
// orientation out^T = W . V^T: the A operand is a weight fragment (LDS, staged once per
// workgroup in fragment order), the B operand is a 16-column matrix: columns 0-6 are the seven
// pooled policy columns of the first sample and column 7 its value-head vector, columns 8-15 the
// same for the second sample - so policy fc and value pool_fc share one operand fetch, and the
// epilogue arithmetic on the accumulators (which every lane executes whether its column is
// live or not) is paid once per two samples.  A lone vector on a 16-wide tile wastes most of
// that MFMA and is still ~10x cheaper than the 64 LDS reads + 64 FMAs per lane of a VALU
// matvec.  The token pass is packed-f32 arithmetic (two elements per VALU instruction) with
// DPP reductions; rounding points (bf16 after every normalisation / linear / activation) are
// those of the reference under bf16 autocast, except that the row gate sees the normalised
// tokens before their rounding.  The kernel is VALU-issue bound (profiles/): the first
// version spent 2.3 k vector instructions per sample, this one ~0.75 k.
// HBM traffic: read tokens (5376 B per sample), write 11 floats.  Replaces az_nn_heads_prep
// + ~25 small PyTorch kernels (0.33 ms per 32768-leaf iteration in profiles/r01).
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "az_nn.h"

namespace {

constexpr int CELLS = 42, ROWS = 6, COLS = 7, C = 64;
constexpr int WPB = 4;          // wavefronts per workgroup, each on its own sample pairs
constexpr int VS = 72;          // bf16 row stride of the B-operand buffer: 144 B keeps b128 reads conflict-free

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct alignas(16) V8 { uint32_t w[4]; };

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ f32x2 unpack2(uint32_t w) { return f32x2{bf_lo(w), bf_hi(w)}; }
__device__ __forceinline__ uint16_t to_bf16(float a)
{
    const __hip_bfloat16 x = __float2bfloat16(a);
    return *reinterpret_cast<const uint16_t *>(&x);
}
// one v_cvt_pk_bf16_f32 (round to nearest even, NaN preserving)
__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(pk_f32x2{a, b}, pk_bf16x2));
}
__device__ __forceinline__ f32x2 rbf2(f32x2 v) { return unpack2(pack2(v.x, v.y)); }       // round to bf16 and back
__device__ __forceinline__ bf16x8 as_bf16x8(const V8 &v)
{
    union { V8 a; bf16x8 b; } r;
    r.a = v;
    return r.b;
}
__device__ __forceinline__ float bf1(const uint16_t *p) { return __uint_as_float(static_cast<uint32_t>(*p) << 16); }
// silu with the hardware exp2 / reciprocal (about 1 ulp each; the result is rounded to bf16)
__device__ __forceinline__ f32x2 silu2(f32x2 x)
{
    const f32x2 t = x * f32x2{-1.44269504f, -1.44269504f};
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + f32x2{1.0f, 1.0f};
    return x * f32x2{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
}
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(1.44269504f * x); }

// lane movement as a DPP operand of the add (no LDS crossbar, no address arithmetic)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float sum8(float v)      // over the 8 lanes that share lane >> 3; result in all of them
{
    v += dpp_mov<0xB1>(v);           // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);           // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);          // row_half_mirror
    return v;
}
__device__ __forceinline__ float max8(float v)
{
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)  // over the wavefront, returned uniform
{
    v = sum8(v);
    v += dpp_mov<0x140>(v);          // row_mirror: every lane of a 16-lane row holds the row sum
    v += dpp_mov<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v += dpp_mov<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float col_sum(float v)   // over the 4 lane groups that share lane & 15
{
    // v_permlane16_swap / v_permlane32_swap (gfx950): the partner row / half arrives through the vector ALU
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
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

enum { K_PFC_B, K_POUT_W, K_DPOOL_B, K_DNORM, K_DFC_B, K_DOUT_NORM, K_DVAL_B, K_DPOOL_NORM, K_N };

__device__ __forceinline__ void load_tokens(V8 (&v)[6], const uint16_t *xs, int sub, int vec)
{
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int t = sub + 8 * k;
        V8 z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0;
        v[k] = t < CELLS ? *reinterpret_cast<const V8 *>(xs + t * C + vec * 8) : z;
    }
}

// dynamic LDS layout (bytes)
constexpr int L_A = 0;                                   // V8 [26*64]: A fragments
constexpr int L_C = L_A + 26 * 64 * 16;                  // float [K_N][64]: per-channel constants
constexpr int L_PN = L_C + K_N * C * 4;                  // per wave: V8 [42*8] normalised tokens (bf16)
constexpr int L_VEC = L_PN + WPB * CELLS * C * 2;        // per wave: u16 [16*VS] B operand, row n = column n of V^T
constexpr int L_PART = L_VEC + WPB * 16 * VS * 2;        // per wave: float [8][64] channel sums of the 8 token slots
constexpr int L_SCORE = L_PART + WPB * 8 * C * 4;        // per wave: float [48] row-gate scores, then weights
constexpr int L_MEAN = L_SCORE + WPB * 48 * 4;           // per wave: float [2][64] token means of the pair
constexpr int L_TOTAL = L_MEAN + WPB * 2 * C * 4;

__global__ void __launch_bounds__(64 * WPB) k_heads(const uint16_t *tok, az_nn_heads_weights w, const uint8_t *mask,
                                                    float *probs, float *wdl, float *moves_left, int64_t B, float eps,
                                                    const int32_t *scatter, const int64_t *batch_dev)
{
    const int64_t rows_total = B;                 // rows of mask / outputs: a compact list may name any of them
    if (batch_dev != nullptr && *batch_dev < B) B = *batch_dev;
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int sub = lane >> 3, vec = lane & 7;
    // A fragments (fragment f, lane l -> 16 bytes at f*64+l): policy fc 0-7, pool_fc 8-15, fc 16-23
    // as [m tile][k step]; 24-25 = rows {value_out 0-2, aux_out} x k step
    V8 *s_a = reinterpret_cast<V8 *>(smem + L_A);
    float (*s_c)[C] = reinterpret_cast<float (*)[C]>(smem + L_C);
    V8 *s_pn = reinterpret_cast<V8 *>(smem + L_PN + wave * CELLS * C * 2);
    uint16_t *s_vec = reinterpret_cast<uint16_t *>(smem + L_VEC + wave * 16 * VS * 2);
    float *s_part = reinterpret_cast<float *>(smem + L_PART + wave * 8 * C * 4);
    float *s_score = reinterpret_cast<float *>(smem + L_SCORE + wave * 48 * 4);
    float *s_mean = reinterpret_cast<float *>(smem + L_MEAN + wave * 2 * C * 4);

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
        s_c[K_DPOOL_NORM][i] = bf1(static_cast<const uint16_t *>(w.d_pool_norm) + i);
    }
    for (int i = lane; i < 16 * VS; i += 64) s_vec[i] = 0;
    __syncthreads();

    // policy-norm weight and norm x row-gate weight of this lane's 8 channels
    f32x2 nw2[4], ngw2[4];
    {
        const V8 a = *reinterpret_cast<const V8 *>(static_cast<const uint16_t *>(w.p_norm) + vec * 8);
        const V8 g = *reinterpret_cast<const V8 *>(static_cast<const uint16_t *>(w.p_gate_w) + vec * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) { nw2[q] = unpack2(a.w[q]); ngw2[q] = nw2[q] * unpack2(g.w[q]); }
    }
    const float dpool_norm = s_c[K_DPOOL_NORM][lane];
    auto afrag = [&](int f) { return as_bf16x8(s_a[f * 64 + lane]); };
    auto bfrag = [&](int ks) { return as_bf16x8(*reinterpret_cast<const V8 *>(&s_vec[l15 * VS + 32 * ks + 8 * l4])); };
    auto cvec2 = [&](int which, int m, int h) { return *reinterpret_cast<const f32x2 *>(&s_c[which][16 * m + 4 * l4 + 2 * h]); };
    const bool dual = (l15 & 7) == 7;          // this lane's accumulator column is a value-head vector
    const int half = l15 >> 3;                 // which sample of the pair the column belongs to
    // a dual lane's 16 accumulator values (channel 16m+4*l4+reg) -> its own B-operand row
    auto put_dual = [&](const f32x2 (&v)[4][2]) {
        if (dual) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                uint32_t *p = reinterpret_cast<uint32_t *>(&s_vec[l15 * VS + 16 * m + 4 * l4]);
                p[0] = pack2(v[m][0].x, v[m][0].y);
                p[1] = pack2(v[m][1].x, v[m][1].y);
            }
        }
    };

    const int64_t npairs = (B + 1) / 2;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * WPB;
    int64_t pr = static_cast<int64_t>(blockIdx.x) * WPB + wave;
    V8 cur[6];
    if (pr < npairs) load_tokens(cur, tok + (2 * pr) * (CELLS * C), sub, vec);
    for (; pr < npairs; pr += stride) {
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            // the sample after this one: second of the pair, or the first of this wave's next pair
            int64_t nb = h == 0 ? 2 * pr + 1 : 2 * (pr + stride);
            if (nb >= B) nb = 2 * pr;                                  // loaded and never used
            V8 nxt[6];
            load_tokens(nxt, tok + nb * (CELLS * C), sub, vec);

            // ---- pass over the tokens: RMS statistics, row-gate score, channel sums, normalised tokens
            f32x2 msum[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int t = sub + 8 * k;
                f32x2 f[4], ss2 = {0.f, 0.f}, sc2 = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f[q] = unpack2(cur[k].w[q]);
                    ss2 = __builtin_elementwise_fma(f[q], f[q], ss2);
                    sc2 = __builtin_elementwise_fma(f[q], ngw2[q], sc2);
                    msum[q] += f[q];
                }
                const float ss = sum8(ss2.x + ss2.y), sc = sum8(sc2.x + sc2.y);
                const float r = rsqrtf(ss * (1.0f / C) + eps);
                if (t < CELLS) {
                    V8 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x2 pn = f[q] * f32x2{r, r} * nw2[q];
                        o.w[q] = pack2(pn.x, pn.y);
                    }
                    s_pn[t * 8 + vec] = o;
                    if (vec == 0) s_score[t] = sc * r + w.p_gate_b;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x2 *>(&s_part[sub * C + vec * 8 + 2 * q]) = msum[q];
            wave_lds_sync();

            // ---- token mean (bf16, like the reference's mean of a bf16 tensor), lane = channel:
            // kept for the residual, and its pool_norm goes to this sample's value column
            {
                float m = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) m += s_part[j * C + lane];
                const f32x2 g0 = rbf2(f32x2{m * (1.0f / CELLS), 0.0f});
                const float r = rsqrtf(wave_sum(g0.x * g0.x) * (1.0f / C) + eps);
                s_mean[h * C + lane] = g0.x;
                s_vec[(8 * h + 7) * VS + lane] = to_bf16(g0.x * r * dpool_norm);
            }
            // ---- softmax over the 6 rows of each column: lane t owns token t's pooling weight
            if (lane < CELLS) {
                const int c = lane % COLS;
                float sc[ROWS], mx = -INFINITY, den = 0.0f;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) { sc[r] = s_score[r * COLS + c]; mx = fmaxf(mx, sc[r]); }
#pragma unroll
                for (int r = 0; r < ROWS; ++r) den += fast_exp(sc[r] - mx);
                const float wt = fast_exp(s_score[lane] - mx) * __builtin_amdgcn_rcpf(den);
                wave_lds_sync();                                       // every lane has read the scores
                s_score[lane] = rbf2(f32x2{wt, 0.0f}).x;
            } else {
                wave_lds_sync();
            }
            wave_lds_sync();
            // ---- weighted column sums: lane = (channel pair, half of the columns), packed f32
            {
                const uint32_t *pn2 = reinterpret_cast<const uint32_t *>(s_pn);
                const int cp = lane & 31, c0 = (lane >> 5) * 4;       // columns c0 .. c0+3 (the 8th does not exist)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int c = c0 + cc;
                    if (c < COLS) {
                        f32x2 acc = {0.0f, 0.0f};
#pragma unroll
                        for (int r = 0; r < ROWS; ++r) {
                            const float wt = s_score[r * COLS + c];
                            acc = __builtin_elementwise_fma(f32x2{wt, wt}, unpack2(pn2[(r * COLS + c) * (C / 2) + cp]), acc);
                        }
                        *reinterpret_cast<uint32_t *>(&s_vec[(8 * h + c) * VS + 2 * cp]) = pack2(acc.x, acc.y);
                    }
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < 6; ++k) cur[k] = nxt[k];
        }

        // ======== both samples of the pair: columns 0-6 | 7 and 8-14 | 15 of the B operand ========
        const int64_t bc = 2 * pr + half;             // the sample this lane's column belongs to
        // compact batch: sample bc stands for row scatter[bc] of the mask and of the outputs; an index
        // outside the rows (a list longer than what was written) is dropped, never dereferenced
        const int64_t b = (bc < B && scatter != nullptr) ? scatter[bc] : bc;
        const bool real = bc < B && b >= 0 && b < rows_total;
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
        // policy: logit[c] = out . silu(fc(col_c) + b), masked softmax over the 7 columns of a sample
        {
            f32x2 part2 = {0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x2 x = rbf2(f32x2{ap[m][2 * hh], ap[m][2 * hh + 1]} + cvec2(K_PFC_B, m, hh));
                    part2 = __builtin_elementwise_fma(rbf2(silu2(x)), cvec2(K_POUT_W, m, hh), part2);
                }
            float logit = col_sum(part2.x + part2.y) + w.p_out_b;
            const bool live = !dual;
            if (live && real && mask != nullptr && mask[b * COLS + (l15 & 7)] == 0) logit = -1e9f;
            if (!live) logit = -INFINITY;
            const float mx = max8(logit);
            const float e = live ? fast_exp(logit - mx) : 0.0f;
            const float den = sum8(e);
            if (live && real && l4 == 0) probs[b * COLS + (l15 & 7)] = e * __builtin_amdgcn_rcpf(den);
        }
        // value head, stage 1 (dual columns): g = mean + silu(pool_fc(pool_norm(mean)) + b); n2 = norm(g)
        f32x2 g[4][2];
        {
            f32x2 ss2 = {0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x2 mean = *reinterpret_cast<const f32x2 *>(&s_mean[half * C + 16 * m + 4 * l4 + 2 * hh]);
                    const f32x2 x = rbf2(f32x2{ad[m][2 * hh], ad[m][2 * hh + 1]} + cvec2(K_DPOOL_B, m, hh));
                    g[m][hh] = rbf2(mean + rbf2(silu2(x)));
                    ss2 = __builtin_elementwise_fma(g[m][hh], g[m][hh], ss2);
                }
            const float rn = rsqrtf(col_sum(ss2.x + ss2.y) * (1.0f / C) + eps);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) g[m][hh] = g[m][hh] * f32x2{rn, rn} * cvec2(K_DNORM, m, hh);
        }
        wave_lds_sync();
        put_dual(g);
        wave_lds_sync();
        // stage 2: h = out_norm(silu(fc(n2) + b))
        {
            const bf16x8 b0 = bfrag(0), b1 = bfrag(1);
            f32x2 ss2 = {0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 acc = MFMA32(afrag(16 + 2 * m), b0, zero);
                acc = MFMA32(afrag(16 + 2 * m + 1), b1, acc);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x2 x = rbf2(f32x2{acc[2 * hh], acc[2 * hh + 1]} + cvec2(K_DFC_B, m, hh));
                    g[m][hh] = rbf2(silu2(x));
                    ss2 = __builtin_elementwise_fma(g[m][hh], g[m][hh], ss2);
                }
            }
            const float rn = rsqrtf(col_sum(ss2.x + ss2.y) * (1.0f / C) + eps);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) g[m][hh] = g[m][hh] * f32x2{rn, rn} * cvec2(K_DOUT_NORM, m, hh);
        }
        wave_lds_sync();
        put_dual(g);
        wave_lds_sync();
        // stage 3: rows 0-2 = value logits, row 3 = moves-left logit, in the dual lanes with l4 == 0
        {
            f32x4 acc = MFMA32(afrag(24), bfrag(0), zero);
            acc = MFMA32(afrag(25), bfrag(1), acc);
            if (dual && l4 == 0 && real) {
                const f32x2 v01 = rbf2(f32x2{acc[0] + s_c[K_DVAL_B][0], acc[1] + s_c[K_DVAL_B][1]});
                const float v2 = rbf2(f32x2{acc[2] + s_c[K_DVAL_B][2], 0.0f}).x;
                const float mx = fmaxf(v01.x, fmaxf(v01.y, v2));
                const float e0 = fast_exp(v01.x - mx), e1 = fast_exp(v01.y - mx), e2 = fast_exp(v2 - mx);
                const float inv = 1.0f / (e0 + e1 + e2);
                wdl[b * 3 + 0] = e0 * inv;
                wdl[b * 3 + 1] = e1 * inv;
                wdl[b * 3 + 2] = e2 * inv;
                moves_left[b] = w.aux_scale / (1.0f + fast_exp(-(acc[3] + w.d_aux_b)));
            }
        }
        wave_lds_sync();
    }
}

}  // namespace

extern "C" int az_nn_heads(const void *tokens, const az_nn_heads_weights *w, const uint8_t *mask, float *probs,
                           float *wdl, float *moves_left, int64_t batch, float eps, const int32_t *scatter,
                           const int64_t *batch_dev, void *stream)
{
    if (batch <= 0 || w == nullptr || tokens == nullptr || probs == nullptr || wdl == nullptr || moves_left == nullptr) return 1;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_heads), hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL) !=
            hipSuccess)
            return 2;
        attr_set = true;
    }
    // 70 KB of LDS per workgroup: two workgroups (8 wavefronts) per CU, each walking its sample pairs
    const int64_t want = ((batch + 1) / 2 + WPB - 1) / WPB;
    const unsigned grid = static_cast<unsigned>(want < 512 ? want : 512);
    hipLaunchKernelGGL(k_heads, dim3(grid), dim3(64 * WPB), L_TOTAL, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(tokens), *w, mask, probs, wdl, moves_left, batch, eps, scatter, batch_dev);
    return 0;
}
