// nn_othello_heads.hip - the thin ends of the Othello evaluator (Othello/Network.py:40-104, 201-211) as HIP
// kernels, so that the whole network can run as one native object (az_nn_model, kind OTHELLO_CNN) inside
// az_mcts_dev_search: the embedding lookup in front of the convolutions and, behind them, both heads.
//
//   k_oth_embed   positions -> (8, 8, 32) bf16 NHWC tokens: a cell shows its orbit's position vector plus
//                 exactly one of {own stone, opponent stone, empty + legal, empty + illegal}: one row of a
//                 (64 cells x 4 kinds, 32) table built on the host (fast_othello.py); the board is the leaf
//                 under its symmetry id (board_sym.h), the legality plane comes from the action mask
//   k_oth_heads   policy: 64 square logits = 1x1 convolution of the policy stem's output, pass logit = Linear(
//                 RMSNorm(mean over squares)), softmax over the 65 (no masking: Network.py:58-62);
//                 value: 3x3 stride-2 convolution 8 -> 8 on the bottleneck's 8x8 map, BatchNorm, SiLU,
//                 Linear(72, 3), softmax; auxiliary: Linear(512, 512), RMSNorm, SiLU, Linear(512, 1), tanh,
//                 then the score utility atan(disc difference / score_scale) * 2/pi the search consumes.
//                 One 256-thread workgroup per 4 samples: the 512 x 512 weight (1 MB) is read once per
//                 workgroup, coalesced (stored transposed), against activations broadcast from LDS.
// fp32 arithmetic on bf16 activations (the reference's autocast rounds the 1x1 convolution's output to bf16
// as well; the difference is inside the evaluator's bf16 tolerance, tests).  Compact batches as everywhere:
// sample b of the launch is row gather[b] of the inputs / scatter[b] of the outputs, the count is on the device.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "az_nn.h"
#include "board_sym.h"

namespace {

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(static_cast<uint32_t>(v) << 16); }

struct alignas(16) V8 { uint32_t w[4]; };

__global__ void __launch_bounds__(256) k_oth_embed(az_nn_positions pos, const uint8_t *mask, const uint16_t *table,
                                                   uint16_t *tokens, int64_t B, const int32_t *gather, const int64_t *batch_dev)
{
    const int64_t rows_total = B;
    if (batch_dev != nullptr && *batch_dev < B) B = *batch_dev;
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t b = gid >> 8;                       // 64 cells x 4 chunks of 8 channels per sample
    if (b >= B) return;
    const int cell = static_cast<int>(gid >> 2) & 63, chunk = static_cast<int>(gid) & 3;
    int64_t row = gather != nullptr ? gather[b] : b;
    if (row < 0 || row >= rows_total) row = 0;
    const int sym = pos.sym[row];
    const bool p1 = pos.turn[row] > 0;
    const uint64_t a = az::othello_sym(pos.bb_p1[row], sym), c = az::othello_sym(pos.bb_p2[row], sym);
    const uint64_t own = p1 ? a : c, opp = p1 ? c : a;
    const int kind = ((own >> cell) & 1ull) ? 0 : (((opp >> cell) & 1ull) ? 1 : (mask[row * 65 + cell] ? 2 : 3));
    const V8 v = *reinterpret_cast<const V8 *>(table + (cell * 4 + kind) * 32 + chunk * 8);
    *reinterpret_cast<V8 *>(tokens + (b * 64 + cell) * 32 + chunk * 8) = v;
}

constexpr int SPW = 4;      // samples per workgroup of the heads kernel

__device__ __forceinline__ float block_sum(float v, float *red, int tid)     // sum over the 256 threads; red: 4 floats
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) k_oth_heads(const uint16_t *p2, const uint16_t *h8, az_nn_othello_heads_weights w,
                                                   float *probs, float *wdl, float *utility, int64_t B,
                                                   const int32_t *scatter, const int64_t *batch_dev)
{
    const int64_t rows_total = B;
    if (batch_dev != nullptr && *batch_dev < B) B = *batch_dev;
    __shared__ float s_h[SPW][512];          // the bottleneck's activations, channel-major: [c * 64 + cell]
    __shared__ float s_a[SPW][512];          // the auxiliary MLP's hidden layer
    __shared__ float s_logit[65];
    __shared__ float s_v[72];
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    const int64_t b0 = static_cast<int64_t>(blockIdx.x) * SPW;
    if (b0 >= B) return;
    const int ns = static_cast<int>(B - b0 < SPW ? B - b0 : SPW);

    // ---- bottleneck activations of the workgroup's samples into LDS (NHWC bf16 -> channel-major fp32)
    for (int i = tid; i < SPW * 512; i += 256) {
        const int s = i >> 9, j = i & 511, cell = j >> 3, c = j & 7;
        s_h[s][c * 64 + cell] = s < ns ? bf2f(h8[(b0 + s) * 512 + j]) : 0.0f;
    }
    __syncthreads();
    // ---- auxiliary head, first layer: thread t owns outputs t and t + 256 of all samples; weights transposed
    // [in][out], so a warp-row of 256 consecutive floats per input
    {
        float acc[SPW][2];
#pragma unroll
        for (int s = 0; s < SPW; ++s) { acc[s][0] = w.a_fc_b[tid]; acc[s][1] = w.a_fc_b[tid + 256]; }
        for (int i = 0; i < 512; ++i) {
            const float w0 = w.a_fc_wt[i * 512 + tid], w1 = w.a_fc_wt[i * 512 + tid + 256];
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                const float x = s_h[s][i];
                acc[s][0] = fmaf(x, w0, acc[s][0]);
                acc[s][1] = fmaf(x, w1, acc[s][1]);
            }
        }
#pragma unroll
        for (int s = 0; s < SPW; ++s) { s_a[s][tid] = acc[s][0]; s_a[s][tid + 256] = acc[s][1]; }
    }
    __syncthreads();

    for (int s = 0; s < ns; ++s) {
        const int64_t bc = b0 + s;
        int64_t row = scatter != nullptr ? scatter[bc] : bc;
        const bool ok = row >= 0 && row < rows_total;
        // ---- auxiliary head: RMSNorm(512), SiLU, Linear(512, 1), tanh; utility
        const float a0 = s_a[s][tid], a1 = s_a[s][tid + 256];
        const float ss = block_sum(a0 * a0 + a1 * a1, s_red, tid);
        const float r = rsqrtf(ss * (1.0f / 512.0f) + w.eps);
        const float n0 = a0 * r * w.a_norm_w[tid], n1 = a1 * r * w.a_norm_w[tid + 256];
        const float y0 = n0 / (1.0f + __expf(-n0)), y1 = n1 / (1.0f + __expf(-n1));
        const float dot = block_sum(y0 * w.a_out_w[tid] + y1 * w.a_out_w[tid + 256], s_red, tid);
        const float aux = tanhf(dot + w.a_out_b);
        // ---- value head: strided 3x3 convolution on the 8x8 map (3x3 outputs), BatchNorm, SiLU
        if (tid < 72) {
            const int co = tid / 9, posi = tid % 9, oy = posi / 3, ox = posi % 3;
            float acc = 0.0f;
            for (int ci = 0; ci < 8; ++ci)
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx)
                        acc = fmaf(s_h[s][ci * 64 + (2 * oy + ky) * 8 + 2 * ox + kx], w.v_conv_w[(ci * 9 + ky * 3 + kx) * 8 + co], acc);
            acc = acc * w.v_bn_s[co] + w.v_bn_b[co];
            s_v[tid] = acc / (1.0f + __expf(-acc));                  // flatten order of (8, 3, 3): co * 9 + position
        }
        // ---- policy head: thread t = channel t of the 256: mean over the 64 squares; 4 threads per square
        // for the 1x1 convolution
        const uint16_t *ps = p2 + bc * (64 * 256);
        float m = 0.0f;
        for (int cell = 0; cell < 64; ++cell) m += bf2f(ps[cell * 256 + tid]);
        m *= (1.0f / 64.0f);
        const float ms = block_sum(m * m, s_red, tid);
        const float pr = rsqrtf(ms * (1.0f / 256.0f) + w.eps);
        const float pass_logit = block_sum(m * pr * w.pass_norm_w[tid] * w.pass_fc_w[tid], s_red, tid) + w.pass_fc_b;
        {
            const int cell = tid >> 2, q = tid & 3;
            float acc = 0.0f;
            for (int c = q * 64; c < q * 64 + 64; ++c) acc = fmaf(bf2f(ps[cell * 256 + c]), bf2f(static_cast<const uint16_t *>(w.board_w)[c]), acc);
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if (q == 0) s_logit[cell] = acc + w.board_b;
        }
        if (tid == 0) s_logit[64] = pass_logit;
        __syncthreads();
        if (tid < 64) {                                              // one wavefront: softmax over 65, value softmax
            float l0 = s_logit[tid], l1 = tid == 0 ? s_logit[64] : -INFINITY;
            float mx = fmaxf(l0, l1);
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            const float e0 = __expf(l0 - mx), e1 = tid == 0 ? __expf(l1 - mx) : 0.0f;
            float den = e0 + e1;
            for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o, 64);
            if (ok) {
                probs[row * 65 + tid] = e0 / den;
                if (tid == 0) probs[row * 65 + 64] = e1 / den;
            }
            if (tid < 3) {
                float acc = w.v_fc_b[tid];
                for (int j = 0; j < 72; ++j) acc = fmaf(s_v[j], w.v_fc_w[tid * 72 + j], acc);
                const float v0 = __shfl(acc, 0, 64), v1 = __shfl(acc, 1, 64), v2 = __shfl(acc, 2, 64);
                const float vm = fmaxf(v0, fmaxf(v1, v2));
                const float d = __expf(v0 - vm) + __expf(v1 - vm) + __expf(v2 - vm);
                if (ok) wdl[row * 3 + tid] = __expf(acc - vm) / d;
            }
            if (tid == 0 && ok) utility[row] = atanf(aux * w.aux_to_score) * (2.0f / 3.14159265358979f);
        }
        __syncthreads();
    }
}

// The same heads with the auxiliary head's Linear(512, 512) on the matrix cores (round 3): a workgroup owns SIXTEEN
// samples, out^T (512 outputs x 16 samples) = W (512 x 512 bf16, rows in the bottleneck's NHWC order: a_fc_w16) . X^T;
// the B operand is the samples' bottleneck rows straight from HBM (16 fragments per lane, kept in registers), the A
// operand streams from L2 one output tile (16 k steps) ahead; wavefront w owns output tiles 8 w .. 8 w + 7.  1 MB of
// fp32 weights per 4 samples becomes 512 KB of bf16 per 16, 1024 fp32 FMAs per thread and sample become 32 MFMAs per
// wavefront and sample.  bf16 weights are what the reference's autocast multiplies with.  The rest of a sample's work
// (norms, value convolution, policy logits, softmaxes) is k_oth_heads' code.
constexpr int SPW16 = 16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) k_oth_heads16(const uint16_t *p2, const uint16_t *h8, az_nn_othello_heads_weights w,
                                                     float *probs, float *wdl, float *utility, int64_t B,
                                                     const int32_t *scatter, const int64_t *batch_dev)
{
    const int64_t rows_total = B;
    if (batch_dev != nullptr && *batch_dev < B) B = *batch_dev;
    __shared__ __align__(16) float s_a[SPW16][512 + 4];      // the auxiliary MLP's hidden layer (+ 4: rows on different banks)
    __shared__ float s_hw[4][512];           // per wavefront: one sample's bottleneck, channel-major fp32 [c * 64 + cell]
    __shared__ float s_vv[4][72];            // per wavefront: the value convolution's outputs
    __shared__ float s_vw[576], s_fw[216];   // value convolution / Linear(72, 3) weights
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t b0 = static_cast<int64_t>(blockIdx.x) * SPW16;
    if (b0 >= B) return;
    const int ns = static_cast<int>(B - b0 < SPW16 ? B - b0 : SPW16);
    for (int i = tid; i < 576; i += 256) s_vw[i] = w.v_conv_w[i];
    if (tid < 216) s_fw[tid] = w.v_fc_w[tid];
    // per-lane constants of the per-sample part: norm / output weights of the auxiliary head's elements 8 lane .. + 7,
    // the 1x1 convolution's weight and the pass branch's weights of channels 8 (lane % 32) .. + 7
    float anw[8], aow[8], bw[8], pnw[8], pfw[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        anw[e] = w.a_norm_w[8 * lane + e]; aow[e] = w.a_out_w[8 * lane + e];
        const int c = 8 * (lane & 31) + e;
        bw[e] = bf2f(static_cast<const uint16_t *>(w.board_w)[c]); pnw[e] = w.pass_norm_w[c]; pfw[e] = w.pass_fc_w[c];
    }
    auto wsum = [](float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; };
    auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    // ---- auxiliary head, first layer
    {
        const int smp = l15 < ns ? l15 : ns - 1;                     // columns past the batch repeat its last sample: never read
        const uint16_t *xs = h8 + (b0 + smp) * 512 + 8 * l4;
        bf16x8 xb[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) xb[ks] = *reinterpret_cast<const bf16x8 *>(xs + 32 * ks);
        const uint16_t *w16 = static_cast<const uint16_t *>(w.a_fc_w16);
        auto fetch = [&](bf16x8 (&a)[16], int mt) {
            const uint16_t *wr = w16 + static_cast<size_t>(16 * mt + l15) * 512 + 8 * l4;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) a[ks] = *reinterpret_cast<const bf16x8 *>(wr + 32 * ks);
        };
        bf16x8 a0[16], a1[16];
        fetch(a0, wave * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int mt = wave * 8 + i;
            bf16x8 (&cur)[16] = (i & 1) ? a1 : a0;
            bf16x8 (&nxt)[16] = (i & 1) ? a0 : a1;
            if (i + 1 < 8) fetch(nxt, mt + 1);
            f32x4 acc = *reinterpret_cast<const f32x4 *>(w.a_fc_b + 16 * mt + 4 * l4);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[ks], xb[ks], acc, 0, 0, 0);
            *reinterpret_cast<f32x4 *>(&s_a[l15][16 * mt + 4 * l4]) = acc;
        }
    }
    __syncthreads();

    // ---- everything else of a sample is ONE wavefront's work (wavefront w: samples w, w + 4, ...): no workgroup
    // barrier, reductions on the DPP network / shuffles only
    for (int s = wave; s < ns; s += 4) {
        const int64_t bc = b0 + s;
        int64_t row = scatter != nullptr ? scatter[bc] : bc;
        const bool ok = row >= 0 && row < rows_total;
        // -- auxiliary head: RMSNorm(512), SiLU, Linear(512, 1), tanh (lane: elements 8 lane .. 8 lane + 7)
        float aux;
        {
            const f32x4 h0 = *reinterpret_cast<const f32x4 *>(&s_a[s][8 * lane]), h1 = *reinterpret_cast<const f32x4 *>(&s_a[s][8 * lane + 4]);
            const float av[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            float ss = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = fmaf(av[e], av[e], ss);
            const float r = rsqrtf(wsum(ss) * (1.0f / 512.0f) + w.eps);
            float dot = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float n = av[e] * r * anw[e];
                dot = fmaf(n / (1.0f + __expf(-n)), aow[e], dot);
            }
            aux = tanhf(wsum(dot) + w.a_out_b);
        }
        // -- value head: this sample's bottleneck (NHWC bf16: lane = cell, 8 channels) -> channel-major fp32 in LDS;
        // strided 3x3 convolution (72 outputs: lane and lane + 64), BatchNorm, SiLU, Linear(72, 3), softmax
        float *sh = s_hw[wave], *sv = s_vv[wave];
        {
            const uint4 q = *reinterpret_cast<const uint4 *>(h8 + bc * 512 + 8 * lane);
            const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sh[(2 * e) * 64 + lane] = __uint_as_float(qw[e] << 16);
                sh[(2 * e + 1) * 64 + lane] = __uint_as_float(qw[e] & 0xffff0000u);
            }
        }
        wave_sync();
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int o = lane + 64 * rep;
            if (o < 72) {
                const int co = o / 9, posi = o % 9, oy = posi / 3, ox = posi % 3;
                float acc = 0.0f;
                for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        acc = fmaf(sh[ci * 64 + (2 * oy + k / 3) * 8 + 2 * ox + k % 3], s_vw[(ci * 9 + k) * 8 + co], acc);
                acc = acc * w.v_bn_s[co] + w.v_bn_b[co];
                sv[o] = acc / (1.0f + __expf(-acc));                 // flatten order of (8, 3, 3): co * 9 + position
            }
        }
        wave_sync();
        {
            float v3[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                float part = sv[lane] * s_fw[o * 72 + lane];
                if (lane < 8) part = fmaf(sv[64 + lane], s_fw[o * 72 + 64 + lane], part);
                v3[o] = wsum(part) + w.v_fc_b[o];
            }
            const float vm = fmaxf(v3[0], fmaxf(v3[1], v3[2]));
            const float d = __expf(v3[0] - vm) + __expf(v3[1] - vm) + __expf(v3[2] - vm);
            if (ok && lane < 3) wdl[row * 3 + lane] = __expf((lane == 0 ? v3[0] : lane == 1 ? v3[1] : v3[2]) - vm) / d;
        }
        // -- policy head: the (64 squares, 256 channels) map in 32 passes of 16 bytes per lane; lane l keeps the 8
        // channels of chunk l % 32 and sees squares 2 i + l / 32: channel sums (-> pass logit) and the partial dot with
        // the 1x1 convolution's weight, one per pass, which a halving exchange then sums over the 32 lanes of a half
        const uint16_t *ps = p2 + bc * (64 * 256);
        float msum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, part[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const uint4 q = *reinterpret_cast<const uint4 *>(ps + (i * 64 + lane) * 8);
            const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
            float d = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = __uint_as_float(qw[e] << 16), hi = __uint_as_float(qw[e] & 0xffff0000u);
                msum[2 * e] += lo; msum[2 * e + 1] += hi;
                d = fmaf(lo, bw[2 * e], d); d = fmaf(hi, bw[2 * e + 1], d);
            }
            part[i] = d;
            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);     // eight loads in flight, not thirty-two
        }
        // halving exchange inside each half of the wavefront: after the step with distance D a lane keeps the passes
        // whose bit (D) equals its own lane bit; five steps leave ONE pass per lane: pass p = lane % 32 - square 2 p + lane / 32
#define AZ_HALVE(D, N)                                                                                   \
        {                                                                                                    \
            const bool up = (lane & (D)) != 0;                                                               \
            _Pragma("unroll") for (int j = 0; j < (N) / 2; ++j) {                                            \
                const float keep = up ? part[j + (N) / 2] : part[j], give = up ? part[j] : part[j + (N) / 2]; \
                part[j] = keep + __shfl_xor(give, (D), 64);                                                  \
            }                                                                                                \
        }
        AZ_HALVE(16, 32) AZ_HALVE(8, 16) AZ_HALVE(4, 8) AZ_HALVE(2, 4) AZ_HALVE(1, 2)
#undef AZ_HALVE
        // the numbering after the five steps: a lane's remaining pass has bit (D) = its lane bit for every D, i.e. pass =
        // lane % 32 with the bit order the steps consumed: step D = 16 split on the TOP bit of the pass index (j vs j + 16)
        const int cell = 2 * (lane & 31) + (lane >> 5);
        const float logit = part[0] + w.board_b;
        // pass logit: mean over the 64 squares of this lane's 8 channels (both halves hold half of the squares)
        float ms = 0.0f, pl = 0.0f;
        {
            float m8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { m8[e] = (msum[e] + __shfl_xor(msum[e], 32, 64)) * (1.0f / 64.0f); ms = fmaf(m8[e], m8[e], ms); }
            ms = lane < 32 ? ms : 0.0f;
            const float pr = rsqrtf(wsum(ms) * (1.0f / 256.0f) + w.eps);
#pragma unroll
            for (int e = 0; e < 8; ++e) pl = fmaf(m8[e] * pr * pnw[e], pfw[e], pl);
            pl = wsum(lane < 32 ? pl : 0.0f) + w.pass_fc_b;
        }
        {
            float mx = fmaxf(logit, pl);
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            const float e0 = __expf(logit - mx), e1 = __expf(pl - mx);
            const float den = wsum(e0) + e1;
            if (ok) {
                probs[row * 65 + cell] = e0 / den;
                if (lane == 0) probs[row * 65 + 64] = e1 / den;
            }
        }
        if (lane == 0 && ok) utility[row] = atanf(aux * w.aux_to_score) * (2.0f / 3.14159265358979f);
    }
}

}  // namespace

extern "C" {

int az_nn_othello_embed(const az_nn_positions *positions, const uint8_t *mask, const void *embed_table, void *tokens,
                        int64_t batch, const int32_t *gather, const int64_t *batch_dev, void *stream)
{
    if (batch <= 0 || positions == nullptr || mask == nullptr || embed_table == nullptr || tokens == nullptr) return 1;
    if (!positions->bb_p1 || !positions->bb_p2 || !positions->turn || !positions->sym) return 1;
    const int64_t threads = batch * 256;
    hipLaunchKernelGGL(k_oth_embed, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *positions, mask, static_cast<const uint16_t *>(embed_table),
                       static_cast<uint16_t *>(tokens), batch, gather, batch_dev);
    return 0;
}

int az_nn_othello_heads(const void *policy_map, const void *bottleneck, const az_nn_othello_heads_weights *w, float *probs,
                        float *wdl, float *utility, int64_t batch, const int32_t *scatter, const int64_t *batch_dev, void *stream)
{
    if (batch <= 0 || policy_map == nullptr || bottleneck == nullptr || w == nullptr || probs == nullptr || wdl == nullptr ||
        utility == nullptr)
        return 1;
    if (w->a_fc_w16 != nullptr) {
        hipLaunchKernelGGL(k_oth_heads16, dim3(static_cast<unsigned>((batch + SPW16 - 1) / SPW16)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), static_cast<const uint16_t *>(policy_map),
                           static_cast<const uint16_t *>(bottleneck), *w, probs, wdl, utility, batch, scatter, batch_dev);
        return 0;
    }
    hipLaunchKernelGGL(k_oth_heads, dim3(static_cast<unsigned>((batch + SPW - 1) / SPW)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint16_t *>(policy_map),
                       static_cast<const uint16_t *>(bottleneck), *w, probs, wdl, utility, batch, scatter, batch_dev);
    return 0;
}

}  // extern "C"
