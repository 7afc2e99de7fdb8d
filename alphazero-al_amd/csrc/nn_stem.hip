// nn_stem.hip - the stem of the evaluator (embedding + 3x3 convolution 32 -> 64 + SiLU; Network.py:168-170,
// 226-239) computed from what it is a function of.
//
// The reference builds tokens  t[cell] = own[cell] * e_own + opp[cell] * e_opp + pos[cell]  with own / opp in {0, 1}
// and convolves them; the convolution is linear, so
//
//   conv(t)[o, cell] + bias[o] = P[cell][o] + sum over the taps (dy, dx) whose neighbour n lies on the board of
//                                own[n] * A[tap][o] + opp[n] * Bm[tap][o]
//   A[tap][o] = sum_c W[o][c][tap] e_own[c],  Bm likewise with e_opp,  P = conv(pos map) + bias
//
// (A, Bm, P: fp32 on the host, once per weight snapshot - fast_net.fold_stem).  That is a GEMM with K = 18 (9 taps
// x 2 planes, padded to one k step of 32) instead of K = 288, whose token operand is a 0/1 matrix that a lane builds
// from the two BITBOARDS of the leaf with a shift and a mask - no token tensor, no padded image in LDS, no barrier:
// a wavefront owns a sample from the bitboards to the stored activation, and a 16 x 16 output tile takes two MFMAs
// (the table split into a bf16 high and a bf16 low part, so that the sum is fp32-accurate - the 0/1 operand is
// exact) where k_conv_block<32, EMBED> issued nine.  What is left is the SiLU and the 5.4 KB store per sample.
//   A operand  = table fragments (64 x 32, hi and lo: 32 registers, loaded once per wavefront).  Row r of channel tile i
//                is channel 32 (i / 2) + 8 (r / 4) + 4 (i % 2) + r % 4 (the host packs them so): the four rows a lane
//                gets from tiles 2 m and 2 m + 1 are EIGHT consecutive channels - one 16-byte store per token;
//   B operand  = lane (g, tl): token tl of the tile, k = 8 g .. 8 g + 7 = taps 4 g .. 4 g + 3 x {own, opp};
//   C operand  = P[token][8 channels] from LDS (rows padded to 272 B: the 16 tokens of a read on distinct banks);
//   D layout   = 2 x 4 channels of one token per lane -> SiLU -> bf16 -> global.
// The next sample's bitboards are requested before the current one is computed (two dependent loads: the compact
// row index, then the position); the accumulators live in VGPRs (-amdgpu-mfma-vgpr-form, build.py: in AGPRs every
// element costs a v_accvgpr_read before the SiLU can touch it).
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "az_nn.h"

namespace {

constexpr int CELLS = 42, COLS = 7, COUT = 64;
constexpr int PROWS = 48, PROW = 68;           // P in LDS: 48 token rows (42 used, the rest zero) of 68 floats (64 used)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// v * sigmoid(v) on two elements: packed multiplies and add around the hardware exp2 / reciprocal
__device__ __forceinline__ f32x2 silu2(f32x2 v)
{
    const f32x2 t = v * f32x2{-1.4426950408889634f, -1.4426950408889634f};
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + f32x2{1.0f, 1.0f};
    return v * f32x2{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
}

struct StemIn {
    const float    *features;          // (rows, 3, 6, 7) relative planes, or nullptr: positions
    const uint64_t *bb_p1, *bb_p2;     // bit = 7 * column + height (Connect4.h:15-29)
    const int32_t  *turn, *sym;        // side to move, symmetry id (1 = columns mirrored)
    const int32_t  *gather;            // compact sample b shows row gather[b] (NULL: b)
};

__global__ void __launch_bounds__(256) k_stem(StemIn in, const uint16_t *wfrag, const float *pmap, uint16_t *y, int64_t B,
                                              const int64_t *batch_dev)
{
    const int64_t rows_total = B;
    if (batch_dev != nullptr && *batch_dev < B) B = *batch_dev;      // compact batch whose size only the device knows
    __shared__ __align__(16) float s_p[PROWS * PROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, tl = lane & 15;
    for (int i = tid; i < PROWS * PROW; i += 256) s_p[i] = pmap[i];
    __syncthreads();

    bf16x8 whi[4], wlo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        whi[i] = *reinterpret_cast<const bf16x8 *>(wfrag + ((0 * 4 + i) * 64 + lane) * 8);
        wlo[i] = *reinterpret_cast<const bf16x8 *>(wfrag + ((1 * 4 + i) * 64 + lane) * 8);
    }
    // this lane's four taps: where each one's neighbour sits in the 3x3 window word (bit 7 dy + dx); taps 9..15 do not
    // exist: bit 31 of the window word is always zero
    int sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tp = g * 4 + j;
        sh[j] = tp < 9 ? (tp / 3) * 7 + tp % 3 : 31;
    }
    // this lane's token in each of the three token tiles: which window bits exist for it (column 0 has no left
    // neighbours, column 6 no right ones - without the mask they would alias the neighbouring row's end)
    uint32_t cmask[3];
    int tok[3];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
        const int token = tt * 16 + tl;
        tok[tt] = token;
        const int c = token % COLS;
        uint32_t m = 0x1C387u;                                      // bits 0-2, 7-9, 14-16
        if (c == 0) m &= ~0x4081u;                                  // bits 0, 7, 14
        if (c == COLS - 1) m &= ~0x10204u;                          // bits 2, 9, 16
        cmask[tt] = token < CELLS ? m : 0u;
    }
    // the cell this lane tests when the board masks are built (cell order: 7 * row + column, row 0 on top)
    const int my_r = lane / COLS, my_c = lane % COLS;

    // what the board masks of sample b are built from: this lane's two plane values (features) or the two bitboards
    // and the mirror flag (positions); requested one sample ahead
    struct Raw { float f_own, f_opp; uint64_t own, opp; bool mir; };
    auto request = [&](int64_t b) {
        Raw r{0.0f, 0.0f, 0ull, 0ull, false};
        if (b >= B) return r;
        int64_t row = in.gather != nullptr ? in.gather[b] : b;
        if (row < 0 || row >= rows_total) row = 0;                  // never dereference an index outside the rows
        if (in.features != nullptr) {
            const float *fs = in.features + row * (3 * CELLS);
            if (lane < CELLS) { r.f_own = fs[lane]; r.f_opp = fs[CELLS + lane]; }
        } else {
            // the planes MCTS_cpp.py:15-20 builds from the (symmetrised) grid, straight from the bitboards
            const bool p1 = in.turn[row] > 0;
            r.mir = in.sym[row] != 0;
            r.own = p1 ? in.bb_p1[row] : in.bb_p2[row];
            r.opp = p1 ? in.bb_p2[row] : in.bb_p1[row];
        }
        return r;
    };
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 4;
    int64_t b = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    Raw cur = request(b);
    for (; b < B; b += stride) {
        const Raw nxt = request(b + stride);
        bool own_here, opp_here;
        if (in.features != nullptr) {
            own_here = cur.f_own != 0.0f; opp_here = cur.f_opp != 0.0f;
        } else {
            const int bit = (cur.mir ? COLS - 1 - my_c : my_c) * 7 + (5 - my_r);
            own_here = lane < CELLS && ((cur.own >> bit) & 1ull);
            opp_here = lane < CELLS && ((cur.opp >> bit) & 1ull);
        }
        // boards in cell order, moved up by 8 so that the window of cell n is bits n .. n + 16 of the word
        const uint64_t om8 = __ballot(own_here) << 8, pm8 = __ballot(opp_here) << 8;

        bf16x8 bq[3];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const uint32_t wo = static_cast<uint32_t>(om8 >> tok[tt]) & cmask[tt];
            const uint32_t wp = static_cast<uint32_t>(pm8 >> tok[tt]) & cmask[tt];
            uint32_t d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)                             // bf16 1.0 = 0x3F80: own in the low half, opp in the high half
                d[j] = (((wo >> sh[j]) & 1u) | (((wp >> sh[j]) & 1u) << 16)) * 0x3F80u;
            bq[tt] = __builtin_bit_cast(bf16x8, uint4{d[0], d[1], d[2], d[3]});
        }
        uint16_t *ys = y + b * (CELLS * COUT);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int ch = m * 32 + g * 8;                      // this lane's eight channels of tiles 2 m and 2 m + 1
                f32x4 a0 = *reinterpret_cast<const f32x4 *>(&s_p[tok[tt] * PROW + ch]);
                f32x4 a1 = *reinterpret_cast<const f32x4 *>(&s_p[tok[tt] * PROW + ch + 4]);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[2 * m], bq[tt], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[2 * m + 1], bq[tt], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[2 * m], bq[tt], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[2 * m + 1], bq[tt], a1, 0, 0, 0);
                const f32x2 s0 = silu2(f32x2{a0[0], a0[1]}), s1 = silu2(f32x2{a0[2], a0[3]});
                const f32x2 s2 = silu2(f32x2{a1[0], a1[1]}), s3 = silu2(f32x2{a1[2], a1[3]});
                const uint4 o = make_uint4(pack2(s0.x, s0.y), pack2(s1.x, s1.y), pack2(s2.x, s2.y), pack2(s3.x, s3.y));
                if (tt < 2 || tok[tt] < CELLS) *reinterpret_cast<uint4 *>(ys + tok[tt] * COUT + ch) = o;
            }
        }
        cur = nxt;
    }
}

int launch(const StemIn &in, const void *wfrag, const float *pmap, void *y, int64_t B, const int64_t *batch_dev, hipStream_t s)
{
    const int64_t blocks = (B + 3) / 4;
    const unsigned grid = static_cast<unsigned>(blocks < 2048 ? blocks : 2048);      // eight workgroups per CU, grid-stride
    hipLaunchKernelGGL(k_stem, dim3(grid), dim3(256), 0, s, in, static_cast<const uint16_t *>(wfrag), pmap,
                       static_cast<uint16_t *>(y), B, batch_dev);
    return 0;
}

}  // namespace

extern "C" {

int az_nn_stem_folded(const float *features, const void *w_frag, const float *pmap, void *y, int64_t batch,
                      const int32_t *gather, const int64_t *batch_dev, void *stream)
{
    if (batch <= 0 || features == nullptr || w_frag == nullptr || pmap == nullptr || y == nullptr) return 1;
    StemIn in{features, nullptr, nullptr, nullptr, nullptr, gather};
    return launch(in, w_frag, pmap, y, batch, batch_dev, static_cast<hipStream_t>(stream));
}

int az_nn_stem_folded_positions(const az_nn_positions *positions, const void *w_frag, const float *pmap, void *y,
                                int64_t batch, const int32_t *gather, const int64_t *batch_dev, void *stream)
{
    if (batch <= 0 || positions == nullptr || !positions->bb_p1 || !positions->bb_p2 || !positions->turn || !positions->sym ||
        w_frag == nullptr || pmap == nullptr || y == nullptr)
        return 1;
    StemIn in{nullptr, positions->bb_p1, positions->bb_p2, positions->turn, positions->sym, gather};
    return launch(in, w_frag, pmap, y, batch, batch_dev, static_cast<hipStream_t>(stream));
}

}  // extern "C"
