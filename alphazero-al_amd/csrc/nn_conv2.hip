// nn_conv2.hip - the evaluator's residual convolution block, second design: ONE wavefront per SIMD, 32x32x16 MFMAs,
// GroupNorm folded into the weights and the epilogue, every non-matrix instruction issued inside the MFMA stream.
//
//   y = x + silu( conv3x3( GroupNorm1(x) * gamma + beta ) + bias )                  (Network.py:27-48,166-170)
//
// Why a second design (nn_conv.hip stays for the stem and as the fallback): at two workgroups per CU the first
// kernel is bound by vector-instruction ISSUE - a 16x16x32 MFMA holds the SIMD's issue port for 8 of its 16
// cycles, the epilogue and the GroupNorm pass need the rest and more, and two wavefronts per SIMD only take
// turns (matrix pipe 44 % busy).  Here
//   * the MFMA is v_mfma_f32_32x32x16_bf16: the same 8 issue cycles buy 32 cycles of matrix work, so ~24 cycles
//     of other instructions hide behind every MFMA (MI355X_MICROARCH.md, cycle constants);
//   * GroupNorm never touches the activations: conv(W, pad(GN(x))) = rstd * (conv(W*gamma, pad(x)) - mean * T1) + T2
//     with T1[class][o] = sum of W*gamma over the taps that fall inside the board for a cell of that border class
//     (9 classes: row top/middle/bottom x column left/middle/right) and T2 = bias + the same sum of W*beta - the
//     RAW bf16 activations are the B operand, the sample's mean and rstd enter in the epilogue as one fma per
//     output.  W*gamma is rounded to bf16 once on the host (where the reference rounds the normalised
//     activations); T1 is summed from those rounded weights, so the mean cancels exactly;
//   * a workgroup is 4 wavefronts with 512 registers each: a wavefront keeps its 32 output channels' weights
//     (144 registers) and three accumulator tiles, and weaves into its MFMA stream the epilogue of the previous
//     token tile, the store of the previous 4-sample tile (whole 128-byte rows from an LDS stage) and the
//     statistics + per-sample epilogue table of the next one (staged HBM -> LDS by global_load_lds, straight
//     into the zero-padded image the MFMAs read);
//   * tokens sit on a 6 x 8 grid (8th column = the halo cell, a dummy token) in 8-KB images of 64 cells whose
//     16-byte chunks are XOR-swizzled with (cell >> 1) & 7: the 32 tokens of a B-fragment read hit 16 distinct
//     bank slots per 16-lane group of ds_read_b128, a step to the next row flips one address bit, and a step to
//     the next token tile (a multiple of 16 cells) is a constant in the read's offset field.
// Two barriers per 4-sample tile.  HBM traffic: x read once, y written once.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "az_nn.h"

namespace {

constexpr int CELLS = 42, TS = 4, CELLB = 128, PITCH = 64, CIN = 64, KSTEPS = 36;
constexpr int IMG = TS * PITCH * CELLB + 2 * CELLB;      // one tile's padded images (+ cells 64, 65 of the last sample)
constexpr int OUTC = 48;                                 // output stage: the token grid itself (dummy column included)
constexpr int OUTB = TS * OUTC * CELLB;
constexpr int NCLS = 9, TABB = NCLS * 64 * 4;
// per-sample epilogue table: class rows 272 bytes apart - lanes of one read that belong to different border classes
// then sit on different bank groups (256 apart they collide: up to six classes meet in a 16-lane group)
constexpr int UROW = 272, USMP = NCLS * UROW, UB = TS * USMP;
constexpr int L_IN = 0, L_OUT = L_IN + 2 * IMG, L_T1 = L_OUT + 2 * OUTB, L_T2 = L_T1 + TABB, L_U = L_T2 + TABB,
              L_RR = L_U + 2 * UB, L_TOTAL = L_RR + 64;
static_assert(L_TOTAL <= 160 * 1024, "LDS budget of one workgroup per CU");
static_assert(L_IN == 0, "the images sit at LDS address 0: every B-fragment read is register + immediate");
constexpr float L2E = 1.44269504f;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct alignas(16) V8 { uint32_t w[4]; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(pk_f32x2{a, b}, pk_bf16x2));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// 16 bytes per lane from global memory straight into LDS at (wave-uniform) lds_off + lane * 16 (see nn_conv.hip:
// inline assembly so that the compiler does not serialise later LDS reads behind the transfer)
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_off)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_off) : "memory");
}
// Pins a value where it stands in program order: the instruction that made it is issued before this point, its users
// after it.  The steps woven into the MFMA stream are single instructions meant for ONE slot each; left alone, the
// compiler pairs them into packed-f32 operations and lets whole dependency chains collapse into one slot.
template <class T>
__device__ __forceinline__ void pin(T &v)
{
    asm volatile("" : "+v"(v));
}
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ uint4 g_zero16[4];          // 64 zero bytes: what the halo lanes of a staging transfer load
__device__ uint4 g_dump[CELLS * 8];
__device__ unsigned long long g_stamp[256 * 4 * 4];   // per workgroup and wavefront: shader cycles, 100 MHz ticks, tiles of the tile loop (az_nn_conv2_stamps)    // where the rows of a sample past the batch go (a partial last tile): no branch in the MFMA stream

// DBG (timing experiments, AZ_NN_CONV2_DBG): bit 0 drops the epilogue steps from the MFMA stream, bit 1 the next
// tile's preparation and the previous tile's store - what is left runs at the speed of what was kept
template <int DBG>
__global__ void __launch_bounds__(256, 1) k_conv2(const uint16_t *x, const uint16_t *wf, const float *t1, const float *t2s,
                                                   uint16_t *y, int64_t B, float eps, const int64_t *batch_dev)
{
    if (batch_dev != nullptr && *batch_dev < B) B = *batch_dev;
    const int64_t ntiles = (B + TS - 1) / TS;
    if (static_cast<int64_t>(blockIdx.x) >= ntiles) return;
    extern __shared__ __align__(16) uint8_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mh = wave & 1, th = wave >> 1;
    const int n = lane & 31, h = lane >> 5;

    // ---- this wavefront's A fragments: 32 output channels x K = 9 taps x 64 channels, resident.  MFMA row rho of the
    // 32x32 tile stands for output channel 32 mh + 16 ((rho >> 2) & 1) + (rho & 3) + 4 (rho >> 3): with the C layout
    // (row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) lane half h then holds channels 32 mh + 16 h + reg, reg = 0..15 -
    // two whole 16-byte chunks of its token's output row.
    bf16x8 aw[KSTEPS];
    {
        const int ch = 32 * mh + 16 * ((n >> 2) & 1) + (n & 3) + 4 * (n >> 3);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            aw[s] = *reinterpret_cast<const bf16x8 *>(wf + static_cast<size_t>(ch) * (9 * CIN) + s * 16 + 8 * h);
        // pinned to the accumulator half of the register file (an MFMA reads its A operand from there directly):
        // the 256 architectural registers stay free for addresses and the epilogue, which vector instructions
        // cannot read from AGPRs without a copy
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) asm volatile("" : "+a"(aw[s]));
    }
    {
        V8 z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0;
        for (int i = tid; i < L_TOTAL / 16; i += 256) reinterpret_cast<V8 *>(smem)[i] = z;      // halos stay zero for good
        __syncthreads();
        for (int i = tid; i < 2 * NCLS * 64; i += 256)
            reinterpret_cast<float *>(smem + L_T1)[i] = i < NCLS * 64 ? t1[i] : t2s[i - NCLS * 64];
    }
    __syncthreads();

    // ---- per-lane geometry.  Token tile tt of this wavefront's pair of samples (2 th, 2 th + 1):
    //   tt 0: sample A, grid tokens 0..31;  tt 1: lanes n < 16 sample A tokens 32..47, n >= 16 sample B tokens 0..15;
    //   tt 2: sample B, tokens 16..47.  Grid token t sits at image cell 9 + t of its sample.
    auto token = [&](int tt, int &smp, int &t) {
        if (tt == 0) { smp = 2 * th; t = n; }
        else if (tt == 1) { smp = n < 16 ? 2 * th : 2 * th + 1; t = n < 16 ? 32 + n : n - 16; }
        else { smp = 2 * th + 1; t = 16 + n; }
    };
    // B-fragment addresses: [set 0: tt 0 (tt 2 = + 80 cells), set 1: tt 1][dx][row parity][k chunk], based one row up
    uint32_t ba[2][3][2][4];
#pragma unroll
    for (int set = 0; set < 2; ++set) {
        int smp, t;
        token(set, smp, t);
        const int c = smp * PITCH + 9 + t;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int q = c + dx - 1;
            const int key = (q >> 1) & 7;
#pragma unroll
            for (int dyp = 0; dyp < 2; ++dyp)
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
                    ba[set][dx][dyp][kc] = static_cast<uint32_t>((q - 8) * CELLB + ((((2 * kc + h) ^ key ^ (dyp ? 4 : 0))) << 4)) & 0xffffu;
        }
    }
    // epilogue addresses of the three token tiles: residual / centre cell (two chunks), output stage (two chunks),
    // this lane's row of the per-sample table U, the sample's scale
    uint32_t ca[3][2], oa[3][2], ua[3], ra[3];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
        int smp, t;
        token(tt, smp, t);
        const int c = smp * PITCH + 9 + t, l0 = 4 * mh + 2 * h;
        const int r = t >> 3, cc = t & 7;
        const int cls = 3 * (r == 0 ? 0 : (r == 5 ? 2 : 1)) + (cc == 0 ? 0 : (cc >= 6 ? 2 : 1));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            ca[tt][k] = static_cast<uint32_t>(c * CELLB + (((l0 + k) ^ ((c >> 1) & 7)) << 4)) & 0xffffu;
            oa[tt][k] = static_cast<uint32_t>(L_OUT + (smp * OUTC + t) * CELLB + (((l0 + k) ^ (t & 7)) << 4)) & 0x3ffffu;
        }
        ua[tt] = static_cast<uint32_t>(L_U + smp * USMP + cls * UROW + (32 * mh + 16 * h) * 4) & 0x3ffffu;
        ra[tt] = static_cast<uint32_t>(L_RR + smp * 4) & 0x3ffffu;
    }
    // staging: an instruction fills one image row (8 cells: the left halo + 7 board cells) of this wavefront's
    // sample; lane = (cell in row, slot); slot j of cell p holds channel chunk j ^ key(p)
    const int st_ci = lane >> 3, st_j = lane & 7;
    uint32_t st_off[2];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {            // rp: parity of the board row r (image row r + 1)
        const int key = (4 * (rp + 1) + (st_ci >> 1)) & 7;
        st_off[rp] = static_cast<uint32_t>((st_ci - 1) * CELLB + ((st_j ^ key) << 4));
    }
    // statistics / output store: chunk i = lane + 64 j of a sample's 336, j = 0..5
    uint32_t sa[6], pl[6], pg[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int i0 = lane + 64 * j;
        const bool ok = i0 < CELLS * 8;
        // the sixth round has 16 chunks left: its other lanes store the chunk they stored in the fifth again
        // (same bytes to the same place) - no predicate, no branch in the MFMA stream
        const int i = ok ? i0 : i0 - 64, cell = i >> 3, s = i & 7;
        const int r = cell / 7, c = cell - 7 * r, t = 8 * r + c;
        sa[j] = static_cast<uint32_t>(ok ? (wave * PITCH + (r + 1) * 8 + c + 1) * CELLB + s * 16 : wave * PITCH * CELLB) & 0xffffu;
        pl[j] = static_cast<uint32_t>(L_OUT + (wave * OUTC + t) * CELLB + s * 16) & 0x3ffffu;
        pg[j] = static_cast<uint32_t>(cell * CELLB + ((s ^ (t & 7)) << 4));
    }

    f32x16 acc0, acc1, acc2;
    {
        const f32x16 z = {};
        acc0 = acc1 = acc2 = z;
    }
    // epilogue state (one token tile at a time)
    // (two sets of loaded operands, alternating from token tile to token tile: the set of the next epilogue is
    // read from LDS while the current one is still in use)
    f32x4 eug[2];                 // this lane's table entries, one group of four outputs at a time (+ the next group's)
    u32x4 err[2][2];
    float ers[2] = {0.0f, 0.0f}, ex[4] = {0.f, 0.f, 0.f, 0.f}, ee[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const f32x4 zf = {0.f, 0.f, 0.f, 0.f};
        const u32x4 zu = {0, 0, 0, 0};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            err[a][0] = err[a][1] = zu;
            eug[a] = zf;
        }
    }
    uint32_t eo[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float dmy[5] = {1.f, 2.f, 3.f, 4.f, 5.f};
    // next-tile preparation state
    float st_sum = 0.0f, st_sq = 0.0f, st_rs = 0.0f, st_mm = 0.0f;

    // ------------------------------------------------------------------------------------------ pieces
    auto stage = [&](int64_t tile, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        const int64_t b = tile * TS + wave;
        if (tile >= ntiles || b >= B) return;
        const uint8_t *xs = reinterpret_cast<const uint8_t *>(x) + b * (CELLS * CIN * 2);
        const uint8_t *zero = reinterpret_cast<const uint8_t *>(g_zero16) + (lane & 3) * 16;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const uint8_t *src = st_ci ? xs + r * (7 * CELLB) + st_off[r & 1] : zero;
            glds16(src, static_cast<uint32_t>(L_IN + PAR * IMG + (wave * PITCH + (r + 1) * 8) * CELLB));
        }
    };
    // operands of the epilogue of token tile (PAR, TT), read ahead of it: the residual (centre cell, two chunks) from
    // the image, the sample's scale and this lane's 16 entries of the sample's table
    auto epi_load = [&](auto par_c, auto tt_c) {
        constexpr int PAR = decltype(par_c)::value, TT = decltype(tt_c)::value, SET = (PAR + TT) & 1;
        err[SET][0] = *reinterpret_cast<const u32x4 *>(&smem[ca[TT][0] + PAR * IMG]);
        err[SET][1] = *reinterpret_cast<const u32x4 *>(&smem[ca[TT][1] + PAR * IMG]);
        ers[SET] = *reinterpret_cast<const float *>(&smem[ra[TT] + PAR * 16]);
        eug[0] = *reinterpret_cast<const f32x4 *>(&smem[ua[TT] + PAR * UB]);
    };
    constexpr int EPI_STEPS = 4 * 26 + 3 + 2;       // four groups of 26, the next group's table read ahead of groups 1-3, two stores
    // one step of the epilogue of token tile (PAR, TT) on its accumulators `pa`
    auto epi_step = [&](auto par_c, auto tt_c, const f32x16 &pa, auto st_c) {
        constexpr int PAR = decltype(par_c)::value, TT = decltype(tt_c)::value, ST = decltype(st_c)::value, SET = (PAR + TT) & 1;
        // step list: [read group 1's entries] group 0 [read 2] group 1 [read 3] group 2, group 3, the two stores
        constexpr int GS = 27;                              // steps of groups 0-2 (with the read-ahead in front)
        if constexpr (ST < 3 * GS + 26) {
            constexpr int G = ST < 3 * GS ? ST / GS : 3, R = ST - G * GS;
            constexpr int W = G < 3 ? R - 1 : R, I = W & 3, E = 4 * G + I;
            if constexpr (W < 0) {
                eug[(G + 1) & 1] = *reinterpret_cast<const f32x4 *>(&smem[ua[TT] + PAR * UB + (G + 1) * 16]);
            } else {
                if constexpr (W == 0) { pin(eug[G & 1]); if constexpr (G == 0) { pin(err[SET][0]); pin(err[SET][1]); } }
                if constexpr (W < 4) {                                                                       // log2(e) * (rstd * acc + U)
                    if constexpr (DBG & 8) ex[I] = __builtin_fmaf(ee[I], ers[SET], eug[G & 1][I]);
                    else ex[I] = __builtin_fmaf(pa[E], ers[SET], eug[G & 1][I]);
                    pin(ex[I]);
                } else if constexpr (W < 8) {
                    if constexpr (DBG & 4) ee[I] = __builtin_fmaf(ex[I], L2E, L2E);
                    else ee[I] = __builtin_amdgcn_exp2f(-ex[I]);
                    pin(ee[I]);
                } else if constexpr (W < 12) { ee[I] = __builtin_fmaf(ee[I], L2E, L2E); pin(ee[I]); }        // log2(e) * (1 + e^-x)
                else if constexpr (W < 16) {
                    if constexpr (DBG & 4) ee[I] = __builtin_fmaf(ee[I], L2E, L2E);
                    else ee[I] = __builtin_amdgcn_rcpf(ee[I]);
                    pin(ee[I]);
                } else if constexpr (W < 20) { ex[I] = ex[I] * ee[I]; pin(ex[I]); }                          // x / (1 + e^-x)
                else if constexpr (W < 24) {
                    const uint32_t wd = err[SET][E >> 3][(E & 7) >> 1];
                    ex[I] += (E & 1) ? __uint_as_float(wd & 0xffff0000u) : __uint_as_float(wd << 16);
                    pin(ex[I]);
                } else if constexpr (W == 24) { eo[2 * G] = pack2(ex[0], ex[1]); pin(eo[2 * G]); }
                else { eo[2 * G + 1] = pack2(ex[2], ex[3]); pin(eo[2 * G + 1]); }
            }
        } else {
            constexpr int K = ST - (3 * GS + 26);
            const u32x4 o = {eo[4 * K], eo[4 * K + 1], eo[4 * K + 2], eo[4 * K + 3]};
            *reinterpret_cast<u32x4 *>(&smem[oa[TT][K] + PAR * OUTB]) = o;
        }
    };
    // store of one finished tile (its output stage, parity PAR) as whole rows: steps 0..11
    u32x4 pv = {0, 0, 0, 0};
    uint8_t *p3_dst = reinterpret_cast<uint8_t *>(g_dump);
    auto p3_begin = [&](int64_t tile) {               // wave-uniform: this wavefront's sample of `tile`, or the dump
        const int64_t b = tile * TS + wave;
        p3_dst = b < B ? reinterpret_cast<uint8_t *>(y) + b * (CELLS * 64 * 2) : reinterpret_cast<uint8_t *>(g_dump);
    };
    auto p3_step = [&](auto par_c, auto st_c) {
        constexpr int PAR = decltype(par_c)::value, ST = decltype(st_c)::value, J = ST >> 1;
        if constexpr ((ST & 1) == 0) pv = *reinterpret_cast<const u32x4 *>(&smem[pl[J] + PAR * OUTB]);
        else *reinterpret_cast<u32x4 *>(p3_dst + pg[J]) = pv;
    };
    constexpr int P3_STEPS = 12;
    // statistics of this wavefront's sample of the staged tile (image parity PAR) and its table
    //   U[class][o] = log2(e) * (T2[class][o] - rstd * mean * T1[class][o]),   scale = log2(e) * rstd
    u32x4 sv[3];
    f32x4 pt1[2], pt2[2];
    {
        const f32x4 zf = {0.f, 0.f, 0.f, 0.f};
        const u32x4 zu = {0, 0, 0, 0};
        sv[0] = sv[1] = sv[2] = zu;
        pt1[0] = pt1[1] = pt2[0] = pt2[1] = zf;
    }
    // (the sum of squares is NOT taken with v_dot2c_f32_bf16: its products carry bf16-like precision - measured: the mean
    // error of the block against fp32 doubled, from 2.2e-3 to 4.7e-3; the plain sum, a product with 1.0, is exact there)
    constexpr int DOTS = 12;                            // steps per 16-byte chunk: per dword one dot2 (sum) and two fma (squares)
    constexpr int PREP_STEPS = 1 + 3 + (6 * DOTS + 3) + 14 + 6 + 18 + 3;
    constexpr int ST_RED = 4 + 6 * DOTS + 3, ST_FIN = ST_RED + 14, ST_TAB = ST_FIN + 6;
    auto prep_step = [&](auto par_c, auto st_c, auto stores_c) {
        constexpr int PAR = decltype(par_c)::value, ST = decltype(st_c)::value;
        if constexpr (ST == 0) {
            if constexpr (decltype(stores_c)::value) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the staged rows have landed (own transfers)
            st_sum = 0.0f; st_sq = 0.0f;
        } else if constexpr (ST < 4) {                                       // the sample's 336 chunks: three reads in flight
            sv[ST - 1] = *reinterpret_cast<const u32x4 *>(&smem[sa[ST - 1] + PAR * IMG]);
        } else if constexpr (ST < ST_RED) {
            // rounds j = 0..5 of DOTS steps; behind rounds 0-2 the register is refilled with chunk j + 3
            constexpr int R = ST - 4, J = R < 3 * (DOTS + 1) ? R / (DOTS + 1) : 3 + (R - 3 * (DOTS + 1)) / DOTS;
            constexpr int W = R < 3 * (DOTS + 1) ? R % (DOTS + 1) : (R - 3 * (DOTS + 1)) % DOTS;
            if constexpr (W == DOTS) {
                sv[J] = *reinterpret_cast<const u32x4 *>(&smem[sa[J + 3] + PAR * IMG]);
            } else {
                if constexpr (W == 0) pin(sv[J % 3]);                       // one 16-byte read, not four narrowed ones at their uses
                const uint32_t wd = sv[J % 3][W / 3];
                if constexpr (W % 3 == 0) {
                    st_sum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, wd), __builtin_bit_cast(bf16x2, 0x3f803f80u), st_sum, false);
                    pin(st_sum);
                } else {
                    const float f = (W % 3 == 1) ? __uint_as_float(wd << 16) : __uint_as_float(wd & 0xffff0000u);
                    st_sq = __builtin_fmaf(f, f, st_sq);
                    pin(st_sq);
                }
            }
        } else if constexpr (ST < ST_FIN) {
            constexpr int W = ST - ST_RED, Q = W % 7;
            float &v = W < 7 ? st_sum : st_sq;
            if constexpr (Q == 0) v += dpp_mov<0xB1>(v);
            else if constexpr (Q == 1) v += dpp_mov<0x4E>(v);
            else if constexpr (Q == 2) v += dpp_mov<0x141>(v);
            else if constexpr (Q == 3) v += dpp_mov<0x140>(v);
            else if constexpr (Q == 4) v += dpp_mov<0x142, 0xa>(v);
            else if constexpr (Q == 5) v += dpp_mov<0x143, 0xc>(v);
            else v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
            pin(v);
        } else if constexpr (ST < ST_TAB) {
            constexpr int W = ST - ST_FIN;
            if constexpr (W == 0) st_sum *= (1.0f / (CELLS * CIN));                                   // mean
            else if constexpr (W == 1) st_sq = __builtin_fmaf(st_sq, 1.0f / (CELLS * CIN), -st_sum * st_sum);
            else if constexpr (W == 2) st_sq = __builtin_amdgcn_rsqf(fmaxf(st_sq, 0.0f) + eps);       // rstd
            else if constexpr (W == 3) st_rs = st_sq * L2E;
            else if constexpr (W == 4) st_mm = -st_rs * st_sum;
            else *reinterpret_cast<float *>(&smem[L_RR + PAR * 16 + wave * 4]) = st_rs;               // every lane, same word
        } else {
            // the sample's table: 144 float4 per table, three rounds (registers of round k: k & 1); in the third, lanes
            // >= 16 redo the item they did in the second (same result, no predicate)
            constexpr int W = ST - ST_TAB;
            auto item = [&](int k) { return k < 2 || lane < 16 ? lane + 64 * k : lane + 64 * (k - 1); };     // float4 number
            auto off = [&](int k) { return static_cast<uint32_t>(item(k)) * 16u; };
            auto uoff = [&](int k) { return static_cast<uint32_t>((item(k) >> 4) * UROW + (item(k) & 15) * 16); };
            auto rd = [&](auto k_c) {
                constexpr int K = decltype(k_c)::value;
                pt1[K & 1] = *reinterpret_cast<const f32x4 *>(&smem[L_T1 + off(K)]);
                pt2[K & 1] = *reinterpret_cast<const f32x4 *>(&smem[L_T2 + off(K)]);
            };
            auto fm = [&](auto k_c, auto i_c) {
                constexpr int K = decltype(k_c)::value, I = decltype(i_c)::value;
                if constexpr (I == 0) { pin(pt1[K & 1]); pin(pt2[K & 1]); }
                pt2[K & 1][I] = __builtin_fmaf(st_mm, pt1[K & 1][I], pt2[K & 1][I]);
                pin(pt2[K & 1]);
            };
            auto wr = [&](auto k_c) {
                constexpr int K = decltype(k_c)::value;
                *reinterpret_cast<f32x4 *>(&smem[L_U + PAR * UB + wave * USMP + uoff(K)]) = pt2[K & 1];
            };
            using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
            // order: read 0, read 1, fma 0 x4, write 0, read 2 (into the registers of 0), fma 1 x4, write 1, fma 2 x4, write 2
            if constexpr (W == 0) rd(K0{});
            else if constexpr (W == 1) rd(K1{});
            else if constexpr (W < 6) fm(K0{}, std::integral_constant<int, W - 2>{});
            else if constexpr (W == 6) wr(K0{});
            else if constexpr (W == 7) rd(K2{});
            else if constexpr (W < 12) fm(K1{}, std::integral_constant<int, W - 8>{});
            else if constexpr (W == 12) wr(K1{});
            else if constexpr (W < 17) fm(K2{}, std::integral_constant<int, W - 13>{});
            else if constexpr (W == 17) wr(K2{});
        }
    };
    static_assert(PREP_STEPS >= ST_TAB + 18, "step numbering of the preparation");
    // the 36 MFMAs of token tile TT on the image of parity PAR; fill(slot) is issued behind MFMA number `slot`
    auto mfma_tile = [&](auto par_c, auto tt_c, f32x16 &acc, auto &&fill) {
        constexpr int PAR = decltype(par_c)::value, TT = decltype(tt_c)::value;
        constexpr int SET = TT == 1 ? 1 : 0;
        constexpr int DISP = PAR * IMG + (TT == 2 ? 80 * CELLB : 0);
        auto fetch = [&](auto s_c) -> bf16x8 {
            constexpr int S = decltype(s_c)::value, TAP = S >> 2, KC = S & 3, DY = TAP / 3, DX = TAP % 3;
            return *reinterpret_cast<const bf16x8 *>(&smem[ba[SET][DX][DY != 1 ? 1 : 0][KC] + DISP + DY * 8 * CELLB]);
        };
        // B fragments are read RING - 1 MFMAs ahead: a ds_read_b128 issued beside three other wavefronts' reads takes
        // well over the 96 cycles that three MFMAs cover
        constexpr int RING = (DBG & 64) ? 4 : 8;
        bf16x8 bq[RING];
        f32x16 accx;
        static_for<0, RING - 1>([&](auto i_c) { bq[decltype(i_c)::value] = fetch(i_c); });
        __builtin_amdgcn_sched_barrier(0);
        // The MFMAs are inline assembly for one reason: the accumulators must live in ARCHITECTURAL registers.  With AGPR
        // accumulators (what the compiler picks once the weights sit in AGPRs) every v_accvgpr_read of the previous
        // tile's results waits for the MFMA in flight - measured: 45 us of a 125 us launch.  A operand: AGPRs, B: VGPRs.
        // Hazards (cdna_hip_programming.md 5.7): the chain takes D whole as C (no wait states); the first MFMA of a chain
        // has the constant 0 as C; the results are read by vector instructions dozens of instructions after the last
        // MFMA of the chain was issued; the B operand comes from an LDS read the compiler waits for.
        static_for<0, KSTEPS>([&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            if constexpr (S + RING - 1 < KSTEPS && !(DBG & 16)) bq[(S + RING - 1) % RING] = fetch(std::integral_constant<int, S + RING - 1>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DBG & 256) {            // experiment: accumulators in AGPRs
                if constexpr (S == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(aw[S]), "v"(bq[S % RING]));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(aw[S]), "v"(bq[S % RING]));
            } else if constexpr (DBG & 32) {      // experiment: two alternating accumulation chains (sum not formed)
                if constexpr (S == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(aw[S]), "v"(bq[S % RING]));
                else if constexpr (S == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(accx) : "a"(aw[S]), "v"(bq[S % RING]));
                else if constexpr (S & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accx) : "a"(aw[S]), "v"(bq[S % RING]));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(aw[S]), "v"(bq[S % RING]));
            } else {
                if constexpr (S == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(aw[S]), "v"(bq[S % RING]));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(aw[S]), "v"(bq[S % RING]));
            }
            __builtin_amdgcn_sched_barrier(0);
            fill(s_c);
            if constexpr (DBG & 128) {            // experiment: five independent plain vector instructions behind every MFMA
                asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3\n\tv_add_f32 %4, %4, %4"
                             : "+v"(dmy[0]), "+v"(dmy[1]), "+v"(dmy[2]), "+v"(dmy[3]), "+v"(dmy[4]));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (DBG & 32) { asm volatile("" :: "v"(accx)); }

    };
    // epilogue steps of slot S: 3 per slot, 4 in the first
    static_assert(EPI_STEPS <= 3 * KSTEPS + 1, "the epilogue fits behind the MFMAs of one token tile");
    auto epi_fill = [&](auto par_c, auto tt_c, const f32x16 &pa, auto s_c) {
        constexpr int S = decltype(s_c)::value, FIRST = S == 0 ? 0 : 3 * S + 1, COUNT = S == 0 ? 4 : 3;
        static_for<0, COUNT>([&](auto j_c) {
            constexpr int ST = FIRST + decltype(j_c)::value;
            if constexpr (ST < EPI_STEPS && !(DBG & 1)) epi_step(par_c, tt_c, pa, std::integral_constant<int, ST>{});
        });
    };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;

    // one 4-sample tile whose image has parity PAR; `prev` = the tile this workgroup did before (-1: none)
    // (`first_c`: the workgroup's first tile - nothing to finish or store from a tile before)
    auto body = [&](auto par_c, auto first_c, int64_t prev, int64_t next) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        using Q = std::integral_constant<int, PAR ^ 1>;
        lds_barrier();        // A: the tables of this tile are written, its image has landed; the other image is free again
        stage(next, Q{});
        if constexpr (!FIRST) p3_begin(prev);
        mfma_tile(par_c, T0{}, acc0, [&](auto s_c) { if constexpr (!FIRST) epi_fill(Q{}, T2{}, acc2, s_c); });
        lds_barrier();        // M: the output stage of `prev` is complete
        epi_load(par_c, T0{});
        mfma_tile(par_c, T1{}, acc1, [&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            epi_fill(par_c, T0{}, acc0, s_c);
            if constexpr (!FIRST && S >= 20 && S < 20 + P3_STEPS && !(DBG & 2)) p3_step(Q{}, std::integral_constant<int, S - 20>{});
        });
        epi_load(par_c, T1{});
        mfma_tile(par_c, T2{}, acc2, [&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            epi_fill(par_c, T1{}, acc1, s_c);
            // the next tile's statistics and table: four steps per slot from slot 2 on
            static_for<0, 4>([&](auto j_c) {
                constexpr int ST = 4 * (S - 2) + decltype(j_c)::value;
                if constexpr (S >= 2 && ST < PREP_STEPS && !(DBG & 2))
                    prep_step(Q{}, std::integral_constant<int, ST>{}, std::integral_constant<bool, !FIRST && !(DBG & 2)>{});
            });
        });
        static_assert(4 * (KSTEPS - 2) >= PREP_STEPS, "the preparation fits behind the MFMAs of one token tile");
        epi_load(par_c, T2{});        // before barrier A: the next staging overwrites this image
    };

    // ------------------------------------------------------------------------------------------ schedule
    const int64_t G = gridDim.x;
    int64_t tile = blockIdx.x, prev = -1;
    const unsigned long long t_c0 = __builtin_amdgcn_s_memtime(), t_r0 = __builtin_amdgcn_s_memrealtime();
    int64_t n_done = 0;
    stage(tile, P0{});
    static_for<0, PREP_STEPS>([&](auto st_c) { prep_step(P0{}, st_c, std::false_type{}); });
    int last_par = 0;
    body(P0{}, std::true_type{}, prev, tile + G);
    prev = tile; tile += G; ++n_done;
    while (tile < ntiles) {
        body(P1{}, std::false_type{}, prev, tile + G);
        prev = tile; tile += G; last_par = 1; ++n_done;
        if (tile >= ntiles) break;
        body(P0{}, std::false_type{}, prev, tile + G);
        prev = tile; tile += G; last_par = 0; ++n_done;
    }
    if constexpr (DBG & 128) { asm volatile("" :: "v"(dmy[0]), "v"(dmy[1]), "v"(dmy[2]), "v"(dmy[3]), "v"(dmy[4])); }
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long *st = g_stamp + (blockIdx.x * 4 + wave) * 4;
        st[0] = __builtin_amdgcn_s_memtime() - t_c0;
        st[1] = __builtin_amdgcn_s_memrealtime() - t_r0;
        st[2] = static_cast<unsigned long long>(n_done);
    }
    // drain: the last token tile's epilogue, then the last tile's rows
    if (last_par == 0) static_for<0, EPI_STEPS>([&](auto st_c) { epi_step(P0{}, T2{}, acc2, st_c); });
    else static_for<0, EPI_STEPS>([&](auto st_c) { epi_step(P1{}, T2{}, acc2, st_c); });
    lds_barrier();
    p3_begin(prev);
    if (last_par == 0) static_for<0, P3_STEPS>([&](auto st_c) { p3_step(P0{}, st_c); });
    else static_for<0, P3_STEPS>([&](auto st_c) { p3_step(P1{}, st_c); });
}

}  // namespace

extern "C" int az_nn_conv2_stamps(unsigned long long *out, int n)
{
    if (n > 256 * 4 * 4) n = 256 * 4 * 4;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 2;
}

extern "C" int az_nn_conv_block2(const void *x, const void *weight_folded_ohwi, const float *t1, const float *t2_scaled, void *y,
                                 int64_t batch, float eps, const int64_t *batch_dev, void *stream)
{
    if (batch <= 0 || !x || !weight_folded_ohwi || !t1 || !t2_scaled || !y) return 1;
    static bool attr_set = false;
    static int n_cu = 256;
    if (!attr_set) {
        for (const void *f : {reinterpret_cast<const void *>(k_conv2<0>), reinterpret_cast<const void *>(k_conv2<1>),
                              reinterpret_cast<const void *>(k_conv2<2>), reinterpret_cast<const void *>(k_conv2<3>),
                              reinterpret_cast<const void *>(k_conv2<6>), reinterpret_cast<const void *>(k_conv2<10>),
                              reinterpret_cast<const void *>(k_conv2<14>), reinterpret_cast<const void *>(k_conv2<19>),
                              reinterpret_cast<const void *>(k_conv2<35>), reinterpret_cast<const void *>(k_conv2<67>),
                              reinterpret_cast<const void *>(k_conv2<64>), reinterpret_cast<const void *>(k_conv2<131>), reinterpret_cast<const void *>(k_conv2<387>),
                              reinterpret_cast<const void *>(k_conv2<259>)})
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL) != hipSuccess) return 2;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cu = v;
        }
        if (const char *e = getenv("AZ_NN_CONV2_GRID")) n_cu = atoi(e) > 0 ? atoi(e) : n_cu;
        attr_set = true;
    }
    const int64_t ntiles = (batch + TS - 1) / TS;
    const unsigned grid = static_cast<unsigned>(ntiles < n_cu ? ntiles : n_cu);
    static const int dbg = getenv("AZ_NN_CONV2_DBG") ? atoi(getenv("AZ_NN_CONV2_DBG")) & 511 : 0;
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), L_TOTAL, static_cast<hipStream_t>(stream),
                           static_cast<const uint16_t *>(x), static_cast<const uint16_t *>(weight_folded_ohwi), t1, t2_scaled,
                           static_cast<uint16_t *>(y), batch, eps, batch_dev);
    };
    if (dbg == 1) go(k_conv2<1>);
    else if (dbg == 2) go(k_conv2<2>);
    else if (dbg == 3) go(k_conv2<3>);
    else if (dbg == 6) go(k_conv2<6>);
    else if (dbg == 10) go(k_conv2<10>);
    else if (dbg == 14) go(k_conv2<14>);
    else if (dbg == 19) go(k_conv2<19>);
    else if (dbg == 35) go(k_conv2<35>);
    else if (dbg == 67) go(k_conv2<67>);
    else if (dbg == 64) go(k_conv2<64>);
    else if (dbg == 131) go(k_conv2<131>);
    else if (dbg == 387) go(k_conv2<387>);
    else if (dbg == 259) go(k_conv2<259>);
    else go(k_conv2<0>);
    return 0;
}
