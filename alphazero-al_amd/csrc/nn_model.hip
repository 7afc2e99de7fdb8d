// nn_model.hip - the leaf evaluator as ONE C call: the six launches of the inference twin
// (stem with the embedding, residual blocks, gated attention, both heads; Network.py:144-288 of
// the reference's Connect4 network) issued from native code on the caller's stream.
//
// The object holds POINTERS to the caller's weight arrays (bf16, the layouts az_nn.h documents
// for each kernel) and nothing else: it is immutable after creation, so any number of host
// threads / streams may run az_nn_model_forward on it at once, each with scratch of its own.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <new>
#include <vector>

#include "az_nn.h"
#include "board_sym.h"

struct az_nn_model {
    int kind = AZ_NN_KIND_CONNECT4_CNN;
    az_nn_model_weights w{};
    az_nn_othello_weights ow{};
};

namespace {
constexpr int64_t kTokenBytes = 42 * 64 * 2;     // one sample's (42, 64) bf16 activations
// Othello: bytes per sample of the embedded tokens, a 10x10 and an 8x8 map of 256 channels, the 8-channel bottleneck
constexpr uint64_t kOtTok = 64 * 32 * 2, kOtMap10 = 100 * 256 * 2, kOtMap8 = 64 * 256 * 2, kOtNeck = 64 * 8 * 2;

// az_nn_model_profile: event pairs around the launches of every stride-th forward call - the stem, the
// FIRST residual block, the attention block and the heads, one ring per kind (process-wide; the
// mutex only matters when several streams share the model)
constexpr int kKinds = 4;                         // AZ_NN_PROFILE_*
struct Profile {
    std::mutex mu;
    bool on = false;
    int stride = 1;
    int64_t seen = 0;
    std::vector<hipEvent_t> start[kKinds], stop[kKinds];
    size_t used[kKinds] = {0, 0, 0, 0};
} g_prof;
constexpr size_t kProfileMax = 4096;

// ---- integer-hash evaluator (AZ_NN_KIND_HASH_*): not a model, a pure function of the position the
// feature planes show, built from 64-bit mixing, small integers and one correctly rounded division,
// so that numpy (tests/scenarios.py hash_eval / ot_hash_eval), torch (src/hash_eval.py) and this
// kernel agree bit for bit.  Lets the whole native loop be checked against the CPU oracle and the
// tree kernels be timed without a network.
__device__ __forceinline__ uint64_t position_hash(uint64_t bb0, uint64_t bb1, bool p1_to_move)
{
    uint64_t x = bb0 * 0x9E3779B97F4A7C15ull;
    x ^= (bb1 + 0x7F4A7C159E3779B9ull) * 0xBF58476D1CE4E5B9ull;
    x += p1_to_move ? 0x94D049BB133111EBull : 0x2545F4914F6CDD1Dull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

// one 64-lane wavefront per sample: lane = cell (Connect4: 42 of them), a ballot rebuilds the bitboards
template <bool OTHELLO>
__global__ void __launch_bounds__(256) k_hash_eval(const float *features, const uint8_t *mask, float *probs, float *wdl,
                                                   float *ml, int64_t batch, const int32_t *rows, const int64_t *n_rows)
{
    constexpr int CELLS = OTHELLO ? 64 : 42, A = OTHELLO ? 65 : 7;
    const int lane = threadIdx.x & 63;
    const int64_t b = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int64_t n = batch;
    if (n_rows != nullptr && *n_rows < n) n = *n_rows;
    if (b >= n) return;
    int64_t row = rows != nullptr ? rows[b] : b;
    if (row < 0 || row >= batch) return;
    const float *f = features + row * 3 * CELLS;
    const bool in = lane < CELLS;
    const float own = in ? f[lane] : 0.0f, opp = in ? f[CELLS + lane] : 0.0f;
    const bool p1 = f[2 * CELLS] > 0.0f;
    // display cell -> bit of the reference's bitboards (Connect4.h:15-29: col*7 + 5-row; Othello.h:18-23: row*8+col)
    const int bit = OTHELLO ? lane : ((lane % 7) * 7 + (5 - lane / 7));
    const bool is_p1 = in && ((p1 && own != 0.0f) || (!p1 && opp != 0.0f));
    const bool is_p2 = in && ((p1 && opp != 0.0f) || (!p1 && own != 0.0f));
    uint64_t bb0 = 0, bb1 = 0;
    if (OTHELLO) {
        bb0 = __ballot(is_p1); bb1 = __ballot(is_p2);
    } else {
        // scatter the ballot's cell order into the bitboard's bit order
        const uint64_t m1 = __ballot(is_p1), m2 = __ballot(is_p2);
        for (int c = 0; c < 42; ++c) {
            const int bt = (c % 7) * 7 + (5 - c / 7);
            bb0 |= ((m1 >> c) & 1ull) << bt;
            bb1 |= ((m2 >> c) & 1ull) << bt;
        }
        (void)bit;
    }
    const uint64_t h = position_hash(bb0, bb1, p1);
    const uint8_t *mk = mask != nullptr ? mask + row * A : nullptr;
    if (!OTHELLO) {
        if (lane < 7) {
            const float p = static_cast<float>(1 + ((h >> (4 * lane)) & 15)) / 16.0f;
            probs[row * 7 + lane] = (mk == nullptr || mk[lane]) ? p : 0.0f;
        }
    } else {
        for (int a = lane; a < 65; a += 64) {
            const int k = a >> 4, j = a & 15;
            uint64_t hk = h + 0x9E3779B97F4A7C15ull * static_cast<uint64_t>(k + 1);
            hk ^= hk >> 29; hk *= 0xBF58476D1CE4E5B9ull; hk ^= hk >> 32;
            const float p = static_cast<float>(1 + ((hk >> (4 * j)) & 15)) / 16.0f;
            probs[row * 65 + a] = (mk == nullptr || mk[a]) ? p : 0.0f;
        }
    }
    if (lane < 3) {
        const uint64_t w0 = 1 + ((h >> 28) & 31), w1 = 1 + ((h >> 33) & 31), w2 = 1 + ((h >> 38) & 31);
        const float tot = static_cast<float>(w0 + w1 + w2);
        const uint64_t mine = lane == 0 ? w0 : (lane == 1 ? w1 : w2);
        wdl[row * 3 + lane] = static_cast<float>(mine) / tot;
    }
    if (lane == 3) {
        const float v = static_cast<float>((h >> 43) & 63);
        ml[row] = OTHELLO ? (v / 32.0f - 1.0f) : (v / 2.0f);
    }
}

// the same function of the position a leaf shows under its symmetry id, from the bitboards: one thread per sample
template <bool OTHELLO>
__global__ void __launch_bounds__(256) k_hash_eval_positions(az_nn_positions pos, const uint8_t *mask, float *probs, float *wdl,
                                                             float *ml, int64_t batch, const int32_t *rows, const int64_t *n_rows)
{
    constexpr int A = OTHELLO ? 65 : 7;
    const int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    int64_t n = batch;
    if (n_rows != nullptr && *n_rows < n) n = *n_rows;
    if (b >= n) return;
    const int64_t row = rows != nullptr ? rows[b] : b;
    if (row < 0 || row >= batch) return;
    uint64_t bb0 = pos.bb_p1[row], bb1 = pos.bb_p2[row];
    const int sym = pos.sym[row];
    if (OTHELLO) { bb0 = az::othello_sym(bb0, sym); bb1 = az::othello_sym(bb1, sym); }
    else if (sym) { bb0 = az::mirror_columns(bb0); bb1 = az::mirror_columns(bb1); }
    const uint64_t h = position_hash(bb0, bb1, pos.turn[row] > 0);
    const uint8_t *mk = mask != nullptr ? mask + row * A : nullptr;
    if (!OTHELLO) {
        for (int a = 0; a < 7; ++a) {
            const float p = static_cast<float>(1 + ((h >> (4 * a)) & 15)) / 16.0f;
            probs[row * 7 + a] = (mk == nullptr || mk[a]) ? p : 0.0f;
        }
    } else {
        for (int k = 0; k < 5; ++k) {
            uint64_t hk = h + 0x9E3779B97F4A7C15ull * static_cast<uint64_t>(k + 1);
            hk ^= hk >> 29; hk *= 0xBF58476D1CE4E5B9ull; hk ^= hk >> 32;
            for (int j = 0; j < 16 && k * 16 + j < 65; ++j) {
                const int a = k * 16 + j;
                const float p = static_cast<float>(1 + ((hk >> (4 * j)) & 15)) / 16.0f;
                probs[row * 65 + a] = (mk == nullptr || mk[a]) ? p : 0.0f;
            }
        }
    }
    const uint64_t w0 = 1 + ((h >> 28) & 31), w1 = 1 + ((h >> 33) & 31), w2 = 1 + ((h >> 38) & 31);
    const float tot = static_cast<float>(w0 + w1 + w2);
    wdl[row * 3 + 0] = static_cast<float>(w0) / tot;
    wdl[row * 3 + 1] = static_cast<float>(w1) / tot;
    wdl[row * 3 + 2] = static_cast<float>(w2) / tot;
    const float v = static_cast<float>((h >> 43) & 63);
    ml[row] = OTHELLO ? (v / 32.0f - 1.0f) : (v / 2.0f);
}
}

extern "C" {

int az_nn_model_create(const az_nn_model_weights *w, az_nn_model **out)
{
    if (w == nullptr || out == nullptr) return 1;
    if (w->n_blocks < 0 || w->n_blocks > AZ_NN_MAX_BLOCKS) return 1;
    if ((w->stem_frag == nullptr) != (w->stem_pmap == nullptr)) return 1;
    const void *need[] = {w->emb_own, w->emb_opp, w->pos, w->stem_w, w->stem_b, w->pre_w, w->qkvg_w,
                          w->qn_w, w->kn_w, w->o_w, w->heads.p_norm, w->heads.d_val_w};
    for (const void *p : need)
        if (p == nullptr) return 1;
    for (int i = 0; i < w->n_blocks; ++i)
        if (!w->block_w[i] || !w->block_b[i] || !w->block_gamma[i] || !w->block_beta[i]) return 1;
    auto *m = new (std::nothrow) az_nn_model();
    if (m == nullptr) return 1;
    m->w = *w;
    *out = m;
    return 0;
}

int az_nn_model_create_hash(int game, az_nn_model **out)
{
    if (out == nullptr || (game != 0 && game != 1)) return 1;
    auto *m = new (std::nothrow) az_nn_model();
    if (m == nullptr) return 1;
    m->kind = game == 0 ? AZ_NN_KIND_HASH_CONNECT4 : AZ_NN_KIND_HASH_OTHELLO;
    *out = m;
    return 0;
}

int az_nn_model_create_othello(const az_nn_othello_weights *w, az_nn_model **out)
{
    if (w == nullptr || out == nullptr || w->embed_table == nullptr) return 1;
    if (w->n_body < 2 || w->n_body % 2 != 0 || w->n_convs != w->n_body + 2 || w->n_convs > AZ_NN_OTHELLO_MAX_CONVS) return 1;
    for (int i = 0; i < w->n_convs; ++i)
        if (!w->conv[i].w_packed || !w->conv[i].post_scale || !w->conv[i].post_shift) return 1;
    if (!w->dual_w16 || !w->dual_scale16 || !w->dual_shift16 || !w->heads.board_w || !w->heads.a_fc_wt) return 1;
    auto *m = new (std::nothrow) az_nn_model();
    if (m == nullptr) return 1;
    m->kind = AZ_NN_KIND_OTHELLO_CNN;
    m->ow = *w;
    *out = m;
    return 0;
}

void az_nn_model_destroy(az_nn_model *m) { delete m; }

int az_nn_model_kind(const az_nn_model *m) { return m ? m->kind : -1; }

uint64_t az_nn_model_scratch_bytes(const az_nn_model *m, int64_t batch)
{
    if (m != nullptr && m->kind == AZ_NN_KIND_OTHELLO_CNN)       // tokens, three 10x10x256 maps, two 8x8x256 maps, the bottleneck
        return batch > 0 ? static_cast<uint64_t>(batch) * (kOtTok + 3 * kOtMap10 + 2 * kOtMap8 + kOtNeck) : 0;
    if (m != nullptr && m->kind != AZ_NN_KIND_CONNECT4_CNN) return 0;
    return batch > 0 ? static_cast<uint64_t>(2 * batch * kTokenBytes) : 0;
}

static int forward_impl(const az_nn_model *m, const float *features, const az_nn_positions *positions, const uint8_t *mask,
                        float *probs, float *wdl, float *moves_left, int64_t batch, const int32_t *rows,
                        const int64_t *n_rows, void *scratch, uint64_t scratch_bytes, void *stream);

int az_nn_model_forward(const az_nn_model *m, const float *features, const uint8_t *mask, float *probs,
                        float *wdl, float *moves_left, int64_t batch, const int32_t *rows,
                        const int64_t *n_rows, void *scratch, uint64_t scratch_bytes, void *stream)
{
    if (features == nullptr) return 1;
    return forward_impl(m, features, nullptr, mask, probs, wdl, moves_left, batch, rows, n_rows, scratch, scratch_bytes, stream);
}

int az_nn_model_forward_positions(const az_nn_model *m, const az_nn_positions *positions, const uint8_t *mask, float *probs,
                                  float *wdl, float *moves_left, int64_t batch, const int32_t *rows,
                                  const int64_t *n_rows, void *scratch, uint64_t scratch_bytes, void *stream)
{
    if (positions == nullptr || !positions->bb_p1 || !positions->bb_p2 || !positions->turn || !positions->sym) return 1;
    return forward_impl(m, nullptr, positions, mask, probs, wdl, moves_left, batch, rows, n_rows, scratch, scratch_bytes, stream);
}

static int forward_impl(const az_nn_model *m, const float *features, const az_nn_positions *positions, const uint8_t *mask,
                        float *probs, float *wdl, float *moves_left, int64_t batch, const int32_t *rows,
                        const int64_t *n_rows, void *scratch, uint64_t scratch_bytes, void *stream)
{
    if (m == nullptr || probs == nullptr || wdl == nullptr || moves_left == nullptr) return 1;
    if (batch <= 0) return batch == 0 ? 0 : 1;
    if ((rows == nullptr) != (n_rows == nullptr)) return 1;
    if (m->kind == AZ_NN_KIND_OTHELLO_CNN) {
        // Othello/Network.py:213-227 on the kernels of nn_othello.hip / nn_othello_heads.hip; needs positions + masks
        if (positions == nullptr || mask == nullptr) return 1;
        if (scratch == nullptr || scratch_bytes < az_nn_model_scratch_bytes(m, batch)) return 1;
        const az_nn_othello_weights &o = m->ow;
        char *base = static_cast<char *>(scratch);
        char *tok = base; base += batch * kOtTok;
        char *map[3];
        for (auto &p : map) { p = base; base += batch * kOtMap10; }
        char *pa = base; base += batch * kOtMap8;
        char *pb = base; base += batch * kOtMap8;
        char *neck = base;
        auto conv = [&](int i, const void *x, const void *res, void *y) {
            const az_nn_othello_conv_layer &l = o.conv[i];
            return az_nn_othello_conv(x, l.w_packed, l.pre_scale, l.pre_shift, l.post_scale, l.post_shift, l.residual ? res : nullptr,
                                      y, batch, l.c_in, l.h_in, l.pad, 1, n_rows, stream);
        };
        int rc = az_nn_othello_embed(positions, mask, o.embed_table, tok, batch, rows, n_rows, stream);
        int cur = 0;                                      // map[cur] holds the running hidden state
        if (rc == 0) rc = conv(0, tok, nullptr, map[cur]);
        for (int i = 1; rc == 0 && i + 1 < o.n_body; i += 2) {
            const int y1 = (cur + 1) % 3, y2 = (cur + 2) % 3;
            rc = conv(i, map[cur], nullptr, map[y1]);
            if (rc == 0) rc = conv(i + 1, map[y1], map[cur], map[y2]);
            cur = y2;
        }
        const int hid = (cur + 1) % 3;
        if (rc == 0) rc = conv(o.n_body - 1, map[cur], nullptr, map[hid]);
        if (rc == 0) rc = conv(o.n_body, map[hid], nullptr, pa);
        if (rc == 0) rc = conv(o.n_body + 1, pa, nullptr, pb);
        if (rc == 0) rc = az_nn_othello_conv_narrow(map[hid], o.dual_w16, o.dual_scale16, o.dual_shift16, neck, batch, n_rows, stream);
        if (rc == 0) rc = az_nn_othello_heads(pb, neck, &o.heads, probs, wdl, moves_left, batch, rows, n_rows, stream);
        return rc;
    }
    if (m->kind != AZ_NN_KIND_CONNECT4_CNN && positions != nullptr) {
        const dim3 grid(static_cast<unsigned>((batch + 255) / 256)), block(256);
        if (m->kind == AZ_NN_KIND_HASH_OTHELLO)
            hipLaunchKernelGGL(k_hash_eval_positions<true>, grid, block, 0, static_cast<hipStream_t>(stream), *positions, mask, probs,
                               wdl, moves_left, batch, rows, n_rows);
        else
            hipLaunchKernelGGL(k_hash_eval_positions<false>, grid, block, 0, static_cast<hipStream_t>(stream), *positions, mask, probs,
                               wdl, moves_left, batch, rows, n_rows);
        return 0;
    }
    if (m->kind != AZ_NN_KIND_CONNECT4_CNN) {
        const dim3 grid(static_cast<unsigned>((batch + 3) / 4)), block(256);
        if (m->kind == AZ_NN_KIND_HASH_OTHELLO)
            hipLaunchKernelGGL(k_hash_eval<true>, grid, block, 0, static_cast<hipStream_t>(stream), features, mask, probs, wdl,
                               moves_left, batch, rows, n_rows);
        else
            hipLaunchKernelGGL(k_hash_eval<false>, grid, block, 0, static_cast<hipStream_t>(stream), features, mask, probs, wdl,
                               moves_left, batch, rows, n_rows);
        return 0;
    }
    if (scratch == nullptr || scratch_bytes < az_nn_model_scratch_bytes(m, batch)) return 1;
    const az_nn_model_weights &w = m->w;
    char *a = static_cast<char *>(scratch);
    char *b = a + batch * kTokenBytes;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    // this call carries event pairs?  (never while the stream is capturing: no timestamps there)
    bool timed = false;
    int slot[kKinds] = {-1, -1, -1, -1};
    if (g_prof.on) {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(hs, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        if (g_prof.on && !capturing && (g_prof.seen++ % g_prof.stride) == 0) {
            timed = true;
            for (int kd = 0; kd < kKinds; ++kd)
                if (g_prof.used[kd] < g_prof.start[kd].size()) slot[kd] = static_cast<int>(g_prof.used[kd]++);
        }
    }
    auto begin = [&](int kd) { if (timed && slot[kd] >= 0) (void)hipEventRecord(g_prof.start[kd][slot[kd]], hs); };
    auto end = [&](int kd) { if (timed && slot[kd] >= 0) (void)hipEventRecord(g_prof.stop[kd][slot[kd]], hs); };
    begin(AZ_NN_PROFILE_STEM);
    int rc;
    if (w.stem_frag != nullptr && w.stem_pmap != nullptr)
        rc = positions != nullptr
            ? az_nn_stem_folded_positions(positions, w.stem_frag, w.stem_pmap, a, batch, rows, n_rows, stream)
            : az_nn_stem_folded(features, w.stem_frag, w.stem_pmap, a, batch, rows, n_rows, stream);
    else
        rc = positions != nullptr
            ? az_nn_stem_embed_positions(positions, w.emb_own, w.emb_opp, w.pos, w.stem_w, w.stem_b, a, batch, rows, n_rows, stream)
            : az_nn_stem_embed(features, w.emb_own, w.emb_opp, w.pos, w.stem_w, w.stem_b, a, batch, rows, n_rows, stream);
    end(AZ_NN_PROFILE_STEM);
    for (int i = 0; rc == 0 && i < w.n_blocks; ++i) {
        if (i == 0) begin(AZ_NN_PROFILE_CONV);
        rc = az_nn_conv_block(a, 64, w.block_w[i], w.block_b[i], w.block_gamma[i], w.block_beta[i], 1, b, batch,
                              w.eps, n_rows, stream);
        if (i == 0) end(AZ_NN_PROFILE_CONV);
        char *t = a; a = b; b = t;
    }
    if (rc == 0) {
        begin(AZ_NN_PROFILE_ATTN);
        rc = az_nn_attn_block(a, w.pre_w, w.qkvg_w, w.qn_w, w.kn_w, w.o_w, b, batch, w.eps, n_rows, stream);
        end(AZ_NN_PROFILE_ATTN);
        char *t = a; a = b; b = t;
    }
    if (rc == 0) {
        begin(AZ_NN_PROFILE_HEADS);
        rc = az_nn_heads(a, &w.heads, mask, probs, wdl, moves_left, batch, w.eps, rows, n_rows, stream);
        end(AZ_NN_PROFILE_HEADS);
    }
    return rc;
}

int az_nn_model_profile(int enable)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (enable) {
        for (int kd = 0; kd < kKinds; ++kd)
            while (g_prof.start[kd].size() < kProfileMax) {
                hipEvent_t a, b;
                if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 2;
                g_prof.start[kd].push_back(a); g_prof.stop[kd].push_back(b);
            }
    }
    g_prof.on = enable != 0;
    g_prof.stride = enable > 1 ? enable : 1;
    g_prof.seen = 0;
    return 0;
}

int az_nn_model_profile_read_kernels(double out_ms[4], int64_t out_launches[4])
{
    if (out_ms == nullptr || out_launches == nullptr) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int kd = 0; kd < kKinds; ++kd) {
        double ms = 0.0;
        for (size_t i = 0; i < g_prof.used[kd]; ++i) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, g_prof.start[kd][i], g_prof.stop[kd][i]) != hipSuccess) return 2;
            ms += t;
        }
        out_ms[kd] = ms;
        out_launches[kd] = static_cast<int64_t>(g_prof.used[kd]);
        g_prof.used[kd] = 0;
    }
    return 0;
}

int az_nn_model_profile_read(double *out_ms, int64_t *out_launches)
{
    if (out_ms == nullptr || out_launches == nullptr) return 1;
    double ms[4]; int64_t n[4];
    const int rc = az_nn_model_profile_read_kernels(ms, n);
    if (rc != 0) return rc;
    *out_ms = ms[AZ_NN_PROFILE_CONV];
    *out_launches = n[AZ_NN_PROFILE_CONV];
    return 0;
}

}  // extern "C"
