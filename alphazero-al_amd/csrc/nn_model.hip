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

struct az_nn_model {
    az_nn_model_weights w;
};

namespace {
constexpr int64_t kTokenBytes = 42 * 64 * 2;     // one sample's (42, 64) bf16 activations

// az_nn_model_profile: event pairs around the FIRST residual block of every stride-th forward call
// (process-wide; the mutex only matters when several streams share the model)
struct Profile {
    std::mutex mu;
    bool on = false;
    int stride = 1;
    int64_t seen = 0;
    std::vector<hipEvent_t> start, stop;
    size_t used = 0;
} g_prof;
constexpr size_t kProfileMax = 4096;
}

extern "C" {

int az_nn_model_create(const az_nn_model_weights *w, az_nn_model **out)
{
    if (w == nullptr || out == nullptr) return 1;
    if (w->n_blocks < 0 || w->n_blocks > AZ_NN_MAX_BLOCKS) return 1;
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

void az_nn_model_destroy(az_nn_model *m) { delete m; }

uint64_t az_nn_model_scratch_bytes(const az_nn_model *, int64_t batch)
{
    return batch > 0 ? static_cast<uint64_t>(2 * batch * kTokenBytes) : 0;
}

int az_nn_model_forward(const az_nn_model *m, const float *features, const uint8_t *mask, float *probs,
                        float *wdl, float *moves_left, int64_t batch, const int32_t *rows,
                        const int64_t *n_rows, void *scratch, uint64_t scratch_bytes, void *stream)
{
    if (m == nullptr || features == nullptr || probs == nullptr || wdl == nullptr || moves_left == nullptr) return 1;
    if (batch <= 0) return batch == 0 ? 0 : 1;
    if (scratch == nullptr || scratch_bytes < az_nn_model_scratch_bytes(m, batch)) return 1;
    if ((rows == nullptr) != (n_rows == nullptr)) return 1;
    const az_nn_model_weights &w = m->w;
    char *a = static_cast<char *>(scratch);
    char *b = a + batch * kTokenBytes;
    int rc = az_nn_stem_embed(features, w.emb_own, w.emb_opp, w.pos, w.stem_w, w.stem_b, a, batch, rows, n_rows, stream);
    for (int i = 0; rc == 0 && i < w.n_blocks; ++i) {
        int slot = -1;
        if (i == 0 && g_prof.on) {
            std::lock_guard<std::mutex> lk(g_prof.mu);
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
            if (g_prof.on && !capturing && (g_prof.seen++ % g_prof.stride) == 0 && g_prof.used < g_prof.start.size()) {
                slot = static_cast<int>(g_prof.used++);
                (void)hipEventRecord(g_prof.start[slot], static_cast<hipStream_t>(stream));
            }
        }
        rc = az_nn_conv_block(a, 64, w.block_w[i], w.block_b[i], w.block_gamma[i], w.block_beta[i], 1, b, batch,
                              w.eps, n_rows, stream);
        if (slot >= 0) (void)hipEventRecord(g_prof.stop[slot], static_cast<hipStream_t>(stream));
        char *t = a; a = b; b = t;
    }
    if (rc == 0) {
        rc = az_nn_attn_block(a, w.pre_w, w.qkvg_w, w.qn_w, w.kn_w, w.o_w, b, batch, w.eps, n_rows, stream);
        char *t = a; a = b; b = t;
    }
    if (rc == 0)
        rc = az_nn_heads(a, &w.heads, mask, probs, wdl, moves_left, batch, w.eps, rows, n_rows, stream);
    return rc;
}

int az_nn_model_profile(int enable)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (enable) {
        while (g_prof.start.size() < kProfileMax) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 2;
            g_prof.start.push_back(a); g_prof.stop.push_back(b);
        }
    }
    g_prof.on = enable != 0;
    g_prof.stride = enable > 1 ? enable : 1;
    g_prof.seen = 0;
    return 0;
}

int az_nn_model_profile_read(double *out_ms, int64_t *out_launches)
{
    if (out_ms == nullptr || out_launches == nullptr) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double ms = 0.0;
    for (size_t i = 0; i < g_prof.used; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.start[i], g_prof.stop[i]) != hipSuccess) return 2;
        ms += t;
    }
    *out_ms = ms;
    *out_launches = static_cast<int64_t>(g_prof.used);
    g_prof.used = 0;
    return 0;
}

}  // extern "C"
