// engine.hip - host side of libaz_mcts.so: owns the HBM arenas, snapshots the live config,
// feeds the reference-compatible host generator, and implements the C ABI of
// include/az_mcts.h on top of the kernels in kernels.hip.
//
// Host entry points are synchronous and use the NULL stream (they mirror the reference's
// blocking pybind calls, mcts_bindings.cpp:126-131 etc.).  Device entry points only enqueue
// work on the caller's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "az_mcts.h"
#include "az_nn.h"
#include "host_rng.h"
#include "kernels.h"
#include "games.h"

namespace {

thread_local std::string g_last_error;

struct AzError : std::runtime_error {
    int code;
    AzError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

void hip_check(hipError_t e, const char *what)
{
    if (e != hipSuccess)
        throw AzError(AZ_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_OK(x) hip_check((x), #x)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void ensure(size_t count, bool zero = false)
    {
        if (count <= n) return;
        if (p) HIP_OK(hipFree(p));
        p = nullptr;
        HIP_OK(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
        n = count;
        if (zero) HIP_OK(hipMemset(p, 0, count * sizeof(T)));
    }
};

struct LeafStore {
    DevBuf<int32_t> slot, turn, aux, path_len, path, sym;
    DevBuf<uint64_t> bb0, bb1;
    DevBuf<uint8_t> flags, nvalid;
    int max_path = az::C4_MAX_PATH;
    void ensure(size_t leaves)
    {
        const bool grow = leaves > slot.n;
        slot.ensure(leaves); turn.ensure(leaves); aux.ensure(leaves, true); bb0.ensure(leaves); bb1.ensure(leaves);
        sym.ensure(leaves, true); nvalid.ensure(leaves, true);
        if (grow) {
            flags.ensure(leaves, true);
            path_len.ensure(leaves, true);   // 0 == "no descent recorded" (current_leaf_idx == -1)
            path.ensure(leaves * max_path);
        }
    }
    az::LeafBuf view()
    {
        az::LeafBuf v;
        v.slot = slot.p; v.bb0 = bb0.p; v.bb1 = bb1.p; v.turn = turn.p; v.aux = aux.p; v.nvalid = nvalid.p;
        v.flags = flags.p; v.path_len = path_len.p; v.path = path.p; v.sym = sym.p;
        return v;
    }
};

// event pairs around one kind of kernel
struct EventRing {
    std::vector<hipEvent_t> start, stop;
    size_t used = 0;
    ~EventRing()
    {
        for (auto e : start) (void)hipEventDestroy(e);
        for (auto e : stop) (void)hipEventDestroy(e);
    }
    void allocate(size_t n)
    {
        while (start.size() < n) {
            hipEvent_t a, b;
            HIP_OK(hipEventCreate(&a));
            HIP_OK(hipEventCreate(&b));
            start.push_back(a); stop.push_back(b);
        }
    }
    bool begin(hipStream_t s)
    {
        if (used >= start.size()) return false;
        // an event recorded into a graph under capture has no timestamp to read back
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) return false;
        HIP_OK(hipEventRecord(start[used], s));
        return true;
    }
    void end(hipStream_t s) { HIP_OK(hipEventRecord(stop[used], s)); ++used; }
    void read(double &ms, int64_t &n)
    {
        ms = 0.0; n = 0;
        for (size_t i = 0; i < used; ++i) {
            float t = 0.f;
            HIP_OK(hipEventElapsedTime(&t, start[i], stop[i]));
            ms += t; ++n;
        }
        used = 0;
    }
};

// static geometry of a game as the host needs it
struct Geo {
    int actions, cells, rows, cols, stats, max_path, sym_choices;
    int max_edges;      // most legal moves a position can have = records an expansion can append (Othello: 33)
};

Geo geo_of(int game)
{
    if (game == AZ_GAME_OTHELLO)
        return Geo{az::OT_ACTIONS, az::OT_CELLS, 8, 8, az::OT_STATS, az::OT_MAX_PATH, 4, 33};
    return Geo{az::C4_ACTIONS, az::C4_CELLS, az::C4_ROWS, az::C4_COLS, az::C4_STATS, az::C4_MAX_PATH, 2, 7};
}

constexpr int64_t kInitialSlots = 4096;
constexpr int kCpuctTab = 1 << 16;

}  // namespace

struct az_mcts {
    int game = AZ_GAME_CONNECT4;
    Geo geo = geo_of(AZ_GAME_CONNECT4);
    int B = 0;
    int device = 0;
    az_search_config cfg;

    // trees
    DevBuf<az::HotRec> hot;
    DevBuf<az::ColdRec> cold;
    DevBuf<int32_t> root, used;
    DevBuf<uint8_t> half;     // which of its two arena halves a tree lives in (tree_layout.h)
    int64_t S = 0;            // records per half
    int64_t used_bound = 1;   // host-side upper bound of max(used[])
    // What the trees occupy after a re-rooting, reported by the prune kernel without stalling the host:
    // prune number q leaves its maximum in live_ring[q % 8] (pinned host memory) through an async copy;
    // growth_after[q % 8] sums the room handed out by ensure_room since that prune was issued, so that
    // `arrived value + growth since` is an upper bound of max(used[]) again.
    DevBuf<int> max_live;
    volatile int *live_ring = nullptr;
    int64_t prune_seq = 0;
    int64_t ring_seq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t growth_after[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t last_extra = 0;   // room asked for by the last ensure_room call
    int64_t growth_since_prune = 0;   // room handed out since the previous re-rooting = one ply's worst-case growth,
                                      // whether it was asked for in one call (device loop) or in one call per
                                      // backprop (host entry points)
    int64_t epoch = 0;        // bumped whenever a buffer the dev_* kernels address moves

    // roots of the current call
    DevBuf<uint64_t> r_bb0, r_bb1;
    DevBuf<int32_t> r_turn, r_last;

    LeafStore vl_leaf, plain_leaf;
    int vl_stride = 0;        // K of the last VL selection (flat = tree*K + k)
    bool last_select_vl = false;

    // c_puct table
    DevBuf<float> tab;
    float tab_c_init = NAN, tab_c_base = NAN;
    DevBuf<float> term_tab;   // Othello terminal_aux by diff*turn + 64
    float term_tab_scale = NAN;

    DevBuf<unsigned long long> counters;
    DevBuf<int> err;
    int *err_host = nullptr;  // pinned copy of `err` (az_mcts_dev_check)
    DevBuf<uint64_t> call_ctr;
    // recorded draws that stand in for the device generator (az_mcts_dev_replay; parity tests)
    const int32_t *replay_sym = nullptr;
    int64_t replay_stride = 0, replay_calls = 0, replay_next = 0;
    const float *replay_noise = nullptr;
    // device transposition table of evaluator outputs (tt_kernels.hip)
    DevBuf<uint8_t> tt_entries;     // 2^n entries of az::tt_entry_bytes(game) bytes
    DevBuf<unsigned long long> tt_stats;
    DevBuf<uint64_t> tt_keys;
    uint64_t tt_mask = 0;
    int64_t select_launches = 0, backprop_launches = 0;
    const float *noise_eps_tree = nullptr;     // caller-owned device array (az_mcts_dev_set_noise_epsilons)
    // leaf batch of az_mcts_dev_search: evaluator inputs, outputs, compact row list, activations
    DevBuf<float> ev_feat, ev_probs, ev_wdl, ev_ml;
    DevBuf<uint8_t> ev_mask, ev_scratch;
    DevBuf<int32_t> ev_rows;
    DevBuf<int64_t> ev_nrows;
    // one chunk of az_mcts_dev_tt_refresh
    DevBuf<float> rf_probs, rf_wdl, rf_ml;
    DevBuf<uint8_t> rf_mask, rf_scratch;
    DevBuf<uint64_t> rf_bb0, rf_bb1;
    DevBuf<int32_t> rf_rows, rf_turn, rf_sym;
    DevBuf<int64_t> rf_count;
    DevBuf<uint64_t> rf_keys;
    bool profiling = false;
    int profile_stride = 1;      // time every profile_stride-th launch of a kind
    int64_t profile_seen[2] = {0, 0};
    EventRing ev_select, ev_backprop;
    const char *timed_select_kernel = "";   // the kernel behind the newest timed selection launch (az_mcts_timed_select_kernel)

    // IO buffers of the host entry points
    DevBuf<int8_t> io_boards_in, io_boards_out;
    DevBuf<int32_t> io_turns_in, io_sym_in, io_actions, io_noise_req, io_counts;
    DevBuf<uint8_t> io_mask_out, io_is_term, io_reset_mask;
    DevBuf<float> io_policy, io_d, io_p1, io_p2, io_ml, io_noise, io_stats;

    // host generator and what it needs to know between search and backprop
    az::HostRng rng;
    uint64_t dev_seed = 0x5eed;
    std::vector<uint8_t> stash_flags_vl, stash_flags_plain;
    std::vector<uint8_t> stash_root_nv;   // open columns of each root (valid moves of an unexpanded root)
    std::vector<uint8_t> pending_reset;
    bool any_pending_reset = false;

    az::TreeArena arena()
    {
        az::TreeArena a;
        a.hot = hot.p; a.cold = cold.p; a.half = half.p; a.root = root.p; a.used = used.p; a.S = S; a.B = B;
        return a;
    }
    az::RootState roots()
    {
        az::RootState r;
        r.bb0 = r_bb0.p; r.bb1 = r_bb1.p; r.turn = r_turn.p; r.aux = r_last.p;
        return r;
    }

    void ensure_table()
    {
        // Othello.h:260-266 with the host libm: atanf(raw / scale) * (2.0f / 3.14159265f)
        if (!term_tab.p || cfg.score_scale != term_tab_scale) {
            std::vector<float> h(129);
            for (int i = 0; i < 129; ++i) {
                const float raw = static_cast<float>(i - 64);
                h[i] = std::atan(raw / cfg.score_scale) * (2.0f / 3.14159265f);
            }
            if (!term_tab.p) ++epoch;
            term_tab.ensure(129);
            HIP_OK(hipMemcpy(term_tab.p, h.data(), sizeof(float) * 129, hipMemcpyHostToDevice));
            term_tab_scale = cfg.score_scale;
        }
        // logf through the host libm, float arithmetic in the reference's order (MCTS.h:213-214)
        if (tab.p && cfg.c_init == tab_c_init && cfg.c_base == tab_c_base) return;
        // second half: sqrtf(parent_n) - correctly rounded on either side, tabulated to take ~17 instructions
        // out of a level of the Connect4 selection kernels
        std::vector<float> h(2 * kCpuctTab);
        const float c_init = cfg.c_init, c_base = cfg.c_base;
        for (int n = 0; n < kCpuctTab; ++n) {
            const float parent_n = static_cast<float>(n);
            h[n] = c_init + std::log((parent_n + c_base + 1.0f) / c_base);
            h[kCpuctTab + n] = std::sqrt(parent_n);
        }
        if (!tab.p) ++epoch;
        tab.ensure(2 * kCpuctTab);
        HIP_OK(hipMemcpy(tab.p, h.data(), sizeof(float) * 2 * kCpuctTab, hipMemcpyHostToDevice));
        tab_c_init = c_init; tab_c_base = c_base;
    }

    az::SearchParams params()
    {
        ensure_table();
        az::SearchParams p;
        p.c_init = cfg.c_init; p.c_base = cfg.c_base; p.noise_eps = cfg.noise_epsilon;
        p.fpu_reduction = cfg.fpu_reduction; p.mlh_slope = cfg.mlh_slope; p.mlh_cap = cfg.mlh_cap;
        p.value_decay = cfg.value_decay; p.alpha = cfg.dirichlet_alpha;
        p.score_utility_factor = cfg.score_utility_factor; p.term_aux_tab = term_tab.p;
        p.vl_count = cfg.vl_count; p.use_symmetry = cfg.use_symmetry ? 1 : 0;
        p.cpuct_tab = tab.p; p.tab_n = kCpuctTab;
        p.seed = dev_seed; p.call_ptr = call_ctr.p;
        p.noise_eps_tree = noise_eps_tree;
        return p;
    }

    void flush_resets(hipStream_t s)
    {
        if (!any_pending_reset) return;
        HIP_OK(hipStreamSynchronize(s));          // an earlier reset launch may still be reading the mask
        io_reset_mask.ensure(B);
        HIP_OK(hipMemcpy(io_reset_mask.p, pending_reset.data(), B, hipMemcpyHostToDevice));
        az::launch_reset_masked(arena(), io_reset_mask.p, s);
        std::fill(pending_reset.begin(), pending_reset.end(), 0);
        any_pending_reset = false;
    }

    int64_t true_max_used()
    {
        std::vector<int32_t> h(B);
        HIP_OK(hipMemcpy(h.data(), used.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
        return *std::max_element(h.begin(), h.end());
    }

    void grow(int64_t new_S)
    {
        HIP_OK(hipDeviceSynchronize());
        const int64_t keep = std::min<int64_t>(S, true_max_used());
        DevBuf<az::HotRec> nh;
        DevBuf<az::ColdRec> nc;
        nh.ensure(static_cast<size_t>(B) * 2 * new_S);
        nc.ensure(static_cast<size_t>(B) * 2 * new_S);
        // one row per half (slots are relative to a half: nothing to renumber)
        HIP_OK(hipMemcpy2D(nh.p, new_S * sizeof(az::HotRec), hot.p, S * sizeof(az::HotRec),
                           keep * sizeof(az::HotRec), static_cast<size_t>(B) * 2, hipMemcpyDeviceToDevice));
        HIP_OK(hipMemcpy2D(nc.p, new_S * sizeof(az::ColdRec), cold.p, S * sizeof(az::ColdRec),
                           keep * sizeof(az::ColdRec), static_cast<size_t>(B) * 2, hipMemcpyDeviceToDevice));
        std::swap(hot.p, nh.p); std::swap(hot.n, nh.n);
        std::swap(cold.p, nc.p); std::swap(cold.n, nc.n);
        S = new_S;
        ++epoch;
    }

    // the newest re-rooting whose occupancy figure has arrived tightens the host-side bound
    void tighten_bound()
    {
        for (int64_t q = prune_seq; q > 0 && q > prune_seq - 8; --q) {
            const int v = live_ring[q % 8];
            if (ring_seq[q % 8] == q && v >= 0) {
                used_bound = std::min<int64_t>(used_bound, static_cast<int64_t>(v) + growth_after[q % 8]);
                return;
            }
        }
    }
    bool room_needs_device(int64_t extra)
    {
        if (used_bound + extra > S) tighten_bound();
        return used_bound + extra > S;
    }

    // room for `extra` more records in every tree (an expansion appends at most A records)
    void ensure_room(int64_t extra)
    {
        if (used_bound + extra > S) tighten_bound();
        if (used_bound + extra > S) {
            used_bound = true_max_used();
            if (used_bound + extra > S) grow(std::max<int64_t>(2 * S, used_bound + extra));
        }
        used_bound += extra;
        for (auto &g : growth_after) g += extra;
        last_extra = extra;
        growth_since_prune += extra;
    }

    void check_device_error()
    {
        int e = 0;
        HIP_OK(hipMemcpy(&e, err.p, sizeof(int), hipMemcpyDeviceToHost));
        if (e) {
            HIP_OK(hipMemset(err.p, 0, sizeof(int)));
            if (err_host) *err_host = 0;
            throw AzError(AZ_ERR_CAPACITY, device_error_text(e));
        }
    }
    static std::string device_error_text(int e)
    {
        std::string msg;
        if (e & az::ERR_ARENA_OVERFLOW) msg += "tree arena overflow on device (expansions were dropped)";
        if (e & az::ERR_LIST_OVERFLOW) msg += std::string(msg.empty() ? "" : "; ") + "compact leaf list overflow on device (entries were dropped)";
        if (msg.empty()) msg = "device error word " + std::to_string(e);
        return msg;
    }
    // re-rooting on `s` (k_prune: the kept subtrees move to the other arena halves); its occupancy figure
    // travels to live_ring behind it
    void prune_on(const int32_t *actions_dev, int32_t *noise_req, bool dev_noise, const float *noise_replay, hipStream_t s)
    {
        const int64_t q = ++prune_seq;
        // the slot's previous figure (prune q - 8) must have landed before the slot is handed out again
        if (ring_seq[q % 8] != 0 && live_ring[q % 8] < 0) HIP_OK(hipStreamSynchronize(s));
        live_ring[q % 8] = -1;
        ring_seq[q % 8] = q;
        growth_after[q % 8] = 0;
        HIP_OK(hipMemsetAsync(max_live.p, 0, sizeof(int), s));
        // compact the trees that could not take two more plies' worth of growth where they are
        // (AZ_COMPACT_ALWAYS=1: every tree at every re-rooting)
        static const bool always = getenv("AZ_COMPACT_ALWAYS") != nullptr && getenv("AZ_COMPACT_ALWAYS")[0] == '1';
        // (one ply = what was reserved since the previous re-rooting; the host entry points reserve per backprop
        // call, so the last call's figure alone would let a tree run into the end of its half mid-search)
        const int64_t ply = std::max<int64_t>(std::max(growth_since_prune, last_extra), geo.max_edges);
        growth_since_prune = 0;
        const int64_t above = always ? 0 : std::max<int64_t>(S / 8, S - 2 * ply);
        az::launch_prune(game, arena(), params(), actions_dev, noise_req, dev_noise, s, noise_replay, max_live.p, err.p,
                         static_cast<int>(above));
        HIP_OK(hipMemcpyAsync(const_cast<int *>(&live_ring[q % 8]), max_live.p, sizeof(int), hipMemcpyDeviceToHost, s));
    }
    ~az_mcts()
    {
        if (err_host) (void)hipHostFree(err_host);
        if (live_ring) (void)hipHostFree(const_cast<int *>(live_ring));
    }
};

namespace {

az_mcts *create_engine(int game, int n_envs, int device)
{
    if (game != AZ_GAME_CONNECT4 && game != AZ_GAME_OTHELLO) throw AzError(AZ_ERR_ARG, "unknown game id");
    if (n_envs <= 0) throw AzError(AZ_ERR_ARG, "n_envs must be positive");
    int count = 0;
    const hipError_t dev_err = hipGetDeviceCount(&count);
    if (dev_err != hipSuccess || count <= 0)
        throw AzError(AZ_ERR_DEVICE, std::string("no HIP device available: the search engine runs on the GPU only (hipGetDeviceCount: ") +
                                         hipGetErrorString(dev_err) + ", " + std::to_string(count) + " devices)");
    if (device >= 0) HIP_OK(hipSetDevice(device));
    auto *m = new az_mcts();
    try {
        HIP_OK(hipGetDevice(&m->device));
        m->game = game;
        m->geo = geo_of(game);
        m->vl_leaf.max_path = m->plain_leaf.max_path = m->geo.max_path;
        m->B = n_envs;
        m->cfg.c_init = 1.25f; m->cfg.c_base = 19652.0f; m->cfg.dirichlet_alpha = 0.3f;
        m->cfg.noise_epsilon = 0.25f; m->cfg.fpu_reduction = 0.4f; m->cfg.mlh_slope = 0.0f;
        m->cfg.mlh_cap = 0.2f; m->cfg.score_utility_factor = 0.0f; m->cfg.score_scale = 8.0f;
        m->cfg.value_decay = 1.0f; m->cfg.use_symmetry = 1; m->cfg.vl_count = 1;
        m->S = kInitialSlots;
        m->hot.ensure(static_cast<size_t>(n_envs) * 2 * m->S);
        m->cold.ensure(static_cast<size_t>(n_envs) * 2 * m->S);
        m->root.ensure(n_envs); m->used.ensure(n_envs); m->half.ensure(n_envs, true);
        m->max_live.ensure(1, true);
        {
            int *ring = nullptr;
            HIP_OK(hipHostMalloc(reinterpret_cast<void **>(&ring), 8 * sizeof(int), hipHostMallocDefault));
            for (int i = 0; i < 8; ++i) ring[i] = -1;
            m->live_ring = ring;
        }
        m->r_bb0.ensure(n_envs, true); m->r_bb1.ensure(n_envs, true);
        m->r_turn.ensure(n_envs, true); m->r_last.ensure(n_envs, true);
        m->counters.ensure(az::CNT_N * az::CNT_STRIPES, true);
        m->err.ensure(1, true);
        HIP_OK(hipHostMalloc(reinterpret_cast<void **>(&m->err_host), sizeof(int), hipHostMallocDefault));
        *m->err_host = 0;
        m->call_ctr.ensure(1, true);
        m->plain_leaf.ensure(n_envs);
        m->pending_reset.assign(n_envs, 0);
        m->stash_root_nv.assign(n_envs, 0);
        az::launch_init_trees(m->arena(), nullptr);
        HIP_OK(hipDeviceSynchronize());
    } catch (...) {
        delete m;
        throw;
    }
    return m;
}

template <class F>
int guarded(F &&f)
{
    try {
        f();
        return AZ_OK;
    } catch (const AzError &e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return AZ_ERR_DEVICE;
    }
}

void require(bool ok, const std::string &msg)
{
    if (!ok) throw AzError(AZ_ERR_ARG, msg);
}

// Shared body of search_batch / search_batch_vl (BatchedMCTS.h:119-171, 227-286)
void host_search(az_mcts *m, int K, bool vl, const int8_t *boards, const int32_t *turns,
                 int8_t *out_boards, float *out_d, float *out_p1w, float *out_p2w,
                 uint8_t *out_is_term, int32_t *out_turns, int32_t *out_sym, uint8_t *out_mask)
{
    HIP_OK(hipSetDevice(m->device));
    const int B = m->B;
    const size_t total = static_cast<size_t>(B) * K;
    hipStream_t s = nullptr;
    m->flush_resets(s);

    const int A = m->geo.actions, CELLS = m->geo.cells;
    m->io_boards_in.ensure(static_cast<size_t>(B) * CELLS);
    m->io_turns_in.ensure(B);
    HIP_OK(hipMemcpy(m->io_boards_in.p, boards, static_cast<size_t>(B) * CELLS, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(m->io_turns_in.p, turns, sizeof(int32_t) * B, hipMemcpyHostToDevice));
    az::launch_import(m->game, m->io_boards_in.p, m->io_turns_in.p, m->roots(), B, s);

    LeafStore &ls = vl ? m->vl_leaf : m->plain_leaf;
    ls.ensure(total);
    if (vl) m->vl_stride = K;
    m->last_select_vl = vl;
    const az::SearchParams p = m->params();
    az::launch_select(m->game, m->arena(), m->roots(), ls.view(), p, K, vl, m->counters.p, s);
    ++m->select_launches;

    std::vector<uint8_t> flags(total), nvalid(total);
    HIP_OK(hipMemcpy(flags.data(), ls.flags.p, total, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(nvalid.data(), ls.nvalid.p, total, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(out_turns, ls.turn.p, sizeof(int32_t) * total, hipMemcpyDeviceToHost));

    // symmetry ids in env order, one draw per NON-terminal leaf (BatchedMCTS.h:148-158,261-271)
    std::vector<int32_t> sym(total, 0);
    const bool use_sym = m->cfg.use_symmetry != 0;
    for (size_t f = 0; f < total; ++f) {
        const bool term = (flags[f] & az::LEAF_TERMINAL) != 0;
        const int code = (flags[f] >> az::LEAF_RESULT_SHIFT) & 3;
        out_is_term[f] = term ? 1 : 0;
        out_d[f] = (term && code == 0) ? 1.0f : 0.0f;
        out_p1w[f] = (term && code == 1) ? 1.0f : 0.0f;
        out_p2w[f] = (term && code == 2) ? 1.0f : 0.0f;
        if (!term && use_sym) {               // Connect4: id in {0,1}; Othello: {0,2,6,7}[index] (Othello.h:363-367)
            const int choice = m->rng.uniform_int(m->geo.sym_choices - 1);
            static const int ot_ids[4] = {0, 2, 6, 7};
            sym[f] = m->game == AZ_GAME_OTHELLO ? ot_ids[choice] : choice;
        }
    }
    if (out_sym) std::memcpy(out_sym, sym.data(), sizeof(int32_t) * total);
    HIP_OK(hipMemcpy(ls.sym.p, sym.data(), sizeof(int32_t) * total, hipMemcpyHostToDevice));

    m->io_boards_out.ensure(total * CELLS);
    m->io_mask_out.ensure(total * A);
    az::launch_export(m->game, ls.view(), p, static_cast<int>(total), false, m->io_boards_out.p,
                      m->io_mask_out.p, nullptr, s);
    HIP_OK(hipMemcpy(out_boards, m->io_boards_out.p, total * CELLS, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(out_mask, m->io_mask_out.p, total * A, hipMemcpyDeviceToHost));

    // what the host generator needs at expansion time: which leaves are unexpanded roots and how
    // many legal moves (= noise draws) such a root has
    for (int i = 0; i < B; ++i)
        for (int k = 0; k < K; ++k) {
            const size_t f = static_cast<size_t>(i) * K + k;
            if (flags[f] & az::LEAF_ROOT_UNEXPANDED) { m->stash_root_nv[i] = nvalid[f]; break; }
        }
    (vl ? m->stash_flags_vl : m->stash_flags_plain) = std::move(flags);
}

// Shared body of backprop_batch / backprop_batch_vl (BatchedMCTS.h:176-199, 296-332)
void host_backprop(az_mcts *m, int K, bool vl, const float *policy, const float *d, const float *p1w,
                   const float *p2w, const float *ml, const uint8_t *is_term, const int32_t *sym_ids)
{
    HIP_OK(hipSetDevice(m->device));
    const int B = m->B;
    const int A = m->geo.actions;
    hipStream_t s = nullptr;
    m->flush_resets(s);
    LeafStore &ls = vl ? m->vl_leaf : m->plain_leaf;
    if (vl) require(m->vl_stride == K, "backprop_batch_vl: K differs from the preceding search_batch_vl");
    const size_t total = static_cast<size_t>(B) * K;
    ls.ensure(total);
    m->ensure_room(static_cast<int64_t>(K) * m->geo.max_edges);

    // Dirichlet noise for roots expanded by this call, drawn in env order (MCTS.h:347-363)
    m->io_noise.ensure(static_cast<size_t>(B) * A, true);
    const std::vector<uint8_t> &fl = vl ? m->stash_flags_vl : m->stash_flags_plain;
    if (m->cfg.dirichlet_alpha > 0.0f && fl.size() == total) {
        std::vector<float> noise(static_cast<size_t>(B) * A, 0.0f);
        bool any = false;
        for (int i = 0; i < B; ++i)
            for (int k = 0; k < K; ++k) {
                const size_t f = static_cast<size_t>(i) * K + k;
                if ((fl[f] & az::LEAF_ROOT_UNEXPANDED) && !is_term[f]) {
                    m->rng.dirichlet(m->cfg.dirichlet_alpha, &noise[static_cast<size_t>(i) * A],
                                     m->stash_root_nv[i]);
                    any = true;
                    break;   // later k find the root expanded (MCTS.h:601-607)
                }
            }
        if (any)
            HIP_OK(hipMemcpy(m->io_noise.p, noise.data(), sizeof(float) * noise.size(), hipMemcpyHostToDevice));
    }
    (vl ? m->stash_flags_vl : m->stash_flags_plain).clear();

    m->io_policy.ensure(total * A); m->io_d.ensure(total); m->io_p1.ensure(total);
    m->io_p2.ensure(total); m->io_ml.ensure(total); m->io_is_term.ensure(total);
    HIP_OK(hipMemcpy(m->io_policy.p, policy, sizeof(float) * total * A, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(m->io_d.p, d, sizeof(float) * total, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(m->io_p1.p, p1w, sizeof(float) * total, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(m->io_p2.p, p2w, sizeof(float) * total, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(m->io_ml.p, ml, sizeof(float) * total, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(m->io_is_term.p, is_term, total, hipMemcpyHostToDevice));
    az::EvalIn in{};
    in.policy = m->io_policy.p; in.d = m->io_d.p; in.p1w = m->io_p1.p; in.p2w = m->io_p2.p;
    in.is_term = m->io_is_term.p; in.moves_left = m->io_ml.p; in.wdl_rel = nullptr;
    in.root_noise = m->io_noise.p;
    in.sym = nullptr;   // plain: the ids stored by search_batch (pending_sym_ids_, BatchedMCTS.h:152)
    if (vl) {
        m->io_sym_in.ensure(total);
        HIP_OK(hipMemcpy(m->io_sym_in.p, sym_ids, sizeof(int32_t) * total, hipMemcpyHostToDevice));
        in.sym = m->io_sym_in.p;
    }
    az::launch_backprop(m->game, m->arena(), ls.view(), m->params(), K, vl, false, in, m->counters.p, m->err.p, s);
    ++m->backprop_launches;
    HIP_OK(hipStreamSynchronize(s));
    m->check_device_error();
}

}  // namespace

// ====================================================================== C ABI

extern "C" {

const char *az_last_error(void) { return g_last_error.c_str(); }

static bool known_game(int game) { return game == AZ_GAME_CONNECT4 || game == AZ_GAME_OTHELLO; }
int az_game_action_size(int game) { return known_game(game) ? geo_of(game).actions : -1; }
int az_game_board_size(int game) { return known_game(game) ? geo_of(game).cells : -1; }
int az_game_board_rows(int game) { return known_game(game) ? geo_of(game).rows : -1; }
int az_game_board_cols(int game) { return known_game(game) ? geo_of(game).cols : -1; }

int az_mcts_create(int game, int n_envs, int device, az_mcts **out)
{
    return guarded([&] {
        require(out != nullptr, "out is null");
        *out = create_engine(game, n_envs, device);
    });
}

void az_mcts_destroy(az_mcts *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    delete m;
}

az_search_config *az_mcts_config(az_mcts *m) { return &m->cfg; }
int az_mcts_num_envs(const az_mcts *m) { return m->B; }

int az_mcts_set_seed(az_mcts *m, int seed)
{
    return guarded([&] {
        if (seed < 0) {                       // BatchedMCTS.h:73-76
            m->rng.seed_random();
            m->dev_seed = (static_cast<uint64_t>(m->rng.next()) << 32) | m->rng.next();
        } else {                              // thread 0: seed + 0*10007 (BatchedMCTS.h:79-81)
            m->rng.seed(static_cast<uint32_t>(seed));
            m->dev_seed = static_cast<uint64_t>(static_cast<uint32_t>(seed)) * 0x9E3779B97F4A7C15ull + 1;
        }
        HIP_OK(hipSetDevice(m->device));
        HIP_OK(hipMemset(m->call_ctr.p, 0, sizeof(uint64_t)));
    });
}

int az_mcts_reset_env(az_mcts *m, int env)
{
    if (env >= 0 && env < m->B) {             // silently ignores out-of-range (BatchedMCTS.h:95)
        m->pending_reset[env] = 1;
        m->any_pending_reset = true;
    }
    return AZ_OK;
}

int az_mcts_prune_roots(az_mcts *m, const int32_t *actions, int64_t n)
{
    return guarded([&] {
        require(n == m->B, "prune_roots: actions size (" + std::to_string(n) + ") must match n_envs (" +
                               std::to_string(m->B) + ")");
        HIP_OK(hipSetDevice(m->device));
        hipStream_t s = nullptr;
        m->flush_resets(s);
        const int B = m->B;
        const int A = m->geo.actions;
        m->io_actions.ensure(B); m->io_noise_req.ensure(B, true);
        HIP_OK(hipMemcpy(m->io_actions.p, actions, sizeof(int32_t) * B, hipMemcpyHostToDevice));
        m->prune_on(m->io_actions.p, m->io_noise_req.p, false, nullptr, s);
        if (m->cfg.dirichlet_alpha > 0.0f) {  // apply_root_noise, env order (MCTS.h:113-132)
            std::vector<int32_t> req(B);
            HIP_OK(hipMemcpy(req.data(), m->io_noise_req.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
            std::vector<float> noise(static_cast<size_t>(B) * A, 0.0f);
            bool any = false;
            for (int i = 0; i < B; ++i)
                if (req[i] > 0) {
                    m->rng.dirichlet(m->cfg.dirichlet_alpha, &noise[static_cast<size_t>(i) * A], req[i]);
                    any = true;
                }
            if (any) {
                m->io_noise.ensure(static_cast<size_t>(B) * A);
                HIP_OK(hipMemcpy(m->io_noise.p, noise.data(), sizeof(float) * noise.size(), hipMemcpyHostToDevice));
                az::launch_apply_noise(m->game, m->arena(), m->io_noise_req.p, m->io_noise.p, s);
            }
        }
        HIP_OK(hipStreamSynchronize(s));
    });
}

int az_mcts_search_batch(az_mcts *m, const int8_t *boards, const int32_t *turns, int64_t n,
                         int8_t *out_boards, float *out_term_d, float *out_term_p1w,
                         float *out_term_p2w, uint8_t *out_is_term, int32_t *out_turns,
                         uint8_t *out_valid_mask)
{
    return guarded([&] {
        require(n == m->B, "search_batch: input_boards batch size (" + std::to_string(n) +
                               ") must match n_envs (" + std::to_string(m->B) + ")");
        host_search(m, 1, false, boards, turns, out_boards, out_term_d, out_term_p1w, out_term_p2w,
                    out_is_term, out_turns, nullptr, out_valid_mask);
    });
}

int az_mcts_backprop_batch(az_mcts *m, const float *policy, const float *d, const float *p1w,
                           const float *p2w, const float *moves_left, const uint8_t *is_term,
                           int64_t n)
{
    return guarded([&] {
        require(n == m->B, "backprop_batch: policy_logits batch size (" + std::to_string(n) +
                               ") must match n_envs (" + std::to_string(m->B) + ")");
        host_backprop(m, 1, false, policy, d, p1w, p2w, moves_left, is_term, nullptr);
    });
}

int az_mcts_remove_all_vl(az_mcts *m, int K)
{
    return guarded([&] {
        HIP_OK(hipSetDevice(m->device));
        if (m->vl_stride <= 0 || K <= 0) return;
        const int kk = std::min(K, m->vl_stride);          // safe_K, MCTS.h:563
        az::launch_remove_vl(m->game, m->arena(), m->vl_leaf.view(), m->params(), kk, m->vl_stride, nullptr);
        HIP_OK(hipStreamSynchronize(nullptr));
    });
}

int az_mcts_search_batch_vl(az_mcts *m, int K, const int8_t *boards, const int32_t *turns,
                            int64_t n, int8_t *out_boards, float *out_term_d,
                            float *out_term_p1w, float *out_term_p2w, uint8_t *out_is_term,
                            int32_t *out_turns, int32_t *out_sym_ids, uint8_t *out_valid_mask)
{
    return guarded([&] {
        require(n == m->B, "search_batch_vl: input batch (" + std::to_string(n) + ") != n_envs (" +
                               std::to_string(m->B) + ")");
        require(K >= 1, "search_batch_vl: K must be >= 1");
        host_search(m, K, true, boards, turns, out_boards, out_term_d, out_term_p1w, out_term_p2w,
                    out_is_term, out_turns, out_sym_ids, out_valid_mask);
    });
}

int az_mcts_backprop_batch_vl(az_mcts *m, int K, const float *policy, const float *d,
                              const float *p1w, const float *p2w, const float *moves_left,
                              const uint8_t *is_term, const int32_t *sym_ids, int64_t total)
{
    return guarded([&] {
        require(K >= 1, "backprop_batch_vl: K must be >= 1");
        require(total == static_cast<int64_t>(m->B) * K,
                "backprop_batch_vl: policy batch (" + std::to_string(total) + ") != N*K (" +
                    std::to_string(static_cast<int64_t>(m->B) * K) + ")");
        host_backprop(m, K, true, policy, d, p1w, p2w, moves_left, is_term, sym_ids);
    });
}

}  // extern "C"

namespace {
// RolloutEvaluator::evaluate_single (RolloutEvaluator.h:23-48) on the host: result of a uniformly
// random playout from `s` - 0 draw, 1 P1 wins, 2 P2 wins - with one uniform_int(0, nv-1) draw per move
template <class G>
int host_playout(az::GameState s, az::HostRng &rng)
{
    for (;;) {
        const int res = G::result(s);
        if (res >= 0) return res;
        const int nv = G::num_valid(s);
        if (nv <= 0) return 0;
        G::step(s, G::nth_valid(s, rng.uniform_int(nv - 1)));
    }
}

void rollout_common_begin(az_mcts *m, const int8_t *boards, const int32_t *turns, int64_t n, int n_playout, hipStream_t s)
{
    require(n == m->B, "search: input_boards batch size (" + std::to_string(n) + ") must match n_envs (" +
                           std::to_string(m->B) + ")");
    require(n_playout >= 0, "search: n_playout must be >= 0");
    HIP_OK(hipSetDevice(m->device));
    const int B = m->B;
    const int A = m->geo.actions, CELLS = m->geo.cells;
    m->flush_resets(s);
    m->io_boards_in.ensure(static_cast<size_t>(B) * CELLS);
    m->io_turns_in.ensure(B);
    HIP_OK(hipMemcpy(m->io_boards_in.p, boards, static_cast<size_t>(B) * CELLS, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(m->io_turns_in.p, turns, sizeof(int32_t) * B, hipMemcpyHostToDevice));
    az::launch_import(m->game, m->io_boards_in.p, m->io_turns_in.p, m->roots(), B, s);
    m->plain_leaf.ensure(B);
    HIP_OK(hipMemset(m->plain_leaf.sym.p, 0, sizeof(int32_t) * B));
    m->io_policy.ensure(static_cast<size_t>(B) * A); m->io_d.ensure(B); m->io_p1.ensure(B);
    m->io_p2.ensure(B); m->io_ml.ensure(B); m->io_is_term.ensure(B);
    m->ensure_room(static_cast<int64_t>(n_playout) * m->geo.max_edges);
    m->last_select_vl = false;
}
}  // namespace

extern "C" {

// BatchedMCTS::search with RolloutEvaluator (BatchedMCTS.h:339-407, RolloutEvaluator.h:23-48) in the
// REFERENCE'S random stream: per playout, selection on the device; then the playout moves of the
// non-terminal leaves in env order and the root-noise rows of the expansions in env order, both from
// the host mt19937 exactly as the reference (OMP_NUM_THREADS=1) consumes them; expansion and backup on
// the device.  Bit-exact against the reference (fixture G9); one host round trip per playout.
int az_mcts_search_rollout(az_mcts *m, const int8_t *boards, const int32_t *turns, int64_t n, int n_playout)
{
    return guarded([&] {
        hipStream_t s = nullptr;
        rollout_common_begin(m, boards, turns, n, n_playout, s);
        const int B = m->B, A = m->geo.actions;
        const az::SearchParams p = m->params();
        m->io_noise.ensure(static_cast<size_t>(B) * A, true);
        az::EvalIn in{};
        in.policy = m->io_policy.p; in.d = m->io_d.p; in.p1w = m->io_p1.p; in.p2w = m->io_p2.p;
        in.is_term = m->io_is_term.p; in.moves_left = m->io_ml.p; in.sym = nullptr; in.root_noise = m->io_noise.p;
        HIP_OK(hipMemset(m->io_ml.p, 0, sizeof(float) * B));
        std::vector<uint8_t> flags(B), nvalid(B), is_term(B);
        std::vector<uint64_t> bb0(B), bb1(B);
        std::vector<int32_t> turn(B), aux(B);
        std::vector<float> pol(static_cast<size_t>(B) * A), d(B), p1(B), p2(B), noise(static_cast<size_t>(B) * A);
        LeafStore &ls = m->plain_leaf;
        for (int it = 0; it < n_playout; ++it) {
            az::launch_select(m->game, m->arena(), m->roots(), ls.view(), p, 1, false, m->counters.p, s);
            HIP_OK(hipMemcpy(flags.data(), ls.flags.p, B, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(nvalid.data(), ls.nvalid.p, B, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(bb0.data(), ls.bb0.p, sizeof(uint64_t) * B, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(bb1.data(), ls.bb1.p, sizeof(uint64_t) * B, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(turn.data(), ls.turn.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(aux.data(), ls.aux.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
            // phase 2: evaluate_batch over the non-terminal leaves, in order
            for (int i = 0; i < B; ++i) {
                const bool term = (flags[i] & az::LEAF_TERMINAL) != 0;
                int code = (flags[i] >> az::LEAF_RESULT_SHIFT) & 3;
                if (!term) {
                    az::GameState st{bb0[i], bb1[i], turn[i], aux[i]};
                    code = m->game == AZ_GAME_OTHELLO ? host_playout<az::OthelloDev>(st, m->rng)
                                                      : host_playout<az::Connect4Dev>(st, m->rng);
                }
                is_term[i] = term ? 1 : 0;
                d[i] = code == 0 ? 1.0f : 0.0f; p1[i] = code == 1 ? 1.0f : 0.0f; p2[i] = code == 2 ? 1.0f : 0.0f;
                std::fill(pol.begin() + static_cast<size_t>(i) * A, pol.begin() + static_cast<size_t>(i + 1) * A, term ? 0.0f : 1.0f);
            }
            // phase 3: root expansions draw their noise in env order (MCTS.h:347-363)
            bool any_noise = false;
            if (m->cfg.dirichlet_alpha > 0.0f)
                for (int i = 0; i < B; ++i)
                    if ((flags[i] & az::LEAF_ROOT_UNEXPANDED) && !is_term[i]) {
                        m->rng.dirichlet(m->cfg.dirichlet_alpha, &noise[static_cast<size_t>(i) * A], nvalid[i]);
                        any_noise = true;
                    }
            if (any_noise) HIP_OK(hipMemcpy(m->io_noise.p, noise.data(), sizeof(float) * noise.size(), hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(m->io_policy.p, pol.data(), sizeof(float) * pol.size(), hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(m->io_d.p, d.data(), sizeof(float) * B, hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(m->io_p1.p, p1.data(), sizeof(float) * B, hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(m->io_p2.p, p2.data(), sizeof(float) * B, hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(m->io_is_term.p, is_term.data(), B, hipMemcpyHostToDevice));
            az::launch_backprop(m->game, m->arena(), ls.view(), p, 1, false, false, in, m->counters.p, m->err.p, s);
        }
        m->select_launches += n_playout;
        m->backprop_launches += n_playout;
        HIP_OK(hipStreamSynchronize(s));
        m->check_device_error();
    });
}

// The same search with the playouts on the device (k_rollout: one thread per tree, moves and root noise
// from the device generator): no host round trip inside the loop, same distribution, a different stream.
int az_mcts_search_rollout_dev(az_mcts *m, const int8_t *boards, const int32_t *turns, int64_t n, int n_playout)
{
    return guarded([&] {
        hipStream_t s = nullptr;
        rollout_common_begin(m, boards, turns, n, n_playout, s);
        const int B = m->B;
        const az::SearchParams p = m->params();
        az::EvalIn in{};
        in.policy = m->io_policy.p; in.d = m->io_d.p; in.p1w = m->io_p1.p; in.p2w = m->io_p2.p;
        in.is_term = m->io_is_term.p; in.moves_left = m->io_ml.p; in.sym = nullptr; in.root_noise = nullptr;
        for (int it = 0; it < n_playout; ++it) {
            az::launch_select(m->game, m->arena(), m->roots(), m->plain_leaf.view(), p, 1, false, m->counters.p, s);
            az::launch_rollout(m->game, m->plain_leaf.view(), p, B, m->io_policy.p, m->io_d.p, m->io_p1.p, m->io_p2.p,
                               m->io_ml.p, m->io_is_term.p, s);
            az::launch_backprop(m->game, m->arena(), m->plain_leaf.view(), p, 1, false, false, in, m->counters.p, m->err.p, s);
            az::launch_bump_call(m->call_ctr.p, s);
        }
        m->select_launches += n_playout;
        m->backprop_launches += n_playout;
        HIP_OK(hipStreamSynchronize(s));
        m->check_device_error();
    });
}

int az_mcts_get_all_counts(az_mcts *m, int32_t *out)
{
    return guarded([&] {
        HIP_OK(hipSetDevice(m->device));
        m->flush_resets(nullptr);
        const int A = m->geo.actions;
        m->io_counts.ensure(static_cast<size_t>(m->B) * A);
        az::launch_counts(m->game, m->arena(), m->io_counts.p, nullptr);
        HIP_OK(hipMemcpy(out, m->io_counts.p, sizeof(int32_t) * m->B * A, hipMemcpyDeviceToHost));
    });
}

int az_mcts_get_all_root_stats(az_mcts *m, float *out)
{
    return guarded([&] {
        HIP_OK(hipSetDevice(m->device));
        m->flush_resets(nullptr);
        const int STATS = m->geo.stats;
        m->io_stats.ensure(static_cast<size_t>(m->B) * STATS);
        az::launch_root_stats(m->game, m->arena(), m->io_stats.p, nullptr);
        HIP_OK(hipMemcpy(out, m->io_stats.p, sizeof(float) * m->B * STATS, hipMemcpyDeviceToHost));
    });
}

// ---------------------------------------------------------------- device entry points

namespace {
// stream == nullptr with whole_device: everything on the device is waited for (callers that do not
// say which stream their trees are being worked on)
void dev_prepare(az_mcts *m, int K, int64_t sims_per_tree, hipStream_t s, bool whole_device)
{
    require(K >= 1, "dev_prepare: K must be >= 1");
    HIP_OK(hipSetDevice(m->device));
    const size_t total = static_cast<size_t>(m->B) * K;
    const int64_t extra = sims_per_tree * m->geo.max_edges;
    // Reading the trees' fill (`used`), moving the arenas or the leaf buffers, refreshing the tables:
    // all of that must see what the kernels already enqueued have done, and must not pull memory
    // from under them.  Wait for them first - once per many calls (the host-side bound `used_bound`
    // runs ahead of the real fill by at most one call's worth).
    const bool touches = m->any_pending_reset || total > m->vl_leaf.slot.n || static_cast<size_t>(m->B) > m->plain_leaf.slot.n ||
                         !m->tab.p || !m->term_tab.p || m->cfg.c_init != m->tab_c_init || m->cfg.c_base != m->tab_c_base ||
                         m->cfg.score_scale != m->term_tab_scale || m->room_needs_device(extra);
    if (touches) {
        if (whole_device) HIP_OK(hipDeviceSynchronize());
        else HIP_OK(hipStreamSynchronize(s));
    }
    m->flush_resets(s);
    if (total > m->vl_leaf.slot.n) ++m->epoch;
    m->vl_leaf.ensure(total);
    m->plain_leaf.ensure(m->B);
    m->ensure_table();
    m->ensure_room(extra);
}
}  // namespace

int az_mcts_dev_prepare(az_mcts *m, int K, int64_t sims_per_tree)
{
    return guarded([&] { dev_prepare(m, K, sims_per_tree, nullptr, true); });
}

int az_mcts_dev_prepare_stream(az_mcts *m, int K, int64_t sims_per_tree, void *stream)
{
    return guarded([&] { dev_prepare(m, K, sims_per_tree, static_cast<hipStream_t>(stream), false); });
}

int az_mcts_dev_check(az_mcts *m, void *stream)
{
    return guarded([&] {
        const int seen = *static_cast<volatile int *>(m->err_host);
        HIP_OK(hipMemcpyAsync(m->err_host, m->err.p, sizeof(int), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
        if (seen) throw AzError(AZ_ERR_CAPACITY, az_mcts::device_error_text(seen));
    });
}

int az_mcts_dev_replay(az_mcts *m, const int32_t *sym_ids, int64_t sym_stride, int64_t n_select_calls,
                       const float *root_noise)
{
    return guarded([&] {
        require(sym_ids == nullptr || (sym_stride > 0 && n_select_calls > 0), "dev_replay: a symmetry tape needs a stride and a length");
        m->replay_sym = sym_ids;
        m->replay_stride = sym_ids ? sym_stride : 0;
        m->replay_calls = sym_ids ? n_select_calls : 0;
        m->replay_next = 0;
        m->replay_noise = root_noise;
    });
}

int az_mcts_dev_set_roots(az_mcts *m, const uint64_t *bb_p1, const uint64_t *bb_p2,
                          const int32_t *turns, void *stream)
{
    return guarded([&] {
        az::launch_set_roots(m->game, bb_p1, bb_p2, turns, m->roots(), m->B, static_cast<hipStream_t>(stream));
    });
}

int az_mcts_dev_import_roots(az_mcts *m, const int8_t *boards, const int32_t *turns, void *stream)
{
    return guarded([&] {
        az::launch_import(m->game, boards, turns, m->roots(), m->B, static_cast<hipStream_t>(stream));
    });
}

namespace {
// zero_count: an int64 in device memory that the selection launch clears on its way (the
// live-leaf count of az_mcts_dev_search: saves the memset in front of the listing kernel)
void select_and_gather(az_mcts *m, int K, int vl, float *features, uint8_t *valid_mask, void *stream, int64_t *zero_count)
{
    require(K >= 1 && (vl || K == 1), "dev_select: K must be 1 without virtual loss");
    LeafStore &ls = vl ? m->vl_leaf : m->plain_leaf;
    const size_t total = static_cast<size_t>(m->B) * K;
    require(ls.slot.n >= total && m->tab.p, "dev_select: call az_mcts_dev_prepare first");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vl) m->vl_stride = K;
    m->last_select_vl = vl != 0;
    const az::SearchParams p = m->params();
    const bool timed = m->profiling && (m->profile_seen[0]++ % m->profile_stride) == 0 && m->ev_select.begin(s);
    const char *kn = az::launch_select(m->game, m->arena(), m->roots(), ls.view(), p, K, vl != 0, m->counters.p, s, m->call_ctr.p, zero_count);
    if (timed) { m->ev_select.end(s); m->timed_select_kernel = kn; }
    bool gen_sym = true;
    if (m->replay_sym != nullptr) {             // recorded symmetry ids instead of the generator's
        if (m->replay_next >= m->replay_calls || static_cast<int64_t>(total) > m->replay_stride)
            throw AzError(AZ_ERR_STATE, "dev_select: the replay tape (az_mcts_dev_replay) is exhausted or too narrow");
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            throw AzError(AZ_ERR_STATE, "dev_select: a replay tape cannot be captured into a graph (its position moves per call)");
        HIP_OK(hipMemcpyAsync(ls.sym.p, m->replay_sym + m->replay_next * m->replay_stride, sizeof(int32_t) * total,
                              hipMemcpyDeviceToDevice, s));
        ++m->replay_next;
        gen_sym = false;
    }
    az::launch_export(m->game, ls.view(), p, static_cast<int>(total), gen_sym, nullptr, valid_mask, features, s);
    ++m->select_launches;
}
}  // namespace

namespace {
// The native loop's form of the same step: selection, then - instead of the gather into feature planes - the
// leaves' symmetry ids, action masks and (unless the table's lookup builds it) the compact list of leaves to
// evaluate, for an evaluator that reads the leaf positions themselves (az_nn_model_forward_positions).
void select_and_prep(az_mcts *m, int K, int vl, uint8_t *valid_mask, int32_t *rows, int64_t *n_rows, void *stream)
{
    require(K >= 1 && (vl || K == 1), "dev_select: K must be 1 without virtual loss");
    LeafStore &ls = vl ? m->vl_leaf : m->plain_leaf;
    const size_t total = static_cast<size_t>(m->B) * K;
    require(ls.slot.n >= total && m->tab.p, "dev_select: call az_mcts_dev_prepare first");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vl) m->vl_stride = K;
    m->last_select_vl = vl != 0;
    const az::SearchParams p = m->params();
    const bool timed = m->profiling && (m->profile_seen[0]++ % m->profile_stride) == 0 && m->ev_select.begin(s);
    const char *kn = az::launch_select(m->game, m->arena(), m->roots(), ls.view(), p, K, vl != 0, m->counters.p, s, m->call_ctr.p, n_rows);
    if (timed) { m->ev_select.end(s); m->timed_select_kernel = kn; }
    bool gen_sym = true;
    if (m->replay_sym != nullptr) {             // recorded symmetry ids instead of the generator's
        if (m->replay_next >= m->replay_calls || static_cast<int64_t>(total) > m->replay_stride)
            throw AzError(AZ_ERR_STATE, "dev_select: the replay tape (az_mcts_dev_replay) is exhausted or too narrow");
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            throw AzError(AZ_ERR_STATE, "dev_select: a replay tape cannot be captured into a graph (its position moves per call)");
        HIP_OK(hipMemcpyAsync(ls.sym.p, m->replay_sym + m->replay_next * m->replay_stride, sizeof(int32_t) * total,
                              hipMemcpyDeviceToDevice, s));
        ++m->replay_next;
        gen_sym = false;
    }
    az::launch_leaf_prep(m->game, ls.view(), p, static_cast<int>(total), gen_sym, valid_mask, rows, n_rows, m->err.p, s);
    ++m->select_launches;
}
}  // namespace

int az_mcts_dev_select(az_mcts *m, int K, int vl, float *features, uint8_t *valid_mask, void *stream)
{
    return guarded([&] { select_and_gather(m, K, vl, features, valid_mask, stream, nullptr); });
}

int az_mcts_dev_backprop(az_mcts *m, int K, int vl, const float *probs, const float *wdl_rel,
                         const float *moves_left, void *stream)
{
    return guarded([&] {
        require(K >= 1 && (vl || K == 1), "dev_backprop: K must be 1 without virtual loss");
        if (vl) require(m->vl_stride == K, "dev_backprop: K differs from the preceding dev_select");
        LeafStore &ls = vl ? m->vl_leaf : m->plain_leaf;
        az::EvalIn in{};
        in.policy = probs; in.wdl_rel = wdl_rel; in.moves_left = moves_left;
        in.root_noise = m->replay_noise; in.sym = nullptr;
        hipStream_t s = static_cast<hipStream_t>(stream);
        const bool timed = m->profiling && (m->profile_seen[1]++ % m->profile_stride) == 0 && m->ev_backprop.begin(s);
        az::launch_backprop(m->game, m->arena(), ls.view(), m->params(), K, vl != 0, true, in, m->counters.p,
                            m->err.p, s);
        if (timed) m->ev_backprop.end(s);
        ++m->backprop_launches;
    });
}

int az_mcts_dev_set_noise_epsilons(az_mcts *m, const float *per_tree)
{
    return guarded([&] { m->noise_eps_tree = per_tree; });
}

int az_mcts_dev_live_leaves(az_mcts *m, int K, int32_t *leaf_idx, int64_t *leaf_count, void *stream)
{
    return guarded([&] {
        LeafStore &ls = m->last_select_vl ? m->vl_leaf : m->plain_leaf;
        const size_t total = static_cast<size_t>(m->B) * K;
        require(K >= 1 && ls.slot.n >= total, "dev_live_leaves: no selection of that width");
        az::launch_live_leaves(ls.view(), static_cast<int>(total), leaf_idx, leaf_count, m->err.p, static_cast<hipStream_t>(stream));
    });
}

// The reference's iteration schedule (MCTS_cpp.py:110-113, 217-264: one plain simulation that
// expands every root, then ceil((n_playout-1)/K) virtual-loss batches) with the evaluator in the
// loop, issued from native code: per iteration selection + gather, the list of leaves to evaluate,
// the six evaluator launches, backup.  Nothing here waits for the device once the buffers exist.
namespace {
// warmup: the schedule of a whole search (one plain simulation first, MCTS_cpp.py:217-248); without it the call
// CONTINUES a search: n_playout more simulations in virtual-loss batches of K (plain ones for K <= 1)
int dev_search_impl(az_mcts *m, const az_nn_model *model, int n_playout, int K, int use_table, bool warmup, void *stream)
{
    return guarded([&] {
        require(model != nullptr, "dev_search: no evaluator model");
        const int kind = az_nn_model_kind(model);
        require(kind == (m->game == AZ_GAME_CONNECT4 ? AZ_NN_KIND_HASH_CONNECT4 : AZ_NN_KIND_HASH_OTHELLO) ||
                    (kind == AZ_NN_KIND_CONNECT4_CNN && m->game == AZ_GAME_CONNECT4) ||
                    (kind == AZ_NN_KIND_OTHELLO_CNN && m->game == AZ_GAME_OTHELLO),
                "dev_search: the evaluator model does not belong to this engine's game");
        require(K >= 1 && n_playout >= 0, "dev_search: K must be >= 1 and n_playout >= 0");
        require(!use_table || m->tt_entries.p != nullptr, "dev_search: no table (az_mcts_dev_tt_create)");
        hipStream_t s = static_cast<hipStream_t>(stream);
        HIP_OK(hipSetDevice(m->device));
        const size_t total = static_cast<size_t>(m->B) * K;
        const size_t scratch = az_nn_model_scratch_bytes(model, static_cast<int64_t>(total));
        const int64_t extra = static_cast<int64_t>(n_playout) * m->geo.max_edges;
        const bool grows = total > m->vl_leaf.slot.n || total > m->ev_rows.n || scratch > m->ev_scratch.n ||
                           m->room_needs_device(extra) || (use_table && m->tt_keys.n < 2 * total);
        // anything below that allocates, frees or reads a buffer the stream's kernels use waits for them first
        if (grows) HIP_OK(hipStreamSynchronize(s));
        dev_prepare(m, K, n_playout, s, false);
        if (total > m->ev_rows.n) {
            m->ev_feat.ensure(total * 3 * m->geo.cells); m->ev_mask.ensure(total * m->geo.actions);
            m->ev_probs.ensure(total * m->geo.actions); m->ev_wdl.ensure(total * 3); m->ev_ml.ensure(total);
            m->ev_rows.ensure(total); m->ev_nrows.ensure(1, true);
        }
        m->ev_scratch.ensure(scratch);
        if (use_table && m->tt_keys.n < 2 * total) { m->tt_keys.ensure(2 * total); ++m->epoch; }

        auto ok = [&](int rc, const char *what) {
            if (rc != AZ_OK) throw AzError(AZ_ERR_DEVICE, std::string("dev_search: ") + what + ": " + g_last_error);
        };
        // AZ_SEARCH_FEATURES=1: the first form of the loop - leaves gathered into feature planes (k_export), the
        // list from k_live_leaves - kept for A/B runs; default: the evaluator reads the leaf positions
        static const bool via_features = getenv("AZ_SEARCH_FEATURES") != nullptr && getenv("AZ_SEARCH_FEATURES")[0] == '1';
        auto iteration = [&](int k, int vl) {
            const int64_t n = static_cast<int64_t>(m->B) * k;
            LeafStore &ls = vl ? m->vl_leaf : m->plain_leaf;
            if (via_features) {
                select_and_gather(m, k, vl, m->ev_feat.p, m->ev_mask.p, stream, use_table ? nullptr : m->ev_nrows.p);
                if (!use_table)
                    az::launch_live_leaves(ls.view(), static_cast<int>(n), m->ev_rows.p, m->ev_nrows.p, m->err.p, s, false);
            } else {
                // the selection launch clears the count; with the table its lookup builds the list instead
                select_and_prep(m, k, vl, m->ev_mask.p, use_table ? nullptr : m->ev_rows.p, use_table ? nullptr : m->ev_nrows.p, stream);
            }
            if (use_table)
                ok(az_mcts_dev_tt_lookup(m, k, m->ev_probs.p, m->ev_wdl.p, m->ev_ml.p, m->ev_rows.p, m->ev_nrows.p, stream), "tt_lookup");
            int rc;
            if (via_features) {
                rc = az_nn_model_forward(model, m->ev_feat.p, m->ev_mask.p, m->ev_probs.p, m->ev_wdl.p, m->ev_ml.p, n,
                                         m->ev_rows.p, m->ev_nrows.p, m->ev_scratch.p, m->ev_scratch.n, stream);
            } else {
                const az_nn_positions pos{ls.bb0.p, ls.bb1.p, ls.turn.p, ls.sym.p};
                rc = az_nn_model_forward_positions(model, &pos, m->ev_mask.p, m->ev_probs.p, m->ev_wdl.p, m->ev_ml.p, n,
                                                   m->ev_rows.p, m->ev_nrows.p, m->ev_scratch.p, m->ev_scratch.n, stream);
            }
            if (rc != 0) throw AzError(AZ_ERR_ARG, "dev_search: the evaluator model refused its arguments");
            if (use_table)
                ok(az_mcts_dev_tt_insert(m, k, m->ev_rows.p, m->ev_nrows.p, m->ev_probs.p, m->ev_wdl.p, m->ev_ml.p, stream), "tt_insert");
            ok(az_mcts_dev_backprop(m, k, vl, m->ev_probs.p, m->ev_wdl.p, m->ev_ml.p, stream), "backprop");
        };
        int remaining = n_playout;
        if (K <= 1) {
            for (; remaining > 0; --remaining) iteration(1, 0);
            return;
        }
        if (warmup && remaining > 0) { iteration(1, 0); --remaining; }
        while (remaining > 0) {
            const int k = std::min(K, remaining);
            remaining -= k;
            iteration(k, 1);
        }
    });
}
}  // namespace

int az_mcts_dev_search(az_mcts *m, const az_nn_model *model, int n_playout, int K, int use_table, void *stream)
{
    return dev_search_impl(m, model, n_playout, K, use_table, true, stream);
}

int az_mcts_dev_search_more(az_mcts *m, const az_nn_model *model, int n_sims, int K, int use_table, void *stream)
{
    return dev_search_impl(m, model, n_sims, K, use_table, false, stream);
}

// ---- device transposition table ------------------------------------------------------------
int az_mcts_dev_tt_create(az_mcts *m, int log2_entries)
{
    return guarded([&] {
        require(log2_entries >= 2 && log2_entries <= 28, "dev_tt_create: log2_entries must be in [2, 28]");
        HIP_OK(hipSetDevice(m->device));
        HIP_OK(hipDeviceSynchronize());
        const size_t n = (static_cast<size_t>(1) << log2_entries) * az::tt_entry_bytes(m->game);
        if (n != m->tt_entries.n) {
            if (m->tt_entries.p) { HIP_OK(hipFree(m->tt_entries.p)); m->tt_entries.p = nullptr; m->tt_entries.n = 0; }
            m->tt_entries.ensure(n);
            ++m->epoch;
        }
        HIP_OK(hipMemset(m->tt_entries.p, 0, n));
        m->tt_stats.ensure(4, true);
        HIP_OK(hipMemset(m->tt_stats.p, 0, 4 * sizeof(unsigned long long)));
        m->tt_mask = (static_cast<uint64_t>(1) << log2_entries) - 1;
    });
}

int az_mcts_dev_tt_clear(az_mcts *m, void *stream)
{
    return guarded([&] {
        require(m->tt_entries.p != nullptr, "dev_tt_clear: no table (az_mcts_dev_tt_create)");
        HIP_OK(hipMemsetAsync(m->tt_entries.p, 0, m->tt_entries.n, static_cast<hipStream_t>(stream)));
    });
}

int az_mcts_dev_tt_lookup(az_mcts *m, int K, float *probs, float *wdl_rel, float *moves_left, int32_t *miss_idx,
                          int64_t *miss_count, void *stream)
{
    return guarded([&] {
        require(m->tt_entries.p != nullptr, "dev_tt_lookup: no table (az_mcts_dev_tt_create)");
        LeafStore &ls = m->last_select_vl ? m->vl_leaf : m->plain_leaf;
        const size_t total = static_cast<size_t>(m->B) * K;
        require(K >= 1 && ls.slot.n >= total, "dev_tt_lookup: no selection of that width");
        if (m->tt_keys.n < 2 * total) {
            HIP_OK(hipDeviceSynchronize());
            m->tt_keys.ensure(2 * total);
            ++m->epoch;
        }
        az::TtTable t{m->tt_entries.p, m->tt_mask, m->tt_stats.p};
        az::launch_tt_lookup(m->game, ls.view(), static_cast<int>(total), t, m->call_ctr.p, probs, wdl_rel, moves_left, miss_idx,
                             miss_count, m->tt_keys.p, m->err.p, static_cast<hipStream_t>(stream));
    });
}

int az_mcts_dev_tt_insert(az_mcts *m, int K, const int32_t *miss_idx, const int64_t *miss_count, const float *probs,
                          const float *wdl_rel, const float *moves_left, void *stream)
{
    return guarded([&] {
        require(m->tt_entries.p != nullptr, "dev_tt_insert: no table (az_mcts_dev_tt_create)");
        const size_t total = static_cast<size_t>(m->B) * K;
        require(K >= 1 && m->tt_keys.n >= 2 * total, "dev_tt_insert: call az_mcts_dev_tt_lookup on this selection first");
        az::TtTable t{m->tt_entries.p, m->tt_mask, m->tt_stats.p};
        az::launch_tt_insert(m->game, static_cast<int>(total), t, m->call_ctr.p, miss_idx, miss_count, m->tt_keys.p, probs, wdl_rel,
                             moves_left, static_cast<hipStream_t>(stream));
    });
}

int az_mcts_dev_tt_refresh(az_mcts *m, const az_nn_model *model, void *stream)
{
    return guarded([&] {
        require(m->tt_entries.p != nullptr, "dev_tt_refresh: no table (az_mcts_dev_tt_create)");
        require(model != nullptr, "dev_tt_refresh: no evaluator model");
        const int kind = az_nn_model_kind(model);
        require(kind == (m->game == AZ_GAME_CONNECT4 ? AZ_NN_KIND_HASH_CONNECT4 : AZ_NN_KIND_HASH_OTHELLO) ||
                    (kind == AZ_NN_KIND_CONNECT4_CNN && m->game == AZ_GAME_CONNECT4) ||
                    (kind == AZ_NN_KIND_OTHELLO_CNN && m->game == AZ_GAME_OTHELLO),
                "dev_tt_refresh: the evaluator model does not belong to this engine's game");
        HIP_OK(hipSetDevice(m->device));
        hipStream_t s = static_cast<hipStream_t>(stream);
        const int64_t chunk = 16384;
        const int A = m->geo.actions;
        const size_t scratch = az_nn_model_scratch_bytes(model, chunk);
        if (m->rf_rows.n < static_cast<size_t>(chunk) || m->rf_scratch.n < scratch) {
            HIP_OK(hipStreamSynchronize(s));
            m->rf_bb0.ensure(chunk); m->rf_bb1.ensure(chunk); m->rf_turn.ensure(chunk); m->rf_sym.ensure(chunk);
            m->rf_mask.ensure(chunk * A); m->rf_probs.ensure(chunk * A);
            m->rf_wdl.ensure(chunk * 3); m->rf_ml.ensure(chunk); m->rf_rows.ensure(chunk); m->rf_keys.ensure(2 * chunk);
            m->rf_count.ensure(1, true); m->rf_scratch.ensure(scratch);
        }
        az::TtTable t{m->tt_entries.p, m->tt_mask, m->tt_stats.p};
        const int64_t entries = static_cast<int64_t>(m->tt_mask) + 1;
        const az_nn_positions pos{m->rf_bb0.p, m->rf_bb1.p, m->rf_turn.p, m->rf_sym.p};
        for (int64_t e0 = 0; e0 < entries; e0 += chunk) {
            const int n = static_cast<int>(std::min<int64_t>(chunk, entries - e0));
            az::launch_tt_refresh_gather(m->game, t, static_cast<uint64_t>(e0), n, m->rf_bb0.p, m->rf_bb1.p, m->rf_turn.p, m->rf_sym.p,
                                         m->rf_mask.p, m->rf_rows.p, m->rf_count.p, m->rf_keys.p, s);
            if (az_nn_model_forward_positions(model, &pos, m->rf_mask.p, m->rf_probs.p, m->rf_wdl.p, m->rf_ml.p, n, m->rf_rows.p,
                                              m->rf_count.p, m->rf_scratch.p, m->rf_scratch.n, stream) != 0)
                throw AzError(AZ_ERR_ARG, "dev_tt_refresh: the evaluator model refused its arguments");
            az::launch_tt_refresh_store(m->game, t, static_cast<uint64_t>(e0), n, m->rf_rows.p, m->rf_count.p, m->rf_keys.p, m->rf_probs.p,
                                        m->rf_wdl.p, m->rf_ml.p, s);
        }
    });
}

int az_mcts_dev_tt_stats(az_mcts *m, int64_t out[4])
{
    return guarded([&] {
        require(m->tt_entries.p != nullptr && out != nullptr, "dev_tt_stats: no table (az_mcts_dev_tt_create)");
        HIP_OK(hipSetDevice(m->device));
        HIP_OK(hipDeviceSynchronize());
        unsigned long long h[4];
        HIP_OK(hipMemcpy(h, m->tt_stats.p, sizeof(h), hipMemcpyDeviceToHost));
        for (int i = 0; i < 4; ++i) out[i] = static_cast<int64_t>(h[i]);
    });
}

int az_mcts_dev_leaves(az_mcts *m, int K, uint64_t *bb_p1, uint64_t *bb_p2, int32_t *turns,
                       uint8_t *flags, void *stream)
{
    return guarded([&] {
        LeafStore &ls = m->last_select_vl ? m->vl_leaf : m->plain_leaf;
        const size_t total = static_cast<size_t>(m->B) * K;
        require(ls.slot.n >= total, "dev_leaves: no selection of that width");
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (bb_p1) HIP_OK(hipMemcpyAsync(bb_p1, ls.bb0.p, 8 * total, hipMemcpyDeviceToDevice, s));
        if (bb_p2) HIP_OK(hipMemcpyAsync(bb_p2, ls.bb1.p, 8 * total, hipMemcpyDeviceToDevice, s));
        if (turns) HIP_OK(hipMemcpyAsync(turns, ls.turn.p, 4 * total, hipMemcpyDeviceToDevice, s));
        if (flags) HIP_OK(hipMemcpyAsync(flags, ls.flags.p, total, hipMemcpyDeviceToDevice, s));
    });
}

int az_mcts_dev_leaf_syms(az_mcts *m, int K, int32_t *sym_ids, void *stream)
{
    return guarded([&] {
        LeafStore &ls = m->last_select_vl ? m->vl_leaf : m->plain_leaf;
        const size_t total = static_cast<size_t>(m->B) * K;
        require(ls.slot.n >= total && sym_ids != nullptr, "dev_leaf_syms: no selection of that width");
        HIP_OK(hipMemcpyAsync(sym_ids, ls.sym.p, 4 * total, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    });
}

int az_mcts_dev_counts(az_mcts *m, int32_t *counts, void *stream)
{
    return guarded([&] { az::launch_counts(m->game, m->arena(), counts, static_cast<hipStream_t>(stream)); });
}

int az_mcts_dev_root_stats(az_mcts *m, float *stats, void *stream)
{
    return guarded([&] { az::launch_root_stats(m->game, m->arena(), stats, static_cast<hipStream_t>(stream)); });
}

int az_mcts_dev_prune_roots(az_mcts *m, const int32_t *actions, void *stream)
{
    return guarded([&] {
        hipStream_t s = static_cast<hipStream_t>(stream);
        HIP_OK(hipSetDevice(m->device));
        m->prune_on(actions, nullptr, true, m->replay_noise, s);
        az::launch_bump_call(m->call_ctr.p, s);
    });
}

int az_mcts_dev_reset_masked(az_mcts *m, const uint8_t *mask, void *stream)
{
    return guarded([&] { az::launch_reset_masked(m->arena(), mask, static_cast<hipStream_t>(stream)); });
}

// ---------------------------------------------------------------- capacity / instrumentation

int az_mcts_reserve(az_mcts *m, int64_t slots_per_tree)
{
    return guarded([&] {
        HIP_OK(hipSetDevice(m->device));
        if (slots_per_tree > m->S) m->grow(slots_per_tree);
    });
}

int64_t az_mcts_capacity(const az_mcts *m) { return m->S; }
int64_t az_mcts_epoch(const az_mcts *m) { return m->epoch; }

int az_c4_dev_step(uint64_t *bb_p1, uint64_t *bb_p2, int32_t *turns, const int32_t *actions,
                   uint8_t *done, int32_t *winner, int64_t n, int reset_finished, void *stream)
{
    return guarded([&] {
        require(n >= 0, "az_c4_dev_step: negative size");
        if (n == 0) return;
        az::launch_game_step(AZ_GAME_CONNECT4, bb_p1, bb_p2, turns, nullptr, actions, done, winner, n, reset_finished != 0,
                             static_cast<hipStream_t>(stream));
    });
}

int az_game_dev_step(int game, uint64_t *bb_p1, uint64_t *bb_p2, int32_t *turns, int32_t *aux, const int32_t *actions,
                     uint8_t *done, int32_t *winner, int64_t n, int reset_finished, void *stream)
{
    return guarded([&] {
        require(game == AZ_GAME_CONNECT4 || game == AZ_GAME_OTHELLO, "az_game_dev_step: unknown game");
        require(n >= 0, "az_game_dev_step: negative size");
        if (n == 0) return;
        az::launch_game_step(game, bb_p1, bb_p2, turns, aux, actions, done, winner, n, reset_finished != 0,
                             static_cast<hipStream_t>(stream));
    });
}

int az_game_dev_valid_mask(int game, const uint64_t *bb_p1, const uint64_t *bb_p2, const int32_t *turns,
                           const int32_t *aux, uint8_t *mask, int64_t n, void *stream)
{
    return guarded([&] {
        require(game == AZ_GAME_CONNECT4 || game == AZ_GAME_OTHELLO, "az_game_dev_valid_mask: unknown game");
        require(n >= 0, "az_game_dev_valid_mask: negative size");
        if (n == 0) return;
        az::launch_game_valid_mask(game, bb_p1, bb_p2, turns, aux, mask, n, static_cast<hipStream_t>(stream));
    });
}

int az_mcts_max_used(az_mcts *m, int64_t *out)
{
    return guarded([&] {
        HIP_OK(hipSetDevice(m->device));
        m->flush_resets(nullptr);
        HIP_OK(hipDeviceSynchronize());
        *out = m->true_max_used();
        m->used_bound = *out;
    });
}

int az_mcts_counters(az_mcts *m, int64_t out[AZ_NUM_COUNTERS])
{
    return guarded([&] {
        HIP_OK(hipSetDevice(m->device));
        unsigned long long h[az::CNT_N * az::CNT_STRIPES];
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(h, m->counters.p, sizeof h, hipMemcpyDeviceToHost));
        for (int i = 0; i < az::CNT_N; ++i) {
            unsigned long long sum = 0;
            for (int st = 0; st < az::CNT_STRIPES; ++st) sum += h[st * az::CNT_N + i];
            out[i] = static_cast<int64_t>(sum);
        }
        out[az::CNT_SELECT_LAUNCHES] = m->select_launches;
        out[az::CNT_BACKPROP_LAUNCHES] = m->backprop_launches;
        m->check_device_error();
    });
}

int az_mcts_counters_reset(az_mcts *m)
{
    return guarded([&] {
        HIP_OK(hipSetDevice(m->device));
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemset(m->counters.p, 0, sizeof(unsigned long long) * az::CNT_N * az::CNT_STRIPES));
        m->select_launches = m->backprop_launches = 0;
    });
}

int az_mcts_profile(az_mcts *m, int enable)
{
    return guarded([&] {
        HIP_OK(hipSetDevice(m->device));
        if (enable) {
            m->ev_select.allocate(AZ_PROFILE_MAX);
            m->ev_backprop.allocate(AZ_PROFILE_MAX);
        }
        m->profiling = enable != 0;
        m->profile_stride = enable > 1 ? enable : 1;
        m->profile_seen[0] = m->profile_seen[1] = 0;
    });
}

const char *az_mcts_timed_select_kernel(az_mcts *m) { return m ? m->timed_select_kernel : ""; }

int az_mcts_profile_read(az_mcts *m, double out_ms[2], int64_t out_launches[2])
{
    return guarded([&] {
        HIP_OK(hipSetDevice(m->device));
        HIP_OK(hipDeviceSynchronize());
        m->ev_select.read(out_ms[0], out_launches[0]);
        m->ev_backprop.read(out_ms[1], out_launches[1]);
    });
}

int az_rng_gamma_selftest(uint32_t seed, float alpha, int count, float *out)
{
    az::HostRng r;
    r.seed(seed);
    r.gamma_fill(alpha, out, count);
    return AZ_OK;
}

}  // extern "C"
