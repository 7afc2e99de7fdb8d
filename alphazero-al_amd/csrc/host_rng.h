// host_rng.h - the host-side random stream of the engine's "reference RNG" mode.
//
// The reference draws symmetry ids and Dirichlet noise from a thread-local std::mt19937
// (MCTS.h:13-17) through std::uniform_int_distribution<int> (BatchedMCTS.h:33-40) and a fresh
// std::gamma_distribution<float>(alpha, 1) per draw site (MCTS.h:119-128, 352-358).  With one
// OpenMP thread all draws come from one engine in env order; this class reproduces that
// stream bit for bit so that `set_seed(s)` means the same thing it means in the reference.
//
// The distributions are written out (GCC 11 libstdc++: bits/random.tcc:1802-1836, 2337-2392,
// 3348-3385; bits/uniform_int_dist.h:243-305) with the fused multiply-adds that g++ -O3 emits
// for them on an FMA machine made explicit, so the values do not depend on which compiler or
// contraction mode builds this file.  Checked against the real std:: classes by
// tests/golden/rng_std.npz through az_rng_gamma_selftest().
#pragma once

#include <cmath>
#include <cstdint>
#include <random>

namespace az {

class HostRng {
public:
    HostRng() { seed_random(); }

    void seed(uint32_t s)
    {
        mt_[0] = s;
        for (int i = 1; i < N; ++i)
            mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + static_cast<uint32_t>(i);
        idx_ = N;
    }

    void seed_random() { seed(std::random_device{}()); }

    uint32_t next()
    {
        if (idx_ >= N) twist();
        uint32_t y = mt_[idx_++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }

    // uniform_int_distribution<int>(0, hi)(rng), hi >= 0: Lemire's nearly divisionless method
    int uniform_int(int hi)
    {
        const uint32_t range = static_cast<uint32_t>(hi) + 1u;
        uint64_t product = static_cast<uint64_t>(next()) * range;
        uint32_t low = static_cast<uint32_t>(product);
        if (low < range) {
            const uint32_t threshold = (0u - range) % range;
            while (low < threshold) {
                product = static_cast<uint64_t>(next()) * range;
                low = static_cast<uint32_t>(product);
            }
        }
        return static_cast<int>(product >> 32);
    }

    // `count` draws from ONE gamma_distribution<float>(alpha, 1) object (its inner normal
    // distribution caches a variate, so the object boundary matters).
    void gamma_fill(float alpha, float *out, int count)
    {
        bool have_saved = false;
        float saved = 0.0f;
        const float malpha = (static_cast<double>(alpha) < 1.0) ? alpha + 1.0f : alpha;
        const float a1 = malpha - 1.0f / 3.0f;
        const float a2 = 1.0f / std::sqrt(9.0f * a1);
        for (int i = 0; i < count; ++i) {
            float u, v, n;
            for (;;) {
                do {
                    n = normal(have_saved, saved);
                    v = std::fma(a2, n, 1.0f);
                } while (static_cast<double>(v) <= 0.0);
                v = v * v * v;
                u = canonical();
                const double dn = n;
                const double squeeze = std::fma(-(((0.0331 * dn) * dn) * dn), dn, 1.0);
                if (!(static_cast<double>(u) > squeeze)) break;
                const double rhs = std::fma(0.5 * dn, dn,
                    static_cast<double>(a1) * ((1.0 - static_cast<double>(v)) +
                                               static_cast<double>(std::log(v))));
                if (!(static_cast<double>(std::log(u)) > rhs)) break;
            }
            if (alpha == malpha) {
                out[i] = a1 * v * 1.0f;
            } else {
                do { u = canonical(); } while (u == 0.0f);
                out[i] = std::pow(u, 1.0f / alpha) * a1 * v * 1.0f;
            }
        }
    }

    // Dirichlet(alpha) noise over `count` edges exactly as MCTS.h:122-131 / 354-362 normalise
    // it: sum in edge order, inv = 1/(sum + 1e-8f), multiply.
    void dirichlet(float alpha, float *out, int count)
    {
        gamma_fill(alpha, out, count);
        float sum = 0.0f;
        for (int i = 0; i < count; ++i) sum += out[i];
        const float inv = 1.0f / (sum + 1e-8f);
        for (int i = 0; i < count; ++i) out[i] = out[i] * inv;
    }

private:
    static constexpr int N = 624;
    uint32_t mt_[N];
    int idx_ = N;

    void twist()
    {
        for (int i = 0; i < N; ++i) {
            const uint32_t y = (mt_[i] & 0x80000000u) | (mt_[(i + 1) % N] & 0x7fffffffu);
            uint32_t v = mt_[(i + 397) % N] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            mt_[i] = v;
        }
        idx_ = 0;
    }

    float canonical()
    {
        const float sum = static_cast<float>(next()) * 1.0f;
        float ret = sum / 4294967296.0f;
        if (ret >= 1.0f) ret = std::nextafter(1.0f, 0.0f);
        return ret;
    }

    float normal(bool &have_saved, float &saved)
    {
        if (have_saved) {
            have_saved = false;
            return saved;
        }
        float x, y, r2;
        do {
            x = static_cast<float>(static_cast<double>(2.0f * canonical()) - 1.0);
            y = static_cast<float>(static_cast<double>(2.0f * canonical()) - 1.0);
            r2 = std::fma(x, x, y * y);
        } while (r2 > 1.0f || r2 == 0.0f);
        const float mult = std::sqrt(-2.0f * std::log(r2) / r2);
        saved = x * mult;
        have_saved = true;
        return y * mult;
    }
};

}  // namespace az
