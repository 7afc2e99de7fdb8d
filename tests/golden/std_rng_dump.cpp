// Draws from the REAL libstdc++ <random> classes the reference uses (MCTS.h:13-17,119-128,
// 352-358; BatchedMCTS.h:33-40; RolloutEvaluator.h:44-46), compiled with the reference's
// own flags (setup.py:24-27).  Output: raw little-endian binary on stdout, parsed by
// make_golden.py into tests/golden/rng_std.npz.  This is our own harness, not reference code.
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

static void put(const void *p, size_t n) { fwrite(p, 1, n, stdout); }

int main()
{
    const unsigned seeds[3] = {0u, 1234u, 20071u};
    const float alphas[4] = {0.3f, 0.03f, 1.0f, 2.5f};
    for (unsigned s : seeds) {
        std::mt19937 g(s);
        for (int i = 0; i < 16; ++i) { uint32_t v = (uint32_t)g(); put(&v, 4); }
        for (int i = 0; i < 1024; ++i) { int32_t v = std::uniform_int_distribution<int>(0, 1)(g); put(&v, 4); }
        for (int i = 0; i < 1024; ++i) {
            int hi = 1 + i % 7;
            int32_t v = std::uniform_int_distribution<int>(0, hi - 1)(g); put(&v, 4);
        }
        for (float a : alphas)
            for (int grp = 0; grp < 200; ++grp) {
                std::gamma_distribution<float> gd(a, 1.0f);   // fresh object per group
                int cnt = 1 + grp % 7;
                for (int i = 0; i < 7; ++i) { float v = (i < cnt) ? gd(g) : 0.0f; put(&v, 4); }
            }
        uint32_t tail = (uint32_t)g(); put(&tail, 4);   // stream position check
    }
    return 0;
}
