/*
 * rng_oracle.c - restatement of the libstdc++-11 <random> pieces the reference draws from.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference uses (MCTS.h:13-17,119-128,352-358; BatchedMCTS.h:33-40):
 *   std::mt19937, std::uniform_int_distribution<int>, std::gamma_distribution<float>.
 * Those live in the host toolchain (GCC 11.4 libstdc++), not in the reference tree, so
 * their published algorithms are restated here from /usr/include/c++/11/bits/random.tcc
 * and bits/uniform_int_dist.h.  g++ -O3 on an FMA machine contracts four expressions in
 * them (seen in the disassembly of the compiled reference); those are written as explicit
 * fma()/fmaf() here because this file is compiled with -ffp-contract=off.
 * Pinned by tests/golden/rng_std.npz (drawn from the real std:: classes with the
 * reference's compile flags) and by golden set G4.
 */
#include "oracle.h"

#include <math.h>

/* ---- std::mt19937 (mersenne_twister_engine<uint_fast32_t,32,624,397,31,...>) ---- */

void orc_mt_seed(orc_mt19937 *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static void mt_twist(orc_mt19937 *g)
{
    for (int i = 0; i < 624; ++i) {
        uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
        uint32_t v = g->mt[(i + 397) % 624] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        g->mt[i] = v;
    }
    g->idx = 0;
}

uint32_t orc_mt_next(orc_mt19937 *g)
{
    if (g->idx >= 624) mt_twist(g);
    uint32_t y = g->mt[g->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* uniform_int_dist.h:243-271 (_S_nd, Lemire) reached through :296-305 because mt19937's
 * range is exactly 32 bits. */
int orc_uniform_int(orc_mt19937 *g, int a, int b)
{
    uint32_t urange = (uint32_t)b - (uint32_t)a;
    if (urange == 0xffffffffu) return (int)(orc_mt_next(g) + (uint32_t)a);
    uint32_t range = urange + 1u;
    uint64_t product = (uint64_t)orc_mt_next(g) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = (uint64_t)orc_mt_next(g) * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (int)((uint32_t)(product >> 32) + (uint32_t)a);
}

/* random.tcc:3348-3385 with _RealType=float, bits=24: one 32-bit draw, /2^32, clamp <1 */
float orc_canonical_float(orc_mt19937 *g)
{
    float sum = (float)orc_mt_next(g) * 1.0f;
    float ret = sum / 4294967296.0f;
    if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
    return ret;
}

/* random.tcc:1802-1836 (Marsaglia polar); mean 0, stddev 1 */
float orc_normal_float(orc_mt19937 *g, orc_normal_state *st)
{
    float ret;
    if (st->saved_available) {
        st->saved_available = 0;
        ret = st->saved;
    } else {
        float x, y, r2;
        do {
            x = (float)((double)(2.0f * orc_canonical_float(g)) - 1.0);
            y = (float)((double)(2.0f * orc_canonical_float(g)) - 1.0);
            r2 = fmaf(x, x, y * y); /* g++ contracts x*x + y*y this way */
        } while (r2 > 1.0f || r2 == 0.0f);
        float mult = sqrtf(-2.0f * logf(r2) / r2);
        st->saved = x * mult;
        st->saved_available = 1;
        ret = y * mult;
    }
    return ret * 1.0f + 0.0f;
}

/* random.tcc:2337-2392 (Marsaglia-Tsang), beta = 1.  One distribution object per call of
 * this function, because the reference constructs a fresh one at each site and its inner
 * normal_distribution caches a variate (MCTS.h:120,353). */
void orc_gamma_fill(orc_mt19937 *g, float alpha, float *out, int count)
{
    orc_normal_state nd = {0.0f, 0};
    const float malpha = ((double)alpha < 1.0) ? alpha + 1.0f : alpha;
    const float a1 = malpha - 1.0f / 3.0f;
    const float a2 = 1.0f / sqrtf(9.0f * a1);

    for (int i = 0; i < count; ++i) {
        float u, v, n;
        for (;;) {
            do {
                n = orc_normal_float(g, &nd);
                v = fmaf(a2, n, 1.0f); /* contracted: 1 + a2*n */
            } while ((double)v <= 0.0);
            v = v * v * v;
            u = orc_canonical_float(g);

            double dn = (double)n;
            double squeeze = fma(-(((0.0331 * dn) * dn) * dn), dn, 1.0);
            if (!((double)u > squeeze)) break;
            double rhs = fma(0.5 * dn, dn,
                             (double)a1 * ((1.0 - (double)v) + (double)logf(v)));
            if (!((double)logf(u) > rhs)) break;
        }
        if (alpha == malpha) {
            out[i] = a1 * v * 1.0f;
        } else {
            do {
                u = orc_canonical_float(g);
            } while (u == 0.0f);
            out[i] = powf(u, 1.0f / alpha) * a1 * v * 1.0f;
        }
    }
}
