// board_sym.h - bitboards under the symmetries the search shows its leaves in (shared by the transposition
// table's keys and the evaluators that read leaf positions directly).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace az {

// Connect4, symmetry 1: columns mirrored, 7 bits per column (Connect4.h:249-262)
__host__ __device__ inline uint64_t mirror_columns(uint64_t b)
{
    uint64_t r = 0;
    for (int c = 0; c < 7; ++c) r |= ((b >> (7 * c)) & 0x7full) << (7 * (6 - c));
    return r;
}

__host__ __device__ inline uint64_t reverse_bits64(uint64_t x)
{
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0full) | ((x & 0x0f0f0f0f0f0f0f0full) << 4);
    x = ((x >> 8) & 0x00ff00ff00ff00ffull) | ((x & 0x00ff00ff00ff00ffull) << 8);
    x = ((x >> 16) & 0x0000ffff0000ffffull) | ((x & 0x0000ffff0000ffffull) << 16);
    return (x >> 32) | (x << 32);
}

__host__ __device__ inline uint64_t transpose8x8(uint64_t x)
{
    uint64_t t;
    t = 0x0f0f0f0f00000000ull & (x ^ (x << 28)); x ^= t ^ (t >> 28);
    t = 0x3333000033330000ull & (x ^ (x << 14)); x ^= t ^ (t >> 14);
    t = 0x5500550055005500ull & (x ^ (x << 7));  x ^= t ^ (t >> 7);
    return x;
}

// Othello boards (bit = 8 * row + col) under the symmetries the search draws ({0, 2, 6, 7}, Othello.h:45,
// 312-326): stone i moves to T_sym(i); 2 = rotation by 180 degrees (i -> 63 - i), 6 = transposition,
// 7 = anti-transposition (transpose, then rotate)
__host__ __device__ inline uint64_t othello_sym(uint64_t b, int sym)
{
    if (sym == 2) return reverse_bits64(b);
    if (sym == 6) return transpose8x8(b);
    if (sym == 7) return reverse_bits64(transpose8x8(b));
    return b;
}

}  // namespace az
