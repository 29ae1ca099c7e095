// mt19937ar.hpp — MATLAB's default random stream, RandStream("mt19937ar", Seed = s), restated from the published algorithm
// (Matsumoto & Nishimura, mt19937ar.c: init_genrand, genrand_int32, genrand_res53).  Host code.
//   rand(stream)       = genrand_res53: (a * 2^26 + b) / 2^53 with a = draw >> 5, b = draw >> 6        MonteCarloTreeSearch.m:53
//   randi(stream, n)   = floor(n * rand(stream)) + 1                                                   PrioritizedExplorativeController.m:283-286
//   Seed = 0 is the generator's default seed 5489.
#pragma once
#include <cstdint>

struct Mt19937ar {
    uint32_t mt[624];
    int mti = 624;
    explicit Mt19937ar(uint32_t seed) {
        mt[0] = seed ? seed : 5489u;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    }
    uint32_t next() {
        if (mti >= 624) {
            for (int k = 0; k < 624; ++k) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            mti = 0;
        }
        uint32_t y = mt[mti++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double rand() {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
    }
    int randi(int n) { return (int)(n * rand()) + 1; }  // (n * rand() >= 0: the cast is floor)
};
