// bft_hash.h -- the Bloom-filter hash table of a BFT.
//
// The reference never hashes at query time: create_hash_v_array (include/Node.h:158-185) fills
// hash_v[2i] = XXH64(key3(i), r1), hash_v[2i+1] = XXH64(key3(i), r2) once per root, and the probe
// is hash_v[...] % 1504 (src/presenceNode.c:1335-1350).  With compressed == 0 (the CLI default,
// src/main.c:180) only the 2^14 keys n2..n8 are used, so the whole post-modulo table is
// 16384 x 2 x u16 = 64 KiB.  XXH64 is restated from the published xxHash specification
// (reference vendors xxHash 0.6.2, src/xxhash.c); only the 3-byte input case is needed.
#pragma once
#include <stdint.h>
#include "bft_image.h"

#define BFT_DEFAULT_R1 1804289383  // glibc rand() #1 without srand(): include/CC.h:246
#define BFT_DEFAULT_R2 846930886   // glibc rand() #2

static inline uint64_t bft_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// XXH64 of a 3-byte input (len < 4: only the byte tail and the avalanche run)
static inline uint64_t bft_xxh64_3(const uint8_t* p, uint64_t seed) {
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL,
                   P5 = 2870177450012600261ULL;
    uint64_t h = seed + P5 + 3;
    for (int i = 0; i < 3; i++) {
        h ^= p[i] * P5;
        h = bft_rotl64(h, 11) * P1;
    }
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return h;
}

// hashmod[key] = (h1 % 1504) | (h2 % 1504) << 16 for the 14-bit Bloom keys
static inline void bft_make_hashmod(int r1, int r2, uint32_t* hashmod /*[16384]*/) {
    for (uint32_t i = 0; i < 16384; i++) {
        uint8_t g[3] = {(uint8_t)((i >> 10) & 0xff), (uint8_t)((i >> 2) & 0xff), (uint8_t)((i << 6) & 0xff)};
        uint32_t h1 = (uint32_t)(bft_xxh64_3(g, (uint64_t)(long long)r1) % BFT_MODULO_HASH);
        uint32_t h2 = (uint32_t)(bft_xxh64_3(g, (uint64_t)(long long)r2) % BFT_MODULO_HASH);
        hashmod[i] = h1 | (h2 << 16);
    }
}
