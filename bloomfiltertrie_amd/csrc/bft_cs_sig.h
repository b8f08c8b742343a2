// bft_cs_sig.h -- the signature of a colour set (the sorted genome-id list of a k-mer): a sum of mixed ids -- order-free, so that the
// elements can be added in any order by any lane -- mixed with the length.  The ORDER of the distinct signatures numbers the colour sets of
// an image (bft_assemble.hip: k_cs_sig over the id lists, then a hash of the signatures): any path that makes signatures must make these.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint64_t bft_mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
// what one genome id adds to the sum
__device__ __forceinline__ uint64_t bft_cs_term(uint32_t id) { return bft_mix64((uint64_t)id + 0x632BE59BD9B4E019ULL); }
// the signature of a list of `len` ids whose terms add up to `sum` (weak: a test hook -- the signature of a list is its length, so that
// different lists collide and the exact pass must run)
__device__ __forceinline__ uint64_t bft_cs_finish(uint64_t sum, uint32_t len, int weak) {
    return weak ? (uint64_t)len : bft_mix64(sum ^ ((uint64_t)len * 0x9E3779B97F4A7C15ULL));
}
