// bft_kernels_load.h -- coalesced loads of packed 2-bit k-mers from a batch (device code shared by the translation units of libbft_gpu.so)
#pragma once
// Packed k-mer i -> X words, straight from global memory (the 64 lanes of a wavefront read one contiguous 64*B-byte span).
// One load instruction per 16 bytes of the k-mer, at its byte address (gfx950 takes unaligned dwordx2 / dwordx4 global loads): a vector
// memory instruction costs the CU's address path the same ~64 cycles whether its 64 lanes ask for 4 or 16 bytes each, and the query kernels
// are bound by those cycles as much as by the lines they miss on (round 4: 8 load instructions per k-mer made 31 G k-mers/s where 6 made
// 47) -- the aligned-dword form of round 3 (2W + 1 dword loads + funnel shifts) spent three instructions on a 7-byte k-mer.  The words
// behind the k-mer's B bytes are masked off; the last k-mers of a buffer, whose 8 W bytes would cross its end, take a byte path.
struct __attribute__((packed, aligned(1))) BftU64u { uint64_t v; };
struct __attribute__((packed, aligned(1))) BftU128u { uint64_t a, b; };
template <int W>
__device__ __forceinline__ void load_x(const uint8_t* __restrict__ packed, uint64_t i, int B, uint64_t end_aligned, uint64_t* x) {
    const uint8_t* q = packed + i * (uint64_t)B;
    if ((uint64_t)q + 8ull * W <= end_aligned) {
        if (W == 1) x[0] = reinterpret_cast<const BftU64u*>(q)->v;
        else {
#pragma unroll
            for (int w = 0; w + 1 < W; w += 2) {
                const BftU128u v = *reinterpret_cast<const BftU128u*>(q + 8 * w);
                x[w] = v.a;
                x[w + 1] = v.b;
            }
            if (W & 1) x[W - 1] = reinterpret_cast<const BftU64u*>(q + 8 * (W - 1))->v;
        }
    } else {
#pragma unroll
        for (int w = 0; w < W; w++) x[w] = 0;
        for (int b = 0; b < B; b++) {
            const uint64_t v = (uint64_t)q[b] << (8 * (b & 7));
#pragma unroll
            for (int w = 0; w < W; w++)
                if ((b >> 3) == w) x[w] |= v;
        }
    }
    const int rem = B - 8 * (W - 1);
    if (rem < 8) x[W - 1] &= (1ull << (8 * rem)) - 1ull;
}
