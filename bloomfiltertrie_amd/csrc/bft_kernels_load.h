// bft_kernels_load.h -- coalesced loads of packed 2-bit k-mers from a batch (device code shared by the translation units of libbft_gpu.so)
#pragma once
// Packed k-mer i -> X words, straight from global memory: the 64 lanes of a wavefront read one
// contiguous 64*B-byte span with aligned dword loads (each lane the <= 2W+1 dwords that cover its
// B bytes), then funnel-shift.  The last k-mers of a buffer whose window would cross the end of the
// buffer take a byte path.
template <int W>
__device__ __forceinline__ void load_x(const uint8_t* __restrict__ packed, uint64_t i, int B, uint64_t end_aligned, uint64_t* x) {
    constexpr int NDW = 2 * W + 1;
    const uint64_t addr = (uint64_t)packed + i * (uint64_t)B;
    const uint64_t a = addr & ~3ull;
    const uint32_t mis = (uint32_t)(addr & 3ull), sh = mis * 8;
    const uint32_t need = (mis + (uint32_t)B + 3u) >> 2;
    uint32_t dw[NDW];
    if (a + 4ull * need <= end_aligned) {
        const uint32_t* p = (const uint32_t*)a;
#pragma unroll
        for (int j = 0; j < NDW; j++) dw[j] = ((uint32_t)j < need) ? p[j] : 0u;
    } else {
#pragma unroll
        for (int j = 0; j < NDW; j++) dw[j] = 0;
        const uint8_t* q = (const uint8_t*)addr;
        for (int b = 0; b < B; b++) {
            const uint32_t pos = mis + (uint32_t)b, v = (uint32_t)q[b] << (8 * (pos & 3));
#pragma unroll
            for (int j = 0; j < NDW; j++)
                if ((pos >> 2) == (uint32_t)j) dw[j] |= v;
        }
    }
#pragma unroll
    for (int w = 0; w < W; w++) {
        const uint64_t lo = (uint64_t)dw[2 * w] | ((uint64_t)dw[2 * w + 1] << 32);
        const uint64_t hi = dw[2 * w + 2];
        x[w] = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    }
    const int rem = B - 8 * (W - 1);
    if (rem < 8) x[W - 1] &= (1ull << (8 * rem)) - 1ull;
}

