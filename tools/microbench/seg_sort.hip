// Segmented radix sort of (64-bit key, 32-bit value) pairs -- the second word of multi-word k-mers, sorted inside the runs of an equal first
// word -- against the device-wide sort of the same pairs: 4x10^7 pairs in segments of ~1440 (config 5: 2000 colours over one locus), ~8
// (a pan-genome at k = 63) and 1.  hipcc --offload-arch=gfx950 -O3 -o seg_sort seg_sort.hip && ./seg_sort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_fill(uint64_t* p, uint32_t* v, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[i] = x;
        v[i] = (uint32_t)i;
    }
}
__global__ void k_check(const uint64_t* p, const uint32_t* off, uint32_t nseg, unsigned* bad) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += gridDim.x * blockDim.x)
        for (uint32_t i = off[s]; i + 1 < off[s + 1]; i++)
            if (p[i] > p[i + 1]) atomicAdd(bad, 1u);
}
int main() {
    const uint64_t n = 40000000;
    uint64_t *in, *out; uint32_t *vi, *vo, *off; unsigned* bad;
    CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&vi, n * 4)); CK(hipMalloc(&vo, n * 4)); CK(hipMalloc(&off, (n + 1) * 4)); CK(hipMalloc(&bad, 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, vi, n);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (uint32_t seg : {1440u, 100u, 8u, 2u, 1u}) {
        std::vector<uint32_t> h;
        uint64_t x = 12345;
        for (uint64_t at = 0; at < n;) {  // lengths in [seg / 2, 3 seg / 2]
            h.push_back((uint32_t)at);
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            at += seg == 1 ? 1 : seg / 2 + (x >> 33) % (seg + 1);
        }
        const uint32_t nseg = (uint32_t)h.size();
        h.push_back((uint32_t)n);
        CK(hipMemcpy(off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        size_t tb = 0;
        CK(rocprim::segmented_radix_sort_pairs(nullptr, tb, in, out, vi, vo, (unsigned)n, nseg, off, off + 1, 0, 64, 0));
        void* tmp; CK(hipMalloc(&tmp, tb));
        float best = 1e9;
        for (int r = 0; r < 4; r++) {
            CK(hipEventRecord(a, 0));
            CK(rocprim::segmented_radix_sort_pairs(tmp, tb, in, out, vi, vo, (unsigned)n, nseg, off, off + 1, 0, 64, 0));
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        CK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, out, off, nseg, bad);
        unsigned nb = 0; CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
        printf("{\"what\": \"segmented_radix_sort_pairs u64+u32\", \"n\": %llu, \"mean_segment\": %u, \"segments\": %u, \"ms\": %.3f, \"tmp_MB\": %.1f, \"bad\": %u}\n", (unsigned long long)n, seg, nseg, best, tb / 1048576.0, nb);
        CK(hipFree(tmp));
    }
    {
        size_t tb = 0;
        CK(rocprim::radix_sort_pairs(nullptr, tb, in, out, vi, vo, (unsigned)n, 0, 64, 0));
        void* tmp; CK(hipMalloc(&tmp, tb));
        float best = 1e9;
        for (int r = 0; r < 4; r++) {
            CK(hipEventRecord(a, 0));
            CK(rocprim::radix_sort_pairs(tmp, tb, in, out, vi, vo, (unsigned)n, 0, 64, 0));
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("{\"what\": \"radix_sort_pairs u64+u32 (device-wide, 64 bits)\", \"n\": %llu, \"ms\": %.3f}\n", (unsigned long long)n, best);
    }
    return 0;
}
