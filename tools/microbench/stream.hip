// Streaming comparators for the write- and copy-shaped kernels (k_color_rows_bm16, the build's scatter passes): a fill (write
// only), a read (sum) and a copy of 1 GiB with 16 bytes per lane, non-temporal and plain stores.  GB/s = bytes named in the row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void k_fill(u32x4* __restrict__ dst, uint64_t n16) {
    const u32x4 v = {1, 2, 3, 4};
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, &dst[i]); else dst[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ src, uint64_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n16) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const u32x4 v = src[i];
        if (NT) __builtin_nontemporal_store(v, &dst[i]); else dst[i] = v;
    }
}
// the shape of k_color_rows_bm16: every 16 bytes written are read from a SMALL source (the 75 MB bitmap dictionary of config 5: L2 /
// Infinity Cache hits, no HBM reads) -- a fill whose data comes through the caches.  src_n16 = 16-byte words of the source.
__global__ __launch_bounds__(256) void k_expand(const u32x4* __restrict__ src, uint64_t src_n16, u32x4* __restrict__ dst, uint64_t n16) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t hsh = (uint32_t)(i >> 4) * 0x9E3779B1u;
        const uint64_t j = (((uint64_t)hsh * (uint32_t)(src_n16 / 16)) >> 32) * 16 + (i & 15);  // a pseudo-random 256-byte row of the source, 16 lanes per row
        const u32x4 v = src[j];
        __builtin_nontemporal_store(v, &dst[i]);
    }
}
template <class F>
static double timeit(F f) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; r++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 5;
}
int main() {
    const uint64_t bytes = 1ull << 30, n16 = bytes / 16;
    uint4 *a, *b; uint32_t* o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const dim3 g(256 * 8), t(256);
    double ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_fill<false>, g, t, 0, 0, (u32x4*)a, n16); });
    printf("{\"kernel\": \"fill\", \"bytes_written\": %llu, \"ms\": %.4f, \"GBps\": %.0f}\n", (unsigned long long)bytes, ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_fill<true>, g, t, 0, 0, (u32x4*)a, n16); });
    printf("{\"kernel\": \"fill_nt\", \"bytes_written\": %llu, \"ms\": %.4f, \"GBps\": %.0f}\n", (unsigned long long)bytes, ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_read, g, t, 0, 0, a, n16, o); });
    printf("{\"kernel\": \"read\", \"bytes_read\": %llu, \"ms\": %.4f, \"GBps\": %.0f}\n", (unsigned long long)bytes, ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_copy<false>, g, t, 0, 0, (const u32x4*)a, (u32x4*)b, n16); });
    printf("{\"kernel\": \"copy\", \"bytes_read_plus_written\": %llu, \"ms\": %.4f, \"GBps\": %.0f}\n", (unsigned long long)(2 * bytes), ms, 2 * bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_copy<true>, g, t, 0, 0, (const u32x4*)a, (u32x4*)b, n16); });
    printf("{\"kernel\": \"copy_nt\", \"bytes_read_plus_written\": %llu, \"ms\": %.4f, \"GBps\": %.0f}\n", (unsigned long long)(2 * bytes), ms, 2 * bytes / ms / 1e6);
    for (uint64_t src_mb : {4ull, 75ull, 512ull}) {
        const uint64_t sn16 = (src_mb << 20) / 16;
        ms = timeit([&] { hipLaunchKernelGGL(k_expand, g, t, 0, 0, (const u32x4*)a, sn16, (u32x4*)b, n16); });
        printf("{\"kernel\": \"expand_nt\", \"source_MiB\": %llu, \"bytes_written\": %llu, \"ms\": %.4f, \"GBps_written\": %.0f}\n", (unsigned long long)src_mb, (unsigned long long)bytes, ms, bytes / ms / 1e6);
    }
    return 0;
}
