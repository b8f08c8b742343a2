// rocPRIM onesweep on the top 18 bits of 2x10^8 64-bit keys: the default configuration (8-bit digits: 3 passes) against 9-bit digits
// (2 passes).  hipcc --offload-arch=gfx950 -O3 -o msd_sort msd_sort.hip && ./msd_sort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_fill(uint64_t* p, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[i] = x >> 3;
    }
}
__global__ void k_check(const uint64_t* p, uint64_t n, int lo, unsigned* bad) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i + 1 < n; i += (uint64_t)gridDim.x * blockDim.x)
        if ((p[i] >> lo) > (p[i + 1] >> lo)) atomicAdd(bad, 1u);
}
template <class Cfg>
int run(const char* name, uint64_t* in, uint64_t* out, uint64_t n, int lo, int hi) {
    size_t tb = 0;
    CK((rocprim::radix_sort_keys<Cfg>(nullptr, tb, in, out, (uint32_t)n, lo, hi, 0)));
    void* tmp; CK(hipMalloc(&tmp, tb));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int r = 0; r < 4; r++) {
        CK(hipEventRecord(a, 0));
        CK((rocprim::radix_sort_keys<Cfg>(tmp, tb, in, out, (uint32_t)n, lo, hi, 0)));
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
    }
    unsigned* bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(k_check, dim3(2048), dim3(256), 0, 0, out, n, lo, bad);
    unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("{\"config\": \"%s\", \"keys\": %llu, \"bits\": %d, \"ms\": %.3f, \"out_of_order\": %u, \"temp_mb\": %.1f}\n", name, (unsigned long long)n, hi - lo, best, hb, tb / 1048576.0);
    (void)hipFree(tmp); (void)hipFree(bad);
    return 0;
}
int main() {
    const uint64_t n = 200000000ull;
    uint64_t *in, *out;
    CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, n * 8));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, in, n);
    using D = rocprim::default_config;
    using C9a = rocprim::radix_sort_config<D, D, rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<512, 12>, 9, rocprim::block_radix_rank_algorithm::match>>;
    using C9b = rocprim::radix_sort_config<D, D, rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<512, 15>, 9, rocprim::block_radix_rank_algorithm::match>>;
    using C9c = rocprim::radix_sort_config<D, D, rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<1024, 8>, 9, rocprim::block_radix_rank_algorithm::match>>;
    using C6 = rocprim::radix_sort_config<D, D, rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<256, 12>, 6, rocprim::block_radix_rank_algorithm::match>>;
    if (run<D>("default", in, out, n, 43, 61)) return 1;
    if (run<C9a>("9 bits, 512x12", in, out, n, 43, 61)) return 1;
    if (run<C9b>("9 bits, 512x15", in, out, n, 43, 61)) return 1;
    if (run<C9c>("9 bits, 1024x8", in, out, n, 43, 61)) return 1;
    if (run<C6>("6 bits, 256x12", in, out, n, 43, 61)) return 1;
    return 0;
}
