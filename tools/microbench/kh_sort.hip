// The k-mer hash build's sort by home line: 4.46x10^7 (u32 key of 24 bits, 12-byte record) pairs, library onesweep with 8-bit digits (3 passes:
// what the build runs), 12-bit digits (2 passes) and in between.  hipcc --offload-arch=gfx950 -O3 -o kh_sort kh_sort.hip && ./kh_sort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct __attribute__((packed, aligned(4))) Rec { uint64_t t; uint32_t v; };
__global__ void k_fill(uint32_t* k, Rec* r, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        k[i] = (uint32_t)(x % 10485760u);
        Rec q; q.t = i; q.v = (uint32_t)x; r[i] = q;
    }
}
__global__ void k_check(const uint32_t* p, const Rec* r, uint64_t n, unsigned* bad) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i + 1 < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (p[i] > p[i + 1] || (p[i] == p[i + 1] && r[i].t > r[i + 1].t)) atomicAdd(bad, 1u);  // sorted and stable
}
template <class Cfg>
int run(const char* name, uint32_t* in, uint32_t* out, Rec* ri, Rec* ro, uint64_t n, unsigned* bad, unsigned lo = 0u) {
    size_t tb = 0;
    CK((rocprim::radix_sort_pairs<Cfg>(nullptr, tb, in, out, ri, ro, (size_t)n, lo, 24u, 0)));
    void* tmp; CK(hipMalloc(&tmp, tb));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int r = 0; r < 4; r++) {
        CK(hipEventRecord(a, 0));
        CK((rocprim::radix_sort_pairs<Cfg>(tmp, tb, in, out, ri, ro, (size_t)n, lo, 24u, 0)));
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CK(hipMemset(bad, 0, 4));
    if (lo == 0) hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, out, ro, n, bad);
    unsigned nb = 0; CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
    printf("{\"what\": \"radix_sort_pairs u32 (24 bits) + 12-byte record\", \"config\": \"%s\", \"n\": %llu, \"ms\": %.3f, \"bad\": %u}\n", name, (unsigned long long)n, best, nb);
    CK(hipFree(tmp));
    return 0;
}
template <int BITS, int BLK, int IPT>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<BLK, IPT>, BITS,
                                                                           rocprim::block_radix_rank_algorithm::match>>;
int main() {
    const uint64_t n = 44600000;
    uint32_t *in, *out; Rec *ri, *ro; unsigned* bad;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&ri, n * 12)); CK(hipMalloc(&ro, n * 12)); CK(hipMalloc(&bad, 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, ri, n);
    if (run<rocprim::default_config>("default (8-bit digits, 3 passes)", in, out, ri, ro, n, bad)) return 1;
    if (run<Cfg<8, 1024, 8>>("8 bits, 1024 x 8", in, out, ri, ro, n, bad)) return 1;
    if (run<Cfg<9, 1024, 8>>("9 bits, 1024 x 8 (3 passes)", in, out, ri, ro, n, bad)) return 1;
    // (12-bit digits: 278 KB of LDS for the sort kernel, 196 KB for the histogram -- does not compile for gfx950's 160 KB)
#ifdef TRY11
    if (run<Cfg<11, 1024, 8>>("11 bits, 1024 x 8, bits [2, 24) only (2 passes: what a sort by groups of 4 lines would cost)", in, out, ri, ro, n, bad, 2u)) return 1;
    if (run<Cfg<11, 512, 8>>("11 bits, 512 x 8, bits [2, 24) only (2 passes)", in, out, ri, ro, n, bad, 2u)) return 1;
#endif
    return 0;
}
