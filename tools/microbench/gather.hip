// Microbenchmark behind DESIGN.md section 6: what bounds a kernel of dependent random 8-byte gathers on MI355X --
// wave-level instruction count or active-lane (address) count?  Each lane chases R dependent pointers through a
// table of T bytes; in the "sparse" variant only every 4th lane is active (same number of wave instructions,
// a quarter of the addresses).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int ACTIVE_EVERY, int WIDTH>
__global__ __launch_bounds__(1024) void chase(const uint64_t* __restrict__ tab, uint64_t mask, int rounds, uint64_t n, uint64_t* out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if ((threadIdx.x % ACTIVE_EVERY) != 0) continue;
        uint64_t idx = (i * 0x9E3779B97F4A7C15ull) & mask;
        uint64_t acc = 0;
        for (int r = 0; r < rounds; r++) {
            uint64_t v;
            if (WIDTH == 16) {
                const uint64_t a = idx & ~1ull;
                const ulonglong2 w = *(const ulonglong2*)(tab + a);
                v = w.x ^ w.y;
            } else
                v = tab[idx];
            acc += v;
            idx = (v ^ (idx * 0xD1B54A32D192ED03ull + r)) & mask;
        }
        out[i] = acc;
    }
}

template <int AE, int WD>
static double run(const uint64_t* tab, uint64_t mask, int rounds, uint64_t n, uint64_t* out, int grid) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((chase<AE, WD>), dim3(grid), dim3(1024), 0, 0, tab, mask, rounds, n, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((chase<AE, WD>), dim3(grid), dim3(1024), 0, 0, tab, mask, rounds, n, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / 3;
}

int main() {
    const uint64_t n = 1ull << 26;  // lanes
    const int rounds = 8;
    uint64_t* out;
    CK(hipMalloc(&out, n * 8));
    for (uint64_t tbytes : {1ull << 21, 1ull << 26, 1ull << 30}) {
        const uint64_t words = tbytes / 8;
        std::vector<uint64_t> h(words);
        uint64_t s = 88172645463325252ull;
        for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = s; }
        uint64_t* tab;
        CK(hipMalloc(&tab, tbytes));
        CK(hipMemcpy(tab, h.data(), tbytes, hipMemcpyHostToDevice));
        for (int grid : {512, 2048}) {
            const double d8 = run<1, 8>(tab, words - 1, rounds, n, out, grid);
            const double s8 = run<4, 8>(tab, words - 1, rounds, n, out, grid);
            const double d16 = run<1, 16>(tab, words - 1, rounds, n, out, grid);
            const double gl = (double)n * rounds;
            printf("{\"table_MiB\": %llu, \"grid\": %d, \"dense8_ms\": %.3f, \"dense8_Gloads_s\": %.1f, \"sparse8_ms\": %.3f, \"sparse8_Gloads_s\": %.1f, "
                   "\"dense16_ms\": %.3f, \"dense16_Gloads_s\": %.1f}\n",
                   (unsigned long long)(tbytes >> 20), grid, d8, gl / d8 / 1e6, s8, gl / 4 / s8 / 1e6, d16, gl / d16 / 1e6);
        }
        CK(hipFree(tab));
    }
    return 0;
}
