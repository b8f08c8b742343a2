// Microbenchmark behind DESIGN.md "What bounds the query kernels": the random-gather rate of MI355X beyond the L2.
//
//   chase<EVERY, 8|16>  one DEPENDENT chain per lane (R rounds): the round-1 measurement; EVERY = 4 leaves every 4th lane active
//                       (same wave instructions, a quarter of the addresses)
//   indep<K>            K INDEPENDENT 8-byte gathers in flight per lane and round (K = 2, 4, 8): is the dependent-chain figure a
//                       latency x occupancy bound, or the request rate of the fabric?
//   block<B>            one gather = a whole aligned B-byte block (B = 32, 64, 128; 16-byte loads), dependent chain: what does a
//                       request move?  If a 128-byte block costs what a 64-byte one costs, the unit of a miss is 128 bytes.
//
// Every kernel makes a KNOWN number of gathers (lanes x rounds x K), so `rocprofv3 --pmc FETCH_SIZE` (or TCC_MISS_sum,
// TCC_EA0_RDREQ_sum ...) of this binary calibrates "bytes tallied per gather" for this access pattern -- the guide's advice for
// anything that is not a wide coalesced stream (MI355X_MICROARCH.md, HBM).  usage: gather [table_MiB ...]   (default 2 64 1024 8192)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

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

template <int K>
__global__ __launch_bounds__(1024) void indep(const uint64_t* __restrict__ tab, uint64_t mask, int rounds, uint64_t n, uint64_t* out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t seed = i * 0x9E3779B97F4A7C15ull, acc = 0;
        for (int r = 0; r < rounds; r++) {
            uint64_t v[K];
#pragma unroll
            for (int j = 0; j < K; j++) v[j] = tab[mix(seed + (uint64_t)j * 0xD1B54A32D192ED03ull) & mask];  // K addresses known up front
            uint64_t s = 0;
#pragma unroll
            for (int j = 0; j < K; j++) s += v[j];
            acc += s;
            seed = mix(seed ^ s) + r;  // the next round depends on all K
        }
        out[i] = acc;
    }
}

template <int BYTES>
__global__ __launch_bounds__(1024) void block(const uint64_t* __restrict__ tab, uint64_t mask, int rounds, uint64_t n, uint64_t* out) {
    constexpr int NW = BYTES / 8;  // words per block
    const uint64_t bmask = mask / NW;  // blocks - 1 (table words are a power of two)
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t b = (i * 0x9E3779B97F4A7C15ull) & bmask, acc = 0;
        for (int r = 0; r < rounds; r++) {
            const ulonglong2* p = (const ulonglong2*)(tab + b * NW);
            uint64_t s = 0;
#pragma unroll
            for (int j = 0; j < NW / 2; j++) { const ulonglong2 w = p[j]; s += w.x ^ w.y; }
            acc += s;
            b = (s ^ (b * 0xD1B54A32D192ED03ull + r)) & bmask;
        }
        out[i] = acc;
    }
}

// quadline: one whole 64-byte line per lane and round, fetched the way k_query_kh fetches home lines (round 4): the four lanes of a quad load
// 16 bytes each of ONE lane's line, four instructions bring the quad's four lines -- every line is requested once, whole, by one
// instruction.  lane64: the same lines, every lane loading its own with four 16-byte loads (four instructions touch 64 lines each).
// The next round's line depends on nothing that was loaded (the product kernel's k-mers are independent): seeds only.
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v, int j) {
    switch (j) {
    case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xF, 0xF, true);
    case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xF, 0xF, true);
    case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xF, 0xF, true);
    default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xF, 0xF, true);
    }
}
template <bool QUAD>
__global__ __launch_bounds__(1024) void lines(const uint64_t* __restrict__ tab, uint64_t mask, int rounds, uint64_t n, uint64_t* out) {
    const uint64_t lmask = mask / 8;  // lines - 1
    const uint32_t ql = threadIdx.x & 3u;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t acc = 0;
        for (int r = 0; r < rounds; r++) {
            const uint64_t ln = mix(i * 0x9E3779B97F4A7C15ull + (uint64_t)r * 0xD1B54A32D192ED03ull) & lmask;
            if (QUAD) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint64_t l = ((uint64_t)quad_bcast((uint32_t)(ln >> 32), q) << 32) | quad_bcast((uint32_t)ln, q);
                    const ulonglong2 w = *(const ulonglong2*)(tab + l * 8 + 2 * ql);
                    acc += w.x ^ w.y;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const ulonglong2 w = *(const ulonglong2*)(tab + ln * 8 + 2 * q);
                    acc += w.x ^ w.y;
                }
            }
        }
        out[i] = acc;
    }
}

typedef void (*kern_t)(const uint64_t*, uint64_t, int, uint64_t, uint64_t*);
static double run(kern_t k, const uint64_t* tab, uint64_t mask, int rounds, uint64_t n, uint64_t* out, int grid) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 0, 0, tab, mask, rounds, n, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 0, 0, tab, mask, rounds, n, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / 3;
}

__global__ void fill(uint64_t* tab, uint64_t words) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) tab[i] = mix(i + 0x632BE59BD9B4E019ull);
}

int main(int argc, char** argv) {
    const uint64_t n = 1ull << 25;  // lanes
    const int rounds = 8;
    std::vector<uint64_t> sizes;
    for (int a = 1; a < argc; a++) sizes.push_back(strtoull(argv[a], nullptr, 10));
    if (sizes.empty()) sizes = {2, 64, 1024, 8192};
    uint64_t* out;
    CK(hipMalloc(&out, n * 8));
    for (uint64_t mib : sizes) {
        const uint64_t tbytes = mib << 20, words = tbytes / 8;
        uint64_t* tab;
        CK(hipMalloc(&tab, tbytes));
        hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, tab, words);
        CK(hipDeviceSynchronize());
        const int grid = 512;  // two 1024-thread workgroups per CU
        const double gl = (double)n * rounds;
        struct { const char* name; kern_t k; double per_lane_round; } v[] = {
            {"chase_dep8", chase<1, 8>, 1}, {"chase_dep8_quarter_lanes", chase<4, 8>, 0.25}, {"chase_dep16", chase<1, 16>, 1},
            {"indep2", indep<2>, 2}, {"indep4", indep<4>, 4}, {"indep8", indep<8>, 8},
            {"block32", block<32>, 1}, {"block64", block<64>, 1}, {"block128", block<128>, 1},
            {"lines64_by_lane", lines<false>, 1}, {"lines64_by_quad", lines<true>, 1},
        };
        for (auto& x : v) {
            const double ms = run(x.k, tab, words - 1, rounds, n, out, grid);
            printf("{\"table_MiB\": %llu, \"kernel\": \"%s\", \"gathers\": %.0f, \"ms\": %.3f, \"G_gathers_per_s\": %.1f}\n", (unsigned long long)mib, x.name,
                   gl * x.per_lane_round, ms, gl * x.per_lane_round / ms / 1e6);
            fflush(stdout);
        }
        CK(hipFree(tab));
    }
    return 0;
}
