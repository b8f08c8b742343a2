// The library's own radix sort (bloomfiltertrie_amd/csrc/bft_sort.h) on its own: correctness against std::stable_sort on small and ragged
// sizes, then the three shapes the build runs -- the root-prefix split (2 x 10^8 composites, 18 bits in two passes), the k-mer hash's sort by
// home line (4.46 x 10^7 u32 keys + 12-byte records, 24 bits) and a small 64-bit sort (1.6 x 10^6 keys + u32) -- beside rocPRIM's onesweep.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o rs_sort rs_sort.hip && ./rs_sort [quick]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <vector>

#include "../../bloomfiltertrie_amd/csrc/bft_sort.h"
#include "../../bloomfiltertrie_amd/csrc/bft_scan.h"

int bft_rs::g_bft_rs_rank_mode = -1;
static std::string g_err;
int bft_fail(int code, const std::string& msg) { g_err = msg; fprintf(stderr, "fail: %s\n", msg.c_str()); return code; }
int bft_pool_alloc(void** p, size_t n, size_t* cap) { *cap = n; return hipMalloc(p, n) == hipSuccess ? 0 : -1; }
void bft_pool_release(void* p, size_t) { (void)hipFree(p); }
int bft_zero_async(void* p, size_t bytes, hipStream_t s) { return hipMemsetAsync(p, 0, bytes, s) == hipSuccess ? 0 : -1; }  // (nothing is captured here)

#define HCK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct __attribute__((packed, aligned(4))) Rec12 { uint64_t t; uint32_t v; };
__host__ __device__ static inline uint64_t mix(uint64_t i) { uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x; }

template <class K, class V>
static int check_case(uint64_t n, unsigned b0, unsigned b1, int skew, const char* name) {
    std::vector<K> hk(n);
    std::vector<V> hv(n);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t x = mix(i + 12345 * n);
        if (skew == 1) x &= 0xFFFFull << b0;              // few distinct digits
        if (skew == 2) x = (i % 7 == 0) ? x : (3ull << b0);  // one heavy digit
        hk[i] = (K)x;
        memset(&hv[i], 0, sizeof(V));
        uint32_t tag = (uint32_t)i;
        if (!std::is_same<V, bft_rs::NoVal>::value) memcpy(&hv[i], &tag, std::min<size_t>(4, sizeof(V)));  // the value remembers the input position: stability is visible
    }
    K *dk, *ok, *tk;
    V *dv, *ov, *tv;
    HCK(hipMalloc(&dk, n * sizeof(K) + 16)); HCK(hipMalloc(&ok, n * sizeof(K) + 16)); HCK(hipMalloc(&tk, n * sizeof(K) + 16));
    HCK(hipMalloc(&dv, n * sizeof(V) + 16)); HCK(hipMalloc(&ov, n * sizeof(V) + 16)); HCK(hipMalloc(&tv, n * sizeof(V) + 16));
    HCK(hipMemcpy(dk, hk.data(), n * sizeof(K), hipMemcpyHostToDevice));
    HCK(hipMemcpy(dv, hv.data(), n * sizeof(V), hipMemcpyHostToDevice));
    DevBuf scratch;
    int rc = bft_rs::sort<K, V>(bft_rs::PtrIn<K, V>{dk, dv}, n, ok, ov, tk, tv, b0, b1, 0, scratch);
    if (rc) return 1;
    HCK(hipDeviceSynchronize());
    std::vector<K> rk(n);
    std::vector<V> rv(n);
    HCK(hipMemcpy(rk.data(), ok, n * sizeof(K), hipMemcpyDeviceToHost));
    HCK(hipMemcpy(rv.data(), ov, n * sizeof(V), hipMemcpyDeviceToHost));
    std::vector<uint32_t> idx(n);
    for (uint64_t i = 0; i < n; i++) idx[i] = (uint32_t)i;
    const uint64_t m = b1 - b0 >= 64 ? ~0ull : ((1ull << (b1 - b0)) - 1ull);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return (((uint64_t)hk[a] >> b0) & m) < (((uint64_t)hk[b] >> b0) & m); });
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint32_t tag = 0, want = idx[i];
        if (!std::is_same<V, bft_rs::NoVal>::value) memcpy(&tag, &rv[i], std::min<size_t>(4, sizeof(V)));
        else want = 0;
        if (sizeof(V) < 4) want &= (1u << (8 * sizeof(V))) - 1u;
        if (rk[i] != hk[idx[i]] || tag != want) bad++;
    }
    printf("{\"check\": \"%s\", \"n\": %llu, \"bits\": [%u, %u], \"skew\": %d, \"bad\": %llu}\n", name, (unsigned long long)n, b0, b1, skew, (unsigned long long)bad);
    fflush(stdout);
    hipFree(dk); hipFree(ok); hipFree(tk); hipFree(dv); hipFree(ov); hipFree(tv);
    return bad ? 1 : 0;
}

__global__ void k_fill64(uint64_t* k, uint64_t n, int gb) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t locus = mix(i % 44600000ull) & ((1ull << 54) - 1ull);  // a pan-genome: the same k-mers again and again, ids ascending
        k[i] = (locus << gb) | (i / 2000000ull);
    }
}
__global__ void k_fill_kh(uint32_t* k, Rec12* r, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t x = mix(i);
        k[i] = (uint32_t)(x % 10485760u);
        Rec12 q; q.t = i; q.v = (uint32_t)x; r[i] = q;
    }
}
__global__ void k_check_split(const uint64_t* c, uint64_t n, unsigned shift, unsigned* bad) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i + 1 < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = c[i] >> shift, b = c[i + 1] >> shift;
        if (a > b || (a == b && (c[i] & 127) > (c[i + 1] & 127))) atomicAdd(bad, 1u);  // sorted on the top bits, ids still ascending inside a bucket
    }
}
__global__ void k_check_kh(const uint32_t* p, const Rec12* r, uint64_t n, unsigned* bad) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i + 1 < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (p[i] > p[i + 1] || (p[i] == p[i + 1] && r[i].t > r[i + 1].t)) atomicAdd(bad, 1u);
}

using Msd9 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                        rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<1024, 8>, 9, rocprim::block_radix_rank_algorithm::match>>;

struct FlagIn {  // flags computed on the fly, two counts in one word (what the de-duplication scans)
    const uint64_t* k;
    __device__ uint64_t operator()(uint64_t i) const { const uint64_t a = k[i], b = i ? k[i - 1] : ~a; return ((uint64_t)((a >> 9) != (b >> 9)) << 32) | (uint64_t)(a != b); }
};
template <class T>
static int check_scan(uint64_t n, int what) {
    std::vector<T> h(n), want(n), got(n);
    for (uint64_t i = 0; i < n; i++) h[i] = (T)(what == 2 ? (mix(i) & 0xFFFFFFFFFFull) : (mix(i) % 7));
    T run = what == 2 ? (T)5 : (T)0;
    for (uint64_t i = 0; i < n; i++) {
        if (what == 0) { want[i] = run; run += h[i]; }                          // exclusive sum
        else if (what == 1) { run += h[i]; want[i] = run; }                     // inclusive sum
        else { run = std::max(run, h[i]); want[i] = run; }                      // inclusive max, init 5
    }
    T *d, *o;
    HCK(hipMalloc(&d, n * sizeof(T) + 16)); HCK(hipMalloc(&o, n * sizeof(T) + 16));
    HCK(hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    static DevBuf sc;  // shared by all the checks: the scans tell their states apart by epoch, nothing is zeroed in between
    unsigned long long* d_total;
    HCK(hipMalloc(&d_total, 8));
    int rc = 0;
    uint64_t bad = 0;
    for (int rep = 0; rep < 3 && !rc; rep++) {  // the same scratch block three times in a row
        HCK(hipMemset(o, 0xEE, n * sizeof(T) + 16));
        if (what == 0) rc = bft_scan::exclusive_sum_ptr<T>(d, o, n, 0, sc, d_total, true);
        else if (what == 1) rc = bft_scan::scan<T, bft_scan::PtrIn<T>, bft_scan::Sum, true>(bft_scan::PtrIn<T>{d}, o, n, (T)0, bft_scan::Sum(), 0, sc, d_total);
        else rc = bft_scan::scan<T, bft_scan::PtrIn<T>, bft_scan::Max, true>(bft_scan::PtrIn<T>{d}, o, n, (T)5, bft_scan::Max(), 0, sc, d_total);
        if (rc) break;
        HCK(hipMemcpy(got.data(), o, n * sizeof(T), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; i++) bad += got[i] != want[i];
        unsigned long long tot = 0;
        T tail = 0;
        HCK(hipMemcpy(&tot, d_total, 8, hipMemcpyDeviceToHost));
        HCK(hipMemcpy(&tail, o + n, sizeof(T), hipMemcpyDeviceToHost));
        bad += tot != (unsigned long long)run;           // the grand total (for the exclusive sum: the sum of everything)
        if (what == 0) bad += tail != run;                // ... and behind the last offset
    }
    hipFree(d_total);
    if (rc) return 1;
    printf("{\"check\": \"scan %s u%d\", \"n\": %llu, \"bad\": %llu}\n", what == 0 ? "exclusive sum" : what == 1 ? "inclusive sum" : "inclusive max", (int)sizeof(T) * 8, (unsigned long long)n, (unsigned long long)bad);
    hipFree(d); hipFree(o);
    return bad ? 1 : 0;
}

template <class F>
static float time_it(F f, int reps = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(a, 0);
        f();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = std::min(best, ms);
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    int fails = 0;
    const bool nocheck = argc > 1 && !strcmp(argv[1], "bench");
    if (!nocheck)
    for (uint64_t n : {1ull, 63ull, 64ull, 65ull, 2047ull, 4096ull, 4097ull, 100000ull, 1000003ull, 5000011ull}) {
        fails += check_case<uint64_t, uint32_t>(n, 7, 25, 0, "u64+u32 18 bits");
        fails += check_case<uint64_t, bft_rs::NoVal>(n, 0, 64, 0, "u64 keys 64 bits");
    }
    if (!nocheck) {
    fails += check_case<uint32_t, Rec12>(3000017, 0, 24, 0, "u32+rec12 24 bits");
    fails += check_case<uint64_t, uint32_t>(3000017, 3, 14, 1, "u64+u32 11 bits few digits");
    fails += check_case<uint64_t, uint32_t>(6000017, 20, 38, 2, "u64+u32 18 bits heavy digit");
    fails += check_case<uint64_t, uint8_t>(4500000, 44, 62, 0, "u64+u8 18 bits");
    fails += check_case<uint32_t, bft_rs::NoVal>(1234567, 5, 5, 0, "no bits (copy)");
    fails += check_case<uint64_t, uint32_t>(20000003, 3, 40, 0, "u64+u32 37 bits, chained");
    }
    if (!nocheck)
        for (uint64_t n : {1ull, 63ull, 4096ull, 4097ull, 1000003ull, 30000001ull}) {
            fails += check_scan<uint32_t>(n, 0);
            fails += check_scan<uint64_t>(n, 0);
            fails += check_scan<uint32_t>(n, 1);
            fails += check_scan<uint64_t>(n, 2);
        }
    if (fails) { printf("{\"failed_checks\": %d}\n", fails); return 1; }
    if (quick) return 0;
    DevBuf scratch;
    unsigned* bad;
    HCK(hipMalloc(&bad, 4));
    {   // the root-prefix split
        const uint64_t n = 200000000ull;
        const int gb = 7;
        uint64_t *in, *out, *tmp;
        HCK(hipMalloc(&in, n * 8)); HCK(hipMalloc(&out, n * 8)); HCK(hipMalloc(&tmp, n * 8));
        hipLaunchKernelGGL(k_fill64, dim3(4096), dim3(256), 0, 0, in, n, gb);
        const unsigned lo = gb + 36, hi = gb + 54;
        float ms = time_it([&] { (void)bft_rs::sort<uint64_t, bft_rs::NoVal>(bft_rs::PtrIn<uint64_t, bft_rs::NoVal>{in, nullptr}, n, out, (bft_rs::NoVal*)nullptr, tmp, (bft_rs::NoVal*)nullptr, lo, hi, 0, scratch); });
        HCK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL(k_check_split, dim3(4096), dim3(256), 0, 0, out, n, lo, bad);
        unsigned nb = 0;
        HCK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
        printf("{\"what\": \"root-prefix split, 2e8 composites, bits [%u, %u)\", \"impl\": \"bft_rs\", \"ms\": %.3f, \"GBps_40B\": %.0f, \"bad\": %u}\n", lo, hi, ms, n * 40.0 / ms / 1e6, nb);
#ifdef BFT_RS_PROF
        {
            unsigned long long z[2][16] = {{0}}, pr[2][16];
            HCK(hipMemcpyToSymbol(HIP_SYMBOL(bft_rs::g_rs_prof), z, sizeof(z)));
            (void)bft_rs::sort<uint64_t, bft_rs::NoVal>(bft_rs::PtrIn<uint64_t, bft_rs::NoVal>{in, nullptr}, n, out, (bft_rs::NoVal*)nullptr, tmp, (bft_rs::NoVal*)nullptr, lo, hi, 0, scratch);
            HCK(hipDeviceSynchronize());
            HCK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(bft_rs::g_rs_prof), sizeof(pr)));
            const char* nm[12] = {"loop top", "claim publish + zero counters", "rank (+ wait for the loads)", "barrier B", "scan part 1", "barriers D, E + part 2", "publish + look-back loads", "reorder in LDS", "next loads issued", "look-back finish", "barrier G", "write-out"};
            for (int k = 0; k < 2; k++) {
                double tot = 0; for (int i = 0; i < 12; i++) tot += (double)pr[k][i];
                for (int i = 0; i < 12; i++) printf("{\"pass\": \"%s\", \"phase\": \"%s\", \"share\": %.3f, \"Mticks\": %.1f}\n", k ? "chains" : "ranged", nm[i], pr[k][i] / tot, pr[k][i] / 1e6);
            }
        }
#endif
        size_t tb = 0;
        (void)rocprim::radix_sort_keys<Msd9>(nullptr, tb, in, out, (size_t)n, lo, hi, 0);
        void* t2;
        HCK(hipMalloc(&t2, tb));
        ms = time_it([&] { (void)rocprim::radix_sort_keys<Msd9>(t2, tb, in, out, (size_t)n, lo, hi, 0); });
        printf("{\"what\": \"root-prefix split, 2e8 composites\", \"impl\": \"rocprim onesweep 9-bit\", \"ms\": %.3f}\n", ms);
        fflush(stdout);
        hipFree(t2); hipFree(in); hipFree(out); hipFree(tmp);
    }
    {   // the k-mer hash's sort by home line
        const uint64_t n = 44600000ull;
        uint32_t *in, *out, *tmp;
        Rec12 *ri, *ro, *rt;
        HCK(hipMalloc(&in, n * 4)); HCK(hipMalloc(&out, n * 4)); HCK(hipMalloc(&tmp, n * 4));
        HCK(hipMalloc(&ri, n * 12)); HCK(hipMalloc(&ro, n * 12)); HCK(hipMalloc(&rt, n * 12));
        hipLaunchKernelGGL(k_fill_kh, dim3(4096), dim3(256), 0, 0, in, ri, n);
        float ms = time_it([&] { (void)bft_rs::sort<uint32_t, Rec12, bft_rs::PtrIn<uint32_t, Rec12>, bft_rs::SHAPE_BACK>(bft_rs::PtrIn<uint32_t, Rec12>{in, ri}, n, out, ro, tmp, rt, 0, 24, 0, scratch); });
        printf("{\"what\": \"k-mer hash sort, 4.46e7 x (u32 + 12 B), 24 bits\", \"impl\": \"bft_rs back shape\", \"ms\": %.3f}\n", ms);
        ms = time_it([&] { (void)bft_rs::sort<uint32_t, Rec12>(bft_rs::PtrIn<uint32_t, Rec12>{in, ri}, n, out, ro, tmp, rt, 0, 24, 0, scratch); });
        HCK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL(k_check_kh, dim3(4096), dim3(256), 0, 0, out, ro, n, bad);
        unsigned nb = 0;
        HCK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
        printf("{\"what\": \"k-mer hash sort, 4.46e7 x (u32 + 12 B), 24 bits\", \"impl\": \"bft_rs\", \"ms\": %.3f, \"bad\": %u}\n", ms, nb);
        size_t tb = 0;
        (void)rocprim::radix_sort_pairs(nullptr, tb, in, out, ri, ro, (size_t)n, 0u, 24u, 0);
        void* t2;
        HCK(hipMalloc(&t2, tb));
        ms = time_it([&] { (void)rocprim::radix_sort_pairs(t2, tb, in, out, ri, ro, (size_t)n, 0u, 24u, 0); });
        printf("{\"what\": \"k-mer hash sort\", \"impl\": \"rocprim default\", \"ms\": %.3f}\n", ms);
        fflush(stdout);
        hipFree(t2); hipFree(in); hipFree(out); hipFree(tmp); hipFree(ri); hipFree(ro); hipFree(rt);
    }
    {   // scans: 4.46e7 u64 (the k-mer hash's running maximum), 2e8 flags on the fly
        const uint64_t n = 44600000ull;
        uint64_t *in, *out;
        HCK(hipMalloc(&in, n * 8)); HCK(hipMalloc(&out, n * 8));
        hipLaunchKernelGGL(k_fill64, dim3(4096), dim3(256), 0, 0, in, n, 10);
        float ms = time_it([&] { (void)bft_scan::scan<uint64_t, bft_scan::PtrIn<uint64_t>, bft_scan::Max, true>(bft_scan::PtrIn<uint64_t>{in}, out, n, 0ull, bft_scan::Max(), 0, scratch); });
        printf("{\"what\": \"inclusive max-scan, 4.46e7 u64\", \"impl\": \"bft_scan\", \"ms\": %.3f, \"GBps\": %.0f}\n", ms, n * 16.0 / ms / 1e6);
        size_t tb = 0;
        (void)rocprim::inclusive_scan(nullptr, tb, in, out, (size_t)n, rocprim::maximum<uint64_t>(), 0);
        void* t2;
        HCK(hipMalloc(&t2, tb));
        ms = time_it([&] { (void)rocprim::inclusive_scan(t2, tb, in, out, (size_t)n, rocprim::maximum<uint64_t>(), 0); });
        printf("{\"what\": \"inclusive max-scan, 4.46e7 u64\", \"impl\": \"rocprim\", \"ms\": %.3f}\n", ms);
        hipFree(t2); hipFree(in); hipFree(out);
    }
    {   // a small sort: 1.6e6 64-bit keys + u32
        const uint64_t n = 1600000ull;
        uint64_t *in, *out, *tmp;
        uint32_t *vi, *vo, *vt;
        HCK(hipMalloc(&in, n * 8)); HCK(hipMalloc(&out, n * 8)); HCK(hipMalloc(&tmp, n * 8));
        HCK(hipMalloc(&vi, n * 4)); HCK(hipMalloc(&vo, n * 4)); HCK(hipMalloc(&vt, n * 4));
        hipLaunchKernelGGL(k_fill64, dim3(4096), dim3(256), 0, 0, in, n, 10);
        float ms = time_it([&] { (void)bft_rs::sort<uint64_t, uint32_t>(bft_rs::PtrIn<uint64_t, uint32_t>{in, vi}, n, out, vo, tmp, vt, 0, 64, 0, scratch); });
        printf("{\"what\": \"small sort, 1.6e6 x (u64 + u32), 64 bits\", \"impl\": \"bft_rs\", \"ms\": %.3f}\n", ms);
        ms = time_it([&] { (void)bft_rs::sort<uint64_t, uint32_t, bft_rs::PtrIn<uint64_t, uint32_t>, bft_rs::SHAPE_LIGHT>(bft_rs::PtrIn<uint64_t, uint32_t>{in, vi}, n, out, vo, tmp, vt, 0, 64, 0, scratch); });
        printf("{\"what\": \"small sort, 1.6e6 x (u64 + u32), 64 bits\", \"impl\": \"bft_rs light shape\", \"ms\": %.3f}\n", ms);
        ms = time_it([&] { (void)bft_rs::sort<uint64_t, uint32_t, bft_rs::PtrIn<uint64_t, uint32_t>, bft_rs::SHAPE_LIGHT>(bft_rs::PtrIn<uint64_t, uint32_t>{in, vi}, n, out, vo, tmp, vt, 0, 30, 0, scratch); });
        printf("{\"what\": \"small sort, 1.6e6 x (u64 + u32), 30 bits\", \"impl\": \"bft_rs light shape\", \"ms\": %.3f}\n", ms);
        ms = time_it([&] { (void)bft_rs::sort<uint64_t, uint32_t, bft_rs::PtrIn<uint64_t, uint32_t>>(bft_rs::PtrIn<uint64_t, uint32_t>{in, vi}, n, out, vo, tmp, vt, 0, 30, 0, scratch); });
        printf("{\"what\": \"small sort, 1.6e6 x (u64 + u32), 30 bits\", \"impl\": \"bft_rs\", \"ms\": %.3f}\n", ms);
        size_t tb = 0;
        (void)rocprim::radix_sort_pairs(nullptr, tb, in, out, vi, vo, (size_t)n, 0u, 64u, 0);
        void* t2;
        HCK(hipMalloc(&t2, tb));
        ms = time_it([&] { (void)rocprim::radix_sort_pairs(t2, tb, in, out, vi, vo, (size_t)n, 0u, 64u, 0); });
        printf("{\"what\": \"small sort\", \"impl\": \"rocprim default\", \"ms\": %.3f}\n", ms);
        hipFree(t2);
    }
    return 0;
}
