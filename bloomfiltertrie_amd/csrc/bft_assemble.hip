// bft_assemble.hip -- GPU construction of the BFT containers and of the colour-set dictionary.
//
// Bulk counterpart of the reference's per-k-mer insertion work (insertKmer_Node / insertKmer_Node_special
// src/insertNode.c:38-423; transform2CC / insertSP_CC / transform_Filter2n3 src/CC.c:40-1664; annotation
// updates src/retrieveAnnotation.c:232-314), level-synchronous over the sorted T-form table:
//   per depth, for all nodes of that depth at once:
//     k_prefix_flags/scatter   runs of equal 18-bit digit = the node's prefixes; runs of equal r>>4 = Bloom keys
//     k_assign_cc              one workgroup per node: CCs are opened while >= 255 k-mers are unassigned; the
//                              first <= 255 unassigned keys seed the Bloom filter (LDS bitset, atomicOr), every
//                              unassigned key the filter holds is claimed (first-BF-positive rule)
//     radix sort (bft_sort.h)  prefixes grouped by (node, CC), prefix order kept
//     k_runs / k_clusters      CC boundaries, filter2 clusters (runs of equal p_u)
//     k_entries                prefix entries {p_v | count | row-or-child-node}, filter2 bits (atomicOr), child nodes
//     k_ranks                  running rank into each filter2 word
//     k_uc_rows, k_bloom_slice node UC rows; bit-sliced Bloom block of each node
// The arrays are bit-identical to the host restatement bft_index.cpp (tests/test_gpu_build.py).

#include <atomic>
#include <mutex>
#include <vector>

#include "bft_cs_sig.h"
#include "bft_dev.h"
#include "bft_image.h"
#include "bft_index.h"
#include "bft_scan.h"
#include "bft_sort.h"
#include "bft_walk.h"

#define ABLK 256

__global__ void k_pin_ticket(uint64_t* __restrict__ slot, uint64_t v) {
    __threadfence_system();
    *reinterpret_cast<volatile uint64_t*>(slot) = v;
}
__global__ void k_zero_words(uint32_t* __restrict__ p, uint64_t words) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
int bft_zero_async(void* p, size_t bytes, hipStream_t s) {
    if (!bytes) return 0;
    if ((bytes & 3u) || ((uintptr_t)p & 3u)) return bft_fail(BFT_GPU_E_ARG, "internal: bft_zero_async wants whole aligned words");
    const uint64_t words = bytes / 4;
    hipLaunchKernelGGL(k_zero_words, dim3(bft_grid_for((words + 255) / 256)), dim3(256), 0, s, (uint32_t*)p, words);
    HIPCK(hipGetLastError());
    return 0;
}
uint64_t bft_pin_next_ticket() {
    static std::atomic<uint64_t> tickets{0};
    return tickets.fetch_add(1) + 1;
}
int bft_pin_post(PinBlock& pin, hipStream_t s, uint64_t* ticket) {
    if (!pin.p) return bft_fail(BFT_GPU_E_HIP, "hipHostMalloc (pinned counts)");
    *ticket = bft_pin_next_ticket();
    hipLaunchKernelGGL(k_pin_ticket, dim3(1), dim3(1), 0, s, pin.p + PIN_SLOTS, *ticket);
    HIPCK(hipGetLastError());
    return 0;
}
int bft_pin_wait_for(PinBlock& pin, hipStream_t s, uint64_t want) {
    volatile uint64_t* t = pin.p + PIN_SLOTS;
    for (uint32_t i = 1;; i++) {
        if (*t == want) break;
        __builtin_ia32_pause();
        if ((i & 0xFFFu) == 0) {  // every 4096 polls: is the stream still at it?
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) {  // through: the ticket is there
                if (*t == want) break;
                HIPCK(hipStreamSynchronize(s));
                if (*t == want) break;
                return bft_fail(BFT_GPU_E_HIP, "pinned ticket not written");
            }
            if (q != hipErrorNotReady) HIPCK(q);
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
}
int bft_pin_wait(PinBlock& pin, hipStream_t s) {
    uint64_t want = 0;
    CK(bft_pin_post(pin, s, &want));
    return bft_pin_wait_for(pin, s, want);
}

namespace {

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
// (PinBlock, bft_dev.h: counts the host needs come back through a pinned block that kernels write)
__global__ void k_set32(uint32_t* __restrict__ p, uint32_t v) { *p = v; }
__global__ void k_publish(const uint32_t* __restrict__ v, int n, uint64_t* __restrict__ slots) {
    if ((int)threadIdx.x < n) slots[threadIdx.x] = v[threadIdx.x];
}
// the same, and behind the values the block's ticket (bft_pin_wait_for): the launch the host waits for
__global__ void k_publish_ticket(const uint32_t* __restrict__ v, int n, uint64_t* __restrict__ slots, uint64_t* __restrict__ ticket_slot, uint64_t ticket) {
    if ((int)threadIdx.x < n) slots[threadIdx.x] = v[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *reinterpret_cast<volatile uint64_t*>(ticket_slot) = ticket;
}

struct Scan {
    DevBuf tmp;
    hipStream_t s;
    PinBlock pin;
    explicit Scan(hipStream_t st) : s(st) {}
    // out = exclusive sum of in (u32); the total lands in slot `slot` of the pinned block once the stream gets there
    // tail: out[n] = total as well
    int enqueue(const uint32_t* in, uint32_t* out, uint64_t n, int slot, bool tail = false) {
        if (!pin.p) return bft_fail(BFT_GPU_E_HIP, "hipHostMalloc (scan totals)");
        if (n == 0) {  // (the slot is not in flight: wait() precedes every reuse)
            pin.p[slot] = 0;
            if (tail) hipLaunchKernelGGL(k_set32, dim3(1), dim3(1), 0, s, out, 0u);
            return 0;
        }
        CK(bft_scan::exclusive_sum_ptr<uint32_t>(in, out, n, s, tmp, (unsigned long long*)(pin.p + slot), tail));  // (the last tile writes the total)
        return 0;
    }
    // n <= PIN_SLOTS device words -> slots [slot, slot + n).  The launch is made by wait() (it carries the ticket the host polls for: one launch
    // instead of two) or by the next publish(): the words are read when it runs, behind everything enqueued until then.
    const uint32_t* pend_vals = nullptr;
    int pend_n = 0, pend_slot = 0;
    int flush() {
        if (pend_vals) hipLaunchKernelGGL(k_publish, dim3(1), dim3(PIN_SLOTS), 0, s, pend_vals, pend_n, pin.p + pend_slot);
        pend_vals = nullptr;
        return 0;
    }
    int publish(const uint32_t* d_vals, int n, int slot) {
        if (!pin.p) return bft_fail(BFT_GPU_E_HIP, "hipHostMalloc (scan totals)");
        CK(flush());
        pend_vals = d_vals;
        pend_n = n;
        pend_slot = slot;
        return 0;
    }
    int wait() {
        if (pend_vals) {
            if (!pin.p) return bft_fail(BFT_GPU_E_HIP, "hipHostMalloc (scan totals)");
            const uint64_t ticket = bft_pin_next_ticket();
            hipLaunchKernelGGL(k_publish_ticket, dim3(1), dim3(PIN_SLOTS), 0, s, pend_vals, pend_n, pin.p + pend_slot, pin.p + PIN_SLOTS, ticket);
            pend_vals = nullptr;
            HIPCK(hipGetLastError());
            return bft_pin_wait_for(pin, s, ticket);
        }
        HIPCK(hipGetLastError());
        return bft_pin_wait(pin, s);
    }
    uint64_t get(int slot) const { return pin.p[slot]; }
    int run(const uint32_t* in, uint32_t* out, uint64_t n, uint64_t* total) {
        if (!total) {
            if (n == 0) return 0;
            CK(bft_scan::exclusive_sum_ptr<uint32_t>(in, out, n, s, tmp));
            return 0;
        }
        CK(enqueue(in, out, n, 0));
        CK(wait());
        *total = get(0);
        return 0;
    }
};

__device__ __forceinline__ uint32_t find_node(const uint32_t* __restrict__ node_off, uint32_t M, uint32_t j) {
    uint32_t lo = 0, hi = M;  // last m with node_off[m] <= j
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (node_off[mid] <= j) lo = mid;
        else hi = mid;
    }
    return lo;
}

__global__ void k_sizes(const uint32_t* lo, const uint32_t* hi, uint32_t* sz, uint32_t M) {
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) sz[m] = hi[m] - lo[m];
}

// Row j of a depth's active rows starts a new prefix of its node (h) / a new Bloom key (kh): the input of ONE exclusive scan, both counts in a
// word -- h << 31 | kh, either sum below 2^31 --, computed from the table's neighbouring rows as the scan reads them (bft_scan.h takes a functor).
// Flag arrays, two scans and their four arrays of positions were 48 bytes per row; this is 8 (the T-form) in and 8 out.
constexpr uint64_t PF_KMASK = (1ull << 31) - 1ull;
template <int W>
struct PrefixFlagsIn {
    const uint64_t* tk;
    int k, d;
    const uint32_t* nd_lo;
    const uint32_t* node_off;
    uint32_t M;
    __device__ __forceinline__ uint64_t operator()(uint64_t j) const {
        const uint32_t m = M == 1 ? 0u : find_node(node_off, M, (uint32_t)j);
        const uint32_t row = nd_lo[m] + ((uint32_t)j - node_off[m]);
        const uint32_t r = bft_digit<W>(tk + (size_t)row * W, k, d);
        uint64_t h = 1, kh = 1;
        if (row != nd_lo[m]) {
            const uint32_t rp = bft_digit<W>(tk + (size_t)(row - 1) * W, k, d);
            h = r != rp;
            kh = (r >> 4) != (rp >> 4);
        }
        return (h << 31) | kh;
    }
};

// pos[j] = prefixes << 31 | keys in front of row j (pos[A] = the totals): a row whose prefix count grows is a prefix's first, likewise its key
template <int W>
__global__ void k_prefix_scatter(const uint64_t* __restrict__ tk, int k, int d, const uint32_t* __restrict__ nd_lo,
                                 const uint32_t* __restrict__ node_off, uint32_t M, uint32_t A, const uint64_t* __restrict__ pos,
                                 uint32_t* __restrict__ pref_r, uint32_t* __restrict__ pref_row, uint32_t* __restrict__ pref_node,
                                 uint32_t* __restrict__ pref_key, uint32_t* __restrict__ key_val, uint32_t* __restrict__ key_row,
                                 uint32_t* __restrict__ key_node, uint32_t* __restrict__ node_kb) {
    // (a row's own word by one coalesced load; the next row's from the neighbouring lane, by a load only in a wavefront's last lane)
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t j0 = blockIdx.x * blockDim.x; j0 < A; j0 += gridDim.x * blockDim.x) {  // (whole wavefronts take part in the shuffles)
        const uint32_t j = j0 + threadIdx.x;
        const uint64_t p0 = j <= A ? pos[j] : 0ull;
        uint64_t p1 = ((uint64_t)(uint32_t)__shfl_down((uint32_t)(p0 >> 32), 1) << 32) | (uint32_t)__shfl_down((uint32_t)p0, 1);
        if (lane == 63u && j < A) p1 = pos[j + 1];
        if (j >= A) continue;
        const uint32_t p = (uint32_t)(p0 >> 31), kp = (uint32_t)(p0 & PF_KMASK);
        if ((uint32_t)(p1 >> 31) == p) continue;  // (not the first row of a prefix)
        const bool khd = (uint32_t)(p1 & PF_KMASK) != kp;
        const uint32_t m = M == 1 ? 0u : find_node(node_off, M, j);
        const uint32_t row = nd_lo[m] + (j - node_off[m]);
        const uint32_t r = bft_digit<W>(tk + (size_t)row * W, k, d);
        const uint32_t kk = kp + (khd ? 1u : 0u) - 1u;
        pref_r[p] = r;
        pref_row[p] = row;
        pref_node[p] = m;
        pref_key[p] = kk;
        if (khd) {
            key_val[kk] = r >> 4;
            key_row[kk] = row;
            key_node[kk] = m;
            if (row == nd_lo[m]) node_kb[m] = kk;
        }
    }
}

// count of rows under each prefix / key: next start in the same node, else the node's end
__global__ void k_counts(const uint32_t* __restrict__ start, const uint32_t* __restrict__ node, const uint32_t* __restrict__ nd_hi, uint32_t n,
                         uint32_t* __restrict__ cnt) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t m = node[i];
        const uint32_t end = (i + 1 < n && node[i + 1] == m) ? start[i + 1] : nd_hi[m];
        cnt[i] = end - start[i];
    }
}

// format limits and counters, reduced on the device (one atomic per wavefront)
__global__ void k_ncc_limits(const uint32_t* __restrict__ node_ncc, uint32_t M, uint32_t* __restrict__ lim) {
    uint32_t mx = 0;
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) mx = max(mx, node_ncc[m]);
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_down(mx, o));
    if ((threadIdx.x & 63u) == 0 && mx) atomicMax(&lim[0], mx);
    if (blockIdx.x == 0 && threadIdx.x == 0) lim[1] = node_ncc[0];
}
__global__ void k_nb_limits(const uint32_t* __restrict__ cc_nb, uint32_t C, uint32_t* __restrict__ lim) {
    uint32_t mx = 0, sum = 0, big = 0;  // (the prefixes of one depth are fewer than 2^32: they index u32 arrays)
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        const uint32_t v = cc_nb[c];
        mx = max(mx, v);
        sum += v;
        big += v >= BFT_TRESH_SUF_PREF;
    }
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, (uint32_t)__shfl_down(mx, o));
        sum += __shfl_down(sum, o);
        big += __shfl_down(big, o);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (mx) atomicMax(&lim[0], mx);
        if (sum) atomicAdd(&lim[1], sum);
        if (big) atomicAdd(&lim[2], big);
    }
}

__global__ void k_cc_upper(const uint32_t* __restrict__ sz, uint32_t M, uint32_t* __restrict__ ub) {
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) ub[m] = sz[m] / BFT_NB_KMERS_PER_UC + 1u;
}

// One workgroup per node.  node_ccb: first Bloom-bitset slot of the node in cc_bits (an upper-bound layout, see assemble()).
__global__ __launch_bounds__(ABLK) void k_assign_cc(const uint32_t* __restrict__ key_val, const uint32_t* __restrict__ key_cnt,
                                                    const uint32_t* __restrict__ node_kb, const uint32_t* __restrict__ nd_lo,
                                                    const uint32_t* __restrict__ nd_hi, const uint32_t* __restrict__ hashmod,
                                                    int32_t* __restrict__ key_cc, uint32_t* __restrict__ node_ncc,
                                                    const uint32_t* __restrict__ node_ccb, uint32_t* __restrict__ cc_bits, uint32_t M) {
    __shared__ uint32_t bits[48];
    __shared__ uint32_t wsum[ABLK / 64];
    __shared__ uint32_t s_total;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t m = blockIdx.x; m < M; m += gridDim.x) {
        const uint32_t kb = node_kb[m], ke = node_kb[m + 1];
        uint32_t U = nd_hi[m] - nd_lo[m];
        uint32_t ncc = 0;
        while (U >= BFT_NB_KMERS_PER_UC) {
            if (tid < 48) bits[tid] = 0;
            __syncthreads();
            // seeds: the first <= 255 unassigned keys, in prefix order
            uint32_t running = 0;
            for (uint32_t base = kb; base < ke && running < BFT_NB_KMERS_PER_UC; base += ABLK) {
                const uint32_t idx = base + tid;
                const bool un = idx < ke && key_cc[idx] < 0;
                const uint64_t bal = __ballot(un);
                const uint32_t inwave = (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) wsum[wave] = (uint32_t)__builtin_popcountll(bal);
                __syncthreads();
                uint32_t before = 0, total = 0;
                for (uint32_t w = 0; w < ABLK / 64; w++) {
                    if (w < wave) before += wsum[w];
                    total += wsum[w];
                }
                const uint32_t rank = running + before + inwave;
                if (un && rank < BFT_NB_KMERS_PER_UC) {
                    const uint32_t hm = hashmod[key_val[idx]];
                    const uint32_t h1 = hm & 0xFFFFu, h2 = hm >> 16;
                    atomicOr(&bits[h1 >> 5], 1u << (h1 & 31));
                    atomicOr(&bits[h2 >> 5], 1u << (h2 & 31));
                }
                running += total;
                __syncthreads();
            }
            __syncthreads();
            // claim every unassigned key the Bloom filter holds
            uint32_t local = 0;
            for (uint32_t idx = kb + tid; idx < ke; idx += ABLK) {
                if (key_cc[idx] >= 0) continue;
                const uint32_t hm = hashmod[key_val[idx]];
                const uint32_t h1 = hm & 0xFFFFu, h2 = hm >> 16;
                if (((bits[h1 >> 5] >> (h1 & 31)) & 1u) && ((bits[h2 >> 5] >> (h2 & 31)) & 1u)) {
                    key_cc[idx] = (int32_t)ncc;
                    local += key_cnt[idx];
                }
            }
            if (tid == 0) s_total = 0;
            __syncthreads();
            if (local) atomicAdd(&s_total, local);
            __syncthreads();
            U -= s_total;
            if (cc_bits && tid < 48) cc_bits[(size_t)(node_ccb[m] + ncc) * 48 + tid] = bits[tid];
            ncc++;
            __syncthreads();
        }
        if (tid == 0) node_ncc[m] = ncc;
        __syncthreads();
    }
}

// The same for a level that is ONE node (the root: up to 2^14 Bloom keys, two dozen CCs on a pan-genome index): one workgroup of 1024 threads
// with the node's keys in LDS -- both hash positions of a key in one word, bit 31 = assigned -- so a CC costs the workgroup LDS passes and ballots
// instead of two passes of dependent global loads (key_cc -> key_val -> hashmod) by 256 threads.  Same seeds (the first <= 255 unassigned keys
// in prefix order), same claims, same Bloom bitsets.  (A version with the keys in registers, 16 per thread, spilt 1306 VGPRs at the 128 a
// 1024-thread workgroup may use and was slower than the kernel it replaced.)
#define AONE_BLK 1024
#define AONE_MAXKEYS 16384u
__global__ __launch_bounds__(AONE_BLK) void k_assign_cc_one(const uint32_t* __restrict__ key_val, const uint32_t* __restrict__ key_cnt,
                                                            const uint32_t* __restrict__ node_kb, const uint32_t* __restrict__ nd_lo,
                                                            const uint32_t* __restrict__ nd_hi, const uint32_t* __restrict__ hashmod,
                                                            int32_t* __restrict__ key_cc, uint32_t* __restrict__ node_ncc,
                                                            const uint32_t* __restrict__ node_ccb, uint32_t* __restrict__ cc_bits) {
    extern __shared__ uint32_t hm[];  // AONE_MAXKEYS words
    __shared__ uint32_t bits[48];
    __shared__ uint32_t wsum[AONE_BLK / 64];
    __shared__ uint32_t s_total;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t kb = node_kb[0], ke = node_kb[1], nkeys = ke - kb;
    for (uint32_t i = tid; i < nkeys; i += AONE_BLK) hm[i] = hashmod[key_val[kb + i]];  // (positions below 1504: bit 31 is free)
    uint32_t U = nd_hi[0] - nd_lo[0], ncc = 0;
    __syncthreads();
    while (U >= BFT_NB_KMERS_PER_UC) {
        if (tid < 48) bits[tid] = 0;
        if (tid == 0) s_total = 0;
        __syncthreads();
        uint32_t running = 0;
        for (uint32_t base = 0; base < nkeys && running < BFT_NB_KMERS_PER_UC; base += AONE_BLK) {
            const uint32_t i = base + tid;
            const uint32_t v = i < nkeys ? hm[i] : 0x80000000u;
            const bool un = !(v >> 31);
            const uint64_t bal = __ballot(un);
            const uint32_t inwave = (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wsum[wave] = (uint32_t)__builtin_popcountll(bal);
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (uint32_t w = 0; w < AONE_BLK / 64; w++) {
                const uint32_t x = wsum[w];
                if (w < wave) before += x;
                total += x;
            }
            const uint32_t rank = running + before + inwave;
            if (un && rank < BFT_NB_KMERS_PER_UC) {
                const uint32_t h1 = v & 0xFFFFu, h2 = v >> 16;
                atomicOr(&bits[h1 >> 5], 1u << (h1 & 31));
                atomicOr(&bits[h2 >> 5], 1u << (h2 & 31));
            }
            running += total;
            __syncthreads();
        }
        __syncthreads();
        uint32_t local = 0;
        for (uint32_t i = tid; i < nkeys; i += AONE_BLK) {
            const uint32_t v = hm[i];
            if (v >> 31) continue;
            const uint32_t h1 = v & 0xFFFFu, h2 = v >> 16;
            if (((bits[h1 >> 5] >> (h1 & 31)) & 1u) && ((bits[h2 >> 5] >> (h2 & 31)) & 1u)) {
                key_cc[kb + i] = (int32_t)ncc;
                hm[i] = v | 0x80000000u;
                local += key_cnt[kb + i];  // (read once per key: when it is claimed)
            }
        }
        for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
        if (lane == 0 && local) atomicAdd(&s_total, local);
        __syncthreads();
        U -= s_total;
        if (cc_bits && tid < 48) cc_bits[(size_t)(node_ccb[0] + ncc) * 48 + tid] = bits[tid];
        ncc++;
        __syncthreads();
    }
    if (tid == 0) node_ncc[0] = ncc;
}

__global__ void k_sort_keys(const uint32_t* __restrict__ pref_node, const uint32_t* __restrict__ pref_key, const int32_t* __restrict__ key_cc,
                            uint32_t P, uint64_t* __restrict__ skey, uint32_t* __restrict__ iota) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        skey[p] = ((uint64_t)pref_node[p] << 17) | (uint32_t)(key_cc[pref_key[p]] + 1);
        iota[p] = p;
    }
}

// run boundaries of the sorted (node, cc+1) keys: real CCs -> cc_qb/cc_qe, UC pseudo-CC -> uc_qb/uc_qe
__global__ void k_runs(const uint64_t* __restrict__ skey, uint32_t P, const uint32_t* __restrict__ node_ccb, uint32_t* __restrict__ cc_qb,
                       uint32_t* __restrict__ cc_qe, uint32_t* __restrict__ uc_qb, uint32_t* __restrict__ uc_qe) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < P; q += gridDim.x * blockDim.x) {
        const uint64_t k = skey[q];
        const uint32_t m = (uint32_t)(k >> 17), c1 = (uint32_t)(k & 0x1FFFFu);
        const bool first = q == 0 || skey[q - 1] != k, last = q + 1 == P || skey[q + 1] != k;
        if (c1 == 0) {
            if (first) uc_qb[m] = q;
            if (last) uc_qe[m] = q + 1;
        } else {
            const uint32_t ci = node_ccb[m] + c1 - 1;
            if (first) cc_qb[ci] = q;
            if (last) cc_qe[ci] = q + 1;
        }
    }
}

__global__ void k_cc_meta(const uint32_t* __restrict__ cc_qb, const uint32_t* __restrict__ cc_qe, uint32_t C, uint32_t* __restrict__ cc_nb,
                          uint32_t* __restrict__ cc_s, uint32_t* __restrict__ cc_nwords) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        const uint32_t nb = cc_qe[c] - cc_qb[c];
        const uint32_t s = nb >= BFT_TRESH_SUF_PREF ? 4u : 8u;
        cc_nb[c] = nb;
        cc_s[c] = s;
        cc_nwords[c] = ((1u << (18 - s)) + BFT_F2_BITS_PER_WORD - 1) / BFT_F2_BITS_PER_WORD;
    }
}

// per sorted prefix q: cluster-head flag, child-node flag, UC row count
__global__ void k_cluster_flags(const uint64_t* __restrict__ skey, const uint32_t* __restrict__ sp, const uint32_t* __restrict__ pref_r,
                                const uint32_t* __restrict__ pref_cnt, uint32_t P, const uint32_t* __restrict__ node_ccb,
                                const uint32_t* __restrict__ cc_qb, const uint32_t* __restrict__ cc_s, int last_level,
                                uint32_t* __restrict__ chead, uint32_t* __restrict__ pend, uint32_t* __restrict__ ucn) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < P; q += gridDim.x * blockDim.x) {
        const uint64_t k = skey[q];
        const uint32_t m = (uint32_t)(k >> 17), c1 = (uint32_t)(k & 0x1FFFFu);
        const uint32_t p = sp[q];
        if (c1 == 0) {
            chead[q] = 0;
            pend[q] = 0;
            ucn[q] = pref_cnt[p];
            continue;
        }
        const uint32_t ci = node_ccb[m] + c1 - 1, s = cc_s[ci];
        uint32_t h = 1;
        if (q != cc_qb[ci]) h = (pref_r[p] >> s) != (pref_r[sp[q - 1]] >> s);
        chead[q] = h;
        pend[q] = (!last_level && pref_cnt[p] > BFT_NB_KMERS_PER_UC) ? 1u : 0u;
        ucn[q] = 0;
    }
}

__global__ void k_cluster_scatter(const uint32_t* __restrict__ chead, const uint32_t* __restrict__ cidx, uint32_t P, uint32_t* __restrict__ clus_q) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < P; q += gridDim.x * blockDim.x)
        if (chead[q]) clus_q[cidx[q]] = q;
}

// cluster length and its number of entries in child[] (0 for a single-prefix cluster)
__global__ void k_cluster_len(const uint32_t* __restrict__ clus_q, uint32_t Q, const uint64_t* __restrict__ skey, const uint32_t* __restrict__ node_ccb,
                              const uint32_t* __restrict__ cc_qe, uint32_t* __restrict__ clus_len, uint32_t* __restrict__ multi) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < Q; c += gridDim.x * blockDim.x) {
        const uint32_t q = clus_q[c];
        const uint64_t k = skey[q];
        const uint32_t ci = node_ccb[(uint32_t)(k >> 17)] + (uint32_t)(k & 0x1FFFFu) - 1;
        uint32_t end = cc_qe[ci];
        if (c + 1 < Q && clus_q[c + 1] < end) end = clus_q[c + 1];
        const uint32_t len = end - q;
        clus_len[c] = len;
        multi[c] = len > 1 ? len : 0;
    }
}

__global__ void k_cc_headers(const uint32_t* __restrict__ cc_qb, const uint32_t* __restrict__ cc_nb, const uint32_t* __restrict__ cc_s,
                             const uint32_t* __restrict__ cc_f2, const uint32_t* __restrict__ cidx, const uint32_t* __restrict__ cpos, uint32_t C,
                             uint32_t f2_base, uint32_t clus_base, uint32_t child_base, BftCC* __restrict__ out) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        BftCC cc;
        const uint32_t firstc = cidx[cc_qb[c]];
        cc.f2_off = f2_base + cc_f2[c];
        cc.clus_off = clus_base + firstc;
        cc.child_off = child_base + cpos[firstc];
        cc.nb_elem = (uint16_t)cc_nb[c];
        cc.s = (uint8_t)cc_s[c];
        cc.pad0 = 0;
        out[c] = cc;
    }
}

// prefix entries, filter2 bits, child nodes of the next depth
__global__ void k_entries(const uint64_t* __restrict__ skey, const uint32_t* __restrict__ sp, const uint32_t* __restrict__ pref_r,
                          const uint32_t* __restrict__ pref_row, const uint32_t* __restrict__ pref_cnt, uint32_t P,
                          const uint32_t* __restrict__ node_ccb, const uint32_t* __restrict__ cc_qb, const uint32_t* __restrict__ cc_s,
                          const uint32_t* __restrict__ cc_f2, const uint32_t* __restrict__ chead, const uint32_t* __restrict__ cidx,
                          const uint32_t* __restrict__ clus_q, const uint32_t* __restrict__ clus_len, const uint32_t* __restrict__ cpos,
                          const uint32_t* __restrict__ pend, const uint32_t* __restrict__ nrank, int last_level, int rb, uint32_t next_node_base,
                          uint64_t* __restrict__ f2w, uint64_t* __restrict__ clus, uint64_t* __restrict__ child, uint32_t* __restrict__ next_lo,
                          uint32_t* __restrict__ next_hi) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < P; q += gridDim.x * blockDim.x) {
        const uint64_t k = skey[q];
        const uint32_t c1 = (uint32_t)(k & 0x1FFFFu);
        if (c1 == 0) continue;
        const uint32_t ci = node_ccb[(uint32_t)(k >> 17)] + c1 - 1, s = cc_s[ci];
        const uint32_t p = sp[q], r = pref_r[p], cnt = pref_cnt[p], row = pref_row[p];
        const uint32_t pu = r >> s, pv = r & ((1u << s) - 1u);
        uint64_t ent = (uint64_t)pv << BFT_CHILD_PV_SHIFT;
        if (last_level && rb == 0) ent |= (1ull << BFT_CHILD_CNT_SHIFT) | row;
        else if (last_level) ent = BFT_REM_ENTRY(pv, cnt, row);
        else if (!pend[q]) ent |= ((uint64_t)cnt << BFT_CHILD_CNT_SHIFT) | row;
        else {
            const uint32_t nn = nrank[q];
            ent |= (uint64_t)(next_node_base + nn);
            next_lo[nn] = row;
            next_hi[nn] = row + cnt;
        }
        const uint32_t c = cidx[q] + chead[q] - 1;  // cluster of q
        const uint32_t len = clus_len[c];
        const uint32_t firstc = cidx[cc_qb[ci]];
        if (len == 1) clus[c] = ent;
        else {
            child[cpos[c] + (q - clus_q[c])] = ent;
            if (chead[q]) clus[c] = BFT_CLUS_MULTI | ((uint64_t)len << BFT_CLUS_LEN_SHIFT) | (uint64_t)(cpos[c] - cpos[firstc]);
        }
        if (chead[q]) atomicOr((unsigned long long*)&f2w[cc_f2[ci] + pu / BFT_F2_BITS_PER_WORD], 1ull << (pu % BFT_F2_BITS_PER_WORD));
    }
}

__global__ void k_ranks(const uint32_t* __restrict__ cc_f2, const uint32_t* __restrict__ cc_nwords, uint32_t C, uint64_t* __restrict__ f2w) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        uint64_t* f2 = f2w + cc_f2[c];
        uint32_t rank = 0;
        for (uint32_t w = 0; w < cc_nwords[c]; w++) {
            const uint32_t pc = (uint32_t)__builtin_popcountll(f2[w]);
            f2[w] |= (uint64_t)rank << 48;
            rank += pc;
        }
    }
}

__global__ void k_uc_rows(const uint64_t* __restrict__ tk, int W, const uint64_t* __restrict__ skey, const uint32_t* __restrict__ sp,
                          const uint32_t* __restrict__ pref_row, const uint32_t* __restrict__ pref_cnt, const uint32_t* __restrict__ ucpos, uint32_t P,
                          uint32_t uc_base_unused, uint64_t* __restrict__ uck, uint32_t* __restrict__ ucrow) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < P; q += gridDim.x * blockDim.x) {
        if ((uint32_t)(skey[q] & 0x1FFFFu) != 0) continue;
        const uint32_t p = sp[q], row0 = pref_row[p], cnt = pref_cnt[p], o = ucpos[q];
        for (uint32_t i = 0; i < cnt; i++) {
            for (int w = 0; w < W; w++) uck[(size_t)(o + i) * W + w] = tk[(size_t)(row0 + i) * W + w];
            ucrow[o + i] = row0 + i;
        }
    }
}

__global__ void k_node_meta(const uint32_t* __restrict__ node_ncc, const uint32_t* __restrict__ uc_qb, const uint32_t* __restrict__ uc_qe,
                            const uint32_t* __restrict__ ucpos, const uint32_t* __restrict__ ucn, uint32_t M, uint32_t* __restrict__ node_ucn,
                            uint32_t* __restrict__ node_bf8) {
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        uint32_t n = 0;
        if (uc_qe[m] > uc_qb[m]) n = ucpos[uc_qe[m] - 1] + ucn[uc_qe[m] - 1] - ucpos[uc_qb[m]];
        node_ucn[m] = n;
        const uint32_t ncc = node_ncc[m];
        const uint32_t wb = ncc == 0 ? 0 : ncc <= 8 ? 1 : ncc <= 16 ? 2 : ncc <= 32 ? 4 : 8 * ((ncc + 63) / 64);
        node_bf8[m] = (BFT_MODULO_HASH * wb) / 8;
    }
}

__global__ void k_node_records(const uint32_t* __restrict__ node_ncc, const uint32_t* __restrict__ node_ccb, const uint32_t* __restrict__ node_ucn,
                               const uint32_t* __restrict__ node_ucoff, const uint32_t* __restrict__ node_bfoff, uint32_t M, uint32_t cc_base,
                               uint32_t uc_base, uint32_t bf_base8, BftNode* __restrict__ out) {
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        BftNode nd;
        const uint32_t ncc = node_ncc[m];
        const uint32_t wb = ncc == 0 ? 0 : ncc <= 8 ? 1 : ncc <= 16 ? 2 : ncc <= 32 ? 4 : 8 * ((ncc + 63) / 64);
        nd.cc_first = cc_base + node_ccb[m];
        nd.bf_off = ncc ? bf_base8 + node_bfoff[m] : 0;
        nd.uc_first = uc_base + node_ucoff[m];
        nd.ncc = (uint16_t)ncc;
        nd.uc_n = (uint8_t)node_ucn[m];
        nd.bf_wb = (uint8_t)wb;
        out[m] = nd;
    }
}

// one workgroup per node: transpose the CC bitsets into the bit-sliced block
__global__ __launch_bounds__(ABLK) void k_bloom_slice(const uint32_t* __restrict__ node_ncc, const uint32_t* __restrict__ node_ccb,
                                                      const uint32_t* __restrict__ node_bfoff, const uint32_t* __restrict__ cc_bits, uint32_t M,
                                                      uint8_t* __restrict__ bfT) {
    for (uint32_t m = blockIdx.x; m < M; m += gridDim.x) {
        const uint32_t ncc = node_ncc[m];
        if (ncc == 0) continue;
        const uint32_t wb = ncc <= 8 ? 1 : ncc <= 16 ? 2 : ncc <= 32 ? 4 : 8 * ((ncc + 63) / 64);
        uint8_t* blk = bfT + (size_t)node_bfoff[m] * 8;
        const uint32_t* bits = cc_bits + (size_t)node_ccb[m] * 48;
        for (uint32_t h = threadIdx.x; h < BFT_MODULO_HASH; h += ABLK) {
            for (uint32_t b = 0; b < wb; b++) {
                uint32_t v = 0;
                for (uint32_t c = b * 8; c < ncc && c < b * 8 + 8; c++) v |= ((bits[(size_t)c * 48 + (h >> 5)] >> (h & 31)) & 1u) << (c & 7);
                blk[(size_t)h * wb + b] = (uint8_t)v;
            }
        }
    }
}

// the per-depth segments into the final arrays (every size is a multiple of 4 bytes)
struct ConcatJob { uint32_t* dst; const uint32_t* src; uint64_t words; };
struct ConcatJobs { ConcatJob j[8 * 14]; int n; };  // (k <= 126: 14 levels at most, 8 arrays each)
__global__ __launch_bounds__(ABLK) void k_concat(const ConcatJobs jobs) {
    const ConcatJob j = jobs.j[blockIdx.y];
    for (uint64_t i = blockIdx.x * (uint64_t)ABLK + threadIdx.x; i < j.words; i += (uint64_t)gridDim.x * ABLK) j.dst[i] = j.src[i];
}

// several arrays set to a value each in one launch (every size is a multiple of 4 bytes; what a depth would otherwise zero by a memset per array)
struct FillJob { uint32_t* p; uint64_t words; uint32_t val; };
struct FillJobs {
    FillJob j[8];
    int n = 0;
    void add(DevBuf& b, uint32_t val = 0u) { add(b.as<uint32_t>(), b.bytes / 4, val); }
    void add(uint32_t* p, uint64_t words, uint32_t val) { if (words) j[n++] = FillJob{p, words, val}; }
};
__global__ __launch_bounds__(ABLK) void k_fill_many(const FillJobs jobs) {
    const FillJob j = jobs.j[blockIdx.y];
    for (uint64_t i = blockIdx.x * (uint64_t)ABLK + threadIdx.x; i < j.words; i += (uint64_t)gridDim.x * ABLK) j.p[i] = j.val;
}
static void fill_many(const FillJobs& jobs, hipStream_t s) {
    if (!jobs.n) return;
    uint64_t mx = 1;
    for (int i = 0; i < jobs.n; i++) mx = std::max(mx, jobs.j[i].words);
    hipLaunchKernelGGL(k_fill_many, dim3(bft_grid_for((mx + ABLK - 1) / ABLK), (unsigned)jobs.n), dim3(ABLK), 0, s, jobs);
}

struct Seg {  // per-depth output segments, concatenated at the end
    DevBuf nodes, bfT, ccs, f2w, clus, child, uck, ucrow;
    uint64_t n_nodes = 0, n_bf8 = 0, n_ccs = 0, n_f2w = 0, n_clus = 0, n_child = 0, n_uc = 0;
};

template <int W>
int assemble(const uint64_t* tk, uint64_t n, int k, const uint32_t* hashmod, hipStream_t s, BftDeviceIndex& out, const BftAssembleHook* hook) {
    const int L = k / 9, rb = 2 * (k - 9 * L);
    Scan scan(s);
    std::vector<Seg> segs;
    DevBuf nd_lo, nd_hi;
    CK(nd_lo.alloc(4));
    CK(nd_hi.alloc(4));
    hipLaunchKernelGGL(k_set32, dim3(1), dim3(1), 0, s, nd_lo.as<uint32_t>(), 0u);  // (the root: every row)
    hipLaunchKernelGGL(k_set32, dim3(1), dim3(1), 0, s, nd_hi.as<uint32_t>(), (uint32_t)n);
    uint64_t M = 1;
    uint64_t T_nodes = 0, T_ccs = 0, T_f2w = 0, T_clus = 0, T_child = 0, T_bf8 = 0, T_uc = 0;
#define G(nelem) dim3(bft_grid_for(((uint64_t)(nelem) + ABLK - 1) / ABLK)), dim3(ABLK), 0, s
    // Waits for counts per depth: prefixes + keys, CCs, clusters + child nodes, Bloom blocks (+ the child entries and the NEXT depth's active rows,
    // read with that last one: their arrays are sized by an upper bound / prepared a depth early) -- four, where rounds 3-4 took six or seven.
    DevBuf nsz_c, node_off_c;  // the next depth's node sizes and their scan, prepared behind this depth's entries
    uint64_t A_c = 0;
    for (int d = 0; d < L && M > 0; d++) {
        const int last_level = d == L - 1;
        Seg sg;
        // ---- active rows ----
        DevBuf nsz, node_off;
        uint64_t A = n;  // (the root's rows are all of them)
        if (d == 0) {
            CK(nsz.alloc(M * 4));
            CK(node_off.alloc((M + 1) * 4));
            hipLaunchKernelGGL(k_sizes, G(M), nd_lo.as<uint32_t>(), nd_hi.as<uint32_t>(), nsz.as<uint32_t>(), (uint32_t)M);
            CK(scan.enqueue(nsz.as<uint32_t>(), node_off.as<uint32_t>(), M, 0, true));  // (nothing to wait for; the slot is written again only behind this scan)
        } else {
            nsz.swap(nsz_c);
            node_off.swap(node_off_c);
            A = A_c;
        }
        bft_trace_mark("  level: active rows");
        const std::string lv = "containers depth " + std::to_string(d) + ": ";
        bft_stage((lv + "active rows").c_str(), (double)M * 12, s);
        // ---- prefixes and keys ----
        DevBuf pos;  // [A + 1] prefixes << 31 | keys in front of every active row
        CK(pos.alloc((A + 1) * 8));
        uint64_t P = 0, K = 0;
        if (A) {
            if (!scan.pin.p) return bft_fail(BFT_GPU_E_HIP, "hipHostMalloc (scan totals)");
            const PrefixFlagsIn<W> pf{tk, k, d, nd_lo.as<uint32_t>(), node_off.as<uint32_t>(), (uint32_t)M};
            CK((bft_scan::exclusive_sum<uint64_t>(pf, pos.as<uint64_t>(), A, s, scan.tmp, (unsigned long long*)(scan.pin.p + 0), true)));
            CK(scan.wait());
            P = scan.get(0) >> 31;
            K = scan.get(0) & PF_KMASK;
        }
        bft_trace_mark("  level: prefix flags + scans");
        bft_stage((lv + "prefix flags + scans").c_str(), (double)A * (8.0 * W + 8), s);
        DevBuf pref_r, pref_row, pref_node, pref_key, pref_cnt, key_val, key_row, key_node, key_cnt, node_kb;
        CK(pref_r.alloc(P * 4));
        CK(pref_row.alloc(P * 4));
        CK(pref_node.alloc(P * 4));
        CK(pref_key.alloc(P * 4));
        CK(pref_cnt.alloc(P * 4));
        CK(key_val.alloc(K * 4));
        CK(key_row.alloc(K * 4));
        CK(key_node.alloc(K * 4));
        CK(key_cnt.alloc(K * 4));
        CK(node_kb.alloc((M + 1) * 4));
        // (allocated here so that one launch presets them all) CC assignment: key -> CC (-1: unassigned); the format's limits
        DevBuf key_cc, lim;  // lim: [0] largest number of CCs in a node, [1] that of node 0, [2] largest CC, [3] prefixes, [4] CCs with >= BFT_TRESH_SUF_PREF, [5] UC rows of node 0
        CK(key_cc.alloc(K * 4));
        CK(lim.alloc(8 * 4));
        {
            FillJobs fj;
            fj.add(node_kb.as<uint32_t>(), M, 0u);  // nodes of an empty trie (n == 0) have no keys: node_kb = 0
            fj.add(node_kb.as<uint32_t>() + M, 1, (uint32_t)K);
            if (K) fj.add(key_cc.as<uint32_t>(), K, 0xFFFFFFFFu);
            fj.add(lim.as<uint32_t>(), 8, 0u);
            fill_many(fj, s);
        }
        if (A) {
            hipLaunchKernelGGL(k_prefix_scatter<W>, G(A), tk, k, d, nd_lo.as<uint32_t>(), node_off.as<uint32_t>(), (uint32_t)M, (uint32_t)A,
                               pos.as<uint64_t>(), pref_r.as<uint32_t>(),
                               pref_row.as<uint32_t>(), pref_node.as<uint32_t>(), pref_key.as<uint32_t>(), key_val.as<uint32_t>(),
                               key_row.as<uint32_t>(), key_node.as<uint32_t>(), node_kb.as<uint32_t>());
            hipLaunchKernelGGL(k_counts, G(P), pref_row.as<uint32_t>(), pref_node.as<uint32_t>(), nd_hi.as<uint32_t>(), (uint32_t)P, pref_cnt.as<uint32_t>());
            hipLaunchKernelGGL(k_counts, G(K), key_row.as<uint32_t>(), key_node.as<uint32_t>(), nd_hi.as<uint32_t>(), (uint32_t)K, key_cnt.as<uint32_t>());
        }
        pos.release();

        // ---- CC assignment: one pass.  A node opens a CC only while >= 255 k-mers are unassigned and every CC but the last claims at
        // least 255, so a node of U k-mers holds at most U / 255 + 1 CCs: the Bloom bitsets are written at those upper-bound slots
        // (ubb) and the real CC numbering comes from a scan of the counts afterwards.  (A counting pass used to run first: the same
        // kernel twice, 3.0 ms of config 3's assembly.) ----
        DevBuf node_ncc, node_ccb, cc_bits, ub, ubb;
        CK(node_ncc.alloc(M * 4));
        CK(node_ccb.alloc((M + 1) * 4));
        CK(ub.alloc(M * 4));
        CK(ubb.alloc((M + 1) * 4));
        hipLaunchKernelGGL(k_cc_upper, G(M), nsz.as<uint32_t>(), (uint32_t)M, ub.as<uint32_t>());
        CK(scan.run(ub.as<uint32_t>(), ubb.as<uint32_t>(), M, nullptr));
        CK(cc_bits.alloc((A / BFT_NB_KMERS_PER_UC + M) * 48 * 4));  // (the sum of the upper bounds is at most that: no count to wait for)
        const dim3 ngrid((unsigned)std::min<uint64_t>(M, 65535ull * 16));
        if (M == 1 && K <= AONE_MAXKEYS) {  // (one node: its keys in the LDS of one large workgroup)
            static std::atomic<uint64_t> attr_devs{0};  // the attribute is per device: one bit per device it was set on
            int dev = 0;
            HIPCK(hipGetDevice(&dev));
            const uint64_t dev_bit = 1ull << (dev & 63);
            if (!(attr_devs.load(std::memory_order_acquire) & dev_bit)) {
                HIPCK(hipFuncSetAttribute((const void*)k_assign_cc_one, hipFuncAttributeMaxDynamicSharedMemorySize, AONE_MAXKEYS * 4));
                attr_devs.fetch_or(dev_bit, std::memory_order_release);
            }
            hipLaunchKernelGGL(k_assign_cc_one, dim3(1), dim3(AONE_BLK), AONE_MAXKEYS * 4, s, key_val.as<uint32_t>(), key_cnt.as<uint32_t>(), node_kb.as<uint32_t>(),
                               nd_lo.as<uint32_t>(), nd_hi.as<uint32_t>(), hashmod, key_cc.as<int32_t>(), node_ncc.as<uint32_t>(),
                               ubb.as<uint32_t>(), cc_bits.as<uint32_t>());
        } else
        hipLaunchKernelGGL(k_assign_cc, ngrid, dim3(ABLK), 0, s, key_val.as<uint32_t>(), key_cnt.as<uint32_t>(), node_kb.as<uint32_t>(),
                           nd_lo.as<uint32_t>(), nd_hi.as<uint32_t>(), hashmod, key_cc.as<int32_t>(), node_ncc.as<uint32_t>(),
                           ubb.as<uint32_t>(), cc_bits.as<uint32_t>(), (uint32_t)M);
        // limits (format): CCs per node, Bloom slice width
        hipLaunchKernelGGL(k_ncc_limits, G(M), node_ncc.as<uint32_t>(), (uint32_t)M, lim.as<uint32_t>());
        CK(scan.enqueue(node_ncc.as<uint32_t>(), node_ccb.as<uint32_t>(), M, 0, true));
        CK(scan.publish(lim.as<uint32_t>(), 2, 1));
        CK(scan.wait());
        const uint64_t C = scan.get(0);
        if (scan.get(1) > 2040) return bft_fail(BFT_GPU_E_LIMIT, "node with too many CCs for bf_wb");
        out.max_ccs_per_node = std::max<uint64_t>(out.max_ccs_per_node, scan.get(1));
        if (d == 0) out.root_ncc = scan.get(2);
        bft_trace_mark("  level: scatter, CC assignment");
        bft_stage((lv + "prefix scatter, CC assignment").c_str(), (double)A * (8.0 * W + 16) + (double)P * 24 + (double)K * 24, s);
        // (the passes over the whole table and the root's CC assignment -- ONE workgroup claiming CC after CC, bound by latency: 0.8 ms
        // alone, 3.5 ms beside a kernel that saturates the memory system -- are done: what follows is a chain of small kernels and
        // counts read back, which leaves most of the GPU to whatever the caller starts beside it now)
        if (d == 0 && hook && hook->after_table_passes) hook->after_table_passes(hook->ctx, s);

        // ---- prefixes grouped by (node, cc) ----
        DevBuf skey, skey_s, iota, sp;
        CK(skey.alloc(P * 8));
        CK(skey_s.alloc(P * 8));
        CK(iota.alloc(P * 4));
        CK(sp.alloc(P * 4));
        if (P) {
            hipLaunchKernelGGL(k_sort_keys, G(P), pref_node.as<uint32_t>(), pref_key.as<uint32_t>(), key_cc.as<int32_t>(), (uint32_t)P,
                               skey.as<uint64_t>(), iota.as<uint32_t>());
            int mbits = 1;
            while (mbits < 32 && (M >> mbits)) mbits++;
            CK((bft_rs::sort_pairs<uint64_t, uint32_t, bft_rs::SHAPE_LIGHT>(skey.as<uint64_t>(), iota.as<uint32_t>(), P, skey_s.as<uint64_t>(), sp.as<uint32_t>(), 0, 17 + mbits, s)));
        }  // (no synchronisation: what is released here is only handed out again in the order of this stream, bft_pool_alloc)
        skey.release(); iota.release();

        // ---- CC runs, clusters ----
        DevBuf cc_qb, cc_qe, uc_qb, uc_qe, cc_nb, cc_s, cc_nwords, cc_f2;
        CK(cc_qb.alloc(C * 4));
        CK(cc_qe.alloc(C * 4));
        CK(uc_qb.alloc(M * 4));
        CK(uc_qe.alloc(M * 4));
        {
            FillJobs fj;
            fj.add(cc_qb.as<uint32_t>(), C, 0u);
            fj.add(cc_qe.as<uint32_t>(), C, 0u);
            fj.add(uc_qb.as<uint32_t>(), M, 0u);
            fj.add(uc_qe.as<uint32_t>(), M, 0u);
            fill_many(fj, s);
        }
        CK(cc_nb.alloc(C * 4));
        CK(cc_s.alloc(C * 4));
        CK(cc_nwords.alloc(C * 4));
        CK(cc_f2.alloc(C * 4));
        if (P) hipLaunchKernelGGL(k_runs, G(P), skey_s.as<uint64_t>(), (uint32_t)P, node_ccb.as<uint32_t>(), cc_qb.as<uint32_t>(), cc_qe.as<uint32_t>(),
                                  uc_qb.as<uint32_t>(), uc_qe.as<uint32_t>());
        if (C) {
            hipLaunchKernelGGL(k_cc_meta, G(C), cc_qb.as<uint32_t>(), cc_qe.as<uint32_t>(), (uint32_t)C, cc_nb.as<uint32_t>(), cc_s.as<uint32_t>(),
                               cc_nwords.as<uint32_t>());
            hipLaunchKernelGGL(k_nb_limits, G(C), cc_nb.as<uint32_t>(), (uint32_t)C, lim.as<uint32_t>() + 2);
        }
        CK(scan.enqueue(cc_nwords.as<uint32_t>(), cc_f2.as<uint32_t>(), C, 0));  // (read with the cluster counts below)
        DevBuf chead, pend, ucn, cidx, nrank, ucpos;
        CK(chead.alloc(P * 4));
        CK(pend.alloc(P * 4));
        CK(ucn.alloc(P * 4));
        CK(cidx.alloc(P * 4));
        CK(nrank.alloc(P * 4));
        CK(ucpos.alloc(P * 4));
        if (P) {
            hipLaunchKernelGGL(k_cluster_flags, G(P), skey_s.as<uint64_t>(), sp.as<uint32_t>(), pref_r.as<uint32_t>(), pref_cnt.as<uint32_t>(), (uint32_t)P,
                               node_ccb.as<uint32_t>(), cc_qb.as<uint32_t>(), cc_s.as<uint32_t>(), last_level, chead.as<uint32_t>(), pend.as<uint32_t>(),
                               ucn.as<uint32_t>());
        }
        CK(scan.enqueue(chead.as<uint32_t>(), cidx.as<uint32_t>(), P, 1));
        CK(scan.enqueue(pend.as<uint32_t>(), nrank.as<uint32_t>(), P, 2));
        CK(scan.enqueue(ucn.as<uint32_t>(), ucpos.as<uint32_t>(), P, 3));
        CK(scan.publish(lim.as<uint32_t>() + 2, 3, 4));
        CK(scan.wait());
        const uint64_t F2 = scan.get(0), Q = scan.get(1), Mnext = scan.get(2), UCR = scan.get(3);
        if (scan.get(4) > 65535) return bft_fail(BFT_GPU_E_LIMIT, "CC with more than 65535 prefixes (nb_elem is uint16, include/CC.h:36)");
        out.n_prefixes += scan.get(5);
        out.n_ccs_s4 += scan.get(6);
        bft_trace_mark("  level: sort by (node, CC), runs, cluster flags");
        bft_stage((lv + "sort by (node, CC), runs, cluster flags").c_str(), (double)P * (12.0 * 2 * 4 + 60), s);
        DevBuf clus_q, clus_len, multi, cpos;
        CK(clus_q.alloc(Q * 4));
        CK(clus_len.alloc(Q * 4));
        CK(multi.alloc(Q * 4));
        CK(cpos.alloc(Q * 4));
        if (Q) {
            hipLaunchKernelGGL(k_cluster_scatter, G(P), chead.as<uint32_t>(), cidx.as<uint32_t>(), (uint32_t)P, clus_q.as<uint32_t>());
            hipLaunchKernelGGL(k_cluster_len, G(Q), clus_q.as<uint32_t>(), (uint32_t)Q, skey_s.as<uint64_t>(), node_ccb.as<uint32_t>(), cc_qe.as<uint32_t>(),
                               clus_len.as<uint32_t>(), multi.as<uint32_t>());
        }
        // (E, the entries of the multi-prefix clusters, is read with the depth's last counts: every prefix lies in one cluster, so P bounds it)
        CK(scan.enqueue(multi.as<uint32_t>(), cpos.as<uint32_t>(), Q, 2));
        bft_trace_mark("  level: clusters");
        bft_stage((lv + "clusters").c_str(), (double)P * 8 + (double)Q * 24, s);
        if (T_f2w + F2 > 0xFFFFFFFFull || T_clus + Q > 0xFFFFFFFFull || T_uc + UCR > 0xFFFFFFFFull)
            return bft_fail(BFT_GPU_E_LIMIT, "index array offset overflow (u32)");

        // ---- outputs of this depth ----
        CK(sg.ccs.alloc(C * sizeof(BftCC)));
        CK(sg.f2w.alloc_zero(F2 * 8, s));
        CK(sg.clus.alloc(Q * 8));
        CK(sg.child.alloc(P * 8));  // (upper bound; E of them are used and concatenated)
        CK(sg.uck.alloc(UCR * W * 8));
        CK(sg.ucrow.alloc(UCR * 4));
        CK(sg.nodes.alloc(M * sizeof(BftNode)));
        DevBuf next_lo, next_hi;
        CK(next_lo.alloc(Mnext * 4));
        CK(next_hi.alloc(Mnext * 4));
        if (C) {
            // cc_f2 is relative to this depth's f2w segment while filling; headers carry global offsets
            hipLaunchKernelGGL(k_cc_headers, G(C), cc_qb.as<uint32_t>(), cc_nb.as<uint32_t>(), cc_s.as<uint32_t>(), cc_f2.as<uint32_t>(), cidx.as<uint32_t>(),
                               cpos.as<uint32_t>(), (uint32_t)C, (uint32_t)T_f2w, (uint32_t)T_clus, (uint32_t)T_child, sg.ccs.as<BftCC>());
            hipLaunchKernelGGL(k_entries, G(P), skey_s.as<uint64_t>(), sp.as<uint32_t>(), pref_r.as<uint32_t>(), pref_row.as<uint32_t>(), pref_cnt.as<uint32_t>(),
                               (uint32_t)P, node_ccb.as<uint32_t>(), cc_qb.as<uint32_t>(), cc_s.as<uint32_t>(), cc_f2.as<uint32_t>(), chead.as<uint32_t>(),
                               cidx.as<uint32_t>(), clus_q.as<uint32_t>(), clus_len.as<uint32_t>(), cpos.as<uint32_t>(), pend.as<uint32_t>(),
                               nrank.as<uint32_t>(), last_level, rb, (uint32_t)(T_nodes + M), sg.f2w.as<uint64_t>(), sg.clus.as<uint64_t>(),
                               sg.child.as<uint64_t>(), next_lo.as<uint32_t>(), next_hi.as<uint32_t>());
            hipLaunchKernelGGL(k_ranks, G(C), cc_f2.as<uint32_t>(), cc_nwords.as<uint32_t>(), (uint32_t)C, sg.f2w.as<uint64_t>());
        }
        if (UCR) hipLaunchKernelGGL(k_uc_rows, G(P), tk, W, skey_s.as<uint64_t>(), sp.as<uint32_t>(), pref_row.as<uint32_t>(), pref_cnt.as<uint32_t>(),
                                    ucpos.as<uint32_t>(), (uint32_t)P, 0u, sg.uck.as<uint64_t>(), sg.ucrow.as<uint32_t>());
        // nodes
        DevBuf node_ucn, node_bf8, node_ucoff, node_bfoff;
        CK(node_ucn.alloc(M * 4));
        CK(node_bf8.alloc(M * 4));
        CK(node_ucoff.alloc(M * 4));
        CK(node_bfoff.alloc(M * 4));
        hipLaunchKernelGGL(k_node_meta, G(M), node_ncc.as<uint32_t>(), uc_qb.as<uint32_t>(), uc_qe.as<uint32_t>(), ucpos.as<uint32_t>(), ucn.as<uint32_t>(),
                           (uint32_t)M, node_ucn.as<uint32_t>(), node_bf8.as<uint32_t>());
        CK(scan.run(node_ucn.as<uint32_t>(), node_ucoff.as<uint32_t>(), M, nullptr));
        CK(scan.enqueue(node_bf8.as<uint32_t>(), node_bfoff.as<uint32_t>(), M, 0));
        if (d == 0) CK(scan.publish(node_ucn.as<uint32_t>(), 1, 1));
        if (Mnext && !last_level) {  // the next depth's node sizes and active rows, a depth early (next_lo / next_hi are k_entries' output)
            CK(nsz_c.alloc(Mnext * 4));
            CK(node_off_c.alloc((Mnext + 1) * 4));
            hipLaunchKernelGGL(k_sizes, G(Mnext), next_lo.as<uint32_t>(), next_hi.as<uint32_t>(), nsz_c.as<uint32_t>(), (uint32_t)Mnext);
            CK(scan.enqueue(nsz_c.as<uint32_t>(), node_off_c.as<uint32_t>(), Mnext, 3, true));
        }
        CK(scan.wait());
        const uint64_t BF8 = scan.get(0), E = scan.get(2);
        A_c = (Mnext && !last_level) ? scan.get(3) : 0;
        if (d == 0) out.root_uc = scan.get(1);
        if (E > P) return bft_fail(BFT_GPU_E_LIMIT, "assembly: more cluster entries than prefixes");  // (cannot happen: the bound above)
        if (T_child + E > 0xFFFFFFFFull) return bft_fail(BFT_GPU_E_LIMIT, "index array offset overflow (u32)");
        bft_trace_mark("  level: entries, node records");
        bft_stage((lv + "entries, ranks, UC rows, node records").c_str(), (double)P * 60 + (double)(F2 + Q + E) * 8 + (double)UCR * (8.0 * W + 4), s);
        if (T_bf8 + BF8 > 0xFFFFFFFFull) return bft_fail(BFT_GPU_E_LIMIT, "Bloom block offset overflow");
        CK(sg.bfT.alloc(BF8 * 8));
        hipLaunchKernelGGL(k_node_records, G(M), node_ncc.as<uint32_t>(), node_ccb.as<uint32_t>(), node_ucn.as<uint32_t>(), node_ucoff.as<uint32_t>(),
                           node_bfoff.as<uint32_t>(), (uint32_t)M, (uint32_t)T_ccs, (uint32_t)T_uc, (uint32_t)T_bf8, sg.nodes.as<BftNode>());
        if (BF8) hipLaunchKernelGGL(k_bloom_slice, ngrid, dim3(ABLK), 0, s, node_ncc.as<uint32_t>(), ubb.as<uint32_t>(), node_bfoff.as<uint32_t>(),
                                    cc_bits.as<uint32_t>(), (uint32_t)M, sg.bfT.as<uint8_t>());
        HIPCK(hipGetLastError());
        sg.n_nodes = M; sg.n_bf8 = BF8; sg.n_ccs = C; sg.n_f2w = F2; sg.n_clus = Q; sg.n_child = E; sg.n_uc = UCR;
        T_nodes += M; T_ccs += C; T_f2w += F2; T_clus += Q; T_child += E; T_bf8 += BF8; T_uc += UCR;
        out.n_child_nodes += Mnext;
        segs.push_back(std::move(sg));
        nd_lo.swap(next_lo);
        nd_hi.swap(next_hi);
        M = Mnext;
    }
#undef G
    // ---- concatenate the per-depth segments ----
    CK(out.nodes.alloc(T_nodes * sizeof(BftNode)));
    CK(out.bfT.alloc(T_bf8 * 8));
    CK(out.ccs.alloc(T_ccs * sizeof(BftCC)));
    CK(out.f2w.alloc(T_f2w * 8));
    CK(out.clus.alloc(T_clus * 8));
    CK(out.child.alloc(T_child * 8));
    CK(out.uck.alloc(T_uc * W * 8));
    CK(out.ucrow.alloc(T_uc * 4));
    // one kernel for every (depth, array) piece: up to 8 L stream-ordered copies of a few microseconds each cost their launch gaps
    uint64_t o_n = 0, o_b = 0, o_c = 0, o_f = 0, o_q = 0, o_e = 0, o_u = 0;
    ConcatJobs jobs;
    jobs.n = 0;
    uint64_t most = 0;
    for (Seg& g : segs) {
#define CP(dst, src, off, nbytes) if (nbytes) { jobs.j[jobs.n++] = ConcatJob{(uint32_t*)((uint8_t*)(dst).p + (off)), (const uint32_t*)(src).p, (uint64_t)(nbytes) / 4}; most = std::max<uint64_t>(most, (uint64_t)(nbytes) / 4); }
        CP(out.nodes, g.nodes, o_n * sizeof(BftNode), g.n_nodes * sizeof(BftNode));
        CP(out.bfT, g.bfT, o_b * 8, g.n_bf8 * 8);
        CP(out.ccs, g.ccs, o_c * sizeof(BftCC), g.n_ccs * sizeof(BftCC));
        CP(out.f2w, g.f2w, o_f * 8, g.n_f2w * 8);
        CP(out.clus, g.clus, o_q * 8, g.n_clus * 8);
        CP(out.child, g.child, o_e * 8, g.n_child * 8);
        CP(out.uck, g.uck, o_u * W * 8, g.n_uc * W * 8);
        CP(out.ucrow, g.ucrow, o_u * 4, g.n_uc * 4);
#undef CP
        o_n += g.n_nodes; o_b += g.n_bf8; o_c += g.n_ccs; o_f += g.n_f2w; o_q += g.n_clus; o_e += g.n_child; o_u += g.n_uc;
    }
    if (jobs.n) {
        const unsigned gx = (unsigned)std::min<uint64_t>(1024, std::max<uint64_t>(1, (most + ABLK * 4 - 1) / (ABLK * 4)));
        hipLaunchKernelGGL(k_concat, dim3(gx, (unsigned)jobs.n), dim3(ABLK), 0, s, jobs);
        HIPCK(hipGetLastError());
    }
    // (no wait: the segments go back to the cache under this stream's tag, and the caller goes on in this stream)
    out.n_nodes = T_nodes; out.n_ccs = T_ccs; out.n_f2w = T_f2w; out.n_clus = T_clus; out.n_child = T_child; out.n_bf8 = T_bf8; out.n_uc = T_uc;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// flat form of the big CCs (bft_image.h): derived from ccs / f2w / clus / child
// ---------------------------------------------------------------------------------------------
__global__ void k_flat_flags(const BftCC* __restrict__ ccs, uint32_t C, uint32_t flat_min, uint32_t* __restrict__ flag, uint32_t* __restrict__ cnt) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        const uint32_t f = ccs[c].nb_elem >= flat_min ? 1u : 0u;
        flag[c] = f;
        cnt[c] = f ? ccs[c].nb_elem : 0u;
    }
}

__global__ void k_ccx(const BftCC* __restrict__ ccs, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ fidx, const uint32_t* __restrict__ foff,
                      uint32_t C, BftCCX* __restrict__ out) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        const BftCC cc = ccs[c];
        BftCCX x;
        x.f2_off = cc.f2_off; x.clus_off = cc.clus_off; x.child_off = cc.child_off; x.nb_elem = cc.nb_elem; x.s = cc.s;
        x.flat = (uint8_t)flag[c];
        x.f18_off = flag[c] ? fidx[c] * BFT_F18_WORDS : 0u;
        x.fent_off = flag[c] ? foff[c] : 0u;
        x.pad[0] = 0; x.pad[1] = 0;
        out[c] = x;
    }
}

#define FLAT_MAX_F2W 352  // filter2 words of one CC: ceil(2^14 / 48) = 342 (s = 4), 22 (s = 8)

// One workgroup per flat CC: walk the clusters in p_u order, emit every prefix entry in r order and set bit r.  A work item is a QUARTER of a
// filter2 word (12 of its 48 prefixes): 1024 threads with a dozen clusters each, where a thread per word walked up to 96 one after the other
// (0.36 ms for config 3's 21 root CCs, all of it latency).
#define FLAT_BLK 1024
#define FLAT_Q 4                                    // items per filter2 word
#define FLAT_QBITS (BFT_F2_BITS_PER_WORD / FLAT_Q)  // 12
static_assert(BFT_F2_BITS_PER_WORD % FLAT_Q == 0, "a filter2 word splits into equal parts");
__global__ __launch_bounds__(FLAT_BLK) void k_flat_fill(const BftCCX* __restrict__ ccx, uint32_t C, const uint64_t* __restrict__ f2w, const uint64_t* __restrict__ clus,
                                                        const uint64_t* __restrict__ child, uint64_t* __restrict__ f18, uint64_t* __restrict__ fent) {
    __shared__ uint32_t ipos[FLAT_MAX_F2W * FLAT_Q];
    __shared__ uint32_t wtot[FLAT_BLK / 64];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t c = blockIdx.x; c < C; c += gridDim.x) {
        const BftCCX cc = ccx[c];
        if (!cc.flat) continue;  // uniform over the workgroup
        const uint32_t nw = ((1u << (18 - cc.s)) + BFT_F2_BITS_PER_WORD - 1) / BFT_F2_BITS_PER_WORD, ni = nw * FLAT_Q;
        // entries of every item
        for (uint32_t it = threadIdx.x; it < ni; it += FLAT_BLK) {
            const uint32_t w = it / FLAT_Q, q = it % FLAT_Q;
            const uint64_t fw = f2w[cc.f2_off + w];
            const uint64_t all = fw & ((1ull << BFT_F2_BITS_PER_WORD) - 1ull);
            uint64_t bits = (all >> (q * FLAT_QBITS)) & ((1ull << FLAT_QBITS) - 1ull);
            uint32_t clu = (uint32_t)(fw >> 48) + (uint32_t)__builtin_popcountll(all & ((1ull << (q * FLAT_QBITS)) - 1ull)), n = 0;
            for (; bits; bits &= bits - 1, clu++) {
                const uint64_t e = clus[cc.clus_off + clu];
                n += (e & BFT_CLUS_MULTI) ? (uint32_t)((e >> BFT_CLUS_LEN_SHIFT) & 0xFFFFu) : 1u;
            }
            ipos[it] = n;
        }
        __syncthreads();
        // exclusive scan of ipos[0, ni): a run of `per` items per thread, wavefront scans of the runs' sums, the wavefronts' totals by the first lanes
        {
            const uint32_t per = (ni + FLAT_BLK - 1) / FLAT_BLK, i0 = threadIdx.x * per, i1 = i0 + per < ni ? i0 + per : ni;
            uint32_t sum = 0;
            for (uint32_t i = i0; i < i1; i++) sum += ipos[i];
            uint32_t inc = sum;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t v = __shfl_up(inc, d);
                if ((int)lane >= d) inc += v;
            }
            if (lane == 63) wtot[wv] = inc;
            __syncthreads();
            uint32_t base = 0;
            for (uint32_t v = 0; v < wv; v++) base += wtot[v];
            uint32_t acc = base + inc - sum;
            for (uint32_t i = i0; i < i1; i++) { const uint32_t n = ipos[i]; ipos[i] = acc; acc += n; }
        }
        __syncthreads();
        for (uint32_t it = threadIdx.x; it < ni; it += FLAT_BLK) {
            const uint32_t w = it / FLAT_Q, q = it % FLAT_Q;
            const uint64_t fw = f2w[cc.f2_off + w];
            const uint64_t all = fw & ((1ull << BFT_F2_BITS_PER_WORD) - 1ull);
            uint64_t bits = (all >> (q * FLAT_QBITS)) & ((1ull << FLAT_QBITS) - 1ull);
            uint32_t clu = (uint32_t)(fw >> 48) + (uint32_t)__builtin_popcountll(all & ((1ull << (q * FLAT_QBITS)) - 1ull)), pos = ipos[it];
            for (; bits; bits &= bits - 1, clu++) {
                const uint32_t pu = w * BFT_F2_BITS_PER_WORD + q * FLAT_QBITS + (uint32_t)__builtin_ctzll(bits);
                const uint64_t e = clus[cc.clus_off + clu];
                const uint32_t len = (e & BFT_CLUS_MULTI) ? (uint32_t)((e >> BFT_CLUS_LEN_SHIFT) & 0xFFFFu) : 1u;
                for (uint32_t j = 0; j < len; j++) {
                    const uint64_t ent = (e & BFT_CLUS_MULTI) ? child[cc.child_off + (uint32_t)e + j] : e;
                    const uint32_t r = (pu << cc.s) | ((uint32_t)(ent >> BFT_CHILD_PV_SHIFT) & 0xFFu);
                    fent[cc.fent_off + pos++] = ent;
                    atomicOr((unsigned long long*)&f18[cc.f18_off + r / BFT_F2_BITS_PER_WORD], 1ull << (r % BFT_F2_BITS_PER_WORD));
                }
            }
        }
        __syncthreads();
    }
}

// running rank into each word of the flat bitmaps (separate launch: the bits were set with atomics)
__global__ __launch_bounds__(ABLK) void k_flat_ranks(const BftCCX* __restrict__ ccx, uint32_t C, uint64_t* __restrict__ f18) {
    __shared__ uint32_t part[ABLK];
    const uint32_t per = (BFT_F18_WORDS + ABLK - 1) / ABLK;
    for (uint32_t c = blockIdx.x; c < C; c += gridDim.x) {
        const BftCCX cc = ccx[c];
        if (!cc.flat) continue;
        uint64_t* f = f18 + cc.f18_off;
        const uint32_t w0 = threadIdx.x * per, w1 = w0 + per < BFT_F18_WORDS ? w0 + per : BFT_F18_WORDS;
        uint32_t n = 0;
        for (uint32_t w = w0; w < w1; w++) n += (uint32_t)__builtin_popcountll(f[w]);
        part[threadIdx.x] = n;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t acc = 0;
            for (uint32_t i = 0; i < ABLK; i++) { const uint32_t v = part[i]; part[i] = acc; acc += v; }
        }
        __syncthreads();
        uint32_t rank = part[threadIdx.x];
        for (uint32_t w = w0; w < w1; w++) {
            const uint64_t v = f[w];
            f[w] = v | ((uint64_t)rank << 48);
            rank += (uint32_t)__builtin_popcountll(v);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// colour sets
// ---------------------------------------------------------------------------------------------
// (the signature function itself: bft_cs_sig.h)

// The genome-id lists of the index are bimodal (a k-mer of the shared ancestor sits in most genomes, a k-mer around a SNP in
// one), so one-thread-per-list loops leave most lanes idle for the length of the longest list of the wavefront.  The lists
// of 64 consecutive k-mers are ONE contiguous range of pg: the kernels below let the wavefront stream that range coalesced,
// each element finding its list by a binary search over the lanes' start offsets (6 shuffles).
// lane s owns the list pg[a_s, ...): the largest lane whose start is <= e (starts are non-decreasing over the lanes)
__device__ __forceinline__ uint32_t wave_list_of(uint32_t e, uint32_t a) {
    uint32_t lo = 0, hi = 63;
#pragma unroll
    for (int it = 0; it < 6; it++) {  // every lane takes part in every shuffle
        const uint32_t mid = (lo + hi + 1) >> 1;
        const uint32_t am = __shfl(a, mid);
        if (am <= e) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// signature of the sorted genome-id list of each k-mer: a sum of mixed ids (order-free, so that the elements can be added
// in any order by any lane) mixed with the length
__global__ __launch_bounds__(ABLK) void k_cs_sig(const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ pg, uint32_t nk, int weak, uint64_t* __restrict__ sig,
                                                 uint32_t* __restrict__ iota) {
    __shared__ unsigned long long acc[ABLK];
    const uint32_t lane = threadIdx.x & 63u, w0 = threadIdx.x & ~63u;
    const uint32_t nblk = (nk + ABLK - 1) / ABLK;
    for (uint32_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {  // whole wavefronts stay in the loop together
        const uint32_t i = blk * ABLK + threadIdx.x;
        const bool valid = i < nk;
        const uint32_t a = seg_off[valid ? i : nk], b = seg_off[valid ? i + 1 : nk];
        const uint32_t base = __shfl(a, 0), end = __shfl(b, 63);
        acc[threadIdx.x] = 0ull;
        for (uint32_t e0 = base; e0 < end; e0 += 64) {
            const uint32_t e = e0 + lane;
            const bool in = e < end;
            const uint32_t s = wave_list_of(in ? e : end - 1, a);
            if (in) atomicAdd(&acc[w0 + s], (unsigned long long)bft_cs_term(pg[e]));
        }
        if (valid) {
            // (weak: a test hook -- the signature of a list is its length, so that different lists collide and the exact pass must run)
            sig[i] = bft_cs_finish((uint64_t)acc[threadIdx.x], b - a, weak);
            iota[i] = i;
        }
    }
}

__device__ __forceinline__ bool same_list(const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ pg, uint32_t a, uint32_t b) {
    const uint32_t la = seg_off[a + 1] - seg_off[a], lb = seg_off[b + 1] - seg_off[b];
    if (la != lb) return false;
    for (uint32_t i = 0; i < la; i++)
        if (pg[seg_off[a] + i] != pg[seg_off[b] + i]) return false;
    return true;
}

// a new colour set starts where the (signature-sorted) list differs from its predecessor.  exact = 0: by the 64-bit signature and the
// length alone (two gathers of a whole list per k-mer saved: 2.0 ms of config 3's 8); k_cs_verify then compares EVERY k-mer's list
// with the dictionary entry it was given, and a mismatch -- two different lists with one signature -- sends the caller back here with
// exact = 1: the lists themselves are compared.
__global__ void k_cs_heads(const uint64_t* __restrict__ sig_s, const uint32_t* __restrict__ order, const uint32_t* __restrict__ seg_off,
                           const uint32_t* __restrict__ pg, uint32_t nk, int exact, uint32_t* __restrict__ head, uint32_t* __restrict__ len) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += gridDim.x * blockDim.x) {
        const uint32_t a = order[i];
        uint32_t h = 1;
        if (i > 0 && sig_s[i] == sig_s[i - 1]) {
            const uint32_t b = order[i - 1];
            if (exact ? same_list(seg_off, pg, a, b) : (seg_off[a + 1] - seg_off[a] == seg_off[b + 1] - seg_off[b])) h = 0;
        }
        head[i] = h;
        len[i] = h ? seg_off[a + 1] - seg_off[a] : 0;
    }
}

// tcol of every k-mer (position i of the signature order) and, for the first k-mer of every run of equal lists, the
// dictionary entry: the lanes of a wavefront copy that list together, 64 ids at a time
__global__ __launch_bounds__(ABLK) void k_cs_assign(const uint32_t* __restrict__ order, const uint32_t* __restrict__ head, const uint32_t* __restrict__ csid_ex,
                                                    const uint32_t* __restrict__ off_ex, const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ pg,
                                                    uint32_t nk, uint32_t* __restrict__ tcol, uint32_t* __restrict__ cs_off, uint32_t* __restrict__ cs_ids) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nblk = (nk + ABLK - 1) / ABLK;
    for (uint32_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const uint32_t i = blk * ABLK + threadIdx.x;
        uint32_t hd = 0, src = 0, len = 0, dst = 0;
        if (i < nk) {
            const uint32_t a = order[i];
            hd = head[i];
            const uint32_t cs = csid_ex[i] + hd - 1;
            tcol[a] = cs;
            if (hd) {
                src = seg_off[a];
                len = seg_off[a + 1] - src;
                dst = off_ex[i];
                cs_off[cs] = dst;
            }
        }
        uint64_t heads = __ballot(hd != 0);
        while (heads) {
            const int t = __builtin_ctzll(heads);
            heads &= heads - 1;
            const uint32_t s0 = __shfl(src, t), l0 = __shfl(len, t), d0 = __shfl(dst, t);
            for (uint32_t j = lane; j < l0; j += 64) cs_ids[d0 + j] = pg[s0 + j];
        }
    }
}

// exactness check: every k-mer's list equals the dictionary entry it was assigned (streamed like k_cs_sig)
__global__ __launch_bounds__(ABLK) void k_cs_verify(const uint32_t* __restrict__ tcol, const uint32_t* __restrict__ cs_off, const uint32_t* __restrict__ cs_ids,
                                                    const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ pg, uint32_t nk, uint32_t* __restrict__ bad) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nblk = (nk + ABLK - 1) / ABLK;
    for (uint32_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const uint32_t i = blk * ABLK + threadIdx.x;
        const bool valid = i < nk;
        const uint32_t a = seg_off[valid ? i : nk], b = seg_off[valid ? i + 1 : nk];
        uint32_t co = 0;
        bool wrong = false;
        if (valid) {
            const uint32_t cs = tcol[i];
            co = cs_off[cs];
            wrong = (cs_off[cs + 1] - co) != (b - a);
        }
        const uint32_t base = __shfl(a, 0), end = __shfl(b, 63);
        for (uint32_t e0 = base; e0 < end; e0 += 64) {
            const uint32_t e = e0 + lane;
            const bool in = e < end;
            const uint32_t s = wave_list_of(in ? e : end - 1, a);
            const uint32_t as = __shfl(a, s), cos = __shfl(co, s);
            if (in && cs_ids[cos + (e - as)] != pg[e]) wrong = true;  // a list of the wrong length is already flagged by its own lane
        }
        if (wrong) atomicAdd(bad, 1u);
    }
}

// ---- interning by a hash of the signatures ----
// The lists of a pan-genome repeat massively (config 3: 4.5 x 10^7 k-mers, 1.6 x 10^6 distinct lists, and nine k-mers in ten carry one
// of the 100 one-genome lists): every k-mer looks its signature up in an open-addressed table -- the popular lists are L2 hits --,
// the few distinct signatures are sorted (that order numbers the sets, whichever thread inserted first), and the first k-mer to claim
// a slot lends the dictionary its list.  Sorting every k-mer's signature instead (6 radix passes over 4.5 x 10^7 pairs, then gathers
// in that order) was 3.5 ms of config 3's 6.5.
struct CsSlot {
    unsigned long long sig;  // 0: free
    uint32_t rep;            // a k-mer with this signature
    uint32_t id;             // number of the set
};
__device__ __forceinline__ unsigned long long cs_key(uint64_t sig) { return sig ? sig : 0x9E3779B97F4A7C15ULL; }

// (Nine lookups in ten ask for one of a hundred slots: as device-scope atomics those queue up at the few memory channels that own
// them -- 13 ms; a run of ONE genome asks 2 x 10^6 times for one slot.  Each workgroup therefore remembers, in LDS, the slot where a
// signature was last found; a remembered slot is confirmed by an ordinary cached load of its signature, which is immutable once
// claimed -- a stale view can only show the slot free and sends the lookup down the atomic path.  And only ONE thread of the workgroup
// at a time takes the atomic path for an entry of that memory (an LDS lock per entry): the others with the same signature find the
// slot remembered when they look again.)
#define CS_CACHE 2048u
__global__ __launch_bounds__(ABLK) void k_cs_hash_insert(const uint64_t* __restrict__ sig, uint32_t nk, CsSlot* __restrict__ tab, uint32_t mask,
                                                         uint32_t* __restrict__ slot_of,
                                                         uint32_t* __restrict__ counters) {  // [0] slots claimed, [1] lookups that gave up (table too full)
    __shared__ uint32_t cache[CS_CACHE];
    __shared__ uint32_t lock[CS_CACHE];
    __shared__ uint32_t s_new, s_fail;  // (one global atomic per workgroup at the end: 10^6 atomics on ONE address take 10 ns each)
    for (uint32_t j = threadIdx.x; j < CS_CACHE; j += blockDim.x) { cache[j] = 0xFFFFFFFFu; lock[j] = 0; }
    if (threadIdx.x == 0) { s_new = 0; s_fail = 0; }
    __syncthreads();
    const uint32_t nblk = (nk + ABLK - 1) / ABLK;
    for (uint32_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const uint32_t i = blk * ABLK + threadIdx.x;
        bool done = i >= nk;
        const unsigned long long key = done ? 0ull : cs_key(sig[i]);
        const uint32_t ch = (uint32_t)(key >> 40) & (CS_CACHE - 1u);
        uint32_t found = 0xFFFFFFFFu;
        // (every lane of the wavefront stays in the loop until all are done: a lane that holds a lock finishes its lookup inside the
        // iteration, so the lanes waiting for it -- of this wavefront or another -- see the slot the next time round)
        for (int round = 0; round < 4096 && __any(!done); round++) {
            if (!done) {
                const uint32_t cand = cache[ch];
                if (cand != 0xFFFFFFFFu && tab[cand].sig == key) {
                    found = cand;
                    done = true;
                } else if (atomicCAS(&lock[ch], 0u, 1u) == 0u) {
                    uint32_t pos = (uint32_t)key & mask;
                    for (int probe = 0; probe < 64; probe++) {  // (beyond that the table is too full: the caller retries with a larger one)
                        unsigned long long cur = __hip_atomic_load(&tab[pos].sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (cur == 0ull) {
                            cur = atomicCAS(&tab[pos].sig, 0ull, key);
                            if (cur == 0ull) {
                                tab[pos].rep = i;
                                atomicAdd(&s_new, 1u);
                                cur = key;
                            }
                        }
                        if (cur == key) { found = pos; break; }
                        pos = (pos + 1u) & mask;
                    }
                    if (found != 0xFFFFFFFFu) cache[ch] = found;
                    __threadfence_block();
                    atomicExch(&lock[ch], 0u);
                    done = true;
                }
            }
        }
        if (i < nk) {
            if (found == 0xFFFFFFFFu) atomicAdd(&s_fail, 1u);
            slot_of[i] = found;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_new) atomicAdd(&counters[0], s_new);
        if (s_fail) atomicAdd(&counters[1], s_fail);
    }
}

// the claimed slots, in any order: every workgroup counts its share of the table, reserves that many places with one atomic, writes
__global__ __launch_bounds__(ABLK) void k_cs_hash_compact(const CsSlot* __restrict__ tab, uint32_t n_slots, uint64_t* __restrict__ keys, uint32_t* __restrict__ slots,
                                                          uint32_t* __restrict__ cnt) {
    __shared__ uint32_t s_cnt, s_base;
    const uint32_t per = (n_slots + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * per, p1 = min(n_slots, p0 + per);
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += blockDim.x) mine += tab[p].sig != 0ull;
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) { s_base = s_cnt ? atomicAdd(cnt, s_cnt) : 0u; s_cnt = 0; }
    __syncthreads();
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const unsigned long long v = tab[p].sig;
        if (v != 0ull) {
            const uint32_t j = s_base + atomicAdd(&s_cnt, 1u);
            keys[j] = v;
            slots[j] = p;
        }
    }
}

// r-th signature in order: the set's number, its representative and the length of its list
__global__ void k_cs_hash_ids(const uint32_t* __restrict__ slots_s, uint32_t n_sets, CsSlot* __restrict__ tab, const uint32_t* __restrict__ seg_off,
                              uint32_t* __restrict__ rep, uint32_t* __restrict__ len) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_sets; r += gridDim.x * blockDim.x) {
        CsSlot* sl = tab + slots_s[r];
        sl->id = r;
        const uint32_t a = sl->rep;
        rep[r] = a;
        len[r] = seg_off[a + 1] - seg_off[a];
    }
}

// the dictionary's ids in the width the resident image keeps them in (behind the verification, on the side stream)
template <class T>
__global__ void k_cs_narrow(const uint32_t* __restrict__ in, uint64_t n, T* __restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (T)in[i];
}

// dictionary entries: the lanes of a wavefront copy each representative's list together, 64 ids at a time
__global__ __launch_bounds__(ABLK) void k_cs_hash_copy(const uint32_t* __restrict__ rep, const uint32_t* __restrict__ cs_off, uint32_t n_sets, const uint32_t* __restrict__ seg_off,
                                                       const uint32_t* __restrict__ pg, uint32_t* __restrict__ cs_ids) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nblk = (n_sets + ABLK - 1) / ABLK;
    for (uint32_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const uint32_t r = blk * ABLK + threadIdx.x;
        uint32_t src = 0, len = 0, dst = 0;
        if (r < n_sets) {
            src = seg_off[rep[r]];
            dst = cs_off[r];
            len = cs_off[r + 1] - dst;
        }
        uint64_t todo = __ballot(len != 0);
        while (todo) {
            const int t = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t s0 = __shfl(src, t), l0 = __shfl(len, t), d0 = __shfl(dst, t);
            for (uint32_t j = lane; j < l0; j += 64) cs_ids[d0 + j] = pg[s0 + j];
        }
    }
}

__global__ void k_cs_hash_tcol(const uint32_t* __restrict__ slot_of, const CsSlot* __restrict__ tab, uint32_t nk, uint32_t* __restrict__ tcol) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += gridDim.x * blockDim.x) tcol[i] = tab[slot_of[i]].id;
}

}  // namespace

static bool g_weak_signature = false;
static unsigned long long g_exact_passes = 0;
void bft_test_weak_signature(bool on) { g_weak_signature = on; }
unsigned long long bft_test_exact_passes(void) { return g_exact_passes; }

int bft_assemble_gpu(const uint64_t* d_tk, uint64_t n, int k, const uint32_t* d_hashmod, hipStream_t s, BftDeviceIndex& out, const BftAssembleHook* hook) {
    if (!bft_valid_k(k)) return bft_fail(BFT_GPU_E_ARG, "k must be in [9, 126]");
    if (n >= 0x7FFFFFFFull) return bft_fail(BFT_GPU_E_LIMIT, "more than 2^31-1 k-mers");
    switch (bft_words_for_k(k)) {
    case 1: return assemble<1>(d_tk, n, k, d_hashmod, s, out, hook);
    case 2: return assemble<2>(d_tk, n, k, d_hashmod, s, out, hook);
    case 3: return assemble<3>(d_tk, n, k, d_hashmod, s, out, hook);
    default: return assemble<4>(d_tk, n, k, d_hashmod, s, out, hook);
    }
}

int bft_flatten_gpu(const BftCC* d_ccs, uint64_t n_ccs, const uint64_t* d_f2w, const uint64_t* d_clus, const uint64_t* d_child, uint32_t flat_min,
                    hipStream_t s, DevBuf& ccx, DevBuf& f18, DevBuf& fent, uint64_t& n_f18, uint64_t& n_fent) {
    n_f18 = 0;
    n_fent = 0;
    CK(ccx.alloc(n_ccs * sizeof(BftCCX)));
    if (n_ccs == 0) {
        CK(f18.alloc(8));
        CK(fent.alloc(8));
        return 0;
    }
    const uint32_t C = (uint32_t)n_ccs;
    Scan scan(s);
    DevBuf flag, cnt, fidx, foff;
    CK(flag.alloc(n_ccs * 4));
    CK(cnt.alloc(n_ccs * 4));
    CK(fidx.alloc(n_ccs * 4));
    CK(foff.alloc(n_ccs * 4));
    const dim3 grid(bft_grid_for((n_ccs + ABLK - 1) / ABLK)), block(ABLK);
    hipLaunchKernelGGL(k_flat_flags, grid, block, 0, s, d_ccs, C, flat_min, flag.as<uint32_t>(), cnt.as<uint32_t>());
    CK(scan.enqueue(flag.as<uint32_t>(), fidx.as<uint32_t>(), n_ccs, 0));
    CK(scan.enqueue(cnt.as<uint32_t>(), foff.as<uint32_t>(), n_ccs, 1));
    CK(scan.wait());  // (one wait for both counts)
    const uint64_t nflat = scan.get(0);
    n_fent = scan.get(1);
    n_f18 = nflat * BFT_F18_WORDS;
    if (n_f18 > 0xFFFFFFFFull) return bft_fail(BFT_GPU_E_LIMIT, "flat prefix bitmaps exceed 2^32 words");
    CK(f18.alloc_zero(n_f18 * 8, s));
    CK(fent.alloc(n_fent * 8));
    hipLaunchKernelGGL(k_ccx, grid, block, 0, s, d_ccs, flag.as<uint32_t>(), fidx.as<uint32_t>(), foff.as<uint32_t>(), C, ccx.as<BftCCX>());
    if (nflat) {
        const dim3 g2((unsigned)std::min<uint64_t>(n_ccs, 65535));
        hipLaunchKernelGGL(k_flat_fill, g2, dim3(FLAT_BLK), 0, s, ccx.as<BftCCX>(), C, d_f2w, d_clus, d_child, f18.as<uint64_t>(), fent.as<uint64_t>());
        hipLaunchKernelGGL(k_flat_ranks, g2, block, 0, s, ccx.as<BftCCX>(), C, f18.as<uint64_t>());
    }
    HIPCK(hipGetLastError());
    return 0;  // (no wait: the caller goes on in this stream, and what is released here is handed out again in its order)
}

// Interning of the colour sets (sorted genome-id list of each distinct k-mer, CSR seg_off/pg) into a dictionary: a 64-bit signature
// per list, equal signatures found through a hash table (k_cs_hash_*), sets numbered in the order of their signatures, then a
// verification pass that compares EVERY k-mer's list with the dictionary entry it was given.  A mismatch -- two different lists with
// one signature -- sends the build through the sorted path, where runs of equal lists are found by comparing the lists themselves.
// Two different sets can therefore never share an id, and two equal sets always do.
int bft_intern_colors_gpu(const uint32_t* d_seg_off, const uint32_t* d_pg, uint64_t nk, uint64_t np, hipStream_t s, DevBuf& d_tcol,
                          DevBuf& d_cs_off, DevBuf& d_cs_ids, uint64_t& n_sets, uint64_t& n_ids, uint64_t distinct_hint, BftInternTail* tail, bool exact) {
    n_sets = 0;
    n_ids = 0;
    CK(d_tcol.alloc(nk * 4));
    if (nk == 0) {
        CK(d_cs_off.alloc_zero(4, s));
        CK(d_cs_ids.alloc(4));
        HIPCK(hipStreamSynchronize(s));
        return 0;
    }
    Scan scan(s);
    DevBuf sig, iota, bad;
    CK(iota.alloc(nk * 4));
    CK(bad.alloc(4));
    uint32_t nbad = 0;
    const dim3 grid(bft_grid_for((nk + ABLK - 1) / ABLK)), block(ABLK);
    CK(sig.alloc(nk * 8));
    hipLaunchKernelGGL(k_cs_sig, grid, block, 0, s, d_seg_off, d_pg, (uint32_t)nk, g_weak_signature ? 1 : 0, sig.as<uint64_t>(), iota.as<uint32_t>());
    bft_stage("colour sets: signatures", (double)np * 4 + (double)nk * (4 + 8 + 4), s);

    // ---- by a hash of the signatures (kernels above).  The table starts at nk / 8 slots (a pan-genome has far fewer distinct lists than
    // k-mers) and is retried at 2 nk when more than half of it fills: then every list may be distinct. ----
    bool done = false;
    if (!exact) {
        DevBuf tab, slot_of, cnt, keys, keys_s, slots, slots_s, rep, len, tmp;
        CK(slot_of.alloc(nk * 4));
        CK(cnt.alloc(3 * 4));
        uint64_t n_slots = 1ull << 16;
        while (n_slots < nk / 8 || n_slots < 3 * distinct_hint) n_slots <<= 1;  // (distinct_hint: lists the caller knows to differ -- a merge's old sets)
        for (int attempt = 0; attempt < 2 && !done; attempt++) {
            if (attempt) {
                while (n_slots < 2 * nk) n_slots <<= 1;
            }
            CK(tab.alloc_zero(n_slots * sizeof(CsSlot), s));
            HIPCK(hipMemsetAsync(cnt.p, 0, 12, s));
            hipLaunchKernelGGL(k_cs_hash_insert, grid, block, 0, s, sig.as<uint64_t>(), (uint32_t)nk, tab.as<CsSlot>(), (uint32_t)(n_slots - 1), slot_of.as<uint32_t>(),
                               cnt.as<uint32_t>());
            CK(scan.publish(cnt.as<uint32_t>(), 2, 0));
            bft_stage("colour sets: hash of the signatures", (double)nk * (8 + 4) + (double)n_slots * 16, s);
            CK(scan.wait());
            if (scan.get(1) == 0 && scan.get(0) * 2 <= n_slots) done = true;
            else if (n_slots >= 2 * nk) break;  // (cannot happen: at most nk signatures in 2 nk slots)
        }
        if (done) {
            n_sets = scan.get(0);
            CK(keys.alloc(n_sets * 8));
            CK(keys_s.alloc(n_sets * 8));
            CK(slots.alloc(n_sets * 4));
            CK(slots_s.alloc(n_sets * 4));
            CK(rep.alloc(n_sets * 4));
            CK(len.alloc(n_sets * 4));
            const dim3 tgrid(bft_grid_for((n_slots + ABLK - 1) / ABLK)), sgrid(bft_grid_for((n_sets + ABLK - 1) / ABLK));
            hipLaunchKernelGGL(k_cs_hash_compact, tgrid, block, 0, s, tab.as<CsSlot>(), (uint32_t)n_slots, keys.as<uint64_t>(), slots.as<uint32_t>(), cnt.as<uint32_t>() + 2);
            CK((bft_rs::sort_pairs<uint64_t, uint32_t, bft_rs::SHAPE_LIGHT>(keys.as<uint64_t>(), slots.as<uint32_t>(), n_sets, keys_s.as<uint64_t>(), slots_s.as<uint32_t>(), 0, 64, s)));
            hipLaunchKernelGGL(k_cs_hash_ids, sgrid, block, 0, s, slots_s.as<uint32_t>(), (uint32_t)n_sets, tab.as<CsSlot>(), d_seg_off, rep.as<uint32_t>(), len.as<uint32_t>());
            CK(d_cs_off.alloc((n_sets + 1) * 4));
            CK(scan.enqueue(len.as<uint32_t>(), d_cs_off.as<uint32_t>(), n_sets, 0, true));
            hipLaunchKernelGGL(k_cs_hash_tcol, grid, block, 0, s, slot_of.as<uint32_t>(), tab.as<CsSlot>(), (uint32_t)nk, d_tcol.as<uint32_t>());
            bft_stage("colour sets: distinct signatures sorted, ids, set per k-mer", (double)n_slots * 16 + (double)n_sets * (12 * 2 * 8 + 24) + (double)nk * 8, s);
            CK(scan.wait());
            n_ids = scan.get(0);
            CK(d_cs_ids.alloc(n_ids * 4));
            if (tail && tail->side && tail->pin.p &&
                (tail->done || hipEventCreateWithFlags(&tail->done, hipEventDisableTiming) == hipSuccess) &&
                (tail->ready || hipEventCreateWithFlags(&tail->ready, hipEventDisableTiming) == hipSuccess)) {
                // the copy and the verification on the side stream, behind everything enqueued so far; the caller collects the verdict later
                hipStream_t t2 = tail->side;
                tail->rep.swap(rep);
                CK(tail->bad.alloc_zero(4, s));
                if (tail->narrow_w < 4) CK(tail->narrow.alloc(n_ids * tail->narrow_w));  // (before `ready`: whatever held the block last was enqueued on s)
                HIPCK(hipEventRecord(tail->ready, s));
                HIPCK(hipStreamWaitEvent(t2, tail->ready, 0));
                hipLaunchKernelGGL(k_cs_hash_copy, sgrid, block, 0, t2, tail->rep.as<uint32_t>(), d_cs_off.as<uint32_t>(), (uint32_t)n_sets, d_seg_off, d_pg, d_cs_ids.as<uint32_t>());
                hipLaunchKernelGGL(k_cs_verify, grid, block, 0, t2, d_tcol.as<uint32_t>(), d_cs_off.as<uint32_t>(), d_cs_ids.as<uint32_t>(), d_seg_off, d_pg, (uint32_t)nk,
                                   tail->bad.as<uint32_t>());
                hipLaunchKernelGGL(k_publish, dim3(1), dim3(PIN_SLOTS), 0, t2, tail->bad.as<uint32_t>(), 1, tail->pin.p);
                if (n_ids && tail->narrow_w == 1) hipLaunchKernelGGL(k_cs_narrow<uint8_t>, grid, block, 0, t2, d_cs_ids.as<uint32_t>(), n_ids, tail->narrow.as<uint8_t>());
                if (n_ids && tail->narrow_w == 2) hipLaunchKernelGGL(k_cs_narrow<uint16_t>, grid, block, 0, t2, d_cs_ids.as<uint32_t>(), n_ids, tail->narrow.as<uint16_t>());
                HIPCK(hipGetLastError());
                HIPCK(hipEventRecord(tail->done, t2));
                bft_stage("+colour sets: dictionary copied, every list verified, ids narrowed (side stream)",
                          (double)n_ids * 8 + (double)n_sets * 12 + (double)np * 8 + (double)nk * 12 + (tail->narrow_w < 4 ? (double)n_ids * (4 + tail->narrow_w) : 0.0), t2);
                tail->pending = true;
                return 0;  // (tab, slot_of ... go back to the cache under this stream's tag: what reads them was enqueued on s before this point)
            }
            hipLaunchKernelGGL(k_cs_hash_copy, sgrid, block, 0, s, rep.as<uint32_t>(), d_cs_off.as<uint32_t>(), (uint32_t)n_sets, d_seg_off, d_pg, d_cs_ids.as<uint32_t>());
            bft_stage("colour sets: dictionary copied", (double)n_ids * 8 + (double)n_sets * 12, s);
            HIPCK(hipMemsetAsync(bad.p, 0, 4, s));
            hipLaunchKernelGGL(k_cs_verify, grid, block, 0, s, d_tcol.as<uint32_t>(), d_cs_off.as<uint32_t>(), d_cs_ids.as<uint32_t>(), d_seg_off, d_pg, (uint32_t)nk,
                               bad.as<uint32_t>());
            CK(scan.publish(bad.as<uint32_t>(), 1, 1));
            bft_stage("colour sets: every list verified", (double)np * 8 + (double)nk * 12, s);
            CK(scan.wait());
            nbad = (uint32_t)scan.get(1);
            done = nbad == 0;  // (else: two different lists with one signature -- the sorted path below compares the lists)
        }
    }
    if (done) return 0;

    // ---- by a sort of every k-mer's signature: equal lists next to each other, runs found by comparison ----
    DevBuf sig_s, order, head, len, csid, off, tmp;
    CK(sig_s.alloc(nk * 8));
    CK(order.alloc(nk * 4));
    CK(head.alloc(nk * 4));
    CK(len.alloc(nk * 4));
    CK(csid.alloc(nk * 4));
    CK(off.alloc(nk * 4));
    // Equal lists only have to end up next to each other: the low 48 bits of the signature order them (6 radix passes instead
    // of 8).  Two different lists that agree on those bits could at worst split a run of equal lists, i.e. cost a duplicate
    // dictionary entry (expected once in ~10^14 / n_sets^2 builds); k_cs_heads still compares whole signatures and lists.
    CK((bft_rs::sort_pairs<uint64_t, uint32_t, bft_rs::SHAPE_LIGHT>(sig.as<uint64_t>(), iota.as<uint32_t>(), nk, sig_s.as<uint64_t>(), order.as<uint32_t>(), 0, 48, s)));
    for (int exact = 1; exact < 2; exact++) {
        g_exact_passes++;
        hipLaunchKernelGGL(k_cs_heads, grid, block, 0, s, sig_s.as<uint64_t>(), order.as<uint32_t>(), d_seg_off, d_pg, (uint32_t)nk, exact, head.as<uint32_t>(), len.as<uint32_t>());
        CK(scan.run(head.as<uint32_t>(), csid.as<uint32_t>(), nk, &n_sets));
        CK(scan.run(len.as<uint32_t>(), off.as<uint32_t>(), nk, &n_ids));
        CK(d_cs_off.alloc((n_sets + 1) * 4));
        CK(d_cs_ids.alloc(n_ids * 4));
        {
            const uint32_t t32 = (uint32_t)n_ids;
            HIPCK(hipMemcpyAsync(d_cs_off.as<uint32_t>() + n_sets, &t32, 4, hipMemcpyHostToDevice, s));
        }
        hipLaunchKernelGGL(k_cs_assign, grid, block, 0, s, order.as<uint32_t>(), head.as<uint32_t>(), csid.as<uint32_t>(), off.as<uint32_t>(), d_seg_off, d_pg,
                           (uint32_t)nk, d_tcol.as<uint32_t>(), d_cs_off.as<uint32_t>(), d_cs_ids.as<uint32_t>());
        HIPCK(hipMemsetAsync(bad.p, 0, 4, s));
        hipLaunchKernelGGL(k_cs_verify, grid, block, 0, s, d_tcol.as<uint32_t>(), d_cs_off.as<uint32_t>(), d_cs_ids.as<uint32_t>(), d_seg_off, d_pg, (uint32_t)nk,
                           bad.as<uint32_t>());
        HIPCK(hipMemcpyAsync(&nbad, bad.p, 4, hipMemcpyDeviceToHost, s));
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(s));
    }
    if (nbad) return bft_fail(BFT_GPU_E_LIMIT, "colour-set interning self-check failed");
    (void)np;
    return 0;
}
