// bft_scan.h -- the library's own device-wide scan (prefix sums of counts and flags, the running maximum that places the k-mer hash's
// rows): one pass over the input, one over the output (device code + host launcher; templates, header only).
//
// The reference needs no such primitive: it grows its containers one k-mer at a time (realloc + memmove, src/CC.c:714-1474); the bulk build
// sizes and fills arrays from counts instead, and every "where does element i go" of it is a scan.  Until round 6 these were rocPRIM /
// hipCUB calls.
//
//   * tiles of 256 x 16 elements, claimed from a counter (a tile is only ever waited for by tiles claimed after it: forward progress does
//     not depend on dispatch order); elements are read and written striped (lane l of a wavefront takes element r 64 + l of the
//     wavefront's stretch in round r: whole lines per instruction) and scanned round by round with wavefront shuffles -- no transposition
//     through LDS;
//   * a tile publishes its total as ONE 8-byte word {flag:2, value:62} (a relaxed agent-scope store: write-through, MI355X_MICROARCH.md), then
//     the 64 lanes of its first wavefront look at the 64 tiles before it at once: the values up to the nearest tile that already knows its
//     inclusive prefix are combined with shuffles, and further windows are fetched only if none of the 64 does;
//   * the operator and the input are template parameters: sums of 32- or 64-bit counts, flags computed on the fly from neighbouring keys
//     (a functor instead of a flag array), max.  Values stay below 2^62;
//   * nothing is zeroed by a launch of its own between two scans that share a scratch block.  The block holds the states twice: launch L
//     uses array L & 1 and zeroes, tile by tile, what launch L - 1 left in the other one (nobody reads that any more); the workgroup whose claim
//     is the launch's last puts the tile counter back to zero.  The host zeroes a block once, when it is new (DevBuf::tag counts the launches,
//     DevBuf::tag2 remembers how many states the last one used).  A CAPTURED launch cannot take part (it runs again with the arguments it was
//     captured with): it is bracketed by zeroing kernels (not memset nodes: bft_zero_async) and leaves the block zeroed;
//   * the grand total, which nearly every caller wants on the host or behind the last offset, is written by the last tile (total_slot, tail)
//     instead of by a launch of its own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bft_dev.h"

namespace bft_scan {

struct Sum {
    template <class T>
    __device__ __forceinline__ T operator()(T a, T b) const { return a + b; }
    static __device__ __host__ __forceinline__ uint64_t identity() { return 0; }
};
struct Max {
    template <class T>
    __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; }
    static __device__ __host__ __forceinline__ uint64_t identity() { return 0; }
};
template <class T>
struct PtrIn {
    const T* p;
    __device__ __forceinline__ T operator()(uint64_t i) const { return p[i]; }
};

constexpr int THREADS = 256, IPT = 16, TILE = THREADS * IPT, WAVES = THREADS / 64;
constexpr uint64_t F_AGG = 1ull << 62, F_INC = 2ull << 62, VMASK = (1ull << 62) - 1ull;

template <class T>
__device__ __forceinline__ T shfl_up_t(T v, int o) {
    if constexpr (sizeof(T) == 8) {
        const uint32_t lo = __shfl_up((uint32_t)v, o), hi = __shfl_up((uint32_t)((uint64_t)v >> 32), o);
        return (T)(((uint64_t)hi << 32) | lo);
    } else
        return (T)__shfl_up((uint32_t)v, o);
}
template <class T>
__device__ __forceinline__ T shfl_t(T v, int l) {
    if constexpr (sizeof(T) == 8) {
        const uint32_t lo = __shfl((uint32_t)v, l), hi = __shfl((uint32_t)((uint64_t)v >> 32), l);
        return (T)(((uint64_t)hi << 32) | lo);
    } else
        return (T)__shfl((uint32_t)v, l);
}

// scratch: [0] tile counter (zero between launches); states: this launch's tile states {flag:2, value:62}, zero on entry; zero_other[0 .. zero_n):
// the states of the launch before, zeroed here
// SINGLE: the whole input is one tile (no counter, no states)
// total_slot (may be NULL; device or pinned host memory): receives init (+) in(0) (+) ... (+) in(n - 1); tail: out[n] receives it too
template <class T, class In, class Op, bool INCLUSIVE, bool SINGLE>
__global__ __launch_bounds__(THREADS) void k_scan(In in, T* __restrict__ out, uint64_t n, T init, Op op, unsigned long long* __restrict__ scratch,
                                                  unsigned long long* __restrict__ states, unsigned long long* __restrict__ zero_other, uint32_t zero_n,
                                                  unsigned long long* __restrict__ total_slot, int tail) {
    __shared__ T s_wave[WAVES];
    __shared__ T s_prefix;
    __shared__ uint32_t s_tile;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t ntiles = (uint32_t)((n + TILE - 1) / TILE);
    for (;;) {
        if (!SINGLE) {
            if (tid == 0) s_tile = (uint32_t)atomicAdd(scratch, 1ull);
            __syncthreads();
        }
        const uint32_t tile = SINGLE ? 0u : s_tile;
        if (tile >= ntiles) {
            // every workgroup makes exactly one claim that fails: the last of them leaves the counter at zero for the next launch
            if (tid == 0 && tile == ntiles + gridDim.x - 1u) __hip_atomic_store(scratch, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const uint64_t base = (uint64_t)tile * TILE + (uint64_t)wave * 64u * IPT;
        T v[IPT];
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint64_t i = base + (uint64_t)r * 64u + lane;
            v[r] = i < n ? in(i) : (T)Op::identity();
        }
        // inclusive scan of the wavefront's stretch: round by round, the rounds chained through lane 63
        T carry = (T)Op::identity();
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            T x = v[r];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const T y = shfl_up_t(x, o);
                if ((int)lane >= o) x = op(y, x);
            }
            x = op(carry, x);
            v[r] = x;
            carry = shfl_t(x, 63);
        }
        if (lane == 63) s_wave[wave] = carry;
        __syncthreads();
        T wpre = (T)Op::identity(), total = (T)Op::identity();
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            if (w < (int)wave) wpre = op(wpre, s_wave[w]);
            total = op(total, s_wave[w]);
        }
        // publish, look back (first wavefront), broadcast the tile's exclusive prefix
        if (wave == 0) {
            T ex = init;
            if (SINGLE) {
            } else if (tile == 0) {
                if (lane == 0) __hip_atomic_store(&states[0], F_INC | ((uint64_t)op(init, total) & VMASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (lane == 0) __hip_atomic_store(&states[tile], F_AGG | ((uint64_t)total & VMASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                T acc = (T)Op::identity();
                int64_t first = (int64_t)tile - 1;  // the window is the tiles first, first - 1, ..., first - 63
                for (;;) {
                    const int64_t t = first - (int64_t)lane;
                    unsigned long long st = F_INC;  // (before tile 0: nothing, and the search ends there)
                    if (t >= 0) {
                        do { st = __hip_atomic_load(&states[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (__builtin_expect((st >> 62) == 0ull, 0) && (__builtin_amdgcn_s_sleep(1), true));
                    }
                    const uint64_t incl = __ballot((st >> 62) == 2ull);
                    const int stop = incl ? __builtin_ctzll(incl) : 64;  // the nearest tile that knows its inclusive prefix
                    // combine the values of the lanes 0 .. stop (in tile order: lane `stop` first): a reduction over lanes with the others neutral
                    T x = ((int)lane <= stop && t >= 0) ? (T)(st & VMASK) : (T)Op::identity();
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const T y = shfl_t(x, (int)((lane + o) & 63u));
                        x = op(x, y);  // (sum and max commute: the order of the lanes does not matter)
                    }
                    acc = op(acc, shfl_t(x, 0));
                    if (stop < 64) break;
                    first -= 64;
                }
                ex = acc;  // (the inclusive state the search ended at already holds `init`: tile 0 starts from it)
                if (lane == 0) __hip_atomic_store(&states[tile], F_INC | ((uint64_t)op(ex, total) & VMASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) s_prefix = ex;
            if (lane == 0 && tile == ntiles - 1u) {  // the grand total, for the host and / or behind the last element
                const T grand = op(ex, total);
                if (total_slot) *total_slot = (unsigned long long)grand;
                if (tail) out[n] = grand;
            }
        }
        __syncthreads();
        const T pre = op(s_prefix, wpre);
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint64_t i = base + (uint64_t)r * 64u + lane;
            if (INCLUSIVE) {
                if (i < n) out[i] = op(pre, v[r]);
            } else {
                // exclusive: the element before me in scan order = lane - 1 of this round, or lane 63 of the round before
                T p = shfl_up_t(v[r], 1);
                const T last_prev = r ? shfl_t(v[r - 1], 63) : (T)Op::identity();
                if (lane == 0) p = last_prev;
                if (i < n) out[i] = op(pre, p);
            }
        }
        if (SINGLE) return;
        {   // this tile's share of the states the launch before left behind
            const uint32_t per = (zero_n + ntiles - 1u) / ntiles;
            for (uint32_t i = tid; i < per; i += THREADS) {
                const uint64_t j = (uint64_t)tile * per + i;
                if (j < zero_n) zero_other[j] = 0ull;
            }
        }
        __syncthreads();  // (s_tile, s_wave, s_prefix are rewritten by the next tile)
    }
}
inline size_t scratch_bytes(uint64_t n) { return ((((n + TILE - 1) / TILE) * 2 + 2) * 8 + 63) & ~(size_t)63; }  // what a scan of n elements needs (callers that must not allocate later)
// out[i] = init (+) in(0) (+) ... (+) in(i - 1) (exclusive) or ... (+) in(i) (inclusive); `scratch` is (re)allocated as needed and may be shared by
// any number of scans that follow one another on their stream(s)
template <class T, class In, class Op, bool INCLUSIVE>
int scan(In in, T* out, uint64_t n, T init, Op op, hipStream_t s, DevBuf& scratch, unsigned long long* total_slot = nullptr, bool tail = false) {
    if (n == 0) return 0;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
    if (ntiles == 1) {
        hipLaunchKernelGGL((k_scan<T, In, Op, INCLUSIVE, true>), dim3(1), dim3(THREADS), 0, s, in, out, n, init, op, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                           (unsigned long long*)nullptr, 0u, total_slot, tail ? 1 : 0);
        HIPCK(hipGetLastError());
        return 0;
    }
    if (scratch.bytes < scratch_bytes(n)) CK(scratch.alloc(scratch_bytes(n)));
    unsigned long long* const base = scratch.as<unsigned long long>();
    const uint64_t half = (scratch.bytes / 8 - 2) / 2;  // states per array
    const uint32_t grid = (uint32_t)std::min<uint64_t>(ntiles, 256ull * 8ull);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { cap = hipStreamCaptureStatusNone; (void)hipGetLastError(); }
    if (cap != hipStreamCaptureStatusNone) {
        CK(bft_zero_async(scratch.p, scratch.bytes & ~(size_t)3, s));  // (kernels, not memset nodes: those replay correctly only once on this runtime, bft_dev.h)
        hipLaunchKernelGGL((k_scan<T, In, Op, INCLUSIVE, false>), dim3(grid), dim3(THREADS), 0, s, in, out, n, init, op, base, base + 2, base + 2 + half, 0u, total_slot, tail ? 1 : 0);
        HIPCK(hipGetLastError());
        CK(bft_zero_async(scratch.p, scratch.bytes & ~(size_t)3, s));
        scratch.tag = 0;  // (whenever the graph runs it leaves zeros; what eager launches left before it is gone by then: start over)
        return 0;
    }
    if (scratch.tag == 0) {  // a new block: anything may be in it
        HIPCK(hipMemsetAsync(scratch.p, 0, scratch.bytes, s));
        scratch.tag = 2;
        scratch.tag2 = 0;
    }
    const uint64_t cur = scratch.tag & 1u;
    hipLaunchKernelGGL((k_scan<T, In, Op, INCLUSIVE, false>), dim3(grid), dim3(THREADS), 0, s, in, out, n, init, op, base, base + 2 + cur * half, base + 2 + (cur ^ 1u) * half,
                       scratch.tag2, total_slot, tail ? 1 : 0);
    HIPCK(hipGetLastError());
    scratch.tag2 = (uint32_t)ntiles;
    if (++scratch.tag == 0) scratch.tag = 2;
    return 0;
}
template <class T, class In>
int exclusive_sum(In in, T* out, uint64_t n, hipStream_t s, DevBuf& scratch, unsigned long long* total_slot = nullptr, bool tail = false) {
    return scan<T, In, Sum, false>(in, out, n, (T)0, Sum(), s, scratch, total_slot, tail);
}
template <class T>
int exclusive_sum_ptr(const T* in, T* out, uint64_t n, hipStream_t s, DevBuf& scratch, unsigned long long* total_slot = nullptr, bool tail = false) {
    return scan<T, PtrIn<T>, Sum, false>(PtrIn<T>{in}, out, n, (T)0, Sum(), s, scratch, total_slot, tail);
}

}  // namespace bft_scan
