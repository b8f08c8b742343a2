// bft_front.hip -- the bulk build's front end behind the root-prefix split: (k-mer, genome) composites c = T << gb | genome, already
// grouped by the top bits of T (the rotated root prefix: 2^18 buckets of ~10^3 composites on a pan-genome index, in insertion order
// inside a bucket), become the sorted distinct k-mer table, the genome ids of every k-mer and their offsets.
//
// The reference reaches the same state one k-mer at a time: insertKmer_Node descends by the root prefix first (src/insertNode.c:38-226),
// keeps every container sorted (insertSP_CC src/CC.c:714-1474, insertKmer_UC src/UC.c:13-79) and appends the genome id to the k-mer's
// annotation when the k-mer is already there (modify_annotations, src/retrieveAnnotation.c:232-314).
//
//   k_bucket_sort   one workgroup per bucket: the bucket's composites are sorted on the remaining T bits by a stable LSD radix sort
//                   that never leaves the CU -- keys in registers, 8-bit digits, ranks from wavefront ballots (the lanes that hold
//                   the same digit find each other with eight __ballot's; one of them bumps the wavefront's own LDS counter for all),
//                   a 256-digit scan, one exchange through LDS per pass --, then duplicates are flagged against the left neighbour
//                   and the bucket goes back in place together with its counts (distinct k-mers, distinct pairs).
//   k_bucket_emit   after one scan of the 2^18 count pairs: every bucket writes its k-mers, their offsets and the genome ids at
//                   its place in the outputs.
// One read and one write of the array for all the remaining bits, where a device-wide LSD sort spends a pass per 8 bits
// (rocPRIM: 7 passes over 2x10^8 composites, 7.6 ms; its segmented sort of the same buckets: 4.5 ms).
#include <hipcub/hipcub.hpp>

#include "bft_dev.h"

#define FB_BLOCK 256
#define FB_WAVES (FB_BLOCK / 64)
#define FB_EMAX 16                      // composites per thread
#define FB_CAP (FB_BLOCK * FB_EMAX)     // largest bucket sorted in LDS (4096 composites = 32 KB)
#define FB_DBITS 8                      // digit width (9-bit digits, four passes instead of five over 36 bits, were slower: 5.9 ms against 4.7 on
                                        // config 3 -- 40 KB of LDS per workgroup leaves three per CU, and the counters' upkeep grows with the digits)
#define FB_DIGITS (1 << FB_DBITS)
// (Also tried: the top 24 remaining bits first -- three passes instead of five --, every pass only for a bucket whose k-mers then fail
// an order check.  On a pan-genome 85 % of the buckets fail it: the SNP variants of a k-mer share all but one base, one variant in
// three differs from its neighbour only below those bits.  Dropped.)

namespace {

// lanes of the wavefront whose (valid) digit equals mine
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid, int nbits) {
    uint64_t m = __ballot(valid);
    for (int j = 0; j < nbits; j++) {
        const bool bit = (d >> j) & 1u;
        const uint64_t bj = __ballot(bit);
        m &= bit ? bj : ~bj;
    }
    return m;
}

// vals != nullptr: c holds whole T-form k-mers grouped by their top bits and vals the genome id of each (vw bytes wide); the composite
// (T's bits below the split) << lo_bit | genome is formed here -- the top bits are the bucket's number, so a k-mer of up to 64 - lo_bit
// bits below the split fits whatever the number of genomes (k = 31: 44 bits and up to 2^20 genomes) -- and goes back into c.
__device__ __forceinline__ uint64_t load_id(const void* vals, uint32_t vw, uint64_t i) {
    if (vw == 1) return reinterpret_cast<const uint8_t*>(vals)[i];
    if (vw == 2) return reinterpret_cast<const uint16_t*>(vals)[i];
    return reinterpret_cast<const uint32_t*>(vals)[i];
}

__global__ __launch_bounds__(FB_BLOCK) void k_bucket_sort(uint64_t* __restrict__ c, const uint32_t* __restrict__ boff, uint32_t nb, uint32_t lo_bit, uint32_t hi_bit,
                                                          uint64_t* __restrict__ counts, const void* __restrict__ vals, uint32_t vw) {
    __shared__ uint64_t keys[FB_CAP];
    __shared__ uint32_t cnt[FB_WAVES][FB_DIGITS];
    __shared__ uint32_t wtot[FB_WAVES];
    __shared__ uint32_t s_nk, s_np;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t a0 = boff[b], n = boff[b + 1] - a0;
        if (n == 0 || n > FB_CAP) {  // (buckets beyond the capacity are excluded by the caller: the whole build then takes the one-sort path)
            if (tid == 0) counts[b] = 0;
            continue;
        }
        const uint32_t E = (n + FB_BLOCK - 1) / FB_BLOCK;       // rounds; wavefront w owns [w E 64, (w + 1) E 64): order = (wave, round, lane)
        const uint32_t wbase = wave * E * 64u;
        uint64_t key[FB_EMAX];
#pragma unroll
        for (uint32_t r = 0; r < FB_EMAX; r++) {
            const uint32_t idx = wbase + r * 64u + lane;
            key[r] = ~0ull;
            if (r < E && idx < n) {
                key[r] = c[a0 + idx];
                if (vals) key[r] = ((key[r] & ((1ull << (hi_bit - lo_bit)) - 1ull)) << lo_bit) | load_id(vals, vw, (uint64_t)a0 + idx);
            }
        }
        for (uint32_t bit = lo_bit; bit < hi_bit; bit += FB_DBITS) {
            const int nbits = (int)min((uint32_t)FB_DBITS, hi_bit - bit);
            const uint32_t mask = (1u << nbits) - 1u;
#pragma unroll
            for (uint32_t j = 0; j < FB_DIGITS / 64; j++) cnt[wave][lane * (FB_DIGITS / 64) + j] = 0;  // the wavefront's own counters (LDS operations of one wavefront are in order)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            uint32_t rank[FB_EMAX];
#pragma unroll
            for (uint32_t r = 0; r < FB_EMAX; r++) {
                rank[r] = 0;
                if (r >= E) continue;  // (uniform)
                const uint32_t idx = wbase + r * 64u + lane;
                const bool valid = idx < n;
                const uint32_t d = (uint32_t)(key[r] >> bit) & mask;
                const uint64_t peers = match_digit(d, valid, nbits);
                if (valid) {
                    const int leader = __builtin_ctzll(peers);
                    uint32_t base = 0;
                    if ((int)lane == leader) {
                        base = cnt[wave][d];
                        cnt[wave][d] = base + (uint32_t)__builtin_popcountll(peers);
                    }
                    base = __shfl(base, leader);
                    rank[r] = base + (uint32_t)__builtin_popcountll(peers & lt_mask);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next round's leaders read what this round's leaders wrote
            }
            __syncthreads();
            {   // digit(s) of this thread: counts of the four wavefronts -> start of (digit, wavefront) in the bucket
                constexpr int DPT = FB_DIGITS / FB_BLOCK;  // digits per thread
                uint32_t cw[DPT][FB_WAVES], tot = 0;
#pragma unroll
                for (int j = 0; j < DPT; j++) {
#pragma unroll
                    for (int w = 0; w < FB_WAVES; w++) { cw[j][w] = cnt[w][tid * DPT + j]; tot += cw[j][w]; }
                }
                uint32_t inc = tot;  // inclusive scan over the threads' totals: shuffles inside a wavefront, four partial sums across
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t v = __shfl_up(inc, o);
                    if ((int)lane >= o) inc += v;
                }
                if (lane == 63) wtot[wave] = inc;
                __syncthreads();
                uint32_t before = 0;
#pragma unroll
                for (int w = 0; w < FB_WAVES; w++)
                    if (w < (int)wave) before += wtot[w];
                uint32_t start = before + inc - tot;
#pragma unroll
                for (int j = 0; j < DPT; j++) {
#pragma unroll
                    for (int w = 0; w < FB_WAVES; w++) { cnt[w][tid * DPT + j] = start; start += cw[j][w]; }
                }
            }
            __syncthreads();
#pragma unroll
            for (uint32_t r = 0; r < FB_EMAX; r++) {
                if (r >= E) continue;
                const uint32_t idx = wbase + r * 64u + lane;
                if (idx < n) keys[cnt[wave][(uint32_t)(key[r] >> bit) & mask] + rank[r]] = key[r];
            }
            __syncthreads();
#pragma unroll
            for (uint32_t r = 0; r < FB_EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                if (r < E && idx < n) key[r] = keys[idx];
            }
            // (no barrier here: the next pass writes `keys` only after two more barriers, and zeroes only its own counters)
        }
        // duplicates against the left neighbour (the first composite of a bucket starts a k-mer: buckets differ in their top bits)
        if (tid == 0) { s_nk = 0; s_np = 0; }
        if (lo_bit >= hi_bit) {  // nothing was sorted (the split covered every T bit): the keys are only in registers yet
#pragma unroll
            for (uint32_t r = 0; r < FB_EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                if (r < E && idx < n) keys[idx] = key[r];
            }
        }
        __syncthreads();
        uint32_t nk = 0, np = 0;
#pragma unroll
        for (uint32_t r = 0; r < FB_EMAX; r++) {
            const uint32_t idx = wbase + r * 64u + lane;
            if (r < E && idx < n) {
                const uint64_t prev = idx ? keys[idx - 1] : ~key[r];
                nk += (key[r] >> lo_bit) != (prev >> lo_bit);
                np += key[r] != prev;
                c[a0 + idx] = key[r];
            }
        }
        for (int o = 32; o > 0; o >>= 1) { nk += __shfl_down(nk, o); np += __shfl_down(np, o); }
        if (lane == 0) { atomicAdd(&s_nk, nk); atomicAdd(&s_np, np); }
        __syncthreads();
        if (tid == 0) counts[b] = ((uint64_t)s_nk << 32) | s_np;
        __syncthreads();
    }
}

// bases[b] = exclusive scan of counts (k-mers << 32 | pairs).  Every bucket places its k-mers (T = c >> gb), the offset of each k-mer's
// first genome id, and the genome ids (c & gmask) of its distinct pairs.
__global__ __launch_bounds__(FB_BLOCK) void k_bucket_emit(const uint64_t* __restrict__ c, const uint32_t* __restrict__ boff, uint32_t nb, uint32_t gb,
                                                          const uint64_t* __restrict__ bases, uint64_t* __restrict__ tk, uint32_t* __restrict__ seg_off,
                                                          uint32_t* __restrict__ pg, int kv, uint32_t rest) {  // kv: the composites lack the bucket's
                                                                                                             // bits (k_bucket_sort, vals); rest: T bits below the split
    __shared__ uint32_t w_nk[FB_WAVES], w_np[FB_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t gmask = (1ull << gb) - 1ull, lt_mask = (1ull << lane) - 1ull;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t a0 = boff[b], n = boff[b + 1] - a0;
        if (n == 0) continue;
        const uint64_t base = bases[b];
        uint32_t kbase = (uint32_t)(base >> 32), pbase = (uint32_t)base;
        for (uint32_t r0 = 0; r0 < n; r0 += FB_BLOCK) {  // 256 composites at a time, in order
            const uint32_t idx = r0 + tid;
            const bool valid = idx < n;
            const uint64_t a = valid ? c[a0 + idx] : 0ull;
            const uint64_t prev = (valid && idx) ? c[a0 + idx - 1] : ~a;
            const bool head = valid && (a >> gb) != (prev >> gb), keep = valid && a != prev;
            const uint64_t hm = __ballot(head), km = __ballot(keep);
            if (lane == 0) { w_nk[wave] = (uint32_t)__builtin_popcountll(hm); w_np[wave] = (uint32_t)__builtin_popcountll(km); }
            __syncthreads();
            uint32_t kb = kbase, pb = pbase, tk_all = 0, tp_all = 0;
#pragma unroll
            for (int w = 0; w < FB_WAVES; w++) {
                if (w < (int)wave) { kb += w_nk[w]; pb += w_np[w]; }
                tk_all += w_nk[w];
                tp_all += w_np[w];
            }
            const uint32_t prank = pb + (uint32_t)__builtin_popcountll(km & lt_mask);
            if (keep) pg[prank] = (uint32_t)(a & gmask);
            if (head) {
                const uint32_t q = kb + (uint32_t)__builtin_popcountll(hm & lt_mask);
                tk[q] = kv ? (((uint64_t)b << rest) | (a >> gb)) : (a >> gb);
                seg_off[q] = prank;  // (a head is always kept: its pair is the k-mer's first)
            }
            kbase += tk_all;
            pbase += tp_all;
            __syncthreads();
        }
    }
}

}  // namespace

uint32_t bft_front_bucket_capacity(void) { return FB_CAP; }

int bft_front_buckets(uint64_t* d_c, uint64_t n, const uint32_t* d_boff, uint32_t nb, uint32_t gb, uint32_t split_bit, hipStream_t s, DevBuf& tk, DevBuf& seg_off,
                      DevBuf& pg, uint64_t& nk, uint64_t& np, const void* d_vals, uint32_t vw) {
    DevBuf counts, bases, tmp;
    CK(counts.alloc(((uint64_t)nb + 1) * 8));
    CK(bases.alloc(((uint64_t)nb + 1) * 8));
    HIPCK(hipMemsetAsync((uint8_t*)counts.p + (uint64_t)nb * 8, 0, 8, s));
    const dim3 grid(std::min<uint32_t>(nb, 256u * 16u)), block(FB_BLOCK);
    hipLaunchKernelGGL(k_bucket_sort, grid, block, 0, s, d_c, d_boff, nb, gb, split_bit, counts.as<uint64_t>(), d_vals, vw);
    size_t tb = 0;
    HIPCK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, counts.as<uint64_t>(), bases.as<uint64_t>(), (int)(nb + 1), s));
    CK(tmp.alloc(tb));
    HIPCK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, counts.as<uint64_t>(), bases.as<uint64_t>(), (int)(nb + 1), s));
    uint64_t total = 0;
    HIPCK(hipMemcpyAsync(&total, bases.as<uint64_t>() + nb, 8, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    nk = total >> 32;
    np = total & 0xFFFFFFFFull;
    (void)n;
    bft_trace_mark("bucket sort done (sync)");
    CK(tk.alloc(nk * 8));
    CK(seg_off.alloc((nk + 1) * 4));
    CK(pg.alloc(np * 4));
    hipLaunchKernelGGL(k_bucket_emit, grid, block, 0, s, d_c, d_boff, nb, gb, bases.as<uint64_t>(), tk.as<uint64_t>(), seg_off.as<uint32_t>(), pg.as<uint32_t>(), d_vals ? 1 : 0, split_bit - gb);
    const uint32_t np32 = (uint32_t)np;
    HIPCK(hipMemcpyAsync(seg_off.as<uint32_t>() + nk, &np32, 4, hipMemcpyHostToDevice, s));
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(s));
    return 0;
}
