// bft_front.hip -- the bulk build's front end behind the root-prefix split: (k-mer, genome) composites c = T << gb | genome, already
// grouped by the top bits of T (the rotated root prefix: 2^18 buckets of ~10^3 composites on a pan-genome index, in insertion order
// inside a bucket), become the sorted distinct k-mer table, the genome ids of every k-mer and their offsets.
//
// The reference reaches the same state one k-mer at a time: insertKmer_Node descends by the root prefix first (src/insertNode.c:38-226),
// keeps every container sorted (insertSP_CC src/CC.c:714-1474, insertKmer_UC src/UC.c:13-79) and appends the genome id to the k-mer's
// annotation when the k-mer is already there (modify_annotations, src/retrieveAnnotation.c:232-314).
//
//   k_bucket_sort_wave   one WAVEFRONT per bucket of up to 1024 composites (a pan-genome bucket holds ~760): a stable LSD radix sort on
//                   the remaining T bits that never leaves the CU and never meets a barrier -- keys in registers, 9-bit digits, the
//                   rank of a key = what one LDS atomic on the wavefront's digit counter returns, a 512-digit scan by shuffles, one
//                   exchange through LDS per pass --, then duplicates are flagged against the left neighbour and the bucket goes back
//                   in place together with its counts (distinct k-mers, distinct pairs).  The atomic ranks are stable only if the LDS
//                   serves the lanes of an instruction in lane order -- it does, undocumented --: the final order is checked over
//                   every bit, and a bucket that fails is sorted again with ranks from wavefront ballots (stable by construction).
//   k_bucket_sort   the same with one workgroup per bucket, for the larger ones (up to 4096 composites; variants by keys per thread)
//   k_bucket_emit   after one scan of the 2^18 count pairs: every bucket writes its k-mers, their offsets and the genome ids at
//                   its place in the outputs.
// Composites that do not fit 63 bits (k = 31; k = 27 beyond 512 genomes) arrive as whole k-mers with the ids beside them: the bits a
// bucket's k-mers share are dropped inside the bucket, which makes room for the id.
// One read and one write of the array for all the remaining bits, where a device-wide LSD sort spends a pass per 8 bits
// (rocPRIM: 7 passes over 2x10^8 composites, 7.6 ms; its segmented sort of the same buckets: 4.5 ms; these kernels: 3.0 ms).
#include "bft_dev.h"
#include "bft_scan.h"

#define FB_BLOCK 256
#define FB_WAVES (FB_BLOCK / 64)
#define FB_EMAX 16                      // composites per thread
#define FB_CAP (FB_BLOCK * FB_EMAX)     // largest bucket sorted in LDS (4096 composites = 32 KB)
#define FB_DBITS 9                      // digit width: four passes over the 36 bits a k = 27 bucket sorts on, five over k = 31's 44 (8-bit digits: five
                                        // and six; 3.7 -> 3.0 ms and 4.3 -> 3.8 ms on config 3 -- the ranks come from LDS atomics, whose cost does
                                        // not grow with the digit as the ballots' does; wider digits cost LDS, i.e. workgroups per CU)
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

// LSD radix passes over the bits [lo_bit, hi_bit) of the bucket's composites: keys in registers (wavefront w owns the slots
// [w E 64, (w + 1) E 64) of the bucket: order = (wave, round, lane)), per-wavefront digit counters, one exchange through `keys` per
// pass; on return key[] and keys[] hold the same arrangement.
//   BALLOT   the lanes that hold the same digit find each other with eight __ballot's, one of them bumps the counter for all: a
//            lane's rank among its peers is its position among them -- stable by construction.
//   !BALLOT  every lane bumps the counter itself with one LDS atomic that returns its rank.  The LDS unit serves the lanes of one
//            instruction that hit one address in lane order on this hardware, but nothing documents that: the caller checks the
//            final order and repeats the bucket with BALLOT when it is off (never seen; counted in *n_redone).
template <bool BALLOT, int EMAX>
__device__ __forceinline__ void radix_passes(uint64_t (&key)[EMAX], uint64_t* keys, uint32_t (*cnt)[FB_DIGITS], uint32_t* wtot, uint32_t n, uint32_t E, uint32_t lo_bit,
                                             uint32_t hi_bit) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t wbase = wave * E * 64u;
    for (uint32_t bit = lo_bit; bit < hi_bit; bit += FB_DBITS) {
        const int nbits = (int)min((uint32_t)FB_DBITS, hi_bit - bit);
        const uint32_t mask = (1u << nbits) - 1u;
#pragma unroll
        for (uint32_t j = 0; j < FB_DIGITS / 64; j++) cnt[wave][lane * (FB_DIGITS / 64) + j] = 0;  // the wavefront's own counters (LDS operations of one wavefront are in order)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t rank[EMAX];
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
            rank[r] = 0;
            if (r >= E) continue;  // (uniform)
            const uint32_t idx = wbase + r * 64u + lane;
            const bool valid = idx < n;
            const uint32_t d = (uint32_t)(key[r] >> bit) & mask;
            if (BALLOT) {
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
            } else if (valid) {
                rank[r] = atomicAdd(&cnt[wave][d], 1u);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next round reads what this round wrote
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
        for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
            if (r >= E) continue;
            const uint32_t idx = wbase + r * 64u + lane;
            if (idx < n) keys[cnt[wave][(uint32_t)(key[r] >> bit) & mask] + rank[r]] = key[r];
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
            const uint32_t idx = wbase + r * 64u + lane;
            if (r < E && idx < n) key[r] = keys[idx];
        }
        // (no barrier here: the next pass writes `keys` only after two more barriers, and zeroes only its own counters)
    }
}

template <int EMAX>
__device__ __forceinline__ void load_bucket(uint64_t (&key)[EMAX], const uint64_t* __restrict__ c, const void* __restrict__ vals, uint32_t vw, uint32_t a0, uint32_t n,
                                            uint32_t E, uint32_t lo_bit, uint32_t hi_bit) {
    const uint32_t lane = threadIdx.x & 63u, wbase = (threadIdx.x >> 6) * E * 64u;
#pragma unroll
    for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
        const uint32_t idx = wbase + r * 64u + lane;
        key[r] = ~0ull;
        if (r < E && idx < n) {
            key[r] = c[a0 + idx];
            if (vals) key[r] = ((key[r] & ((1ull << (hi_bit - lo_bit)) - 1ull)) << lo_bit) | load_id(vals, vw, (uint64_t)a0 + idx);
        }
    }
}

// mode: 0 = ranks from LDS atomics, checked, BALLOT on failure; 1 = BALLOT only; 2 = test hook: the check always fails
// EMAX: composites per thread, i.e. buckets of up to 256 EMAX composites (the host picks the smallest that holds the largest bucket:
// the kernel is bound by the latency of its LDS round trips and barriers, and both the registers and the LDS of a workgroup -- hence
// the workgroups a CU holds -- go with EMAX: 4 per CU at 16, 7 at 9)
template <int EMAX>
__global__ __launch_bounds__(FB_BLOCK, (EMAX <= 6 ? 6 : EMAX <= 9 ? 5 : EMAX <= 12 ? 4 : 3)) void k_bucket_sort(uint64_t* __restrict__ c, const uint32_t* __restrict__ boff, uint32_t nb, uint32_t lo_bit, uint32_t hi_bit,
                                                          uint64_t* __restrict__ counts, const void* __restrict__ vals, uint32_t vw, int mode, uint32_t* __restrict__ n_redone,
                                                          uint32_t min_n) {  // buckets of up to min_n composites are k_bucket_sort_wave's
    __shared__ uint64_t keys[FB_BLOCK * EMAX];
    __shared__ uint32_t cnt[FB_WAVES][FB_DIGITS];
    __shared__ uint32_t wtot[FB_WAVES];
    __shared__ uint32_t s_nk, s_np;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t a0 = boff[b], n = boff[b + 1] - a0;
        if (n <= min_n && min_n) continue;
        if (n == 0 || n > (uint32_t)(FB_BLOCK * EMAX)) {  // (buckets beyond the capacity are excluded by the caller: the whole build then takes the one-sort path)
            if (tid == 0) counts[b] = 0;
            continue;
        }
        const uint32_t E = (n + FB_BLOCK - 1) / FB_BLOCK;       // rounds
        const uint32_t wbase = wave * E * 64u;
        uint64_t key[EMAX];
        load_bucket<EMAX>(key, c, vals, vw, a0, n, E, lo_bit, hi_bit);
        bool sorted = false;
        if (mode != 1 && lo_bit < hi_bit) {
            radix_passes<false, EMAX>(key, keys, cnt, wtot, n, E, lo_bit, hi_bit);
            // In order over EVERY bit?  (The ids below lo_bit are not sorted on: equal k-mers must have kept their insertion order, i.e.
            // ascending ids, which only a stable sort does.)
            int off = mode == 2;
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                if (r < E && idx < n && idx) off |= keys[idx - 1] > key[r];
            }
            sorted = __syncthreads_or(off) == 0;
            if (!sorted) {
                if (tid == 0) atomicAdd(n_redone, 1u);
                load_bucket<EMAX>(key, c, vals, vw, a0, n, E, lo_bit, hi_bit);  // (the insertion order again)
            }
        }
        if (!sorted) radix_passes<true, EMAX>(key, keys, cnt, wtot, n, E, lo_bit, hi_bit);
        // duplicates against the left neighbour (the first composite of a bucket starts a k-mer: buckets differ in their top bits)
        if (tid == 0) { s_nk = 0; s_np = 0; }
        if (lo_bit >= hi_bit) {  // nothing was sorted (the split covered every T bit): the keys are only in registers yet
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                if (r < E && idx < n) keys[idx] = key[r];
            }
        }
        __syncthreads();
        uint32_t nk = 0, np = 0;
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
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


// The same for the small buckets, ONE WAVEFRONT per bucket (up to 64 EW composites): no barrier anywhere -- the LDS operations of a
// wavefront are in order --, four independent buckets per workgroup.  The workgroup version above spends most of a pass waiting at
// its four barriers (1300 cycles per bucket and pass on a CU where the arithmetic is 200); a pan-genome bucket is ~760 composites.
template <int EW>
__global__ __launch_bounds__(FB_BLOCK, (EW <= 8 ? 5 : 4)) void k_bucket_sort_wave(uint64_t* __restrict__ c, const uint32_t* __restrict__ boff, uint32_t nb, uint32_t lo_bit, uint32_t hi_bit,
                                                               uint64_t* __restrict__ counts, const void* __restrict__ vals, uint32_t vw, int mode,
                                                               uint32_t* __restrict__ n_redone) {
    __shared__ uint64_t keys_all[FB_WAVES][64 * EW];
    __shared__ uint32_t cnt_all[FB_WAVES][FB_DIGITS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint64_t* keys = keys_all[wave];
    uint32_t* cnt = cnt_all[wave];
    for (uint32_t b = blockIdx.x * FB_WAVES + wave; b < nb; b += gridDim.x * FB_WAVES) {
        const uint32_t a0 = boff[b], n = boff[b + 1] - a0;
        if (n > 64u * EW) continue;  // (k_bucket_sort's)
        if (n == 0) {
            if (lane == 0) counts[b] = 0;
            continue;
        }
        const uint32_t E = (n + 63u) / 64u;  // rounds: slot (round r, lane l) = composite r 64 + l of the bucket
        uint64_t key[EW];
        bool sorted = false;
        for (int attempt = (mode == 1 ? 1 : 0); attempt < 2 && !sorted; attempt++) {  // 0: ranks from LDS atomics, checked; 1: from ballots (see radix_passes)
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EW; r++) {
                const uint32_t idx = r * 64u + lane;
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
                for (uint32_t j = 0; j < FB_DIGITS / 64; j++) cnt[lane * (FB_DIGITS / 64) + j] = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                uint32_t rank[EW];
#pragma unroll
                for (uint32_t r = 0; r < (uint32_t)EW; r++) {
                    rank[r] = 0;
                    if (r >= E) continue;  // (uniform)
                    const bool valid = r * 64u + lane < n;
                    const uint32_t d = (uint32_t)(key[r] >> bit) & mask;
                    if (attempt == 0) {
                        if (valid) rank[r] = atomicAdd(&cnt[d], 1u);
                    } else {
                        const uint64_t peers = match_digit(d, valid, nbits);
                        if (valid) {
                            const int leader = __builtin_ctzll(peers);
                            uint32_t base = 0;
                            if ((int)lane == leader) {
                                base = cnt[d];
                                cnt[d] = base + (uint32_t)__builtin_popcountll(peers);
                            }
                            base = __shfl(base, leader);
                            rank[r] = base + (uint32_t)__builtin_popcountll(peers & lt_mask);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
                {   // the lane's digits: counts -> starts
                    constexpr int DPL = FB_DIGITS / 64;
                    uint32_t cw[DPL], tot = 0;
#pragma unroll
                    for (int j = 0; j < DPL; j++) { cw[j] = cnt[lane * DPL + j]; tot += cw[j]; }
                    uint32_t inc = tot;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const uint32_t v = __shfl_up(inc, o);
                        if ((int)lane >= o) inc += v;
                    }
                    uint32_t start = inc - tot;
#pragma unroll
                    for (int j = 0; j < DPL; j++) { cnt[lane * DPL + j] = start; start += cw[j]; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                for (uint32_t r = 0; r < (uint32_t)EW; r++) {
                    if (r >= E) continue;
                    if (r * 64u + lane < n) keys[cnt[(uint32_t)(key[r] >> bit) & mask] + rank[r]] = key[r];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                for (uint32_t r = 0; r < (uint32_t)EW; r++) {
                    const uint32_t idx = r * 64u + lane;
                    if (r < E && idx < n) key[r] = keys[idx];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            if (lo_bit >= hi_bit) {  // nothing to sort: the keys are only in registers yet
#pragma unroll
                for (uint32_t r = 0; r < (uint32_t)EW; r++) {
                    const uint32_t idx = r * 64u + lane;
                    if (r < E && idx < n) keys[idx] = key[r];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            sorted = true;
            if (attempt == 0 && lo_bit < hi_bit) {  // in order over EVERY bit (see k_bucket_sort)?
                int off = mode == 2;
#pragma unroll
                for (uint32_t r = 0; r < (uint32_t)EW; r++) {
                    const uint32_t idx = r * 64u + lane;
                    if (r < E && idx < n && idx) off |= keys[idx - 1] > key[r];
                }
                sorted = __ballot(off != 0) == 0ull;
                if (!sorted && lane == 0) atomicAdd(n_redone, 1u);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        uint32_t nk = 0, np = 0;
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)EW; r++) {
            const uint32_t idx = r * 64u + lane;
            if (r < E && idx < n) {
                const uint64_t prev = idx ? keys[idx - 1] : ~key[r];
                nk += (key[r] >> lo_bit) != (prev >> lo_bit);
                np += key[r] != prev;
                c[a0 + idx] = key[r];
            }
        }
        for (int o = 32; o > 0; o >>= 1) { nk += __shfl_down(nk, o); np += __shfl_down(np, o); }
        if (lane == 0) counts[b] = ((uint64_t)nk << 32) | np;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next bucket overwrites keys)
    }
}

// what the host waits for: slots[0] = *a (when given), slots[1] = *b
__global__ void k_front_publish(const uint64_t* __restrict__ a, const uint32_t* __restrict__ b, uint64_t* __restrict__ slots) {
    if (a) { slots[0] = *a; slots[1] = *b; }
    else slots[0] = *b;
}

// bases[b] = exclusive scan of counts (k-mers << 32 | pairs).  Every bucket places its k-mers (T = c >> gb), the offset of each k-mer's
// first genome id, and the genome ids (c & gmask) of its distinct pairs.
__global__ __launch_bounds__(FB_BLOCK) void k_bucket_emit(const uint64_t* __restrict__ c, const uint32_t* __restrict__ boff, uint32_t nb, uint32_t gb,
                                                          const uint64_t* __restrict__ bases, uint64_t* __restrict__ tk, uint32_t* __restrict__ seg_off,
                                                          uint32_t* __restrict__ pg, int kv, uint32_t rest,  // kv: the composites lack the bucket's
                                                          uint32_t nk, uint32_t np) {                        // bits (k_bucket_sort, vals); rest: T bits below the split
    __shared__ uint32_t w_nk[FB_WAVES], w_np[FB_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t gmask = (1ull << gb) - 1ull, lt_mask = (1ull << lane) - 1ull;
    if (blockIdx.x == 0 && tid == 0) seg_off[nk] = np;  // the offsets array has nk + 1 entries
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


// ---- two-word keys (33 <= k <= 64) --------------------------------------------------------------------------------------------------------
// insertKmer_Node_special (src/insertNode.c:241-423) for keys of two words.  The split hands over items (hk, lo, id): hk = the top 64 bits of
// the T-form, left-aligned (its top 18 bits = the root prefix = the bucket), lo = the 2k - 64 bits below them, id = the genome.  Inside a
// bucket 46 + (2k - 64) key bits are left -- 108 at k = 63: twelve 9-bit passes where a one-word bucket takes four.  But a pan-genome
// bucket holds few DISTINCT k-mers (config 5: one locus, ~2000 items of ~450 k-mers), so the items are GROUPED first and only the distinct
// k-mers are ordered:
//   1. every item gets a 27-bit hash of its key bits; (hash << 13 | position in the bucket) is sorted on the hash by the same three LDS
//      radix passes the one-word buckets use (radix_passes): equal k-mers end up next to each other, in insertion order (ascending ids);
//   2. an item that differs from its left neighbour starts a k-mer; two DIFFERENT neighbours with one hash are a collision: the bucket
//      is done again with another seed (expected for ~10^-3 of the buckets, four seeds);
//   3. the distinct k-mers (at most half the capacity) go to LDS and are ranked by comparison -- m^2 / 256 comparisons of 16 bytes per
//      thread: ~3 us at m = 450 --; a scan over the ranks gives every k-mer's place among the bucket's kept pairs;
//   4. the items go back in place in (k-mer, id) order, duplicated pairs dropped; counts as for one-word buckets; k_bucket2_emit
//      then streams k-mers, offsets and ids out.
struct __attribute__((packed, aligned(4))) BftItem2 {
    uint64_t lo;
    uint32_t id;
};

__device__ __forceinline__ uint32_t hash27(uint64_t a, uint64_t b, uint32_t seed) {
    uint64_t x = (a ^ (0x9E3779B97F4A7C15ull * (seed + 1))) * 0xBF58476D1CE4E5B9ull;
    x ^= x >> 29;
    x += b * 0x94D049BB133111EBull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 31;
    return (uint32_t)(x >> 37);
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int l) {
    return ((uint64_t)(uint32_t)__shfl((uint32_t)(v >> 32), l) << 32) | (uint32_t)__shfl((uint32_t)v, l);
}

// fail[0]: buckets that could not be done here (beyond the capacity, more than half of it distinct, four colliding seeds): the caller falls
// back to the device-wide sort.  min_n / max_n: the sizes this launch takes (the variants differ in EMAX).
template <int EMAX>
__global__ __launch_bounds__(FB_BLOCK, (EMAX <= 8 ? 4 : EMAX <= 16 ? 2 : 1)) void k_bucket2_sort(uint64_t* __restrict__ hk, BftItem2* __restrict__ it, const uint32_t* __restrict__ boff, uint32_t nb,
                                                                                          uint64_t* __restrict__ counts, uint32_t* __restrict__ fail, uint32_t min_n, uint32_t max_n) {
    constexpr uint32_t CAP = FB_BLOCK * EMAX, MC = CAP / 2;
    __shared__ uint64_t keys[CAP];                 // the sort's exchange buffer, then the distinct k-mers: (hk46, lo) x MC
    __shared__ uint32_t cnt[FB_WAVES][FB_DIGITS];  // digit counters, then ranks of the distinct k-mers (MC <= 4 x 512 words at EMAX = 16; EMAX = 32: see dextra)
    __shared__ uint32_t dkoff[MC + 1];             // first kept pair of a distinct k-mer (in hash order)
    __shared__ uint32_t lenr[MC];                  // pairs of the k-mer of a rank, then its first pair among the bucket's
    __shared__ uint32_t drank_x[EMAX > 16 ? MC : 1];
    __shared__ uint32_t wtot[FB_WAVES];
    __shared__ uint64_t wlast_k[FB_WAVES][2];
    __shared__ uint32_t wlast_i[FB_WAVES], wlast_h[FB_WAVES];
    __shared__ uint32_t w_nh[FB_WAVES], w_nk[FB_WAVES];
    __shared__ uint32_t s_flag;
    uint32_t* drank = EMAX > 16 ? drank_x : &cnt[0][0];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t a0 = boff[b], n = boff[b + 1] - a0;
        if (n < min_n || n > max_n) continue;
        if (n == 0) {
            if (tid == 0) counts[b] = 0;
            continue;
        }
        if (n > CAP) {  // (excluded by the caller; counted in case)
            if (tid == 0) { counts[b] = 0; atomicAdd(fail, 1u); }
            continue;
        }
        const uint32_t E = (n + FB_BLOCK - 1) / FB_BLOCK;  // rounds
        const uint32_t wbase = wave * E * 64u;
        bool ok = false;
        uint64_t ihk[EMAX], ilo[EMAX];
        uint32_t iid[EMAX];
        uint32_t hidx[EMAX], kidx[EMAX];  // per item: index of its k-mer among the distinct ones (hash order); index among the kept pairs
        uint64_t flags_head = 0, flags_keep = 0;  // bit r: the item of round r starts a k-mer / is kept
        uint32_t m = 0, np = 0;
        for (uint32_t seed = 0; seed < 4 && !ok; seed++) {
            // 1. hash + position, sorted on the hash
            uint64_t key[EMAX];
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                key[r] = ~0ull;
                if (r < E && idx < n) {
                    const uint64_t a = hk[a0 + idx] & ((1ull << 46) - 1ull);
                    const uint64_t l = it[a0 + idx].lo;
                    key[r] = ((uint64_t)hash27(a, l, seed) << 13) | idx;
                }
            }
            radix_passes<true, EMAX>(key, keys, cnt, wtot, n, E, 13, 40);  // (ranks from ballots: stable by construction -- the order of equal hashes IS the result here)
            __syncthreads();
            // 2. the items in hash order; neighbours: lane - 1 of the round, lane 63 of the round before, the last item of the wavefront before
            if (tid == 0) s_flag = 0;
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                ihk[r] = 0; ilo[r] = 0; iid[r] = 0;
                if (r < E && idx < n) {
                    const uint32_t src = (uint32_t)(key[r] & 0x1FFFu);
                    ihk[r] = hk[a0 + src] & ((1ull << 46) - 1ull);
                    const BftItem2 x = it[a0 + src];
                    ilo[r] = x.lo;
                    iid[r] = x.id;
                }
            }
            {   // the last item of every wavefront, for the first of the next
                const uint32_t nw = wbase < n ? min(n - wbase, E * 64u) : 0u;  // items of this wavefront
                if (nw) {
                    const uint32_t lr = (nw - 1) / 64u, ll = (nw - 1) % 64u;
                    uint64_t lk = 0, llo = 0;
                    uint32_t li = 0, lh = 0;
#pragma unroll
                    for (uint32_t r = 0; r < (uint32_t)EMAX; r++)
                        if (r == lr) { lk = ihk[r]; llo = ilo[r]; li = iid[r]; lh = (uint32_t)(key[r] >> 13); }
                    if (lane == ll) { wlast_k[wave][0] = lk; wlast_k[wave][1] = llo; wlast_i[wave] = li; wlast_h[wave] = lh; }
                }
            }
            __syncthreads();
            flags_head = 0;
            flags_keep = 0;
            uint32_t nh_w = 0, nk_w = 0;
            bool coll = false;
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                const bool valid = r < E && idx < n;
                uint64_t pk = shfl64(ihk[r], (int)((lane + 63u) & 63u)), pl = shfl64(ilo[r], (int)((lane + 63u) & 63u));
                uint32_t pi = __shfl(iid[r], (int)((lane + 63u) & 63u)), ph = __shfl((uint32_t)(key[r] >> 13), (int)((lane + 63u) & 63u));
                // (lane 0: lane 63 of the round before -- every round but a wavefront's last is full --, or the last item of the wavefront before; the
                // shuffles by every lane: no lane-dependent control flow around them)
                const uint64_t qk = r ? shfl64(ihk[r ? r - 1 : 0], 63) : 0ull, ql = r ? shfl64(ilo[r ? r - 1 : 0], 63) : 0ull;
                const uint32_t qi = r ? __shfl(iid[r ? r - 1 : 0], 63) : 0u, qh = r ? __shfl((uint32_t)(key[r ? r - 1 : 0] >> 13), 63) : 0u;
                if (lane == 0) {
                    if (r > 0) { pk = qk; pl = ql; pi = qi; ph = qh; }
                    else if (wave > 0) { pk = wlast_k[wave - 1][0]; pl = wlast_k[wave - 1][1]; pi = wlast_i[wave - 1]; ph = wlast_h[wave - 1]; }
                }
                const bool first = idx == 0;
                const bool same = valid && !first && pk == ihk[r] && pl == ilo[r];
                const bool head = valid && !same, keep = valid && (!same || pi != iid[r]);
                coll |= valid && !first && !same && ph == (uint32_t)(key[r] >> 13);
                const uint64_t hm = __ballot(head), km = __ballot(keep);
                hidx[r] = nh_w + (uint32_t)__builtin_popcountll(hm & lt_mask) + (head ? 1u : 0u);  // (inclusive: heads up to and including me, in the wavefront)
                kidx[r] = nk_w + (uint32_t)__builtin_popcountll(km & lt_mask);                     // (exclusive)
                nh_w += (uint32_t)__builtin_popcountll(hm);
                nk_w += (uint32_t)__builtin_popcountll(km);
                flags_head |= (uint64_t)head << r;
                flags_keep |= (uint64_t)keep << r;
            }
            if (__ballot(coll) && lane == 0) s_flag = 1;
            if (lane == 0) { w_nh[wave] = nh_w; w_nk[wave] = nk_w; }
            __syncthreads();
            ok = s_flag == 0;
            uint32_t hb = 0, kb = 0;
            m = 0; np = 0;
#pragma unroll
            for (int w = 0; w < FB_WAVES; w++) {
                if (w < (int)wave) { hb += w_nh[w]; kb += w_nk[w]; }
                m += w_nh[w];
                np += w_nk[w];
            }
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EMAX; r++) { hidx[r] += hb; kidx[r] += kb; }  // hidx: 1-based index of the item's k-mer
            __syncthreads();  // (s_flag, w_nh are rewritten by the next seed / bucket)
        }
        if (!ok || m > MC) {
            if (tid == 0) { counts[b] = 0; atomicAdd(fail, 1u); }
            continue;
        }
        // 3. the distinct k-mers to LDS (over the exchange buffer: every wavefront holds its items in registers), ranked by comparison
        uint64_t* dk = keys;
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)EMAX; r++)
            if ((flags_head >> r) & 1ull) {
                dk[2 * (hidx[r] - 1)] = ihk[r];
                dk[2 * (hidx[r] - 1) + 1] = ilo[r];
                dkoff[hidx[r] - 1] = kidx[r];
            }
        if (tid == 0) dkoff[m] = np;
        __syncthreads();
        for (uint32_t i = tid; i < m; i += FB_BLOCK) {
            const uint64_t a = dk[2 * i], c = dk[2 * i + 1];
            uint32_t rk = 0;
            for (uint32_t j = 0; j < m; j++) {
                const uint64_t x = dk[2 * j], y = dk[2 * j + 1];
                rk += (x < a || (x == a && y < c)) ? 1u : 0u;
            }
            drank[i] = rk;
        }
        __syncthreads();
        // lengths by rank -> starts by rank (a scan over m <= MC values, FB_BLOCK x (MC / FB_BLOCK) each)
        for (uint32_t i = tid; i < m; i += FB_BLOCK) lenr[drank[i]] = dkoff[i + 1] - dkoff[i];
        __syncthreads();
        {
            constexpr uint32_t PT = MC / FB_BLOCK;
            uint32_t v[PT], sum = 0;
#pragma unroll
            for (uint32_t q = 0; q < PT; q++) {
                const uint32_t i = tid * PT + q;
                v[q] = i < m ? lenr[i] : 0u;
                sum += v[q];
            }
            uint32_t inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(inc, o);
                if ((int)lane >= o) inc += y;
            }
            if (lane == 63) wtot[wave] = inc;
            __syncthreads();
            uint32_t run = inc - sum;
#pragma unroll
            for (int w = 0; w < FB_WAVES; w++)
                if (w < (int)wave) run += wtot[w];
#pragma unroll
            for (uint32_t q = 0; q < PT; q++) {
                const uint32_t i = tid * PT + q;
                if (i < m) lenr[i] = run;
                run += v[q];
            }
        }
        __syncthreads();
        // 4. back in place, in (k-mer, id) order
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)EMAX; r++)
            if ((flags_keep >> r) & 1ull) {
                const uint32_t hi = hidx[r] - 1;
                const uint32_t dest = a0 + lenr[drank[hi]] + (kidx[r] - dkoff[hi]);
                hk[dest] = ((uint64_t)b << 46) | ihk[r];
                BftItem2 x;
                x.lo = ilo[r];
                x.id = iid[r];
                it[dest] = x;
            }
        if (tid == 0) counts[b] = ((uint64_t)m << 32) | np;
        __syncthreads();
    }
}

// bases[b] = exclusive scan of counts (k-mers << 32 | pairs).  Every bucket places its k-mers (two words: the T-form back from hk and lo), the offset
// of each k-mer's first genome id, and the genome ids of its pairs (the bucket's first `pairs` items, in (k-mer, id) order).
__global__ __launch_bounds__(FB_BLOCK) void k_bucket2_emit(const uint64_t* __restrict__ hk, const BftItem2* __restrict__ it, const uint32_t* __restrict__ boff, uint32_t nb,
                                                           const uint64_t* __restrict__ counts, const uint64_t* __restrict__ bases, uint64_t* __restrict__ tk, uint32_t* __restrict__ seg_off,
                                                           uint32_t* __restrict__ pg, uint32_t sh, uint32_t nk, uint32_t np) {  // sh = 2k - 64: bits of lo
    __shared__ uint32_t w_nk[FB_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    if (blockIdx.x == 0 && tid == 0) seg_off[nk] = np;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t a0 = boff[b], n = (uint32_t)counts[b];  // (the kept pairs)
        if (n == 0) continue;
        const uint64_t base = bases[b];
        uint32_t kbase = (uint32_t)(base >> 32);
        const uint32_t pbase = (uint32_t)base;
        for (uint32_t r0 = 0; r0 < n; r0 += FB_BLOCK) {
            const uint32_t idx = r0 + tid;
            const bool valid = idx < n;
            uint64_t a = 0, l = 0, pa = 0, pl = 0;
            uint32_t id = 0;
            if (valid) {
                a = hk[a0 + idx];
                const BftItem2 x = it[a0 + idx];
                l = x.lo;
                id = x.id;
                if (idx) { pa = hk[a0 + idx - 1]; pl = it[a0 + idx - 1].lo; }
            }
            const bool head = valid && (idx == 0 || a != pa || l != pl);
            const uint64_t hm = __ballot(head);
            if (lane == 0) w_nk[wave] = (uint32_t)__builtin_popcountll(hm);
            __syncthreads();
            uint32_t kb = kbase, tk_all = 0;
#pragma unroll
            for (int w = 0; w < FB_WAVES; w++) {
                if (w < (int)wave) kb += w_nk[w];
                tk_all += w_nk[w];
            }
            if (valid) pg[pbase + idx] = id;
            if (head) {
                const uint32_t q = kb + (uint32_t)__builtin_popcountll(hm & lt_mask);
                // T-form: hk holds its top 64 bits left-aligned, lo the sh bits below them
                const uint64_t t0 = sh == 64 ? a : a >> (64 - sh);
                const uint64_t t1 = sh == 64 ? l : ((a << sh) | l);
                tk[2ull * q] = t0;
                tk[2ull * q + 1] = t1;
                seg_off[q] = pbase + idx;
            }
            kbase += tk_all;
            __syncthreads();
        }
    }
}

}  // namespace

uint32_t bft_front_bucket_capacity(void) { return FB_CAP; }
static int g_rank_mode = 0;                 // k_bucket_sort's mode ("test_front_rank_mode")
void bft_test_front_rank_mode(int mode) { g_rank_mode = mode; }

int bft_front_buckets(uint64_t* d_c, uint64_t n, const uint32_t* d_boff, uint32_t nb, uint32_t gb, uint32_t split_bit, hipStream_t s, DevBuf& tk, DevBuf& seg_off,
                      DevBuf& pg, uint64_t& nk, uint64_t& np, const uint32_t* d_max_bucket, uint32_t* max_bucket, bool* done, uint32_t* n_redone,
                      const void* d_vals, uint32_t vw) {
    *done = false;
    PinBlock pin;  // [0] the largest bucket, [1] k-mers << 32 | pairs, [2] buckets sorted again
    if (!pin.p) return bft_fail(BFT_GPU_E_HIP, "hipHostMalloc (front end counts)");
    DevBuf counts, bases, tmp;
    CK(counts.alloc(((uint64_t)nb + 1) * 8));
    CK(bases.alloc(((uint64_t)nb + 1) * 8));
    HIPCK(hipMemsetAsync((uint8_t*)counts.p + (uint64_t)nb * 8, 0, 8, s));
    const dim3 grid(std::min<uint32_t>(nb, 256u * 16u)), block(FB_BLOCK);
    DevBuf redone;
    CK(redone.alloc_zero(4, s));
    // The size of the largest bucket decides which workgroup variant the larger buckets need, and whether the buckets fit at all; it
    // travels to the host behind the split while the wavefront kernel -- whose own variant follows the MEAN bucket -- is already
    // running (a synchronisation in front of it left the GPU idle for ~0.15 ms).
    hipEvent_t ev;
    HIPCK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipLaunchKernelGGL(k_front_publish, dim3(1), dim3(1), 0, s, (const uint64_t*)nullptr, d_max_bucket, pin.p);
    hipError_t e = hipEventRecord(ev, s);
    // the small buckets a wavefront each, the others a workgroup each (one after the other: on two streams the two kernels --
    // both latency-bound, different LDS footprints -- got in each other's way: 8.6 ms instead of 2.0 + 0.8)
    const uint32_t wave_cap = n / std::max(nb, 1u) <= 192u ? 512u : 1024u;
    const dim3 wgrid(std::min<uint32_t>((nb + FB_WAVES - 1) / FB_WAVES, 256u * 16u));
    if (wave_cap == 512u) hipLaunchKernelGGL(k_bucket_sort_wave<8>, wgrid, block, 0, s, d_c, d_boff, nb, gb, split_bit, counts.as<uint64_t>(), d_vals, vw, g_rank_mode, redone.as<uint32_t>());
    else hipLaunchKernelGGL(k_bucket_sort_wave<16>, wgrid, block, 0, s, d_c, d_boff, nb, gb, split_bit, counts.as<uint64_t>(), d_vals, vw, g_rank_mode, redone.as<uint32_t>());
    if (e == hipSuccess) e = hipEventSynchronize(ev);
    (void)hipEventDestroy(ev);
    HIPCK(e);
    const uint32_t mx = (uint32_t)pin.p[0];
    *max_bucket = mx;
    bft_trace_mark("root-prefix split done (largest bucket known)");
    if (mx > FB_CAP) return 0;
#define FB_LAUNCH(E) hipLaunchKernelGGL(k_bucket_sort<E>, grid, block, 0, s, d_c, d_boff, nb, gb, split_bit, counts.as<uint64_t>(), d_vals, vw, g_rank_mode, redone.as<uint32_t>(), wave_cap)
    if (mx <= wave_cap) {}
    else if (mx <= 256u * 6u) FB_LAUNCH(6);
    else if (mx <= 256u * 9u) FB_LAUNCH(9);
    else if (mx <= 256u * 12u) FB_LAUNCH(12);
    else FB_LAUNCH(16);
#undef FB_LAUNCH
    CK(bft_scan::exclusive_sum_ptr<uint64_t>(counts.as<uint64_t>(), bases.as<uint64_t>(), (uint64_t)nb + 1, s, tmp));
    hipLaunchKernelGGL(k_front_publish, dim3(1), dim3(1), 0, s, bases.as<uint64_t>() + nb, redone.as<uint32_t>(), pin.p + 1);
    HIPCK(hipGetLastError());
    bft_stage("bucket sorts in LDS (+ scan of the counts)", (double)n * 16 + (d_vals ? (double)n * vw : 0.0), s);
    HIPCK(hipStreamSynchronize(s));
    const uint64_t total = pin.p[1];
    if (n_redone) *n_redone = (uint32_t)pin.p[2];
    nk = total >> 32;
    np = total & 0xFFFFFFFFull;
    bft_trace_mark("bucket sort done (sync)");
    CK(tk.alloc(nk * 8));
    CK(seg_off.alloc((nk + 1) * 4));
    CK(pg.alloc(np * 4));
    // (seg_off[nk] = np is the kernel's too: nothing of this call is left on the host's side when it returns, the stream goes on)
    hipLaunchKernelGGL(k_bucket_emit, grid, block, 0, s, d_c, d_boff, nb, gb, bases.as<uint64_t>(), tk.as<uint64_t>(), seg_off.as<uint32_t>(), pg.as<uint32_t>(), d_vals ? 1 : 0, split_bit - gb,
                       (uint32_t)nk, (uint32_t)np);
    HIPCK(hipGetLastError());
    bft_stage("bucket emit (k-mers, offsets, genome ids)", (double)n * 8 + (double)nk * 12 + (double)np * 4, s);
    *done = true;
    return 0;
}

uint32_t bft_front2_bucket_capacity(void) { return FB_BLOCK * 32; }

// the two-word front end behind the split (items grouped by the top 18 bits of hk): see k_bucket2_sort
int bft_front2_buckets(uint64_t* d_hk, void* d_items, uint64_t n, const uint32_t* d_boff, uint32_t nb, int k, hipStream_t s, DevBuf& tk, DevBuf& seg_off, DevBuf& pg, uint64_t& nk, uint64_t& np,
                       const uint32_t* d_max_bucket, uint32_t* max_bucket, bool* done) {
    *done = false;
    BftItem2* d_it = (BftItem2*)d_items;
    PinBlock pin;  // [0] the largest bucket, [1] k-mers << 32 | pairs, [2] buckets that failed
    if (!pin.p) return bft_fail(BFT_GPU_E_HIP, "hipHostMalloc (front end counts)");
    DevBuf counts, bases, tmp, fail;
    CK(counts.alloc(((uint64_t)nb + 1) * 8));
    CK(bases.alloc(((uint64_t)nb + 1) * 8));
    CK(fail.alloc_zero(4, s));
    HIPCK(hipMemsetAsync((uint8_t*)counts.p + (uint64_t)nb * 8, 0, 8, s));
    hipLaunchKernelGGL(k_front_publish, dim3(1), dim3(1), 0, s, (const uint64_t*)nullptr, d_max_bucket, pin.p);
    HIPCK(hipStreamSynchronize(s));
    const uint32_t mx = (uint32_t)pin.p[0];
    *max_bucket = mx;
    if (mx > FB_BLOCK * 32) return 0;
    const dim3 block(FB_BLOCK);
    // by size: the variants differ in the items a thread holds (registers, LDS, workgroups per CU)
    hipLaunchKernelGGL(k_bucket2_sort<8>, dim3(std::min<uint32_t>(nb, 256u * 8u)), block, 0, s, d_hk, d_it, d_boff, nb, counts.as<uint64_t>(), fail.as<uint32_t>(), 0u, FB_BLOCK * 8u);
    if (mx > FB_BLOCK * 8u)
        hipLaunchKernelGGL(k_bucket2_sort<16>, dim3(std::min<uint32_t>(nb, 256u * 8u)), block, 0, s, d_hk, d_it, d_boff, nb, counts.as<uint64_t>(), fail.as<uint32_t>(), FB_BLOCK * 8u + 1u,
                           FB_BLOCK * 16u);
    if (mx > FB_BLOCK * 16u)
        hipLaunchKernelGGL(k_bucket2_sort<32>, dim3(std::min<uint32_t>(nb, 256u * 4u)), block, 0, s, d_hk, d_it, d_boff, nb, counts.as<uint64_t>(), fail.as<uint32_t>(), FB_BLOCK * 16u + 1u,
                           FB_BLOCK * 32u);
    CK(bft_scan::exclusive_sum_ptr<uint64_t>(counts.as<uint64_t>(), bases.as<uint64_t>(), (uint64_t)nb + 1, s, tmp));
    hipLaunchKernelGGL(k_front_publish, dim3(1), dim3(1), 0, s, bases.as<uint64_t>() + nb, fail.as<uint32_t>(), pin.p + 1);
    HIPCK(hipGetLastError());
    bft_stage("two-word buckets: grouped by hash, distinct k-mers ranked", (double)n * 2 * 20, s);
    HIPCK(hipStreamSynchronize(s));
    if (pin.p[2] != 0) return 0;  // (a bucket beyond this front end: the caller sorts device-wide)
    const uint64_t total = pin.p[1];
    nk = total >> 32;
    np = total & 0xFFFFFFFFull;
    CK(tk.alloc(nk * 16));
    CK(seg_off.alloc((nk + 1) * 4));
    CK(pg.alloc(np * 4));
    hipLaunchKernelGGL(k_bucket2_emit, dim3(std::min<uint32_t>(nb, 256u * 16u)), block, 0, s, d_hk, d_it, d_boff, nb, counts.as<uint64_t>(), bases.as<uint64_t>(), tk.as<uint64_t>(),
                       seg_off.as<uint32_t>(), pg.as<uint32_t>(), (uint32_t)(2 * k - 64), (uint32_t)nk, (uint32_t)np);
    HIPCK(hipGetLastError());
    bft_stage("two-word buckets: emit (k-mers, offsets, genome ids)", (double)np * 24 + (double)nk * 20, s);
    *done = true;
    return 0;
}
