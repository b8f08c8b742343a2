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
//                   every bit, and a bucket that fails goes on a list that a second launch (k_bucket_sort<.., true>) sorts again with
//                   ranks from wavefront ballots (stable by construction; a launch of its own so that its registers do not make the
//                   main kernels spill).  From the second pass on, one atomic serves a whole run of equal digits (rank_runs).
//   k_bucket_sort   the same with one workgroup per bucket, for the larger ones (up to 4096 composites; variants by keys per thread)
//   k_bucket2_*     the same front end for two-word keys (33 <= k <= 64): see BftItem2
//   k_bucket_emit   after one scan of the 2^18 count pairs: every bucket writes its k-mers, their offsets and the genome ids at
//                   its place in the outputs.
// Composites that do not fit 63 bits (k = 31; k = 27 beyond 512 genomes) arrive as whole k-mers with the ids beside them: the bits a
// bucket's k-mers share are dropped inside the bucket, which makes room for the id.
// One read and one write of the array for all the remaining bits, where a device-wide LSD sort spends a pass per 8 bits
// (a library radix sort: 7 passes over 2x10^8 composites, 7.6 ms; its segmented sort of the same buckets: 4.5 ms; these kernels: 2.0 ms).
#include "bft_dev.h"
#include "bft_scan.h"
#include "bft_sort.h"

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

// Ranks from ONE LDS atomic per run of equal neighbouring digits.  From a bucket's second pass on, the composites of one k-mer lie side by
// side (they agree in every digit, and the pass before was stable): a k-mer that ten genomes hold is ten atomics on one address, which the LDS
// serves one after the other.  Lanes whose digit equals their left neighbour's form a run; the run's first lane reserves the run's ranks
// with one atomic and hands its base on.  Runs of one digit that do not touch are ordered as single lanes are (see radix_passes: checked).
// Measured on config 3's buckets (~760 composites, 4.5 per k-mer; profiles/r06/pmc_build.txt): k_bucket_sort_wave 1.90 -> 1.39 ms,
// SQ_LDS_BANK_CONFLICT 5.0 -> 1.6 x10^8 cycles, SQ_WAIT_INST_LDS 12.6 -> 0.5 x10^8 (profiles/r06/bucket_experiments_2.txt).
__device__ __forceinline__ uint32_t rank_runs(uint32_t* cnt, uint32_t d, bool valid, uint32_t lane) {
    const uint32_t dprev = (uint32_t)__builtin_amdgcn_update_dpp((int)d, (int)d, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const uint64_t hm = __ballot(valid && (lane == 0 || d != dprev)), vm = __ballot(valid);
    uint32_t rank = 0;
    if (valid) {  // (the lanes that are not valid are the wavefront's last: a valid lane's left neighbour is valid)
        const uint64_t upto = hm & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull)), above = lane == 63 ? 0ull : (hm >> (lane + 1));
        const uint32_t start = 63u - (uint32_t)__builtin_clzll(upto);
        uint32_t base = 0;
        if (start == lane) {
            const uint32_t next = above ? lane + 1u + (uint32_t)__builtin_ctzll(above) : (uint32_t)__builtin_popcountll(vm);
            base = atomicAdd(&cnt[d], next - lane);
        }
        rank = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(start * 4u), (int)base) + (lane - start);
    }
    return rank;
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
            } else if (bit > lo_bit) {
                rank[r] = rank_runs(cnt[wave], d, valid, lane);
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
// REDO: the fallback launch -- the buckets on the `redo` list (*n_redone of them), ranks from ballots.  It is a launch of its own because the
// ballot passes beside the atomic ones cost the kernel registers it then spills: a kernel that uses scratch memory starts ~0.13 ms late
// whenever the kernels before it used none (the runtime hands a queue's scratch back and has to set it up again).
template <int EMAX, bool REDO>
__global__ __launch_bounds__(FB_BLOCK, (EMAX <= 6 ? 6 : EMAX <= 9 ? 5 : EMAX <= 12 ? 4 : 3)) void k_bucket_sort(uint64_t* __restrict__ c, const uint32_t* __restrict__ boff, uint32_t nb, uint32_t lo_bit, uint32_t hi_bit,
                                                          uint64_t* __restrict__ counts, const void* __restrict__ vals, uint32_t vw, int mode, uint32_t* __restrict__ n_redone,
                                                          uint32_t* __restrict__ redo, uint32_t min_n) {  // buckets of up to min_n composites are k_bucket_sort_wave's
    __shared__ uint64_t keys[FB_BLOCK * EMAX];
    __shared__ uint32_t cnt[FB_WAVES][FB_DIGITS];
    __shared__ uint32_t wtot[FB_WAVES];
    __shared__ uint32_t s_nk, s_np;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_items = REDO ? min(*n_redone, nb) : nb;
    for (uint32_t bi = blockIdx.x; bi < n_items; bi += gridDim.x) {
        const uint32_t b = REDO ? redo[bi] : bi;
        const uint32_t a0 = boff[b], n = boff[b + 1] - a0;
        if (!REDO && n <= min_n && min_n) continue;
        if (n == 0 || n > (uint32_t)(FB_BLOCK * EMAX)) {  // (buckets beyond the capacity are excluded by the caller: the whole build then takes the one-sort path)
            if (tid == 0) counts[b] = 0;
            continue;
        }
        const uint32_t E = (n + FB_BLOCK - 1) / FB_BLOCK;       // rounds
        const uint32_t wbase = wave * E * 64u;
        if (!REDO && mode == 1 && lo_bit < hi_bit) {  // (ballots only: everything is the fallback's)
            if (tid == 0) redo[atomicAdd(n_redone, 1u)] = b;
            continue;
        }
        uint64_t key[EMAX];
        load_bucket<EMAX>(key, c, vals, vw, a0, n, E, lo_bit, hi_bit);
        if (!REDO && lo_bit < hi_bit) {
            radix_passes<false, EMAX>(key, keys, cnt, wtot, n, E, lo_bit, hi_bit);
            // In order over EVERY bit?  (The ids below lo_bit are not sorted on: equal k-mers must have kept their insertion order, i.e.
            // ascending ids, which only a stable sort does.)
            int off = mode == 2;
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                if (r < E && idx < n && idx) off |= keys[idx - 1] > key[r];
            }
            if (__syncthreads_or(off) != 0) {  // out of order: the fallback launch sorts the bucket again, from the insertion order still in `c`
                if (tid == 0) redo[atomicAdd(n_redone, 1u)] = b;
                continue;
            }
        }
        if (REDO) radix_passes<true, EMAX>(key, keys, cnt, wtot, n, E, lo_bit, hi_bit);
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
                                                               uint32_t* __restrict__ n_redone, uint32_t* __restrict__ redo) {
    __shared__ uint64_t keys_all[FB_WAVES][64 * EW];
    __shared__ uint32_t cnt_all[FB_WAVES][FB_DIGITS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
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
        if (mode == 1 && lo_bit < hi_bit) {  // (ballots only: everything is the fallback launch's, k_bucket_sort<.., true>)
            if (lane == 0) redo[atomicAdd(n_redone, 1u)] = b;
            continue;
        }
        uint64_t key[EW];
        {   // ranks from LDS atomics, the order checked; a bucket that fails goes on the fallback launch's list
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
                uint32_t rank2[(EW + 1) / 2];  // two 16-bit ranks per register (a bucket holds <= 1024 composites): at one rank per register the kernel
#pragma unroll                                 // spills, and a kernel that uses scratch memory starts ~0.13 ms late behind kernels that use none
                for (int j = 0; j < (EW + 1) / 2; j++) rank2[j] = 0;
#pragma unroll
                for (uint32_t r = 0; r < (uint32_t)EW; r++) {
                    uint32_t rk = 0;
                    if (r >= E) continue;  // (uniform)
                    const bool valid = r * 64u + lane < n;
                    const uint32_t d = (uint32_t)(key[r] >> bit) & mask;
                    if (bit > lo_bit) rk = rank_runs(cnt, d, valid, lane);
                    else if (valid) rk = atomicAdd(&cnt[d], 1u);
                    rank2[r >> 1] |= rk << ((r & 1u) * 16u);
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
                    if (r * 64u + lane < n) keys[cnt[(uint32_t)(key[r] >> bit) & mask] + ((rank2[r >> 1] >> ((r & 1u) * 16u)) & 0xFFFFu)] = key[r];
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
            if (lo_bit < hi_bit) {  // in order over EVERY bit (see k_bucket_sort)?
                int off = mode == 2;
#pragma unroll
                for (uint32_t r = 0; r < (uint32_t)EW; r++) {
                    const uint32_t idx = r * 64u + lane;
                    if (r < E && idx < n && idx) off |= keys[idx - 1] > key[r];
                }
                const bool sorted = __ballot(off != 0) == 0ull;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (!sorted) {  // (nothing of this bucket has been written: `c` still holds the insertion order)
                    if (lane == 0) redo[atomicAdd(n_redone, 1u)] = b;
                    continue;
                }
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
// ticket != 0: the block's ticket behind the values (what the host polls for: bft_pin_wait_for)
__global__ void k_front_publish(const uint64_t* __restrict__ a, const uint32_t* __restrict__ b, uint64_t* __restrict__ slots, uint64_t* __restrict__ ticket_slot = nullptr,
                                uint64_t ticket = 0) {
    if (a) { slots[0] = *a; slots[1] = *b; }
    else slots[0] = *b;
    if (ticket) {
        __threadfence_system();
        *reinterpret_cast<volatile uint64_t*>(ticket_slot) = ticket;
    }
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
// the T-form, left-aligned (its top 18 bits = the root prefix = the bucket), lo = the sh = 2k - 64 bits below them, id = the genome.  Inside
// a bucket 46 + sh key bits are left (108 at k = 63): the same stable LSD radix sort in LDS as for one-word buckets, over two words -- the
// items in registers (five per item), 9-bit digits first over lo, then over the low 46 bits of hk, ranks from LDS atomics (checked over the
// whole order, ballots on failure), one exchange per pass in two steps so that the buffer holds 12 bytes per item instead of 20.
// (Built and measured first, and dropped: grouping a bucket's items by a 27-bit hash of their key bits -- three passes over 8-byte entries --
// and ordering only the distinct k-mers, a quarter of the items on a pan-genome, by comparison.  The ranking alone is m^2 comparisons of
// 16 bytes: 35 x 10^3 wavefront instructions per bucket of 1850 items / 445 k-mers where these thirteen passes take 13 x 10^3 all told;
// 7.0 ms for config 5's 19 x 10^3 buckets.)
// (rank_runs in the two-word kernels: measured on config 5 -- 2000 genomes, runs of hundreds of equal digits -- 4.79 ms for the bucket sorts
// against 4.08 with one atomic per lane; the hardware serves a wavefront's atomics on ONE address faster than the ballots and the permute cost.)
#ifndef FB2_RUNS
#define FB2_RUNS 0
#endif
struct __attribute__((packed, aligned(4))) BftItem2 {
    uint64_t lo;
    uint32_t id;
};

// digit of the (sh + 46)-bit key hi46 : lo over the bits [b, b + 9): the passes run over the two words as over one number (twelve passes at k = 63,
// where passes that stop at the words' border made thirteen)
__device__ __forceinline__ uint32_t digit2(uint64_t hi, uint64_t lo, uint32_t b, uint32_t sh, uint32_t mask) {
    uint64_t v;
    if (b >= sh) v = hi >> (b - sh);
    else {
        v = lo >> b;
        if (b + FB_DBITS > sh) v |= hi << (sh - b);  // (1 <= sh - b <= 8)
    }
    return (uint32_t)v & mask;
}

template <bool BALLOT, int EMAX>
__device__ __forceinline__ void radix_pass2(uint64_t (&khi)[EMAX], uint64_t (&klo)[EMAX], uint32_t (&kid)[EMAX], uint32_t bit, int nbits, uint32_t sh, uint64_t* buf8, uint32_t* buf4,
                                            uint32_t (*cnt)[FB_DIGITS], uint32_t* wtot, uint32_t n, uint32_t E) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t wbase = wave * E * 64u;
    const uint32_t mask = (1u << nbits) - 1u;
#pragma unroll
    for (uint32_t j = 0; j < FB_DIGITS / 64; j++) cnt[wave][lane * (FB_DIGITS / 64) + j] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t rank[EMAX];
#pragma unroll
    for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
        rank[r] = 0;
        if (r >= E) continue;  // (uniform)
        const uint32_t idx = wbase + r * 64u + lane;
        const bool valid = idx < n;
        const uint32_t d = digit2(khi[r], klo[r], bit, sh, mask);
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
        } else if (FB2_RUNS && bit) {  // (not the first pass: rank_runs)
            rank[r] = rank_runs(cnt[wave], d, valid, lane);
        } else if (valid) {
            rank[r] = atomicAdd(&cnt[wave][d], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __syncthreads();
    {   // digit(s) of this thread: counts of the four wavefronts -> start of (digit, wavefront) in the bucket
        constexpr int DPT = FB_DIGITS / FB_BLOCK;
        uint32_t cw[DPT][FB_WAVES], tot = 0;
#pragma unroll
        for (int j = 0; j < DPT; j++) {
#pragma unroll
            for (int w = 0; w < FB_WAVES; w++) { cw[j][w] = cnt[w][tid * DPT + j]; tot += cw[j][w]; }
        }
        uint32_t inc = tot;
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
    // the exchange: every item to its place; hk and id first, lo behind them through the same buffer
    uint32_t dst[EMAX];
#pragma unroll
    for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
        const uint32_t idx = wbase + r * 64u + lane;
        dst[r] = 0xFFFFFFFFu;
        if (r < E && idx < n) {
            dst[r] = cnt[wave][digit2(khi[r], klo[r], bit, sh, mask)] + rank[r];
            buf8[dst[r]] = khi[r];
            buf4[dst[r]] = kid[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
        const uint32_t idx = wbase + r * 64u + lane;
        if (r < E && idx < n) { khi[r] = buf8[idx]; kid[r] = buf4[idx]; }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < (uint32_t)EMAX; r++)
        if (dst[r] != 0xFFFFFFFFu) buf8[dst[r]] = klo[r];
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
        const uint32_t idx = wbase + r * 64u + lane;
        if (r < E && idx < n) klo[r] = buf8[idx];
    }
    __syncthreads();
}

template <bool BALLOT, int EMAX>
__device__ __forceinline__ void radix_passes2(uint64_t (&khi)[EMAX], uint64_t (&klo)[EMAX], uint32_t (&kid)[EMAX], uint32_t sh, uint64_t* buf8, uint32_t* buf4, uint32_t (*cnt)[FB_DIGITS],
                                              uint32_t* wtot, uint32_t n, uint32_t E) {
    for (uint32_t bit = 0; bit < sh + 46u; bit += FB_DBITS) radix_pass2<BALLOT, EMAX>(khi, klo, kid, bit, (int)min((uint32_t)FB_DBITS, sh + 46u - bit), sh, buf8, buf4, cnt, wtot, n, E);
}

// the non-empty buckets by size class (lists[c * nb ..], n_lists[c]): up to 64 items (class 5: k_bucket2_tiny), up to 512 / 1024 / 2048 -- a wavefront each, 8 / 16 / 32 items per lane --,
// up to 4096 / 8192 -- a workgroup each --, so that the sort kernels walk real buckets only (those of a pan-genome are a few per cent of the 2^18 root
// prefixes); the empty ones get their zero counts here
#define FB2_CLASSES 6
__global__ void k_bucket2_lists(const uint32_t* __restrict__ boff, uint32_t nb, uint32_t* __restrict__ lists, uint32_t* __restrict__ n_lists, uint64_t* __restrict__ counts) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    const uint32_t n = b < nb ? boff[b + 1] - boff[b] : 0u;
    if (b < nb && n == 0) counts[b] = 0;
    const int cls = n == 0 ? -1 : n <= 64u ? 5 : n <= 512u ? 0 : n <= 1024u ? 1 : n <= 2048u ? 2 : n <= 4096u ? 3 : 4;
    __shared__ uint32_t s_cnt[FB2_CLASSES], s_base[FB2_CLASSES];
    if (threadIdx.x < FB2_CLASSES) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (int c = 0; c < FB2_CLASSES; c++) {  // (a slot in the workgroup's part of the class's list: one atomic per wavefront in LDS, one per workgroup on the list's counter)
        const uint64_t mk = __ballot(cls == c);
        if (!mk) continue;
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(mk)) base = atomicAdd(&s_cnt[c], (uint32_t)__builtin_popcountll(mk));
        base = __shfl(base, __builtin_ctzll(mk));
        if (cls == c) mine = base + (uint32_t)__builtin_popcountll(mk & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x < FB2_CLASSES) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&n_lists[threadIdx.x], s_cnt[threadIdx.x]) : 0u;
    __syncthreads();
    if (cls >= 0) lists[(size_t)cls * nb + s_base[cls] + mine] = b;
}

// a bucket of up to 64 items, one per lane of ONE wavefront: every item's rank by comparison with the others (a uniform loop over the items,
// each broadcast from its lane), ties in lane order (= insertion order: ascending ids).  On a pan-genome most non-empty buckets are of this kind --
// the few copies of a locus whose root prefix carries a SNP: 2.3 x 10^5 of config 5's 2.5 x 10^5 buckets, a dozen items each; thirteen radix passes
// apiece took 1.8 ms.
__global__ __launch_bounds__(FB_BLOCK) void k_bucket2_tiny(uint64_t* __restrict__ hk, BftItem2* __restrict__ it, const uint32_t* __restrict__ boff, const uint32_t* __restrict__ list,
                                                           const uint32_t* __restrict__ n_list, uint64_t* __restrict__ counts) {
    __shared__ uint64_t s_h[FB_WAVES][64], s_l[FB_WAVES][64];
    __shared__ uint32_t s_i[FB_WAVES][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t m46 = (1ull << 46) - 1ull;
    const uint32_t n_b = *n_list;
    for (uint32_t li = blockIdx.x * FB_WAVES + wave; li < n_b; li += gridDim.x * FB_WAVES) {
        const uint32_t b = list[li], a0 = boff[b], n = boff[b + 1] - a0;
        uint64_t h = ~0ull, l = ~0ull;
        uint32_t id = 0;
        if (lane < n) {
            const BftItem2 x = it[a0 + lane];
            h = hk[a0 + lane] & m46;
            l = x.lo;
            id = x.id;
        }
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j++) {
            const uint64_t bh = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h >> 32), (int)j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h, (int)j);
            const uint64_t bl = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(l >> 32), (int)j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)l, (int)j);
            const bool less = bh < h || (bh == h && bl < l), eq = bh == h && bl == l;
            rank += (less || (eq && j < lane)) ? 1u : 0u;
        }
        if (lane < n) { s_h[wave][rank] = h; s_l[wave][rank] = l; s_i[wave][rank] = id; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t nk = 0, np = 0;
        if (lane < n) {
            h = s_h[wave][lane]; l = s_l[wave][lane]; id = s_i[wave][lane];
            const bool head = lane == 0 || s_h[wave][lane - 1] != h || s_l[wave][lane - 1] != l;
            nk = head;
            np = head || s_i[wave][lane - 1] != id;
            hk[a0 + lane] = ((uint64_t)b << 46) | h;
            BftItem2 x;
            x.lo = l;
            x.id = id;
            it[a0 + lane] = x;
        }
        for (int o = 32; o > 0; o >>= 1) { nk += __shfl_down(nk, o); np += __shfl_down(np, o); }
        if (lane == 0) counts[b] = ((uint64_t)nk << 32) | np;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next bucket overwrites the slots)
    }
}

// ONE WAVEFRONT per bucket of up to 64 EW items: no barrier anywhere (the LDS operations of a wavefront are in order), the items in registers
// (five per item), per pass: ranks from one LDS atomic per item, a 512-digit scan by shuffles, the exchange word by word through 8 bytes per
// item of LDS.  (With a workgroup per bucket the thirteen passes of k = 63 cost seven barriers each: 12.4 ms for config 5's 19 x 10^3 buckets.)
// PACK: genome ids below 2^18 ride in the 18 free bits on top of hk's 46: four registers and two exchanged words per item instead of five and three
template <int EW, bool PACK>
__global__ __launch_bounds__(FB_BLOCK, (EW <= 8 ? 3 : (EW >= 32 && !PACK) ? 1 : 2)) void k_bucket2_sort_wave(uint64_t* __restrict__ hk, BftItem2* __restrict__ it, const uint32_t* __restrict__ boff, const uint32_t* __restrict__ list,
                                                                const uint32_t* __restrict__ n_list, uint32_t sh, uint64_t* __restrict__ counts, int mode, uint32_t* __restrict__ n_redone, uint32_t* __restrict__ redo) {
    __shared__ uint64_t buf_all[FB_WAVES][64 * EW];
    __shared__ uint32_t cnt_all[FB_WAVES][FB_DIGITS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t m46 = (1ull << 46) - 1ull;
    uint64_t* buf = buf_all[wave];
    uint32_t* cnt = cnt_all[wave];
    const uint32_t n_b = *n_list;
    for (uint32_t li = blockIdx.x * FB_WAVES + wave; li < n_b; li += gridDim.x * FB_WAVES) {
        const uint32_t b = list[li], a0 = boff[b], n = boff[b + 1] - a0;
        const uint32_t E = (n + 63u) / 64u;  // rounds: slot (round r, lane l) = item r 64 + l of the bucket
        uint64_t khi[EW], klo[EW];
        uint32_t kid[PACK ? 1 : EW];
        if (mode == 1) {  // (ballots only: everything is the fallback launch's, k_bucket2_sort<32, true>)
            if (lane == 0) redo[atomicAdd(n_redone, 1u)] = b;
            continue;
        }
        {   // ranks from LDS atomics, the order checked; a bucket that fails goes on the fallback launch's list
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EW; r++) {
                const uint32_t idx = r * 64u + lane;
                khi[r] = ~0ull; klo[r] = ~0ull;
                if (!PACK) kid[r] = 0;
                if (r < E && idx < n) {
                    const BftItem2 x = it[a0 + idx];
                    khi[r] = hk[a0 + idx] & m46;
                    klo[r] = x.lo;
                    if (PACK) khi[r] |= (uint64_t)x.id << 46;
                    else kid[r] = x.id;
                }
            }
            for (uint32_t p = 0, bit = 0; bit < sh + 46u; p++, bit += FB_DBITS) {  // (the two words as one number: digit2)
                const uint32_t width = sh + 46u;
                const int nbits = (int)min((uint32_t)FB_DBITS, width - bit);
                const uint32_t mask = (1u << nbits) - 1u;
#pragma unroll
                for (uint32_t j = 0; j < FB_DIGITS / 64; j++) cnt[lane * (FB_DIGITS / 64) + j] = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                uint32_t rank2[(EW + 1) / 2];  // two 16-bit ranks / slots per register (a bucket holds <= 2048 items)
#pragma unroll
                for (int j = 0; j < (EW + 1) / 2; j++) rank2[j] = 0;
#pragma unroll
                for (uint32_t r = 0; r < (uint32_t)EW; r++) {
                    if (r >= E) continue;  // (uniform)
                    const bool valid = r * 64u + lane < n;
                    const uint32_t d = digit2(khi[r], klo[r], bit, sh, mask);
                    uint32_t rk = 0;
                    if (FB2_RUNS && p) rk = rank_runs(cnt, d, valid, lane);
                    else if (valid) rk = atomicAdd(&cnt[d], 1u);
                    rank2[r >> 1] |= rk << ((r & 1u) * 16u);
                    if (EW > 16 && (r & 7u) == 7u) __builtin_amdgcn_sched_barrier(0);  // (bounds what the scheduler keeps in flight: registers)
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
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
                for (uint32_t r = 0; r < (uint32_t)EW; r++)  // (rank becomes the item's new slot: rank + start < 2048, no carry into the other half)
                {
                    if (r < E && r * 64u + lane < n) rank2[r >> 1] += cnt[digit2(khi[r], klo[r], bit, sh, mask)] << ((r & 1u) * 16u);
                    if (EW > 16 && (r & 7u) == 7u) __builtin_amdgcn_sched_barrier(0);
                }
                // the exchange, word by word through the wavefront's 8 bytes per item
#pragma unroll
                for (int f = 0; f < (PACK ? 2 : 3); f++) {
#pragma unroll
                    for (uint32_t r = 0; r < (uint32_t)EW; r++)
                    {
                        if (r < E && r * 64u + lane < n) buf[(rank2[r >> 1] >> ((r & 1u) * 16u)) & 0xFFFFu] = f == 0 ? khi[r] : f == 1 ? klo[r] : (uint64_t)kid[PACK ? 0 : r];
                        if (EW > 16 && (r & 7u) == 7u) __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                    for (uint32_t r = 0; r < (uint32_t)EW; r++) {
                        const uint32_t idx = r * 64u + lane;
                        if (r < E && idx < n) {
                            const uint64_t v = buf[idx];
                            if (f == 0) khi[r] = v; else if (f == 1) klo[r] = v; else kid[PACK ? 0 : r] = (uint32_t)v;
                        }
                        if (EW > 16 && (r & 7u) == 7u) __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            }
            {   // in order over the key AND the ids (a stable sort keeps equal k-mers in insertion order: ascending ids)?
                int off = mode == 2;
#pragma unroll
                for (uint32_t r = 0; r < (uint32_t)EW; r++) {
                    // the left neighbour: lane - 1 of the round, or lane 63 of the round before
                    const uint32_t idx = r * 64u + lane;
                    uint64_t ph = ((uint64_t)(uint32_t)__shfl((uint32_t)(khi[r] >> 32), (int)((lane + 63u) & 63u)) << 32) | (uint32_t)__shfl((uint32_t)khi[r], (int)((lane + 63u) & 63u));
                    uint64_t pl = ((uint64_t)(uint32_t)__shfl((uint32_t)(klo[r] >> 32), (int)((lane + 63u) & 63u)) << 32) | (uint32_t)__shfl((uint32_t)klo[r], (int)((lane + 63u) & 63u));
                    uint32_t pi = PACK ? 0u : __shfl(kid[PACK ? 0 : r], (int)((lane + 63u) & 63u));
                    const uint32_t rp = r ? r - 1 : 0;
                    const uint64_t qh = ((uint64_t)(uint32_t)__shfl((uint32_t)(khi[rp] >> 32), 63) << 32) | (uint32_t)__shfl((uint32_t)khi[rp], 63);
                    const uint64_t ql = ((uint64_t)(uint32_t)__shfl((uint32_t)(klo[rp] >> 32), 63) << 32) | (uint32_t)__shfl((uint32_t)klo[rp], 63);
                    const uint32_t qi = PACK ? 0u : __shfl(kid[PACK ? 0 : rp], 63);
                    if (lane == 0) { ph = qh; pl = ql; pi = qi; }
                    if (r < E && idx < n && idx) {
                        // (PACK: the id sits above the 46 key bits of hk: the order is (hk46, lo, id))
                        const uint64_t a46 = khi[r] & m46, p46 = ph & m46;
                        const uint32_t ai = PACK ? (uint32_t)(khi[r] >> 46) : kid[PACK ? 0 : r], bi = PACK ? (uint32_t)(ph >> 46) : pi;
                        off |= p46 > a46 || (p46 == a46 && (pl > klo[r] || (pl == klo[r] && bi > ai)));
                    }
                    if (EW > 16 && (r & 3u) == 3u) __builtin_amdgcn_sched_barrier(0);
                }
                if (__ballot(off != 0) != 0ull) {  // (nothing of the bucket has been written: the fallback launch starts from the insertion order)
                    if (lane == 0) redo[atomicAdd(n_redone, 1u)] = b;
                    continue;
                }
            }
        }
        uint32_t nk = 0, np = 0;
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)EW; r++) {
            const uint32_t idx = r * 64u + lane;
            uint64_t ph = ((uint64_t)(uint32_t)__shfl((uint32_t)(khi[r] >> 32), (int)((lane + 63u) & 63u)) << 32) | (uint32_t)__shfl((uint32_t)khi[r], (int)((lane + 63u) & 63u));
            uint64_t pl = ((uint64_t)(uint32_t)__shfl((uint32_t)(klo[r] >> 32), (int)((lane + 63u) & 63u)) << 32) | (uint32_t)__shfl((uint32_t)klo[r], (int)((lane + 63u) & 63u));
            uint32_t pi = PACK ? 0u : __shfl(kid[PACK ? 0 : r], (int)((lane + 63u) & 63u));
            const uint32_t rp = r ? r - 1 : 0;
            const uint64_t qh = ((uint64_t)(uint32_t)__shfl((uint32_t)(khi[rp] >> 32), 63) << 32) | (uint32_t)__shfl((uint32_t)khi[rp], 63);
            const uint64_t ql = ((uint64_t)(uint32_t)__shfl((uint32_t)(klo[rp] >> 32), 63) << 32) | (uint32_t)__shfl((uint32_t)klo[rp], 63);
            const uint32_t qi = PACK ? 0u : __shfl(kid[PACK ? 0 : rp], 63);
            if (lane == 0) { ph = qh; pl = ql; pi = qi; }
            if (r < E && idx < n) {
                const uint64_t a46 = khi[r] & m46;
                const uint32_t ai = PACK ? (uint32_t)(khi[r] >> 46) : kid[PACK ? 0 : r], bi = PACK ? (uint32_t)(ph >> 46) : pi;
                const bool head = idx == 0 || (ph & m46) != a46 || pl != klo[r];
                nk += head;
                np += head || bi != ai;
                hk[a0 + idx] = ((uint64_t)b << 46) | a46;
                BftItem2 x;
                x.lo = klo[r];
                x.id = ai;
                it[a0 + idx] = x;
            }
            if (EW > 16 && (r & 3u) == 3u) __builtin_amdgcn_sched_barrier(0);
        }
        for (int o = 32; o > 0; o >>= 1) { nk += __shfl_down(nk, o); np += __shfl_down(np, o); }
        if (lane == 0) counts[b] = ((uint64_t)nk << 32) | np;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next bucket overwrites buf)
    }
}

// mode as k_bucket_sort's: 0 = ranks from LDS atomics, the order checked, ballots on failure; 1 = ballots only; 2 = the check always fails
// REDO: the fallback launch over the `redo` list (ballot ranks), as k_bucket_sort's.
template <int EMAX, bool REDO>
__global__ __launch_bounds__(FB_BLOCK) void k_bucket2_sort(uint64_t* __restrict__ hk, BftItem2* __restrict__ it, const uint32_t* __restrict__ boff, const uint32_t* __restrict__ list,
                                                           const uint32_t* __restrict__ n_list, uint32_t sh, uint64_t* __restrict__ counts, int mode, uint32_t* __restrict__ n_redone,
                                                           uint32_t* __restrict__ redo, uint32_t* __restrict__ fail) {
    constexpr uint32_t CAP = FB_BLOCK * EMAX;
    __shared__ uint64_t buf8[CAP];
    __shared__ uint32_t buf4[CAP];
    __shared__ uint32_t cnt[FB_WAVES][FB_DIGITS];
    __shared__ uint32_t wtot[FB_WAVES];
    __shared__ uint32_t s_nk, s_np;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t m46 = (1ull << 46) - 1ull;
    const uint32_t n_b = *n_list;
    for (uint32_t li = blockIdx.x; li < n_b; li += gridDim.x) {
        const uint32_t b = list[li], a0 = boff[b], n = boff[b + 1] - a0;
        if (n > CAP) {  // (excluded by the caller; counted in case)
            if (tid == 0) { counts[b] = 0; atomicAdd(fail, 1u); }
            continue;
        }
        const uint32_t E = (n + FB_BLOCK - 1) / FB_BLOCK;  // rounds
        const uint32_t wbase = wave * E * 64u;
        uint64_t khi[EMAX], klo[EMAX];
        uint32_t kid[EMAX];
        auto load = [&]() {
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                khi[r] = ~0ull; klo[r] = ~0ull; kid[r] = 0;
                if (r < E && idx < n) {
                    const BftItem2 x = it[a0 + idx];
                    khi[r] = hk[a0 + idx] & m46;
                    klo[r] = x.lo;
                    kid[r] = x.id;
                }
            }
        };
        // in order over the key AND the ids (equal k-mers must have kept their insertion order -- ascending ids --, which only a stable sort does)?
        auto out_of_order = [&]() -> int {
            int off = 0;
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                if (r < E && idx < n) { buf8[idx] = khi[r]; buf4[idx] = kid[r]; }
            }
            __syncthreads();
            uint64_t ph[EMAX];
            uint32_t pi[EMAX];
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                ph[r] = 0; pi[r] = 0;
                if (r < E && idx < n && idx) { ph[r] = buf8[idx - 1]; pi[r] = buf4[idx - 1]; }
            }
            __syncthreads();
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                if (r < E && idx < n) buf8[idx] = klo[r];
            }
            __syncthreads();
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
                const uint32_t idx = wbase + r * 64u + lane;
                if (r < E && idx < n && idx) {
                    const uint64_t pl = buf8[idx - 1];
                    off |= ph[r] > khi[r] || (ph[r] == khi[r] && (pl > klo[r] || (pl == klo[r] && pi[r] > kid[r])));
                }
            }
            return __syncthreads_or(off);
        };
        if (!REDO && mode == 1) {  // (ballots only: everything is the fallback launch's)
            if (tid == 0) redo[atomicAdd(n_redone, 1u)] = b;
            continue;
        }
        load();
        if (!REDO) {
            radix_passes2<false, EMAX>(khi, klo, kid, sh, buf8, buf4, cnt, wtot, n, E);
            if (out_of_order() | (mode == 2)) {  // (nothing of the bucket has been written: the fallback launch starts from the insertion order)
                if (tid == 0) redo[atomicAdd(n_redone, 1u)] = b;
                continue;
            }
        } else
            radix_passes2<true, EMAX>(khi, klo, kid, sh, buf8, buf4, cnt, wtot, n, E);
        // duplicates against the left neighbour; the bucket back in place, sorted
        if (tid == 0) { s_nk = 0; s_np = 0; }
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
            const uint32_t idx = wbase + r * 64u + lane;
            if (r < E && idx < n) { buf8[idx] = khi[r]; buf4[idx] = kid[r]; }
        }
        __syncthreads();
        uint32_t headm = 0, samem = 0;  // bit r: hk differs from the left neighbour's / (hk, id) equal to it
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
            const uint32_t idx = wbase + r * 64u + lane;
            if (r < E && idx < n) {
                const bool first = idx == 0;
                const uint64_t ph = first ? 0ull : buf8[idx - 1];
                const uint32_t pi = first ? 0u : buf4[idx - 1];
                if (first || ph != khi[r]) headm |= 1u << r;
                if (!first && ph == khi[r] && pi == kid[r]) samem |= 1u << r;
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
            const uint32_t idx = wbase + r * 64u + lane;
            if (r < E && idx < n) buf8[idx] = klo[r];
        }
        __syncthreads();
        uint32_t nk = 0, np = 0;
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)EMAX; r++) {
            const uint32_t idx = wbase + r * 64u + lane;
            if (r < E && idx < n) {
                const bool lo_diff = idx == 0 || buf8[idx - 1] != klo[r];
                const bool head = ((headm >> r) & 1u) || lo_diff;
                const bool dup = ((samem >> r) & 1u) && !lo_diff;
                nk += head;
                np += !dup;
                hk[a0 + idx] = ((uint64_t)b << 46) | khi[r];
                BftItem2 x;
                x.lo = klo[r];
                x.id = kid[r];
                it[a0 + idx] = x;
            }
        }
        for (int o = 32; o > 0; o >>= 1) { nk += __shfl_down(nk, o); np += __shfl_down(np, o); }
        if (lane == 0) { atomicAdd(&s_nk, nk); atomicAdd(&s_np, np); }
        __syncthreads();
        if (tid == 0) counts[b] = ((uint64_t)s_nk << 32) | s_np;
        __syncthreads();
    }
}

// bases[b] = exclusive scan of counts (k-mers << 32 | pairs).  Every bucket places its k-mers (two words: the T-form back from hk and lo), the offset
// of each k-mer's first genome id, and the genome ids of its distinct pairs (the bucket lies sorted by (k-mer, id): duplicates are neighbours).
__global__ __launch_bounds__(FB_BLOCK) void k_bucket2_emit(const uint64_t* __restrict__ hk, const BftItem2* __restrict__ it, const uint32_t* __restrict__ boff, uint32_t nb,
                                                           const uint64_t* __restrict__ bases, uint64_t* __restrict__ tk, uint32_t* __restrict__ seg_off, uint32_t* __restrict__ pg, uint32_t sh,
                                                           uint32_t nk, uint32_t np) {  // sh = 2k - 64: bits of lo
    __shared__ uint32_t w_nk[FB_WAVES], w_np[FB_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    if (blockIdx.x == 0 && tid == 0) seg_off[nk] = np;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t a0 = boff[b], n = boff[b + 1] - a0;
        if (n == 0) continue;
        const uint64_t base = bases[b];
        uint32_t kbase = (uint32_t)(base >> 32), pbase = (uint32_t)base;
        for (uint32_t r0 = 0; r0 < n; r0 += FB_BLOCK) {
            const uint32_t idx = r0 + tid;
            const bool valid = idx < n;
            uint64_t a = 0, l = 0, pa = 0, pl = 0;
            uint32_t id = 0, pid = 0;
            if (valid) {
                a = hk[a0 + idx];
                const BftItem2 x = it[a0 + idx];
                l = x.lo;
                id = x.id;
                if (idx) { pa = hk[a0 + idx - 1]; const BftItem2 y = it[a0 + idx - 1]; pl = y.lo; pid = y.id; }
            }
            const bool head = valid && (idx == 0 || a != pa || l != pl), keep = valid && (head || id != pid);
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
            if (keep) pg[prank] = id;
            if (head) {
                const uint32_t q = kb + (uint32_t)__builtin_popcountll(hm & lt_mask);
                // T-form: hk holds its top 64 bits left-aligned, lo the sh bits below them
                tk[2ull * q] = sh == 64 ? a : a >> (64 - sh);
                tk[2ull * q + 1] = sh == 64 ? l : ((a << sh) | l);
                seg_off[q] = prank;
            }
            kbase += tk_all;
            pbase += tp_all;
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
    DevBuf redone, redo;  // buckets to sort again with ballot ranks: their number, their list
    CK(redone.alloc_zero(4, s));
    CK(redo.alloc((size_t)nb * 4));
    // The size of the largest bucket decides which workgroup variant the larger buckets need, and whether the buckets fit at all; it
    // travels to the host behind the split while the wavefront kernel -- whose own variant follows the MEAN bucket -- is already
    // running (a synchronisation in front of it left the GPU idle for ~0.15 ms).
    const uint64_t ticket = bft_pin_next_ticket();
    hipLaunchKernelGGL(k_front_publish, dim3(1), dim3(1), 0, s, (const uint64_t*)nullptr, d_max_bucket, pin.p, pin.p + PIN_SLOTS, ticket);
    // the small buckets a wavefront each, the others a workgroup each (one after the other: on two streams the two kernels --
    // both latency-bound, different LDS footprints -- got in each other's way: 8.6 ms instead of 2.0 + 0.8)
    const uint32_t wave_cap = n / std::max(nb, 1u) <= 192u ? 512u : 1024u;
    const dim3 wgrid(std::min<uint32_t>((nb + FB_WAVES - 1) / FB_WAVES, 256u * 16u));
    if (wave_cap == 512u) hipLaunchKernelGGL(k_bucket_sort_wave<8>, wgrid, block, 0, s, d_c, d_boff, nb, gb, split_bit, counts.as<uint64_t>(), d_vals, vw, g_rank_mode, redone.as<uint32_t>(), redo.as<uint32_t>());
    else hipLaunchKernelGGL(k_bucket_sort_wave<16>, wgrid, block, 0, s, d_c, d_boff, nb, gb, split_bit, counts.as<uint64_t>(), d_vals, vw, g_rank_mode, redone.as<uint32_t>(), redo.as<uint32_t>());
    CK(bft_pin_wait_for(pin, s, ticket));
    const uint32_t mx = (uint32_t)pin.p[0];
    *max_bucket = mx;
    bft_trace_mark("root-prefix split done (largest bucket known)");
    if (mx > FB_CAP) return 0;
#define FB_LAUNCH(E) hipLaunchKernelGGL((k_bucket_sort<E, false>), grid, block, 0, s, d_c, d_boff, nb, gb, split_bit, counts.as<uint64_t>(), d_vals, vw, g_rank_mode, redone.as<uint32_t>(), redo.as<uint32_t>(), wave_cap)
    if (mx <= wave_cap) {}
    else if (mx <= 256u * 6u) FB_LAUNCH(6);
    else if (mx <= 256u * 9u) FB_LAUNCH(9);
    else if (mx <= 256u * 12u) FB_LAUNCH(12);
    else FB_LAUNCH(16);
#undef FB_LAUNCH
    // the fallback launch: the buckets whose order check failed (none, on every run so far), or all of them ("test_front_rank_mode" 1)
    hipLaunchKernelGGL((k_bucket_sort<FB_EMAX, true>), dim3(g_rank_mode ? grid.x : 64u), block, 0, s, d_c, d_boff, nb, gb, split_bit, counts.as<uint64_t>(), d_vals, vw, g_rank_mode,
                       redone.as<uint32_t>(), redo.as<uint32_t>(), 0u);
    CK(bft_scan::exclusive_sum_ptr<uint64_t>(counts.as<uint64_t>(), bases.as<uint64_t>(), (uint64_t)nb + 1, s, tmp));
    const uint64_t ticket2 = bft_pin_next_ticket();
    hipLaunchKernelGGL(k_front_publish, dim3(1), dim3(1), 0, s, bases.as<uint64_t>() + nb, redone.as<uint32_t>(), pin.p + 1, pin.p + PIN_SLOTS, ticket2);
    HIPCK(hipGetLastError());
    bft_stage("bucket sorts in LDS (+ scan of the counts)", (double)n * 16 + (d_vals ? (double)n * vw : 0.0), s);
    CK(bft_pin_wait_for(pin, s, ticket2));
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
                       const uint32_t* d_max_bucket, uint32_t* max_bucket, bool* done, uint32_t* n_redone, uint32_t max_gid) {
    *done = false;
    const bool pack = max_gid < (1u << 18);  // (ids that fit the 18 free bits above hk's 46: k_bucket2_sort_wave)
    BftItem2* d_it = (BftItem2*)d_items;
    PinBlock pin;  // [0] the largest bucket, [1] k-mers << 32 | pairs, [2] buckets that failed, [3] buckets sorted again
    if (!pin.p) return bft_fail(BFT_GPU_E_HIP, "hipHostMalloc (front end counts)");
    DevBuf counts, bases, tmp, fail, redone, redo, lists, n_lists;
    CK(counts.alloc(((uint64_t)nb + 1) * 8));
    CK(bases.alloc(((uint64_t)nb + 1) * 8));
    CK(fail.alloc_zero(4, s));
    CK(redone.alloc_zero(4, s));
    CK(redo.alloc((size_t)nb * 4));  // (buckets to sort again with ballot ranks)
    CK(lists.alloc((size_t)FB2_CLASSES * nb * 4));
    CK(n_lists.alloc_zero(32, s));
    HIPCK(hipMemsetAsync((uint8_t*)counts.p + (uint64_t)nb * 8, 0, 8, s));
    const dim3 block(FB_BLOCK);
    const uint64_t ticket = bft_pin_next_ticket();
    hipLaunchKernelGGL(k_front_publish, dim3(1), dim3(1), 0, s, (const uint64_t*)nullptr, d_max_bucket, pin.p, pin.p + PIN_SLOTS, ticket);
    // (the size of the largest bucket travels to the host while the lists are made and the buckets of a wavefront each -- all of them, usually -- are sorted)
    hipLaunchKernelGGL(k_bucket2_lists, dim3((nb + FB_BLOCK - 1) / FB_BLOCK), block, 0, s, d_boff, nb, lists.as<uint32_t>(), n_lists.as<uint32_t>(), counts.as<uint64_t>());
    const uint32_t sh = (uint32_t)(2 * k - 64);
#define FB2_WAVE(EW_, CLS, GRID)                                                                                                                                                \
    do {                                                                                                                                                                       \
        if (pack) hipLaunchKernelGGL((k_bucket2_sort_wave<EW_, true>), dim3(GRID), block, 0, s, d_hk, d_it, d_boff, lists.as<uint32_t>() + (size_t)CLS * nb, n_lists.as<uint32_t>() + CLS, sh, \
                                     counts.as<uint64_t>(), g_rank_mode, redone.as<uint32_t>(), redo.as<uint32_t>());                                                          \
        else hipLaunchKernelGGL((k_bucket2_sort_wave<EW_, false>), dim3(GRID), block, 0, s, d_hk, d_it, d_boff, lists.as<uint32_t>() + (size_t)CLS * nb, n_lists.as<uint32_t>() + CLS, sh,   \
                                counts.as<uint64_t>(), g_rank_mode, redone.as<uint32_t>(), redo.as<uint32_t>());                                                               \
    } while (0)
#define FB2_LAUNCH(EM, CLS, GRID)                                                                                                                                           \
    hipLaunchKernelGGL((k_bucket2_sort<EM, false>), dim3(GRID), block, 0, s, d_hk, d_it, d_boff, lists.as<uint32_t>() + (size_t)CLS * nb, n_lists.as<uint32_t>() + CLS, sh, counts.as<uint64_t>(), \
                       g_rank_mode, redone.as<uint32_t>(), redo.as<uint32_t>(), fail.as<uint32_t>())
    hipLaunchKernelGGL(k_bucket2_tiny, dim3(256u * 8u), block, 0, s, d_hk, d_it, d_boff, lists.as<uint32_t>() + (size_t)5 * nb, n_lists.as<uint32_t>() + 5, counts.as<uint64_t>());
    FB2_WAVE(8, 0, 256u * 8u);
    FB2_WAVE(16, 1, 256u * 4u);
    FB2_WAVE(32, 2, 256u * 2u);
    CK(bft_pin_wait_for(pin, s, ticket));
    const uint32_t mx = (uint32_t)pin.p[0];
    *max_bucket = mx;
    if (mx > FB_BLOCK * 32) return 0;
    if (mx > 2048u) FB2_LAUNCH(16, 3, 256u * 4u);
    if (mx > 4096u) FB2_LAUNCH(32, 4, 256u * 2u);
#undef FB2_WAVE
#undef FB2_LAUNCH
    // the fallback launch: the buckets whose order check failed (none, on every run so far), or all of them ("test_front_rank_mode" 1): the list is `redo`
    hipLaunchKernelGGL((k_bucket2_sort<32, true>), dim3(g_rank_mode ? 256u * 2u : 32u), block, 0, s, d_hk, d_it, d_boff, redo.as<uint32_t>(), redone.as<uint32_t>(), sh, counts.as<uint64_t>(),
                       g_rank_mode, redone.as<uint32_t>(), redo.as<uint32_t>(), fail.as<uint32_t>());
    CK(bft_scan::exclusive_sum_ptr<uint64_t>(counts.as<uint64_t>(), bases.as<uint64_t>(), (uint64_t)nb + 1, s, tmp));
    const uint64_t ticket2 = bft_pin_next_ticket();
    hipLaunchKernelGGL(k_front_publish, dim3(1), dim3(1), 0, s, bases.as<uint64_t>() + nb, fail.as<uint32_t>(), pin.p + 1);
    hipLaunchKernelGGL(k_front_publish, dim3(1), dim3(1), 0, s, (const uint64_t*)nullptr, redone.as<uint32_t>(), pin.p + 3, pin.p + PIN_SLOTS, ticket2);
    HIPCK(hipGetLastError());
    bft_stage("two-word buckets: sorts in LDS (+ scan of the counts)", (double)n * 2 * 20, s);
    CK(bft_pin_wait_for(pin, s, ticket2));
    if (n_redone) *n_redone = (uint32_t)pin.p[3];
    if (pin.p[2] != 0) return 0;  // (a bucket beyond this front end: the caller sorts device-wide)
    const uint64_t total = pin.p[1];
    nk = total >> 32;
    np = total & 0xFFFFFFFFull;
    CK(tk.alloc(nk * 16));
    CK(seg_off.alloc((nk + 1) * 4));
    CK(pg.alloc(np * 4));
    hipLaunchKernelGGL(k_bucket2_emit, dim3(std::min<uint32_t>(nb, 256u * 16u)), block, 0, s, d_hk, d_it, d_boff, nb, bases.as<uint64_t>(), tk.as<uint64_t>(), seg_off.as<uint32_t>(),
                       pg.as<uint32_t>(), sh, (uint32_t)nk, (uint32_t)np);
    HIPCK(hipGetLastError());
    bft_stage("two-word buckets: emit (k-mers, offsets, genome ids)", (double)n * 20 + (double)np * 4 + (double)nk * 20, s);
    *done = true;
    return 0;
}
