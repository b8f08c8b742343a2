// bft_sort.h -- the library's own stable radix sort / radix partition (device code + host launcher; templates, header only).
//
// What it is for: the reference keeps every container sorted by inserting one k-mer at a time (insertKmer_Node src/insertNode.c:38-226,
// transform2CC's sort by rotated prefix src/CC.c:40-367, insertKmer_UC src/UC.c:13-79); the bulk build sorts instead, and until round 6
// every device-wide sort of it was a rocPRIM / hipCUB call (42 % of a config-3 build's device time).  This is the replacement, written for
// the part it runs on:
//
//   * LSD passes of up to 9 bits with ONE read and ONE write of the array per pass ("onesweep"), behind ONE histogram kernel that reads the
//     keys once for all passes;
//   * a tile (THREADS x IPT keys, up to 12288) is ranked in registers, REORDERED IN LDS and written out by digit runs, so that a digit's keys
//     leave as one contiguous piece instead of one transaction per key;
//   * THE FIRST PASS NEEDS NO LOOK-BACK: the input is cut into one contiguous range of tiles per workgroup, the histogram kernel (same
//     ranges) counts per (range, digit), a scan over the ranges tells every workgroup where its keys of every digit start, and the workgroup
//     carries those positions from tile to tile in LDS.  Consecutive tiles of a range write consecutive pieces of every digit's output from
//     the same CU: the partial lines at their seams meet in one L2;
//   * the later passes read an array whose order the histogram kernel cannot know, so their tiles find their place by a decoupled
//     look-back -- but in up to 64 INDEPENDENT CHAINS instead of one: chain c of pass p is the stretch of its input that holds the keys
//     whose PREVIOUS digit has the top bits c (a contiguous stretch: the previous pass sorted on that digit), and the histogram kernel can
//     count per (chain, digit) from the keys alone.  A chain is worked on by the few workgroups of one XCD (HW_REG_XCC_ID picks the chains
//     a workgroup prefers; any workgroup may take any tile, placement is never a matter of correctness), so a look-back is one round trip
//     of four tile states instead of one per resident workgroup, and neighbouring tiles again share an L2;
//   * tile states are 32-bit words {inclusive:1, count + 1:31}, four digits per 16-byte write-through (sc1) store / load (MI355X_MICROARCH.md,
//     inter-workgroup visibility: narrow sc1 stores are a fabric write each); a tile publishes its counts before it looks back;
//   * tiles are claimed from a counter per chain one tile AHEAD (the returning atomic travels while a tile is worked on), and the next
//     tile's keys are loaded into the registers the current tile's keys have just left (they sit in LDS by then);
//   * ranks: a key's rank among the keys of its digit in its wavefront comes from ONE returning LDS atomic per lane on the wavefront's own
//     counters (rounds of 64 keys: the atomics of a wavefront are served in the order they were issued).  Inside one instruction the
//     lanes that share a digit are served in lane order on this hardware -- measured, not documented -- which is what makes the sort
//     stable; bft_rs::rank_mode() checks that behaviour on the device once per process (k_rs_selftest) and every sort falls back to ranks
//     from wavefront ballots (stable by construction, ~0.35 ms more per pass over 2 x 10^8 keys) if it does not hold or "sort_ballots" is set.
//
// n < 2^31 per call (the callers' arrays are rows and pairs counted in 32 bits; the insertion log is flushed at 2^30 pairs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "bft_dev.h"

// (see bft_rs::sort for the two)
#ifndef BFT_RS_CHAIN_MIN
#define BFT_RS_CHAIN_MIN (1u << 23)
#endif
#ifndef BFT_RS_BIG_THREADS
#define BFT_RS_BIG_THREADS 1024
#endif

#ifndef BFT_RS_RFL
#define BFT_RS_RFL 1
#endif
#if BFT_RS_RFL
#define BFT_RS_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define BFT_RS_UNI(x) (x)
#endif
namespace bft_rs {

constexpr int DBITS = 9;
constexpr int DIGITS = 1 << DBITS;
constexpr int MAXP = 8;
constexpr int MAXCB = 6;  // up to 64 chains
// a tile's state per digit: 0 = not published; otherwise count + 1 in the low 31 bits, bit 31 set when the count includes every tile before it
constexpr uint32_t ST_VAL = 0x7FFFFFFFu;
constexpr uint32_t ST_AGG = 1u, ST_INC = 0x80000001u;  // (added to a count)
constexpr uint32_t NONE = 0xFFFFFFFFu;

struct NoVal {};

struct Plan {
    int P;
    uint32_t bit[MAXP], nbits[MAXP];
    uint32_t cb[MAXP];    // chains of pass p = 2^cb[p] (p >= 1): the top cb bits of digit p - 1
    uint32_t hoff[MAXP];  // words: where pass p's counters start in a histogram workgroup's block ([1 or chains][2^nbits])
    uint32_t hwords;
};
static inline Plan make_plan(unsigned begin_bit, unsigned end_bit) {
    Plan pl;
    const unsigned bits = end_bit - begin_bit;
    pl.P = (int)((bits + DBITS - 1) / DBITS);
    unsigned b = begin_bit, maxnb = 0;
    for (int p = 0; p < pl.P; p++) {
        const unsigned left = end_bit - b, nb = (left + (pl.P - p) - 1) / (pl.P - p);
        pl.bit[p] = b;
        pl.nbits[p] = nb;
        maxnb = std::max(maxnb, nb);
        b += nb;
    }
    // chains: as many as the histogram kernel's LDS holds counters for (128 KB), at most 64
    unsigned cb = MAXCB;
    while (cb > 0 && (size_t)(pl.P - 1) * ((size_t)1 << (cb + maxnb)) * 4 > 128 * 1024) cb--;
    uint32_t o = 0;
    for (int p = 0; p < MAXP; p++) {
        if (p >= pl.P) { pl.bit[p] = pl.nbits[p] = pl.cb[p] = pl.hoff[p] = 0; continue; }
        pl.cb[p] = p == 0 ? 0 : std::min<unsigned>(cb, pl.nbits[p - 1]);
        pl.hoff[p] = o;
        o += (1u << pl.cb[p]) << pl.nbits[p];
    }
    pl.hwords = o;
    return pl;
}

// plain arrays as the input of a pass
template <class K, class V>
struct PtrIn {
    const K* k;
    const V* v;
    __device__ __forceinline__ K key(uint32_t i) const { return k[i]; }
    __device__ __forceinline__ V val(uint32_t i) const {
        if constexpr (std::is_same<V, NoVal>::value) return V{};
        else return v[i];
    }
};

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }  // hwreg(HW_REG_XCC_ID, 0, 4)

template <class K>
__device__ __forceinline__ uint32_t digit_of(K key, uint32_t bit, uint32_t mask) { return (uint32_t)(key >> bit) & mask; }

constexpr size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// ---- histogram of every pass's digits in one read of the keys ---------------------------------------------------------------------------
// Workgroup r counts range r of the input (tiles [r tpr, (r + 1) tpr)): pass 0 per digit, the later passes per (chain, digit).
template <class K, class In, int THREADS>
__global__ __launch_bounds__(THREADS) void k_rs_hist(In in, uint32_t n, Plan pl, uint32_t range_len, uint32_t* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) uint32_t h[];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < pl.hwords; i += THREADS) h[i] = 0;
    __syncthreads();
    const uint64_t a64 = (uint64_t)blockIdx.x * range_len;
    const uint32_t a = (uint32_t)(a64 < n ? a64 : n), b = (uint32_t)(a64 + range_len < n ? a64 + range_len : n);
    constexpr uint32_t U = 8;
    for (uint32_t i0 = a; i0 < b; i0 += THREADS * U) {
        K key[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t idx = i0 + u * THREADS + tid;
            key[u] = in.key(idx < b ? idx : a);
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t idx = i0 + u * THREADS + tid;
            if (idx < b) {
                uint32_t prev = 0;
                for (int p = 0; p < pl.P; p++) {
                    const uint32_t d = digit_of(key[u], pl.bit[p], (1u << pl.nbits[p]) - 1u);
                    const uint32_t ch = p ? prev >> (pl.nbits[p - 1] - pl.cb[p]) : 0u;
                    atomicAdd(&h[pl.hoff[p] + (ch << pl.nbits[p]) + d], 1u);
                    prev = d;
                }
            }
        }
    }
    __syncthreads();
    uint32_t* out = partial + (size_t)blockIdx.x * pl.hwords;
    for (uint32_t i = tid; i < pl.hwords; i += THREADS) out[i] = h[i];
}

// the chain counters of the later passes: cnt[w] += sum over a slice of the histogram workgroups (grid.y slices; cnt zeroed)
[[maybe_unused]] static __global__ __launch_bounds__(256) void k_rs_reduce(const uint32_t* __restrict__ partial, uint32_t nwg, uint32_t stride, uint32_t first, uint32_t words, uint32_t* __restrict__ cnt) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= words) return;
    const uint32_t per = (nwg + gridDim.y - 1) / gridDim.y, g0 = blockIdx.y * per, g1 = min(nwg, g0 + per);
    uint32_t s = 0;
#pragma unroll 8
    for (uint32_t g = g0; g < g1; g++) s += partial[(size_t)g * stride + first + w];
    if (s) atomicAdd(&cnt[w], s);
}

// One workgroup per (digit, pass): exclusive scan of the digit's counts over the rows (ranges of pass 0 -- in place in the histogram
// workgroups' blocks --, chains of a later pass), in place; the digit's total.  rows <= 1024.
[[maybe_unused]] static __global__ __launch_bounds__(256) void k_rs_rowscan(uint32_t* __restrict__ partial, uint32_t* __restrict__ cnt, Plan pl, uint32_t ranges, uint32_t* __restrict__ tot_all) {
    __shared__ uint32_t wsum[4];
    const uint32_t d = blockIdx.x, p = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (d >= (1u << pl.nbits[p])) return;
    uint32_t* m = p == 0 ? partial + pl.hoff[0] : cnt + pl.hoff[p];
    const uint32_t stride = p == 0 ? pl.hwords : 1u << pl.nbits[p], rows = p == 0 ? ranges : 1u << pl.cb[p];
    uint32_t* tot = tot_all + p * DIGITS;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t r = tid * 4 + q;
        v[q] = r < rows ? m[(size_t)r * stride + d] : 0u;
        s += v[q];
    }
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(inc, o);
        if ((int)lane >= o) inc += x;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - s;
    for (uint32_t w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t r = tid * 4 + q;
        if (r < rows) m[(size_t)r * stride + d] = run;
        run += v[q];
    }
    if (tid == 255) tot[d] = run;
}

// One workgroup per pass: where every digit starts in the pass's output; the chains and tiles of the NEXT pass.
// chain table of a pass: start[65], tile_first[65]
constexpr int CT = 2 * ((1 << MAXCB) + 1);
[[maybe_unused]] static __global__ __launch_bounds__(DIGITS) void k_rs_digits(const uint32_t* __restrict__ tot, uint32_t* __restrict__ dbase, uint32_t* __restrict__ chain, Plan pl, uint32_t n, uint32_t tile) {
    __shared__ uint32_t wsum[DIGITS / 64];
    __shared__ uint32_t db[DIGITS + 1];
    const uint32_t p = blockIdx.x, d = threadIdx.x, lane = d & 63u, wave = d >> 6, nd = 1u << pl.nbits[p];
    const uint32_t t = d < nd ? tot[p * DIGITS + d] : 0u;
    uint32_t inc = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(inc, o);
        if ((int)lane >= o) inc += x;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - t;
    for (uint32_t w = 0; w < wave; w++) run += wsum[w];
    db[d] = run;
    dbase[p * DIGITS + d] = run;
    __syncthreads();
    if (p + 1 < (uint32_t)pl.P) {  // (uniform)
        __shared__ uint32_t cst[(1 << MAXCB) + 1];
        uint32_t* cs = chain + (size_t)(p + 1) * CT;
        const uint32_t nch = 1u << pl.cb[p + 1], sh = pl.nbits[p] - pl.cb[p + 1];
        if (d <= nch) {
            cst[d] = d < nch ? db[d << sh] : n;
            cs[d] = cst[d];
        }
        __syncthreads();
        if (d < 64) {  // (wave 0: tiles per chain -> first tile of every chain)
            const uint32_t nt = d < nch ? (cst[d + 1] - cst[d] + tile - 1) / tile : 0u;
            uint32_t ti = nt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t x = __shfl_up(ti, o);
                if ((int)d >= o) ti += x;
            }
            if (d < nch) cs[(1 << MAXCB) + 1 + d] = ti - nt;
            if (d == nch - 1) cs[(1 << MAXCB) + 1 + nch] = ti;
        }
    }
}

// ---- do the lanes of one LDS atomic instruction that hit one address get their turns in lane order? ------------------------------------
[[maybe_unused]] static __global__ __launch_bounds__(64) void k_rs_selftest(uint32_t* __restrict__ bad) {
    __shared__ uint32_t c[64];
    const uint32_t lane = threadIdx.x;
    uint32_t wrong = 0;
    c[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t x = 0x9E3779B9u * (blockIdx.x + 1);
    for (int it = 0; it < 512; it++) {
        // every lane picks one of `groups` addresses (1 .. 64 groups: from all lanes on one address to a permutation) from a sequence every lane computes
        x = x * 1664525u + 1013904223u;
        const uint32_t groups = 1u + ((x >> 9) % 64u);
        uint32_t y = x ^ (lane * 0x85EBCA6Bu);
        y ^= y >> 15; y *= 0x2C1B3C6Du; y ^= y >> 12;
        const uint32_t a = y % groups;
        const bool on = ((x >> (lane & 31u)) & 1u) || (it & 1);  // some lanes sit out
        uint32_t r = 0;
        if (on) r = atomicAdd(&c[a], 1u);
        // what the ranks must be: my position among the active lanes below me with my address, plus the counter's value before
        uint64_t peers = __ballot(on);
        for (int b = 0; b < 6; b++) {
            const bool bit = (a >> b) & 1u;
            const uint64_t bj = __ballot(bit);
            peers &= bit ? bj : ~bj;
        }
        const int leader = on ? __builtin_ctzll(peers) : (int)lane;
        const uint32_t r0 = __shfl(r, leader);
        if (on && r != r0 + (uint32_t)__builtin_popcountll(peers & ((1ull << lane) - 1ull))) wrong++;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (wrong) atomicAdd(bad, wrong);
}

// ---- one pass --------------------------------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifdef BFT_RS_PROF  // (microbenchmark only: cycles per phase of a tile, summed over the tiles of every workgroup's thread 0)
__device__ unsigned long long g_rs_prof[2][16];
#define RS_T(i) do { if (tid == 0) { const unsigned long long t_ = clock64(); atomicAdd(&g_rs_prof[RANGED ? 0 : 1][i], t_ - t_prev); t_prev = t_; } } while (0)
#else
#define RS_T(i) do {} while (0)
#endif
#ifndef BFT_RS_KO
#define BFT_RS_KO 0  // (microbenchmark only: knock-outs -- 1 no look-back, 2 no stores of the reordered tile; wrong results)
#endif

// RANGED: the first pass (no look-back: workgroup r owns the tiles [r tpr, (r + 1) tpr) and knows from `rows` where its keys of every digit
// start).  !RANGED: tiles claimed from the chains' counters, positions by look-back.  BALLOT: ranks from wavefront ballots.
template <class K, class V, class In, int THREADS, int IPT, bool RANGED, bool BALLOT, int LBG>
__global__ __launch_bounds__(THREADS) void k_rs_pass(In in, K* __restrict__ ok, V* __restrict__ ov, uint32_t n, uint32_t bit, uint32_t nbits, const uint32_t* __restrict__ rows,
                                                     uint32_t row_stride, const uint32_t* __restrict__ dbase, uint32_t tpr, const uint32_t* __restrict__ chain_start,
                                                     const uint32_t* __restrict__ tile_first, uint32_t nch, uint32_t* __restrict__ heads, uint32_t* __restrict__ states,
                                                     uint32_t states_bytes) {
    constexpr bool HASV = !std::is_same<V, NoVal>::value;
    constexpr int WAVES = THREADS / 64, TILE = THREADS * IPT;
    constexpr int DT = THREADS < DIGITS ? THREADS : DIGITS, DPT = DIGITS / DT;  // the threads that own DPT digits each (scan of the tile's counts)
    constexpr int LBT = DIGITS / 4, LB = 4;                                     // look-back threads (four digits each), tiles fetched per round
    static_assert(THREADS >= LBT && THREADS % 64 == 0, "workgroup too small");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    K* lk = reinterpret_cast<K*>(smem);  // [TILE + 1] (the last slot takes the writes of the lanes beyond a partial tile)
    constexpr size_t OFF_V = align16(sizeof(K) * (size_t)(TILE + 1));
    V* lv = reinterpret_cast<V*>(smem + OFF_V);
    constexpr size_t OFF_C = OFF_V + (HASV ? align16(sizeof(V) * (size_t)(TILE + 1)) : 0);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem + OFF_C);  // [WAVES][DIGITS]
    uint32_t* tstart = cnt + WAVES * DIGITS;                    // [DIGITS] first slot of a digit in the reordered tile
    uint32_t* tcnt = tstart + DIGITS;                           // [DIGITS] keys of a digit in the tile
    uint32_t* gpos = tcnt + DIGITS;                             // [DIGITS] out = gpos[d] + slot
    uint32_t* grun = gpos + DIGITS;                             // [DIGITS] RANGED: where the range's next key of a digit goes
    uint32_t* wsum = grun + DIGITS;                             // [16]
    uint32_t* shd = wsum + 16;                                  // [4] the tile claimed next: chain, number
    uint32_t* lbs = cnt;  // [LBG][LBT][5] what a look-back group found in its window (four sums, meta): in the counters, which are free between a tile's reordering and the next tile's ranking
    static_assert(LBG * LBT * 5 <= WAVES * DIGITS, "look-back windows do not fit the counters");
    static_assert(LBG >= 1 && LBG * LBT <= THREADS, "look-back groups");

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t mask = (1u << nbits) - 1u;
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(states, 0, (int)states_bytes, 0x00020000);

    // ---- which tiles
    const uint32_t total_tiles = (n + TILE - 1) / TILE;
    uint32_t rt_end = 0;  // RANGED: the range's end
    uint32_t home = 0;    // !RANGED (thread 0): the chain claimed from
    uint32_t pend_j = 0;  //                     the claim in flight (a tile number of chain `home`)
    auto pref_chain = [&](uint32_t k) -> uint32_t {  // the k-th chain in this workgroup's order of preference: its XCD's chains first
        if (nch < 8) return (home + k) % nch;
        const uint32_t per = nch >> 3, res = ((home & 7u) + k / per) & 7u, sub = ((home >> 3) + k % per) % per;
        return res + 8u * sub;
    };
    auto chain_tiles = [&](uint32_t c) -> uint32_t { return tile_first[c + 1] - tile_first[c]; };
    // the home chain is used up: any chain.  Called by the whole FIRST WAVEFRONT (uniformly): lane k looks at the k-th chain of the workgroup's
    // order of preference -- one round trip says which chains have tiles left -- and the claims are then tried in that order.
    auto claim_slow = [&](uint32_t& cc, uint32_t& jj) -> bool {
        const uint32_t h0 = __builtin_amdgcn_readfirstlane(home);
        home = h0;
        const uint32_t k = lane, c2 = k < nch ? pref_chain(k) : 0u;
        const uint32_t nt = k < nch ? chain_tiles(c2) : 0u;
        uint32_t hd = NONE;
        if (nt) hd = __hip_atomic_load(&heads[c2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t open = __ballot(nt && hd < nt);
        bool got = false;
        while (open && !got) {
            const int l = __builtin_ctzll(open);
            open &= open - 1ull;
            uint32_t j2 = 0;
            if ((int)lane == l) j2 = atomicAdd(&heads[c2], 1u);
            j2 = __shfl(j2, l);
            const uint32_t ntl = __shfl(nt, l), cl = __shfl(c2, l);
            if (j2 < ntl) { cc = cl; jj = j2; home = cl; got = true; }
        }
        return got;
    };
    K key[IPT];
    V val[IPT];
    uint32_t a0 = 0, tile_n = 0;
    auto load_tile = [&](uint32_t c, uint32_t j) {
        uint32_t rem;
        if (RANGED) { a0 = j * (uint32_t)TILE; rem = n - a0; }
        else {
            const uint32_t cs0 = BFT_RS_UNI(chain_start[c]), cs1 = BFT_RS_UNI(chain_start[c + 1]);
            a0 = cs0 + j * (uint32_t)TILE;
            rem = cs1 - a0;
        }
        tile_n = rem < (uint32_t)TILE ? rem : (uint32_t)TILE;
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t idx = wave * 64u * IPT + r * 64u + lane;
            const uint32_t at = a0 + (idx < tile_n ? idx : 0u);  // (lanes beyond a partial tile read its first key: no branch, nothing out of bounds)
            key[r] = in.key(at);
            if constexpr (HASV) val[r] = in.val(at);
        }
    };

    uint32_t cur_c = 0, cur_j = 0;
    if (RANGED) {
        cur_j = blockIdx.x * tpr;
        rt_end = min(total_tiles, cur_j + tpr);
        if (cur_j >= rt_end) return;
        if (nch == 0u) {
            // (a pass that counted its own digit: chain_start = the digits' totals.  Where the digits start in the output -- their exclusive
            // scan -- is 512 additions, made here by the first wavefront, eight digits per lane, instead of by a launch of its own; workgroup 0
            // leaves it in `heads` for the caller.)
            if (wave == 0) {
                uint32_t t[8], sum = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint32_t d = lane * 8u + (uint32_t)q;
                    t[q] = d <= mask ? chain_start[d] : 0u;
                    sum += t[q];
                }
                uint32_t inc = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t x = __shfl_up(inc, o);
                    if ((int)lane >= o) inc += x;
                }
                uint32_t run = inc - sum;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    grun[lane * 8u + (uint32_t)q] = run;
                    run += t[q];
                }
            }
            __syncthreads();
            for (uint32_t d = tid; d < DIGITS; d += THREADS) {
                const uint32_t db = grun[d];
                if (blockIdx.x == 0) heads[d] = db;
                grun[d] = d <= mask ? db + rows[(size_t)blockIdx.x * row_stride + d] : 0u;
            }
        } else
            for (uint32_t d = tid; d < DIGITS; d += THREADS) grun[d] = d <= mask ? dbase[d] + rows[(size_t)blockIdx.x * row_stride + d] : 0u;
    } else {
        if (wave == 0) {  // (uniform in the wavefront; lane 0 keeps `home` and the claim in flight)
            const uint32_t x = xcc_id();
            home = nch < 8 ? (blockIdx.x % nch) : x + 8u * ((blockIdx.x >> 3) % (nch >> 3));
            uint32_t cc = NONE, jj = 0;
            const uint32_t nt = chain_tiles(home);
            uint32_t j0 = 0;
            if (nt && lane == 0) j0 = atomicAdd(&heads[home], 2u);  // this tile and the one after it
            j0 = __builtin_amdgcn_readfirstlane(j0);
            if (nt && j0 < nt) { cc = home; jj = j0; pend_j = j0 + 1u; }
            else if (claim_slow(cc, jj)) {
                uint32_t pj = 0;
                if (lane == 0) pj = atomicAdd(&heads[home], 1u);
                pend_j = __builtin_amdgcn_readfirstlane(pj);
            } else
                cc = NONE;
            if (lane == 0) { shd[0] = cc; shd[1] = jj; }
        }
        __syncthreads();
        cur_c = BFT_RS_UNI(shd[0]);  // (uniform: scalar registers, not one vector register per value)
        cur_j = BFT_RS_UNI(shd[1]);
        if (cur_c == NONE) return;
    }
    load_tile(cur_c, cur_j);
    __syncthreads();  // (shd is rewritten below; grun is read behind barriers)

#ifdef BFT_RS_PROF
    unsigned long long t_prev = clock64();
#endif
    for (;;) {
        RS_T(0);
        // ---- rank: wave w owns the tile's keys [w 64 IPT, (w + 1) 64 IPT), key (round r, lane l) = r 64 + l of them
#pragma unroll
        for (int q = 0; q < DIGITS / 64; q++) cnt[wave * DIGITS + q * 64 + lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        RS_T(1);
        uint32_t rk2[(IPT + 1) / 2];  // ranks, two to a register (a tile holds fewer than 2^16 keys)
#pragma unroll
        for (int r = 0; r < (IPT + 1) / 2; r++) rk2[r] = 0;
        static_assert(TILE < 65536, "ranks are kept in 16 bits");
        if (BALLOT) {
            const uint64_t lt_mask = (1ull << lane) - 1ull;
            uint32_t b0v[IPT];
            int leader[IPT];
#pragma unroll
            for (int r = 0; r < IPT; r++) {
                const uint32_t idx = wave * 64u * IPT + r * 64u + lane;
                const bool valid = idx < tile_n;
                const uint32_t d = digit_of(key[r], bit, mask);
                uint64_t peers = __ballot(valid);
#pragma unroll
                for (int b = 0; b < DBITS; b++) {  // (bits beyond the digit are zero in every lane: they change nothing)
                    const bool on = (d >> b) & 1u;
                    const uint64_t bj = __ballot(on);
                    peers &= on ? bj : ~bj;
                }
                leader[r] = valid ? __builtin_ctzll(peers) : (int)lane;
                rk2[r >> 1] |= (uint32_t)__builtin_popcountll(peers & lt_mask) << (16 * (r & 1));
                b0v[r] = 0;
                if (valid && (int)lane == leader[r]) b0v[r] = atomicAdd(&cnt[wave * DIGITS + d], (uint32_t)__builtin_popcountll(peers));
            }
#pragma unroll
            for (int r = 0; r < IPT; r++) rk2[r >> 1] += __shfl(b0v[r], leader[r]) << (16 * (r & 1));
        } else {
#pragma unroll
            for (int r = 0; r < IPT; r++) {
                const uint32_t idx = wave * 64u * IPT + r * 64u + lane;
                uint32_t v = 0;
                if (idx < tile_n) v = atomicAdd(&cnt[wave * DIGITS + digit_of(key[r], bit, mask)], 1u);
                rk2[r >> 1] |= v << (16 * (r & 1));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (!RANGED && wave == 0) {  // the claim made a tile ago has returned by now (asked for here and not at the loop's top: behind it the
            uint32_t cc = home, jj = __builtin_amdgcn_readfirstlane(pend_j);  // stores of the last tile are in flight, and a wavefront's memory operations return in order)
            if (jj >= chain_tiles(home) && !claim_slow(cc, jj)) cc = NONE;
            if (lane == 0) { shd[0] = cc; shd[1] = jj; }
        }
        RS_T(2);
        __syncthreads();  // B
        RS_T(3);
        uint32_t nxt_c = 0, nxt_j = 0;
        bool more;
        if (RANGED) { nxt_j = cur_j + 1; more = nxt_j < rt_end; }
        else { nxt_c = BFT_RS_UNI(shd[0]); nxt_j = BFT_RS_UNI(shd[1]); more = nxt_c != NONE; }
        // ---- per digit: counts of the waves -> starts of (digit, wave) relative to the digit; the digit's total
        uint32_t s[DPT], tot = 0, inc = 0;
        if (tid < DT) {
#pragma unroll
            for (int q = 0; q < DPT; q++) {
                const uint32_t d = tid * DPT + q;
                uint32_t cw[WAVES], run = 0;
#pragma unroll
                for (int w = 0; w < WAVES; w++) cw[w] = cnt[w * DIGITS + d];  // (every read first: the loads in flight together, not one round trip each)
#pragma unroll
                for (int w = 0; w < WAVES; w++) {
                    cnt[w * DIGITS + d] = run;
                    run += cw[w];
                }
                s[q] = run;
                tot += run;
            }
            inc = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(inc, o);
                if ((int)lane >= o) inc += v;
            }
            if (lane == 63) wsum[wave] = inc;
        }
        RS_T(4);
        __syncthreads();  // D
        if (tid < DT) {
            uint32_t before = 0;
#pragma unroll
            for (int w = 0; w < DT / 64; w++)
                if (w < (int)wave) before += wsum[w];
            uint32_t start = before + inc - tot;
#pragma unroll
            for (int q = 0; q < DPT; q++) {
                const uint32_t d = tid * DPT + q;
                tstart[d] = start;
                if (RANGED) {  // the range's running positions: no look-back
                    const uint32_t g = grun[d];
                    gpos[d] = g - start;
                    grun[d] = g + s[q];
                } else
                    tcnt[d] = s[q];
                start += s[q];
            }
        }
        __syncthreads();  // E
        RS_T(5);
        // ---- publish the tile's counts, start looking back; claim the tile after the next
        const uint32_t tile_g = RANGED ? 0u : tile_first[cur_c] + cur_j;
        uint32_t my[4] = {0, 0, 0, 0}, ex[4] = {0, 0, 0, 0};
        u32x4 xb[LB];
        const uint32_t lg = tid / LBT, lt = tid % LBT;  // look-back group, thread of it: digits 4 lt .. 4 lt + 3; group g looks at the tiles cur_j - 1 - (g LB + i)
        if (!RANGED && tid < LBG * LBT) {
            if (lg == 0) {
                const uint32_t fl = cur_j == 0 ? ST_INC : ST_AGG;
#pragma unroll
                for (int q = 0; q < 4; q++) my[q] = tcnt[lt * 4 + q];
                u32x4 a;
                a.x = fl + my[0]; a.y = fl + my[1]; a.z = fl + my[2]; a.w = fl + my[3];
                __builtin_amdgcn_raw_buffer_store_b128(a, srsrc, (int)((tile_g * (uint32_t)DIGITS + lt * 4u) * 4u), 0, 16);
            }
            if (tid == 0 && more) pend_j = atomicAdd(&heads[nxt_c], 1u);  // (nxt_c is the chain the first wavefront claims from now: `home`)
        }
        RS_T(6);
        // ---- reorder in LDS
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t idx = wave * 64u * IPT + r * 64u + lane;
            const uint32_t d = digit_of(key[r], bit, mask);
            const uint32_t slot = idx < tile_n ? tstart[d] + cnt[wave * DIGITS + d] + ((rk2[r >> 1] >> (16 * (r & 1))) & 0xFFFFu) : (uint32_t)TILE;
            lk[slot] = key[r];
            if constexpr (HASV) lv[slot] = val[r];
        }
        if (!RANGED && tid < LBG * LBT) {
            if (cur_j > 0 && !(BFT_RS_KO & 1)) {
#pragma unroll
                for (int i = 0; i < LB; i++) {
                    const uint32_t o = lg * LB + i;
                    if (o < cur_j) xb[i] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, (int)(((tile_g - 1u - o) * (uint32_t)DIGITS + lt * 4u) * 4u), 0, 16);
                    else { xb[i].x = ST_INC; xb[i].y = ST_INC; xb[i].z = ST_INC; xb[i].w = ST_INC; }  // (before the chain's first tile: nothing)
                }
            }
        }
        RS_T(7);
        // ---- the next tile's keys: into the registers this tile's keys have just left
        const uint32_t this_n = tile_n;
        if (more && (RANGED || tid >= LBG * LBT)) load_tile(nxt_c, nxt_j);  // (the look-back threads: once they have looked back -- their registers are taken until then)
        RS_T(8);
        // ---- finish the look-back: every group sums its window up to the first tile that knows its inclusive counts (per digit), or up
        // to a tile that has not published yet; group 0 puts the windows together, and goes on alone in the rare case that is not enough
        if (!RANGED && cur_j > 0 && !(BFT_RS_KO & 1)) {
            uint32_t done = 0, used = 0;  // digits that met an inclusive state; window entries taken in
            if (LBG > 1) __syncthreads();  // (the window summaries go where the counters are: every wavefront has reordered its keys by now)
            if (tid < LBG * LBT) {
#pragma unroll
                for (int i = 0; i < LB; i++) {
                    if (done == 15u || used != (uint32_t)i) break;
                    const u32x4 x = xb[i];
                    const uint32_t xv[4] = {x.x, x.y, x.z, x.w};
                    bool wait = false;
#pragma unroll
                    for (int q = 0; q < 4; q++) wait |= !((done >> q) & 1u) && xv[q] == 0u;
                    if (wait) break;
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (!((done >> q) & 1u)) {
                            ex[q] += (xv[q] & ST_VAL) - 1u;
                            if (xv[q] >> 31) done |= 1u << q;
                        }
                    used = i + 1;
                }
                if (LBG > 1) {
                    uint32_t* o = lbs + (lg * LBT + lt) * 5;
                    o[0] = ex[0]; o[1] = ex[1]; o[2] = ex[2]; o[3] = ex[3];
                    o[4] = done | (used << 4);
                }
            }
            if (LBG > 1) __syncthreads();
            if (tid < LBT) {
                uint32_t pj;  // the next tile to look at, if any is left to
                if (LBG > 1) {
                    bool blocked = done != 15u && used < (uint32_t)LB;
                    uint32_t seen = used;
                    for (int g2 = 1; g2 < LBG && done != 15u && !blocked; g2++) {
                        const uint32_t* o = lbs + (g2 * LBT + lt) * 5;
                        const uint32_t meta = o[4];
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (!((done >> q) & 1u)) ex[q] += o[q];
                        done |= meta & 15u;
                        seen += meta >> 4;
                        blocked = (meta >> 4) < (uint32_t)LB;
                    }
                    pj = cur_j - 1u - seen;
                } else
                    pj = cur_j - 1u - used;
                while (done != 15u) {  // (one tile at a time; tile 0 of a chain is published inclusive: `done` is complete before pj wraps)
                    asm volatile("" ::: "memory");
                    const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(srsrc, (int)(((tile_g - cur_j + pj) * (uint32_t)DIGITS + lt * 4u) * 4u), 0, 16);
                    const uint32_t xv[4] = {x.x, x.y, x.z, x.w};
                    bool wait = false;
#pragma unroll
                    for (int q = 0; q < 4; q++) wait |= !((done >> q) & 1u) && xv[q] == 0u;
                    if (wait) { __builtin_amdgcn_s_sleep(2); continue; }
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (!((done >> q) & 1u)) {
                            ex[q] += (xv[q] & ST_VAL) - 1u;
                            if (xv[q] >> 31) done |= 1u << q;
                        }
                    pj--;
                }
                u32x4 a;
                a.x = ST_INC + (ex[0] + my[0]); a.y = ST_INC + (ex[1] + my[1]); a.z = ST_INC + (ex[2] + my[2]); a.w = ST_INC + (ex[3] + my[3]);
                __builtin_amdgcn_raw_buffer_store_b128(a, srsrc, (int)((tile_g * (uint32_t)DIGITS + lt * 4u) * 4u), 0, 16);
            }
        }
        if (!RANGED && more && tid < LBG * LBT) load_tile(nxt_c, nxt_j);
        if (!RANGED && tid < LBT) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t d = tid * 4 + q;
                gpos[d] = (d <= mask ? dbase[d] + rows[cur_c * row_stride + d] : 0u) + ex[q] - tstart[d];
            }
        }
        RS_T(9);
        __syncthreads();  // G
        RS_T(10);
        // ---- write out: slot by slot, i.e. digit run by digit run (in two halves: all of a tile's LDS reads scheduled ahead of its stores
        // cost more registers than the kernel has beside the next tile's keys)
        constexpr int WH = (IPT % 2 == 0 && IPT >= 8) ? 2 : 1;
#pragma unroll 1
        for (int h = 0; h < WH; h++) {
#pragma unroll
            for (int i2 = 0; i2 < IPT / WH; i2++) {
                const uint32_t slot = (uint32_t)(h * (IPT / WH) + i2) * THREADS + tid;
                if (slot < this_n) {
                    const K kk = lk[slot];
                    const uint32_t pos = gpos[digit_of(kk, bit, mask)] + slot;
                    if (!(BFT_RS_KO & 2) || kk == K(12345)) {
                        ok[pos] = kk;
                        if constexpr (HASV) ov[pos] = lv[slot];
                    }
                }
            }
        }
        RS_T(11);
        if (!more) break;
        cur_c = nxt_c;
        cur_j = nxt_j;
        // (no barrier: the next round writes shd before B -- every thread has read it behind B of this round --, the counters before B, the rest behind D)
    }
}

// ---- an array that is ONE tile: every pass in the LDS of one workgroup, one launch ----------------------------------------------------------
template <class K, class V, class In, int THREADS, int IPT, bool BALLOT>
__global__ __launch_bounds__(THREADS) void k_rs_tiny(In in, K* __restrict__ ok, V* __restrict__ ov, uint32_t n, Plan pl) {
    constexpr bool HASV = !std::is_same<V, NoVal>::value;
    constexpr int WAVES = THREADS / 64, TILE = THREADS * IPT;
    constexpr int DT = THREADS < DIGITS ? THREADS : DIGITS, DPT = DIGITS / DT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    K* lk = reinterpret_cast<K*>(smem);
    constexpr size_t OFF_V = align16(sizeof(K) * (size_t)(TILE + 1));
    V* lv = reinterpret_cast<V*>(smem + OFF_V);
    constexpr size_t OFF_C = OFF_V + (HASV ? align16(sizeof(V) * (size_t)(TILE + 1)) : 0);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem + OFF_C);  // [WAVES][DIGITS]
    uint32_t* tstart = cnt + WAVES * DIGITS;                    // [DIGITS]
    uint32_t* wsum = tstart + DIGITS;                           // [16]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    K key[IPT];
    V val[IPT];
#pragma unroll
    for (int r = 0; r < IPT; r++) {
        const uint32_t idx = wave * 64u * IPT + r * 64u + lane;
        const uint32_t at = idx < n ? idx : 0u;
        key[r] = in.key(at);
        if constexpr (HASV) val[r] = in.val(at);
    }
    for (int p = 0; p < pl.P; p++) {
        const uint32_t bit = pl.bit[p], mask = (1u << pl.nbits[p]) - 1u;
#pragma unroll
        for (int q = 0; q < DIGITS / 64; q++) cnt[wave * DIGITS + q * 64 + lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t rank[IPT];
        if (BALLOT) {
            const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
            for (int r = 0; r < IPT; r++) {
                const uint32_t idx = wave * 64u * IPT + r * 64u + lane;
                const bool valid = idx < n;
                const uint32_t d = digit_of(key[r], bit, mask);
                uint64_t peers = __ballot(valid);
#pragma unroll
                for (int b = 0; b < DBITS; b++) {
                    const bool on = (d >> b) & 1u;
                    const uint64_t bj = __ballot(on);
                    peers &= on ? bj : ~bj;
                }
                const int leader = valid ? __builtin_ctzll(peers) : (int)lane;
                uint32_t b0 = 0;
                if (valid && (int)lane == leader) b0 = atomicAdd(&cnt[wave * DIGITS + d], (uint32_t)__builtin_popcountll(peers));
                rank[r] = __shfl(b0, leader) + (uint32_t)__builtin_popcountll(peers & lt_mask);
            }
        } else {
#pragma unroll
            for (int r = 0; r < IPT; r++) {
                const uint32_t idx = wave * 64u * IPT + r * 64u + lane;
                rank[r] = 0;
                if (idx < n) rank[r] = atomicAdd(&cnt[wave * DIGITS + digit_of(key[r], bit, mask)], 1u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __syncthreads();
        uint32_t sq[DPT], tot = 0, inc = 0;
        if (tid < DT) {
#pragma unroll
            for (int q = 0; q < DPT; q++) {
                const uint32_t d = tid * DPT + q;
                uint32_t cw[WAVES], run = 0;
#pragma unroll
                for (int w = 0; w < WAVES; w++) cw[w] = cnt[w * DIGITS + d];
#pragma unroll
                for (int w = 0; w < WAVES; w++) {
                    cnt[w * DIGITS + d] = run;
                    run += cw[w];
                }
                sq[q] = run;
                tot += run;
            }
            inc = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(inc, o);
                if ((int)lane >= o) inc += v;
            }
            if (lane == 63) wsum[wave] = inc;
        }
        __syncthreads();
        if (tid < DT) {
            uint32_t before = 0;
#pragma unroll
            for (int w = 0; w < DT / 64; w++)
                if (w < (int)wave) before += wsum[w];
            uint32_t start = before + inc - tot;
#pragma unroll
            for (int q = 0; q < DPT; q++) {
                tstart[tid * DPT + q] = start;
                start += sq[q];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t idx = wave * 64u * IPT + r * 64u + lane;
            const uint32_t d = digit_of(key[r], bit, mask);
            const uint32_t slot = idx < n ? tstart[d] + cnt[wave * DIGITS + d] + rank[r] : (uint32_t)TILE;
            lk[slot] = key[r];
            if constexpr (HASV) lv[slot] = val[r];
        }
        __syncthreads();
        if (p + 1 < pl.P) {  // the next pass ranks the new order
#pragma unroll
            for (int r = 0; r < IPT; r++) {
                const uint32_t idx = wave * 64u * IPT + r * 64u + lane;
                const uint32_t at = idx < n ? idx : 0u;
                key[r] = lk[at];
                if constexpr (HASV) val[r] = lv[at];
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        const uint32_t slot = i * THREADS + tid;
        if (slot < n) {
            ok[slot] = lk[slot];
            if constexpr (HASV) ov[slot] = lv[slot];
        }
    }
}

template <class K, class V, class In>
__global__ void k_rs_copy(In in, uint32_t n, K* __restrict__ ok, V* __restrict__ ov) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        ok[i] = in.key(i);
        if constexpr (!std::is_same<V, NoVal>::value) ov[i] = in.val(i);
    }
}

template <class K, class V, int THREADS, int IPT>
constexpr size_t pass_lds_bytes() {
    return align16(sizeof(K) * ((size_t)THREADS * IPT + 1)) + (std::is_same<V, NoVal>::value ? 0 : align16(sizeof(V) * ((size_t)THREADS * IPT + 1))) +
           ((size_t)(THREADS / 64) * DIGITS + 4 * DIGITS + 16 + 4) * 4;
}

// the dynamic-LDS limit of a kernel, raised once per device (the call costs the host tens of microseconds: between two passes that is a gap on the GPU)
template <auto KERNEL>
static inline void lds_limit(int bytes) {
    static unsigned done[1] = {0u};  // (bit d: device d has the limit; a kernel is launched with one size)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); return; }
    if (done[0] & (1u << dev)) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done[0] |= 1u << dev;
}
// does the kernel use scratch memory (spilled registers)?  Asked once per kernel.  A kernel that does starts ~0.13 ms late behind kernels that do not
// (the runtime sets the queue's scratch up again at the dispatch): where there is a choice, the variant without is launched.
template <auto KERNEL>
static inline bool uses_scratch() {
    static int known = -1;
    if (known < 0) {
        hipFuncAttributes at;
        if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(KERNEL)) == hipSuccess) known = at.localSizeBytes > 0 ? 1 : 0;
        else { known = 1; (void)hipGetLastError(); }
    }
    return known == 1;
}
static inline int cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// 0: ranks from LDS atomics (the device passed the lane-order check), 1: from ballots.  force >= 0 sets it ("sort_ballots" option, tests).
extern int g_bft_rs_rank_mode;  // (one per process: bft_gpu.hip; -1 = not checked yet)
static inline int rank_mode(hipStream_t s, int force = -1) {
    int& mode = g_bft_rs_rank_mode;
    if (force >= 0) { mode = force; return mode; }
    if (mode >= 0) return mode;
    uint32_t* d = nullptr;
    uint32_t bad = 1;
    if (hipMalloc((void**)&d, 4) == hipSuccess) {
        if (hipMemsetAsync(d, 0, 4, s) == hipSuccess) {
            hipLaunchKernelGGL(k_rs_selftest, dim3(512), dim3(64), 0, s, d);
            if (hipMemcpyAsync(&bad, d, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) bad = 1;
        }
        (void)hipFree(d);
    }
    (void)hipGetLastError();
    mode = bad ? 1 : 0;
    return mode;
}

// scratch of one sort call (u32 words)
struct Layout {
    size_t heads, cnt, states, zero_words;  // zeroed before every call: [P][64]; the later passes' (chain, digit) counters; [P - 1][max_tiles][DIGITS]
    size_t tot, dbase, chain, partial, total_words;
    uint32_t max_tiles, ranges;
};

// Stable sort of n (key, value) pairs on the key bits [begin_bit, end_bit).  The input is read through `in` (once by the histogram kernel,
// once by the first pass); the result lands in out_k / out_v; tmp_k / tmp_v (n entries each) carry the passes in between and may be NULL
// for a one-pass sort.  `in` may read out_k / out_v only when the number of passes is even.  V = NoVal: keys only.
template <class K, class V, class In, int THREADS, int IPT>
int sort_cfg(In in, uint64_t n, K* out_k, V* out_v, K* tmp_k, V* tmp_v, unsigned begin_bit, unsigned end_bit, hipStream_t s, DevBuf& scratch, const uint32_t** last_dbase) {
    if (last_dbase) *last_dbase = nullptr;
    if (n >= (1ull << 31) - 1) return bft_fail(BFT_GPU_E_LIMIT, "internal: radix sort of 2^31 entries or more");
    if (end_bit < begin_bit || end_bit - begin_bit > (unsigned)(MAXP * DBITS) || end_bit > sizeof(K) * 8) return bft_fail(BFT_GPU_E_ARG, "internal: radix sort bit range");
    constexpr uint32_t TILE = (uint32_t)THREADS * IPT;
    const Plan pl = make_plan(begin_bit, end_bit);
    if (n == 0) return 0;
    if (pl.P == 0) {
        hipLaunchKernelGGL((k_rs_copy<K, V, In>), dim3(bft_grid_for((n + 255) / 256)), dim3(256), 0, s, in, (uint32_t)n, out_k, out_v);
        HIPCK(hipGetLastError());
        return 0;
    }
    if (pl.P > 1 && (!tmp_k || (!std::is_same<V, NoVal>::value && !tmp_v))) return bft_fail(BFT_GPU_E_ARG, "internal: radix sort without a second buffer");
    const bool ballot = rank_mode(s) != 0;
    const int cus = cu_count();
    constexpr size_t lds = pass_lds_bytes<K, V, THREADS, IPT>();
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds, 2048 / THREADS));
    const uint32_t tiles = (uint32_t)((n + TILE - 1) / TILE);
    const uint32_t wgs = std::min<uint32_t>(tiles, (uint32_t)std::min(1024, cus * per_cu));  // (<= 1024 ranges: k_rs_rowscan)
    const uint32_t tpr = (tiles + wgs - 1) / wgs, ranges = (tiles + tpr - 1) / tpr;
    constexpr int HT = THREADS;  // (the histogram kernel in the shape of the passes: a 1024-thread workgroup waits long for a CU beside another stream's small ones)
    if (tiles == 1) {  // one tile: every pass in one workgroup's LDS
        if (ballot) {
            lds_limit<&k_rs_tiny<K, V, In, THREADS, IPT, true>>((int)lds);
            hipLaunchKernelGGL((k_rs_tiny<K, V, In, THREADS, IPT, true>), dim3(1), dim3(THREADS), lds, s, in, out_k, out_v, (uint32_t)n, pl);
        } else {
            lds_limit<&k_rs_tiny<K, V, In, THREADS, IPT, false>>((int)lds);
            hipLaunchKernelGGL((k_rs_tiny<K, V, In, THREADS, IPT, false>), dim3(1), dim3(THREADS), lds, s, in, out_k, out_v, (uint32_t)n, pl);
        }
        HIPCK(hipGetLastError());
        return 0;
    }
    if (n < BFT_RS_CHAIN_MIN || pl.P == 1) {
        // Passes that each count their own digit first: a histogram kernel per pass (one more read of the keys), the scan of the digit's counts over
        // the ranges, the pass (which finds the digits' starts itself: nch = 0), and no pass ever looks back --
        // every workgroup owns a contiguous range of tiles in every pass.  For arrays of up to a few 10^7 entries the extra read costs less than
        // the look-back's round trips and the chains' bookkeeping.
        const size_t words = (size_t)2 * DIGITS + (size_t)CT * 2 + (size_t)ranges * DIGITS + 8;
        if (scratch.bytes < words * 4) CK(scratch.alloc(words * 4));
        uint32_t* W = scratch.as<uint32_t>();
        uint32_t *tot = W, *dbase = W + DIGITS, *chain = W + 2 * DIGITS, *partial = W + 2 * DIGITS + 2 * CT;
        const K* src_k = nullptr;
        const V* src_v = nullptr;
        for (int p = 0; p < pl.P; p++) {
            const bool to_out = ((pl.P - 1 - p) % 2) == 0;
            K* dk = to_out ? out_k : tmp_k;
            V* dv = to_out ? out_v : tmp_v;
            const Plan p1 = make_plan(pl.bit[p], pl.bit[p] + pl.nbits[p]);
#define BFT_RS_ONEPASS(INTYPE, INVAL, BAL)                                                                                                                                     \
    do {                                                                                                                                                                       \
        hipLaunchKernelGGL((k_rs_hist<K, INTYPE, HT>), dim3(ranges), dim3(HT), (size_t)p1.hwords * 4, s, INVAL, (uint32_t)n, p1, tpr * TILE, partial);                          \
        hipLaunchKernelGGL(k_rs_rowscan, dim3(1u << p1.nbits[0], 1), dim3(256), 0, s, partial, partial, p1, ranges, tot);                                                       \
        lds_limit<&k_rs_pass<K, V, INTYPE, THREADS, IPT, true, BAL, 1>>((int)lds);   \
        hipLaunchKernelGGL((k_rs_pass<K, V, INTYPE, THREADS, IPT, true, BAL, 1>), dim3(ranges), dim3(THREADS), lds, s, INVAL, dk, dv, (uint32_t)n, p1.bit[0], p1.nbits[0],      \
                           partial, p1.hwords, dbase, tpr, tot, chain, 0u, dbase, chain, 16u);                                                                                 \
    } while (0)
            if (p == 0) {
                if (ballot) BFT_RS_ONEPASS(In, in, true);
                else BFT_RS_ONEPASS(In, in, false);
            } else {
                const PtrIn<K, V> pin{src_k, src_v};
                if (ballot) BFT_RS_ONEPASS(decltype(pin), pin, true);
                else BFT_RS_ONEPASS(decltype(pin), pin, false);
            }
#undef BFT_RS_ONEPASS
            src_k = dk;
            src_v = dv;
        }
        HIPCK(hipGetLastError());
        if (last_dbase) *last_dbase = dbase;
        return 0;
    }
    Layout L;
    {
        L.max_tiles = tiles + (1u << MAXCB);
        L.ranges = ranges;
        size_t o = 0;
        L.heads = o; o += (size_t)pl.P * (1 << MAXCB);
        L.cnt = o; o += pl.hwords;  // (the part of pass 0 is unused: its rows are the histogram workgroups' own blocks)
        o = (o + 3) & ~(size_t)3;
        L.states = o; o += (size_t)(pl.P - 1) * L.max_tiles * DIGITS;
        L.zero_words = o;
        L.tot = o; o += (size_t)pl.P * DIGITS;
        L.dbase = o; o += (size_t)pl.P * DIGITS;
        L.chain = o; o += (size_t)(pl.P + 1) * CT;
        o = (o + 3) & ~(size_t)3;
        L.partial = o; o += (size_t)ranges * pl.hwords;
        L.total_words = o;
    }
    if (scratch.bytes < L.total_words * 4) CK(scratch.alloc(L.total_words * 4));
    uint32_t* W = scratch.as<uint32_t>();
    HIPCK(hipMemsetAsync(W, 0, L.zero_words * 4, s));
    lds_limit<&k_rs_hist<K, In, HT>>(160 * 1024 - 64);
    hipLaunchKernelGGL((k_rs_hist<K, In, HT>), dim3(ranges), dim3(HT), (size_t)pl.hwords * 4, s, in, (uint32_t)n, pl, tpr * TILE, W + L.partial);
    // per digit: the ranges' counts (pass 0, in place in the histogram blocks) and the chains' (later passes) -> where each of them starts inside the digit
    uint32_t maxnb = 0;
    for (int p = 0; p < pl.P; p++) maxnb = std::max(maxnb, pl.nbits[p]);
    if (pl.P > 1) {
        const uint32_t words = pl.hwords - pl.hoff[1];
        hipLaunchKernelGGL(k_rs_reduce, dim3((words + 255) / 256, std::max(1u, std::min(8u, ranges / 32u))), dim3(256), 0, s, W + L.partial, ranges, pl.hwords, pl.hoff[1], words,
                           W + L.cnt + pl.hoff[1]);
    }
    hipLaunchKernelGGL(k_rs_rowscan, dim3(1u << maxnb, pl.P), dim3(256), 0, s, W + L.partial, W + L.cnt, pl, ranges, W + L.tot);
    hipLaunchKernelGGL(k_rs_digits, dim3(pl.P), dim3(DIGITS), 0, s, W + L.tot, W + L.dbase, W + L.chain, pl, (uint32_t)n, TILE);
    HIPCK(hipGetLastError());
    const K* src_k = nullptr;
    const V* src_v = nullptr;
    for (int p = 0; p < pl.P; p++) {
        const bool to_out = ((pl.P - 1 - p) % 2) == 0;
        K* dk = to_out ? out_k : tmp_k;
        V* dv = to_out ? out_v : tmp_v;
        const uint32_t* ch = W + L.chain + (size_t)p * CT;
        if (p == 0) {
#define BFT_RS_LAUNCH0(BAL)                                                                                                                                            \
    do {                                                                                                                                                               \
        lds_limit<&k_rs_pass<K, V, In, THREADS, IPT, true, BAL, 1>>((int)lds); \
        hipLaunchKernelGGL((k_rs_pass<K, V, In, THREADS, IPT, true, BAL, 1>), dim3(ranges), dim3(THREADS), lds, s, in, dk, dv, (uint32_t)n, pl.bit[0], pl.nbits[0],       \
                           W + L.partial + pl.hoff[0], pl.hwords, W + L.dbase, tpr, ch, ch, 1u, W + L.heads, W + L.states, 16u);                                       \
    } while (0)
            if (ballot) BFT_RS_LAUNCH0(true);
            else BFT_RS_LAUNCH0(false);
#undef BFT_RS_LAUNCH0
        } else {
            uint32_t* st = W + L.states + (size_t)(p - 1) * L.max_tiles * DIGITS;
            const uint32_t st_bytes = L.max_tiles * DIGITS * 4;
#define BFT_RS_LAUNCH1(BAL, G)                                                                                                                                                      \
    do {                                                                                                                                                                            \
        lds_limit<&k_rs_pass<K, V, PtrIn<K, V>, THREADS, IPT, false, BAL, G>>((int)lds); \
        hipLaunchKernelGGL((k_rs_pass<K, V, PtrIn<K, V>, THREADS, IPT, false, BAL, G>), dim3(grid), dim3(THREADS), lds, s, PtrIn<K, V>{src_k, src_v}, dk, dv, (uint32_t)n,         \
                           pl.bit[p], pl.nbits[p], W + L.cnt + pl.hoff[p], 1u << pl.nbits[p], W + L.dbase + (size_t)p * DIGITS, 0u, ch, ch + (1 << MAXCB) + 1, 1u << pl.cb[p],     \
                           W + L.heads + (size_t)p * (1 << MAXCB), st, st_bytes);                                                                                                   \
    } while (0)
            // (a tile looks back over the tiles its chain's other workgroups are working on: one group of look-back threads sees four of them per
            // round trip, all the groups of the workgroup THREADS / 32 -- the grid is kept within what they see at once)
            constexpr int GMAX = THREADS / (DIGITS / 4);
            const uint32_t nchp = 1u << pl.cb[p];
            uint32_t grid = std::min<uint32_t>(tiles + nchp, (uint32_t)(cus * per_cu));
            // (the one-group form where the grid is small enough for it -- unless it uses scratch memory: for 8-byte keys without a payload it
            // spills five registers, the eight-group form none, and a kernel with scratch starts ~0.13 ms late behind kernels without)
            if (grid <= nchp * 4u && ballot) {
                BFT_RS_LAUNCH1(true, 1);
            } else if (grid <= nchp * 4u && !ballot && !uses_scratch<&k_rs_pass<K, V, PtrIn<K, V>, THREADS, IPT, false, false, 1>>()) {
                BFT_RS_LAUNCH1(false, 1);
            } else {
                grid = std::min<uint32_t>(grid, nchp * 4u * GMAX);
                if (ballot) BFT_RS_LAUNCH1(true, GMAX);
                else BFT_RS_LAUNCH1(false, GMAX);
            }
#undef BFT_RS_LAUNCH1
        }
        src_k = dk;
        src_v = dv;
    }
    HIPCK(hipGetLastError());
    if (last_dbase) *last_dbase = W + L.dbase + (size_t)(pl.P - 1) * DIGITS;
    return 0;
}

// The tile shape by the bytes of an entry: 1024-thread workgroups, one per CU, with the largest tile the LDS holds (a digit's piece of a tile
// is TILE / 512 entries: the longer, the fewer partial lines).  By size: one tile -> one launch, every pass in LDS (k_rs_tiny); up to
// BFT_RS_CHAIN_MIN entries -> a histogram kernel and a ranged pass per digit; beyond -> one histogram kernel, a ranged first pass, chained
// passes behind it (2 x 10^8 composites: 2.2-2.3 ms against 2.75 with a histogram per pass; 1.6 x 10^6 pairs, eight passes: 0.18 ms against
// 0.43 chained -- and 0.27 for rocPRIM's onesweep --; 4.5 x 10^7 k-mer hash records: 1.33 against 1.50).
// *last_dbase (optional): where, in `scratch`, the last pass's digits start in the output (512 words; NULL for an array of one tile)
// SHAPE: BIG (default) -- 1024-thread workgroups, the largest tile the LDS holds, one workgroup per CU; LIGHT -- 256 threads and ~40 KB of LDS,
// for the sorts of 10^4 .. 10^7 entries on the build's main stream; BACK -- 1024 threads and ~105 KB, for the k-mer hash's sort on the
// build's second stream: a BACK and a LIGHT workgroup fit one CU together, so the persistent workgroups of the one sort do not keep the
// other's from being placed (side by side in the BIG shape the (node, CC) sorts of the assembly took 1.6 ms instead of 0.3).
enum { SHAPE_BIG = 0, SHAPE_LIGHT = 1, SHAPE_BACK = 2 };
template <class K, class V, class In, int SHAPE = SHAPE_BIG>
int sort(In in, uint64_t n, K* out_k, V* out_v, K* tmp_k, V* tmp_v, unsigned begin_bit, unsigned end_bit, hipStream_t s, DevBuf& scratch, const uint32_t** last_dbase = nullptr) {
    constexpr size_t E = sizeof(K) + (std::is_same<V, NoVal>::value ? 0 : sizeof(V));
    if constexpr (SHAPE == SHAPE_LIGHT) {
        constexpr int IPT = E <= 8 ? 12 : E <= 12 ? 8 : E <= 16 ? 6 : E <= 24 ? 4 : E <= 32 ? 3 : 2;
        return sort_cfg<K, V, In, 256, IPT>(in, n, out_k, out_v, tmp_k, tmp_v, begin_bit, end_bit, s, scratch, last_dbase);
    } else if constexpr (SHAPE == SHAPE_BACK) {
        constexpr int IPT = E <= 8 ? 8 : E <= 12 ? 5 : E <= 16 ? 4 : E <= 24 ? 3 : E <= 32 ? 2 : 1;
        return sort_cfg<K, V, In, 1024, IPT>(in, n, out_k, out_v, tmp_k, tmp_v, begin_bit, end_bit, s, scratch, last_dbase);
    } else {
        // (ten 8-byte keys per thread, not twelve: at twelve the chained pass spills two registers, and a kernel that uses scratch memory starts
        // ~0.13 ms late whenever the kernels before it used none -- the runtime hands the queue's scratch back and has to find it again)
        constexpr int IPT_BIG = E <= 8 ? 12 : E <= 12 ? 8 : E <= 16 ? 6 : E <= 24 ? 4 : E <= 32 ? 3 : 2;
#ifdef BFT_RS_BIG_IPT
        return sort_cfg<K, V, In, BFT_RS_BIG_THREADS, (E <= 8 ? BFT_RS_BIG_IPT : IPT_BIG)>(in, n, out_k, out_v, tmp_k, tmp_v, begin_bit, end_bit, s, scratch, last_dbase);
#else
        return sort_cfg<K, V, In, BFT_RS_BIG_THREADS, IPT_BIG>(in, n, out_k, out_v, tmp_k, tmp_v, begin_bit, end_bit, s, scratch, last_dbase);
#endif
    }
}

// the same with the buffers between the passes and the scratch taken from the device-memory cache for the duration of the call (they go back
// to it stream-ordered: DevBuf)
template <class K, class V, class In, int SHAPE = SHAPE_BIG>
int sort_in(In in, uint64_t n, K* out_k, V* out_v, unsigned begin_bit, unsigned end_bit, hipStream_t s) {
    DevBuf tk, tv, scratch;
    if (make_plan(begin_bit, end_bit).P > 1 && n) {
        CK(tk.alloc(n * sizeof(K)));
        if (!std::is_same<V, NoVal>::value) CK(tv.alloc(n * sizeof(V)));
    }
    return sort<K, V, In, SHAPE>(in, n, out_k, out_v, tk.as<K>(), tv.as<V>(), begin_bit, end_bit, s, scratch);
}
template <class K, class V, int SHAPE = SHAPE_BIG>
int sort_pairs(const K* in_k, const V* in_v, uint64_t n, K* out_k, V* out_v, unsigned begin_bit, unsigned end_bit, hipStream_t s) {
    return sort_in<K, V, PtrIn<K, V>, SHAPE>(PtrIn<K, V>{in_k, in_v}, n, out_k, out_v, begin_bit, end_bit, s);
}
template <class K>
int sort_keys(const K* in_k, uint64_t n, K* out_k, unsigned begin_bit, unsigned end_bit, hipStream_t s) {
    return sort_in<K, NoVal, PtrIn<K, NoVal>>(PtrIn<K, NoVal>{in_k, nullptr}, n, out_k, (NoVal*)nullptr, begin_bit, end_bit, s);
}

}  // namespace bft_rs
