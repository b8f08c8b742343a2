// bft_sort.h -- the library's own stable radix sort / radix partition (device code + host launcher; templates, header only).
//
// What it is for: the reference keeps every container sorted by inserting one k-mer at a time (insertKmer_Node src/insertNode.c:38-226,
// transform2CC's sort by rotated prefix src/CC.c:40-367, insertKmer_UC src/UC.c:13-79); the bulk build sorts instead, and until round 6
// every device-wide sort of it was a rocPRIM / hipCUB call (42 % of a config-3 build's device time).  This is the replacement, written for
// the part it runs on:
//
//   * LSD passes of up to 9 bits, "onesweep" style: ONE read and ONE write of the array per pass, the global position of a tile's keys found
//     by a decoupled look-back over the tiles before it instead of a second pass over per-tile histograms;
//   * a tile (THREADS x IPT keys, up to 12288) is RANKED in registers (wavefront ballots: stable by construction), REORDERED IN LDS and
//     written out by digit runs, so that a digit's keys leave as one contiguous piece instead of one transaction per key;
//   * the look-back runs in EIGHT INDEPENDENT CHAINS, one per XCD: the input of a pass is cut into eight contiguous ranges, a workgroup takes
//     its tiles from the range of the XCD it runs on (HW_REG_XCC_ID; any other workgroup may steal: placement is never a matter of
//     correctness), and the histogram kernel counts per (chain, digit) so that every chain knows where its digits start.  Neighbouring
//     tiles of a chain write neighbouring pieces of every digit's output: the partial lines at the seams meet in ONE L2 instead of being
//     written back, byte-masked, by two.  For the passes after the first the chains are ranges of the previous pass's digit (its top
//     three bits), which the one histogram kernel in front of all passes can count from the keys alone;
//   * tile states are 32-bit words {flag:2, count:30}, four digits per 16-byte write-through (sc1) store / load (MI355X_MICROARCH.md,
//     inter-workgroup visibility: narrow sc1 stores are a fabric write each); the aggregate is published before the tile looks back;
//   * the next tile's keys are loaded into the registers the current tile's keys have just left (they sit in LDS by then), so the memory
//     system is never idle while a tile is ranked.
//
// n < 2^30 per call (the callers' arrays are rows and pairs counted in 32 bits; the insertion log is flushed before 2^30 pairs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "bft_dev.h"

namespace bft_rs {

constexpr int DBITS = 9;
constexpr int DIGITS = 1 << DBITS;
constexpr int CHAINS = 8;
constexpr int MAXP = 8;
constexpr uint32_t ST_VAL = 0x3FFFFFFFu;
constexpr uint32_t ST_AGG = 1u << 30, ST_INC = 2u << 30;

struct NoVal {};

struct Plan {
    int P;
    uint32_t bit[MAXP], nbits[MAXP], csh[MAXP];  // csh: chain of the NEXT pass = digit >> csh
};
static inline Plan make_plan(unsigned begin_bit, unsigned end_bit) {
    Plan pl;
    const unsigned bits = end_bit - begin_bit;
    pl.P = (int)((bits + DBITS - 1) / DBITS);
    unsigned b = begin_bit;
    for (int p = 0; p < pl.P; p++) {
        const unsigned left = end_bit - b, nb = (left + (pl.P - p) - 1) / (pl.P - p);
        pl.bit[p] = b;
        pl.nbits[p] = nb;
        pl.csh[p] = nb > 3 ? nb - 3 : 0;
        b += nb;
    }
    for (int p = pl.P; p < MAXP; p++) pl.bit[p] = pl.nbits[p] = pl.csh[p] = 0;
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

// scratch of one sort call (u32 words)
struct Layout {
    size_t heads, states, zero_words;  // [P][CHAINS]; [P][max_tiles][DIGITS]  -- zeroed before every call
    size_t cnt, base, chain, partial, total_words;
    uint32_t max_tiles, hist_wgs;
};
static inline Layout make_layout(uint64_t n, int P, uint32_t tile, uint32_t hist_wgs) {
    Layout L;
    L.max_tiles = (uint32_t)((n + tile - 1) / tile) + CHAINS;
    L.hist_wgs = hist_wgs;
    size_t o = 0;
    L.heads = o; o += (size_t)P * CHAINS;
    o = (o + 3) & ~(size_t)3;
    L.states = o; o += (size_t)P * L.max_tiles * DIGITS;
    L.zero_words = o;
    L.cnt = o; o += (size_t)P * CHAINS * DIGITS;
    L.base = o; o += (size_t)P * CHAINS * DIGITS;
    L.chain = o; o += (size_t)(P + 1) * 2 * (CHAINS + 1);  // per pass: chain_start[9], tile_first[9]
    o = (o + 3) & ~(size_t)3;
    L.partial = o; o += (size_t)hist_wgs * P * CHAINS * DIGITS;
    L.total_words = o;
    return L;
}

// ---- histogram of every pass's digits, per (chain, digit), in one read of the keys ------------------------------------------------------
// grid = CHAINS x wg_per_chain; workgroup (c, i) counts a slice of chain c of the INPUT (chain c = keys [c per0, (c + 1) per0)).
template <class K, class In, int THREADS>
__global__ __launch_bounds__(THREADS) void k_rs_hist(In in, uint32_t n, Plan pl, uint32_t per0, uint32_t wg_per_chain, uint32_t* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) uint32_t h[];  // [P][CHAINS][DIGITS]
    const uint32_t tid = threadIdx.x, words = (uint32_t)pl.P * CHAINS * DIGITS;
    for (uint32_t i = tid; i < words; i += THREADS) h[i] = 0;
    __syncthreads();
    const uint32_t c = blockIdx.x / wg_per_chain, wi = blockIdx.x % wg_per_chain;
    const uint64_t lo64 = (uint64_t)c * per0;
    const uint32_t lo = (uint32_t)(lo64 < n ? lo64 : n), hi = (uint32_t)(lo64 + per0 < n ? lo64 + per0 : n), len = hi - lo;
    constexpr uint32_t U = 8;
    uint32_t sl = (len + wg_per_chain - 1) / wg_per_chain;
    sl = (sl + THREADS * U - 1) / (THREADS * U) * (THREADS * U);
    const uint64_t a64 = (uint64_t)lo + (uint64_t)wi * sl;
    const uint32_t a = (uint32_t)(a64 < hi ? a64 : hi), b = (uint32_t)(a64 + sl < hi ? a64 + sl : hi);
    for (uint32_t i0 = a; i0 < b; i0 += THREADS * U) {
        K key[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t idx = i0 + u * THREADS + tid;
            key[u] = idx < b ? in.key(idx) : K(0);
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t idx = i0 + u * THREADS + tid;
            if (idx < b) {
                uint32_t ch = c;
                for (int p = 0; p < pl.P; p++) {
                    const uint32_t d = digit_of(key[u], pl.bit[p], (1u << pl.nbits[p]) - 1u);
                    atomicAdd(&h[((uint32_t)p * CHAINS + ch) * DIGITS + d], 1u);
                    ch = d >> pl.csh[p];
                }
            }
        }
    }
    __syncthreads();
    uint32_t* out = partial + (size_t)blockIdx.x * words;
    for (uint32_t i = tid; i < words; i += THREADS) out[i] = h[i];
}

// cnt[w] = sum over the histogram workgroups
__global__ __launch_bounds__(256) void k_rs_reduce(const uint32_t* __restrict__ partial, uint32_t nwg, uint32_t words, uint32_t* __restrict__ cnt) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= words) return;
    uint32_t s = 0;
#pragma unroll 8
    for (uint32_t g = 0; g < nwg; g++) s += partial[(size_t)g * words + w];
    cnt[w] = s;
}

// one workgroup of DIGITS threads per pass: where every (chain, digit) of the pass starts in its output; the chains and tiles of the NEXT pass
__global__ __launch_bounds__(DIGITS) void k_rs_scan(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ base, uint32_t* __restrict__ chain, Plan pl, uint32_t n,
                                                   uint32_t per0, uint32_t tile) {
    __shared__ uint32_t wsum[DIGITS / 64];
    __shared__ uint32_t dbase[DIGITS + 1];
    const uint32_t p = blockIdx.x, d = threadIdx.x, lane = d & 63u, wave = d >> 6;
    const uint32_t* c0 = cnt + (size_t)p * CHAINS * DIGITS;
    uint32_t cc[CHAINS], tot = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { cc[c] = c0[c * DIGITS + d]; tot += cc[c]; }
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o);
        if ((int)lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; w++) before += wsum[w];
    uint32_t run = before + inc - tot;
    dbase[d] = run;
    if (d == DIGITS - 1) dbase[DIGITS] = n;
    uint32_t* b0 = base + (size_t)p * CHAINS * DIGITS;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { b0[c * DIGITS + d] = run; run += cc[c]; }
    __syncthreads();
    if (d == 0) {
        // chains of pass p + 1: ranges of this pass's digit; of pass 0: ranges of the input
        uint32_t* cs = chain + (size_t)(p + 1) * 2 * (CHAINS + 1);
        const uint32_t ndig = 1u << pl.nbits[p];
        for (uint32_t c = 0; c <= CHAINS; c++) {
            const uint32_t first = c << pl.csh[p];
            cs[c] = (c < CHAINS && first < ndig) ? dbase[first] : n;
        }
        uint32_t t = 0;
        for (uint32_t c = 0; c <= CHAINS; c++) {
            cs[CHAINS + 1 + c] = t;
            if (c < CHAINS) t += (cs[c + 1] - cs[c] + tile - 1) / tile;
        }
        if (p == 0) {
            uint32_t* c0s = chain;
            for (uint32_t c = 0; c <= CHAINS; c++) {
                const uint64_t v = (uint64_t)c * per0;
                c0s[c] = (uint32_t)(v < n ? v : n);
            }
            t = 0;
            for (uint32_t c = 0; c <= CHAINS; c++) {
                c0s[CHAINS + 1 + c] = t;
                if (c < CHAINS) t += (c0s[c + 1] - c0s[c] + tile - 1) / tile;
            }
        }
    }
}

// ---- one pass --------------------------------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <class K, class V, class In, int THREADS, int IPT>
__global__ __launch_bounds__(THREADS) void k_rs_pass(In in, K* __restrict__ ok, V* __restrict__ ov, uint32_t bit, uint32_t nbits, const uint32_t* __restrict__ chain_start,
                                                     const uint32_t* __restrict__ tile_first, const uint32_t* __restrict__ base, uint32_t* __restrict__ heads,
                                                     uint32_t* __restrict__ states, uint32_t states_bytes) {
    constexpr bool HASV = !std::is_same<V, NoVal>::value;
    constexpr int WAVES = THREADS / 64, TILE = THREADS * IPT;
    constexpr int DT = THREADS < DIGITS ? THREADS : DIGITS, DPT = DIGITS / DT;  // the threads that own DPT digits each (scan of the tile's counts)
    constexpr int LBT = DIGITS / 4, LB = 4;                                     // look-back threads (four digits each), tiles fetched per round
    static_assert(THREADS >= LBT && THREADS % 64 == 0, "workgroup too small");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    K* lk = reinterpret_cast<K*>(smem);
    constexpr size_t OFF_V = align16(sizeof(K) * (size_t)TILE);
    V* lv = reinterpret_cast<V*>(smem + OFF_V);
    constexpr size_t OFF_C = OFF_V + (HASV ? align16(sizeof(V) * (size_t)TILE) : 0);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem + OFF_C);  // [WAVES][DIGITS]
    uint32_t* tstart = cnt + WAVES * DIGITS;                    // [DIGITS] first slot of a digit in the reordered tile
    uint32_t* tcnt = tstart + DIGITS;                           // [DIGITS] keys of a digit in the tile
    uint32_t* gpos = tcnt + DIGITS;                             // [DIGITS] global position of slot 0 as seen from a digit: out = gpos[d] + slot
    uint32_t* wsum = gpos + DIGITS;                             // [16]
    uint32_t* shd = wsum + 16;                                  // [4] the tile claimed next: chain, number

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t mask = (1u << nbits) - 1u;
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(states, 0, (int)states_bytes, 0x00020000);

    uint32_t home = xcc_id();
    auto claim = [&](uint32_t& cc, uint32_t& jj) -> bool {  // (thread 0)
        for (int t = 0; t < CHAINS; t++) {
            const uint32_t c2 = (home + t) & (CHAINS - 1);
            const uint32_t nt = tile_first[c2 + 1] - tile_first[c2];
            if (nt == 0) continue;
            const uint32_t j2 = atomicAdd(&heads[c2], 1u);
            if (j2 < nt) { cc = c2; jj = j2; home = c2; return true; }
        }
        return false;
    };
    K key[IPT];
    V val[IPT];
    auto load_tile = [&](uint32_t c, uint32_t j, uint32_t& tn) {
        const uint32_t a0 = chain_start[c] + j * (uint32_t)TILE, rem = chain_start[c + 1] - a0;
        tn = rem < (uint32_t)TILE ? rem : (uint32_t)TILE;
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t idx = wave * 64u * IPT + r * 64u + lane;
            if (idx < tn) {
                key[r] = in.key(a0 + idx);
                if constexpr (HASV) val[r] = in.val(a0 + idx);
            } else
                key[r] = K(0);
        }
    };

    if (tid == 0) {
        uint32_t cc = 0xFFFFFFFFu, jj = 0;
        if (!claim(cc, jj)) cc = 0xFFFFFFFFu;
        shd[0] = cc;
        shd[1] = jj;
    }
    __syncthreads();
    uint32_t cur_c = shd[0], cur_j = shd[1], tile_n = 0;
    if (cur_c == 0xFFFFFFFFu) return;
    load_tile(cur_c, cur_j, tile_n);
    __syncthreads();  // (shd is rewritten below)

    for (;;) {
        // the tile after this one: claimed now, used once this tile's keys sit in LDS
        if (tid == 0) {
            uint32_t cc = 0xFFFFFFFFu, jj = 0;
            if (!claim(cc, jj)) cc = 0xFFFFFFFFu;
            shd[0] = cc;
            shd[1] = jj;
        }
        // ---- rank: wave w owns the tile's keys [w 64 IPT, (w + 1) 64 IPT), key (round r, lane l) = r 64 + l of them
#pragma unroll
        for (int q = 0; q < DIGITS / 64; q++) cnt[wave * DIGITS + q * 64 + lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t rank[IPT];
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t idx = wave * 64u * IPT + r * 64u + lane;
            const bool valid = idx < tile_n;
            const uint32_t d = digit_of(key[r], bit, mask);
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < DBITS; b++) {
                if (b < (int)nbits) {
                    const bool on = (d >> b) & 1u;
                    const uint64_t bj = __ballot(on);
                    peers &= on ? bj : ~bj;
                }
            }
            const int leader = valid ? __builtin_ctzll(peers) : (int)lane;
            uint32_t b0 = 0;
            if (valid && (int)lane == leader) {
                b0 = cnt[wave * DIGITS + d];
                cnt[wave * DIGITS + d] = b0 + (uint32_t)__builtin_popcountll(peers);
            }
            b0 = __shfl(b0, leader);
            rank[r] = b0 + (uint32_t)__builtin_popcountll(peers & lt_mask);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        __syncthreads();  // B
        const uint32_t nxt_c = shd[0], nxt_j = shd[1];
        // ---- per digit: counts of the waves -> starts of (digit, wave) relative to the digit; the digit's total
        uint32_t s[DPT], tot = 0, inc = 0;
        if (tid < DT) {
#pragma unroll
            for (int q = 0; q < DPT; q++) {
                const uint32_t d = tid * DPT + q;
                uint32_t run = 0;
#pragma unroll
                for (int w = 0; w < WAVES; w++) {
                    const uint32_t c = cnt[w * DIGITS + d];
                    cnt[w * DIGITS + d] = run;
                    run += c;
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
        __syncthreads();  // D
        if (tid < DT) {
            uint32_t before = 0;
#pragma unroll
            for (int w = 0; w < DT / 64; w++)
                if (w < (int)wave) before += wsum[w];
            uint32_t start = before + inc - tot;
#pragma unroll
            for (int q = 0; q < DPT; q++) {
                tstart[tid * DPT + q] = start;
                tcnt[tid * DPT + q] = s[q];
                start += s[q];
            }
        }
        __syncthreads();  // E
        // ---- publish the tile's counts, start looking back
        const uint32_t tile_g = tile_first[cur_c] + cur_j;
        uint32_t my[4] = {0, 0, 0, 0}, ex[4] = {0, 0, 0, 0};
        u32x4 xb[LB];
        if (tid < LBT) {
            const uint32_t fl = cur_j == 0 ? ST_INC : ST_AGG;
#pragma unroll
            for (int q = 0; q < 4; q++) my[q] = tcnt[tid * 4 + q];
            u32x4 a;
            a.x = fl | my[0]; a.y = fl | my[1]; a.z = fl | my[2]; a.w = fl | my[3];
            __builtin_amdgcn_raw_buffer_store_b128(a, srsrc, (int)((tile_g * (uint32_t)DIGITS + tid * 4u) * 4u), 0, 16);
            if (cur_j > 0) {
#pragma unroll
                for (int i = 0; i < LB; i++) {
                    const uint32_t pj = cur_j - 1 >= (uint32_t)i ? cur_j - 1 - i : 0u;
                    xb[i] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, (int)(((tile_g - cur_j + pj) * (uint32_t)DIGITS + tid * 4u) * 4u), 0, 16);
                }
            }
        }
        // ---- reorder in LDS
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t idx = wave * 64u * IPT + r * 64u + lane;
            if (idx < tile_n) {
                const uint32_t d = digit_of(key[r], bit, mask);
                const uint32_t slot = tstart[d] + cnt[wave * DIGITS + d] + rank[r];
                lk[slot] = key[r];
                if constexpr (HASV) lv[slot] = val[r];
            }
        }
        // ---- the next tile's keys: into the registers this tile's keys have just left
        const uint32_t this_n = tile_n;
        if (nxt_c != 0xFFFFFFFFu) load_tile(nxt_c, nxt_j, tile_n);
        // ---- finish the look-back
        if (tid < LBT) {
            if (cur_j > 0) {
                uint32_t done = 0;
                uint32_t pj = cur_j - 1;  // the tile xb[0] stands for
                for (;;) {
                    bool stall = false;
#pragma unroll
                    for (int i = 0; i < LB; i++) {
                        if (done == 15u || stall) break;
                        const u32x4 x = xb[i];
                        const uint32_t xv[4] = {x.x, x.y, x.z, x.w};
                        bool wait = false;
#pragma unroll
                        for (int q = 0; q < 4; q++) wait |= !((done >> q) & 1u) && (xv[q] >> 30) == 0u;
                        if (wait) { stall = true; break; }
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (!((done >> q) & 1u)) {
                                ex[q] += xv[q] & ST_VAL;
                                if ((xv[q] >> 30) == 2u) done |= 1u << q;
                            }
                        pj--;  // (tile 0 of a chain is published inclusive: `done` is complete before pj wraps)
                    }
                    if (done == 15u) break;
                    if (stall) __builtin_amdgcn_s_sleep(4);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < LB; i++) {
                        const uint32_t pi = pj >= (uint32_t)i ? pj - i : 0u;
                        xb[i] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, (int)(((tile_g - cur_j + pi) * (uint32_t)DIGITS + tid * 4u) * 4u), 0, 16);
                    }
                }
                u32x4 a;
                a.x = ST_INC | (ex[0] + my[0]); a.y = ST_INC | (ex[1] + my[1]); a.z = ST_INC | (ex[2] + my[2]); a.w = ST_INC | (ex[3] + my[3]);
                __builtin_amdgcn_raw_buffer_store_b128(a, srsrc, (int)((tile_g * (uint32_t)DIGITS + tid * 4u) * 4u), 0, 16);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) gpos[tid * 4 + q] = base[cur_c * DIGITS + tid * 4 + q] + ex[q] - tstart[tid * 4 + q];
        }
        __syncthreads();  // G
        // ---- write out: slot by slot, i.e. digit run by digit run
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const uint32_t slot = i * THREADS + tid;
            if (slot < this_n) {
                const K kk = lk[slot];
                const uint32_t pos = gpos[digit_of(kk, bit, mask)] + slot;
                ok[pos] = kk;
                if constexpr (HASV) ov[pos] = lv[slot];
            }
        }
        if (nxt_c == 0xFFFFFFFFu) break;
        cur_c = nxt_c;
        cur_j = nxt_j;
        // (no barrier: the next round writes shd before B -- every thread has read it behind B of this round --, the counters before B, everything else behind E)
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
    return align16(sizeof(K) * (size_t)THREADS * IPT) + (std::is_same<V, NoVal>::value ? 0 : align16(sizeof(V) * (size_t)THREADS * IPT)) +
           ((size_t)(THREADS / 64) * DIGITS + 3 * DIGITS + 16 + 4) * 4;
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

// Stable sort of n (key, value) pairs on the key bits [begin_bit, end_bit).  The input is read through `in` (once by the histogram kernel,
// once by the first pass); the result lands in out_k / out_v; tmp_k / tmp_v (n entries each) carry the passes in between and may be NULL
// for a one-pass sort.  `in` may read out_k / out_v only when the number of passes is even.  V = NoVal: keys only.
template <class K, class V, class In, int THREADS, int IPT>
int sort_cfg(In in, uint64_t n, K* out_k, V* out_v, K* tmp_k, V* tmp_v, unsigned begin_bit, unsigned end_bit, hipStream_t s, DevBuf& scratch) {
    if (n >= (1ull << 30)) return bft_fail(BFT_GPU_E_LIMIT, "internal: radix sort of 2^30 entries or more");
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
    const int cus = cu_count();
    const uint32_t tiles = (uint32_t)((n + TILE - 1) / TILE);
    const uint32_t per0 = (tiles + CHAINS - 1) / CHAINS * TILE;
    // histogram workgroups: enough to stream (1024 threads x 8 keys a turn), no more than the input has turns
    constexpr int HT = 1024;
    uint32_t wpc = (uint32_t)std::max(1, cus / CHAINS);
    while (wpc > 1 && (uint64_t)CHAINS * wpc * HT * 8 > n * 2) wpc >>= 1;
    const Layout L = make_layout(n, pl.P, TILE, CHAINS * wpc);
    if (scratch.bytes < L.total_words * 4) CK(scratch.alloc(L.total_words * 4));
    uint32_t* W = scratch.as<uint32_t>();
    HIPCK(hipMemsetAsync(W, 0, L.zero_words * 4, s));
    const size_t hist_lds = (size_t)pl.P * CHAINS * DIGITS * 4;
    {
        static bool attr_h = false;
        if (!attr_h) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_hist<K, In, HT>), hipFuncAttributeMaxDynamicSharedMemorySize, MAXP * CHAINS * DIGITS * 4); attr_h = true; }
    }
    hipLaunchKernelGGL((k_rs_hist<K, In, HT>), dim3(CHAINS * wpc), dim3(HT), hist_lds, s, in, (uint32_t)n, pl, per0, wpc, W + L.partial);
    const uint32_t words = (uint32_t)pl.P * CHAINS * DIGITS;
    hipLaunchKernelGGL(k_rs_reduce, dim3((words + 255) / 256), dim3(256), 0, s, W + L.partial, CHAINS * wpc, words, W + L.cnt);
    hipLaunchKernelGGL(k_rs_scan, dim3(pl.P), dim3(DIGITS), 0, s, W + L.cnt, W + L.base, W + L.chain, pl, (uint32_t)n, per0, TILE);
    HIPCK(hipGetLastError());
    constexpr size_t lds = pass_lds_bytes<K, V, THREADS, IPT>();
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds, 2048 / THREADS));
    const uint32_t grid = std::min<uint32_t>(tiles + CHAINS, (uint32_t)(cus * per_cu));
    const K* src_k = nullptr;
    const V* src_v = nullptr;
    for (int p = 0; p < pl.P; p++) {
        const bool to_out = ((pl.P - 1 - p) % 2) == 0;
        K* dk = to_out ? out_k : tmp_k;
        V* dv = to_out ? out_v : tmp_v;
        const uint32_t* ch = W + L.chain + (size_t)p * 2 * (CHAINS + 1);
        uint32_t* st = W + L.states + (size_t)p * L.max_tiles * DIGITS;
        const uint32_t st_bytes = L.max_tiles * DIGITS * 4;
        if (p == 0) {
            static bool attr0 = false;
            if (!attr0) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_pass<K, V, In, THREADS, IPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr0 = true; }
            hipLaunchKernelGGL((k_rs_pass<K, V, In, THREADS, IPT>), dim3(grid), dim3(THREADS), lds, s, in, dk, dv, pl.bit[p], pl.nbits[p], ch, ch + CHAINS + 1,
                               W + L.base + (size_t)p * CHAINS * DIGITS, W + L.heads + (size_t)p * CHAINS, st, st_bytes);
        } else {
            static bool attr1 = false;
            if (!attr1) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_pass<K, V, PtrIn<K, V>, THREADS, IPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr1 = true; }
            hipLaunchKernelGGL((k_rs_pass<K, V, PtrIn<K, V>, THREADS, IPT>), dim3(grid), dim3(THREADS), lds, s, PtrIn<K, V>{src_k, src_v}, dk, dv, pl.bit[p], pl.nbits[p], ch,
                               ch + CHAINS + 1, W + L.base + (size_t)p * CHAINS * DIGITS, W + L.heads + (size_t)p * CHAINS, st, st_bytes);
        }
        src_k = dk;
        src_v = dv;
    }
    HIPCK(hipGetLastError());
    return 0;
}

// the tile shape by the bytes of an entry: large arrays stream through 1024-thread workgroups, one per CU, with the largest tile the LDS
// holds (a digit's piece of a tile is TILE / 512 entries: the longer, the fewer partial lines); small arrays take 256-thread workgroups so
// that there are tiles for every CU
template <class K, class V, class In>
int sort(In in, uint64_t n, K* out_k, V* out_v, K* tmp_k, V* tmp_v, unsigned begin_bit, unsigned end_bit, hipStream_t s, DevBuf& scratch) {
    constexpr size_t E = sizeof(K) + (std::is_same<V, NoVal>::value ? 0 : sizeof(V));
    constexpr int IPT_BIG = E <= 8 ? 12 : E <= 12 ? 8 : E <= 16 ? 6 : E <= 24 ? 4 : E <= 32 ? 3 : 2;
    constexpr int IPT_SMALL = E <= 8 ? 16 : E <= 16 ? 8 : E <= 32 ? 4 : 2;
    if (n >= (1u << 22)) return sort_cfg<K, V, In, 1024, IPT_BIG>(in, n, out_k, out_v, tmp_k, tmp_v, begin_bit, end_bit, s, scratch);
    return sort_cfg<K, V, In, 256, IPT_SMALL>(in, n, out_k, out_v, tmp_k, tmp_v, begin_bit, end_bit, s, scratch);
}

}  // namespace bft_rs
