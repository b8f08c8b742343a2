// bft_kh.hip -- the k-mer hash (bft_image.h, BFT_KH_*): its build on the GPU, its dump ("compact_table"), and the queries of
// include/bft_gpu.h through it: presence / colour set, branching, sequence positions.  Its own translation unit: these kernels stage
// nothing and walk nothing -- T-form, region of the root prefix, home line, compare.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstring>

#include "bft_dev.h"
#include "bft_image.h"
#include "bft_kh.h"
#include "bft_claims.h"
#include "bft_walk.h"
#include "bft_kernels_load.h"
#include "bft_kernels_seqwin.h"
#include "bft_kh_dev.h"
#include "bft_scan.h"
#include "bft_rows16.h"
#include "bft_sort.h"

// ---------------------------------------------------------------------------------------------------------------------------------
// queries
// ---------------------------------------------------------------------------------------------------------------------------------
// Batched isKmerPresent (src/presenceNode.c:1823-1921; loop of src/file_io.c:726-768): bit i = k-mer i is stored; out32 (optional):
// its colour-set id (what get_annotation locates, src/bft.c:363-387), 0xFFFFFFFF when absent.  One lane per k-mer, 64 presence bits
// per wavefront through __ballot, persistent grid.
// (Two or four k-mers per lane and pass, their home lines loaded together, were measured on the 100-genome index: 41.5 / 37.5 G k-mers/s
// against 44.6 with one -- the fabric's request rate is the limit, not the number of requests a lane keeps in flight; tools/probe_kh.py.)
// The blocks of 256 k-mers are dealt out in rounds of `chunk` blocks: the first round by workgroup number, the others claimed from a counter
// in device memory (bft_claims.h; ctr == NULL: every round by workgroup number).
// The presence words of a round (one per wavefront and block) are gathered in LDS and leave as ONE coalesced store of whole lines: a round
// of four blocks is exactly one 128-byte line of the bitmap.  Stored wavefront by wavefront, a line was written 8 bytes at a time over
// ~40 us while the gathers turn the L2 over every ~10 us -- evicted in pieces --, and the launch time followed where the 15 MB bitmap lay
// (2.60 / 2.82 ms on the same table and batch, fresh bitmaps flipping it inside a process: profiles/r03/probe_dynamic_inputs.jsonl).
#define BFT_KH_MIN_CLAIM 4u
#define BFT_KH_MAX_CLAIM 64u
typedef BftClaims<BFT_KH_MIN_CLAIM> KhClaims;
// (Round 4 also measured four k-mers per lane answered stage by stage -- four packed k-mers, then four regions, then four home lines in
// flight together: 28.8 G k-mers/s against 31.5 with one per lane on the table's first form.  What bound that kernel was not the length of
// its chain of dependent loads but the number of vector memory instructions per k-mer: see bft_kh_scan.)
// What the home line costs (round 4, measured on the config-4 share, profiles/r04/kh_forms.jsonl):
//   * all 64 bytes by four 16-byte loads per lane: 37 G k-mers/s -- a vector memory instruction of 64 lanes that touch 64 different lines
//     occupies the CU's address path for its 64 tag lookups, and six such instructions per k-mer (packed k-mer, region, line) bound the kernel;
//   * the 16-byte header first, the body of the slot that matches second (bft_kh_scan): 40 G -- fewer instructions, but the body comes
//     ~5 us after the header (returns are in order behind other wavefronts' misses) and the L2 of an XCD turns over in that time: four body
//     loads in ten fetch the line from HBM again (1.18 -> 1.41 L2 misses per k-mer, 54 G misses/s: the fabric's ceiling);
//   * software-pipelined blocks (the next block's region and the block after's packed k-mers in flight beside the header): 38.7 G -- the
//     chain of dependent loads is not what binds.
// So the line is fetched by the QUAD: the four lanes of a quad load 16 bytes each of ONE lane's home line -- one instruction brings
// sixteen whole lines per wavefront instead of a quarter of sixty-four, so four instructions bring the 64 lines with a quarter of the
// tag lookups each, every line is requested once and whole, and nothing is read twice.  The pieces go to the wavefront's LDS, where the
// owner lane reads its header and the body of the slot that matches (bft_kh_dev.h; gathered into registers by DPP and scanned there the
// kernel was bound by its own instructions: 48 G k-mers/s against 52.5).
template <int W, int S>
__global__ __launch_bounds__(BFT_KH_BLOCK) void k_query_kh(BftImage im, const uint8_t* __restrict__ packed, uint64_t n, int B, uint64_t* __restrict__ bits64,
                                                          uint32_t* __restrict__ out32, BftClaimCtr ctr, uint32_t chunk) {
    const uint64_t end_aligned = ((uint64_t)packed + n * (uint64_t)B) & ~3ull;
    const uint64_t nblk = (n + BFT_KH_BLOCK - 1) / BFT_KH_BLOCK, nwords = (n + 63) / 64;
    constexpr uint32_t WPB = BFT_KH_BLOCK / 64;  // presence words per block
    __shared__ uint32_t s_next[2];
    __shared__ uint64_t s_bits[BFT_KH_MAX_CLAIM * WPB];
    __shared__ uint4 s_lines[WPB][64 * BFT_KH_LDS_LINE];
    uint4* const wave_lines = s_lines[threadIdx.x >> 6];
    const uint4* const mine = wave_lines + (threadIdx.x & 63u) * BFT_KH_LDS_LINE;
    KhClaims cl(ctr, chunk, nblk, s_next);
    cl.first();
    while (cl.blk < nblk) {
        const uint64_t i = cl.blk * BFT_KH_BLOCK + threadIdx.x;
        const bool live = i < n;
        BftKhKey<W> key;
        key.home = 0;
        key.field = 0;
#pragma unroll
        for (int w = 0; w < W; w++) { key.body[w] = 0; key.bmask[w] = 0; }
        uint64_t t[W];
#pragma unroll
        for (int w = 0; w < W; w++) t[w] = 0;
        if (live) {
            uint64_t x[W];
            load_x<W>(packed, i, B, end_aligned, x);
            bft_tform_from_x<W>(x, im.k, t);
            bft_kh_key<W>(t, im.k, im.kh, key);
        }
        kh_fetch_quad(im, key.home, live, wave_lines);
        bool present = false;
        uint32_t val = 0xFFFFFFFFu;
        if (live) {
            int res = kh_lds_scan<W, S>(im, mine, key, 0u, &val);
            for (uint32_t d = 1; res < 0 && d <= im.kh.maxd; d++) {  // full line without the key: on from the home line, on this lane's own (a few per cent)
                const uint64_t* line = im.kh_lines + (key.home + d) * BFT_KH_LINE_WORDS;
                uint64_t hd[2];
                bft_kh_load_header(line, hd);
                res = bft_kh_scan<W, S>(im, line, hd, key, d, &val);
            }
            if (res < 0 && im.kh_ovf_n) res = bft_kh_overflow_find<W>(im, t, &val) ? 1 : 0;  // (a run of full lines as long as any displacement: the overflow list)
            present = res > 0;
        }
        const uint64_t mask = __ballot(present);
        if (out32 && i < n) out32[i] = present ? val : 0xFFFFFFFFu;
        if ((threadIdx.x & 63u) == 0) s_bits[(cl.blk - cl.start) * WPB + (threadIdx.x >> 6)] = mask;
        if (cl.last_of_round()) {  // the round is answered: its words leave together
            __syncthreads();
            const uint64_t w0 = cl.start * WPB, w1 = min(nwords, cl.blk_end * WPB);
            if (w0 + threadIdx.x < w1) __builtin_nontemporal_store(s_bits[threadIdx.x], &bits64[w0 + threadIdx.x]);
        }
        cl.advance();  // (its barriers stand between these reads of s_bits and the next round's writes)
    }
    cl.done();
}

// Colour rows of a resident batch (get_annotation + the bitmap form of get_list_id_genomes, src/bft.c:363-387, 622-641) in ONE launch: the
// tiles belong to wavefronts as in k_color_rows_bm16 (bft_kernels_color.h), but the wavefront looks its tile's k-mers up itself -- 64 at a
// time, the lookup of k_query_kh -- and keeps their dictionary rows in its LDS slice; nothing but the presence words and the rows is
// written.  Round 5 ran the lookup as its own launch, writing a colour-set id per k-mer that the row kernel read back.
// tile_rows: a multiple of 64 (a presence word per lookup round; tiles start 16-byte aligned), at most BFT_KH_ROWS_TILE.
#ifndef BFT_KH_ROWS_TILE
#define BFT_KH_ROWS_TILE 256u
#endif
template <int W, int S>
__global__ __launch_bounds__(256) void k_color_rows_kh(BftImage im, const uint8_t* __restrict__ packed, uint64_t n, int B, uint64_t* __restrict__ bits64,
                                                       const uint8_t* __restrict__ bm, uint32_t stride, uint32_t rowbytes, uint32_t tile_rows, uint32_t div_m, uint32_t div_l,
                                                       uint8_t* __restrict__ out) {
    __shared__ uint4 s_lines[4][64 * BFT_KH_LDS_LINE];
    __shared__ uint32_t s_cs_all[4][BFT_KH_ROWS_TILE + 1];
    const uint64_t end_aligned = ((uint64_t)packed + n * (uint64_t)B) & ~3ull;
    const uint64_t ntiles = (n + tile_rows - 1) / tile_rows;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint4* const wave_lines = s_lines[wave];
    const uint4* const mine = wave_lines + lane * BFT_KH_LDS_LINE;
    uint32_t* const s_cs = s_cs_all[wave];
    for (uint64_t tile = (uint64_t)blockIdx.x * 4u + wave; tile < ntiles; tile += (uint64_t)gridDim.x * 4u) {
        const uint64_t q0 = tile * tile_rows;
        const uint32_t nt = (uint32_t)min((uint64_t)tile_rows, n - q0);
        __builtin_amdgcn_wave_barrier();  // (the last tile's reads of s_cs come before these writes: same wavefront, in order)
#pragma unroll 1
        for (uint32_t j0 = 0; j0 < nt; j0 += 64u) {
            const uint64_t i = q0 + j0 + lane;
            const bool live = i < n;
            BftKhKey<W> key;
            key.home = 0;
            key.field = 0;
#pragma unroll
            for (int w = 0; w < W; w++) { key.body[w] = 0; key.bmask[w] = 0; }
            uint64_t t[W];
#pragma unroll
            for (int w = 0; w < W; w++) t[w] = 0;
            if (live) {
                uint64_t x[W];
                load_x<W>(packed, i, B, end_aligned, x);
                bft_tform_from_x<W>(x, im.k, t);
                bft_kh_key<W>(t, im.k, im.kh, key);
            }
            kh_fetch_quad(im, key.home, live, wave_lines);
            bool present = false;
            uint32_t val = 0;
            if (live) {
                int res = kh_lds_scan<W, S>(im, mine, key, 0u, &val);
                for (uint32_t d = 1; res < 0 && d <= im.kh.maxd; d++) {
                    const uint64_t* line = im.kh_lines + (key.home + d) * BFT_KH_LINE_WORDS;
                    uint64_t hd[2];
                    bft_kh_load_header(line, hd);
                    res = bft_kh_scan<W, S>(im, line, hd, key, d, &val);
                }
                if (res < 0 && im.kh_ovf_n) res = bft_kh_overflow_find<W>(im, t, &val) ? 1 : 0;
                present = res > 0;
            }
            s_cs[j0 + lane] = present ? val * (stride >> 2) : CR16_ABSENT;
            const uint64_t mask = __ballot(present);
            if (lane == 0) bits64[(q0 + j0) >> 6] = mask;
        }
        if (lane == 0) s_cs[nt] = CR16_ABSENT;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        cr16_stream_tile(s_cs, nt, rowbytes, div_m, div_l, bm, out + q0 * rowbytes, lane);
    }
}

// get_annotation + get_list_id_genomes of a resident batch (src/bft.c:363-387, 622-641; src/annotation.c:2086-2250) in ONE launch: lookup, offsets and
// ids.  A workgroup claims tiles of CT x 256 k-mers in order; per block of the tile: the lookup of k_query_kh (quad fetch into LDS, scan of the
// line) hands every lane its colour set, whose list length comes out of the dictionary's offsets; the tile's total is published as one
// 8-byte word and the first wavefront looks back over the 64 tiles before it for the tile's 64-bit offset (the scan of bft_scan.h, inside
// this kernel); then every wavefront writes its k-mers' offsets and streams their ids out coalesced (every output element finds its k-mer by
// six shuffles over the lanes' starts, as k_color_fill_cs did).  Round 5 ran three launches -- lookup writing a colour-set id per k-mer,
// a library scan reading it back, the fill reading both: 24.8 G k-mers/s on the config-4 index.
// scratch: [0] the tile counter, [1 ..] the tiles' states {flag:2, value:62}; zeroed before the launch.  ids beyond ids_cap are not written
// (the caller compares *needed = offsets[n] with its capacity).
#ifndef BFT_KH_CT
#define BFT_KH_CT 4
#endif
#ifndef BFT_KH_LEN_T
#define BFT_KH_LEN_T uint16_t
#endif
template <int W, int S>
__global__ __launch_bounds__(BFT_KH_BLOCK) void k_colors_kh(BftImage im, const uint8_t* __restrict__ packed, uint64_t n, int B, uint64_t* __restrict__ bits64,
                                                           unsigned long long* __restrict__ offsets, uint32_t* __restrict__ ids, uint64_t ids_cap, unsigned long long* __restrict__ needed,
                                                           unsigned long long* __restrict__ scratch) {
    constexpr uint32_t WPB = BFT_KH_BLOCK / 64, CT = BFT_KH_CT;
    constexpr unsigned long long F_AGG = 1ull << 62, F_INC = 2ull << 62, VMASK = (1ull << 62) - 1ull;
    __shared__ uint4 s_lines[WPB][64 * BFT_KH_LDS_LINE];
    __shared__ uint64_t s_bits[CT * WPB];
    __shared__ uint32_t s_sum[CT * WPB];
    __shared__ BFT_KH_LEN_T s_len[CT * BFT_KH_BLOCK];  // (16 bits: the launcher sends indexes of 2^16 genomes and more the three-launch way; with 32 the LDS
    __shared__ uint32_t s_src[CT * BFT_KH_BLOCK];  // holds five workgroups per CU instead of six: 2.04 -> 1.97 ms on config 4)  // a k-mer's list: ids, where they start in the dictionary (in LDS, not registers:
    __shared__ uint32_t s_tile;                                              // the lookups need the wavefronts the registers would cost)
    __shared__ unsigned long long s_prefix;
    const uint64_t end_aligned = ((uint64_t)packed + n * (uint64_t)B) & ~3ull;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint4* const wave_lines = s_lines[wave];
    const uint4* const mine = wave_lines + lane * BFT_KH_LDS_LINE;
    const uint64_t ntiles = (n + (uint64_t)CT * BFT_KH_BLOCK - 1) / ((uint64_t)CT * BFT_KH_BLOCK), nwords = (n + 63) / 64;
    unsigned long long* states = scratch + 1;
    for (;;) {
        if (threadIdx.x == 0) s_tile = (uint32_t)atomicAdd(scratch, 1ull);
        __syncthreads();
        const uint64_t tile = s_tile;
        if (tile >= ntiles) return;
#pragma unroll 1
        for (uint32_t c = 0; c < CT; c++) {
            const uint64_t i = (tile * CT + c) * BFT_KH_BLOCK + threadIdx.x;
            const bool live = i < n;
            BftKhKey<W> key;
            key.home = 0;
            key.field = 0;
#pragma unroll
            for (int w = 0; w < W; w++) { key.body[w] = 0; key.bmask[w] = 0; }
            uint64_t t[W];
#pragma unroll
            for (int w = 0; w < W; w++) t[w] = 0;
            if (live) {
                uint64_t x[W];
                load_x<W>(packed, i, B, end_aligned, x);
                bft_tform_from_x<W>(x, im.k, t);
                bft_kh_key<W>(t, im.k, im.kh, key);
            }
            kh_fetch_quad(im, key.home, live, wave_lines);
            bool present = false;
            uint32_t val = 0;
            if (live) {
                int res = kh_lds_scan<W, S>(im, mine, key, 0u, &val);
                for (uint32_t d = 1; res < 0 && d <= im.kh.maxd; d++) {
                    const uint64_t* line = im.kh_lines + (key.home + d) * BFT_KH_LINE_WORDS;
                    uint64_t hd[2];
                    bft_kh_load_header(line, hd);
                    res = bft_kh_scan<W, S>(im, line, hd, key, d, &val);
                }
                if (res < 0 && im.kh_ovf_n) res = bft_kh_overflow_find<W>(im, t, &val) ? 1 : 0;
                present = res > 0;
            }
            // (the colour set for now -- NONE when absent --: the dictionary is asked about all CT blocks together below, CT loads in flight per lane
            // instead of one behind every lookup)
            s_src[c * BFT_KH_BLOCK + threadIdx.x] = present ? val : 0xFFFFFFFFu;
            const uint64_t mask = __ballot(present);
            if (lane == 0) s_bits[c * WPB + wave] = mask;
        }
        {
            uint32_t cs[CT], o0[CT], o1[CT];
#pragma unroll
            for (uint32_t c = 0; c < CT; c++) cs[c] = s_src[c * BFT_KH_BLOCK + threadIdx.x];  // (this thread's own words: no barrier)
#pragma unroll
            for (uint32_t c = 0; c < CT; c++) {
                o0[c] = 0; o1[c] = 0;
                if (cs[c] != 0xFFFFFFFFu) { o0[c] = im.cs_off[cs[c]]; o1[c] = im.cs_off[cs[c] + 1]; }
            }
#pragma unroll
            for (uint32_t c = 0; c < CT; c++) {
                const uint32_t len = o1[c] - o0[c];
                s_len[c * BFT_KH_BLOCK + threadIdx.x] = (BFT_KH_LEN_T)len;
                s_src[c * BFT_KH_BLOCK + threadIdx.x] = o0[c];
                uint32_t ws = len;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) ws += __shfl_down(ws, o);
                if (lane == 0) s_sum[c * WPB + wave] = ws;
            }
        }
        __syncthreads();
        {   // the presence words of the tile leave together; the tile's total; its place among all ids
            const uint64_t w0 = tile * CT * WPB;
            if (threadIdx.x < CT * WPB && w0 + threadIdx.x < nwords) __builtin_nontemporal_store(s_bits[threadIdx.x], &bits64[w0 + threadIdx.x]);
        }
        if (wave == 0) {
            unsigned long long total = lane < CT * WPB ? s_sum[lane] : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o);
            total = __shfl(total, 0);
            unsigned long long ex = 0;
            if (tile == 0) {
                if (lane == 0) __hip_atomic_store(&states[0], F_INC | (total & VMASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (lane == 0) __hip_atomic_store(&states[tile], F_AGG | (total & VMASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                long long first = (long long)tile - 1;
                for (;;) {
                    const long long tt = first - (long long)lane;
                    unsigned long long st = F_INC;
                    if (tt >= 0) {
                        do { st = __hip_atomic_load(&states[tt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((st >> 62) == 0ull && (__builtin_amdgcn_s_sleep(1), true));
                    }
                    const uint64_t incl = __ballot((st >> 62) == 2ull);
                    const int stop = incl ? __builtin_ctzll(incl) : 64;
                    unsigned long long x = ((int)lane <= stop && tt >= 0) ? (st & VMASK) : 0ull;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
                    ex += __shfl(x, 0);
                    if (stop < 64) break;
                    first -= 64;
                }
                if (lane == 0) __hip_atomic_store(&states[tile], F_INC | ((ex + total) & VMASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) {
                s_prefix = ex;
                if (tile == ntiles - 1) {  // (the last tile knows the total)
                    offsets[n] = ex + total;
                    if (needed) *needed = ex + total;
                }
            }
        }
        __syncthreads();
        unsigned long long run = s_prefix;
#pragma unroll 1
        for (uint32_t c = 0; c < CT; c++) {
            const uint32_t len = s_len[c * BFT_KH_BLOCK + threadIdx.x], src = s_src[c * BFT_KH_BLOCK + threadIdx.x];
            // the lane's start: the tile's, the (block, wavefront) sums before it, the lanes before it
            unsigned long long wbase = run;
#pragma unroll
            for (uint32_t w = 0; w < WPB; w++) {
                if (w < wave) wbase += s_sum[c * WPB + w];
                run += s_sum[c * WPB + w];
            }
            uint32_t inc = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(inc, o);
                if ((int)lane >= o) inc += y;
            }
            const uint32_t ar = inc - len, span = __shfl(inc, 63);  // start relative to the wavefront's first id; ids of the wavefront's 64 k-mers
            const uint64_t i = (tile * CT + c) * BFT_KH_BLOCK + threadIdx.x;
            if (i < n) offsets[i] = wbase + ar;
            if (!ids) continue;
            for (uint32_t r0 = 0; r0 < span; r0 += 64) {
                const uint32_t r = r0 + lane;
                const bool in = r < span;
                const uint32_t rr = in ? r : span - 1;
                uint32_t lo = 0, hi = 63;  // the last lane whose list starts at or before rr (starts are non-decreasing over the lanes)
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    const uint32_t mid = (lo + hi + 1) >> 1;
                    const uint32_t am = __shfl(ar, mid);
                    if (am <= rr) lo = mid; else hi = mid - 1;
                }
                const uint32_t as = __shfl(ar, lo);
                const uint32_t ss = __shfl(src, lo);
                if (in && wbase + r < ids_cap) ids[wbase + r] = bft_cs_id_at(im.cs_ids, im.cs_w, ss + (r - as));
            }
        }
        __syncthreads();  // (s_tile, s_sum, s_bits, s_prefix are rewritten by the next tile)
    }
}

// How many of the four successors (or the four predecessors) of a k-mer are stored.  They share their home line and differ in two stored key
// bits (bft_image.h: the bits that tell them apart are not hashed), so ONE line is fetched by the quad and looked through ONCE, the two bits
// left out of the comparison: every slot that matches is one of the four (src/presenceNode.c:15-1211 shares one descent between the four;
// here they share one line and one scan).  t0: the candidate whose wildcard nucleotide is 0; b: where the wildcard lies in the rest.  When the
// line is full the lines behind it follow on the lane's own (a few per cent), and a run of full lines as long as any displacement hands the
// four over to the overflow list.  Every lane of the wavefront calls this; `live` = this lane has candidates.
template <int W, int S>
__device__ __forceinline__ int kh_count4(const BftImage& im, const uint64_t* t0, uint32_t b, int wild_word, int wild_shift, bool live, uint4* wave_lines) {
    BftKhKey<W> key;
    key.home = 0; key.field = 0;
#pragma unroll
    for (int w = 0; w < W; w++) { key.body[w] = 0; key.bmask[w] = 0; }
    if (live) bft_kh_key<W>(t0, im.k, im.kh, key);
    kh_fetch_quad(im, key.home, live, wave_lines);
    if (!live) return 0;
    BftKhFamily<W> fam;
    bft_kh_family<W>(im.kh, key, b, fam);
    int count = 0;
    int res = kh_lds_count<W, S>(im, wave_lines + (threadIdx.x & 63u) * BFT_KH_LDS_LINE, key, fam, 0u, &count);
    for (uint32_t d = 1; res < 0 && count < 4 && d <= im.kh.maxd; d++) res = bft_kh_count_line<W, S>(im, im.kh_lines + (key.home + d) * BFT_KH_LINE_WORDS, key, fam, d, &count);
    if (res < 0 && count < 4 && im.kh_ovf_n) {
        for (int v = 0; v < 4; v++) {
            uint64_t c[W];
            uint32_t val;
#pragma unroll
            for (int w = 0; w < W; w++) c[w] = t0[w] | (w == wild_word ? (uint64_t)v << wild_shift : 0ull);
            count += bft_kh_overflow_find<W>(im, c, &val) ? 1 : 0;
        }
    }
    return count;
}

// Batched isBranchingRight / isBranchingLeft (src/branchingNode.c:16-112, :240-340; loop of src/file_io.c:943-998: the right side first,
// the left side only when the right one does not branch -- whether the k-mer itself is stored is never asked).  Successors / predecessors
// of a k-mer differ in 2 bits of the last / first T-form digit.  Rounds of blocks as in k_query_kh; the branching bits of a round leave
// LDS as whole lines.
#define BFT_KH_BR_MAX_CLAIM 16u
template <int W, int S>
__global__ __launch_bounds__(BFT_KH_BLOCK) void k_branching_kh(BftImage im, const uint8_t* __restrict__ packed, uint64_t n, int B, uint64_t* __restrict__ bits64,
                                                              uint8_t* __restrict__ counts, BftClaimCtr ctr, uint32_t chunk) {
    const uint64_t end_aligned = ((uint64_t)packed + n * (uint64_t)B) & ~3ull;
    const uint64_t nblk = (n + BFT_KH_BLOCK - 1) / BFT_KH_BLOCK, nwords = (n + 63) / 64;
    constexpr uint32_t WPB = BFT_KH_BLOCK / 64;
    const int k = im.k, L = im.L, rb = 2 * (k - 9 * L);
    __shared__ uint32_t s_next[2];
    __shared__ uint64_t s_bits[BFT_KH_BR_MAX_CLAIM * WPB];
    __shared__ uint4 s_lines[WPB][64 * BFT_KH_LDS_LINE];
    uint4* const wave_lines = s_lines[threadIdx.x >> 6];
    KhClaims cl(ctr, chunk, nblk, s_next);
    cl.first();
    while (cl.blk < nblk) {
        const uint64_t i = cl.blk * BFT_KH_BLOCK + threadIdx.x;
        const bool live = i < n;
        uint64_t x[W], y[W], t[W];
#pragma unroll
        for (int w = 0; w < W; w++) x[w] = 0;
        if (live) load_x<W>(packed, i, B, end_aligned, x);
        // successors: drop the first nucleotide, the last one is the wildcard (bits vo.. of the T-form's last word)
#pragma unroll
        for (int w = 0; w < W; w++) y[w] = (x[w] >> 2) | (w + 1 < W ? x[w + 1] << 62 : 0ull);
        bft_tform_from_x<W>(y, k, t);
        const int vo = rb ? 0 : 2;
        const int cr = kh_count4<W, S>(im, t, (uint32_t)vo, W - 1, vo, live, wave_lines);
        int cl_ = 0;
        const bool left = live && (counts || cr < 2);
        if (__any(left)) {  // (the quad fetch is the whole wavefront's)
            // predecessors: shift in a wildcard first nucleotide (bits 0..1 of the first digit), drop the last one
#pragma unroll
            for (int w = W - 1; w >= 0; w--) y[w] = (x[w] << 2) | (w > 0 ? x[w - 1] >> 62 : 0ull);
            const int top = 2 * k - 64 * (W - 1);
            if (top < 64) y[W - 1] &= (1ull << top) - 1ull;
            bft_tform_from_x<W>(y, k, t);
            const int o = rb + 18 * (L - 1), ow = W - 1 - (o >> 6), osh = o & 63;
            // (the first nucleotide's bits sit on top of the rest when they come out of the hashed bits, else where they are)
            cl_ = kh_count4<W, S>(im, t, im.kh.po < 32u ? im.kh.restb - 2u : (uint32_t)o, ow, osh, left, wave_lines);
        }
        const int branching = live && (cr > 1 || cl_ > 1);
        if (counts && live) counts[i] = (uint8_t)((cr << 4) | cl_);  // (a wavefront's 64 bytes: one coalesced store)
        const uint64_t mask = __ballot(branching);
        if ((threadIdx.x & 63u) == 0) s_bits[(cl.blk - cl.start) * WPB + (threadIdx.x >> 6)] = mask;
        if (cl.last_of_round()) {
            __syncthreads();
            const uint64_t w0 = cl.start * WPB, w1 = min(nwords, cl.blk_end * WPB);
            if (w0 + threadIdx.x < w1) __builtin_nontemporal_store(s_bits[threadIdx.x], &bits64[w0 + threadIdx.x]);
        }
        cl.advance();
    }
    cl.done();
}

// Sequence positions: the colour set of a position sits in the slot that says the k-mer is stored -- one line per position, fetched by the quad.
template <int W, int S>
__global__ __launch_bounds__(256) void k_seq_kh(BftImage im, const uint64_t* __restrict__ codes, const uint32_t* __restrict__ bad, const uint64_t* __restrict__ seq_off,
                                                const uint64_t* __restrict__ pos_off, const uint32_t* __restrict__ tile_seq, uint32_t n_seqs, int canonical,
                                                uint32_t* __restrict__ csout, BftClaimCtr ctr, uint32_t chunk) {
    const uint64_t P = pos_off[n_seqs];
    const uint64_t nblk = (P + 255) / 256;
    __shared__ uint32_t s_next[2];
    __shared__ uint4 s_lines[4][64 * BFT_KH_LDS_LINE];
    uint4* const wave_lines = s_lines[threadIdx.x >> 6];
    KhClaims cl(ctr, chunk, nblk, s_next);  // (the number of positions is only known on the device: the grid is the resident one, rounds beyond nblk are empty)
    cl.first();
    for (; cl.blk < nblk; cl.advance()) {
        const uint64_t p = cl.blk * 256 + threadIdx.x;
        uint32_t cs = 0xFFFFFFFFu;
        uint64_t x[W], t[W];
        BftKhKey<W> key;
        key.home = 0; key.field = 0;
#pragma unroll
        for (int w = 0; w < W; w++) { key.body[w] = 0; key.bmask[w] = 0; }
        bool live = false;
        if (p < P) {
            uint32_t lo = tile_seq[p >> 6];
            while (lo + 1 < n_seqs && pos_off[lo + 1] <= p) lo++;
            if (seq_window<W>(codes, bad, seq_off[lo] + (p - pos_off[lo]), im.k, canonical, x)) {
                bft_tform_from_x<W>(x, im.k, t);
                bft_kh_key<W>(t, im.k, im.kh, key);
                live = true;
            }
        }
        kh_fetch_quad(im, key.home, live, wave_lines);  // (the whole wavefront's: bft_kh_dev.h)
        if (live) {
            uint32_t val = 0;
            int res = kh_lds_scan<W, S>(im, wave_lines + (threadIdx.x & 63u) * BFT_KH_LDS_LINE, key, 0u, &val);
            for (uint32_t d = 1; res < 0 && d <= im.kh.maxd; d++) {
                const uint64_t* line = im.kh_lines + (key.home + d) * BFT_KH_LINE_WORDS;
                uint64_t hd[2];
                bft_kh_load_header(line, hd);
                res = bft_kh_scan<W, S>(im, line, hd, key, d, &val);
            }
            if (res < 0 && im.kh_ovf_n) res = bft_kh_overflow_find<W>(im, t, &val) ? 1 : 0;
            if (res > 0) cs = val;
        }
        if (p < P) csout[p] = cs;
    }
    cl.done();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// build (the canonical layout of bft_image.h: the k-mers in (home line, T-form) order, slot-level linear probing)
// ---------------------------------------------------------------------------------------------------------------------------------
// A k-mer on its way through the sort by home line: its T-form and its value travel with the key, so that the lines are assembled from
// a sequential read (gathering the rows by index afterwards costs two random lines per k-mer).
template <int W>
struct __attribute__((packed, aligned(4))) KhRec {  // (8 W + 4 bytes, no padding: the sort moves it three times)
    uint64_t t[W];
    uint32_t v;
};
// what the sort by home line reads: row i of the sorted table as (home line, {T-form, value}) -- computed on the fly, twice (the sort's
// histogram kernel and its first pass), instead of being written out and read back
template <int W>
struct KhSortIn {
    const uint64_t* tk;
    const uint32_t* vals;
    int k;
    BftKhGeo g;
    __device__ __forceinline__ uint32_t key(uint32_t i) const {
        uint64_t t[W];
        BftKhKey<W> kk;
        bft_load_row<W>(tk + (uint64_t)i * W, t);
        bft_kh_key<W>(t, k, g, kk);
        return (uint32_t)kk.home;
    }
    __device__ __forceinline__ KhRec<W> val(uint32_t i) const {
        KhRec<W> r;
        bft_load_row<W>(tk + (uint64_t)i * W, r.t);
        r.v = vals[i];
        return r;
    }
};
// Slot-level linear probing over the rows in sorted order: p_s = max(home slot_s, p_(s-1) + 1) = s + max_(j <= s)(home slot_j - j): one
// device-wide inclusive max-scan of (bias + home slot_j - j).
#define BFT_KH_SCAN_BIAS (1ull << 40)
struct KhScanIn {  // (the scan's input, from the sorted home lines)
    const uint32_t* key_s;
    uint32_t S;
    __device__ __forceinline__ uint64_t operator()(uint64_t s) const { return BFT_KH_SCAN_BIAS + (uint64_t)key_s[s] * S - s; }
};
// The lines, each written once and whole.  Every sorted row makes its own slot image (all lanes busy); the rows of a line -- neighbours in
// the sorted order, at most S of them -- OR their images together in their wavefront's LDS, and the row that comes first in the line stores
// its 64 bytes.  A line whose rows straddle two wavefronts is OR-ed into the (zeroed) table with atomics instead; lines without a k-mer
// keep the zeros of the memset.  A row displaced further from home than the slots' displacement bits hold goes to the overflow list
// (status[3] counts them; beyond BFT_KH_OVF_CAP: status[0] |= 1, no table) and leaves a tombstone -- its slot in use, value 0 -- so
// that the lines before the rows behind it stay full.  status[2] = the largest displacement in the table.
template <int W>
__global__ __launch_bounds__(256) void k_kh_assemble(const uint32_t* __restrict__ key_s, const KhRec<W>* __restrict__ rec_s, const uint64_t* __restrict__ vscan, uint64_t n, int k,
                                                     BftKhGeo g, uint32_t* __restrict__ status, uint64_t* __restrict__ kh, uint64_t* __restrict__ ovf_k,
                                                     uint32_t* __restrict__ ovf_v) {
    __shared__ unsigned long long s_line[4][64][BFT_KH_LINE_WORDS];  // per wavefront: the lines its 64 rows touch (at most 64)
    uint32_t dmax = 0;
    const uint64_t n_lines = g.nl + BFT_KH_TAIL_LINES;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t nblk = (n + 255) / 256;
    for (uint64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const uint64_t s = blk * 256 + threadIdx.x;
        const bool live = s < n;
        uint64_t p = 0, ln = ~0ull;
        bool first_of_line = false, last_of_line = false;
        if (live) {
            p = vscan[s] - BFT_KH_SCAN_BIAS + s;
            ln = p / g.S;
            first_of_line = s == 0 || (vscan[s - 1] - BFT_KH_SCAN_BIAS + (s - 1)) / g.S != ln;
            last_of_line = s + 1 == n || (vscan[s + 1] - BFT_KH_SCAN_BIAS + (s + 1)) / g.S != ln;
        }
        // the rank of this row's line among the lines of its wavefront's rows; a line is whole in the wavefront when its first and last row are
        const uint64_t heads = __ballot(live && (first_of_line || lane == 0));
        const uint32_t rank = (uint32_t)__popcll(heads & ((2ull << lane) - 1ull)) - 1u;
        const uint64_t firsts = __ballot(live && first_of_line), lasts = __ballot(live && last_of_line);
        // first / last row of my line inside the wavefront: the head lane of my rank, the lane before the next head
        const uint64_t my_head_bit = live ? (1ull << (63 - __builtin_clzll((heads & ((2ull << lane) - 1ull)) | 1ull))) : 0ull;
        const uint64_t above = heads & ~((my_head_bit << 1) - 1ull);
        const uint32_t my_last_lane = above ? (uint32_t)__builtin_ctzll(above) - 1u : 63u;
        const bool whole = live && (firsts & my_head_bit) && my_last_lane < 64u && ((lasts >> my_last_lane) & 1ull);
        // zero the wavefront's lines
#pragma unroll
        for (uint32_t q = 0; q < BFT_KH_LINE_WORDS; q++) s_line[wave][lane][q] = 0ull;
        __builtin_amdgcn_wave_barrier();
        uint64_t img[BFT_KH_LINE_WORDS];
#pragma unroll
        for (uint32_t q = 0; q < BFT_KH_LINE_WORDS; q++) img[q] = 0;
        if (live) {
            const uint64_t home = key_s[s], d = ln - home;
            const KhRec<W> r = rec_s[s];
            if (ln >= n_lines) atomicOr(&status[0], 1u);
            else if (d > g.maxd) {
                const uint32_t at = atomicAdd(&status[3], 1u);
                if (at >= BFT_KH_OVF_CAP) atomicOr(&status[0], 1u);
                else {
#pragma unroll
                    for (int w = 0; w < W; w++) ovf_k[(size_t)at * W + w] = r.t[w];
                    ovf_v[at] = r.v;
                }
                img[1] = 1ull << (64u - g.S + (uint32_t)(p % g.S));
            } else {
                dmax = max(dmax, (uint32_t)d);
                bft_kh_slot_image<W>(r.t, k, g, (uint32_t)(p % g.S), (uint32_t)d, r.v, img);
            }
            if (ln < n_lines) {
                if (whole) {
#pragma unroll
                    for (uint32_t q = 0; q < BFT_KH_LINE_WORDS; q++)
                        if (img[q]) atomicOr(&s_line[wave][rank][q], (unsigned long long)img[q]);
                } else {  // (a line that straddles two wavefronts)
                    unsigned long long* line = (unsigned long long*)(kh + ln * BFT_KH_LINE_WORDS);
#pragma unroll
                    for (uint32_t q = 0; q < BFT_KH_LINE_WORDS; q++)
                        if (img[q]) atomicOr(&line[q], (unsigned long long)img[q]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (whole && first_of_line && ln < n_lines) {
            ulonglong2* dst = reinterpret_cast<ulonglong2*>(kh + ln * BFT_KH_LINE_WORDS);
#pragma unroll
            for (int q = 0; q < 4; q++) dst[q] = make_ulonglong2(s_line[wave][rank][2 * q], s_line[wave][rank][2 * q + 1]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, (uint32_t)__shfl_down(dmax, o));
    if ((threadIdx.x & 63u) == 0 && dmax) atomicMax(&status[2], dmax);
}

// phase A: the k-mers sorted by home line (needs the sorted table only -- not the colour sets)
template <int W>
static int kh_sort_w(const uint64_t* d_tk, const uint32_t* d_vals, uint64_t n, int k, const BftKhGeo& g, BftKhScratch& sc, hipStream_t s) {
    DevBuf &key = sc.b[0], &rec = sc.b[1], &key_s = sc.b[2], &rec_s = sc.b[3], &tmp = sc.b[6];
    // stable sort by home line: rows of a line stay in T-form order (key / rec: the buffers between the passes)
    unsigned bits = 1;
    while (bits < 32 && (g.nl >> bits)) bits++;
    CK(key_s.alloc(n * 4));
    CK(rec_s.alloc(n * sizeof(KhRec<W>)));
    if (bft_rs::make_plan(0u, bits).P > 1) {
        CK(key.alloc(n * 4));
        CK(rec.alloc(n * sizeof(KhRec<W>)));
    }
    CK((bft_rs::sort<uint32_t, KhRec<W>, KhSortIn<W>, bft_rs::SHAPE_BACK>(KhSortIn<W>{d_tk, d_vals, k, g}, n, key_s.as<uint32_t>(), rec_s.as<KhRec<W>>(), key.as<uint32_t>(), rec.as<KhRec<W>>(), 0u, bits, s, tmp)));
    HIPCK(hipGetLastError());
    return 0;
}
// phase B: positions by one max-scan, then every line assembled and stored once
template <int W>
static int kh_lay_w(uint64_t n, int k, const BftKhGeo& g, uint64_t* d_kh, uint64_t* d_ovf_k, uint32_t* d_ovf_v, uint32_t* d_status, BftKhScratch& sc,
                    hipStream_t s) {
    DevBuf &key_s = sc.b[2], &rec_s = sc.b[3], &vs = sc.b[5], &tmp = sc.b[6];
    CK(vs.alloc(n * 8));
    const dim3 b(256);
    HIPCK(hipMemsetAsync(d_status, 0, 16, s));
    HIPCK(hipMemsetAsync(d_kh, 0, (g.nl + BFT_KH_TAIL_LINES) * BFT_KH_LINE_WORDS * 8, s));
    CK((bft_scan::scan<uint64_t, KhScanIn, bft_scan::Max, true>(KhScanIn{key_s.as<uint32_t>(), g.S}, vs.as<uint64_t>(), n, 0ull, bft_scan::Max(), s, tmp)));
    hipLaunchKernelGGL(k_kh_assemble<W>, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 256ull * 32)), b, 0, s, key_s.as<uint32_t>(), rec_s.as<KhRec<W>>(), vs.as<uint64_t>(), n, k, g,
                       d_status, d_kh, d_ovf_k, d_ovf_v);
    HIPCK(hipGetLastError());
    return 0;  // (the transients stay in `sc` until the caller has seen `s` drain)
}

int bft_kh_sort(const uint64_t* d_tk, const uint32_t* d_vals, uint64_t n, int k, int W, const BftKhGeo& g, BftKhScratch& sc, hipStream_t s) {
    switch (W) {
    case 1: return kh_sort_w<1>(d_tk, d_vals, n, k, g, sc, s);
    case 2: return kh_sort_w<2>(d_tk, d_vals, n, k, g, sc, s);
    case 3: return kh_sort_w<3>(d_tk, d_vals, n, k, g, sc, s);
    default: return kh_sort_w<4>(d_tk, d_vals, n, k, g, sc, s);
    }
}
int bft_kh_lay(uint64_t n, int k, int W, const BftKhGeo& g, uint64_t* d_kh, uint64_t* d_ovf_k, uint32_t* d_ovf_v, uint32_t* d_status, BftKhScratch& sc, hipStream_t s) {
    switch (W) {
    case 1: return kh_lay_w<1>(n, k, g, d_kh, d_ovf_k, d_ovf_v, d_status, sc, s);
    case 2: return kh_lay_w<2>(n, k, g, d_kh, d_ovf_k, d_ovf_v, d_status, sc, s);
    case 3: return kh_lay_w<3>(n, k, g, d_kh, d_ovf_k, d_ovf_v, d_status, sc, s);
    default: return kh_lay_w<4>(n, k, g, d_kh, d_ovf_k, d_ovf_v, d_status, sc, s);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dump ("compact_table": the sorted table comes back from here)
// ---------------------------------------------------------------------------------------------------------------------------------
// Every (k-mer, value) the table holds, in any order: one thread per line rebuilds the T-form of every used slot (the hashed high bits
// from the slot's home line and q, the rest from the slot), reserves places with one atomic per workgroup.  Word w of the j-th k-mer goes
// to keys[w * stride + j].
template <int W>
__global__ __launch_bounds__(256) void k_kh_dump(BftImage im, uint64_t* __restrict__ keys, uint64_t stride, uint32_t* __restrict__ vals, unsigned long long* __restrict__ cnt) {
    __shared__ uint32_t s_cnt;
    __shared__ unsigned long long s_base;
    const uint64_t n_lines = im.kh.nl + BFT_KH_TAIL_LINES;
    const uint32_t S = im.kh.S;
    for (uint64_t l0 = (uint64_t)blockIdx.x * blockDim.x; l0 < n_lines; l0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t ln = l0 + threadIdx.x;
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        uint64_t hd[2] = {0, 0};
        uint32_t occ = 0;
        const uint64_t* line = im.kh_lines + ln * BFT_KH_LINE_WORDS;
        if (ln < n_lines) {
            bft_kh_load_header(line, hd);
            occ = (uint32_t)(hd[1] >> (64u - S));
        }
        {   // (slots in use with value 0 are tombstones: not k-mers)
            uint32_t real = 0;
            const uint64_t vmask = (1ull << im.kh.cb) - 1ull;
            for (uint32_t o2 = occ; o2;) {
                const uint32_t s = (uint32_t)__builtin_ctz(o2);
                o2 &= o2 - 1u;
                uint64_t body[W];
                bft_kh_load_body<W>(line, s, im.kh.wb, body);
                if (body[0] & vmask) real |= 1u << s;
            }
            occ = real;
        }
        const uint32_t mine = (uint32_t)__popc(occ);
        uint32_t my_off = mine ? atomicAdd(&s_cnt, mine) : 0u;
        __syncthreads();
        if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(cnt, (unsigned long long)s_cnt) : 0ull;
        __syncthreads();
        uint64_t j = s_base + my_off;
        while (occ) {
            const uint32_t s = (uint32_t)__builtin_ctz(occ);
            occ &= occ - 1u;
            uint64_t t[W];
            uint32_t v;
            bft_kh_slot_decode<W>(im, line, hd, ln, s, t, &v);
#pragma unroll
            for (int i = 0; i < W; i++) keys[(uint64_t)i * stride + j] = t[i];
            vals[j] = v;
            j++;
        }
        __syncthreads();
    }
    // the overflow list: one workgroup appends it
    if (blockIdx.x == 0) {
        __syncthreads();
        if (threadIdx.x == 0) s_base = im.kh_ovf_n ? atomicAdd(cnt, (unsigned long long)im.kh_ovf_n) : 0ull;
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < im.kh_ovf_n; e += blockDim.x) {
#pragma unroll
            for (int i = 0; i < W; i++) keys[(uint64_t)i * stride + s_base + e] = im.kh_ovf[(size_t)e * W + i];
            vals[s_base + e] = im.kh_ovf_val[e];
        }
    }
}

int bft_kh_dump(const BftImage& im, uint64_t* d_keys, uint64_t stride, uint32_t* d_vals, unsigned long long* d_cnt, hipStream_t s) {
    const dim3 grid(256 * 8), block(256);
    switch (im.W) {
    case 1: hipLaunchKernelGGL(k_kh_dump<1>, grid, block, 0, s, im, d_keys, stride, d_vals, d_cnt); break;
    case 2: hipLaunchKernelGGL(k_kh_dump<2>, grid, block, 0, s, im, d_keys, stride, d_vals, d_cnt); break;
    case 3: hipLaunchKernelGGL(k_kh_dump<3>, grid, block, 0, s, im, d_keys, stride, d_vals, d_cnt); break;
    default: hipLaunchKernelGGL(k_kh_dump<4>, grid, block, 0, s, im, d_keys, stride, d_vals, d_cnt); break;
    }
    HIPCK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// launchers (called from bft_gpu.hip).  The kernels are instantiated for the (key words, slots per line) pairs an index can have:
// one-word keys (k <= 32) 10..4 slots, two-word keys 10..3, three-word keys 4..2, four-word keys 2..1.
// ---------------------------------------------------------------------------------------------------------------------------------
bool bft_kh_has_kernels(int W, uint32_t S) {
    switch (W) {
    case 1: return S >= 4 && S <= 10;
    case 2: return S >= 3 && S <= 10;
    case 3: return S >= 2 && S <= 4;
    default: return S >= 1 && S <= 2;
    }
}

// Rounds of `chunk` blocks (bft_claims.h).  d_ctr != NULL: the rounds after the first are claimed, and no more workgroups than are resident
// are launched; else every round is dealt out by workgroup number, over four times as many (44.6 -> 47.2 G k-mers/s on the 100-genome
// index when that was the only form: the tail of a persistent grid is shorter).
static dim3 kh_round_grid(uint64_t n, uint32_t chunk, int mult, bool claimed) {
    const uint64_t nblk = (n + BFT_KH_BLOCK - 1) / BFT_KH_BLOCK, rounds = (nblk + chunk - 1) / chunk;
    const uint64_t resident = 256ull * 8 * (uint64_t)std::max(1, mult) * (claimed ? 1ull : 4ull);
    return dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(rounds, resident)));
}

int bft_kh_query(const BftImage& im, int grid_mult, const uint8_t* d_kmers, uint64_t n, int rec, uint64_t* d_bits64, uint32_t* d_out32, BftClaimCtr d_ctr, uint32_t chunk,
                 hipStream_t s) {
    const dim3 block(BFT_KH_BLOCK);
    chunk = std::max(1u, std::min(chunk, BFT_KH_MAX_CLAIM));
    const dim3 grid = kh_round_grid(n, chunk, grid_mult, d_ctr.p != nullptr);
    KH_DISPATCH(im.W, (int)im.kh.S, hipLaunchKernelGGL((k_query_kh<KW, KS>), grid, block, 0, s, im, d_kmers, n, rec, d_bits64, d_out32, d_ctr, chunk));
    HIPCK(hipGetLastError());
    return 0;
}

// presence bits, offsets [n + 1] and genome ids of n packed k-mers in one launch (k_colors_kh); d_scratch: bft_kh_colors_scratch_bytes(n) bytes
size_t bft_kh_colors_scratch_bytes(uint64_t n) { return ((n + (uint64_t)BFT_KH_CT * BFT_KH_BLOCK - 1) / ((uint64_t)BFT_KH_CT * BFT_KH_BLOCK) + 1) * 8; }
int bft_kh_colors(const BftImage& im, const uint8_t* d_kmers, uint64_t n, int rec, uint64_t* d_bits64, uint64_t* d_offsets, uint32_t* d_ids, uint64_t ids_cap, uint64_t* d_needed,
                  void* d_scratch, hipStream_t s) {
    const uint64_t ntiles = (n + (uint64_t)BFT_KH_CT * BFT_KH_BLOCK - 1) / ((uint64_t)BFT_KH_CT * BFT_KH_BLOCK);
    CK(bft_zero_async(d_scratch, (ntiles + 1) * 8, s));  // (a kernel, not a memset: the call may be recorded into a graph, bft_dev.h)
    const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ntiles, 256ull * 8))), block(BFT_KH_BLOCK);
    KH_DISPATCH(im.W, (int)im.kh.S,
                hipLaunchKernelGGL((k_colors_kh<KW, KS>), grid, block, 0, s, im, d_kmers, n, rec, d_bits64, (unsigned long long*)d_offsets, d_ids, ids_cap, (unsigned long long*)d_needed,
                                   (unsigned long long*)d_scratch));
    HIPCK(hipGetLastError());
    return 0;
}

template <int W, int S>
static void kh_color_rows_launch(const BftImage& im, const uint8_t* d_kmers, uint64_t n, int rec, uint64_t* d_bits64, const uint8_t* bm, uint32_t stride, uint32_t rowbytes,
                                 uint32_t tile_rows, uint32_t div_m, uint32_t div_l, uint8_t* d_out, int device, hipStream_t s) {
    // the tiles are dealt out by workgroup number: no more workgroups than are resident at once (per device: CU count and partition mode may differ)
    static std::atomic<int> resident_dev[64];
    std::atomic<int>& res = resident_dev[device & 63];
    int r = res.load(std::memory_order_relaxed);
    if (!r) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_color_rows_kh<W, S>, 256, 0) != hipSuccess || per_cu < 1) { per_cu = 4; (void)hipGetLastError(); }
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) { cus = 256; (void)hipGetLastError(); }
        r = per_cu * cus;
        res.store(r, std::memory_order_relaxed);
    }
    const uint64_t tiles = (n + tile_rows - 1) / tile_rows;
    const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((tiles + 3) / 4, (uint64_t)r)));
    hipLaunchKernelGGL((k_color_rows_kh<W, S>), grid, dim3(256), 0, s, im, d_kmers, n, rec, d_bits64, bm, stride, rowbytes, tile_rows, div_m, div_l, d_out);
}

// presence bits and colour rows (rowbytes >= 16 bytes each, d_out 16-byte aligned) of n packed k-mers in one launch (k_color_rows_kh); bm: the bitmap
// dictionary (rows of `stride` bytes, slack on either side: ensure_cs_bitmaps)
int bft_kh_color_rows(const BftImage& im, const uint8_t* d_kmers, uint64_t n, int rec, uint64_t* d_bits64, const uint8_t* bm, uint32_t stride, uint32_t rowbytes, uint8_t* d_out,
                      int device, hipStream_t s) {
    // tiles of about 16 KiB of output, a multiple of 64 k-mers.  (Config 5, 250-byte rows, 4x10^6 k-mers: tiles of 64 / 128 / 256 k-mers at 4, 5, 6
    // workgroups per CU all take 0.37-0.40 ms, the smallest tiles and the most workgroups the least -- the launch costs what the lookups and the
    // rows cost one after the other, whichever way they are interleaved: DESIGN.md.)
    const uint32_t tile_rows = std::max(64u, std::min(BFT_KH_ROWS_TILE, ((16u << 10) / rowbytes) & ~63u));
    uint32_t div_l = 0;
    while ((1ull << div_l) < rowbytes) div_l++;
    const uint32_t div_m = (uint32_t)(((1ull << 32) * ((1ull << div_l) - rowbytes)) / rowbytes + 1ull);
    KH_DISPATCH(im.W, (int)im.kh.S, (kh_color_rows_launch<KW, KS>(im, d_kmers, n, rec, d_bits64, bm, stride, rowbytes, tile_rows, div_m, div_l, d_out, device, s)));
    HIPCK(hipGetLastError());
    return 0;
}

int bft_kh_branching(const BftImage& im, const uint8_t* d_kmers, uint64_t n, int B, uint64_t* d_bits64, uint8_t* d_counts, BftClaimCtr d_ctr, uint32_t chunk, hipStream_t s) {
    const dim3 block(BFT_KH_BLOCK);
    chunk = std::max(1u, std::min(chunk, BFT_KH_BR_MAX_CLAIM));
    const dim3 grid = kh_round_grid(n, chunk, 1, d_ctr.p != nullptr);
    KH_DISPATCH(im.W, (int)im.kh.S, hipLaunchKernelGGL((k_branching_kh<KW, KS>), grid, block, 0, s, im, d_kmers, n, B, d_bits64, d_counts, d_ctr, chunk));
    HIPCK(hipGetLastError());
    return 0;
}

int bft_kh_seq(const BftImage& im, const uint64_t* d_codes, const uint32_t* d_bad, const uint64_t* d_seq_off, const uint64_t* d_pos_off, const uint32_t* d_tile_seq,
               uint32_t n_seqs, int canonical, uint32_t* d_csout, BftClaimCtr d_ctr, uint32_t chunk, hipStream_t s) {
    const dim3 grid(256 * 8), block(256);
    chunk = std::max(1u, std::min(chunk, BFT_KH_MAX_CLAIM));
    KH_DISPATCH(im.W, (int)im.kh.S,
                hipLaunchKernelGGL((k_seq_kh<KW, KS>), grid, block, 0, s, im, d_codes, d_bad, d_seq_off, d_pos_off, d_tile_seq, n_seqs, canonical, d_csout, d_ctr, chunk));
    HIPCK(hipGetLastError());
    return 0;
}
