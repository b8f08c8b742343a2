// bft_kh.hip -- the k-mer hash (bft_image.h, BFT_KH_*): its build on the GPU, its dump ("compact_table"), and the queries of
// include/bft_gpu.h through it: presence / colour set, branching, sequence positions.  Its own translation unit: these kernels stage
// nothing and walk nothing -- T-form, region of the root prefix, home line, compare.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "bft_dev.h"
#include "bft_image.h"
#include "bft_kh.h"
#include "bft_claims.h"
#include "bft_walk.h"
#include "bft_kernels_load.h"
#include "bft_kernels_seqwin.h"

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
// A lookup is a chain of loads -- the packed k-mer, its region (an L2 hit), the header of its home line (a miss), the body of the slot that
// matches -- and the vector memory path returns in order: a hit waits behind the misses in front of it, so every link costs a full trip.
// The links of DIFFERENT k-mers do not depend on each other, so the blocks of a round are software-pipelined: while the header of block
// b is in flight, the region of block b + 1 and the packed k-mers of block b + 2 are too, and a lane waits for one trip per k-mer (two for
// a stored one: the body) instead of three or four.
template <int W, int S>
__global__ __launch_bounds__(BFT_KH_BLOCK) void k_query_kh(BftImage im, const uint8_t* __restrict__ packed, uint64_t n, int B, uint64_t* __restrict__ bits64,
                                                          uint32_t* __restrict__ out32, uint32_t* __restrict__ ctr, uint32_t chunk) {
    const uint64_t end_aligned = ((uint64_t)packed + n * (uint64_t)B) & ~3ull;
    const uint64_t nblk = (n + BFT_KH_BLOCK - 1) / BFT_KH_BLOCK, nwords = (n + 63) / 64;
    constexpr uint32_t WPB = BFT_KH_BLOCK / 64;  // presence words per block
    __shared__ uint32_t s_next[2];
    __shared__ uint64_t s_bits[BFT_KH_MAX_CLAIM * WPB];
    KhClaims cl(ctr, chunk, nblk, s_next);
    cl.first();
    while (cl.blk < nblk) {
        // ---- one round: blocks cl.start .. cl.blk_end - 1 ----
        uint64_t x_nx[W], t_cur[W];   // packed k-mer of the block after next; T-form of the current block's k-mer
        BftKhRegion g_cur;
        g_cur.first = g_cur.lines = g_cur.mh = g_cur.special = 0;
        {   // prologue: k-mer and region of the round's first block, k-mer of its second
            const uint64_t i0 = cl.blk * BFT_KH_BLOCK + threadIdx.x, i1 = i0 + BFT_KH_BLOCK;
#pragma unroll
            for (int w = 0; w < W; w++) { t_cur[w] = 0; x_nx[w] = 0; }
            if (i0 < n) {
                uint64_t x[W];
                load_x<W>(packed, i0, B, end_aligned, x);
                bft_tform_from_x<W>(x, im.k, t_cur);
                g_cur = bft_kh_region(im.kreg, bft_digit<W>(t_cur, im.k, 0));
            }
            if (cl.blk + 1 < cl.blk_end && i1 < n) load_x<W>(packed, i1, B, end_aligned, x_nx);
        }
        for (; cl.blk < cl.blk_end; cl.blk++) {
            const uint64_t i = cl.blk * BFT_KH_BLOCK + threadIdx.x;
            const bool live = i < n && g_cur.lines != 0;  // (an empty region: absent without a table line)
            // (1) the header of this block's home line
            uint64_t hd[2] = {0, 0}, at = 0;
            BftKhKey<W> key;
            if (live) {
                uint64_t remle[W];
                bft_kh_rem<W>(t_cur, im.k, remle);
                bft_kh_key<W>(remle, im.kh_rb, im.kh_f, im.kh_cb, key);
                at = (uint64_t)g_cur.first + bft_kh_home_of(bft_kh_hash<W>(remle), g_cur.mh);
                bft_kh_load_header(im.kh + at * BFT_KH_LINE_WORDS, hd);
            }
            // (2) the region of the next block's k-mer, (3) the packed k-mer of the block after it -- in flight beside the header
            uint64_t t_nx[W];
            BftKhRegion g_nx;
            g_nx.first = g_nx.lines = g_nx.mh = g_nx.special = 0;
            const uint64_t i1 = i + BFT_KH_BLOCK, i2 = i1 + BFT_KH_BLOCK;
            const bool has1 = cl.blk + 1 < cl.blk_end && i1 < n, has2 = cl.blk + 2 < cl.blk_end && i2 < n;
#pragma unroll
            for (int w = 0; w < W; w++) t_nx[w] = 0;
            if (has1) {
                bft_tform_from_x<W>(x_nx, im.k, t_nx);
                g_nx = bft_kh_region(im.kreg, bft_digit<W>(t_nx, im.k, 0));
            }
            if (has2) load_x<W>(packed, i2, B, end_aligned, x_nx);
            // (4) this block's answer
            bool present = false;
            uint32_t val = 0xFFFFFFFFu;
            if (live) {
                int res = bft_kh_scan<W, S>(im, im.kh + at * BFT_KH_LINE_WORDS, hd, key, &val);
                const uint64_t end = (uint64_t)g_cur.first + g_cur.lines;
                while (res < 0 && ++at < end) {  // full line without the key: on from the home line (a few per cent)
                    bft_kh_load_header(im.kh + at * BFT_KH_LINE_WORDS, hd);
                    res = bft_kh_scan<W, S>(im, im.kh + at * BFT_KH_LINE_WORDS, hd, key, &val);
                }
                present = res > 0;
            }
            const uint64_t mask = __ballot(present);
            if (out32 && i < n) out32[i] = present ? val : 0xFFFFFFFFu;
            if ((threadIdx.x & 63u) == 0) s_bits[(cl.blk - cl.start) * WPB + (threadIdx.x >> 6)] = mask;
#pragma unroll
            for (int w = 0; w < W; w++) t_cur[w] = t_nx[w];
            g_cur = g_nx;
            if (cl.ctr && threadIdx.x == 0 && cl.blk == cl.start) cl.claim(cl.blk_end);  // (the next round's claim travels from here on)
        }
        // the round is answered: its words leave together
        __syncthreads();
        {
            const uint64_t w0 = cl.start * WPB, w1 = min(nwords, cl.blk_end * WPB);
            if (w0 + threadIdx.x < w1) __builtin_nontemporal_store(s_bits[threadIdx.x], &bits64[w0 + threadIdx.x]);
        }
        cl.take();  // (its barriers stand between these reads of s_bits and the next round's writes)
    }
    cl.done();
}

// How many of four candidate k-mers are stored: the four regions, then the four home lines' headers, are loaded before any is looked at --
// four independent misses in flight instead of four dependent walks (src/presenceNode.c:15-1211 shares one descent between the four;
// here there is no descent to share).  A candidate whose home line is full without holding it continues line by line (a few per cent).
template <int W, int S>
__device__ __forceinline__ int kh_count4(const BftImage& im, const uint64_t (*cand)[W]) {
    BftKhRegion g[4];
    BftKhKey<W> key[4];
    uint64_t at[4], hd[4][2];
#pragma unroll
    for (int v = 0; v < 4; v++) g[v] = bft_kh_region(im.kreg, bft_digit<W>(cand[v], im.k, 0));
#pragma unroll
    for (int v = 0; v < 4; v++) {
        uint64_t remle[W];
        bft_kh_rem<W>(cand[v], im.k, remle);
        bft_kh_key<W>(remle, im.kh_rb, im.kh_f, im.kh_cb, key[v]);
        at[v] = (uint64_t)g[v].first + (g[v].lines ? bft_kh_home_of(bft_kh_hash<W>(remle), g[v].mh) : 0u);
    }
#pragma unroll
    for (int v = 0; v < 4; v++)
        if (g[v].lines) bft_kh_load_header(im.kh + at[v] * BFT_KH_LINE_WORDS, hd[v]);
    int count = 0;
#pragma unroll
    for (int v = 0; v < 4; v++) {
        if (!g[v].lines) continue;
        uint32_t val;
        int res = bft_kh_scan<W, S>(im, im.kh + at[v] * BFT_KH_LINE_WORDS, hd[v], key[v], &val);
        const uint64_t end = (uint64_t)g[v].first + g[v].lines;
        while (res < 0 && ++at[v] < end) {  // full line without the key: on from the home line
            uint64_t h2[2];
            bft_kh_load_header(im.kh + at[v] * BFT_KH_LINE_WORDS, h2);
            res = bft_kh_scan<W, S>(im, im.kh + at[v] * BFT_KH_LINE_WORDS, h2, key[v], &val);
        }
        count += res > 0;
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
                                                              uint8_t* __restrict__ counts, uint32_t* __restrict__ ctr, uint32_t chunk) {
    const uint64_t end_aligned = ((uint64_t)packed + n * (uint64_t)B) & ~3ull;
    const uint64_t nblk = (n + BFT_KH_BLOCK - 1) / BFT_KH_BLOCK, nwords = (n + 63) / 64;
    constexpr uint32_t WPB = BFT_KH_BLOCK / 64;
    const int k = im.k, L = im.L, rb = 2 * (k - 9 * L);
    __shared__ uint32_t s_next[2];
    __shared__ uint64_t s_bits[BFT_KH_BR_MAX_CLAIM * WPB];
    KhClaims cl(ctr, chunk, nblk, s_next);
    cl.first();
    while (cl.blk < nblk) {
        const uint64_t i = cl.blk * BFT_KH_BLOCK + threadIdx.x;
        int branching = 0;
        if (i < n) {
            uint64_t x[W], y[W], t[W], cand[4][W];
            load_x<W>(packed, i, B, end_aligned, x);
            // successors: drop the first nucleotide, the last one is the wildcard (bits vo.. of the T-form's last word)
#pragma unroll
            for (int w = 0; w < W; w++) y[w] = (x[w] >> 2) | (w + 1 < W ? x[w + 1] << 62 : 0ull);
            bft_tform_from_x<W>(y, k, t);
            const int vo = rb ? 0 : 2;
#pragma unroll
            for (int v = 0; v < 4; v++) {
#pragma unroll
                for (int w = 0; w < W; w++) cand[v][w] = t[w] | (w == W - 1 ? (uint64_t)v << vo : 0ull);
            }
            const int cr = kh_count4<W, S>(im, cand);
            int cl_ = 0;
            if (counts || cr < 2) {
                // predecessors: shift in a wildcard first nucleotide (bits 0..1 of the first digit), drop the last one
#pragma unroll
                for (int w = W - 1; w >= 0; w--) y[w] = (x[w] << 2) | (w > 0 ? x[w - 1] >> 62 : 0ull);
                const int top = 2 * k - 64 * (W - 1);
                if (top < 64) y[W - 1] &= (1ull << top) - 1ull;
                bft_tform_from_x<W>(y, k, t);
                const int o = rb + 18 * (L - 1), ow = W - 1 - (o >> 6), osh = o & 63;
#pragma unroll
                for (int v = 0; v < 4; v++) {
#pragma unroll
                    for (int w = 0; w < W; w++) cand[v][w] = t[w] | (w == ow ? (uint64_t)v << osh : 0ull);
                }
                cl_ = kh_count4<W, S>(im, cand);
            }
            branching = cr > 1 || cl_ > 1;
            if (counts) counts[i] = (uint8_t)((cr << 4) | cl_);  // (a wavefront's 64 bytes: one coalesced store)
        }
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

// Sequence positions: the colour set of a position sits in the slot that says the k-mer is stored -- one line per position, nothing staged.
template <int W, int S>
__global__ __launch_bounds__(256) void k_seq_kh(BftImage im, const uint64_t* __restrict__ codes, const uint32_t* __restrict__ bad, const uint64_t* __restrict__ seq_off,
                                                const uint64_t* __restrict__ pos_off, const uint32_t* __restrict__ tile_seq, uint32_t n_seqs, int canonical,
                                                uint32_t* __restrict__ csout, uint32_t* __restrict__ ctr, uint32_t chunk) {
    const uint64_t P = pos_off[n_seqs];
    const uint64_t nblk = (P + 255) / 256;
    __shared__ uint32_t s_next[2];
    KhClaims cl(ctr, chunk, nblk, s_next);  // (the number of positions is only known on the device: the grid is the resident one, rounds beyond nblk are empty)
    cl.first();
    for (; cl.blk < nblk; cl.advance()) {
        const uint64_t p = cl.blk * 256 + threadIdx.x;
        if (p >= P) continue;
        uint32_t lo = tile_seq[p >> 6];
        while (lo + 1 < n_seqs && pos_off[lo + 1] <= p) lo++;
        uint32_t cs = 0xFFFFFFFFu;
        uint64_t x[W], t[W];
        if (seq_window<W>(codes, bad, seq_off[lo] + (p - pos_off[lo]), im.k, canonical, x)) {
            bft_tform_from_x<W>(x, im.k, t);
            uint32_t val;
            if (bft_kh_lookup<W, S>(im, t, &val)) cs = val;
        }
        csout[p] = cs;
    }
    cl.done();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// build (the canonical layout of bft_image.h: rows of a region in (home line, T-form) order, slot-level linear probing)
// ---------------------------------------------------------------------------------------------------------------------------------
// rr[r] = first row of the sorted table whose root prefix is >= r (r = 0 .. 2^18)
template <int W>
__global__ void k_kh_rows(const uint64_t* __restrict__ tk, uint64_t n, int k, uint32_t* __restrict__ rr) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > BFT_KH_REGIONS) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        uint64_t row[W];
#pragma unroll
        for (int w = 0; w < W; w++) row[w] = tk[mid * W + w];
        if (bft_digit<W>(row, k, 0) < r) lo = mid + 1; else hi = mid;
    }
    rr[r] = (uint32_t)lo;
}
// provisional lines per region: home lines + one tail line
__global__ void k_kh_plan(const uint32_t* __restrict__ rr, uint32_t S, uint32_t load, uint32_t* __restrict__ prov) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > BFT_KH_REGIONS) return;
    prov[r] = r < BFT_KH_REGIONS && rr[r + 1] > rr[r] ? bft_kh_home_lines(rr[r + 1] - rr[r], S, load) + 1u : 0u;
}
// sort key of row i: its provisional global home line (monotone in (region, home line)); value: the row
template <int W>
__global__ void k_kh_keys(const uint64_t* __restrict__ tk, uint64_t n, int k, const uint32_t* __restrict__ rr, const uint32_t* __restrict__ base0, uint32_t S, uint32_t load,
                          uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t t[W], remle[W];
        bft_load_row<W>(tk + i * W, t);
        const uint32_t r = bft_digit<W>(t, k, 0);
        const uint32_t mh = bft_kh_home_lines(rr[r + 1] - rr[r], S, load);
        bft_kh_rem<W>(t, k, remle);
        key[i] = base0[r] + bft_kh_home_of(bft_kh_hash<W>(remle), mh);
        val[i] = (uint32_t)i;
    }
}
// Slot-level linear probing over the rows of a region in sorted order: p_i = max(home slot_i, p_(i-1) + 1) = i + max_(j <= i)(home slot_j - j).
// The values carry the region in their top bits, so ONE device-wide inclusive max-scan restarts at every region by itself.
#define BFT_KH_SCAN_BIAS (1ull << 39)
template <int W>
__global__ void k_kh_scanvals(const uint64_t* __restrict__ tk, const uint32_t* __restrict__ key_s, const uint32_t* __restrict__ val_s, uint64_t n, int k,
                              const uint32_t* __restrict__ rr, const uint32_t* __restrict__ base0, uint32_t S, uint64_t* __restrict__ v) {
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < n; s += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t t[W];
        bft_load_row<W>(tk + (uint64_t)val_s[s] * W, t);
        const uint64_t r = bft_digit<W>(t, k, 0);
        const uint64_t li = s - rr[r], j = key_s[s] - base0[r];
        v[s] = (r << 40) | (BFT_KH_SCAN_BIAS + j * S - li);
    }
}
struct BftMaxU64 {
    __host__ __device__ uint64_t operator()(const uint64_t& a, const uint64_t& b) const { return a > b ? a : b; }
};
// p of every sorted row (slot offset inside its region); the last row of a region sizes it: lines, tail.  status[0] != 0: the table cannot
// be built (a region beyond BFT_KH_MAX_TAIL tail lines, a region beyond 2^32 slots).
template <int W>
__global__ void k_kh_place(const uint64_t* __restrict__ tk, const uint32_t* __restrict__ val_s, const uint64_t* __restrict__ vscan, uint64_t n, int k,
                           const uint32_t* __restrict__ rr, uint32_t S, uint32_t load, uint32_t* __restrict__ p_out, uint32_t* __restrict__ lines, uint32_t* __restrict__ tails,
                           uint32_t* __restrict__ status) {
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < n; s += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t t[W];
        bft_load_row<W>(tk + (uint64_t)val_s[s] * W, t);
        const uint32_t r = bft_digit<W>(t, k, 0);
        const uint64_t li = s - rr[r];
        const uint64_t p = (vscan[s] & ((1ull << 40) - 1ull)) - BFT_KH_SCAN_BIAS + li;
        if (p >> 32) atomicOr(&status[0], 2u);
        p_out[s] = (uint32_t)p;
        if (s + 1 == rr[r + 1]) {  // the region's last row in sorted order holds its highest slot
            const uint32_t mh = bft_kh_home_lines(rr[r + 1] - rr[r], S, load);
            uint64_t used = p / S + 1;
            if (p % S == S - 1) used++;  // (the last line keeps a free slot)
            const uint32_t code = bft_kh_tail_code(used > mh ? used - mh : 1);
            if (code > 3u) atomicOr(&status[0], 1u);
            lines[r] = mh + BFT_KH_TAIL_OF(code & 3u);
            tails[r] = code & 3u;
        }
    }
}
__global__ void k_kh_zero_u32(uint32_t* __restrict__ a, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = 0;
}
// kreg[r] = first line | tail - 1; kreg[2^18] = lines in use.  status[1] = lines in use; more than the table holds: status[0] |= 4
__global__ void k_kh_kreg(const uint32_t* __restrict__ first, const uint32_t* __restrict__ tails, uint64_t lines_cap, uint32_t* __restrict__ kreg, uint32_t* __restrict__ status) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > BFT_KH_REGIONS) return;
    const uint32_t f = first[r];
    if (r == BFT_KH_REGIONS) {
        status[1] = f;
        if ((uint64_t)f > lines_cap || f > BFT_KREG_LINE_MASK) atomicOr(&status[0], 4u);
        kreg[r] = f & BFT_KREG_LINE_MASK;
        return;
    }
    kreg[r] = (f & BFT_KREG_LINE_MASK) | ((tails[r] & 3u) << BFT_KREG_TAIL_SHIFT);
}
// every row ORs its slot -- header field, occupancy bit, body -- into the (zeroed) table
template <int W>
__global__ void k_kh_write(const uint64_t* __restrict__ tk, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ val_s, const uint32_t* __restrict__ p_s, uint64_t n,
                           int k, const uint32_t* __restrict__ kreg, BftKhGeometry g, const uint32_t* __restrict__ status, uint64_t* __restrict__ kh) {
    if (status[0]) return;
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < n; s += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = val_s[s];
        uint64_t t[W], img[BFT_KH_LINE_WORDS];
        bft_load_row<W>(tk + i * W, t);
        const uint32_t r = bft_digit<W>(t, k, 0);
        const uint32_t p = p_s[s];
        bft_kh_slot_image<W>(t, k, g.S, g.rb, g.f, g.wb, g.cb, p % g.S, vals[i], img);
        unsigned long long* line = (unsigned long long*)(kh + ((uint64_t)(kreg[r] & BFT_KREG_LINE_MASK) + p / g.S) * BFT_KH_LINE_WORDS);
#pragma unroll
        for (uint32_t q = 0; q < BFT_KH_LINE_WORDS; q++)
            if (img[q]) atomicOr(&line[q], (unsigned long long)img[q]);
    }
}

static int kh_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, DevBuf& tmp, hipStream_t s) {
    size_t tb = 0;
    HIPCK(rocprim::exclusive_scan(nullptr, tb, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
    if (tb > tmp.bytes) CK(tmp.alloc(tb));
    HIPCK(rocprim::exclusive_scan(tmp.p, tb, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
    return 0;
}

uint64_t bft_kh_lines_bound(uint64_t n, uint32_t S, uint32_t load_pct) {
    if (n == 0) return 0;
    const uint64_t per = (uint64_t)S * load_pct;
    const uint64_t home = (n * 100ull + per - 1) / per;
    const uint64_t regions = std::min<uint64_t>(n, BFT_KH_REGIONS);
    return home + 2 * regions + std::max<uint64_t>(4096, home / 128);
}

template <int W>
static int kh_build_w(const uint64_t* d_tk, const uint32_t* d_vals, uint64_t n, int k, const BftKhGeometry& g, uint32_t load, uint64_t* d_kh, uint64_t lines_cap,
                      uint32_t* d_kreg, uint32_t* d_status, BftKhScratch& sc, hipStream_t s) {
    const uint32_t NR = BFT_KH_REGIONS + 1;
    DevBuf &rr = sc.b[0], &prov = sc.b[1], &base0 = sc.b[2], &key = sc.b[3], &val = sc.b[4], &key_s = sc.b[5], &val_s = sc.b[6], &v = sc.b[7], &vs = sc.b[8], &p = sc.b[9],
           &lines = sc.b[10], &tails = sc.b[11], &first = sc.b[12], &tmp = sc.b[13];
    CK(rr.alloc((NR + 1) * 4));
    CK(prov.alloc(NR * 4));
    CK(base0.alloc(NR * 4));
    CK(key.alloc(n * 4));
    CK(val.alloc(n * 4));
    CK(key_s.alloc(n * 4));
    CK(val_s.alloc(n * 4));
    CK(v.alloc(n * 8));
    CK(vs.alloc(n * 8));
    CK(p.alloc(n * 4));
    CK(lines.alloc(NR * 4));
    CK(tails.alloc(NR * 4));
    CK(first.alloc(NR * 4));
    const dim3 gr((NR + 255) / 256), b(256), gn((unsigned)std::min<uint64_t>((n + 255) / 256, 256ull * 32));
    HIPCK(hipMemsetAsync(d_status, 0, 8, s));
    HIPCK(hipMemsetAsync(d_kh, 0, lines_cap * BFT_KH_LINE_WORDS * 8, s));
    hipLaunchKernelGGL(k_kh_rows<W>, gr, b, 0, s, d_tk, n, k, rr.as<uint32_t>());
    hipLaunchKernelGGL(k_kh_plan, gr, b, 0, s, rr.as<uint32_t>(), g.S, load, prov.as<uint32_t>());
    CK(kh_scan_u32(prov.as<uint32_t>(), base0.as<uint32_t>(), NR, tmp, s));
    hipLaunchKernelGGL(k_kh_keys<W>, gn, b, 0, s, d_tk, n, k, rr.as<uint32_t>(), base0.as<uint32_t>(), g.S, load, key.as<uint32_t>(), val.as<uint32_t>());
    {   // stable sort by provisional home line: rows of a line stay in T-form order.  (Every bit: the largest key is not known on the host.)
        size_t tb = 0;
        HIPCK(rocprim::radix_sort_pairs(nullptr, tb, key.as<uint32_t>(), key_s.as<uint32_t>(), val.as<uint32_t>(), val_s.as<uint32_t>(), (size_t)n, 0u, 32u, s));
        if (tb > tmp.bytes) CK(tmp.alloc(tb));
        HIPCK(rocprim::radix_sort_pairs(tmp.p, tb, key.as<uint32_t>(), key_s.as<uint32_t>(), val.as<uint32_t>(), val_s.as<uint32_t>(), (size_t)n, 0u, 32u, s));
    }
    hipLaunchKernelGGL(k_kh_scanvals<W>, gn, b, 0, s, d_tk, key_s.as<uint32_t>(), val_s.as<uint32_t>(), n, k, rr.as<uint32_t>(), base0.as<uint32_t>(), g.S, v.as<uint64_t>());
    {
        size_t tb = 0;
        HIPCK(rocprim::inclusive_scan(nullptr, tb, v.as<uint64_t>(), vs.as<uint64_t>(), (size_t)n, BftMaxU64(), s));
        if (tb > tmp.bytes) CK(tmp.alloc(tb));
        HIPCK(rocprim::inclusive_scan(tmp.p, tb, v.as<uint64_t>(), vs.as<uint64_t>(), (size_t)n, BftMaxU64(), s));
    }
    hipLaunchKernelGGL(k_kh_zero_u32, gr, b, 0, s, lines.as<uint32_t>(), NR);
    hipLaunchKernelGGL(k_kh_zero_u32, gr, b, 0, s, tails.as<uint32_t>(), NR);
    hipLaunchKernelGGL(k_kh_place<W>, gn, b, 0, s, d_tk, val_s.as<uint32_t>(), vs.as<uint64_t>(), n, k, rr.as<uint32_t>(), g.S, load, p.as<uint32_t>(), lines.as<uint32_t>(),
                       tails.as<uint32_t>(), d_status);
    CK(kh_scan_u32(lines.as<uint32_t>(), first.as<uint32_t>(), NR, tmp, s));
    hipLaunchKernelGGL(k_kh_kreg, gr, b, 0, s, first.as<uint32_t>(), tails.as<uint32_t>(), lines_cap, d_kreg, d_status);
    hipLaunchKernelGGL(k_kh_write<W>, gn, b, 0, s, d_tk, d_vals, val_s.as<uint32_t>(), p.as<uint32_t>(), n, k, d_kreg, g, d_status, d_kh);
    HIPCK(hipGetLastError());
    return 0;  // (the transients stay in `sc` until the caller has seen `s` drain)
}

BftKhGeometry bft_kh_geometry(int k, uint64_t n_values) {
    BftKhGeometry g;
    g.rb = bft_kh_rb(k);
    g.cb = bft_kh_value_bits(n_values);
    g.S = bft_kh_slots_for(g.rb, g.cb);
    g.f = bft_kh_field_bits(g.S, g.rb);
    g.wb = bft_kh_body_bytes(g.S);
    return g;
}

int bft_kh_build(const uint64_t* d_tk, const uint32_t* d_vals, uint64_t n, int k, int W, const BftKhGeometry& g, uint32_t load, uint64_t* d_kh, uint64_t lines_cap,
                 uint32_t* d_kreg, uint32_t* d_status, BftKhScratch& sc, hipStream_t s) {
    switch (W) {
    case 1: return kh_build_w<1>(d_tk, d_vals, n, k, g, load, d_kh, lines_cap, d_kreg, d_status, sc, s);
    case 2: return kh_build_w<2>(d_tk, d_vals, n, k, g, load, d_kh, lines_cap, d_kreg, d_status, sc, s);
    case 3: return kh_build_w<3>(d_tk, d_vals, n, k, g, load, d_kh, lines_cap, d_kreg, d_status, sc, s);
    default: return kh_build_w<4>(d_tk, d_vals, n, k, g, load, d_kh, lines_cap, d_kreg, d_status, sc, s);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dump ("compact_table": the sorted table comes back from here)
// ---------------------------------------------------------------------------------------------------------------------------------
// Every (k-mer, value) the table holds, in any order: one thread per line finds the line's region (binary search of kreg), rebuilds the
// T-form of every used slot (root prefix from the region, the rest from the slot), reserves places with one atomic per workgroup.
// Word w of the j-th k-mer goes to keys[w * stride + j].
template <int W>
__global__ __launch_bounds__(256) void k_kh_dump(BftImage im, uint64_t* __restrict__ keys, uint64_t stride, uint32_t* __restrict__ vals, unsigned long long* __restrict__ cnt) {
    __shared__ uint32_t s_cnt;
    __shared__ unsigned long long s_base;
    const uint64_t n_lines = im.kreg[BFT_KH_REGIONS] & BFT_KREG_LINE_MASK;
    const uint32_t S = im.kh_S;
    for (uint64_t l0 = (uint64_t)blockIdx.x * blockDim.x; l0 < n_lines; l0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t ln = l0 + threadIdx.x;
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        uint64_t hd[2] = {0, 0};
        uint32_t occ = 0, r = 0;
        const uint64_t* line = im.kh + ln * BFT_KH_LINE_WORDS;
        if (ln < n_lines) {
            bft_kh_load_header(line, hd);
            occ = (uint32_t)(hd[1] >> (64u - S));
            uint32_t lo = 0, hi = BFT_KH_REGIONS;  // last r with kreg[r] <= ln
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((im.kreg[mid] & BFT_KREG_LINE_MASK) <= ln) lo = mid; else hi = mid;
            }
            r = lo;
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
            uint64_t tl[W];
            uint32_t v;
            bft_kh_slot_decode<W>(im, line, hd, s, tl, &v);
            bft_or18_le<W>(tl, 2 * im.k - 18, (uint64_t)r);  // root prefix r at bits [2k - 18, 2k) of the T-form
#pragma unroll
            for (int i = 0; i < W; i++) keys[(uint64_t)i * stride + j] = tl[W - 1 - i];
            vals[j] = v;
            j++;
        }
        __syncthreads();
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
// one-word keys (k <= 32: 0..46 key bits + 1..32 value bits) 10..6 slots, two-word keys 10..3, three-word keys 4..2, four-word keys 2..1.
// ---------------------------------------------------------------------------------------------------------------------------------
#define KH_DISPATCH(W_, S_, CALL)                                                                                                                   \
    switch ((W_) * 16 + (S_)) {                                                                                                                      \
    case 1 * 16 + 10: { constexpr int KW = 1, KS = 10; CALL; } break;                                                                                \
    case 1 * 16 + 9: { constexpr int KW = 1, KS = 9; CALL; } break;                                                                                  \
    case 1 * 16 + 8: { constexpr int KW = 1, KS = 8; CALL; } break;                                                                                  \
    case 1 * 16 + 7: { constexpr int KW = 1, KS = 7; CALL; } break;                                                                                  \
    case 1 * 16 + 6: { constexpr int KW = 1, KS = 6; CALL; } break;                                                                                  \
    case 2 * 16 + 10: { constexpr int KW = 2, KS = 10; CALL; } break;                                                                                \
    case 2 * 16 + 9: { constexpr int KW = 2, KS = 9; CALL; } break;                                                                                  \
    case 2 * 16 + 8: { constexpr int KW = 2, KS = 8; CALL; } break;                                                                                  \
    case 2 * 16 + 7: { constexpr int KW = 2, KS = 7; CALL; } break;                                                                                  \
    case 2 * 16 + 6: { constexpr int KW = 2, KS = 6; CALL; } break;                                                                                  \
    case 2 * 16 + 5: { constexpr int KW = 2, KS = 5; CALL; } break;                                                                                  \
    case 2 * 16 + 4: { constexpr int KW = 2, KS = 4; CALL; } break;                                                                                  \
    case 2 * 16 + 3: { constexpr int KW = 2, KS = 3; CALL; } break;                                                                                  \
    case 3 * 16 + 4: { constexpr int KW = 3, KS = 4; CALL; } break;                                                                                  \
    case 3 * 16 + 3: { constexpr int KW = 3, KS = 3; CALL; } break;                                                                                  \
    case 3 * 16 + 2: { constexpr int KW = 3, KS = 2; CALL; } break;                                                                                  \
    case 4 * 16 + 2: { constexpr int KW = 4, KS = 2; CALL; } break;                                                                                  \
    case 4 * 16 + 1: { constexpr int KW = 4, KS = 1; CALL; } break;                                                                                  \
    default: return bft_fail(BFT_GPU_E_LIMIT, "k-mer hash: no kernel for this key width / slots per line");                                            \
    }

bool bft_kh_has_kernels(int W, uint32_t S) {
    switch (W) {
    case 1: return S >= 6 && S <= 10;
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

int bft_kh_query(const BftImage& im, int grid_mult, const uint8_t* d_kmers, uint64_t n, int rec, uint64_t* d_bits64, uint32_t* d_out32, uint32_t* d_ctr, uint32_t chunk,
                 hipStream_t s) {
    const dim3 block(BFT_KH_BLOCK);
    chunk = std::max(1u, std::min(chunk, BFT_KH_MAX_CLAIM));
    const dim3 grid = kh_round_grid(n, chunk, grid_mult, d_ctr != nullptr);
    KH_DISPATCH(im.W, (int)im.kh_S, hipLaunchKernelGGL((k_query_kh<KW, KS>), grid, block, 0, s, im, d_kmers, n, rec, d_bits64, d_out32, d_ctr, chunk));
    HIPCK(hipGetLastError());
    return 0;
}

int bft_kh_branching(const BftImage& im, const uint8_t* d_kmers, uint64_t n, int B, uint64_t* d_bits64, uint8_t* d_counts, uint32_t* d_ctr, uint32_t chunk, hipStream_t s) {
    const dim3 block(BFT_KH_BLOCK);
    chunk = std::max(1u, std::min(chunk, BFT_KH_BR_MAX_CLAIM));
    const dim3 grid = kh_round_grid(n, chunk, 1, d_ctr != nullptr);
    KH_DISPATCH(im.W, (int)im.kh_S, hipLaunchKernelGGL((k_branching_kh<KW, KS>), grid, block, 0, s, im, d_kmers, n, B, d_bits64, d_counts, d_ctr, chunk));
    HIPCK(hipGetLastError());
    return 0;
}

int bft_kh_seq(const BftImage& im, const uint64_t* d_codes, const uint32_t* d_bad, const uint64_t* d_seq_off, const uint64_t* d_pos_off, const uint32_t* d_tile_seq,
               uint32_t n_seqs, int canonical, uint32_t* d_csout, uint32_t* d_ctr, uint32_t chunk, hipStream_t s) {
    const dim3 grid(256 * 8), block(256);
    chunk = std::max(1u, std::min(chunk, BFT_KH_MAX_CLAIM));
    KH_DISPATCH(im.W, (int)im.kh_S,
                hipLaunchKernelGGL((k_seq_kh<KW, KS>), grid, block, 0, s, im, d_codes, d_bad, d_seq_off, d_pos_off, d_tile_seq, n_seqs, canonical, d_csout, d_ctr, chunk));
    HIPCK(hipGetLastError());
    return 0;
}
