// bft_kh.hip -- the queries of include/bft_gpu.h through the k-mer hash (bft_image.h, BFT_KH_*): fill, presence / colour set, branching,
// sequence positions.  Its own translation unit: these kernels stage nothing and walk nothing -- T-form, home line, compare.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "bft_dev.h"
#include "bft_image.h"
#include "bft_kh.h"
#include "bft_claims.h"
#include "bft_walk.h"
#include "bft_kernels_load.h"
#include "bft_kernels_seqwin.h"

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
template <int W>
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
        const uint64_t i = cl.blk * BFT_KH_BLOCK + threadIdx.x;
        bool present = false;
        uint32_t val = 0xFFFFFFFFu;
        if (i < n) {
            uint64_t x[W], t[W];
            load_x<W>(packed, i, B, end_aligned, x);
            bft_tform_from_x<W>(x, im.k, t);
            uint64_t at = 0;
            present = bft_kh_find<W>(im.kh, im.kh_lines, t, at);
            if (present && out32) val = reinterpret_cast<const uint32_t*>(im.kh)[at];
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

// How many of four candidate k-mers are stored: the four home lines are loaded before any is looked at -- four independent
// misses in flight instead of four dependent walks (src/presenceNode.c:15-1211 shares one descent between the four; here there
// is no descent to share).  A candidate whose home line is full without holding it continues line by line (a few per cent).
template <int W>
__device__ __forceinline__ int kh_count4(const BftImage& im, const uint64_t (*cand)[W]) {
    constexpr uint32_t S = BFT_KH_SLOTS(W);
    uint64_t ln[4], key[4][S][W];
#pragma unroll
    for (int v = 0; v < 4; v++) ln[v] = bft_kh_home<W>(cand[v], im.kh_lines);
#pragma unroll
    for (int v = 0; v < 4; v++) bft_kh_load_keys<W>(im.kh + ln[v] * BFT_KH_LINE_WORDS, key[v]);
    int count = 0;
#pragma unroll
    for (int v = 0; v < 4; v++) {
        bool hit = false, free_slot = false;
#pragma unroll
        for (uint32_t s = 0; s < S; s++) {
            hit = hit || bft_cmp<W>(key[v][s], cand[v]) == 0;
            free_slot = free_slot || key[v][s][0] == BFT_KH_EMPTY;
        }
        if (!hit && !free_slot) {  // full line without the key: the general lookup walks on from the home line
            uint64_t at = 0;
            hit = bft_kh_find<W>(im.kh, im.kh_lines, cand[v], at);
        }
        count += hit;
    }
    return count;
}

// Batched isBranchingRight / isBranchingLeft (src/branchingNode.c:16-112, :240-340; loop of src/file_io.c:943-998), see branching_body.
// Rounds of blocks as in k_query_kh; the branching bits and the neighbour counts of a round leave LDS as whole lines.
#define BFT_KH_BR_MAX_CLAIM 16u
template <int W>
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
            const int cr = kh_count4<W>(im, cand);
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
                cl_ = kh_count4<W>(im, cand);
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

// Fills the table: one thread per stored k-mer claims the first free slot at or after its home line with a compare-and-swap on the
// slot's first key word (a key word is never all ones: bft_kh_usable), then writes the rest of the key and the value.  Nothing
// reads the table before the kernel is done.
template <int W>
__global__ __launch_bounds__(256) void k_kh_insert(const uint64_t* __restrict__ tk, const uint32_t* __restrict__ tcol, uint64_t n, uint64_t* __restrict__ kh,
                                                   uint64_t n_lines) {
    constexpr uint32_t S = BFT_KH_SLOTS(W);
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t t[W];
        bft_load_row<W>(tk + i * W, t);
        const uint32_t val = tcol[i];
        uint64_t ln = bft_kh_home<W>(t, n_lines);
        bool placed = false;
        while (!placed) {
            uint64_t* line = kh + ln * BFT_KH_LINE_WORDS;
            // An ordinary, cached look at the slots' first key words picks the candidates: a stale view can only show a taken slot as
            // free, never the reverse, and the compare-and-swap decides.  (Agent-scope atomic loads slot by slot -- each an L2
            // transaction of its own -- made the fill 2.9 ms on config 3; all five at once, 4.3 ms.)
            uint64_t view[S];
#pragma unroll
            for (uint32_t s = 0; s < S; s++) view[s] = line[s * W];
#pragma unroll
            for (uint32_t s = 0; s < S; s++) {
                if (placed || view[s] != BFT_KH_EMPTY) continue;
                unsigned long long* slot = (unsigned long long*)(line + s * W);
                if (atomicCAS(slot, (unsigned long long)BFT_KH_EMPTY, (unsigned long long)t[0]) == BFT_KH_EMPTY) {
#pragma unroll
                    for (int w = 1; w < W; w++) line[s * W + w] = t[w];
                    reinterpret_cast<uint32_t*>(line + S * W)[s] = val;
                    placed = true;
                }
            }
            ln = ln + 1 == n_lines ? 0 : ln + 1;
        }
    }
}

// The same through the k-mer hash (BFT_KH_*): the colour set of a position sits in the cache line that says the k-mer is stored --
// one line per position, nothing staged.
template <int W>
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
            if (bft_kh_lookup<W>(im.kh, im.kh_lines, t, &val)) cs = val;
        }
        csout[p] = cs;
    }
    cl.done();
}


// ---- launchers (called from bft_gpu.hip) ------------------------------------------------------------------------------------
// Every (key, value) the table holds, in any order: each workgroup counts the occupied slots of its share of the lines, reserves that
// many places with one atomic, writes (word w of the j-th key at keys[w * stride + j]).  The "compact_table" option rebuilds the sorted
// table from this.
template <int W>
__global__ __launch_bounds__(256) void k_kh_dump(const uint64_t* __restrict__ kh, uint64_t n_lines, uint64_t* __restrict__ keys, uint64_t stride, uint32_t* __restrict__ vals,
                                                 unsigned long long* __restrict__ cnt) {
    constexpr uint32_t S = BFT_KH_SLOTS(W);
    __shared__ uint32_t s_cnt;
    __shared__ unsigned long long s_base;
    const uint64_t per = (n_lines + gridDim.x - 1) / gridDim.x, l0 = blockIdx.x * per, l1 = min(n_lines, l0 + per);
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint64_t ln = l0 + threadIdx.x; ln < l1; ln += blockDim.x) {
        const uint64_t* line = kh + ln * BFT_KH_LINE_WORDS;
#pragma unroll
        for (uint32_t sl = 0; sl < S; sl++) mine += line[sl * W] != BFT_KH_EMPTY;
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) { s_base = s_cnt ? atomicAdd(cnt, (unsigned long long)s_cnt) : 0ull; s_cnt = 0; }
    __syncthreads();
    for (uint64_t ln = l0 + threadIdx.x; ln < l1; ln += blockDim.x) {
        const uint64_t* line = kh + ln * BFT_KH_LINE_WORDS;
        const uint32_t* v = reinterpret_cast<const uint32_t*>(line + S * W);
#pragma unroll
        for (uint32_t sl = 0; sl < S; sl++) {
            if (line[sl * W] != BFT_KH_EMPTY) {
                const uint64_t j = s_base + atomicAdd(&s_cnt, 1u);
#pragma unroll
                for (int w = 0; w < W; w++) keys[(uint64_t)w * stride + j] = line[sl * W + w];
                vals[j] = v[sl];
            }
        }
    }
}

int bft_kh_dump(const uint64_t* d_kh, uint64_t n_lines, int W, uint64_t* d_keys, uint64_t stride, uint32_t* d_vals, unsigned long long* d_cnt, hipStream_t s) {
    if (W == 1) hipLaunchKernelGGL(k_kh_dump<1>, dim3(256 * 8), dim3(256), 0, s, d_kh, n_lines, d_keys, stride, d_vals, d_cnt);
    else hipLaunchKernelGGL(k_kh_dump<2>, dim3(256 * 8), dim3(256), 0, s, d_kh, n_lines, d_keys, stride, d_vals, d_cnt);
    HIPCK(hipGetLastError());
    return 0;
}

int bft_kh_fill(const uint64_t* d_tk, const uint32_t* d_tcol, uint64_t n, int W, uint64_t* d_kh, uint64_t n_lines, hipStream_t s) {
    // (one key per thread, not a persistent grid: the fill runs beside the container assembly on a low-priority stream, and a
    // workgroup that ends gives its CU slots to the assembly's next small kernel)
    const dim3 grid((unsigned)std::min<uint64_t>((n + 255) / 256, 0x7FFFFFFFull)), block(256);
    if (W == 1) hipLaunchKernelGGL(k_kh_insert<1>, grid, block, 0, s, d_tk, d_tcol, n, d_kh, n_lines);
    else hipLaunchKernelGGL(k_kh_insert<2>, grid, block, 0, s, d_tk, d_tcol, n, d_kh, n_lines);
    HIPCK(hipGetLastError());
    return 0;
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
    if (im.W == 1) hipLaunchKernelGGL(k_query_kh<1>, grid, block, 0, s, im, d_kmers, n, rec, d_bits64, d_out32, d_ctr, chunk);
    else hipLaunchKernelGGL(k_query_kh<2>, grid, block, 0, s, im, d_kmers, n, rec, d_bits64, d_out32, d_ctr, chunk);
    HIPCK(hipGetLastError());
    return 0;
}

int bft_kh_branching(const BftImage& im, const uint8_t* d_kmers, uint64_t n, int B, uint64_t* d_bits64, uint8_t* d_counts, uint32_t* d_ctr, uint32_t chunk, hipStream_t s) {
    const dim3 block(BFT_KH_BLOCK);
    chunk = std::max(1u, std::min(chunk, BFT_KH_BR_MAX_CLAIM));
    const dim3 grid = kh_round_grid(n, chunk, 1, d_ctr != nullptr);
    if (im.W == 1) hipLaunchKernelGGL(k_branching_kh<1>, grid, block, 0, s, im, d_kmers, n, B, d_bits64, d_counts, d_ctr, chunk);
    else hipLaunchKernelGGL(k_branching_kh<2>, grid, block, 0, s, im, d_kmers, n, B, d_bits64, d_counts, d_ctr, chunk);
    HIPCK(hipGetLastError());
    return 0;
}

int bft_kh_seq(const BftImage& im, const uint64_t* d_codes, const uint32_t* d_bad, const uint64_t* d_seq_off, const uint64_t* d_pos_off, const uint32_t* d_tile_seq,
               uint32_t n_seqs, int canonical, uint32_t* d_csout, uint32_t* d_ctr, uint32_t chunk, hipStream_t s) {
    const dim3 grid(256 * 8), block(256);
    chunk = std::max(1u, std::min(chunk, BFT_KH_MAX_CLAIM));
    if (im.W == 1) hipLaunchKernelGGL(k_seq_kh<1>, grid, block, 0, s, im, d_codes, d_bad, d_seq_off, d_pos_off, d_tile_seq, n_seqs, canonical, d_csout, d_ctr, chunk);
    else hipLaunchKernelGGL(k_seq_kh<2>, grid, block, 0, s, im, d_codes, d_bad, d_seq_off, d_pos_off, d_tile_seq, n_seqs, canonical, d_csout, d_ctr, chunk);
    HIPCK(hipGetLastError());
    return 0;
}
