// bft_kernels_seq.h -- sequence queries: k_seq_encode / k_seq_plan / k_seq_walk (window + walk + colour set) / k_seq_tally (counters + threshold per sequence)
// Device code of libbft_gpu.so, included by bft_gpu.hip only (one translation unit: the kernels are templates launched from
// the host code there).
#pragma once
// ---- query_sequence (src/bft.c:1241-1351, harness src/file_io.c:1464-1574): every k-mer of every sequence ----
// ASCII -> 2-bit code (A C G T/U = 0 1 2 3, either case), -1 for anything else; branch-free: bits 1 and 2 of the character code
// already separate the four letters ((c >> 1) ^ (c >> 2)) & 3, and a 21-bit mask over 'A'..'U' says which letters count.
__device__ __forceinline__ int nt_code(char ch) {
    const uint32_t c = (uint8_t)ch, idx = (c & 0xDFu) - 0x41u;  // upper-cased, 'A' = 0
    const uint32_t valid_mask = (1u << 0) | (1u << 2) | (1u << 6) | (1u << 19) | (1u << 20);  // A C G T U
    const bool ok = idx < 21u && ((valid_mask >> idx) & 1u);
    return ok ? (int)(((c >> 1) ^ (c >> 2)) & 3u) : -1;
}

// Sequence queries, step 0.  The ASCII blob -> 2 bits per character (32 characters per u64, character c at bits 2(c%32) of
// word c/32: the packed layout of src/fasta.c:11-23 continued over the whole blob) + one "not ACGTU" bit per character.
// One thread per 32 characters; characters past n_chars count as 'A' / good (no window of a sequence reaches them).  A blob that
// is 16-byte aligned is read 32 bytes at a time, any other one byte by byte.
__global__ void k_seq_encode(const char* __restrict__ seqs, uint64_t n_chars, uint64_t n_words, uint64_t* __restrict__ codes, uint32_t* __restrict__ bad) {
    const bool aligned = ((uintptr_t)seqs & 15u) == 0;
    for (uint64_t wi = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; wi < n_words; wi += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t d[8];
        if (aligned && wi * 32 + 32 <= n_chars) {
            const uint4* src = (const uint4*)(seqs + wi * 32);
            const uint4 a = src[0], b = src[1];
            d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint32_t v = 0;
                for (int c = 0; c < 4; c++) {
                    const uint64_t at = wi * 32 + 4 * j + c;
                    v |= (uint32_t)(uint8_t)(at < n_chars ? seqs[at] : 'A') << (8 * c);
                }
                d[j] = v;
            }
        }
        uint64_t cw = 0;
        uint32_t bw = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int code = nt_code((char)((d[j] >> (8 * c)) & 0xFFu));
                const int i = 4 * j + c;
                cw |= (uint64_t)(code & 3) << (2 * i);
                bw |= (code < 0 ? 1u : 0u) << i;
            }
        }
        codes[wi] = cw;
        bad[wi] = bw;
    }
}

#include "bft_kernels_seqwin.h"

// ---- plan (positions per sequence) -> window + walk + colour set -> counters ------------------------------------------------
// k_seq_plan: k-mer positions of every sequence of a chunk, on the device (the device-resident entry point never sees the offsets
// on the host): npos[s] = max(len - k + 1, 0).  An exclusive scan of npos gives pos_off.
__global__ void k_seq_plan(const uint64_t* __restrict__ seq_off, uint64_t n_seqs, int k, uint64_t* __restrict__ npos) {
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s <= n_seqs; s += (uint64_t)gridDim.x * blockDim.x) {
        if (s == n_seqs) { npos[s] = 0; continue; }  // the scan's last element: the total
        const uint64_t len = seq_off[s + 1] - seq_off[s];
        npos[s] = len >= (uint64_t)k ? len - (uint64_t)k + 1 : 0;
    }
}

// k_seq_tiles: the sequence of the first position of every 64-position tile (last s with pos_off[s] <= 64 t), one binary search
// per tile, once -- the wavefronts of k_seq_walk start from there with one load.  (A search per wavefront pass
// in that kernel was measured: a scalar binary search costs 20 dependent loads on the critical path of every pass, a 64-ary
// wavefront-wide search 256 L2 requests per pass -- the path went from 4.9 to 8 ms per 10^6 reads with it.)
__global__ void k_seq_tiles(const uint64_t* __restrict__ pos_off, uint32_t n_seqs, uint32_t* __restrict__ tile_seq) {
    const uint64_t P = pos_off[n_seqs], ntiles = (P + 63) / 64;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < ntiles; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p0 = t * 64;
        uint32_t lo = 0, hi = n_seqs;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pos_off[mid] <= p0) lo = mid; else hi = mid;
        }
        tile_seq[t] = lo;
    }
}

// k_seq_walk8 / k_seq_walk6: one lane per k-mer position of the chunk, persistent grid like k_query (same LDS staging).
// A wavefront holds 64 consecutive positions: sequence of its first position from k_seq_tiles, then each lane steps forward;
// window from the code stream; bft_walk; row -> colour set (one gather); the colour set of the position (0xFFFFFFFF: no
// k-mer there, or absent) goes to csout[p] -- 4 bytes per position, the only per-position array of the path.
// (Counting inside this kernel was tried: the counter lines and the range table are pushed out of the L2 by the walk's gathers,
// 2.4 L2 misses per position instead of ~1.9, 6.0 ms instead of 4.2 per 10^6 reads of 150 nt.)
template <int W, int BLOCK, bool STAGED, int PROBE>
__device__ __forceinline__ void seq_walk_body(const BftImage& im, const uint64_t* __restrict__ codes, const uint32_t* __restrict__ bad,
                                              const uint64_t* __restrict__ seq_off, const uint64_t* __restrict__ pos_off, const uint32_t* __restrict__ tile_seq,
                                              uint32_t n_seqs, int canonical, uint32_t* __restrict__ csout) {
    extern __shared__ __align__(16) uint8_t lds[];
    uint32_t* l_hm = (uint32_t*)lds;
    uint8_t* l_bf = lds + BFT_LDS_HM_BYTES;
    const BftNode root = im.nodes[0];
    const bool stage_root = STAGED && im.rdir == nullptr;
    const uint32_t bf_bytes = stage_root ? ((BFT_MODULO_HASH * (uint32_t)root.bf_wb + 15u) & ~15u) : 0u;
    BftCCX* l_cc = (BftCCX*)(l_bf + bf_bytes);
    {
        const uint4* g = (const uint4*)im.hashmod;
        uint4* l = (uint4*)l_hm;
        for (uint32_t i = threadIdx.x; i < BFT_LDS_HM_BYTES / 16; i += BLOCK) l[i] = g[i];
        if (stage_root) {
            const uint64_t* gb = (const uint64_t*)(im.bfT + (size_t)root.bf_off * 8);
            uint64_t* lb = (uint64_t*)l_bf;
            const uint32_t nb8 = (BFT_MODULO_HASH * (uint32_t)root.bf_wb) / 8;
            for (uint32_t i = threadIdx.x; i < nb8; i += BLOCK) lb[i] = gb[i];
            for (uint32_t i = threadIdx.x; i < root.ncc; i += BLOCK) l_cc[i] = im.ccx[root.cc_first + i];
        }
    }
    __syncthreads();
    const BftRootLds<STAGED> acc{im, l_hm, l_bf, l_cc};
    const uint64_t P = pos_off[n_seqs];
    const uint64_t nblk = (P + BLOCK - 1) / BLOCK;
    for (uint64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {  // whole wavefronts stay in the loop together
        const uint64_t p = blk * BLOCK + threadIdx.x;
        if (p >= P) continue;
        uint32_t lo = tile_seq[p >> 6];
        while (lo + 1 < n_seqs && pos_off[lo + 1] <= p) lo++;  // (sequences shorter than k own no position and are stepped over)
        uint32_t cs = 0xFFFFFFFFu;
        uint64_t x[W], t[W];
        if (seq_window<W>(codes, bad, seq_off[lo] + (p - pos_off[lo]), im.k, canonical, x)) {
            bft_tform_from_x<W>(x, im.k, t);
            const BftHit h = bft_walk<W, BftRootLds<STAGED>, PROBE>(im, acc, root, t);
            if (h.present) cs = bft_hit_out(im, h);  // (im.emit_cs is set for this launch)
        }
        csout[p] = cs;
    }
}

// Two builds (k_query8 / k_query6): the walk of the two-word rows needs 70 VGPRs (6 wavefronts per SIMD), the others fit 8 x 64.
template <int W, bool STAGED, int PROBE>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_seq_walk8(
    BftImage im, const uint64_t* __restrict__ codes, const uint32_t* __restrict__ bad, const uint64_t* __restrict__ seq_off, const uint64_t* __restrict__ pos_off,
    const uint32_t* __restrict__ tile_seq, uint32_t n_seqs, int canonical, uint32_t* __restrict__ csout) {
    seq_walk_body<W, 1024, STAGED, PROBE>(im, codes, bad, seq_off, pos_off, tile_seq, n_seqs, canonical, csout);
}
template <int W, bool STAGED, int PROBE>
__global__ __launch_bounds__(BFT_BLOCK6) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_seq_walk6(
    BftImage im, const uint64_t* __restrict__ codes, const uint32_t* __restrict__ bad, const uint64_t* __restrict__ seq_off, const uint64_t* __restrict__ pos_off,
    const uint32_t* __restrict__ tile_seq, uint32_t n_seqs, int canonical, uint32_t* __restrict__ csout) {
    seq_walk_body<W, BFT_BLOCK6, STAGED, PROBE>(im, codes, bad, seq_off, pos_off, tile_seq, n_seqs, canonical, csout);
}

// Per-(sequence, genome) counters and the threshold, one WAVEFRONT per sequence, counters in LDS: no counter matrix in HBM, no
// global atomics (the first version added run lengths into an n_seqs x G matrix with device-scope atomics -- 0.8 per k-mer
// position, 98 M per 10^6 reads, half of the path's time -- and thresholded it in another pass).  Consecutive k-mers of a read
// mostly carry the same colour set, so counting works on runs: the 64 lanes hold 64 consecutive positions, run boundaries come
// from a shuffle + __ballot, and the first lane of every run adds the run length once per genome of the set.  Genomes are
// handled a window at a time (at most SEQ_TALLY_G = 2048, a multiple of 8; one pass over the sequence's positions per window).
// Row s of `out` (rowbytes bytes): bit g set iff genome g holds at least ceil(npos(s) * threshold) > 0 of the sequence's k-mers
// (src/bft.c:1281, :1320-1340).
#define SEQ_TALLY_G 2048u
#define SEQ_TALLY_WAVES 4
__global__ __launch_bounds__(64 * SEQ_TALLY_WAVES) void k_seq_tally(const uint32_t* __restrict__ csin, const uint64_t* __restrict__ pos_off, uint32_t n_seqs,
                                                                   const uint32_t* __restrict__ cs_off, const void* __restrict__ cs_ids, uint32_t cs_w, uint32_t G,
                                                                   uint32_t rowbytes, double threshold, uint32_t win, uint8_t* __restrict__ out) {
    extern __shared__ uint32_t s_cnt[];  // [SEQ_TALLY_WAVES][win]: win = the genome window, sized by the host (few genomes: more workgroups per CU)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* cnt = s_cnt + (size_t)wave * win;
    for (uint64_t s = (uint64_t)blockIdx.x * SEQ_TALLY_WAVES + wave; s < n_seqs; s += (uint64_t)gridDim.x * SEQ_TALLY_WAVES) {
        const uint64_t pa = pos_off[s], m = pos_off[s + 1] - pa;
        const uint64_t minv = (uint64_t)(int64_t)ceil((double)m * threshold);  // nb_kmers_query_min
        for (uint32_t win0 = 0; win0 < G; win0 += win) {
            const uint32_t wn = min(win, G - win0);
            for (uint32_t j = lane; j < wn; j += 64) cnt[j] = 0;
            // Two passes of 64 positions per turn, their loads issued together -- colour sets of both, then the list bounds of both, then the
            // first eight ids of both: a 150-nt read is one turn of three dependent round trips where pass after pass made six (1.33 -> 1.05 ms
            // per 10^6 reads).  Two SEQUENCES per turn as well were measured: 1.17 ms -- the chain's latency is no longer what is left.
            for (uint64_t base = 0; base < m; base += 128) {  // m is wavefront-uniform: every lane takes every turn
                uint32_t cs[2], len[2], qa[2], qb[2];
                bool act[2];
#pragma unroll
                for (int u = 0; u < 2; u++) cs[u] = base + 64u * u + lane < m ? csin[pa + base + 64u * u + lane] : 0xFFFFFFFFu;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t pcs = __shfl_up(cs[u], 1);
                    const bool boundary = lane == 0 || pcs != cs[u];
                    const uint64_t bmask = __ballot(boundary);
                    const uint64_t above = lane == 63 ? 0ull : bmask >> (lane + 1);
                    len[u] = above ? (uint32_t)__builtin_ctzll(above) + 1u : 64u - lane;
                    act[u] = boundary && cs[u] != 0xFFFFFFFFu;
                    qa[u] = qb[u] = 0;
                }
#pragma unroll
                for (int u = 0; u < 2; u++)
                    if (act[u]) { qa[u] = cs_off[cs[u]]; qb[u] = cs_off[cs[u] + 1]; }
                uint32_t id[2][8];
#pragma unroll
                for (int u = 0; u < 2; u++) {
#pragma unroll
                    for (int j = 0; j < 8; j++) id[u][j] = act[u] && qb[u] > qa[u] ? bft_cs_id_at(cs_ids, cs_w, min(qa[u] + (uint32_t)j, qb[u] - 1u)) : 0u;
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (!act[u]) continue;
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (qa[u] + (uint32_t)j < qb[u] && id[u][j] - win0 < wn) atomicAdd(&cnt[id[u][j] - win0], len[u]);
                    for (uint32_t q = qa[u] + 8u; q < qb[u]; q += 8) {  // (sets of more than eight genomes: eight ids per step, loaded together)
                        uint32_t more[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) more[j] = bft_cs_id_at(cs_ids, cs_w, min(q + (uint32_t)j, qb[u] - 1u));
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if (q + (uint32_t)j < qb[u] && more[j] - win0 < wn) atomicAdd(&cnt[more[j] - win0], len[u]);
                    }
                }
            }
            // (LDS operations of one wavefront complete in order: the adds above are visible to the reads below)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (uint32_t b = lane; b < (wn + 7u) / 8u; b += 64) {
                uint32_t v = 0;
                for (uint32_t j = 0; j < 8 && b * 8 + j < wn; j++) {
                    const uint32_t c = cnt[b * 8 + j];
                    if (c && c >= minv) v |= 1u << j;
                }
                out[s * rowbytes + win0 / 8 + b] = (uint8_t)v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}
