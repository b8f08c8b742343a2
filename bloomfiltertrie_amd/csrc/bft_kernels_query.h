// bft_kernels_query.h -- T-form conversion into the insert log; k_query / k_query8 / k_query6 and k_branching* (presence and branching as a container walk;
// the same queries through the k-mer hash: bft_kh.hip)
// Device code of libbft_gpu.so, included by bft_gpu.hip only (one translation unit: the kernels are templates launched from
// the host code there).
#pragma once
// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------

#include "bft_kernels_load.h"
#include "bft_claims.h"
#include "bft_kh_dev.h"

template <int W>
__global__ __launch_bounds__(BFT_BLOCK) void k_pack_to_tform(const uint8_t* __restrict__ packed, uint64_t n, int B, int k,
                                                             uint64_t* __restrict__ out, uint64_t stride, uint64_t off,
                                                             uint32_t* __restrict__ gout, uint32_t gid, uint32_t cgb) {
    // cgb != 0 (one-word keys only): the log holds COMPOSITES T << cgb | genome -- what the build's root-prefix split sorts -- and no id array
    const uint64_t end_aligned = ((uint64_t)packed + n * (uint64_t)B) & ~3ull;
    for (uint64_t i = blockIdx.x * (uint64_t)BFT_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BFT_BLOCK) {
        uint64_t x[W], t[W];
        load_x<W>(packed, i, B, end_aligned, x);
        bft_tform_from_x<W>(x, k, t);
        if (W == 1 && cgb) {
            out[off + i] = (t[0] << cgb) | (uint64_t)gid;
            continue;
        }
#pragma unroll
        for (int w = 0; w < W; w++) out[(uint64_t)w * stride + off + i] = t[w];
        gout[off + i] = gid;
    }
}

// The hash table (64 KiB) and the root node's bit-sliced Bloom block and CC headers live in LDS:
// every query of a batch reads them, and a divergent LDS read costs a few cycles where a divergent
// vector-memory read occupies the CU's texture path for ~64.
template <bool STAGED>
struct BftRootLds {
    const BftImage& im;
    const uint32_t* hm;
    const uint8_t* bf;
    const BftCCX* cc;
    __device__ __forceinline__ uint32_t hashmod(uint32_t key) const { return hm[key]; }
    __device__ __forceinline__ int root_first_cc(const BftNode& nd, uint32_t h1, uint32_t h2) const {
        if (STAGED) return bft_first_cc_blk(bf, nd.bf_wb, h1, h2);
        return bft_first_cc_blk(im.bfT + (size_t)nd.bf_off * 8, nd.bf_wb, h1, h2);
    }
    __device__ __forceinline__ BftCCX root_cc(const BftNode& nd, int c) const {
        if (STAGED) return cc[c];
        return im.ccx[nd.cc_first + c];
    }
};

#define BFT_LDS_HM_BYTES 65536u
#define BFT_LDS_ROOT_MAX_CC 64u

// The batch is dealt out WAVEFRONT by wavefront, in chunks of 1024 k-mers = 16 passes of 64 = the 16 presence words of one 128-byte line
// of the bitmap: the first chunk of a wavefront by its number, the others claimed from the stream's counter (NULL: all by number; see bft_claims.h for
// why batches are claimed at all).  A wavefront gathers the words of its chunk in its own 128 bytes of LDS and stores the line with one
// instruction.  Nothing here makes the wavefronts of a workgroup wait for each other: a barrier per pass costs this latency-bound kernel
// 10-30 %, and rounds of four blocks per WORKGROUP (three barriers per round, the scheme of k_query_kh) took 7.6 ms where this takes
// (measured, profiles/r04/probe_walk.jsonl) -- a wavefront of the walk is as slow as its slowest lane, a workgroup would be as slow as its
// slowest wavefront.
#define BFT_WALK_PASSES 16u   // passes of 64 k-mers per chunk: one 128-byte line of presence bits

// KS > 0 (with WKH: the kernel of "walk_hash", bft_walkh.hip): the slots per line of the k-mer hash.  Plain root groups are then looked up
// HERE, all lanes of the wavefront together -- the line fetched by the quad and scanned in LDS like k_query_kh's (bft_kh_dev.h) -- and only the
// special prefixes walk; the 64 KiB hash table of the Bloom filters stays in global memory (the few parked lanes read it through the L2) and
// its LDS holds the wavefronts' lines.
template <int W, int BLOCK, bool STAGED, int PROBE, bool WKH = false, int KS = 0>
__device__ __forceinline__ void query_body(const BftImage& im, const uint8_t* __restrict__ packed, uint64_t n, int B,
                                           uint64_t* __restrict__ bits64, uint32_t* __restrict__ rows, BftClaimCtr cc) {
    extern __shared__ __align__(16) uint8_t lds[];
    uint32_t* l_hm = KS > 0 ? const_cast<uint32_t*>(im.hashmod) : (uint32_t*)lds;
    uint8_t* l_bf = KS > 0 ? lds : lds + BFT_LDS_HM_BYTES;
    constexpr uint32_t WPW = BLOCK / 64;  // wavefronts per workgroup
    __shared__ uint64_t s_bits[WPW][BFT_WALK_PASSES];
    __shared__ uint4 s_lines[KS > 0 ? WPW : 1][KS > 0 ? 64 * BFT_KH_LDS_LINE : 1];
    uint4* const wave_lines = s_lines[KS > 0 ? threadIdx.x >> 6 : 0];
    const BftNode root = im.nodes[0];
    // the root's Bloom block and CC headers are only read when the root level goes through the containers: with the derived
    // root tables (im.rdir) that LDS space holds the queue of deferred lanes instead (below)
    const bool stage_root = STAGED && im.rdir == nullptr;  // host side: root.ncc in [1, 64]
    const uint32_t bf_bytes = stage_root ? ((BFT_MODULO_HASH * (uint32_t)root.bf_wb + 15u) & ~15u) : 0u;
    BftCCX* l_cc = (BftCCX*)(l_bf + bf_bytes);
    {
        if (KS == 0) {
            const uint4* g = (const uint4*)im.hashmod;
            uint4* l = (uint4*)l_hm;
            for (uint32_t i = threadIdx.x; i < BFT_LDS_HM_BYTES / 16; i += BLOCK) l[i] = g[i];
        }
        if (stage_root) {
            const uint64_t* gb = (const uint64_t*)(im.bfT + (size_t)root.bf_off * 8);
            uint64_t* lb = (uint64_t*)l_bf;
            const uint32_t nb8 = (BFT_MODULO_HASH * (uint32_t)root.bf_wb) / 8;  // 1504*wb is a multiple of 8
            for (uint32_t i = threadIdx.x; i < nb8; i += BLOCK) lb[i] = gb[i];
            for (uint32_t i = threadIdx.x; i < root.ncc; i += BLOCK) l_cc[i] = im.ccx[root.cc_first + i];
        }
    }
    __syncthreads();
    const BftRootLds<STAGED> acc{im, l_hm, l_bf, l_cc};
    const uint64_t end_aligned = ((uint64_t)packed + n * (uint64_t)B) & ~3ull;
    constexpr uint64_t CHUNK = 64ull * BFT_WALK_PASSES;
    const uint64_t n_chunks = (n + CHUNK - 1) / CHUNK, nwords = (n + 63) / 64;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t n_waves = (uint64_t)gridDim.x * WPW;
    unsigned long long* ctr = n_waves >= n_chunks ? nullptr : cc.p;  // (the first round covers the batch: no counter involved)
    if (ctr && lane == 0) atomicMax(ctr, cc.base);  // (bft_claims.h: the counter enters this launch's range before the wavefront's first claim)
    uint64_t chunk = (uint64_t)blockIdx.x * WPW + wave;
    uint32_t next_claim = 0;
    volatile uint64_t* my_bits = s_bits[wave];
    const bool with_queue = im.rstart != nullptr || im.walk_kh;
    // With the range table: every lane takes the short path (range table -> suffix group); a lane whose root prefix is "special"
    // (child Node, UC rows: the long container path) only parks its k-mer in its wavefront's queue in LDS, and the wavefront walks
    // the queue when it is full.  A wavefront is as slow as its slowest lane, and with ~5 % special prefixes (config 4) nearly
    // every wavefront had one: parked, the long path is walked by (nearly) full wavefronts, once per ~20 passes.  A parked lane's
    // answer is OR-ed into its presence word: in the wavefront's LDS words while its chunk is still being gathered there, else in the
    // bitmap itself (the line was stored earlier by this very wavefront: same wavefront, same address, in order).
    constexpr uint32_t QCAP = W == 1 ? 64u : (W == 2 ? 32u : 16u);   // entries per wavefront: 16 x QCAP x (8 W + 4) bytes <= the 14 KiB root area
    uint64_t* q_t = (uint64_t*)l_bf + (size_t)wave * QCAP * W;                          // [QCAP * W] parked T-forms of this wavefront
    uint32_t* q_i = (uint32_t*)((uint64_t*)l_bf + (size_t)(BLOCK / 64) * QCAP * W) + (size_t)wave * QCAP;  // [QCAP] their query (offset from qbase)
    uint32_t qn = 0;          // entries parked (wavefront-uniform)
    uint64_t qbase = 0;       // queries are parked as 32-bit offsets from the first query of the pass the queue was last empty in
    auto answer_late = [&](uint64_t i, const BftHit& h, bool chunk_open) {
        if (h.present) {
            if (chunk_open && i / CHUNK == chunk) atomicOr((unsigned long long*)&s_bits[wave][(i >> 6) % BFT_WALK_PASSES], 1ull << (i & 63u));
            else atomicOr((unsigned long long*)&bits64[i >> 6], 1ull << (i & 63u));
        }
        if (rows) rows[i] = h.present ? bft_hit_out(im, h) : BFT_ABSENT_ROW;
    };
    auto drain = [&](bool chunk_open) {
        if (lane < qn) {
            uint64_t t[W];
#pragma unroll
            for (int w = 0; w < W; w++) t[w] = q_t[(size_t)lane * W + w];
            const uint64_t i = qbase + q_i[lane];
            const BftHit h = bft_walk<W, BftRootLds<STAGED>, PROBE, 2>(im, acc, root, t);
            answer_late(i, h, chunk_open);
        }
        qn = 0;
    };
    while (chunk < n_chunks) {
        for (uint32_t pass = 0; pass < BFT_WALK_PASSES; pass++) {
            const uint64_t q0 = chunk * CHUNK + (uint64_t)pass * 64;  // first query of this pass
            if (q0 >= n) break;
            const uint64_t i = q0 + lane;
            int present = 0;
            bool parked = false;
            uint64_t t[W];
            if constexpr (WKH && KS > 0) {
                bool hashed = false;
                BftKhKey<W> key;
                key.home = 0; key.field = 0;
#pragma unroll
                for (int w = 0; w < W; w++) { key.body[w] = 0; key.bmask[w] = 0; }
                if (i < n) {
                    uint64_t x[W];
                    load_x<W>(packed, i, B, end_aligned, x);
                    bft_tform_from_x<W>(x, im.k, t);
                    const uint32_t r = bft_digit<W>(t, im.k, 0);
                    if ((im.rspec[r >> 5] >> (r & 31u)) & 1u) parked = true;  // child Node / UC rows under this prefix: the containers
                    else { hashed = true; bft_kh_key<W>(t, im.k, im.kh, key); }
                }
                kh_fetch_quad(im, key.home, hashed, wave_lines);
                if (hashed) {
                    uint32_t val = 0;
                    int res = kh_lds_scan<W, KS>(im, wave_lines + lane * BFT_KH_LDS_LINE, key, 0u, &val);
                    for (uint32_t d = 1; res < 0 && d <= im.kh.maxd; d++) {
                        const uint64_t* line = im.kh_lines + (key.home + d) * BFT_KH_LINE_WORDS;
                        uint64_t hd[2];
                        bft_kh_load_header(line, hd);
                        res = bft_kh_scan<W, KS>(im, line, hd, key, d, &val);
                    }
                    if (res < 0 && im.kh_ovf_n) res = bft_kh_overflow_find<W>(im, t, &val) ? 1 : 0;
                    present = res > 0;
                    if (rows) rows[i] = present ? (im.emit_cs ? val : 0u) : BFT_ABSENT_ROW;
                }
            } else
            if (i < n) {
                uint64_t x[W];
                load_x<W>(packed, i, B, end_aligned, x);
                bft_tform_from_x<W>(x, im.k, t);
                if (with_queue) {
                    const BftHit h = bft_walk<W, BftRootLds<STAGED>, PROBE, 1, WKH>(im, acc, root, t);
                    if (h.present == BFT_HIT_DEFERRED) parked = true;
                    else {
                        present = h.present;
                        if (rows) rows[i] = present ? bft_hit_out(im, h) : BFT_ABSENT_ROW;
                    }
                } else {  // no range table: every lane walks its k-mer to the end
                    const BftHit h = bft_walk<W, BftRootLds<STAGED>, PROBE>(im, acc, root, t);
                    present = h.present;
                    if (rows) rows[i] = present ? bft_hit_out(im, h) : BFT_ABSENT_ROW;
                }
            }
            const uint64_t mask = __ballot(present);
            if (lane == 0) my_bits[pass] = mask;
            // the claim for the next chunk travels while this one is answered (sent after the first pass: at the start of a launch
            // every wavefront would ask at the same instant)
            if (pass == 0 && ctr && lane == 0) next_claim = (uint32_t)(atomicAdd(ctr, 1ull) - cc.base);
            const uint64_t pm = __ballot(parked);
            if (pm) {
                const uint32_t np = (uint32_t)__popcll(pm);
                if (qn == 0) qbase = q0;
                if (qn + np > QCAP || q0 + 64 - qbase > 0xFFFFFFFFull) { drain(true); qbase = q0; }
                if (np > QCAP) {  // more special lanes than the queue holds (a deep trie: every prefix is a child Node): walk them here
                    if (parked) {
                        const BftHit h = bft_walk<W, BftRootLds<STAGED>, PROBE, 2>(im, acc, root, t);
                        answer_late(i, h, true);
                    }
                } else {
                    if (parked) {
                        const uint32_t qp = qn + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
#pragma unroll
                        for (int w = 0; w < W; w++) q_t[(size_t)qp * W + w] = t[w];
                        q_i[qp] = (uint32_t)(i - qbase);
                    }
                    qn += np;
                }
            }
        }
        // the chunk's line of presence words: lanes 0..15, one store instruction
        __builtin_amdgcn_wave_barrier();
        {
            const uint64_t wi = chunk * BFT_WALK_PASSES + lane;
            if (lane < BFT_WALK_PASSES && wi < nwords) bits64[wi] = my_bits[lane];
        }
        __builtin_amdgcn_wave_barrier();
        chunk = ctr ? n_waves + (uint64_t)__builtin_amdgcn_readfirstlane(next_claim) : chunk + n_waves;
    }
    if (qn) drain(false);
}

// Two builds of the same body.  k_query: registers as the compiler likes them (106 SGPRs: the BftImage pointers live in
// SGPRs), which caps a SIMD at 7 waves, i.e. ONE 1024-thread workgroup per CU -- the fastest arrangement for a one-level
// index (4 waves per SIMD keep the beyond-L2 gather path full, more only thrash it).  k_query8: held to 8 waves per SIMD
// (78 SGPRs) so that two workgroups share a CU -- +10..40 % on deep tries and on L2-resident ones.
// PROBE: suffix-group probe mode fixed at compile time (0 = 4-row blocks, 1 = 8-row blocks; the 1024-thread kernels) or read
// from the image (-1; the other workgroup sizes): the 4-row code alone fits the 64 VGPRs of k_query8 without spilling.
template <int W, int BLOCK, bool STAGED, int PROBE>
__global__ __launch_bounds__(BLOCK) void k_query(BftImage im, const uint8_t* __restrict__ packed, uint64_t n, int B, uint64_t* __restrict__ bits64,
                                                 uint32_t* __restrict__ rows, BftClaimCtr ctr) {
    query_body<W, BLOCK, STAGED, PROBE>(im, packed, n, B, bits64, rows, ctr);
}
template <int W, int BLOCK, bool STAGED, int PROBE>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_query8(BftImage im, const uint8_t* __restrict__ packed, uint64_t n, int B,
                                                                                             uint64_t* __restrict__ bits64, uint32_t* __restrict__ rows,
                                                                                             BftClaimCtr ctr) {
    query_body<W, BLOCK, STAGED, PROBE>(im, packed, n, B, bits64, rows, ctr);
}
// k_query6: the arrangement in between -- two 768-thread workgroups per CU, 6 wavefronts per SIMD with 84 VGPRs each.  The walk
// of the two-word rows (k = 36..63) needs 73-81 VGPRs: k_query8 spills 20-40 of them, k_query runs 4 wavefronts per SIMD.
#define BFT_BLOCK6 768
template <int W, bool STAGED, int PROBE>
__global__ __launch_bounds__(BFT_BLOCK6) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_query6(BftImage im, const uint8_t* __restrict__ packed, uint64_t n, int B,
                                                                                                 uint64_t* __restrict__ bits64, uint32_t* __restrict__ rows,
                                                                                                 BftClaimCtr ctr) {
    query_body<W, BFT_BLOCK6, STAGED, PROBE>(im, packed, n, B, bits64, rows, ctr);
}

// Batched isBranchingRight / isBranchingLeft (src/branchingNode.c:16-112, :240-340; loop of src/file_io.c:943-998):
// successors of a k-mer = present k-mers kmer[1..k-1]+N, predecessors = present k-mers N+kmer[0..k-2], N in ACGT.
// In T-form the four successors differ only in bits 2..3 of the last digit (n9 of the last prefix) and the four
// predecessors only in bits 0..1 of the first digit (n1 of the first prefix): one conversion per side, then four walks
// that share every container down to the last cluster / suffix group (src/presenceNode.c:15-1211 exploits the same).
// counts[i] = (successors << 4) | predecessors when requested; the bit = successors > 1 || predecessors > 1.
template <int W, int BLOCK, bool STAGED, int PROBE>
__device__ __forceinline__ void branching_body(const BftImage& im, const uint8_t* __restrict__ packed, uint64_t n, int B,
                                               uint64_t* __restrict__ bits64, uint8_t* __restrict__ counts) {
    extern __shared__ __align__(16) uint8_t lds[];
    uint32_t* l_hm = (uint32_t*)lds;
    uint8_t* l_bf = lds + BFT_LDS_HM_BYTES;
    const BftNode root = im.nodes[0];
    const uint32_t bf_bytes = STAGED ? ((BFT_MODULO_HASH * (uint32_t)root.bf_wb + 15u) & ~15u) : 0u;
    BftCCX* l_cc = (BftCCX*)(l_bf + bf_bytes);
    {
        const uint4* g = (const uint4*)im.hashmod;
        uint4* l = (uint4*)l_hm;
        for (uint32_t i = threadIdx.x; i < BFT_LDS_HM_BYTES / 16; i += BLOCK) l[i] = g[i];
        if (STAGED) {
            const uint64_t* gb = (const uint64_t*)(im.bfT + (size_t)root.bf_off * 8);
            uint64_t* lb = (uint64_t*)l_bf;
            const uint32_t nb8 = (BFT_MODULO_HASH * (uint32_t)root.bf_wb) / 8;
            for (uint32_t i = threadIdx.x; i < nb8; i += BLOCK) lb[i] = gb[i];
            for (uint32_t i = threadIdx.x; i < root.ncc; i += BLOCK) l_cc[i] = im.ccx[root.cc_first + i];
        }
    }
    __syncthreads();
    const BftRootLds<STAGED> acc{im, l_hm, l_bf, l_cc};
    const uint64_t end_aligned = ((uint64_t)packed + n * (uint64_t)B) & ~3ull;
    const uint64_t nblk = (n + BLOCK - 1) / BLOCK;
    const int k = im.k, L = im.L, rb = 2 * (k - 9 * L);
    for (uint64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const uint64_t i = blk * BLOCK + threadIdx.x;
        int branching = 0;
        if (i < n) {
            uint64_t x[W], y[W], t[W];
            load_x<W>(packed, i, B, end_aligned, x);
            // successors: drop the first nucleotide, the last one is the wildcard
#pragma unroll
            for (int w = 0; w < W; w++) y[w] = (x[w] >> 2) | (w + 1 < W ? x[w + 1] << 62 : 0ull);
            bft_tform_from_x<W>(y, k, t);
            int cl = 0, cr = 0;
            // the four successors differ in the last nucleotide only (bits vo.. of the T-form's last word)
            const int vo = rb ? 0 : 2;
            uint64_t cand[4][W];
#pragma unroll
            for (int v = 0; v < 4; v++) {
#pragma unroll
                for (int w = 0; w < W; w++) cand[v][w] = t[w] | (w == W - 1 ? (uint64_t)v << vo : 0ull);
            }
            // one shared descent, four finishes (any index, any level)
            cr = bft_walk_last4<W, BftRootLds<STAGED>>(im, acc, root, t, counts != nullptr);
            if (counts || cr < 2) {
                // predecessors: shift in a wildcard first nucleotide, drop the last one
#pragma unroll
                for (int w = W - 1; w >= 0; w--) y[w] = (x[w] << 2) | (w > 0 ? x[w - 1] >> 62 : 0ull);
                const int top = 2 * k - 64 * (W - 1);  // bits used in the last word
                if (top < 64) y[W - 1] &= (1ull << top) - 1ull;
                bft_tform_from_x<W>(y, k, t);
                const int o = rb + 18 * (L - 1), ow = W - 1 - (o >> 6), osh = o & 63;  // digit 0 starts at bit o of the T-form integer
#pragma unroll
                for (int v = 0; v < 4; v++) {
#pragma unroll
                    for (int w = 0; w < W; w++) cand[v][w] = t[w] | (w == ow ? (uint64_t)v << osh : 0ull);
                }
                for (int v = 0; v < 4 && (counts || cl < 2); v++) cl += bft_walk<W, BftRootLds<STAGED>, PROBE>(im, acc, root, cand[v]).present;
            }
            branching = cr > 1 || cl > 1;
            if (counts) counts[i] = (uint8_t)((cr << 4) | cl);
        }
        const uint64_t mask = __ballot(branching);
        const uint64_t q0 = i & ~63ull;
        if ((threadIdx.x & 63u) == 0 && q0 < n) bits64[q0 >> 6] = mask;
    }
}

// the three register budgets of k_query / k_query8 / k_query6 (see there)
template <int W, int BLOCK, bool STAGED, int PROBE>
__global__ __launch_bounds__(BLOCK) void k_branching(BftImage im, const uint8_t* __restrict__ packed, uint64_t n, int B, uint64_t* __restrict__ bits64,
                                                     uint8_t* __restrict__ counts) {
    branching_body<W, BLOCK, STAGED, PROBE>(im, packed, n, B, bits64, counts);
}
template <int W, int BLOCK, bool STAGED, int PROBE>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_branching8(BftImage im, const uint8_t* __restrict__ packed, uint64_t n,
                                                                                                 int B, uint64_t* __restrict__ bits64,
                                                                                                 uint8_t* __restrict__ counts) {
    branching_body<W, BLOCK, STAGED, PROBE>(im, packed, n, B, bits64, counts);
}
template <int W, bool STAGED, int PROBE>
__global__ __launch_bounds__(BFT_BLOCK6) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_branching6(BftImage im, const uint8_t* __restrict__ packed, uint64_t n,
                                                                                                     int B, uint64_t* __restrict__ bits64,
                                                                                                     uint8_t* __restrict__ counts) {
    branching_body<W, BFT_BLOCK6, STAGED, PROBE>(im, packed, n, B, bits64, counts);
}

