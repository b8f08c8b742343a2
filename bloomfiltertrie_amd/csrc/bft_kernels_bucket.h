// bft_kernels_bucket.h -- prefix-bucketed query batches ("sort by the rotated prefix so that a
// wavefront works on one CC / cluster"; the key is the rotated root prefix of src/presenceNode.c:1327-1371).
// Device code of libbft_gpu.so, included by bft_gpu.hip only.
//
// Why: a k-mer's walk ends in the slice of the sorted table `tk` (and of the flat root tables) that its root prefix r owns.
// In batch order those slices are hit at random, every query costs 1.7 L2 misses and the kernel sits at the chip's
// beyond-L2 gather rate (~57 G misses/s) whatever the index size.  Partitioned by the top bits of r, all queries of a
// bucket share one contiguous slice: each XCD walks ITS buckets one after the other, the slice stays in that XCD's 4 MiB L2
// and the table is streamed from HBM once per launch instead of being gathered line by line.
//
//   k_bucket_hist     tile of BK_TILE queries -> bucket histogram (LDS atomics)            -> hist[bucket][tile]
//   k_bucket_rowsum   per-bucket totals (one workgroup per bucket)
//   k_bucket_plan     64-aligned bucket bases, bucket q -> XCD q % 8, per-XCD chunk lists (one workgroup, parallel scans)
//   k_bucket_rowscan  off[bucket][tile] = start of the bucket + exclusive prefix of its row
//   k_bucket_scatter  tile -> T-form records counting-sorted in LDS, written in runs per bucket; lrank[i] = rank of query i in its tile's sorted order
//   k_query_bk        the walk of k_query on the T-form records, chunks claimed per XCD list (one atomic per chunk); permuted answer bits
//   k_unpermute       per tile: the tile's runs back into LDS in sorted order, then answer bit (and row) of query i = sorted[lrank[i]]
//
// Same answers as k_query by construction (same bft_walk on the same T-form); tests run both and compare.
#pragma once

#define BK_BLOCK 1024
#define BK_MAX_BUCKETS 1024
#define BK_NXCD 8

// queries per tile of the partition passes: the sorted records of a tile are staged in 64 KiB of LDS
template <int W>
struct BkTile {
    static constexpr int value = W == 1 ? 8192 : (W == 2 ? 4096 : 2048);
};

struct BkPlan {                              // device-resident, written by k_bucket_plan
    uint32_t base[BK_MAX_BUCKETS + 1];       // first permuted position of each bucket (a multiple of 64); [nb] = padded total
    uint32_t size[BK_MAX_BUCKETS];           // queries per bucket
    uint32_t start[BK_MAX_BUCKETS];          // unpadded start of each bucket (prefix of the sizes)
    uint32_t delta[BK_MAX_BUCKETS];          // base[b] - start[b]: added (mod 2^32) to the scanned offsets
    uint32_t xl_bucket[BK_NXCD][BK_MAX_BUCKETS];      // the buckets each XCD walks, in order
    uint32_t xl_chunk0[BK_NXCD][BK_MAX_BUCKETS + 1];  // first chunk (of BK_BLOCK queries) of each of them in that XCD's list
    uint32_t xl_n[BK_NXCD];                  // buckets per XCD
    uint32_t cursor[BK_NXCD];                // next unclaimed chunk of each list (zeroed by k_bucket_plan, bumped by k_query_bk)
};

// bucket of a packed k-mer: the top `bits` bits of its rotated root prefix r = n2..n9,n1
template <int W>
__device__ __forceinline__ uint32_t bk_bucket_of_x(const uint64_t* x, int bits) {
    return bft_rot_prefix((uint32_t)x[0] & 0x3FFFFu) >> (18 - bits);
}

template <int W>
__global__ __launch_bounds__(BK_BLOCK) void k_bucket_hist(const uint8_t* __restrict__ packed, uint64_t n, int B, int bits, uint32_t ntiles,
                                                          uint32_t* __restrict__ hist) {
    constexpr int TILE = BkTile<W>::value;
    __shared__ uint32_t cnt[BK_MAX_BUCKETS];
    const uint32_t nb = 1u << bits;
    const uint64_t end_aligned = ((uint64_t)packed + n * (uint64_t)B) & ~3ull;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (uint32_t b = threadIdx.x; b < nb; b += BK_BLOCK) cnt[b] = 0;
        __syncthreads();
        const uint64_t q0 = (uint64_t)tile * TILE;
#pragma unroll
        for (int j = 0; j < TILE / BK_BLOCK; j++) {
            const uint64_t i = q0 + (uint64_t)j * BK_BLOCK + threadIdx.x;
            if (i < n) {
                uint64_t x[W];
                load_x<W>(packed, i, B, end_aligned, x);
                atomicAdd(&cnt[bk_bucket_of_x<W>(x, bits)], 1u);
            }
        }
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < nb; b += BK_BLOCK) hist[(uint64_t)b * ntiles + tile] = cnt[b];
        __syncthreads();
    }
}

// exclusive scan of one value per thread over a workgroup of BK_BLOCK threads (wave shuffles + one LDS hop)
__device__ __forceinline__ uint32_t bk_block_excl_scan(uint32_t mine, uint32_t* wsum /* [BK_BLOCK/64] in LDS */) {
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63u) >= d) incl += up;
    }
    __syncthreads();  // wsum may still be read from a previous use
    if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += wsum[w];
    return wbase + incl - mine;
}

// hist[bucket][tile] -> per-bucket totals: one workgroup per bucket, coalesced reads of its row
__global__ __launch_bounds__(BK_BLOCK) void k_bucket_rowsum(const uint32_t* __restrict__ hist, uint32_t ntiles, uint32_t* __restrict__ totals) {
    __shared__ uint32_t wsum[BK_BLOCK / 64];
    const uint32_t* row = hist + (uint64_t)blockIdx.x * ntiles;
    uint32_t acc = 0;
    for (uint32_t t = threadIdx.x; t < ntiles; t += BK_BLOCK) acc += row[t];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < BK_BLOCK / 64; w++) tot += wsum[w];
        totals[blockIdx.x] = tot;
    }
}

// One workgroup of BK_BLOCK threads: bucket bases and the per-XCD lists, all in parallel.  Buckets start on multiples of 64
// permuted positions (a wavefront's 64 answer bits are one aligned word).  Bucket q goes to XCD q % 8: neighbouring buckets (and
// so any hot region of the prefix space) spread over all XCDs, each XCD's list walks the table in ascending order, and whatever
// imbalance is left is evened out at run time by the workgroups that move on to other lists when theirs is done.
__global__ __launch_bounds__(BK_BLOCK) void k_bucket_plan(const uint32_t* __restrict__ totals, int bits, BkPlan* __restrict__ plan) {
    __shared__ uint32_t wsum[BK_BLOCK / 64], ebase[BK_BLOCK + 1];
    const uint32_t nb = 1u << bits, b = threadIdx.x;
    const uint32_t size = b < nb ? totals[b] : 0u;
    const uint32_t start = bk_block_excl_scan(size, wsum);                   // unpadded start of the bucket
    const uint32_t base = bk_block_excl_scan((size + 63u) & ~63u, wsum);     // padded
    if (b < nb) {
        plan->base[b] = base;
        plan->size[b] = size;
        plan->start[b] = start;
        plan->delta[b] = base - start;
        if (b == nb - 1) plan->base[nb] = base + ((size + 63u) & ~63u);
    }
    // thread t = (x, i): the i-th bucket of XCD x is bucket i * 8 + x; chunk prefix over t, rebased per list
    const uint32_t per = nb / BK_NXCD, x = b / per, i = b % per, q = i * BK_NXCD + x;
    const uint32_t chunks = b < nb ? (totals[q] + BK_BLOCK - 1) / BK_BLOCK : 0u;
    const uint32_t e = bk_block_excl_scan(chunks, wsum);
    if (b <= nb) ebase[b] = e;  // (thread nb, when it exists, holds the grand total; see below for nb == BK_BLOCK)
    if (b == nb - 1) ebase[nb] = e + chunks;
    __syncthreads();
    if (b < nb) {
        const uint32_t e0 = ebase[x * per];
        plan->xl_bucket[x][i] = q;
        plan->xl_chunk0[x][i] = e - e0;
        if (i == per - 1) plan->xl_chunk0[x][per] = e + chunks - e0;
    }
    if (b < BK_NXCD) { plan->xl_n[b] = per; plan->cursor[b] = 0; }
}

// totals -> off[bucket][tile] = unpadded start of the bucket + exclusive prefix of its row: one workgroup per bucket
__global__ __launch_bounds__(BK_BLOCK) void k_bucket_rowscan(const uint32_t* __restrict__ hist, uint32_t ntiles, const BkPlan* __restrict__ plan,
                                                             uint32_t* __restrict__ off) {
    __shared__ uint32_t wsum[BK_BLOCK / 64];
    const uint32_t* row = hist + (uint64_t)blockIdx.x * ntiles;
    uint32_t* orow = off + (uint64_t)blockIdx.x * ntiles;
    uint32_t carry = plan->start[blockIdx.x];
    for (uint32_t t0 = 0; t0 < ntiles; t0 += BK_BLOCK) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < ntiles ? row[t] : 0u;
        const uint32_t ex = bk_block_excl_scan(v, wsum);
        if (t < ntiles) orow[t] = carry + ex;
        uint32_t tot = 0;
        for (int w = 0; w < BK_BLOCK / 64; w++) tot += wsum[w];  // wsum holds the per-wavefront sums of this round
        carry += tot;
        __syncthreads();
    }
}

// T-form records of a tile, counting-sorted by bucket in LDS and written out in one run per bucket (coalesced stores of
// ~TILE/nb records each); lrank[i] = the position of query i in its tile's sorted order (u16, stored in batch order).
// The loop is latency-bound (a handful of barriers per tile), so the next tile's k-mers and the tile's run offsets are
// requested before the LDS work of the current tile starts.
template <int W>
__global__ __launch_bounds__(BK_BLOCK) void k_bucket_scatter(const uint8_t* __restrict__ packed, uint64_t n, int B, int k, int bits, uint32_t ntiles,
                                                             const uint32_t* __restrict__ off, const BkPlan* __restrict__ plan,
                                                             uint64_t* __restrict__ trec, uint16_t* __restrict__ lrank) {
    constexpr int TILE = BkTile<W>::value, PER = TILE / BK_BLOCK;
    __shared__ uint64_t stage[TILE * W];        // 64 KiB
    __shared__ uint32_t cnt[BK_MAX_BUCKETS];     // per-bucket count, then the local start (exclusive scan)
    __shared__ uint32_t gdst[BK_MAX_BUCKETS];    // permuted position of the tile's first record of each bucket, minus its local start
    __shared__ uint32_t wsum[BK_BLOCK / 64];
    const uint32_t nb = 1u << bits;
    const uint64_t end_aligned = ((uint64_t)packed + n * (uint64_t)B) & ~3ull;
    uint64_t x[PER][W];
    auto load_tile = [&](uint32_t tile) {
        const uint64_t q0 = (uint64_t)tile * TILE;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint64_t i = q0 + (uint64_t)j * BK_BLOCK + threadIdx.x;
            if (i < n) load_x<W>(packed, i, B, end_aligned, x[j]);
        }
    };
    if (blockIdx.x < ntiles) load_tile(blockIdx.x);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t q0 = (uint64_t)tile * TILE;
        // run offsets of this tile: requested now, used after the count phase
        const uint32_t goff = threadIdx.x < nb ? off[(uint64_t)threadIdx.x * ntiles + tile] + plan->delta[threadIdx.x] : 0u;
        for (uint32_t b = threadIdx.x; b < nb; b += BK_BLOCK) cnt[b] = 0;
        __syncthreads();
        uint64_t t[PER][W];
        uint32_t bk[PER], rk[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint64_t i = q0 + (uint64_t)j * BK_BLOCK + threadIdx.x;
            bk[j] = 0xFFFFFFFFu;
            rk[j] = 0;
            if (i < n) {
                bft_tform_from_x<W>(x[j], k, t[j]);
                bk[j] = bk_bucket_of_x<W>(x[j], bits);
                rk[j] = atomicAdd(&cnt[bk[j]], 1u);  // rank inside (tile, bucket): any order will do, lrank records it
            }
        }
        if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x);  // next tile's k-mers: in flight during the rest of this one
        __syncthreads();
        const uint32_t lstart = bk_block_excl_scan(threadIdx.x < nb ? cnt[threadIdx.x] : 0u, wsum);
        if (threadIdx.x < nb) {
            cnt[threadIdx.x] = lstart;
            gdst[threadIdx.x] = goff - lstart;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; j++) {
            if (bk[j] == 0xFFFFFFFFu) continue;
            const uint32_t s = cnt[bk[j]] + rk[j];
#pragma unroll
            for (int w = 0; w < W; w++) stage[(size_t)s * W + w] = t[j][w];
            lrank[q0 + (uint64_t)j * BK_BLOCK + threadIdx.x] = (uint16_t)s;
        }
        __syncthreads();
        const uint32_t nt = (uint32_t)min((uint64_t)TILE, n - q0);
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint32_t s = (uint32_t)j * BK_BLOCK + threadIdx.x;
            if (s >= nt) continue;
            uint64_t r[W];
#pragma unroll
            for (int w = 0; w < W; w++) r[w] = stage[(size_t)s * W + w];
            const uint32_t b = bft_digit<W>(r, k, 0) >> (18 - bits);
            const uint64_t d = (uint64_t)(gdst[b] + s);
#pragma unroll
            for (int w = 0; w < W; w++) trec[d * W + w] = r[w];
        }
        __syncthreads();
    }
}

// The XCD this workgroup runs on.  HW_REG_XCC_ID (hwreg 20), bits 3:0; placement only changes speed, never answers.
__device__ __forceinline__ uint32_t bk_xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

// Presence (and row) of the bucketed T-form records.  Work = the chunks (BK_BLOCK records) of the buckets dealt to each XCD
// by k_bucket_plan; each list has a cursor.  Every WAVEFRONT claims chunks on its own -- no workgroup barrier in the loop, a
// slow lane only holds up its own 64 -- from the list of the XCD its workgroup sits on: one atomicAdd per chunk, issued a
// chunk ahead so that its latency hides behind the 16 x 64 walks of the current chunk; when the list is exhausted the
// wavefront moves on to the next XCD's list (the tail only).  All wavefronts of an XCD therefore sweep the same bucket at
// the same time and the bucket's slice of the index stays in that XCD's L2 (measured: 0.13 L2 misses per query on the 1.8 GB
// image of config 4).  Every chunk is claimed exactly once whatever the placement: placement changes speed, never answers.
// Answers are written in permuted order: bit p%64 of word p/64, row at prow[p].
template <int W, bool STAGED, int PROBE>
__device__ __forceinline__ void query_bk_body(const BftImage& im, const uint64_t* __restrict__ trec, BkPlan* __restrict__ plan,
                                              uint64_t* __restrict__ pbits, uint32_t* __restrict__ prow) {
    extern __shared__ __align__(16) uint8_t lds[];
    uint32_t* l_hm = (uint32_t*)lds;
    uint8_t* l_bf = lds + BFT_LDS_HM_BYTES;
    const BftNode root = im.nodes[0];
    const uint32_t bf_bytes = STAGED ? ((BFT_MODULO_HASH * (uint32_t)root.bf_wb + 15u) & ~15u) : 0u;
    BftCCX* l_cc = (BftCCX*)(l_bf + bf_bytes);
    const uint32_t my_xcd = bk_xcc_id();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t claim = 0;
    if (lane == 0) claim = atomicAdd(&plan->cursor[my_xcd], 1u);  // in flight during the staging below
    {
        const uint4* g = (const uint4*)im.hashmod;
        uint4* l = (uint4*)l_hm;
        for (uint32_t i = threadIdx.x; i < BFT_LDS_HM_BYTES / 16; i += blockDim.x) l[i] = g[i];
        if (STAGED) {
            const uint64_t* gb = (const uint64_t*)(im.bfT + (size_t)root.bf_off * 8);
            uint64_t* lb = (uint64_t*)l_bf;
            const uint32_t nb8 = (BFT_MODULO_HASH * (uint32_t)root.bf_wb) / 8;
            for (uint32_t i = threadIdx.x; i < nb8; i += blockDim.x) lb[i] = gb[i];
            for (uint32_t i = threadIdx.x; i < root.ncc; i += blockDim.x) l_cc[i] = im.ccx[root.cc_first + i];
        }
    }
    __syncthreads();
    const BftRootLds<STAGED> acc{im, l_hm, l_bf, l_cc};
    for (uint32_t step = 0; step < BK_NXCD; step++) {
        const uint32_t y = (my_xcd + step) & 7u;
        const uint32_t nseg = plan->xl_n[y];
        const uint32_t nchunks = plan->xl_chunk0[y][nseg];
        if (step > 0 && lane == 0) claim = atomicAdd(&plan->cursor[y], 1u);  // first claim on a list this wavefront moves on to
        uint32_t seg = 0;
        for (;;) {
            const uint32_t c = (uint32_t)__shfl((int)claim, 0, 64);
            if (c >= nchunks) break;
            if (lane == 0) claim = atomicAdd(&plan->cursor[y], 1u);  // the next claim: used after this chunk
            while (plan->xl_chunk0[y][seg + 1] <= c) seg++;  // the claims a wavefront gets from a list only grow: forward scan
            const uint32_t b = plan->xl_bucket[y][seg];
            const uint32_t size = plan->size[b];
            const uint32_t in0 = (c - plan->xl_chunk0[y][seg]) * BK_BLOCK;  // first position of the chunk inside the bucket
            const uint64_t p0 = (uint64_t)plan->base[b] + in0;
            for (uint32_t it = 0; it < BK_BLOCK / 64 && in0 + it * 64 < size; it++) {
                const uint32_t inb = in0 + it * 64 + lane;
                const uint64_t p = p0 + it * 64 + lane;
                int present = 0;
                if (inb < size) {
                    uint64_t t[W];
                    bft_load_row<W>(trec + p * W, t);
                    const BftHit h = bft_walk<W, BftRootLds<STAGED>, PROBE>(im, acc, root, t);
                    present = h.present;
                    if (prow) prow[p] = present ? bft_hit_out(im, h) : BFT_ABSENT_ROW;
                }
                const uint64_t mask = __ballot(present);
                if (lane == 0) pbits[p >> 6] = mask;
            }
        }
    }
}

template <int W, bool STAGED, int PROBE>
__global__ __launch_bounds__(BK_BLOCK) void k_query_bk(BftImage im, const uint64_t* __restrict__ trec, BkPlan* __restrict__ plan, uint64_t* __restrict__ pbits,
                                                       uint32_t* __restrict__ prow) {
    query_bk_body<W, STAGED, PROBE>(im, trec, plan, pbits, prow);
}
// k_query_bk8: two 768-thread workgroups per CU = 6 wavefronts per SIMD, 84 VGPRs each: with 1024 threads and the 64 VGPRs of 8
// per SIMD the walk spills two registers inside the chunk loop once the node prefix hash is part of it.  The wavefronts claim
// their chunks on their own, so the workgroup size is free.
#define BK_WALK_BLOCK 768
template <int W, bool STAGED, int PROBE>
__global__ __launch_bounds__(BK_WALK_BLOCK) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_query_bk8(BftImage im, const uint64_t* __restrict__ trec,
                                                                                                   BkPlan* __restrict__ plan, uint64_t* __restrict__ pbits,
                                                                                                   uint32_t* __restrict__ prow) {
    query_bk_body<W, STAGED, PROBE>(im, trec, plan, pbits, prow);
}

// Answers back in batch order, tile by tile (the tiles of k_bucket_scatter).  The tile's records sit in one run per bucket of the
// permuted arrays; phase 1 rebuilds the tile's sorted order in LDS (thread s finds its bucket by a binary search over the tile's
// local starts: consecutive threads read consecutive permuted positions, a few words per wavefront), phase 2 lets every query
// pick its answer by lrank -- LDS lookups instead of 10^8 random gathers from the permuted bitmap.  All global loads of a phase
// are issued before the first is used (the loop is latency-bound otherwise).
template <int TILE, bool ROWS>
__global__ __launch_bounds__(BK_BLOCK) void k_unpermute(const uint16_t* __restrict__ lrank, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ off,
                                                        const BkPlan* __restrict__ plan, int bits, uint32_t ntiles, const uint64_t* __restrict__ pbits,
                                                        const uint32_t* __restrict__ prow, uint64_t n, uint64_t* __restrict__ bits64, uint32_t* __restrict__ rows) {
    constexpr int PER = TILE / BK_BLOCK;
    __shared__ uint32_t lstart[BK_MAX_BUCKETS + 1], g0[BK_MAX_BUCKETS], wsum[BK_BLOCK / 64];
    __shared__ uint64_t lbits[TILE / 64];
    __shared__ uint32_t lrows[ROWS ? TILE : 1];
    const uint32_t nb = 1u << bits;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t q0 = (uint64_t)tile * TILE;
        const uint32_t nt = (uint32_t)min((uint64_t)TILE, n - q0);
        const uint32_t b = threadIdx.x;
        const uint32_t c = b < nb ? hist[(uint64_t)b * ntiles + tile] : 0u;
        const uint32_t go = b < nb ? off[(uint64_t)b * ntiles + tile] + plan->delta[b] : 0u;
        uint16_t lr[PER];  // this thread's queries: requested now, used in phase 2
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint64_t i = q0 + (uint64_t)j * BK_BLOCK + threadIdx.x;
            lr[j] = i < n ? lrank[i] : (uint16_t)0;
        }
        const uint32_t ls = bk_block_excl_scan(c, wsum);
        if (b < nb) {
            lstart[b] = ls;
            g0[b] = go - ls;  // permuted position of sorted position s (in b): g0[b] + s
        }
        if (b == 0) lstart[nb] = nt;
        __syncthreads();
        uint32_t p[PER];
        uint64_t wv[PER];
        uint32_t rv[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint32_t s = (uint32_t)j * BK_BLOCK + threadIdx.x;
            p[j] = 0;
            if (s < nt) {
                uint32_t lo = 0, hi = nb;  // last bucket whose local start is <= s (empty buckets share their start with the next one)
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (lstart[mid] <= s) lo = mid; else hi = mid;
                }
                p[j] = g0[lo] + s;
            }
        }
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint32_t s = (uint32_t)j * BK_BLOCK + threadIdx.x;
            wv[j] = s < nt ? pbits[p[j] >> 6] : 0ull;
            rv[j] = (ROWS && s < nt) ? prow[p[j]] : 0u;
        }
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint32_t s = (uint32_t)j * BK_BLOCK + threadIdx.x;
            const uint64_t mask = __ballot((int)((wv[j] >> (p[j] & 63u)) & 1ull));
            if ((threadIdx.x & 63u) == 0) lbits[s >> 6] = mask;
            if (ROWS) lrows[s] = rv[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint64_t i = q0 + (uint64_t)j * BK_BLOCK + threadIdx.x;
            const uint32_t s = lr[j];
            const int present = i < n ? (int)((lbits[s >> 6] >> (s & 63u)) & 1ull) : 0;
            if (ROWS && i < n) rows[i] = lrows[s];
            const uint64_t mask = __ballot(present);
            if ((threadIdx.x & 63u) == 0 && (i & ~63ull) < n) bits64[i >> 6] = mask;
        }
        __syncthreads();
    }
}
