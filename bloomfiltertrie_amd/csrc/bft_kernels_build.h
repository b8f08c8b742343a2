// bft_kernels_build.h -- de-duplication of the sorted (k-mer, genome) pairs for the bulk build: k_iota, k_gather, k_flags, k_scatter
// Device code of libbft_gpu.so, included by bft_gpu.hip only (one translation unit: the kernels are templates launched from
// the host code there).
#pragma once
// a log of composites back into T-form k-mers and their ids (in place + the id array)
__global__ void k_log_decompose(uint64_t* __restrict__ log_k, uint64_t n, uint32_t cgb, uint32_t* __restrict__ log_g) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t c = log_k[i];
        log_g[i] = (uint32_t)(c & ((1ull << cgb) - 1ull));
        log_k[i] = c >> cgb;
    }
}
__global__ void k_iota(uint32_t* p, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
template <class T>
__global__ void k_gather(const T* __restrict__ in, const uint32_t* __restrict__ perm, T* __restrict__ out, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[perm[i]];
}

// the sorted pairs: every key word and the genome id of entry perm[i] (SoA in, SoA out)
// nlb != 0: the id of position p is that of the insert call it came with -- the first of the nlb calls that ends behind p (lb_end ascending)
__global__ void k_gather_pairs(const uint64_t* __restrict__ keys, uint64_t stride, int W, const uint32_t* __restrict__ g, const uint32_t* __restrict__ perm,
                               uint64_t* __restrict__ okeys, uint64_t ostride, uint32_t* __restrict__ og, uint64_t n, const uint64_t* __restrict__ lb_end,
                               const uint32_t* __restrict__ lb_gid, uint32_t nlb) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t p = perm[i];
        for (int w = 0; w < W; w++) okeys[(uint64_t)w * ostride + i] = keys[(uint64_t)w * stride + p];
        if (nlb) {
            uint32_t lo = 0, hi = nlb - 1;  // first call with lb_end > p
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (lb_end[mid] > p) hi = mid; else lo = mid + 1;
            }
            og[i] = lb_gid[lo];
        } else
            og[i] = g[p];
    }
}

// sorted (T, g) pairs -> head-of-k-mer flag and keep-pair flag
__global__ void k_flags(const uint64_t* __restrict__ keys, uint64_t stride, int W, const uint32_t* __restrict__ g, uint64_t n,
                        uint32_t* __restrict__ head, uint32_t* __restrict__ keep) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        bool same = i > 0;
        if (same)
            for (int w = 0; w < W; w++) same = same && (keys[(uint64_t)w * stride + i] == keys[(uint64_t)w * stride + i - 1]);
        head[i] = same ? 0u : 1u;
        keep[i] = (!same || g[i] != g[i - 1]) ? 1u : 0u;
    }
}

__global__ void k_scatter(const uint64_t* __restrict__ keys, uint64_t stride, int W, const uint32_t* __restrict__ g, uint64_t n,
                          const uint32_t* __restrict__ head, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ posK,
                          const uint32_t* __restrict__ posP, uint32_t* __restrict__ pg, uint64_t* __restrict__ tk, uint32_t* __restrict__ seg_off) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (keep[i]) pg[posP[i]] = g[i];
        if (head[i]) {
            const uint32_t q = posK[i];
            for (int w = 0; w < W; w++) tk[(uint64_t)q * W + w] = keys[(uint64_t)w * stride + i];
            seg_off[q] = posP[i];
        }
    }
}

// ---- composite path: one-word keys whose genome ids arrive in ascending order and fit the key's spare low bits ----------------
// c = (T << gb) | genome: ONE 8-byte array is sorted (on the T bits only: the sort is stable, so the ids keep their ascending
// order inside a k-mer) instead of 8-byte keys + 4-byte values -- a third less traffic in each of the seven radix passes -- and the
// de-duplication flags come out of neighbouring composites on the fly instead of through two flag arrays and two scans.
struct BftCompose {  // input "iterator" of the sort: composite i from the insertion log
    const uint64_t* k;
    const uint32_t* g;
    uint32_t gb;
    __host__ __device__ uint64_t operator()(uint32_t i) const { return g ? (k[i] << gb) | (uint64_t)g[i] : k[i]; }  // (g == NULL: the log already holds composites)
    __device__ uint64_t key(uint32_t i) const { return (*this)(i); }  // (as the input of bft_rs::sort)
    __device__ bft_rs::NoVal val(uint32_t) const { return bft_rs::NoVal{}; }
};
// two-word keys: what the root-prefix split moves -- the top 64 bits of the T-form (left-aligned: their top 18 bits are the root prefix) as the
// key, the bits below them and the genome id as the value (bft_front.hip: BftItem2)
struct __attribute__((packed, aligned(4))) BftSplit2Val {
    uint64_t lo;
    uint32_t id;
};
struct BftSplit2In {
    const uint64_t* k0;  // word 0 (most significant), word 1 of the log
    const uint64_t* k1;
    const uint32_t* g;
    uint32_t sh;  // 2k - 64: the bits of word 1 that belong to the T-form's low part once the top 64 are taken (2 .. 64)
    __device__ uint64_t key(uint32_t i) const { return sh == 64 ? k0[i] : (k0[i] << (64 - sh)) | (k1[i] >> sh); }
    __device__ BftSplit2Val val(uint32_t i) const {
        BftSplit2Val v;
        v.lo = sh == 64 ? k1[i] : (k1[i] & ((1ull << sh) - 1ull));
        v.id = g[i];
        return v;
    }
};
template <class GT>
struct BftPairIn {  // input of the sorts that move (k-mer, id) pairs: the log's k-mers, its ids narrowed to GT
    const uint64_t* k;
    const uint32_t* g;
    __device__ uint64_t key(uint32_t i) const { return k[i]; }
    __device__ GT val(uint32_t i) const { return (GT)g[i]; }
};
struct BftPairFlags {  // input of the scan: (first pair of its k-mer) << 32 | (first pair of its (k-mer, genome))
    const uint64_t* c;
    uint32_t gb;
    __host__ __device__ uint64_t operator()(uint32_t i) const {
        const uint64_t a = c[i], b = i ? c[i - 1] : ~a;
        const uint64_t head = (a >> gb) != (b >> gb), keep = a != b;
        return (head << 32) | keep;
    }
};
// root-prefix buckets of the composites sorted on their top bits: off[r] = first composite whose prefix is >= r (r = 0..nb), and the
// size of the largest bucket.  dbase (optional): where the top `dbits` bits of the prefix change -- the last pass of the sort that made the
// order left that table behind (bft_rs::sort, last_dbase) --: a bucket is then searched inside its digit's stretch (19 steps instead of 28
// on 2 x 10^8 composites).  One search per bucket (a bucket ends where the next one starts); k_msd_max takes the sizes from the offsets.
__global__ void k_msd_bounds(const uint64_t* __restrict__ c, uint64_t n, uint32_t shift, uint32_t nb, uint32_t* __restrict__ off, const uint32_t* __restrict__ dbase, uint32_t dbits,
                             uint32_t top_bits) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nb) return;
    uint64_t lo = 0, hi = n;
    if (dbase && r < nb) {
        const uint32_t d = r >> (top_bits - dbits);
        lo = dbase[d];
        hi = d + 1 < (1u << dbits) ? dbase[d + 1] : n;
    } else if (r == nb)
        lo = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((c[mid] >> shift) < (uint64_t)r) lo = mid + 1; else hi = mid;
    }
    off[r] = (uint32_t)lo;
}
__global__ void k_msd_max(const uint32_t* __restrict__ off, uint32_t nb, uint32_t* __restrict__ max_bucket) {
    __shared__ uint32_t wm[16];
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t m = r < nb ? off[r + 1] - off[r] : 0u;
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_down(m, o));
    if ((threadIdx.x & 63u) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {  // (one atomic per workgroup: 4096 on one word took 47 us)
        for (uint32_t w = 1; w < (blockDim.x >> 6); w++) m = max(m, wm[w]);
        if (m) atomicMax(max_bucket, m);
    }
}
__global__ void k_scatter_c(const uint64_t* __restrict__ c, uint32_t gb, uint64_t n, const uint64_t* __restrict__ pos, uint32_t* __restrict__ pg,
                            uint64_t* __restrict__ tk, uint32_t* __restrict__ seg_off) {
    const uint64_t gmask = (1ull << gb) - 1ull;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = c[i], b = i ? c[i - 1] : ~a, ps = pos[i];
        if (a != b) pg[(uint32_t)ps] = (uint32_t)(a & gmask);
        if ((a >> gb) != (b >> gb)) {
            const uint32_t q = (uint32_t)(ps >> 32);
            tk[q] = a >> gb;
            seg_off[q] = (uint32_t)ps;
        }
    }
}

// The same on-the-fly flags for sorted one-word keys with their genome ids in a second array (the general sort's output).  GT: the
// width the ids were sorted at (uint8_t / uint16_t when every id fits: the values are a third of the sort's traffic at 4 bytes).
template <class GT>
struct BftNarrowIds {  // value "iterator" of that sort: genome id i of the log, narrowed
    const uint32_t* g;
    __host__ __device__ GT operator()(uint32_t i) const { return (GT)g[i]; }
};
template <class GT>
struct BftPairFlags2 {
    const uint64_t* k;
    const GT* g;
    __host__ __device__ uint64_t operator()(uint32_t i) const {
        const uint64_t head = i == 0 || k[i] != k[i - 1], keep = head || g[i] != g[i - 1];
        return (head << 32) | keep;
    }
};
template <class GT>
__global__ void k_scatter_2(const uint64_t* __restrict__ k, const GT* __restrict__ g, uint64_t n, const uint64_t* __restrict__ pos,
                            uint32_t* __restrict__ pg, uint64_t* __restrict__ tk, uint32_t* __restrict__ seg_off) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = k[i], ps = pos[i];
        const GT ga = g[i];
        const bool head = i == 0 || a != k[i - 1], keep = head || ga != g[i - 1];
        if (keep) pg[(uint32_t)ps] = (uint32_t)ga;
        if (head) {
            const uint32_t q = (uint32_t)(ps >> 32);
            tk[q] = a;
            seg_off[q] = (uint32_t)ps;
        }
    }
}
