// bft_walk.h -- per-query trie walk over a BftImage, shared by the HIP kernels (device) and by
// the host-side index checker used in the CPU-only unit tests (tests/ only; the bft_gpu_* C-ABI
// never walks on the host).
//
// One call = isKmerPresent (reference src/presenceNode.c:1823-1921): per level presenceKmer
// (:1284-1576) = Bloom probe over the node's CCs -> filter2 bit -> rank -> cluster -> filter3
// search -> child (suffix group / child Node / leaf annotation), else the node's UC (:1554-1573).
#pragma once
#include "bft_image.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BFT_HD __host__ __device__ __forceinline__
#else
#define BFT_HD inline
#endif

// 18-bit raw prefix (nucleotide j at bits 2j, the packed layout of src/fasta.c:11-23)
//   -> P = n1..n9 with n1 in the top bits (what rev[] yields, src/presenceNode.c:1327-1329)
//   -> rotated r = n2..n9,n1 (src/presenceNode.c:1367-1371).
BFT_HD uint32_t bft_rot_prefix(uint32_t raw) {
    uint32_t x;
#if defined(__HIP_DEVICE_COMPILE__)
    x = __brev(raw) >> 14;
#else
    x = 0;
    for (int b = 0; b < 18; b++) x |= ((raw >> b) & 1u) << (17 - b);
#endif
    x = ((x & 0x2AAAAu) >> 1) | ((x & 0x15555u) << 1);  // P
    return ((x & 0xFFFFu) << 2) | (x >> 16);
}

// inverse of bft_rot_prefix (used when k-mers are rebuilt from the table)
BFT_HD uint32_t bft_unrot_prefix(uint32_t r) {
    uint32_t P = ((r & 3u) << 16) | (r >> 2);
    uint32_t x = ((P & 0x2AAAAu) >> 1) | ((P & 0x15555u) << 1);
    uint32_t raw = 0;
    for (int b = 0; b < 18; b++) raw |= ((x >> b) & 1u) << (17 - b);
    return raw;
}

// X (W little-endian u64 words of the packed k-mer) -> T-form (W words, word 0 most significant).
template <int W>
BFT_HD void bft_tform_from_x(const uint64_t* x, int L, uint64_t* t) {
    uint64_t tl[W];
#pragma unroll
    for (int w = 0; w < W; w++) tl[w] = 0;
    for (int d = 0; d < L; d++) {
        int o = 18 * d, wi = o >> 6, sh = o & 63;
        uint64_t v = x[wi] >> sh;
        if (sh > 46 && wi + 1 < W) v |= x[wi + 1] << (64 - sh);
        uint64_t r = bft_rot_prefix((uint32_t)v & 0x3FFFFu);
        int oo = 18 * (L - 1 - d), wo = oo >> 6, so = oo & 63;
        tl[wo] |= r << so;
        if (so > 46 && wo + 1 < W) tl[wo + 1] |= r >> (64 - so);
    }
#pragma unroll
    for (int w = 0; w < W; w++) t[w] = tl[W - 1 - w];
}

// T-form -> X (inverse), host side helpers only need it for extraction
template <int W>
BFT_HD void bft_x_from_tform(const uint64_t* t, int L, uint64_t* x) {
    uint64_t tl[W];
#pragma unroll
    for (int w = 0; w < W; w++) { tl[w] = t[W - 1 - w]; x[w] = 0; }
    for (int d = 0; d < L; d++) {
        int oo = 18 * (L - 1 - d), wo = oo >> 6, so = oo & 63;
        uint64_t v = tl[wo] >> so;
        if (so > 46 && wo + 1 < W) v |= tl[wo + 1] << (64 - so);
        uint64_t raw = bft_unrot_prefix((uint32_t)v & 0x3FFFFu);
        int o = 18 * d, wi = o >> 6, sh = o & 63;
        x[wi] |= raw << sh;
        if (sh > 46 && wi + 1 < W) x[wi + 1] |= raw >> (64 - sh);
    }
}

// rotated prefix of level d (0 = root) out of a T-form k-mer
template <int W>
BFT_HD uint32_t bft_digit(const uint64_t* t, int L, int d) {
    int oo = 18 * (L - 1 - d), wo = oo >> 6, so = oo & 63;
    uint64_t v = t[W - 1 - wo] >> so;
    if (so > 46 && wo + 1 < W) v |= t[W - 2 - wo] << (64 - so);
    return (uint32_t)v & 0x3FFFFu;
}

template <int W>
BFT_HD int bft_cmp(const uint64_t* a, const uint64_t* b) {
#pragma unroll
    for (int w = 0; w < W; w++) {
        if (a[w] < b[w]) return -1;
        if (a[w] > b[w]) return 1;
    }
    return 0;
}

// lower bound of t among n sorted rows; returns index in [0, n]
template <int W>
BFT_HD uint32_t bft_rows_lower_bound(const uint64_t* rows, uint32_t n, const uint64_t* t) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        uint64_t r[W];
#pragma unroll
        for (int w = 0; w < W; w++) r[w] = rows[(size_t)mid * W + w];
        if (bft_cmp<W>(r, t) < 0) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// "first CC of the node whose Bloom filter holds both bits" on the bit-sliced block.
BFT_HD int bft_first_cc(const BftImage& im, const BftNode& nd, uint32_t h1, uint32_t h2) {
    const uint8_t* blk = im.bfT + (size_t)nd.bf_off * 8;
    if (nd.bf_wb == 1) {
        uint32_t m = blk[h1] & blk[h2];
        return m ? __builtin_ctz(m) : -1;
    } else if (nd.bf_wb == 2) {
        const uint16_t* b = (const uint16_t*)blk;
        uint32_t m = b[h1] & b[h2];
        return m ? __builtin_ctz(m) : -1;
    } else if (nd.bf_wb == 4) {
        const uint32_t* b = (const uint32_t*)blk;
        uint32_t m = b[h1] & b[h2];
        return m ? __builtin_ctz(m) : -1;
    }
    const uint64_t* b = (const uint64_t*)blk;
    int nw = nd.bf_wb >> 3;
    for (int w = 0; w < nw; w++) {
        uint64_t m = b[(size_t)h1 * nw + w] & b[(size_t)h2 * nw + w];
        if (m) return w * 64 + __builtin_ctzll(m);
    }
    return -1;
}

struct BftHit {
    int present;
    uint64_t row;  // row of the k-mer in tk (valid when present)
};

template <int W>
BFT_HD BftHit bft_walk(const BftImage& im, const uint64_t* t) {
    BftHit hit;
    hit.present = 0;
    hit.row = 0;
    uint32_t node = 0;
    const int L = im.L;
    for (int d = 0; d < L; d++) {
        const BftNode nd = im.nodes[node];
        const uint32_t r = bft_digit<W>(t, L, d);
        int c = -1;
        if (nd.ncc) {
            const uint32_t hm = im.hashmod[r >> 4];  // Bloom key = n2..n8 (src/presenceNode.c:1341-1343)
            c = bft_first_cc(im, nd, hm & 0xFFFFu, hm >> 16);
        }
        if (c < 0) {
            // no Bloom-positive CC: the node's UC (src/presenceNode.c:1554-1573)
            if (nd.uc_n) {
                const uint64_t* rows = im.uck + (size_t)nd.uc_first * W;
                uint32_t z = bft_rows_lower_bound<W>(rows, nd.uc_n, t);
                if (z < nd.uc_n) {
                    uint64_t q[W];
#pragma unroll
                    for (int w = 0; w < W; w++) q[w] = rows[(size_t)z * W + w];
                    if (bft_cmp<W>(q, t) == 0) { hit.present = 1; hit.row = im.ucrow[nd.uc_first + z]; }
                }
            }
            return hit;
        }
        const BftCC cc = im.ccs[nd.cc_first + c];
        const uint32_t pu = r >> cc.s, pv = r & ((1u << cc.s) - 1u);
        const uint32_t wi = pu / BFT_F2_BITS_PER_WORD, bi = pu % BFT_F2_BITS_PER_WORD;
        const uint64_t fw = im.f2w[cc.f2_off + wi];
        if (!((fw >> bi) & 1ull)) return hit;  // filter2 miss => absent (src/presenceNode.c:1546-1548)
        const uint32_t clu = (uint32_t)(fw >> 48) + (uint32_t)__builtin_popcountll(fw & ((1ull << bi) - 1ull));
        uint32_t lo = im.clus[cc.clus_off + clu], hi = im.clus[cc.clus_off + clu + 1];
        const uint32_t end = hi;
        const uint8_t* f3 = im.f3 + cc.f3_off;
        if (cc.s == 8) {  // src/presenceNode.c:1399-1410
            while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (f3[mid] < pv) lo = mid + 1; else hi = mid; }
            if (lo >= end || f3[lo] != pv) return hit;
        } else {          // nibble-packed, src/presenceNode.c:1472-1489
            while (lo < hi) {
                uint32_t mid = (lo + hi) >> 1;
                uint32_t v = (f3[mid >> 1] >> ((mid & 1) * 4)) & 0xFu;
                if (v < pv) lo = mid + 1; else hi = mid;
            }
            if (lo >= end || (((uint32_t)f3[lo >> 1] >> ((lo & 1) * 4)) & 0xFu) != pv) return hit;
        }
        const uint64_t ch = im.child[cc.child_off + lo];
        const uint32_t cnt = (uint32_t)(ch >> BFT_CHILD_CNT_SHIFT) & 0xFFu;
        const uint64_t idx = ch & BFT_CHILD_IDX_MASK;
        if (d == L - 1) { hit.present = 1; hit.row = idx; return hit; }  // leaf: annotation row
        if (cnt == 0) { node = (uint32_t)idx; continue; }                 // child Node (src/presenceNode.c:1867)
        // suffix group of cnt rows (src/presenceNode.c:1874-1915)
        const uint64_t* rows = im.tk + idx * W;
        uint32_t z = bft_rows_lower_bound<W>(rows, cnt, t);
        if (z < cnt) {
            uint64_t q[W];
#pragma unroll
            for (int w = 0; w < W; w++) q[w] = rows[(size_t)z * W + w];
            if (bft_cmp<W>(q, t) == 0) { hit.present = 1; hit.row = idx + z; }
        }
        return hit;
    }
    return hit;
}
