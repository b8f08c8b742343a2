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
// (Tried: the rare paths -- node UC, galloping searches -- as real calls, __attribute__((noinline)).  The calling convention cost
// more scratch than the inlined arrays: 84-150 spilled VGPRs instead of 0-30.)
#define BFT_HD_RARE __host__ __device__ __forceinline__
#else
#define BFT_HD inline
#define BFT_HD_RARE inline
#endif

// Suffix groups of at least this many rows are searched with aligned block probes (bft_group_probe), smaller ones by
// galloping from the interpolated guess (0 disables the probes).  Measured on MI355X: 4 (k=63: +6 % over 8; non-temporal
// loads of the table rows: 1.7x slower).
#ifndef BFT_WINDOW_PROBE
#define BFT_WINDOW_PROBE 4
#endif

// Stage probes of tools/perf_probe.py: compiled in only with -DBFT_PERF_PROBE (make probe); a constant 0 otherwise, so the
// shipped walk carries none of these branches and no option can make its answers wrong.
#if defined(BFT_PERF_PROBE)
#define BFT_DBG_STOP(im) ((im).debug_stop)
#else
#define BFT_DBG_STOP(im) 0u
#endif

// Random 8-byte gathers from the big tables (each touched cache line is used once by the wavefront): on the
// device they can be issued as non-temporal loads; selected at compile time with -DBFT_NT_LOADS=1.
#if defined(__HIP_DEVICE_COMPILE__) && defined(BFT_NT_LOADS) && BFT_NT_LOADS == 1
#define BFT_GATHER(p) __builtin_nontemporal_load(p)
#elif defined(__HIP_DEVICE_COMPILE__) && defined(BFT_NT_LOADS) && BFT_NT_LOADS == 2
#define BFT_GATHER(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)  /* sc1: served by L2, no L1 line fill */
#else
#define BFT_GATHER(p) (*(p))
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

// 18 bits at bit offset `off` of a little-endian multiword integer (register arrays are indexed
// through unrolled selects: a dynamic index would push them to scratch memory on the GPU).
template <int W>
BFT_HD uint32_t bft_get18_le(const uint64_t* le, int off) {
    const int wi = off >> 6, sh = off & 63;
    if (W == 1) return (uint32_t)(le[0] >> sh) & 0x3FFFFu;
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int w = 0; w < W; w++) {
        if (w == wi) lo = le[w];
        if (w == wi + 1) hi = le[w];
    }
    uint64_t v = lo >> sh;
    if (sh > 46) v |= hi << (64 - sh);
    return (uint32_t)v & 0x3FFFFu;
}
template <int W>
BFT_HD void bft_or18_le(uint64_t* le, int off, uint64_t r) {
    const int wi = off >> 6, sh = off & 63;
    if (W == 1) { le[0] |= r << sh; return; }
    const uint64_t lo = r << sh, hi = sh > 46 ? r >> (64 - sh) : 0;
#pragma unroll
    for (int w = 0; w < W; w++) {
        if (w == wi) le[w] |= lo;
        if (w == wi + 1) le[w] |= hi;
    }
}

// T-form of a k-mer (any k >= 9): the L = k/9 rotated prefixes, most significant first, then -- when k is not a
// multiple of 9, an extension the reference does not have (it requires k % 9 == 0, src/main.c:61-63) -- the R = k % 9
// remaining nucleotides, first one most significant.  2k bits in W words, word 0 most significant.
// X (W little-endian u64 words of the packed k-mer) -> T-form.
template <int W>
BFT_HD void bft_tform_from_x(const uint64_t* x, int k, uint64_t* t) {
    const int L = k / 9, R = k - 9 * L, rb = 2 * R;
    uint64_t tl[W];
#pragma unroll
    for (int w = 0; w < W; w++) tl[w] = 0;
    for (int d = 0; d < L; d++) {
        const uint64_t r = bft_rot_prefix(bft_get18_le<W>(x, 18 * d));
        bft_or18_le<W>(tl, rb + 18 * (L - 1 - d), r);
    }
    if (R) {
        const uint32_t raw = bft_get18_le<W>(x, 18 * L);  // the R remaining nucleotides, first one in the low bits
        uint64_t rem = 0;
        for (int j = 0; j < R; j++) rem |= (uint64_t)((raw >> (2 * j)) & 3u) << (2 * (R - 1 - j));
        tl[0] |= rem;  // rb <= 16 bits at offset 0
    }
#pragma unroll
    for (int w = 0; w < W; w++) t[w] = tl[W - 1 - w];
}

// T-form -> X (inverse), host side helpers only need it for extraction
template <int W>
BFT_HD void bft_x_from_tform(const uint64_t* t, int k, uint64_t* x) {
    const int L = k / 9, R = k - 9 * L, rb = 2 * R;
    uint64_t tl[W];
#pragma unroll
    for (int w = 0; w < W; w++) { tl[w] = t[W - 1 - w]; x[w] = 0; }
    for (int d = 0; d < L; d++) {
        const uint64_t raw = bft_unrot_prefix(bft_get18_le<W>(tl, rb + 18 * (L - 1 - d)));
        bft_or18_le<W>(x, 18 * d, raw);
    }
    if (R) {
        const uint64_t rem = tl[0] & ((1ull << rb) - 1ull);
        uint64_t raw = 0;
        for (int j = 0; j < R; j++) raw |= ((rem >> (2 * (R - 1 - j))) & 3ull) << (2 * j);
        bft_or18_le<W>(x, 18 * L, raw);
    }
}

// rotated prefix of level d (0 = root) out of a T-form k-mer
template <int W>
BFT_HD uint32_t bft_digit(const uint64_t* t, int k, int d) {
    const int L = k / 9, rb = 2 * (k - 9 * L);
    uint64_t tl[W];
#pragma unroll
    for (int w = 0; w < W; w++) tl[w] = t[W - 1 - w];
    return bft_get18_le<W>(tl, rb + 18 * (L - 1 - d));
}

// the (up to) 36 bits that follow level d in a T-form k-mer, left-aligned in 36 bits: interpolation key of a suffix group
template <int W>
BFT_HD uint64_t bft_next36(const uint64_t* t, int k, int d) {
    const int L = k / 9, rb = 2 * (k - 9 * L);
    const int remaining = rb + 18 * (L - 1 - d);  // bits below level d
    uint64_t tl[W];
#pragma unroll
    for (int w = 0; w < W; w++) tl[w] = t[W - 1 - w];
    if (remaining >= 36) return ((uint64_t)bft_get18_le<W>(tl, remaining - 18) << 18) | bft_get18_le<W>(tl, remaining - 36);
    if (remaining > 18) {
        const uint64_t hi = bft_get18_le<W>(tl, remaining - 18), lo = tl[0] & ((1ull << (remaining - 18)) - 1ull);
        return (hi << 18) | (lo << (36 - remaining));
    }
    if (remaining > 0) return (tl[0] & ((1ull << remaining) - 1ull)) << (36 - remaining);
    return 0;
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

// One table row (W words).  Rows of 2 or 4 words start on 16-byte boundaries (the tables are 256-byte aligned), so the
// device reads them with 16-byte loads: a divergent load costs per lane and per instruction, not per byte.
template <int W>
BFT_HD void bft_load_row(const uint64_t* p, uint64_t* r) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (W == 2 || W == 4) {
#pragma unroll
        for (int w = 0; w < W; w += 2) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(p + w);
            r[w] = v.x;
            r[w + 1] = v.y;
        }
        return;
    }
#endif
#pragma unroll
    for (int w = 0; w < W; w++) r[w] = BFT_GATHER(&p[w]);
}

// lower bound of t among n sorted rows; returns index in [0, n]
template <int W>
BFT_HD uint32_t bft_rows_lower_bound(const uint64_t* rows, uint32_t n, const uint64_t* t) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        uint64_t r[W];
        bft_load_row<W>(rows + (size_t)mid * W, r);
        if (bft_cmp<W>(r, t) < 0) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// Exact search of t among the n sorted rows of a suffix group, starting from an interpolated guess g
// (suffix values are close to uniform inside a group, so the guess is a few rows off): probe, gallop
// towards the target, finish with a binary search in the bracket.  Same result as the reference's
// binary_search_UC + memcmp (src/UC.c:81-124, src/presenceNode.c:1886-1913), ~1-2 cache lines instead
// of ~log2(n).  Returns the row index or -1.
template <int W>
BFT_HD_RARE int bft_rows_find(const uint64_t* rows, uint32_t n, const uint64_t* t, uint32_t g) {
    uint64_t r[W];
    if (g >= n) g = n - 1;
    uint32_t lo = 0, hi = n;
    bft_load_row<W>(rows + (size_t)g * W, r);
    int c = bft_cmp<W>(r, t);
    if (c == 0) return (int)g;
    uint32_t step = 1;
    if (c < 0) {
        lo = g + 1;
        while (lo < hi) {
            uint32_t p = lo + step - 1;
            if (p >= hi) p = hi - 1;
            bft_load_row<W>(rows + (size_t)p * W, r);
            c = bft_cmp<W>(r, t);
            if (c == 0) return (int)p;
            if (c < 0) { lo = p + 1; step <<= 1; }
            else { hi = p; break; }
        }
    } else {
        hi = g;
        while (lo < hi) {
            uint32_t p = hi - lo > step ? hi - step : lo;
            bft_load_row<W>(rows + (size_t)p * W, r);
            c = bft_cmp<W>(r, t);
            if (c == 0) return (int)p;
            if (c > 0) { hi = p; step <<= 1; }
            else { lo = p + 1; break; }
        }
    }
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        bft_load_row<W>(rows + (size_t)mid * W, r);
        c = bft_cmp<W>(r, t);
        if (c == 0) return (int)mid;
        if (c < 0) lo = mid + 1;
        else hi = mid;
    }
    return -1;
}

// "first CC of the node whose Bloom filter holds both bits" on a bit-sliced block (blk = the node's
// block, wb = bytes per bit position).
BFT_HD int bft_first_cc_blk(const uint8_t* blk, uint32_t wb, uint32_t h1, uint32_t h2) {
    if (wb == 1) {
        uint32_t m = blk[h1] & blk[h2];
        return m ? __builtin_ctz(m) : -1;
    } else if (wb == 2) {
        const uint16_t* b = (const uint16_t*)blk;
        uint32_t m = b[h1] & b[h2];
        return m ? __builtin_ctz(m) : -1;
    } else if (wb == 4) {
        const uint32_t* b = (const uint32_t*)blk;
        uint32_t m = b[h1] & b[h2];
        return m ? __builtin_ctz(m) : -1;
    }
    const uint64_t* b = (const uint64_t*)blk;
    const int nw = (int)(wb >> 3);
    for (int w = 0; w < nw; w++) {
        uint64_t m = b[(size_t)h1 * nw + w] & b[(size_t)h2 * nw + w];
        if (m) return w * 64 + __builtin_ctzll(m);
    }
    return -1;
}

// Where the walk reads the hash table, and the root node's Bloom block and CC headers, from.
// Host tests and non-staged kernels read the image in global memory; k_query stages them in LDS
// (BftRootLds in bft_gpu.hip) because every query of a batch goes through them.
struct BftRootGlobal {
    const BftImage& im;
    BFT_HD explicit BftRootGlobal(const BftImage& i) : im(i) {}
    BFT_HD uint32_t hashmod(uint32_t key) const { return im.hashmod[key]; }
    BFT_HD int root_first_cc(const BftNode& nd, uint32_t h1, uint32_t h2) const {
        return bft_first_cc_blk(im.bfT + (size_t)nd.bf_off * 8, nd.bf_wb, h1, h2);
    }
    BFT_HD BftCCX root_cc(const BftNode& nd, int c) const { return im.ccx[nd.cc_first + c]; }
};

struct BftHit {
    int present;
    uint64_t row;      // row of the k-mer in tk (valid when present and !from_kh)
    uint32_t cs;       // its colour set, when it was found in the k-mer hash (from_kh; im.walk_kh: no row is known then)
    int from_kh;
};
// what a query kernel writes for a found k-mer: its row, or -- im.emit_cs, the colour-row and sequence paths -- its colour set
BFT_HD uint32_t bft_hit_out(const BftImage& im, const BftHit& h) { return im.emit_cs ? (h.from_kh ? h.cs : im.tcol[h.row]) : (uint32_t)h.row; }

// genome id q of the colour-set dictionary (stored in the narrowest width that holds every id of the index)
BFT_HD uint32_t bft_cs_id_at(const void* cs_ids, uint32_t cs_w, uint64_t q) {
    if (cs_w == 1) return reinterpret_cast<const uint8_t*>(cs_ids)[q];
    if (cs_w == 2) return reinterpret_cast<const uint16_t*>(cs_ids)[q];
    return reinterpret_cast<const uint32_t*>(cs_ids)[q];
}

// ---- k-mer hash (bft_image.h, BFT_KH_*) ----
// Geometry of the table of an index (BftKhGeo, bft_image.h): computed once per build on the host.
BFT_HD uint32_t bft_kh_value_bits(uint64_t n_values) {  // values 0 .. n_values - 1 are stored as 1 .. n_values
    uint32_t b = 1;
    while (b < 32 && (n_values >> b)) b++;
    return b;
}
// bits of a header field: a function of the slots per line alone (the kernels know it at compile time)
BFT_HD constexpr uint32_t bft_kh_field_bits(uint32_t S) { return 128u / S - 1u > 32u ? 32u : 128u / S - 1u; }
BFT_HD uint32_t bft_kh_body_bytes(uint32_t S) { return 48u / S; }
// n k-mers of length k, values below n_values, `load_pct` per cent of the home lines' slots in use
BFT_HD BftKhGeo bft_kh_geometry(int k, uint64_t n, uint64_t n_values, uint32_t load_pct) {
    BftKhGeo g;
    const uint32_t tb = (uint32_t)(2 * k);
    {   // the last nucleotide (bits vo, vo + 1 of the T-form) and the first (bits o, o + 1) stay out of the home line
        const uint32_t L = (uint32_t)k / 9u, rb = 2u * ((uint32_t)k - 9u * L), vo = rb ? 0u : 2u, o = rb + 18u * (L - 1u);
        g.hb0 = tb - 4u < 32u ? tb - 4u : 32u;
        const uint32_t restb0 = tb - g.hb0;
        g.mm = ~(3ull << vo);
        if (o >= restb0) { g.po = o - restb0; g.hb = g.hb0 - 2u; }
        else { g.po = 32u; g.hb = g.hb0; g.mm &= ~(3ull << o); }
        g.restb = tb - g.hb;
    }
    g.cb = bft_kh_value_bits(n_values);
    g.db = BFT_KH_DBITS_FOR(1u); g.maxd = (1u << g.db) - 1u;
    g.S = 1; g.f = 0; g.wb = 48; g.kb = 0; g.qb = 0; g.t = 0; g.m = 1; g.inv = 0; g.nl = 1;
    // Every number of slots per line, twice: with as many home lines as the occupancy asks for (nl = 2^(hb - t) m, m in [16, 32]: any size
    // within 6 %), and with that number rounded up to a power of two -- m lines per 2^t values of c cost q up to one bit more than log2 says
    // unless m is a power of two, and that bit can be the one a body lacks (config 5: k = 63, 3x10^5 colour sets -- four slots per line fit
    // with 2^22 lines where 4.05x10^6 lines allow three).  The geometry with the fewest lines wins (ties: more slots, then the exact size).
    uint64_t best_nl = ~0ull;
    for (uint32_t S = BFT_KH_MAX_SLOTS; S >= 1; S--) {
        for (uint32_t pow2 = 0; pow2 < 2; pow2++) {
            // (t <= 27: the magic division of bft_kh_key needs frac < 2^27)
            const uint64_t per = (uint64_t)S * load_pct;
            uint64_t want = (n * 100ull + per - 1) / per;
            if (want < 1) want = 1;
            uint32_t lg = 0;
            while ((want >> (lg + 1)) != 0) lg++;  // floor(log2(want))
            if (pow2) {
                if (want == (1ull << lg)) continue;  // (already a power of two: the exact geometry is this one)
                lg++;
                want = 1ull << lg;
            }
            uint32_t abits = lg > 4 ? lg - 4 : 0;   // bits of the part `a` of the hashed high bits
            if (abits > g.hb) abits = g.hb;
            if (g.hb - abits > 27u) abits = g.hb - 27u;
            const uint32_t t = g.hb - abits;
            uint64_t m = (want + (1ull << abits) - 1) >> abits;
            if (m < 2) m = 2;  // (m = 1 has no 32-bit magic number; a table is never smaller than 2^(hb - 27) * 2 lines)
            // q < ceil(2^t / m): its bits
            const uint64_t span = ((1ull << t) + m - 1) / m;
            uint32_t qb = 0;
            while (qb < 32 && ((span - 1) >> qb)) qb++;
            // (a stored key never has fewer bits than a header field: short keys -- k < 18 or so -- are padded with zeros on top)
            const uint32_t f = bft_kh_field_bits(S), kb = g.restb + qb > f ? g.restb + qb : f, wb = bft_kh_body_bytes(S);
            const uint32_t db = BFT_KH_DBITS_FOR(S);
            const uint64_t nl = m << abits;
            if ((S == 1 || g.cb + db + kb - f <= 8u * wb) && nl < best_nl) {
                best_nl = nl;
                g.db = db; g.maxd = (1u << db) - 1u;
                g.S = S; g.f = f; g.wb = wb; g.kb = kb; g.qb = qb; g.t = t; g.m = (uint32_t)m;
                g.inv = (uint32_t)(((1ull << 32) + m - 1) / m);
                g.nl = nl;
            }
        }
    }
    return g;
}
// T-form k-mer -> its hashed bits (g.hb of them: the top hb0 bits without the first nucleotide's two) and the rest (g.restb bits: the bits
// below the top hb0, the first nucleotide's two on top of them), as W little-endian words
template <int W>
BFT_HD uint32_t bft_kh_split(const uint64_t* t, int k, const BftKhGeo& g, uint64_t* restle) {
#pragma unroll
    for (int i = 0; i < W; i++) restle[i] = t[W - 1 - i];
    const int tb0 = 2 * k - 64 * (W - 1);  // bits of the T-form in its top word t[0]
    const uint32_t hb = g.hb0;
    uint32_t hi;
    if (tb0 >= (int)hb) {
        hi = (uint32_t)(t[0] >> (tb0 - (int)hb));
        restle[W - 1] = tb0 > (int)hb ? restle[W - 1] & ((1ull << (tb0 - (int)hb)) - 1ull) : 0ull;
    } else {  // (W > 1: the hashed bits straddle the two top words)
        const int below = (int)hb - tb0;  // bits taken from t[1]
        hi = (uint32_t)((t[0] << below) | (t[W > 1 ? 1 : 0] >> (64 - below)));
        restle[W - 1] = 0;
        restle[W > 1 ? W - 2 : 0] &= (1ull << (64 - below)) - 1ull;
    }
    if (hb < 32u) hi &= (1u << hb) - 1u;
    if (g.po < 32u) {  // the first nucleotide's bits leave the hashed ones for the top of the rest (an even position: never across two words)
        const uint64_t e = (hi >> g.po) & 3u;
        hi = (hi & ((1u << g.po) - 1u)) | ((hi >> (g.po + 2u)) << g.po);
        const uint32_t at = g.restb - 2u, wi = at >> 6, sh = at & 63u;
#pragma unroll
        for (int i = 0; i < W; i++)
            if ((uint32_t)i == wi) restle[i] |= e << sh;
    }
    return hi;
}
// 32 well-mixed bits of the rest -- without the bits of the k-mer's last and first nucleotide
template <int W>
BFT_HD uint32_t bft_kh_mix(const uint64_t* restle, const BftKhGeo& g) {
    uint64_t r[W];
#pragma unroll
    for (int i = 0; i < W; i++) r[i] = restle[i];
    r[0] &= g.mm;
    if (g.po < 32u) {
        const uint32_t at = g.restb - 2u, wi = at >> 6, sh = at & 63u;
#pragma unroll
        for (int i = 0; i < W; i++)
            if ((uint32_t)i == wi) r[i] &= ~(3ull << sh);
    }
    uint64_t h = r[0] ^ 0x2545F4914F6CDD1Dull;
#pragma unroll
    for (int w = 1; w < W; w++) h = (h ^ (h >> 29)) * 0x9E3779B97F4A7C15ull + r[w];
    h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33;  // 64-bit finalizer (murmur3)
    return (uint32_t)(h >> 32);
}
// A bijection of the hb-bit integers that scatters them (murmur3's 32-bit finalizer, cut down to hb bits: xor-shifts and odd multipliers
// are invertible modulo 2^hb) and its inverse.  The hashed high bits of the T-form are the root prefix and the start of the next digit:
// k-mers that share them (a deep trie) must not share their home lines.
BFT_HD uint32_t bft_kh_perm(uint32_t x, uint32_t hb) {
    const uint32_t M = hb < 32u ? (1u << hb) - 1u : 0xFFFFFFFFu, s1 = hb / 2u ? hb / 2u : 1u, s2 = (hb * 13u) / 32u ? (hb * 13u) / 32u : 1u;
    x &= M;
    x ^= x >> s1; x = (x * 0x85EBCA6Bu) & M;
    x ^= x >> s2; x = (x * 0xC2B2AE35u) & M;
    x ^= x >> s1;
    return x;
}
BFT_HD uint32_t bft_kh_unshift(uint32_t y, uint32_t s) {  // x with y = x ^ (x >> s)
    uint32_t x = y;
    for (uint32_t k = s; k < 32u; k += s) x ^= y >> k;
    return x;
}
BFT_HD uint32_t bft_kh_perm_inv(uint32_t x, uint32_t hb) {
    const uint32_t M = hb < 32u ? (1u << hb) - 1u : 0xFFFFFFFFu, s1 = hb / 2u ? hb / 2u : 1u, s2 = (hb * 13u) / 32u ? (hb * 13u) / 32u : 1u;
    x &= M;
    x = bft_kh_unshift(x, s1); x = (x * 0x7ED1B41Du) & M;  // (0xC2B2AE35^-1 mod 2^32)
    x = bft_kh_unshift(x, s2); x = (x * 0xA5CB9243u) & M;  // (0x85EBCA6B^-1 mod 2^32)
    x = bft_kh_unshift(x, s1);
    return x & M;
}
// A k-mer as the table sees it: its home line; `field` = the low f bits of its stored key (what a line's header holds); `body` = value
// (to be OR-ed in) at [0, cb), displacement at [cb, cb + db) (to be OR-ed in), the other key bits above; `bmask` = displacement + key bits.
// The stored key = the bits below the hashed ones, shifted up by qb, | q: the home line stands for the rest of the hashed bits.
//   hi' = perm(hi ^ mix(rest))  (a bijection of hi for every rest)  = (a : hb - t bits | c : t bits)
//   line = a m + sub,  sub = floor(c m / 2^t),  q = floor((c m mod 2^t) / m) = c - ceil(sub 2^t / m)
template <int W>
struct BftKhKey {
    uint64_t home, field, body[W], bmask[W];
};
template <int W>
BFT_HD void bft_kh_key(const uint64_t* t, int k, const BftKhGeo& g, BftKhKey<W>& key) {
    uint64_t restle[W], kle[W];
    const uint32_t hi = bft_kh_split<W>(t, k, g, restle);
    const uint32_t hp = bft_kh_perm(hi ^ bft_kh_mix<W>(restle, g), g.hb);
    const uint64_t a = g.t < 32u ? (uint64_t)(hp >> g.t) : 0ull, c = g.t < 32u ? (uint64_t)(hp & ((1u << g.t) - 1u)) : (uint64_t)hp;
    const uint64_t cm = c * g.m, sub = cm >> g.t, frac = cm & ((1ull << g.t) - 1ull);
    const uint64_t q = (frac * g.inv) >> 32;  // = frac / m (frac < 2^27, m <= 32)
    key.home = a * g.m + sub;
    // stored key (kb bits) = rest << qb | q
#pragma unroll
    for (int i = 0; i < W; i++) kle[i] = g.qb ? (restle[i] << g.qb) | (i > 0 ? restle[i - 1] >> (64 - g.qb) : 0ull) : restle[i];
    kle[0] |= q;
    const uint32_t f = g.f, sh = g.cb + g.db, hbits = g.kb - f;
    key.field = f ? kle[0] & ((1ull << f) - 1ull) : 0ull;
    uint64_t hi_k[W], m[W];
#pragma unroll
    for (int i = 0; i < W; i++) {
        hi_k[i] = f ? (kle[i] >> f) | (i + 1 < W ? kle[i + 1] << (64 - f) : 0ull) : kle[i];
        const int lo = 64 * i;
        m[i] = (int)hbits >= lo + 64 ? ~0ull : ((int)hbits > lo ? (1ull << (hbits - lo)) - 1ull : 0ull);
        hi_k[i] &= m[i];
    }
    key.body[0] = hi_k[0] << sh;
    key.bmask[0] = (m[0] << sh) | ((((uint64_t)1 << g.db) - 1ull) << g.cb);
#pragma unroll
    for (int i = 1; i < W; i++) {
        key.body[i] = (hi_k[i] << sh) | (hi_k[i - 1] >> (64 - sh));
        key.bmask[i] = (m[i] << sh) | (m[i - 1] >> (64 - sh));
    }
}
// 64 bits of a multiword little-endian bit string from bit `o` on (zeros beyond its NW words)
template <int NW>
BFT_HD uint64_t bft_kh_bits_at(const uint64_t* ln, uint32_t o) {
    const uint32_t wi = o >> 6, sh = o & 63u;
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {  // (selects, not a dynamic index: the words stay in registers)
        if ((uint32_t)i == wi) lo = ln[i];
        if ((uint32_t)i == wi + 1) hi = ln[i];
    }
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}
// the header of a line (two words): one 16-byte load
BFT_HD void bft_kh_load_header(const uint64_t* line, uint64_t* hd) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(line);
    hd[0] = v.x;
    hd[1] = v.y;
#else
    hd[0] = line[0];
    hd[1] = line[1];
#endif
}
// The body of slot s (W words, zero-extended): byte offset 16 + s wb of the line, wb bytes.  One load instruction per 16 bytes, at the
// body's byte address; a load that would run past the line's end starts earlier and is shifted (a load never leaves its own 64 bytes).
template <int W>
BFT_HD void bft_kh_load_body(const uint64_t* line, uint32_t s, uint32_t wb, uint64_t* body) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(line) + 16u + s * wb;
#pragma unroll
    for (int i = 0; i < W; i++) body[i] = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    struct __attribute__((packed, aligned(1))) U8 { uint64_t v; };
    struct __attribute__((packed, aligned(1))) U16 { uint64_t a, b; };
    if (W == 1) {  // wb <= 8: eight bytes that end no later than the line
        const uint32_t off = 16u + s * wb, back = off + 8u > 64u ? off + 8u - 64u : 0u;
        const uint64_t v = reinterpret_cast<const U8*>(p - back)->v;
        body[0] = v >> (8u * back);
    } else {
        const uint32_t off = 16u + s * wb;
#pragma unroll
        for (int i = 0; i < W; i += 2) {
            if (8u * (uint32_t)i >= wb) break;
            const uint32_t at = off + 8u * (uint32_t)i, back = at + 16u > 64u ? at + 16u - 64u : 0u;  // back < 16
            const U16 v = *reinterpret_cast<const U16*>(p + 8 * i - back);
            const uint64_t a = back >= 8u ? v.b : v.a, b2 = back >= 8u ? 0ull : v.b;
            const uint32_t sh = 8u * (back & 7u);
            const uint64_t lo = sh ? (a >> sh) | (b2 << (64 - sh)) : a, hi = sh ? b2 >> sh : b2;
            body[i] = lo;
            if (i + 1 < W) body[i + 1] = hi;
        }
    }
#else
    for (uint32_t b = 0; b < wb && b < 8u * W; b++) body[b >> 3] |= (uint64_t)p[b] << (8 * (b & 7));
#endif
    if (wb < 8u * W) {  // (bytes behind the body belong to the next slot)
#pragma unroll
        for (int i = 0; i < W; i++) {
            const uint32_t lo = 8u * (uint32_t)i;
            if (wb <= lo) body[i] = 0;
            else if (wb < lo + 8u) body[i] &= (1ull << (8u * (wb - lo))) - 1ull;
        }
    }
}
// A full line, d >= 1 lines past the home line of the k-mer(s) looked for, that does not hold them: may the search stop here?  The table is
// filled in (home line, T-form) order, so the displacements of a line's entries do not increase from slot to slot; when the LAST slot's entry
// is displaced by less than d, its home line lies behind the one looked for, every k-mer of that home line lies before it -- in the lines
// already seen -- and the k-mer is absent (an overflowed k-mer cannot be concerned: its place would lie more than maxd lines from home,
// behind this entry).  A tombstone in the last slot says nothing: the search goes on.  One more load from the line, on the rare path only;
// with one or two slots per line (k >= 97) it halves the lines an absent k-mer costs.
template <int W>
BFT_HD bool bft_kh_line_ends_search(const BftImage& im, const uint64_t* line, uint32_t S, uint32_t wb, uint32_t d) {
    if (d == 0) return false;
    uint64_t b0[1];
    bft_kh_load_body<1>(line, S - 1u, wb, b0);
    const uint32_t cb = im.kh.cb;
    const uint64_t v = b0[0] & ((1ull << cb) - 1ull);
    const uint32_t dl = (uint32_t)(b0[0] >> cb) & ((1u << im.kh.db) - 1u);
    return v != 0 && dl < d;
}
// One line, `d` lines past the k-mer's home line, against the k-mer: 1 = found (*val = its value), 0 = not here and the line has a free
// slot (absent), -1 = not here, line full.  The header (16 bytes: S fields of f bits -- the low key bits of the slots --, the S occupancy
// bits on top) says which slots can hold the k-mer at all; only those slots' bodies are read: an absent k-mer costs one load instruction,
// a stored one two.  SS > 0: the slots per line as a compile-time constant (the kernels of bft_kh.hip); SS == 0: read from the image.
template <int W, int SS>
BFT_HD int bft_kh_scan(const BftImage& im, const uint64_t* line, const uint64_t* hd, const BftKhKey<W>& key, uint32_t d, uint32_t* val) {
    const uint32_t S = SS > 0 ? (uint32_t)SS : im.kh.S, f = SS > 0 ? bft_kh_field_bits(SS > 0 ? SS : 1) : im.kh.f, wb = SS > 0 ? 48u / (SS > 0 ? SS : 1) : im.kh.wb, cb = im.kh.cb;
    const uint64_t fmask = f ? (1ull << f) - 1ull : 0ull, vmask = (1ull << cb) - 1ull;
    const uint32_t occ = (uint32_t)(hd[1] >> (64u - S));  // bit s: slot s is in use
    uint32_t cand = 0;
#pragma unroll
    for (uint32_t s = 0; s < (SS > 0 ? (uint32_t)SS : BFT_KH_MAX_SLOTS); s++) {
        if (SS == 0 && s >= S) break;
        const uint64_t fld = f ? bft_kh_bits_at<2>(hd, s * f) & fmask : 0ull;
        cand |= (fld == key.field ? 1u : 0u) << s;
    }
    cand &= occ;
    while (cand) {  // (a second candidate: two slots whose keys share their low f bits -- once in thousands of lines)
        const uint32_t s = (uint32_t)__builtin_ctz(cand);
        cand &= cand - 1u;
        uint64_t body[W];
        bft_kh_load_body<W>(line, s, wb, body);
        bool same = ((body[0] ^ (key.body[0] | ((uint64_t)d << cb))) & key.bmask[0]) == 0 && (body[0] & vmask) != 0;  // (value 0: a tombstone, below)
#pragma unroll
        for (int i = 1; i < W; i++) same = same && ((body[i] ^ key.body[i]) & key.bmask[i]) == 0;
        if (same) { *val = (uint32_t)(body[0] & vmask) - 1u; return 1; }
    }
    if (occ != (1u << S) - 1u) return 0;
    return bft_kh_line_ends_search<W>(im, line, S, wb, d) ? 0 : -1;
}
// The k-mers that differ from `key`'s only in the two bits b, b + 1 of the rest (the four successors of a k-mer: its last nucleotide; the four
// predecessors: its first -- they share their home line, bft_image.h): the comparison masks without those two stored key bits.
template <int W>
struct BftKhFamily {
    uint64_t fkeep, bkeep[W];
};
template <int W>
BFT_HD void bft_kh_family(const BftKhGeo& g, const BftKhKey<W>& key, uint32_t b, BftKhFamily<W>& fam) {
    fam.fkeep = g.f ? (1ull << g.f) - 1ull : 0ull;
#pragma unroll
    for (int i = 0; i < W; i++) fam.bkeep[i] = key.bmask[i];
#pragma unroll
    for (uint32_t j = 0; j < 2; j++) {
        const uint32_t pos = g.qb + b + j;
        if (pos < g.f) fam.fkeep &= ~(1ull << pos);
        else {
            const uint32_t bp = pos - g.f + g.cb + g.db;
#pragma unroll
            for (int i = 0; i < W; i++)
                if ((uint32_t)i == (bp >> 6)) fam.bkeep[i] &= ~(1ull << (bp & 63u));
        }
    }
}
// how many k-mers of the family the line `line` (in memory) holds, d lines past their home: header first, the bodies of the slots whose fields
// match; returns -1 when the line is full (the family may go on behind it)
template <int W, int SS>
BFT_HD int bft_kh_count_line(const BftImage& im, const uint64_t* line, const BftKhKey<W>& key, const BftKhFamily<W>& fam, uint32_t d, int* count) {
    const uint32_t S = SS > 0 ? (uint32_t)SS : im.kh.S, f = SS > 0 ? bft_kh_field_bits(SS > 0 ? SS : 1) : im.kh.f, wb = SS > 0 ? 48u / (SS > 0 ? SS : 1) : im.kh.wb, cb = im.kh.cb;
    const uint64_t vmask = (1ull << cb) - 1ull;
    uint64_t hd[2];
    bft_kh_load_header(line, hd);
    const uint32_t occ = (uint32_t)(hd[1] >> (64u - S));
    uint32_t cand = 0;
#pragma unroll
    for (uint32_t s = 0; s < (SS > 0 ? (uint32_t)SS : BFT_KH_MAX_SLOTS); s++) {
        if (SS == 0 && s >= S) break;
        const uint64_t fld = f ? bft_kh_bits_at<2>(hd, s * f) : 0ull;
        cand |= (((fld ^ key.field) & fam.fkeep) == 0 ? 1u : 0u) << s;
    }
    cand &= occ;
    while (cand) {
        const uint32_t s = (uint32_t)__builtin_ctz(cand);
        cand &= cand - 1u;
        uint64_t body[W];
        bft_kh_load_body<W>(line, s, wb, body);
        bool same = ((body[0] ^ (key.body[0] | ((uint64_t)d << cb))) & fam.bkeep[0]) == 0 && (body[0] & vmask) != 0;
#pragma unroll
        for (int i = 1; i < W; i++) same = same && ((body[i] ^ key.body[i]) & fam.bkeep[i]) == 0;
        *count += same ? 1 : 0;
    }
    if (occ != (1u << S) - 1u) return 0;
    return bft_kh_line_ends_search<W>(im, line, S, wb, d) ? 0 : -1;
}
// The overflow list (bft_image.h): binary search of the sorted k-mers.
template <int W>
BFT_HD bool bft_kh_overflow_find(const BftImage& im, const uint64_t* t, uint32_t* val) {
    uint32_t lo = 0, hi = im.kh_ovf_n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        uint64_t r[W];
#pragma unroll
        for (int w = 0; w < W; w++) r[w] = im.kh_ovf[(size_t)mid * W + w];
        const int c = bft_cmp<W>(r, t);
        if (c == 0) { *val = im.kh_ovf_val[mid]; return true; }
        if (c < 0) lo = mid + 1; else hi = mid;
    }
    return false;
}
// Lookup of a T-form k-mer: true when stored, *val = its value (the colour-set id).  Lines from the home line on: the key, or a line with
// a free slot, ends it; a run of full lines as long as the table's largest displacement hands over to the overflow list.
template <int W, int SS>
BFT_HD bool bft_kh_lookup(const BftImage& im, const uint64_t* t, uint32_t* val) {
    BftKhKey<W> key;
    bft_kh_key<W>(t, im.k, im.kh, key);
    for (uint32_t d = 0; d <= im.kh.maxd; d++) {
        const uint64_t* line = im.kh_lines + (key.home + d) * BFT_KH_LINE_WORDS;
        uint64_t hd[2];
        bft_kh_load_header(line, hd);
        const int res = bft_kh_scan<W, SS>(im, line, hd, key, d, val);
        if (res > 0) return true;
        if (res == 0) return false;
    }
    return im.kh_ovf_n ? bft_kh_overflow_find<W>(im, t, val) : false;
}
// What a k-mer stored d lines past its home line adds to slot s of that line: the 8 line words to OR in (header field, occupancy, body).
template <int W>
BFT_HD void bft_kh_slot_image(const uint64_t* t, int k, const BftKhGeo& g, uint32_t s, uint32_t d, uint32_t val, uint64_t* img) {
    BftKhKey<W> key;
    bft_kh_key<W>(t, k, g, key);
    key.body[0] |= ((uint64_t)val + 1ull) | ((uint64_t)d << g.cb);
#pragma unroll
    for (uint32_t i = 0; i < BFT_KH_LINE_WORDS; i++) img[i] = 0;
    // header: field at bits [s f, (s + 1) f), occupancy at bit 128 - S + s
    const uint32_t fo = s * g.f;
    if (g.f) {
        img[fo >> 6] |= key.field << (fo & 63u);
        if ((fo & 63u) + g.f > 64u) img[(fo >> 6) + 1] |= key.field >> (64u - (fo & 63u));
    }
    img[1] |= 1ull << (64u - g.S + s);
    // body at byte 16 + s wb
    const uint32_t bo = 128u + 8u * s * g.wb;
#pragma unroll
    for (int i = 0; i < W; i++) {
        const uint32_t o = bo + 64u * (uint32_t)i, wi = o >> 6, sh = o & 63u;
        if (wi < BFT_KH_LINE_WORDS) img[wi] |= key.body[i] << sh;
        if (sh && wi + 1 < BFT_KH_LINE_WORDS) img[wi + 1] |= key.body[i] >> (64u - sh);
    }
}
// A used slot of line `ln` back to its k-mer (T-form, W words, word 0 most significant) and its value.
template <int W>
BFT_HD void bft_kh_slot_decode(const BftImage& im, const uint64_t* line, const uint64_t* hd, uint64_t ln, uint32_t s, uint64_t* t, uint32_t* val) {
    const BftKhGeo& g = im.kh;
    const uint32_t f = g.f, cb = g.cb, sh = cb + g.db, hbits = g.kb - f;
    uint64_t body[W], kle[W], restle[W];
    bft_kh_load_body<W>(line, s, g.wb, body);
    *val = (uint32_t)(body[0] & ((1ull << cb) - 1ull)) - 1u;
    const uint32_t d = (uint32_t)(body[0] >> cb) & ((1u << g.db) - 1u);
    const uint64_t fld = f ? bft_kh_bits_at<2>(hd, s * f) & ((1ull << f) - 1ull) : 0ull;
#pragma unroll
    for (int i = 0; i < W; i++) {
        uint64_t x = (body[i] >> sh) | (i + 1 < W ? body[i + 1] << (64 - sh) : 0ull);
        const int lo = 64 * i;
        x &= (int)hbits >= lo + 64 ? ~0ull : ((int)hbits > lo ? (1ull << (hbits - lo)) - 1ull : 0ull);
        restle[i] = x;  // (key bits above the field, for now)
    }
#pragma unroll
    for (int i = 0; i < W; i++) kle[i] = f ? (restle[i] << f) | (i > 0 ? restle[i - 1] >> (64 - f) : 0ull) : restle[i];
    kle[0] |= fld;
    // stored key = rest << qb | q; home line = a m + sub -> c = ceil(sub 2^t / m) + q -> hi' -> hi
    const uint64_t q = g.qb ? kle[0] & ((1ull << g.qb) - 1ull) : 0ull;
#pragma unroll
    for (int i = 0; i < W; i++) restle[i] = g.qb ? (kle[i] >> g.qb) | (i + 1 < W ? kle[i + 1] << (64 - g.qb) : 0ull) : kle[i];
    const uint64_t home = ln - d, a = home / g.m, sub = home % g.m;
    const uint64_t c = ((sub << g.t) + g.m - 1) / g.m + q;
    const uint32_t hp = (uint32_t)((g.t < 32u ? a << g.t : 0ull) | c);
    uint32_t hi = bft_kh_perm_inv(hp, g.hb) ^ bft_kh_mix<W>(restle, g);
    if (g.hb < 32u) hi &= (1u << g.hb) - 1u;
    uint32_t restb0 = g.restb;
    if (g.po < 32u) {  // the first nucleotide's bits: from the top of the rest back into the hashed bits
        restb0 = g.restb - 2u;
        const uint32_t wi = restb0 >> 6, s2 = restb0 & 63u;
        uint64_t e = 0;
#pragma unroll
        for (int i = 0; i < W; i++)
            if ((uint32_t)i == wi) { e = (restle[i] >> s2) & 3ull; restle[i] &= ~(3ull << s2); }
        hi = (hi & ((1u << g.po) - 1u)) | ((uint32_t)e << g.po) | ((hi >> g.po) << (g.po + 2u));
    }
    // T = hi << restb0 | rest
    uint64_t tl[W];
#pragma unroll
    for (int i = 0; i < W; i++) tl[i] = restle[i];
    {
        const uint32_t o = restb0, wi = o >> 6, s2 = o & 63u;
#pragma unroll
        for (int i = 0; i < W; i++) {
            if ((uint32_t)i == wi) tl[i] |= (uint64_t)hi << s2;
            if ((uint32_t)i == wi + 1 && s2 > 32u) tl[i] |= (uint64_t)hi >> (64u - s2);
        }
    }
#pragma unroll
    for (int i = 0; i < W; i++) t[i] = tl[W - 1 - i];
}

// The node's UC (src/presenceNode.c:1554-1573): exact search among its < 255 rows.
template <int W>
BFT_HD_RARE void bft_uc_search(const BftImage& im, const BftNode& nd, const uint64_t* t, BftHit& hit) {
    if (!nd.uc_n) return;
    const uint64_t* rows = im.uck + (size_t)nd.uc_first * W;
    const uint32_t z = bft_rows_lower_bound<W>(rows, nd.uc_n, t);
    if (z < nd.uc_n) {
        uint64_t q[W];
        bft_load_row<W>(rows + (size_t)z * W, q);
        if (bft_cmp<W>(q, t) == 0) { hit.present = 1; hit.row = im.ucrow[nd.uc_first + z]; }
    }
}

// Prefix r inside one CC: filter2 bit -> rank -> cluster -> filter3 search (src/presenceNode.c:1381-1489, findCluster
// :1578-1821), or the two loads of the flat form.  Returns false when the CC does not hold r; else *e = its prefix entry.
// `stop` (perf probing only) != 0 ends the walk early with a dummy result in *e.
BFT_HD bool bft_cc_lookup(const BftImage& im, const BftCCX& cc, uint32_t r, uint64_t* e, bool* stop) {
    *stop = false;
    if (cc.flat) {
        // bit r of the prefix bitmap (= filter2 bit of p_u and p_v in that cluster's filter3 run); its rank = the entry
        const uint32_t wi = r / BFT_F2_BITS_PER_WORD, bi = r % BFT_F2_BITS_PER_WORD;
        const uint64_t fw = BFT_GATHER(&im.f18[cc.f18_off + wi]);
        if (BFT_DBG_STOP(im) == 2 || BFT_DBG_STOP(im) == 3) { *e = fw; *stop = true; return true; }
        if (!((fw >> bi) & 1ull)) return false;
        *e = BFT_GATHER(&im.fent[cc.fent_off + (uint32_t)(fw >> 48) + (uint32_t)__builtin_popcountll(fw & ((1ull << bi) - 1ull))]);
        return true;
    }
    const uint32_t pu = r >> cc.s, pv = r & ((1u << cc.s) - 1u);
    const uint32_t wi = pu / BFT_F2_BITS_PER_WORD, bi = pu % BFT_F2_BITS_PER_WORD;
    const uint64_t fw = im.f2w[cc.f2_off + wi];
    if (BFT_DBG_STOP(im) == 2) { *e = fw; *stop = true; return true; }
    if (!((fw >> bi) & 1ull)) return false;  // filter2 miss (src/presenceNode.c:1546-1548)
    const uint32_t clu = (uint32_t)(fw >> 48) + (uint32_t)__builtin_popcountll(fw & ((1ull << bi) - 1ull));
    uint64_t v = BFT_GATHER(&im.clus[cc.clus_off + clu]);
    if (BFT_DBG_STOP(im) == 3) { *e = v; *stop = true; return true; }
    if (v & BFT_CLUS_MULTI) {
        // p_v search inside the cluster (src/presenceNode.c:1399-1410 / :1472-1489) on the fused entries
        const uint64_t* ch = im.child + cc.child_off;
        uint32_t lo = (uint32_t)v, hi = lo + (uint32_t)((v >> BFT_CLUS_LEN_SHIFT) & 0xFFFFu);
        const uint32_t end = hi;
        v = 0;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint64_t m = BFT_GATHER(&ch[mid]);
            if (((uint32_t)(m >> BFT_CHILD_PV_SHIFT) & 0xFFu) < pv) lo = mid + 1;
            else { hi = mid; v = m; }
        }
        if (lo >= end) return false;
    }
    if (((uint32_t)(v >> BFT_CHILD_PV_SHIFT) & 0xFFu) != pv) return false;
    *e = v;
    return true;
}

// ---- node prefix hash (bft_image.h): prefix entries of the nodes below the root, keyed by (node, rotated prefix) ----
BFT_HD uint64_t bft_nph_key(uint32_t node, uint32_t r) { return ((uint64_t)node << 18) | r; }
BFT_HD uint64_t bft_nph_bucket(uint64_t key, uint64_t mask) {
    uint64_t h = key * 0x9E3779B97F4A7C15ull;
    h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 29;
    return h & mask;
}
// 1: found (*e = the entry); 0: the bucket has a free slot and no such key -- no CC of the node holds r; -1: full bucket, undecided
BFT_HD int bft_nph_lookup(const BftImage& im, uint32_t node, uint32_t r, uint64_t* e) {
    const uint64_t key = bft_nph_key(node, r);
    const uint64_t* b = im.nph + bft_nph_bucket(key, im.nph_mask) * (2 * BFT_NPH_SLOTS);
    uint64_t kv[BFT_NPH_SLOTS][2];
#pragma unroll
    for (int s = 0; s < BFT_NPH_SLOTS; s++) bft_load_row<2>(b + 2 * s, kv[s]);  // four 16-byte loads of one 64-byte line
    bool free_slot = false;
#pragma unroll
    for (int s = 0; s < BFT_NPH_SLOTS; s++) {
        if (kv[s][0] == key) { *e = kv[s][1]; return 1; }
        free_slot = free_slot || kv[s][0] == BFT_NPH_EMPTY;
    }
    return free_slot ? 0 : -1;
}
// Host-side / single-thread insertion (the GPU kernel k_nph_fill claims slots with atomicCAS instead): false when the bucket is full.
BFT_HD bool bft_nph_insert_seq(uint64_t* tab, uint64_t mask, uint32_t node, uint32_t r, uint64_t ent) {
    const uint64_t key = bft_nph_key(node, r);
    uint64_t* b = tab + bft_nph_bucket(key, mask) * (2 * BFT_NPH_SLOTS);
    for (int s = 0; s < BFT_NPH_SLOTS; s++)
        if (b[2 * s] == BFT_NPH_EMPTY) { b[2 * s] = key; b[2 * s + 1] = ent; return true; }
    return false;
}

// One entry of the root direct table (BFT_RDIR_*, bft_image.h): the root level's Bloom probe + CC lookup for prefix r, i.e.
// steps (2)-(6) of presenceKmer (src/presenceNode.c:1341-1489) evaluated for one of the 2^18 possible prefixes.
template <class Root>
BFT_HD uint64_t bft_root_direct_entry(const BftImage& im, const Root& root, const BftNode& nd, uint32_t r) {
    if (!nd.ncc) return BFT_RDIR_NO_CC;
    const uint32_t hm = root.hashmod(r >> 4);
    const int c = root.root_first_cc(nd, hm & 0xFFFFu, hm >> 16);
    if (c < 0) return BFT_RDIR_NO_CC;
    uint64_t e = 0;
    bool stop = false;
    const BftCCX cc = root.root_cc(nd, c);
    if (!bft_cc_lookup(im, cc, r, &e, &stop)) return BFT_RDIR_ABSENT;
    return e | BFT_RDIR_VALID;
}

// rstart[r] (bft_image.h) for one r in [0, 2^18]: lower bound of r among the root prefixes of the sorted table, tagged special
// unless the rdir entry v of r says "plain suffix group of exactly these rows" or the range is empty and the prefix absent.
template <int W>
BFT_HD uint32_t bft_root_range_entry(const BftImage& im, uint32_t r, uint64_t v /* rdir[r], ignored for r == 2^18 */, uint32_t root_uc_n) {
    uint64_t lo = 0, hi = im.n_kmers;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        uint64_t row[W];
#pragma unroll
        for (int w = 0; w < W; w++) row[w] = im.tk[mid * W + w];
        if (bft_digit<W>(row, im.k, 0) < r) lo = mid + 1; else hi = mid;
    }
    const uint32_t s = (uint32_t)lo;
    if (r >= (1u << 18)) return s;
    bool plain = false;
    if (v == BFT_RDIR_ABSENT || (v == BFT_RDIR_NO_CC && root_uc_n == 0)) plain = true;  // no k-mer under r (bft_root_range_check verifies the range is empty)
    else if (v & BFT_RDIR_VALID) {
        const uint64_t e = v & ~BFT_RDIR_VALID;
        const uint32_t cnt = (uint32_t)(e >> BFT_CHILD_CNT_SHIFT) & 0xFFu;
        plain = im.L > 1 && cnt >= 1 && (e & BFT_CHILD_IDX_MASK) == lo;  // a suffix group (count 0 = child Node) that starts at this row
    }
    return plain ? s : (s | BFT_RSTART_SPECIAL);
}
// second pass, once every rstart word is written: a plain prefix must own exactly the rows its rdir entry says (none for an
// absent prefix); anything else is demoted to special -- the tables can then only ever send a lookup down a path that is exact.
BFT_HD bool bft_root_range_ok(uint32_t a, uint32_t b, uint64_t v) {
    const uint32_t cnt = (b & ~BFT_RSTART_SPECIAL) - (a & ~BFT_RSTART_SPECIAL);
    if (v & BFT_RDIR_VALID) return cnt == ((uint32_t)((v & ~BFT_RDIR_VALID) >> BFT_CHILD_CNT_SHIFT) & 0xFFu);
    return cnt == 0;
}

// rq[r] (BFT_RQ_*, bft_image.h) of a root group [a, b): offsets of the first rows whose next two key bits are >= 1, 2, 3; 0 for a
// special, empty or oversized range (the walk never consults those).
template <int W>
BFT_HD uint32_t bft_root_quartile_entry(const BftImage& im, uint32_t a_raw, uint32_t b_raw) {
    const uint32_t a = a_raw, b = b_raw & ~BFT_RSTART_SPECIAL;
    if ((a & BFT_RSTART_SPECIAL) || b <= a || b - a > 255u) return 0u;
    uint32_t q = 0;
    for (uint32_t j = 1; j < 4; j++) {
        uint32_t lo = a, hi = b;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            uint64_t row[W];
#pragma unroll
            for (int w = 0; w < W; w++) row[w] = im.tk[(uint64_t)mid * W + w];
            if ((uint32_t)(bft_next36<W>(row, im.k, 0) >> 34) < j) lo = mid + 1; else hi = mid;
        }
        q |= (lo - a) << (8 * (j - 1));
    }
    return q;
}

// Two adjacent one-word rows of the table, first one at an even index: one 16-byte load.  Device buffers carry 256 bytes
// of slack (bft_pool_alloc), so a pair that straddles the end of the table is still readable; the host copy is clamped.
BFT_HD void bft_load_pair(const BftImage& im, uint64_t gi, uint64_t* a, uint64_t* b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(im.tk + gi);
    *a = v.x;
    *b = v.y;
#else
    *a = gi < im.n_kmers ? im.tk[gi] : ~0ull;
    *b = gi + 1 < im.n_kmers ? im.tk[gi + 1] : ~0ull;
#endif
}

// Search of a one-word suffix group [idx, idx+cnt) of the table for tv, from the interpolated guess g (src/UC.c:81-124 and
// src/presenceNode.c:1886-1913 give the same answer by binary search + memcmp).  A probe = NR/2 independent 16-byte loads
// of the NR-row ALIGNED block of the table that holds the guess (32 or 64 bytes: never straddles a cache line); rows of the
// block outside the bracket are ignored.  tv inside the block's range ends the search; otherwise the bracket shrinks to the
// side that can still hold tv and the next probe goes to the adjacent block (im.probe_big == 0, 4-row blocks: best when
// groups hold ~10-20 rows) or to a guess re-interpolated from the nearest row just read (im.probe_big == 1, 8-row blocks:
// best for groups of dozens to 255 rows).  The mode is one flag per image, so the branch is uniform across a wavefront.
// After BFT_PROBE_STEPS probes, binary search in what is left of the bracket.
#ifndef BFT_PROBE_STEPS
#define BFT_PROBE_STEPS 3
#endif
#ifndef BFT_PROBE_MAX_W
#define BFT_PROBE_MAX_W 2   // rows of one or two words (k <= 64); longer rows keep the galloping search
#endif

// One probe of NR rows of W words (W = 1: NR/2 loads of a pair of rows; W = 2: one load per row).  Returns true when the
// search is over (hit filled when found).  Else: dir < 0, bracket cut to rows below the block, edge = first bracket row
// of the block; dir > 0, bracket cut to rows above it, edge = the last one.
template <int W, int NR>
BFT_HD bool bft_probe_block(const BftImage& im, uint64_t guess, const uint64_t* t, uint64_t* lo2, uint64_t* hi2, uint64_t* edge, int* dir, BftHit& hit) {
    const uint64_t ga = guess & ~(uint64_t)(NR - 1);
    uint64_t w[NR][W];
    if (W == 1) {
#pragma unroll
        for (int j = 0; j < NR; j += 2) bft_load_pair(im, ga + j, &w[j][0], &w[j + 1][0]);
    } else {
#pragma unroll
        for (int j = 0; j < NR; j++) {
#if defined(__HIP_DEVICE_COMPILE__)
            bft_load_row<W>(im.tk + (ga + j) * W, w[j]);  // slack behind the table: see bft_load_pair
#else
            for (int x = 0; x < W; x++) w[j][x] = ga + j < im.n_kmers ? im.tk[(ga + j) * W + x] : ~0ull;
#endif
        }
    }
    uint32_t nin = 0, nlt = 0;  // block rows inside the bracket; those below t
    uint64_t found = ~0ull;
    int jf = 0, jl = 0;
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const uint64_t gi = ga + j;
        const bool in = gi >= *lo2 && gi < *hi2;
        const int c = bft_cmp<W>(w[j], t);
        if (in && !nin) jf = j;
        if (in) jl = j;
        nin += in;
        nlt += in && c < 0;
        if (in && c == 0) found = gi;
    }
    if (found != ~0ull) { hit.present = 1; hit.row = found; return true; }
    if (nlt != 0 && nlt != nin) return true;  // strictly between two rows of the block: absent
    const int je = nlt == 0 ? jf : jl;
#pragma unroll
    for (int j = 0; j < NR; j++)
        if (j == je) {
#pragma unroll
            for (int x = 0; x < W; x++) edge[x] = w[j][x];
        }
    if (nlt == 0) { *hi2 = ga > *lo2 ? ga : *lo2; *dir = -1; }
    else { *lo2 = ga + NR < *hi2 ? ga + NR : *hi2; *dir = 1; }
    return *lo2 >= *hi2;
}

// d = the level that owns the group (its rows differ only in the key bits below that level)
// PROBE: 0 / 1 fixes the mode at compile time (k_query: the 4-row code needs fewer registers than the 8-row one), -1 reads
// im.probe_big.  qbits: the rows [idx, idx+cnt) share the top 36 - qbits of the 36 key bits that follow level d (36: a whole suffix
// group; 34: one quarter of a root group, BFT_RQ_*), so positions are interpolated on the low qbits of them.
template <int W, int PROBE>
BFT_HD_RARE void bft_group_probe(const BftImage& im, uint64_t idx, uint32_t cnt, uint32_t g, const uint64_t* t, int d, BftHit& hit, int qbits) {
    uint64_t lo2 = idx, hi2 = idx + cnt, guess = idx + g, edge[W] = {};
    int dir = 0;
    const int mode = PROBE < 0 ? (int)im.probe_big : PROBE;
    constexpr int NS = 4, NB = W == 1 ? 8 : 4;  // one-word rows: 32-byte / 64-byte blocks; two-word rows: 64 bytes in both modes (+4 % over 32)
    // (128-byte blocks -- 16 one-word rows, one fabric request in tools/microbench/gather.hip -- were measured in round 4: 10.3 ms against
    // 8.1 with 64-byte blocks on the config-4 share, profiles/r04/probe_walk.jsonl: the walk pays for the registers and the bytes)
#pragma unroll
    for (int step = 0; step < BFT_PROBE_STEPS; step++) {
        bool over;
        if (mode == 1) over = bft_probe_block<W, NB>(im, guess, t, &lo2, &hi2, edge, &dir, hit);
        else over = bft_probe_block<W, NS>(im, guess, t, &lo2, &hi2, edge, &dir, hit);
        if (over) return;
        uint64_t dist = 0;
        if (mode != 0) {  // rows between the edge row and t, by the group's mean density
            const uint64_t a = bft_next36<W>(t, im.k, d), b = bft_next36<W>(edge, im.k, d);
            dist = ((a > b ? a - b : b - a) * cnt) >> qbits;
        }
        if (dir < 0) guess = hi2 - lo2 > dist + 1 ? hi2 - 1 - dist : lo2;
        else guess = hi2 - lo2 > dist ? lo2 + dist : hi2 - 1;
    }
    while (lo2 < hi2) {
        const uint64_t mid = (lo2 + hi2) >> 1;
        uint64_t r[W];
        bft_load_row<W>(im.tk + mid * W, r);
        const int c = bft_cmp<W>(r, t);
        if (c == 0) { hit.present = 1; hit.row = mid; return; }
        if (c < 0) lo2 = mid + 1; else hi2 = mid;
    }
}

// Search of the suffix group [idx, idx+cnt) of level d for t (src/presenceNode.c:1874-1915): interpolate on the next two
// prefixes, then block probes (groups of >= BFT_WINDOW_PROBE rows of one or two words) or a galloping search.
template <int W, int PROBE>
BFT_HD void bft_group_search(const BftImage& im, uint64_t idx, uint32_t cnt, const uint64_t* t, int d, BftHit& hit, int qbits = 36) {
    const uint64_t next = bft_next36<W>(t, im.k, d) & ((1ull << qbits) - 1ull);
    const uint32_t g = (uint32_t)((next * cnt) >> qbits);
#if defined(BFT_WINDOW_PROBE) && BFT_WINDOW_PROBE
    if (W <= BFT_PROBE_MAX_W && cnt >= BFT_WINDOW_PROBE) {
        bft_group_probe<W, PROBE>(im, idx, cnt, g, t, d, hit, qbits);
        return;
    }
#endif
    const uint64_t* rows = im.tk + idx * W;
    const int z = bft_rows_find<W>(rows, cnt, t, g);
    if (z >= 0) { hit.present = 1; hit.row = idx + (uint32_t)z; }
}

// start_node / d0: the walk normally starts at the root (level 0); bft_walk_last4 resumes it at the last level of a node
// it has already reached.
// ROOTMODE (root level with the derived tables): 0 = range table when the prefix is plain, else its rdir entry, inline;
// 1 = as 0, but a special prefix is NOT walked: the hit comes back with present == BFT_HIT_DEFERRED and the caller runs it
// again later with mode 2 (k_query queues those lanes and walks them together: a wavefront no longer pays the long
// container path for its few special lanes); 2 = rdir entry for every prefix (skips the range table).
#define BFT_HIT_DEFERRED 2
// WKH: the lookup of plain root groups in the k-mer hash (im.walk_kh) is compiled in at all -- the host restatement, and the kernels
// launched for "walk_hash" (k_query6h); everywhere else its registers would be dead weight (the 8-row probe kernels spilled 50-110 VGPRs
// with it).
#if defined(__HIP_DEVICE_COMPILE__)
#define BFT_WKH_DEFAULT false
#else
#define BFT_WKH_DEFAULT true
#endif
template <int W, class Root, int PROBE = -1, int ROOTMODE = 0, bool WKH = BFT_WKH_DEFAULT>
BFT_HD BftHit bft_walk(const BftImage& im, const Root& root, const BftNode& start_node, const uint64_t* t, const int d0 = 0) {
    BftHit hit;
    hit.present = 0;
    hit.row = 0;
    hit.cs = 0;
    hit.from_kh = 0;
    uint32_t node = 0;
    const int L = im.L, rb = 2 * (im.k - 9 * im.L);  // rb: bits of the k % 9 remainder (0 for reference-compatible k)
    for (int d = d0; d < L; d++) {
        BftNode nd;
        const uint32_t r = bft_digit<W>(t, im.k, d);
        uint64_t e = 0;
        bool stop = false, found = false;
        // Below the root: the node prefix hash (bft_image.h) answers "which entry does prefix r have in node `node`" with one
        // 64-byte bucket instead of node record -> CC header -> filter2 word -> cluster entry.
        if (d > d0 && im.nph != nullptr) {
            const int res = bft_nph_lookup(im, node, r, &e);
            if (res > 0) found = true;
            else if (res == 0 && im.nph_no_uc) return hit;  // in no CC of the node, and no node below the root holds UC rows: absent
        }
        if (!found) {
        if (d == d0) nd = start_node;
        else nd = im.nodes[node];
        // Which CC: the first one whose Bloom filter holds the key (src/presenceNode.c:1353-1362).  A node with ONE CC needs
        // no filter below the root: every prefix of a CC is Bloom-positive in it and every UC row is Bloom-negative
        // (SURVEY A.7/A.8), so "CC first, then the UC" gives what "Bloom, then CC or UC" gives, two gathers earlier.
#ifdef BFT_NO_SINGLE
        const bool single = false;
#else
        const bool single = d > 0 && nd.ncc == 1;
#endif
        int c = single ? 0 : -1;
        if (d == 0 && im.rdir != nullptr) {
            // Root level through the derived tables (bft_image.h).  Plain suffix groups: two adjacent words of the 1 MiB range
            // table give {first row, count}.
            if (WKH && ROOTMODE != 2 && im.walk_kh) {
                // A plain suffix group of the root in its hashed form: the k-mer hash (bft_image.h, BFT_KH_*) holds the group's suffixes with
                // their colour sets -- one line instead of the probes of the sorted rows (src/UC.c:81-124 finds the suffix by binary
                // search; same answer).  Special prefixes (child Node, UC rows: one bit per prefix in rspec) keep the containers.
                if (!((im.rspec[r >> 5] >> (r & 31u)) & 1u)) {
                    uint32_t cs = 0;
                    if (bft_kh_lookup<W, 0>(im, t, &cs)) { hit.present = 1; hit.cs = cs; hit.from_kh = 1; }
                    return hit;
                }
                if (ROOTMODE == 1) { hit.present = BFT_HIT_DEFERRED; return hit; }
            } else if (ROOTMODE != 2 && im.rstart != nullptr) {
                // rstart[r] and rstart[r + 1] in ONE 8-byte load (4-byte aligned: the hardware takes unaligned dwordx2 loads); a
                // gather costs per load instruction, and loading the second word only after testing the first would add a round trip
                struct __attribute__((packed, aligned(4))) Pair { uint32_t a, b; };
                const Pair pr = *reinterpret_cast<const Pair*>(im.rstart + r);
                const uint32_t a = pr.a;
                if (!(a & BFT_RSTART_SPECIAL)) {
                    const uint32_t b = pr.b & ~BFT_RSTART_SPECIAL;
                    if (BFT_DBG_STOP(im) >= 1 && BFT_DBG_STOP(im) <= 4) { hit.present = (int)(b & 1); return hit; }
                    if (b == a) return hit;
                    if (im.rq != nullptr) {
                        // the quarter of the group the k-mer's next two key bits select (BFT_RQ_*): exact bounds, interpolated on its own
                        const uint32_t q = im.rq[r], seg = (uint32_t)(bft_next36<W>(t, im.k, 0) >> 34);
                        const uint32_t s0 = seg ? BFT_RQ_OFF(q, seg) : 0u, s1 = seg < 3 ? BFT_RQ_OFF(q, seg + 1) : b - a;
                        if (s1 > s0) bft_group_search<W, PROBE>(im, a + s0, s1 - s0, t, 0, hit, 34);
                        return hit;
                    }
                    bft_group_search<W, PROBE>(im, a, b - a, t, 0, hit);
                    return hit;
                }
                if (ROOTMODE == 1) { hit.present = BFT_HIT_DEFERRED; return hit; }
            }
            // rdir[r] is what "first Bloom-positive CC, then filter2 / cluster / filter3 of that CC" (src/presenceNode.c:1353-1489)
            // yields for r, evaluated once per prefix when the image is bound: one gather instead of hash + Bloom + header +
            // bitmap word + entry.
            const uint64_t v = BFT_GATHER(&im.rdir[r]);
            if (BFT_DBG_STOP(im) >= 1 && BFT_DBG_STOP(im) <= 3) { hit.present = (int)(v & 1); return hit; }
            if (v == BFT_RDIR_ABSENT) return hit;        // a CC's Bloom filter holds the key but the CC not the prefix (:1546-1548)
            if (v == BFT_RDIR_NO_CC) {                    // no Bloom-positive CC: the node's UC (:1554-1573)
                bft_uc_search<W>(im, nd, t, hit);
                return hit;
            }
            e = v & ~BFT_RDIR_VALID;
            found = true;
            c = -1;
        } else if (nd.ncc && !single) {
            const uint32_t hm = root.hashmod(r >> 4);  // Bloom key = n2..n8 (src/presenceNode.c:1341-1343)
            if (d == 0) c = root.root_first_cc(nd, hm & 0xFFFFu, hm >> 16);
            else c = bft_first_cc_blk(im.bfT + (size_t)nd.bf_off * 8, nd.bf_wb, hm & 0xFFFFu, hm >> 16);
        }
        if (BFT_DBG_STOP(im) == 1) { hit.present = c >= 0; return hit; }
        if (c >= 0) {
            BftCCX cc;
            if (d == 0) cc = root.root_cc(nd, c);
            else {
                // one 16-byte load (the first half is laid out like BftCC, `flat` in place of pad0); the flat offsets only when needed
                const BftCCX* px = &im.ccx[nd.cc_first + c];
                const BftCC hd = *(const BftCC*)px;
                cc.f2_off = hd.f2_off; cc.clus_off = hd.clus_off; cc.child_off = hd.child_off; cc.nb_elem = hd.nb_elem; cc.s = hd.s; cc.flat = hd.pad0;
                cc.f18_off = 0; cc.fent_off = 0;
                if (cc.flat) { cc.f18_off = px->f18_off; cc.fent_off = px->fent_off; }
            }
            found = bft_cc_lookup(im, cc, r, &e, &stop);
            if (!found && !single) return hit;  // Bloom-positive CC without the prefix: absent (src/presenceNode.c:1546-1548)
        }
        if (!found) {
            bft_uc_search<W>(im, nd, t, hit);   // no Bloom-positive CC (or a single CC without the prefix): the node's UC
            return hit;
        }
        }  // (!found by the node prefix hash)
        if (stop) { hit.present = (int)(e & 1); return hit; }
        uint32_t cnt = (uint32_t)(e >> BFT_CHILD_CNT_SHIFT) & 0xFFu;
        uint64_t idx = e & BFT_CHILD_IDX_MASK;
        if (BFT_DBG_STOP(im) == 4) { hit.present = (int)(cnt & 1); return hit; }
        if (d == L - 1) {
            if (rb == 0) { hit.present = 1; hit.row = idx; return hit; }  // leaf: annotation row
            cnt = BFT_REM_COUNT(e);  // remainder group: count-1 on 16 bits
            idx = BFT_REM_ROW(e);
        } else if (cnt == 0) { node = (uint32_t)idx; continue; }          // child Node (src/presenceNode.c:1867)
        bft_group_search<W, PROBE>(im, idx, cnt, t, d, hit);  // suffix group of cnt rows
        return hit;
    }
    return hit;
}

// How many of the four rows t | v << vo (v = 0..3; t has those two bits clear) the suffix group [idx, idx+cnt) holds.  The
// four values lie within 13 consecutive integers, so their rows are neighbours in the sorted group: ONE search for the
// first row >= t (64-byte block probes from the interpolated guess, as in bft_group_probe, then a binary search), then a
// scan of the few rows up to t | 3 << vo.  Four separate searches cost four times the gathers even though they end in the
// same cache lines (a lane's line rarely survives in the L1 between two of its own probes).
template <int W>
BFT_HD int bft_group_count4(const BftImage& im, uint64_t idx, uint32_t cnt, const uint64_t* t, int d, int vo, bool need_all) {
    constexpr int NR = W == 1 ? 8 : 4;
    const uint64_t end = idx + cnt;
    uint64_t th[W];
#pragma unroll
    for (int w = 0; w < W; w++) th[w] = t[w];
    th[W - 1] |= 3ull << vo;
    uint64_t lo2 = idx, hi2 = end;  // rows before lo2 are < t; rows from hi2 on are >= t
    uint64_t guess = idx + (uint32_t)((bft_next36<W>(t, im.k, d) * cnt) >> 36);
    uint64_t p = ~0ull;             // position of the first row >= t, once known
#pragma unroll
    for (int step = 0; step < BFT_PROBE_STEPS && p == ~0ull; step++) {
        const uint64_t ga = guess & ~(uint64_t)(NR - 1);
        uint64_t w[NR][W];
        if (W == 1) {
#pragma unroll
            for (int j = 0; j < NR; j += 2) bft_load_pair(im, ga + j, &w[j][0], &w[j + 1][0]);
        } else {
#pragma unroll
            for (int j = 0; j < NR; j++) {
#if defined(__HIP_DEVICE_COMPILE__)
                bft_load_row<W>(im.tk + (ga + j) * W, w[j]);
#else
                for (int x = 0; x < W; x++) w[j][x] = ga + j < im.n_kmers ? im.tk[(ga + j) * W + x] : ~0ull;
#endif
            }
        }
        uint32_t nin = 0, nlt = 0;
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const uint64_t gi = ga + j;
            const bool in = gi >= lo2 && gi < hi2;
            nin += in;
            nlt += in && bft_cmp<W>(w[j], t) < 0;
        }
        const uint64_t first_in = ga > lo2 ? ga : lo2, end_in = ga + NR < hi2 ? ga + NR : hi2;
        if (nlt == nin) {              // every bracket row of the block is below t
            lo2 = end_in;
            if (lo2 >= hi2) p = lo2; else guess = lo2;
        } else if (nlt == 0 && first_in > lo2) {  // every one is >= t and rows remain on the left
            hi2 = first_in;
            guess = hi2 - 1;
        } else
            p = first_in + nlt;
    }
    if (p == ~0ull) {
        while (lo2 < hi2) {
            const uint64_t mid = (lo2 + hi2) >> 1;
            uint64_t r[W];
            bft_load_row<W>(im.tk + mid * W, r);
            if (bft_cmp<W>(r, t) < 0) lo2 = mid + 1; else hi2 = mid;
        }
        p = lo2;
    }
    int count = 0;
    for (uint64_t j = p; j < end; j++) {
        uint64_t r[W];
        bft_load_row<W>(im.tk + j * W, r);
        if (bft_cmp<W>(r, th) > 0) break;
        bool same = ((r[W - 1] ^ t[W - 1]) & ~(3ull << vo)) == 0;
#pragma unroll
        for (int w = 0; w + 1 < W; w++) same = same && r[w] == t[w];
        count += same;
        if (!need_all && count >= 2) break;
    }
    return count;
}

// The four k-mers t | v << vo (v = 0..3) that differ only in their LAST nucleotide (vo = 2: n9 of the last prefix; vo = 0:
// the end of the k % 9 remainder) -- the successors of a k-mer (src/branchingNode.c:16-112; src/presenceNode.c:15-1211
// shares the descent the same way).  They take the same path through every level that does not hold the varying bits: one
// descent with the routing of bft_walk (kept as its own copy so that the presence kernel's code is untouched), then four
// finishes in the container the path ends in -- the same suffix group, the same node UC, or the same last-level node.
// Returns how many of the four are present (stops at two unless need_all).
template <int W, class Root>
BFT_HD int bft_walk_last4(const BftImage& im, const Root& root, const BftNode& root_node, const uint64_t* t, bool need_all) {
    const int L = im.L, rb = 2 * (im.k - 9 * im.L);
    const int vo = rb ? 0 : 2;
    int count = 0;
    uint32_t node = 0;
    uint64_t tt[W];
#pragma unroll
    for (int w = 0; w < W; w++) tt[w] = t[w];
    for (int d = 0; d < L; d++) {
        BftNode nd;
        if (d == 0) nd = root_node;
        else nd = im.nodes[node];
        if (rb == 0 && d == L - 1) {  // the last prefix itself varies: four prefixes of this node
            for (uint64_t v = 0; v < 4 && (need_all || count < 2); v++) {
                tt[W - 1] = t[W - 1] | (v << vo);
                count += bft_walk<W, Root, 0>(im, root, nd, tt, d).present;
            }
            return count;
        }
        const uint32_t r = bft_digit<W>(t, im.k, d);
#ifdef BFT_NO_SINGLE
        const bool single = false;
#else
        const bool single = d > 0 && nd.ncc == 1;
#endif
        int c = single ? 0 : -1;
        uint64_t e = 0;
        bool stop = false, found = false;
        if (d == 0 && im.rdir != nullptr) {  // root level through the derived direct table, as in bft_walk
            const uint64_t v = BFT_GATHER(&im.rdir[r]);
            if (v == BFT_RDIR_ABSENT) return 0;
            c = -1;
            if (v != BFT_RDIR_NO_CC) { e = v & ~BFT_RDIR_VALID; found = true; }  // else: the UC branch below
        } else if (nd.ncc && !single) {
            const uint32_t hm = root.hashmod(r >> 4);
            if (d == 0) c = root.root_first_cc(nd, hm & 0xFFFFu, hm >> 16);
            else c = bft_first_cc_blk(im.bfT + (size_t)nd.bf_off * 8, nd.bf_wb, hm & 0xFFFFu, hm >> 16);
        }
        if (c >= 0) {
            BftCCX cc;
            if (d == 0) cc = root.root_cc(nd, c);
            else {
                const BftCCX* px = &im.ccx[nd.cc_first + c];
                const BftCC hd = *(const BftCC*)px;
                cc.f2_off = hd.f2_off; cc.clus_off = hd.clus_off; cc.child_off = hd.child_off; cc.nb_elem = hd.nb_elem; cc.s = hd.s; cc.flat = hd.pad0;
                cc.f18_off = 0; cc.fent_off = 0;
                if (cc.flat) { cc.f18_off = px->f18_off; cc.fent_off = px->fent_off; }
            }
            found = bft_cc_lookup(im, cc, r, &e, &stop);
            if (!found && !single) return 0;  // Bloom-positive CC without the prefix: all four absent
        }
        if (!found) {  // the node's UC holds whole k-mers: four searches
            for (uint64_t v = 0; v < 4 && (need_all || count < 2); v++) {
                BftHit hit;
                hit.present = 0;
                hit.row = 0;
                tt[W - 1] = t[W - 1] | (v << vo);
                bft_uc_search<W>(im, nd, tt, hit);
                count += hit.present;
            }
            return count;
        }
        uint32_t cnt = (uint32_t)(e >> BFT_CHILD_CNT_SHIFT) & 0xFFu;
        uint64_t idx = e & BFT_CHILD_IDX_MASK;
        if (d == L - 1) {  // rb != 0 here: the remainder group of this prefix holds all four
            cnt = BFT_REM_COUNT(e);
            idx = BFT_REM_ROW(e);
        } else if (cnt == 0) { node = (uint32_t)idx; continue; }
        if (W <= BFT_PROBE_MAX_W) return bft_group_count4<W>(im, idx, cnt, t, d, vo, need_all);
        for (uint64_t v = 0; v < 4 && (need_all || count < 2); v++) {
            BftHit hit;
            hit.present = 0;
            hit.row = 0;
            tt[W - 1] = t[W - 1] | (v << vo);
            bft_group_search<W, 0>(im, idx, cnt, tt, d, hit);
            count += hit.present;
        }
        return count;
    }
    return count;
}
