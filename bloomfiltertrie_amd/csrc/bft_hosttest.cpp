// bft_hosttest.cpp -- TEST-ONLY helper library (libbft_hosttest.so), never loaded by the package.
//
// The development container has no GPU, so the host-side logic that feeds the kernels is unit
// tested on the CPU through this file: T-form conversion, container assembly (bft_index.cpp) and
// the shared per-query walk of bft_walk.h (the same source the HIP kernel compiles for the device).
// Nothing here is reachable from the bft_gpu_* C-ABI, which has no CPU path.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "bft_hash.h"
#include "bft_index.h"
#include "bft_walk.h"
#include "bft_kh_host.h"

struct HostTrie {
    int k, L, W, B;
    std::vector<uint32_t> hashmod;
    std::vector<uint64_t> tk;
    std::vector<uint32_t> tcol, cs_off, cs_ids;
    std::vector<uint64_t> rdir, nph;
    BftKhHostTable kh;
    std::vector<uint32_t> rstart, rq, rspec;
    uint64_t rstart_plain = 0;
    BftHostIndex idx;
    BftImage im;
};

template <int W>
static void to_tform(const uint8_t* packed, uint64_t n, int B, int L, std::vector<uint64_t>& out) {
    out.resize(n * W);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t x[W] = {0}, t[W];
        for (int b = 0; b < B; b++) x[b >> 3] |= (uint64_t)packed[i * B + b] << (8 * (b & 7));
        bft_tform_from_x<W>(x, L, t);  // L holds k here
        for (int w = 0; w < W; w++) out[i * W + w] = t[w];
    }
}

static void tform_any(const uint8_t* packed, uint64_t n, int k, std::vector<uint64_t>& out) {
    int W = bft_words_for_k(k), B = bft_bytes_for_k(k);
    switch (W) {
    case 1: to_tform<1>(packed, n, B, k, out); break;
    case 2: to_tform<2>(packed, n, B, k, out); break;
    case 3: to_tform<3>(packed, n, B, k, out); break;
    default: to_tform<4>(packed, n, B, k, out); break;
    }
}

extern "C" void* bft_hosttest_build(const uint8_t* kmers, uint64_t n, int k, int r1, int r2) {
    if (!bft_valid_k(k)) return nullptr;
    HostTrie* t = new HostTrie();
    t->k = k; t->L = k / 9; t->W = bft_words_for_k(k); t->B = bft_bytes_for_k(k);
    t->hashmod.resize(16384);
    bft_make_hashmod(r1 > 0 ? r1 : BFT_DEFAULT_R1, r2 > 0 ? r2 : BFT_DEFAULT_R2, t->hashmod.data());
    std::vector<uint64_t> all;
    tform_any(kmers, n, k, all);
    const int W = t->W;
    std::vector<uint64_t> order(n);
    for (uint64_t i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
        for (int w = 0; w < W; w++) if (all[a * W + w] != all[b * W + w]) return all[a * W + w] < all[b * W + w];
        return false;
    });
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t* cur = &all[order[i] * W];
        if (!t->tk.empty() && memcmp(&t->tk[t->tk.size() - W], cur, W * 8) == 0) continue;
        t->tk.insert(t->tk.end(), cur, cur + W);
    }
    uint64_t nk = t->tk.size() / W;
    t->tcol.assign(nk, 0);
    t->cs_off = {0, 1};
    t->cs_ids = {0};
    if (!bft_build_index(t->tk.data(), nk, k, t->hashmod.data(), t->idx)) { delete t; return nullptr; }
    bft_flatten_index(t->idx, BFT_TRESH_SUF_PREF);
    BftImage& im = t->im;
    memset(&im, 0, sizeof(im));
    im.k = k; im.L = t->L; im.W = W; im.nb_genomes = 1; im.n_kmers = nk;
    im.hashmod = t->hashmod.data();
    im.nodes = t->idx.nodes.data(); im.bfT = t->idx.bfT.data(); im.ccs = t->idx.ccs.data();
    im.f2w = t->idx.f2w.data(); im.clus = t->idx.clus.data(); im.child = t->idx.child.data();
    im.ccx = t->idx.ccx.data(); im.f18 = t->idx.f18.data(); im.fent = t->idx.fent.data();
    im.tk = t->tk.data(); im.tcol = t->tcol.data(); im.uck = t->idx.uck.data(); im.ucrow = t->idx.ucrow.data();
    im.cs_off = t->cs_off.data(); im.cs_ids = t->cs_ids.data();
    return t;
}

template <int W>
static uint64_t query(HostTrie* t, const uint8_t* q, uint64_t nq, uint8_t* bits, uint32_t* rows) {
    std::vector<uint64_t> tq;
    to_tform<W>(q, nq, t->B, t->k, tq);
    uint64_t cnt = 0;
    memset(bits, 0, (nq + 7) / 8);
    for (uint64_t i = 0; i < nq; i++) {
        const BftRootGlobal root(t->im);
        const BftHit h = bft_walk<W, BftRootGlobal, -1>(t->im, root, t->im.nodes[0], &tq[i * W]);
        if (h.present) { bits[i >> 3] |= (uint8_t)(1u << (i & 7)); cnt++; }
        if (rows) rows[i] = h.present ? (h.from_kh ? h.cs : (uint32_t)h.row) : 0xFFFFFFFFu;  // (the host table's values are the rows)
    }
    return cnt;
}

// k-mer hash (BFT_KH_*): the sequential restatement of the canonical build (bft_kh_host.h) with the value = the row of the k-mer (the GPU
// stores the colour set there), then the same lookup the kernels run.  load_pct: occupancy of the home lines in per cent (the product's
// "kmer_hash_load"); 0 = drop the table.  Returns the number of home lines.
extern "C" uint64_t bft_hosttest_kmer_hash(void* hv, uint32_t load_pct) {
    HostTrie* t = (HostTrie*)hv;
    t->im.kh_lines = nullptr;
    t->im.kh_ovf = nullptr;
    t->im.kh_ovf_val = nullptr;
    t->im.kh_ovf_n = 0;
    t->im.walk_kh = 0;
    t->kh = BftKhHostTable();
    const uint64_t n = t->tk.size() / t->W;
    if (!load_pct || n == 0) return 0;
    std::vector<uint32_t> rows(n);
    for (uint64_t i = 0; i < n; i++) rows[i] = (uint32_t)i;
    switch (t->W) {
    case 1: bft_kh_build_host<1>(t->tk.data(), rows.data(), n, t->k, n, load_pct, t->kh); break;
    case 2: bft_kh_build_host<2>(t->tk.data(), rows.data(), n, t->k, n, load_pct, t->kh); break;
    case 3: bft_kh_build_host<3>(t->tk.data(), rows.data(), n, t->k, n, load_pct, t->kh); break;
    default: bft_kh_build_host<4>(t->tk.data(), rows.data(), n, t->k, n, load_pct, t->kh); break;
    }
    if (!t->kh.ok) return 0;
    t->im.kh_lines = t->kh.lines.data();
    t->im.kh = t->kh.g;
    t->im.kh_ovf = t->kh.ovf_k.data();
    t->im.kh_ovf_val = t->kh.ovf_v.data();
    t->im.kh_ovf_n = (uint32_t)t->kh.ovf_v.size();
    return t->kh.g.nl;
}
// the walk looks plain root groups up in the table (BftImage::walk_kh; needs the table and the root range table): returns 1 when on
extern "C" int bft_hosttest_walk_kh(void* hv, int on) {
    HostTrie* t = (HostTrie*)hv;
    t->im.walk_kh = 0;
    t->im.rspec = nullptr;
    if (!on || !t->im.kh_lines || !t->im.rstart) return 0;
    t->rspec.assign((1u << 18) / 32, 0u);
    for (uint32_t r = 0; r < (1u << 18); r++)
        if (t->rstart[r] & BFT_RSTART_SPECIAL) t->rspec[r >> 5] |= 1u << (r & 31u);
    t->im.rspec = t->rspec.data();
    t->im.walk_kh = 1;
    return 1;
}
static void geo_out(const BftKhHostTable& tab, uint32_t* out) {
    const BftKhGeo& g = tab.g;
    out[0] = g.S; out[1] = g.f; out[2] = g.wb; out[3] = g.cb; out[4] = g.kb; out[5] = g.qb; out[6] = g.hb; out[7] = g.restb; out[8] = g.t; out[9] = g.m;
    out[10] = (uint32_t)g.nl; out[11] = tab.max_d; out[12] = g.db; out[13] = (uint32_t)tab.ovf_v.size();
}
// The sequential restatement on arrays handed in (tests compare the GPU-built table with it): tk = n sorted T-form rows, vals < n_values.
// Returns the words of the table (8 per line); geo[0..11] as bft_hosttest_kh_geometry.  0 when the table cannot be built.
extern "C" uint64_t bft_hosttest_kh_build(const uint64_t* tk, const uint32_t* vals, uint64_t n, int k, uint64_t n_values, uint32_t load_pct, uint64_t* lines_out,
                                          uint64_t cap_words, uint32_t* geo, uint64_t* ovf_k_out, uint32_t* ovf_v_out) {
    BftKhHostTable tab;
    switch (bft_words_for_k(k)) {
    case 1: bft_kh_build_host<1>(tk, vals, n, k, n_values, load_pct, tab); break;
    case 2: bft_kh_build_host<2>(tk, vals, n, k, n_values, load_pct, tab); break;
    case 3: bft_kh_build_host<3>(tk, vals, n, k, n_values, load_pct, tab); break;
    default: bft_kh_build_host<4>(tk, vals, n, k, n_values, load_pct, tab); break;
    }
    if (geo) geo_out(tab, geo);
    if (!tab.ok || tab.lines.size() > cap_words) return 0;
    memcpy(lines_out, tab.lines.data(), tab.lines.size() * 8);
    if (ovf_k_out && !tab.ovf_k.empty()) memcpy(ovf_k_out, tab.ovf_k.data(), tab.ovf_k.size() * 8);  // (room for BFT_KH_OVF_CAP rows)
    if (ovf_v_out && !tab.ovf_v.empty()) memcpy(ovf_v_out, tab.ovf_v.data(), tab.ovf_v.size() * 4);
    return tab.lines.size();
}
// A table built elsewhere (the GPU's fast build: slot order by arrival) against the sorted table it was built from: every used slot decodes
// to a row of tk with that row's value, every row is there exactly once, and every line between a k-mer's home line and its own is full
// (what the lookup's early exit relies on).  Returns 1 when all of that holds, 0 or a negative code otherwise.
template <int W>
static int kh_verify(const uint64_t* tk, const uint32_t* vals, uint64_t n, int k, const BftKhGeo& g, const uint64_t* lines, uint64_t n_lines, const uint64_t* ovf_k,
                     const uint32_t* ovf_v, uint32_t n_ovf) {
    BftImage im;
    memset(&im, 0, sizeof(im));
    im.k = k; im.W = W; im.L = k / 9;
    im.kh = g;
    im.kh_lines = lines;
    if (n_lines != g.nl + BFT_KH_TAIL_LINES) return -1;
    std::vector<uint8_t> seen(n, 0);
    uint64_t cnt = 0;
    for (uint64_t ln = 0; ln < n_lines; ln++) {
        const uint64_t* line = lines + ln * BFT_KH_LINE_WORDS;
        const uint32_t occ = (uint32_t)(line[1] >> (64u - g.S));
        for (uint32_t s = 0; s < g.S; s++) {
            if (!((occ >> s) & 1u)) continue;
            uint64_t key[W];
            uint32_t v;
            bft_kh_slot_decode<W>(im, line, line, ln, s, key, &v);
            if (v == 0xFFFFFFFFu) continue;  // (a tombstone)
            // the row: lower bound of key in tk
            uint64_t lo = 0, hi = n;
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (bft_cmp<W>(tk + mid * W, key) < 0) lo = mid + 1; else hi = mid;
            }
            if (lo >= n || bft_cmp<W>(tk + lo * W, key) != 0) return -2;
            if (seen[lo] || vals[lo] != v) return -3;
            seen[lo] = 1;
            cnt++;
            BftKhKey<W> kk;
            bft_kh_key<W>(key, k, g, kk);
            if (kk.home > ln || ln - kk.home > g.maxd) return -4;
            for (uint64_t x = kk.home; x < ln; x++)
                if ((uint32_t)(lines[x * BFT_KH_LINE_WORDS + 1] >> (64u - g.S)) != (1u << g.S) - 1u) return -5;
        }
    }
    // the overflow list: sorted rows of tk, not in the table, and every line from home to home + maxd is full
    for (uint32_t e = 0; e < n_ovf; e++) {
        const uint64_t* key = ovf_k + (size_t)e * W;
        if (e && bft_cmp<W>(ovf_k + (size_t)(e - 1) * W, key) >= 0) return -7;
        uint64_t lo = 0, hi = n;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (bft_cmp<W>(tk + mid * W, key) < 0) lo = mid + 1; else hi = mid;
        }
        if (lo >= n || bft_cmp<W>(tk + lo * W, key) != 0 || seen[lo] || vals[lo] != ovf_v[e]) return -8;
        seen[lo] = 1;
        cnt++;
        BftKhKey<W> kk;
        bft_kh_key<W>(key, k, g, kk);
        for (uint64_t x = kk.home; x <= kk.home + g.maxd; x++)
            if ((uint32_t)(lines[x * BFT_KH_LINE_WORDS + 1] >> (64u - g.S)) != (1u << g.S) - 1u) return -9;
    }
    return cnt == n ? 1 : -6;
}
extern "C" int bft_hosttest_kh_verify(const uint64_t* tk, const uint32_t* vals, uint64_t n, int k, uint64_t n_values, uint32_t load_pct, uint32_t maxd,
                                      const uint64_t* lines, uint64_t n_lines, const uint64_t* ovf_k, const uint32_t* ovf_v, uint32_t n_ovf) {
    BftKhGeo g = bft_kh_geometry(k, n, n_values, load_pct);
    g.maxd = maxd;
    switch (bft_words_for_k(k)) {
    case 1: return kh_verify<1>(tk, vals, n, k, g, lines, n_lines, ovf_k, ovf_v, n_ovf);
    case 2: return kh_verify<2>(tk, vals, n, k, g, lines, n_lines, ovf_k, ovf_v, n_ovf);
    case 3: return kh_verify<3>(tk, vals, n, k, g, lines, n_lines, ovf_k, ovf_v, n_ovf);
    default: return kh_verify<4>(tk, vals, n, k, g, lines, n_lines, ovf_k, ovf_v, n_ovf);
    }
}
// geometry of the host table: out[0..13] (12: displacement bits, 13: k-mers in the overflow list) = slots per line, field bits, body bytes, value bits, key bits, q bits, hashed bits, bits below,
// t, m, home lines, largest displacement
extern "C" void bft_hosttest_kh_geometry(void* hv, uint32_t* out) { geo_out(((HostTrie*)hv)->kh, out); }
template <int W>
static uint64_t query_kh(HostTrie* t, const uint8_t* q, uint64_t nq, uint8_t* bits, uint32_t* vals) {
    std::vector<uint64_t> tq;
    to_tform<W>(q, nq, t->B, t->k, tq);
    uint64_t cnt = 0;
    memset(bits, 0, (nq + 7) / 8);
    for (uint64_t i = 0; i < nq; i++) {
        uint32_t v = 0xFFFFFFFFu;
        const bool hit = bft_kh_lookup<W, 0>(t->im, &tq[i * W], &v);
        if (hit) { bits[i >> 3] |= (uint8_t)(1u << (i & 7)); cnt++; }
        if (vals) vals[i] = hit ? v : 0xFFFFFFFFu;
    }
    return cnt;
}
extern "C" int64_t bft_hosttest_query_kh(void* hv, const uint8_t* q, uint64_t nq, uint8_t* bits, uint32_t* vals) {
    HostTrie* t = (HostTrie*)hv;
    if (!t->im.kh_lines) return -1;
    switch (t->W) {
    case 1: return (int64_t)query_kh<1>(t, q, nq, bits, vals);
    case 2: return (int64_t)query_kh<2>(t, q, nq, bits, vals);
    case 3: return (int64_t)query_kh<3>(t, q, nq, bits, vals);
    default: return (int64_t)query_kh<4>(t, q, nq, bits, vals);
    }
}
// home line of every packed k-mer under the host table's geometry (what a lookup computes first): successors / predecessors of a k-mer share it
template <int W>
static void kh_homes(HostTrie* t, const uint8_t* q, uint64_t nq, uint64_t* out) {
    std::vector<uint64_t> tq;
    to_tform<W>(q, nq, t->B, t->k, tq);
    for (uint64_t i = 0; i < nq; i++) {
        BftKhKey<W> kk;
        bft_kh_key<W>(&tq[i * W], t->k, t->im.kh, kk);
        out[i] = kk.home;
    }
}
extern "C" int bft_hosttest_kh_homes(void* hv, const uint8_t* q, uint64_t nq, uint64_t* out) {
    HostTrie* t = (HostTrie*)hv;
    if (!t->im.kh_lines) return -1;
    switch (t->W) {
    case 1: kh_homes<1>(t, q, nq, out); break;
    case 2: kh_homes<2>(t, q, nq, out); break;
    case 3: kh_homes<3>(t, q, nq, out); break;
    default: kh_homes<4>(t, q, nq, out); break;
    }
    return 0;
}
// every slot of the host table decoded back (bft_kh_slot_decode, what "compact_table" rebuilds the sorted table from): 1 when the decoded
// (k-mer, value) pairs are exactly the rows of the sorted table with their row numbers
template <int W>
static int kh_roundtrip(HostTrie* t) {
    const uint64_t n = t->tk.size() / W, nl = t->kh.g.nl + BFT_KH_TAIL_LINES;
    std::vector<uint8_t> seen(n, 0);
    uint64_t cnt = 0;
    for (uint64_t ln = 0; ln < nl; ln++) {
        const uint64_t* line = t->im.kh_lines + ln * BFT_KH_LINE_WORDS;
        const uint32_t occ = (uint32_t)(line[1] >> (64u - t->kh.g.S));
        for (uint32_t s = 0; s < t->kh.g.S; s++) {
            if (!((occ >> s) & 1u)) continue;
            uint64_t key[W];
            uint32_t v;
            bft_kh_slot_decode<W>(t->im, line, line, ln, s, key, &v);
            if (v == 0xFFFFFFFFu) continue;  // (a tombstone: in use, value 0)
            if (v >= n || seen[v] || memcmp(key, &t->tk[(uint64_t)v * W], W * 8) != 0) return 0;
            seen[v] = 1;
            cnt++;
        }
    }
    for (uint32_t e = 0; e < t->im.kh_ovf_n; e++) {
        const uint32_t v = t->im.kh_ovf_val[e];
        if (v >= n || seen[v] || memcmp(t->im.kh_ovf + (size_t)e * W, &t->tk[(uint64_t)v * W], W * 8) != 0) return 0;
        seen[v] = 1;
        cnt++;
    }
    return cnt == n;
}
extern "C" int bft_hosttest_kh_roundtrip(void* hv) {
    HostTrie* t = (HostTrie*)hv;
    if (!t->im.kh_lines) return -1;
    switch (t->W) {
    case 1: return kh_roundtrip<1>(t);
    case 2: return kh_roundtrip<2>(t);
    case 3: return kh_roundtrip<3>(t);
    default: return kh_roundtrip<4>(t);
    }
}
// mean lines read per lookup of the stored k-mers themselves, and the longest run (diagnostics of the table's occupancy)
template <int W>
static double kh_probe_stats(HostTrie* t, uint64_t* longest) {
    const uint64_t n = t->tk.size() / W;
    uint64_t total = 0, worst = 0;
    for (uint64_t i = 0; i < n; i++) {
        BftKhKey<W> kk;
        bft_kh_key<W>(&t->tk[i * W], t->k, t->im.kh, kk);
        uint64_t steps = 1;
        for (uint32_t d = 0;; d++, steps++) {
            uint32_t v;
            const uint64_t* line = t->im.kh_lines + (kk.home + d) * BFT_KH_LINE_WORDS;
            if (bft_kh_scan<W, 0>(t->im, line, line, kk, d, &v) > 0) break;
            if (d >= t->im.kh.maxd) {  // the overflow list, or a stored k-mer the lookup cannot find (then the caller's bound fails)
                if (!bft_kh_overflow_find<W>(t->im, &t->tk[i * W], &v)) steps = ~0ull >> 1;
                break;
            }
        }
        total += steps;
        worst = std::max(worst, steps);
    }
    if (longest) *longest = worst;
    return n ? (double)total / (double)n : 0.0;
}
extern "C" double bft_hosttest_kh_probe_stats(void* hv, uint64_t* longest) {
    HostTrie* t = (HostTrie*)hv;
    if (!t->im.kh_lines) return 0.0;
    switch (t->W) {
    case 1: return kh_probe_stats<1>(t, longest);
    case 2: return kh_probe_stats<2>(t, longest);
    case 3: return kh_probe_stats<3>(t, longest);
    default: return kh_probe_stats<4>(t, longest);
    }
}

// node prefix hash (BFT_NPH_*): the enumeration of k_nph_fill, sequential; on / off.  tiny != 0 sizes the table far too small so
// that most buckets fill up and the "full bucket -> container path" branch is exercised.
extern "C" void bft_hosttest_node_hash(void* hv, int on, int tiny) {
    HostTrie* t = (HostTrie*)hv;
    BftImage& im = t->im;
    im.nph = nullptr; im.nph_mask = 0; im.nph_no_uc = 0;
    if (!on || t->idx.nodes.size() <= 1) return;
    uint64_t nbk = tiny ? 8 : 1024;
    while (!tiny && nbk < t->idx.n_prefixes) nbk <<= 1;
    t->nph.assign(nbk * BFT_NPH_SLOTS * 2, BFT_NPH_EMPTY);
    bool any_uc = false;
    for (uint32_t m = 1; m < t->idx.nodes.size(); m++) {
        const BftNode nd = t->idx.nodes[m];
        any_uc = any_uc || nd.uc_n;
        for (uint32_t c = 0; c < nd.ncc; c++) {
            const BftCC cc = t->idx.ccs[nd.cc_first + c];
            const uint32_t nw = ((1u << (18 - cc.s)) + BFT_F2_BITS_PER_WORD - 1) / BFT_F2_BITS_PER_WORD;
            uint32_t clu = 0;
            for (uint32_t w = 0; w < nw; w++) {
                uint64_t bits = t->idx.f2w[cc.f2_off + w] & ((1ull << BFT_F2_BITS_PER_WORD) - 1ull);
                while (bits) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(bits);
                    bits &= bits - 1ull;
                    const uint32_t pu = w * BFT_F2_BITS_PER_WORD + b;
                    const uint64_t ce = t->idx.clus[cc.clus_off + clu++];
                    const uint32_t len = (ce & BFT_CLUS_MULTI) ? (uint32_t)((ce >> BFT_CLUS_LEN_SHIFT) & 0xFFFFu) : 1u;
                    for (uint32_t j = 0; j < len; j++) {
                        const uint64_t ent = (ce & BFT_CLUS_MULTI) ? t->idx.child[cc.child_off + (uint32_t)ce + j] : ce;
                        const uint32_t r = (pu << cc.s) | ((uint32_t)(ent >> BFT_CHILD_PV_SHIFT) & 0xFFu);
                        bft_nph_insert_seq(t->nph.data(), nbk - 1, m, r, ent);
                    }
                }
            }
        }
    }
    im.nph = t->nph.data();
    im.nph_mask = nbk - 1;
    im.nph_no_uc = any_uc ? 0u : 1u;
}

// root direct table (BFT_RDIR_*): derived with the same per-prefix function as the GPU kernel k_root_direct; on / off
extern "C" void bft_hosttest_root_direct(void* hv, int on) {
    HostTrie* t = (HostTrie*)hv;
    t->im.rdir = nullptr;
    t->im.rstart = nullptr;
    t->im.rq = nullptr;
    if (!on || t->idx.nodes.empty() || !t->idx.nodes[0].ncc) return;
    t->rdir.assign(1u << 18, 0);
    const BftRootGlobal root(t->im);
    const BftNode nd = t->im.nodes[0];
    for (uint32_t r = 0; r < (1u << 18); r++) t->rdir[r] = bft_root_direct_entry(t->im, root, nd, r);
    t->im.rdir = t->rdir.data();
    if (on < 2) return;  // 2: also the root range table (BFT_RSTART_*), the two passes of k_root_ranges / k_root_ranges_check
    t->rstart.assign((1u << 18) + 2, 0);
    for (uint32_t r = 0; r <= (1u << 18); r++) {
        const uint64_t v = r < (1u << 18) ? t->rdir[r] : 0ull;
        switch (t->W) {
        case 1: t->rstart[r] = bft_root_range_entry<1>(t->im, r, v, nd.uc_n); break;
        case 2: t->rstart[r] = bft_root_range_entry<2>(t->im, r, v, nd.uc_n); break;
        case 3: t->rstart[r] = bft_root_range_entry<3>(t->im, r, v, nd.uc_n); break;
        default: t->rstart[r] = bft_root_range_entry<4>(t->im, r, v, nd.uc_n); break;
        }
    }
    for (uint32_t r = 0; r < (1u << 18); r++)
        if (!(t->rstart[r] & BFT_RSTART_SPECIAL) && !bft_root_range_ok(t->rstart[r], t->rstart[r + 1], t->rdir[r])) t->rstart[r] |= BFT_RSTART_SPECIAL;
    t->im.rstart = t->rstart.data();
    uint64_t plain = 0;
    for (uint32_t r = 0; r < (1u << 18); r++) plain += !(t->rstart[r] & BFT_RSTART_SPECIAL);
    t->rstart_plain = plain;
    if (on < 3 || t->im.L < 2) return;  // 3: also the root quartile table (BFT_RQ_*), k_root_quartiles
    t->rq.assign(1u << 18, 0);
    for (uint32_t r = 0; r < (1u << 18); r++) {
        switch (t->W) {
        case 1: t->rq[r] = bft_root_quartile_entry<1>(t->im, t->rstart[r], t->rstart[r + 1]); break;
        case 2: t->rq[r] = bft_root_quartile_entry<2>(t->im, t->rstart[r], t->rstart[r + 1]); break;
        case 3: t->rq[r] = bft_root_quartile_entry<3>(t->im, t->rstart[r], t->rstart[r + 1]); break;
        default: t->rq[r] = bft_root_quartile_entry<4>(t->im, t->rstart[r], t->rstart[r + 1]); break;
        }
    }
    t->im.rq = t->rq.data();
}
extern "C" uint64_t bft_hosttest_root_plain(void* hv) { return ((HostTrie*)hv)->rstart_plain; }

// suffix-group probe mode of the walk (BftImage::probe_big): same answers either way
extern "C" void bft_hosttest_set_probe(void* hv, int big) {  // 0: 4-row blocks, 1: 8-row blocks
    ((HostTrie*)hv)->im.probe_big = big == 1 ? 1u : 0u;
}

extern "C" uint64_t bft_hosttest_query(void* hv, const uint8_t* q, uint64_t nq, uint8_t* bits, uint32_t* rows) {
    HostTrie* t = (HostTrie*)hv;
    switch (t->W) {
    case 1: return query<1>(t, q, nq, bits, rows);
    case 2: return query<2>(t, q, nq, bits, rows);
    case 3: return query<3>(t, q, nq, bits, rows);
    default: return query<4>(t, q, nq, bits, rows);
    }
}

// out[0]=k-mers [1]=nodes [2]=CCs [3]=UC rows [4]=child nodes [5]=prefixes [6]=CCs s=4 [7]=max CCs/node [8]=root CCs [9]=root UC rows
extern "C" void bft_hosttest_stats(void* hv, uint64_t* out) {
    HostTrie* t = (HostTrie*)hv;
    out[0] = t->tk.size() / t->W; out[1] = t->idx.nodes.size(); out[2] = t->idx.ccs.size(); out[3] = t->idx.ucrow.size();
    out[4] = t->idx.n_child_nodes; out[5] = t->idx.n_prefixes; out[6] = t->idx.n_ccs_s4; out[7] = t->idx.max_ccs_per_node;
    out[8] = t->idx.nodes[0].ncc; out[9] = t->idx.nodes[0].uc_n;
    out[10] = 0; out[11] = 0;
}

// T-form round trip: packed -> T -> packed
extern "C" void bft_hosttest_roundtrip(const uint8_t* kmers, uint64_t n, int k, uint8_t* out, uint64_t* tform_out) {
    std::vector<uint64_t> all;
    tform_any(kmers, n, k, all);
    int W = bft_words_for_k(k), B = bft_bytes_for_k(k), L = k;  // the helpers take k
    for (uint64_t i = 0; i < n; i++) {
        uint64_t x[BFT_MAX_W] = {0};
        switch (W) {
        case 1: bft_x_from_tform<1>(&all[i * W], L, x); break;
        case 2: bft_x_from_tform<2>(&all[i * W], L, x); break;
        case 3: bft_x_from_tform<3>(&all[i * W], L, x); break;
        default: bft_x_from_tform<4>(&all[i * W], L, x); break;
        }
        for (int b = 0; b < B; b++) out[i * B + b] = (uint8_t)(x[b >> 3] >> (8 * (b & 7)));
        if (tform_out) for (int w = 0; w < W; w++) tform_out[i * W + w] = all[i * W + w];
    }
}

// re-derive the flat form with another threshold (mirrors bft_gpu_set_option "flat_min")
extern "C" void bft_hosttest_flatten(void* hv, uint32_t flat_min) {
    HostTrie* t = (HostTrie*)hv;
    bft_flatten_index(t->idx, flat_min);
    t->im.ccx = t->idx.ccx.data(); t->im.f18 = t->idx.f18.data(); t->im.fent = t->idx.fent.data();
}

extern "C" void bft_hosttest_hashmod(int r1, int r2, uint32_t* out) { bft_make_hashmod(r1, r2, out); }
extern "C" void bft_hosttest_free(void* hv) { delete (HostTrie*)hv; }

// raw copy of one host-built index array (same names as bft_gpu_debug_get_array)
extern "C" int bft_hosttest_get_array(void* hv, const char* name, void* out, uint64_t cap, uint64_t* nbytes) {
    HostTrie* t = (HostTrie*)hv;
    const std::string nm(name);
    const void* p = nullptr;
    uint64_t n = 0;
    if (nm == "nodes") { p = t->idx.nodes.data(); n = t->idx.nodes.size() * sizeof(BftNode); }
    else if (nm == "bfT") { p = t->idx.bfT.data(); n = t->idx.bfT.size(); }
    else if (nm == "ccs") { p = t->idx.ccs.data(); n = t->idx.ccs.size() * sizeof(BftCC); }
    else if (nm == "f2w") { p = t->idx.f2w.data(); n = t->idx.f2w.size() * 8; }
    else if (nm == "clus") { p = t->idx.clus.data(); n = t->idx.clus.size() * 8; }
    else if (nm == "ccx") { p = t->idx.ccx.data(); n = t->idx.ccx.size() * sizeof(BftCCX); }
    else if (nm == "f18") { p = t->idx.f18.data(); n = t->idx.f18.size() * 8; }
    else if (nm == "fent") { p = t->idx.fent.data(); n = t->idx.fent.size() * 8; }
    else if (nm == "child") { p = t->idx.child.data(); n = t->idx.child.size() * 8; }
    else if (nm == "uck") { p = t->idx.uck.data(); n = t->idx.uck.size() * 8; }
    else if (nm == "ucrow") { p = t->idx.ucrow.data(); n = t->idx.ucrow.size() * 4; }
    else if (nm == "tk") { p = t->tk.data(); n = t->tk.size() * 8; }
    else if (nm == "kh") { p = t->kh.lines.data(); n = t->kh.lines.size() * 8; }
    else return -1;
    if (nbytes) *nbytes = n;
    if (out) { if (cap < n) return -6; memcpy(out, p, n); }
    return 0;
}

// ---- .bft reader / writer (bft_file.cpp) on the CPU ----
#include "bft_file.h"

// product writer fed by the host-built index (single genome "genome_0")
extern "C" int bft_hosttest_write_bft(void* hv, const char* path) {
    HostTrie* t = (HostTrie*)hv;
    if (!bft_reference_k(t->k)) return -1;
    BftHostImage hi;
    hi.k = t->k; hi.r1 = BFT_DEFAULT_R1; hi.r2 = BFT_DEFAULT_R2;
    hi.genomes = {"genome_0"};
    hi.nodes = t->idx.nodes; hi.ccs = t->idx.ccs; hi.f2w = t->idx.f2w; hi.clus = t->idx.clus; hi.child = t->idx.child;
    hi.ucrow = t->idx.ucrow; hi.cs_off = t->cs_off;
    hi.tk.assign(t->tk.begin(), t->tk.end()); hi.tcol.assign(t->tcol.begin(), t->tcol.end()); hi.cs_ids.assign(t->cs_ids.begin(), t->cs_ids.end());
    std::string err;
    return bft_file_write(path, hi, err) ? 0 : -1;
}

// product reader: returns a handle whose per-genome k-mer lists can be copied out
extern "C" void* bft_hosttest_read_bft(const char* path, int* k, int* nb_genomes, uint64_t* n_kmers) {
    BftFileContent* fc = new BftFileContent();
    std::string err;
    if (!bft_file_read(path, *fc, err)) { delete fc; return nullptr; }
    *k = fc->k; *nb_genomes = (int)fc->per_genome.size(); *n_kmers = fc->n_kmers;
    return fc;
}
extern "C" uint64_t bft_hosttest_read_genome(void* fv, int g, uint8_t* out) {
    BftFileContent* fc = (BftFileContent*)fv;
    const std::vector<uint8_t>& v = fc->per_genome[g];
    if (out && !v.empty()) memcpy(out, v.data(), v.size());
    return v.size();
}
extern "C" void bft_hosttest_read_free(void* fv) { delete (BftFileContent*)fv; }

// the product's annotation encoder (bft_file.cpp: what bft_gpu_colorset_annot and the .bft writer emit), for the CPU tests
extern "C" int bft_hosttest_annot_encode(const uint32_t* ids, uint32_t n, uint8_t* out, uint32_t cap) {
    std::vector<uint8_t> enc;
    bft_annot_encode(ids, n, enc);
    if (enc.size() > cap) return -1;
    memcpy(out, enc.data(), enc.size());
    return (int)enc.size();
}
