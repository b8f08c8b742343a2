// bft_kh_host.h -- sequential restatement of the k-mer hash build (bft_image.h, BFT_KH_*; the GPU build is bft_kh.hip): the canonical
// layout written down the slow, obvious way.  Host code of the TEST helper library only (bft_hosttest.cpp): the product builds on the GPU.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "bft_image.h"
#include "bft_walk.h"

struct BftKhHostTable {
    BftKhGeo g;
    std::vector<uint64_t> lines;   // 8 words per line, g.nl + BFT_KH_TAIL_LINES lines
    std::vector<uint64_t> ovf_k;   // overflow list: sorted T-form rows (W words each) ...
    std::vector<uint32_t> ovf_v;   // ... and their values
    bool ok = false;
    uint32_t max_d = 0;
    uint64_t lines_used = 0;
};

// tk: n sorted T-form rows of W words; vals[i] < n_values.  Rows in (home line, row) order; slot p_i = max(home slot, p_(i-1) + 1); a row
// further from home than the slots' displacement bits hold goes to the overflow list (rows come in T order within a line and the list is
// filled in (home line, T) order, then sorted by T).
template <int W>
static void bft_kh_build_host(const uint64_t* tk, const uint32_t* vals, uint64_t n, int k, uint64_t n_values, uint32_t load_pct, BftKhHostTable& out) {
    out.g = bft_kh_geometry(k, n, n_values, load_pct);
    const BftKhGeo& g = out.g;
    out.lines.assign((g.nl + BFT_KH_TAIL_LINES) * BFT_KH_LINE_WORDS, 0ull);
    out.ovf_k.clear();
    out.ovf_v.clear();
    out.ok = true;
    out.max_d = 0;
    out.lines_used = 0;
    std::vector<std::pair<uint64_t, uint64_t>> order(n);  // (home line, row)
    for (uint64_t i = 0; i < n; i++) {
        BftKhKey<W> key;
        bft_kh_key<W>(tk + i * W, k, g, key);
        order[i] = {key.home, i};
    }
    std::stable_sort(order.begin(), order.end(), [](const std::pair<uint64_t, uint64_t>& a, const std::pair<uint64_t, uint64_t>& b) { return a.first < b.first; });
    std::vector<uint64_t> ovf_rows;
    uint64_t p = 0;
    for (uint64_t x = 0; x < n; x++) {
        const uint64_t home = order[x].first;
        p = x == 0 ? home * g.S : std::max(home * g.S, p + 1);
        const uint64_t ln = p / g.S, d = ln - home;
        if (d > g.maxd) {  // the overflow list; its place in the probe sequence stays taken (a tombstone: in use, value 0)
            ovf_rows.push_back(order[x].second);
            if (ln >= g.nl + BFT_KH_TAIL_LINES) out.ok = false;
            else out.lines[ln * BFT_KH_LINE_WORDS + 1] |= 1ull << (64u - g.S + (uint32_t)(p % g.S));
            continue;
        }
        out.max_d = std::max<uint32_t>(out.max_d, (uint32_t)d);
        uint64_t img[BFT_KH_LINE_WORDS];
        bft_kh_slot_image<W>(tk + order[x].second * W, k, g, (uint32_t)(p % g.S), (uint32_t)d, vals[order[x].second], img);
        uint64_t* line = out.lines.data() + ln * BFT_KH_LINE_WORDS;
        for (uint32_t q = 0; q < BFT_KH_LINE_WORDS; q++) line[q] |= img[q];
        out.lines_used = ln + 1;
    }
    if (ovf_rows.size() > BFT_KH_OVF_CAP) { out.ok = false; return; }
    std::sort(ovf_rows.begin(), ovf_rows.end());  // (rows of the sorted table: T order)
    for (uint64_t r : ovf_rows) {
        for (int w = 0; w < W; w++) out.ovf_k.push_back(tk[r * W + w]);
        out.ovf_v.push_back(vals[r]);
    }
    out.g.maxd = out.max_d;
}
