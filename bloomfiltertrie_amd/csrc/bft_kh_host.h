// bft_kh_host.h -- sequential restatement of the k-mer hash build (bft_image.h, BFT_KH_*; the GPU build is bft_kh.hip): the canonical
// layout written down the slow, obvious way.  Host code of the TEST helper library only (bft_hosttest.cpp): the product builds on the GPU.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "bft_image.h"
#include "bft_walk.h"

struct BftKhHostTable {
    uint32_t S = 0, f = 0, wb = 0, rb = 0, cb = 0;
    std::vector<uint64_t> lines;   // 8 words per line
    std::vector<uint32_t> kreg;    // 2^18 + 1
    bool ok = false;
    uint32_t max_tail = 0;
};

// tk: n sorted T-form rows of W words; vals[i] < n_values.  Rows of a region in (home line, row) order; slot p_i = max(home slot, p_(i-1) + 1);
// lines of the region = max(home lines + 1, lines touched, + 1 when the last touched line is full).
template <int W>
static void bft_kh_build_host(const uint64_t* tk, const uint32_t* vals, uint64_t n, int k, uint64_t n_values, uint32_t load_pct, BftKhHostTable& out) {
    out.rb = bft_kh_rb(k);
    out.cb = bft_kh_value_bits(n_values);
    out.S = bft_kh_slots_for(out.rb, out.cb);
    out.f = bft_kh_field_bits(out.S, out.rb);
    out.wb = bft_kh_body_bytes(out.S);
    out.kreg.assign(BFT_KH_REGIONS + 1, 0);
    out.lines.clear();
    out.ok = true;
    out.max_tail = 0;
    const uint32_t S = out.S;
    uint64_t i = 0, line0 = 0;
    std::vector<std::pair<uint32_t, uint64_t>> order;  // (home line, row)
    for (uint32_t r = 0; r < BFT_KH_REGIONS; r++) {
        uint64_t e = i;
        while (e < n && bft_digit<W>(tk + e * W, k, 0) == r) e++;
        const uint64_t nr = e - i;
        uint32_t L = 0, tail = 0;
        if (nr) {
            const uint32_t mh = bft_kh_home_lines(nr, S, load_pct);
            order.clear();
            for (uint64_t x = i; x < e; x++) {
                uint64_t remle[W];
                bft_kh_rem<W>(tk + x * W, k, remle);
                order.push_back({bft_kh_home_of(bft_kh_hash<W>(remle), mh), x});
            }
            std::stable_sort(order.begin(), order.end(), [](const std::pair<uint32_t, uint64_t>& a, const std::pair<uint32_t, uint64_t>& b) { return a.first < b.first; });
            std::vector<uint64_t> pos(nr);
            uint64_t p = 0;
            for (uint64_t x = 0; x < nr; x++) {
                const uint64_t home = (uint64_t)order[x].first * S;
                p = x == 0 ? home : std::max(home, p + 1);
                pos[x] = p;
            }
            uint64_t used = p / S + 1;
            if (p % S == S - 1) used++;
            const uint32_t code = bft_kh_tail_code(used > mh ? used - mh : 1);
            if (code > 3u) out.ok = false;
            tail = code & 3u;  // (the code; the lines: BFT_KH_TAIL_OF)
            L = mh + BFT_KH_TAIL_OF(tail);
            out.max_tail = std::max(out.max_tail, BFT_KH_TAIL_OF(tail));
            out.lines.resize((line0 + L) * BFT_KH_LINE_WORDS, 0ull);
            for (uint64_t x = 0; x < nr; x++) {
                uint64_t img[BFT_KH_LINE_WORDS];
                bft_kh_slot_image<W>(tk + order[x].second * W, k, S, out.rb, out.f, out.wb, out.cb, (uint32_t)(pos[x] % S), vals[order[x].second], img);
                uint64_t* line = out.lines.data() + (line0 + pos[x] / S) * BFT_KH_LINE_WORDS;
                for (uint32_t q = 0; q < BFT_KH_LINE_WORDS; q++) line[q] |= img[q];
            }
        }
        out.kreg[r] = (uint32_t)line0 | (tail << BFT_KREG_TAIL_SHIFT);
        line0 += L;
        i = e;
    }
    out.kreg[BFT_KH_REGIONS] = (uint32_t)line0;
}
