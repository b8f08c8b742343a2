// bft_index.cpp -- see bft_index.h.
#include "bft_index.h"

#include <string.h>

#include "bft_walk.h"

namespace {

struct Pref {
    uint32_t r;
    uint64_t s, e;  // rows [s, e) of tk under this prefix
};

struct KeyEnt {
    uint32_t key;
    uint32_t p0, p1;  // prefs [p0, p1)
    uint64_t nk;      // k-mers under the key
    int cc;           // -1 = unassigned (node UC)
};

struct BloomBits {
    uint64_t w[(BFT_MODULO_HASH + 63) / 64];
    int nseeds;
    bool test(uint32_t h) const { return (w[h >> 6] >> (h & 63)) & 1ull; }
    void set(uint32_t h) { w[h >> 6] |= 1ull << (h & 63); }
};

template <int W>
struct Builder {
    const uint64_t* tk;
    int k, L;
    const uint32_t* hm;
    BftHostIndex& o;

    uint32_t build_node(uint64_t lo, uint64_t hi, int d) {
        const uint32_t id = (uint32_t)o.nodes.size();
        o.nodes.push_back(BftNode{});
        const uint64_t n = hi - lo;
        BftNode nd;
        memset(&nd, 0, sizeof(nd));

        if (n < BFT_NB_KMERS_PER_UC) {  // the UC never filled up: no CC (src/insertNode.c:183-223)
            nd.uc_first = (uint32_t)o.ucrow.size();
            nd.uc_n = (uint8_t)n;
            for (uint64_t row = lo; row < hi; row++) {
                for (int w = 0; w < W; w++) o.uck.push_back(tk[row * W + w]);
                o.ucrow.push_back((uint32_t)row);
            }
            o.nodes[id] = nd;
            return id;
        }

        // distinct prefixes of this level, in filter3 order
        std::vector<Pref> prefs;
        {
            uint32_t cur = 0xFFFFFFFFu;
            for (uint64_t row = lo; row < hi; row++) {
                uint32_t r = bft_digit<W>(tk + row * W, L, d);
                if (r != cur) {
                    if (!prefs.empty()) prefs.back().e = row;
                    prefs.push_back(Pref{r, row, hi});
                    cur = r;
                }
            }
        }
        // Bloom keys n2..n8 = r >> 4: the 16 prefixes of a key are adjacent in r order
        std::vector<KeyEnt> keys;
        for (uint32_t p = 0; p < prefs.size(); p++) {
            uint32_t key = prefs[p].r >> 4;
            if (keys.empty() || keys.back().key != key) keys.push_back(KeyEnt{key, p, p + 1, 0, -1});
            keys.back().p1 = p + 1;
            keys.back().nk += prefs[p].e - prefs[p].s;
        }

        // CC assignment (invariants (i), (ii), (vii) of bft_index.h)
        std::vector<BloomBits> blooms;
        uint64_t unassigned = n;
        while (unassigned >= BFT_NB_KMERS_PER_UC) {
            BloomBits b;
            memset(&b, 0, sizeof(b));
            const int ci = (int)blooms.size();
            for (auto& ke : keys) {
                if (ke.cc >= 0) continue;
                const uint32_t h1 = hm[ke.key] & 0xFFFFu, h2 = hm[ke.key] >> 16;
                bool take = b.test(h1) && b.test(h2);
                if (!take && b.nseeds < BFT_NB_KMERS_PER_UC) {
                    b.set(h1);
                    b.set(h2);
                    b.nseeds++;
                    take = true;
                }
                if (take) {
                    ke.cc = ci;
                    unassigned -= ke.nk;
                }
            }
            blooms.push_back(b);
        }
        const int ncc = (int)blooms.size();
        if (ncc > 65535) { o.error = "node with more than 65535 CCs"; return id; }
        if ((uint64_t)ncc > o.max_ccs_per_node) o.max_ccs_per_node = ncc;

        // bit-sliced Bloom block
        int wb = ncc <= 8 ? 1 : ncc <= 16 ? 2 : ncc <= 32 ? 4 : 8 * ((ncc + 63) / 64);
        if (wb > 255) { o.error = "node with too many CCs for bf_wb"; return id; }
        while (o.bfT.size() % 8) o.bfT.push_back(0);
        if (o.bfT.size() / 8 > 0xFFFFFFFFull) { o.error = "Bloom block offset overflow"; return id; }
        nd.bf_off = (uint32_t)(o.bfT.size() / 8);
        nd.bf_wb = (uint8_t)wb;
        {
            size_t base = o.bfT.size();
            o.bfT.resize(base + (size_t)BFT_MODULO_HASH * wb, 0);
            for (int c = 0; c < ncc; c++)
                for (uint32_t h = 0; h < BFT_MODULO_HASH; h++)
                    if (blooms[c].test(h)) o.bfT[base + (size_t)h * wb + (c >> 3)] |= (uint8_t)(1u << (c & 7));
        }

        // per-CC prefix lists
        std::vector<std::vector<uint32_t>> ccprefs(ncc);
        nd.uc_first = (uint32_t)o.ucrow.size();
        uint32_t ucn = 0;
        for (auto& ke : keys) {
            if (ke.cc >= 0) {
                for (uint32_t p = ke.p0; p < ke.p1; p++) ccprefs[ke.cc].push_back(p);
            } else {
                for (uint32_t p = ke.p0; p < ke.p1; p++)
                    for (uint64_t row = prefs[p].s; row < prefs[p].e; row++) {
                        for (int w = 0; w < W; w++) o.uck.push_back(tk[row * W + w]);
                        o.ucrow.push_back((uint32_t)row);
                        ucn++;
                    }
            }
        }
        nd.uc_n = (uint8_t)ucn;
        nd.ncc = (uint16_t)ncc;
        nd.cc_first = (uint32_t)o.ccs.size();
        o.ccs.resize(o.ccs.size() + ncc);

        struct Pending { bool in_clus; size_t slot; uint64_t s, e; };
        std::vector<Pending> pending;

        for (int c = 0; c < ncc; c++) {
            const std::vector<uint32_t>& pl = ccprefs[c];
            const size_t ne = pl.size();
            if (ne > 65535) { o.error = "CC with more than 65535 prefixes (nb_elem is uint16, include/CC.h:36)"; return id; }
            BftCC cc;
            memset(&cc, 0, sizeof(cc));
            cc.nb_elem = (uint16_t)ne;
            cc.s = ne >= BFT_TRESH_SUF_PREF ? 4 : 8;
            if (cc.s == 4) o.n_ccs_s4++;
            o.n_prefixes += ne;
            const int p = 18 - cc.s;
            const size_t nwords = ((size_t(1) << p) + BFT_F2_BITS_PER_WORD - 1) / BFT_F2_BITS_PER_WORD;
            cc.f2_off = (uint32_t)o.f2w.size();
            cc.clus_off = (uint32_t)o.clus.size();
            cc.child_off = (uint32_t)o.child.size();
            if (o.f2w.size() + nwords > 0xFFFFFFFFull || o.clus.size() + ne > 0xFFFFFFFFull ||
                o.child.size() + ne > 0xFFFFFFFFull) {
                o.error = "index array offset overflow (u32)";
                return id;
            }
            o.f2w.resize(o.f2w.size() + nwords, 0);
            uint64_t* f2 = &o.f2w[cc.f2_off];
            // clusters = runs of equal p_u; a run of one prefix is stored inline in clus[]
            for (size_t j = 0; j < ne;) {
                const uint32_t pu = prefs[pl[j]].r >> cc.s;
                size_t j2 = j;
                while (j2 + 1 < ne && (prefs[pl[j2 + 1]].r >> cc.s) == pu) j2++;
                const size_t len = j2 - j + 1;
                f2[pu / BFT_F2_BITS_PER_WORD] |= 1ull << (pu % BFT_F2_BITS_PER_WORD);
                const size_t clus_slot = o.clus.size();
                if (len == 1) o.clus.push_back(0);
                else o.clus.push_back(BFT_CLUS_MULTI | ((uint64_t)len << BFT_CLUS_LEN_SHIFT) | (uint64_t)(o.child.size() - cc.child_off));
                for (size_t q = j; q <= j2; q++) {
                    const Pref& pf = prefs[pl[q]];
                    const uint32_t pv = pf.r & ((1u << cc.s) - 1u);
                    const uint64_t cnt = pf.e - pf.s;
                    const uint64_t pvf = (uint64_t)pv << BFT_CHILD_PV_SHIFT;
                    uint64_t ent;
                    bool pend = false;
                    if (d == L - 1) ent = pvf | (1ull << BFT_CHILD_CNT_SHIFT) | pf.s;  // leaf: one annotation per prefix
                    else if (cnt <= BFT_NB_KMERS_PER_UC) ent = pvf | (cnt << BFT_CHILD_CNT_SHIFT) | pf.s;
                    else { ent = pvf; pend = true; }  // > 255 suffixes: child Node, id patched in below
                    if (len == 1) {
                        o.clus[clus_slot] = ent;
                        if (pend) pending.push_back(Pending{true, clus_slot, pf.s, pf.e});
                    } else {
                        if (pend) pending.push_back(Pending{false, o.child.size(), pf.s, pf.e});
                        o.child.push_back(ent);
                    }
                }
                j = j2 + 1;
            }
            uint32_t rank = 0;
            for (size_t w = 0; w < nwords; w++) {
                uint32_t pc = (uint32_t)__builtin_popcountll(f2[w]);
                f2[w] |= (uint64_t)rank << 48;
                rank += pc;
            }
            o.ccs[nd.cc_first + c] = cc;
        }
        o.nodes[id] = nd;

        for (const Pending& pe : pending) {
            uint32_t child = build_node(pe.s, pe.e, d + 1);
            if (!o.error.empty()) return id;
            (pe.in_clus ? o.clus[pe.slot] : o.child[pe.slot]) |= (uint64_t)child;  // count field 0 => child Node
            o.n_child_nodes++;
        }
        return id;
    }
};

template <int W>
bool run(const uint64_t* tk, uint64_t n, int k, const uint32_t* hashmod, BftHostIndex& out) {
    Builder<W> b{tk, k, k / 9, hashmod, out};
    b.build_node(0, n, 0);
    return out.error.empty();
}

}  // namespace

bool bft_build_index(const uint64_t* tk, uint64_t n, int k, const uint32_t* hashmod, BftHostIndex& out) {
    out = BftHostIndex();
    if (!bft_valid_k(k)) { out.error = "k must be a multiple of 9 in [9, 126]"; return false; }
    if (n >= 0xFFFFFFFFull) { out.error = "more than 2^32-1 k-mers: row index overflow"; return false; }
    switch (bft_words_for_k(k)) {
    case 1: return run<1>(tk, n, k, hashmod, out);
    case 2: return run<2>(tk, n, k, hashmod, out);
    case 3: return run<3>(tk, n, k, hashmod, out);
    default: return run<4>(tk, n, k, hashmod, out);
    }
}
