// bft_index.cpp -- host-side assembly of the BFT index arrays; see bft_index.h.
//
// Level-synchronous (breadth-first) and written as the sequence of data-parallel steps the GPU
// assembly (bft_assemble.hip) performs, so that both produce bit-identical arrays:
//   per depth: prefixes of the active nodes (runs of equal digit) -> Bloom keys -> per-node CC
//   assignment -> prefixes grouped by (node, CC) -> clusters -> prefix entries / child nodes -> UC rows.
// This file is only linked into the test helper library (libbft_hosttest.so); the product builds
// its image on the GPU.
#include "bft_index.h"

#include <string.h>

#include <algorithm>

#include "bft_walk.h"

namespace {

struct NodeRange {
    uint64_t lo, hi;
};

struct BloomBits {
    uint64_t w[(BFT_MODULO_HASH + 63) / 64];
    bool test(uint32_t h) const { return (w[h >> 6] >> (h & 63)) & 1ull; }
    void set(uint32_t h) { w[h >> 6] |= 1ull << (h & 63); }
};

template <int W>
bool run(const uint64_t* tk, uint64_t n, int k, const uint32_t* hm, BftHostIndex& o) {
    const int L = k / 9, rb = 2 * (k - 9 * L);
    std::vector<NodeRange> cur{{0, n}};
    uint32_t node_base = 0;  // global id of cur[0]
    for (int d = 0; d < L && !cur.empty(); d++) {
        const size_t M = cur.size();
        o.nodes.resize(node_base + M);
        // ---- 1. prefixes and keys of the active nodes ----
        std::vector<uint32_t> pref_r, pref_node, pref_key;
        std::vector<uint64_t> pref_row, pref_cnt;
        std::vector<uint32_t> key_val, key_node;
        std::vector<uint64_t> key_cnt;
        std::vector<uint32_t> node_kb(M + 1, 0), node_pb(M + 1, 0);
        for (size_t m = 0; m < M; m++) {
            node_kb[m] = (uint32_t)key_val.size();
            node_pb[m] = (uint32_t)pref_r.size();
            uint32_t cur_r = 0xFFFFFFFFu;
            for (uint64_t row = cur[m].lo; row < cur[m].hi; row++) {
                const uint32_t r = bft_digit<W>(tk + row * W, k, d);
                if (row == cur[m].lo || r != cur_r) {
                    if (row == cur[m].lo || (r >> 4) != (cur_r >> 4)) {
                        key_val.push_back(r >> 4);
                        key_node.push_back((uint32_t)m);
                        key_cnt.push_back(0);
                    }
                    pref_r.push_back(r);
                    pref_row.push_back(row);
                    pref_node.push_back((uint32_t)m);
                    pref_key.push_back((uint32_t)key_val.size() - 1);
                    pref_cnt.push_back(0);
                    cur_r = r;
                }
                pref_cnt.back()++;
                key_cnt.back()++;
            }
        }
        node_kb[M] = (uint32_t)key_val.size();
        node_pb[M] = (uint32_t)pref_r.size();
        const size_t P = pref_r.size(), K = key_val.size();

        // ---- 2. CC assignment per node (invariants (i), (ii), (vii) of bft_index.h) ----
        // One CC per pass while >= 255 k-mers are unassigned: its seeds are the first (at most) 255
        // still-unassigned keys in prefix order -- they set the Bloom bits -- and it claims every
        // unassigned key its Bloom filter holds.
        std::vector<int32_t> key_cc(K, -1);
        std::vector<uint32_t> node_ncc(M, 0), node_ccb(M + 1, 0);
        std::vector<BloomBits> blooms;  // in (node, cc) order
        for (size_t m = 0; m < M; m++) {
            node_ccb[m] = (uint32_t)blooms.size();
            uint64_t unassigned = cur[m].hi - cur[m].lo;
            uint32_t ncc = 0;
            while (unassigned >= BFT_NB_KMERS_PER_UC) {
                BloomBits b;
                memset(&b, 0, sizeof(b));
                uint32_t seeds = 0;
                for (uint32_t q = node_kb[m]; q < node_kb[m + 1] && seeds < BFT_NB_KMERS_PER_UC; q++) {
                    if (key_cc[q] >= 0) continue;
                    b.set(hm[key_val[q]] & 0xFFFFu);
                    b.set(hm[key_val[q]] >> 16);
                    seeds++;
                }
                for (uint32_t q = node_kb[m]; q < node_kb[m + 1]; q++) {
                    if (key_cc[q] >= 0) continue;
                    if (b.test(hm[key_val[q]] & 0xFFFFu) && b.test(hm[key_val[q]] >> 16)) {
                        key_cc[q] = (int32_t)ncc;
                        unassigned -= key_cnt[q];
                    }
                }
                blooms.push_back(b);
                ncc++;
                if (ncc > 65535) { o.error = "node with more than 65535 CCs"; return false; }
            }
            node_ncc[m] = ncc;
            if (ncc > o.max_ccs_per_node) o.max_ccs_per_node = ncc;
        }
        node_ccb[M] = (uint32_t)blooms.size();
        const size_t C = blooms.size();

        // ---- 3. prefixes grouped by (node, cc), prefix order kept; the UC pseudo-CC (cc = -1) first ----
        std::vector<uint32_t> sp(P);
        for (size_t p = 0; p < P; p++) sp[p] = (uint32_t)p;
        std::stable_sort(sp.begin(), sp.end(), [&](uint32_t a, uint32_t b) {
            if (pref_node[a] != pref_node[b]) return pref_node[a] < pref_node[b];
            return key_cc[pref_key[a]] < key_cc[pref_key[b]];
        });

        // ---- 4. CCs: headers, filter2 words, clusters, prefix entries; UC rows; child nodes ----
        const uint32_t cc_base = (uint32_t)o.ccs.size();
        o.ccs.resize(cc_base + C);
        std::vector<NodeRange> next;
        size_t q = 0;
        for (size_t m = 0; m < M; m++) {
            BftNode nd;
            memset(&nd, 0, sizeof(nd));
            nd.cc_first = cc_base + node_ccb[m];
            nd.ncc = (uint16_t)node_ncc[m];
            // UC rows
            nd.uc_first = (uint32_t)o.ucrow.size();
            uint32_t ucn = 0;
            while (q < P && pref_node[sp[q]] == m && key_cc[pref_key[sp[q]]] < 0) {
                const uint32_t p = sp[q];
                for (uint64_t row = pref_row[p]; row < pref_row[p] + pref_cnt[p]; row++) {
                    for (int w = 0; w < W; w++) o.uck.push_back(tk[row * W + w]);
                    o.ucrow.push_back((uint32_t)row);
                    ucn++;
                }
                q++;
            }
            nd.uc_n = (uint8_t)ucn;
            // bit-sliced Bloom block
            const uint32_t ncc = node_ncc[m];
            if (ncc) {
                const int wb = ncc <= 8 ? 1 : ncc <= 16 ? 2 : ncc <= 32 ? 4 : 8 * (int)((ncc + 63) / 64);
                if (wb > 255) { o.error = "node with too many CCs for bf_wb"; return false; }
                if (o.bfT.size() / 8 > 0xFFFFFFFFull) { o.error = "Bloom block offset overflow"; return false; }
                nd.bf_off = (uint32_t)(o.bfT.size() / 8);
                nd.bf_wb = (uint8_t)wb;
                const size_t base = o.bfT.size();
                o.bfT.resize(base + (size_t)BFT_MODULO_HASH * wb, 0);  // 1504*wb is a multiple of 8
                for (uint32_t c = 0; c < ncc; c++)
                    for (uint32_t h = 0; h < BFT_MODULO_HASH; h++)
                        if (blooms[node_ccb[m] + c].test(h)) o.bfT[base + (size_t)h * wb + (c >> 3)] |= (uint8_t)(1u << (c & 7));
            }
            // the node's CCs
            for (uint32_t c = 0; c < ncc; c++) {
                const size_t qb = q;
                while (q < P && pref_node[sp[q]] == m && key_cc[pref_key[sp[q]]] == (int32_t)c) q++;
                const size_t ne = q - qb;
                if (ne > 65535) { o.error = "CC with more than 65535 prefixes (nb_elem is uint16, include/CC.h:36)"; return false; }
                BftCC cc;
                memset(&cc, 0, sizeof(cc));
                cc.nb_elem = (uint16_t)ne;
                cc.s = ne >= BFT_TRESH_SUF_PREF ? 4 : 8;
                if (cc.s == 4) o.n_ccs_s4++;
                o.n_prefixes += ne;
                const size_t nwords = ((size_t(1) << (18 - cc.s)) + BFT_F2_BITS_PER_WORD - 1) / BFT_F2_BITS_PER_WORD;
                if (o.f2w.size() + nwords > 0xFFFFFFFFull || o.clus.size() + ne > 0xFFFFFFFFull || o.child.size() + ne > 0xFFFFFFFFull) {
                    o.error = "index array offset overflow (u32)";
                    return false;
                }
                cc.f2_off = (uint32_t)o.f2w.size();
                cc.clus_off = (uint32_t)o.clus.size();
                cc.child_off = (uint32_t)o.child.size();
                o.f2w.resize(o.f2w.size() + nwords, 0);
                uint64_t* f2 = &o.f2w[cc.f2_off];
                for (size_t j = qb; j < q;) {
                    const uint32_t pu = pref_r[sp[j]] >> cc.s;
                    size_t j2 = j;
                    while (j2 + 1 < q && (pref_r[sp[j2 + 1]] >> cc.s) == pu) j2++;
                    const size_t len = j2 - j + 1;
                    f2[pu / BFT_F2_BITS_PER_WORD] |= 1ull << (pu % BFT_F2_BITS_PER_WORD);
                    const size_t clus_slot = o.clus.size();
                    if (len == 1) o.clus.push_back(0);
                    else o.clus.push_back(BFT_CLUS_MULTI | ((uint64_t)len << BFT_CLUS_LEN_SHIFT) | (uint64_t)(o.child.size() - cc.child_off));
                    for (size_t z = j; z <= j2; z++) {
                        const uint32_t p = sp[z];
                        const uint32_t pv = pref_r[p] & ((1u << cc.s) - 1u);
                        const uint64_t cnt = pref_cnt[p];
                        uint64_t ent = (uint64_t)pv << BFT_CHILD_PV_SHIFT;
                        if (d == L - 1 && rb == 0) ent |= (1ull << BFT_CHILD_CNT_SHIFT) | pref_row[p];  // leaf: one annotation per prefix
                        else if (d == L - 1)  // k % 9 != 0: the rows of the prefix differ in the remaining nucleotides; count-1 on 16 bits
                            ent = BFT_REM_ENTRY(pv, cnt, pref_row[p]);
                        else if (cnt <= BFT_NB_KMERS_PER_UC) ent |= (cnt << BFT_CHILD_CNT_SHIFT) | pref_row[p];
                        else {  // > 255 suffixes: child Node (count field 0), ids in breadth-first order
                            ent |= (uint64_t)(node_base + M + next.size());
                            next.push_back(NodeRange{pref_row[p], pref_row[p] + cnt});
                            o.n_child_nodes++;
                        }
                        if (len == 1) o.clus[clus_slot] = ent;
                        else o.child.push_back(ent);
                    }
                    j = j2 + 1;
                }
                uint32_t rank = 0;
                for (size_t w = 0; w < nwords; w++) {
                    const uint32_t pc = (uint32_t)__builtin_popcountll(f2[w]);
                    f2[w] |= (uint64_t)rank << 48;
                    rank += pc;
                }
                o.ccs[cc_base + node_ccb[m] + c] = cc;
            }
            o.nodes[node_base + m] = nd;
        }
        node_base += (uint32_t)M;
        cur.swap(next);
    }
    return o.error.empty();
}

}  // namespace

bool bft_build_index(const uint64_t* tk, uint64_t n, int k, const uint32_t* hashmod, BftHostIndex& out) {
    out = BftHostIndex();
    if (!bft_valid_k(k)) { out.error = "k must be in [9, 126]"; return false; }
    if (n >= 0xFFFFFFFFull) { out.error = "more than 2^32-1 k-mers: row index overflow"; return false; }
    switch (bft_words_for_k(k)) {
    case 1: return run<1>(tk, n, k, hashmod, out);
    case 2: return run<2>(tk, n, k, hashmod, out);
    case 3: return run<3>(tk, n, k, hashmod, out);
    default: return run<4>(tk, n, k, hashmod, out);
    }
}

void bft_flatten_index(BftHostIndex& io, uint32_t flat_min) {
    io.ccx.assign(io.ccs.size(), BftCCX());
    io.f18.clear();
    io.fent.clear();
    for (size_t c = 0; c < io.ccs.size(); c++) {
        const BftCC& cc = io.ccs[c];
        BftCCX x;
        memset(&x, 0, sizeof(x));
        x.f2_off = cc.f2_off; x.clus_off = cc.clus_off; x.child_off = cc.child_off; x.nb_elem = cc.nb_elem; x.s = cc.s;
        if (cc.nb_elem >= flat_min) {
            x.flat = 1;
            x.f18_off = (uint32_t)io.f18.size();
            x.fent_off = (uint32_t)io.fent.size();
            io.f18.resize(io.f18.size() + BFT_F18_WORDS, 0);
            uint64_t* f = &io.f18[x.f18_off];
            const uint32_t nw = ((1u << (18 - cc.s)) + BFT_F2_BITS_PER_WORD - 1) / BFT_F2_BITS_PER_WORD;
            for (uint32_t w = 0; w < nw; w++) {
                const uint64_t fw = io.f2w[cc.f2_off + w];
                uint32_t clu = (uint32_t)(fw >> 48);
                for (uint32_t b = 0; b < BFT_F2_BITS_PER_WORD; b++) {
                    if (!((fw >> b) & 1ull)) continue;
                    const uint32_t pu = w * BFT_F2_BITS_PER_WORD + b;
                    const uint64_t e = io.clus[cc.clus_off + clu++];
                    const uint32_t len = (e & BFT_CLUS_MULTI) ? (uint32_t)((e >> BFT_CLUS_LEN_SHIFT) & 0xFFFFu) : 1u;
                    for (uint32_t j = 0; j < len; j++) {
                        const uint64_t ent = (e & BFT_CLUS_MULTI) ? io.child[cc.child_off + (uint32_t)e + j] : e;
                        const uint32_t r = (pu << cc.s) | ((uint32_t)(ent >> BFT_CHILD_PV_SHIFT) & 0xFFu);
                        io.fent.push_back(ent);
                        f[r / BFT_F2_BITS_PER_WORD] |= 1ull << (r % BFT_F2_BITS_PER_WORD);
                    }
                }
            }
            uint32_t rank = 0;
            for (uint32_t w = 0; w < BFT_F18_WORDS; w++) {
                const uint32_t pc = (uint32_t)__builtin_popcountll(f[w]);
                f[w] |= (uint64_t)rank << 48;
                rank += pc;
            }
        }
        io.ccx[c] = x;
    }
}
