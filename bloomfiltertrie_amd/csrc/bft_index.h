// bft_index.h -- host-side assembly of the BFT index arrays (BftImage) over a sorted T-form table.
//
// This is the bulk (static) counterpart of the reference's one-k-mer-at-a-time insertion
// (src/insertNode.c:38-226, :241-423; src/CC.c transform2CC / insertSP_CC): given ALL k-mers of a
// node it produces containers that satisfy the same invariants:
//   (i)   a prefix lives in the FIRST CC of its node whose Bloom filter holds its key n2..n8;
//         Bloom bits are only set by a CC's seed keys (<= 255 of them, the UC-burst size),
//   (ii)  the node UC holds < 255 k-mers, all Bloom-negative in every CC of the node,
//   (iii) a suffix group holds <= 255 rows, a larger one is a child Node (src/insertNode.c:291),
//   (v)   s = 4 / p = 14 iff the CC holds >= 3584 prefixes (src/insertNode.c:134-135),
//   (vi)  nb_elem fits uint16,
//   (vii) a non-empty UC implies the last CC holds >= 255 prefixes.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "bft_image.h"

struct BftHostIndex {
    std::vector<BftNode> nodes;
    std::vector<uint8_t> bfT;
    std::vector<BftCC> ccs;
    std::vector<uint64_t> f2w;
    std::vector<uint64_t> clus;
    std::vector<uint64_t> child;
    std::vector<uint64_t> uck;
    std::vector<uint32_t> ucrow;
    // derived by bft_flatten_index: extended CC headers and the flat form of the CCs with >= flat_min prefixes
    std::vector<BftCCX> ccx;
    std::vector<uint64_t> f18, fent;
    std::string error;
    // shape counters (printMemory.c-style)
    uint64_t n_child_nodes = 0, n_prefixes = 0, n_ccs_s4 = 0, max_ccs_per_node = 0;
};

// tk: n sorted distinct T-form k-mers (W = ceil(2k/64) words each, word 0 most significant).
bool bft_build_index(const uint64_t* tk, uint64_t n, int k, const uint32_t* hashmod, BftHostIndex& out);

// Flat form (bft_image.h) of the CCs holding at least flat_min prefixes, from ccs / f2w / clus / child.
void bft_flatten_index(BftHostIndex& io, uint32_t flat_min);

static inline int bft_words_for_k(int k) { return (2 * k + 63) / 64; }
static inline int bft_bytes_for_k(int k) { return (2 * k + 7) / 8; }
static inline bool bft_valid_k(int k) { return k >= 9 && k <= 126; }  // the reference additionally requires k % 9 == 0 (src/main.c:61-63)
static inline bool bft_reference_k(int k) { return bft_valid_k(k) && k % 9 == 0; }
