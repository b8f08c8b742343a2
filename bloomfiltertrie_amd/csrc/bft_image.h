// bft_image.h -- flattened, pointer-free HBM image of a Bloom Filter Trie.
//
// The reference links Node -> CC[] -> {BF_filter2, filter3, extra_filter3, children_type,
// children UC[], children_Node_container} through heap pointers (include/Node.h:55-58,
// include/CC.h:34-67) and finds children by counting from the left (include/CC.h:471-550).
// The serialized .bft holds neither the Bloom filters nor the skip tables, so the
// device layout is ours; only query results must match.  Layout, MI355X-first:
//
//  * k-mers are kept in "T-form": the L = k/9 rotated 18-bit prefixes r_d = n2..n9,n1 of
//    src/presenceNode.c:1367-1371, most significant first, in W = ceil(2k/64) u64 words
//    (word 0 most significant).  Sorting T-form integers is a DFS of the trie with every CC's
//    prefixes in filter3 order, so ONE sorted table `tk` holds every suffix group and every child
//    Node as a contiguous run; containers are offsets into it.  Stripping 9 nt per level
//    (src/presenceNode.c:1853-1861) becomes a digit index.
//  * per node, the CC Bloom filters are bit-sliced: for each of the 1504 bit positions a mask over
//    the node's CCs, so "first BF-positive CC" (src/presenceNode.c:1353-1362) is
//    two loads, an AND and a count-trailing-zeros instead of 2 byte loads per CC.
//  * filter2 is stored 48 bits + 16-bit running rank per u64: bit test and rank
//    (src/presenceNode.c:1619-1636, SkipFilter2) are one load + one popcount.
//  * filter3 (the p_v of each prefix, src/presenceNode.c:1399-1410,1472-1489) and children_type counting
//    (include/CC.h:471-550) are fused into ONE u64 "prefix entry" {p_v:8 | count:8 | row-or-node:40}.
//  * extra_filter3 select + cluster length (src/presenceNode.c:1648-1688, SkipFilter3) is a u64 table
//    indexed by the filter2 rank: a cluster of one prefix (the common case) holds that prefix entry
//    inline; a longer cluster holds {bit63 | length:16 | start:32} into the CC's run of `child[]`,
//    where its entries are binary-searched on p_v.
//  * CCs in s = 4 mode (>= 3584 prefixes: the big CCs every query of a level funnels through) also get a FLAT form,
//    derived from the arrays above when the image is bound (bft_flatten_gpu): one bit per 18-bit rotated prefix r,
//    48 bits + 16-bit running rank per u64 (`f18`), and all prefix entries of the CC in r order (`fent`).  filter2 bit
//    test + rank + cluster select + filter3 search then cost two dependent loads instead of three to five.  The walk reads
//    CC headers in their 32-byte extended form `ccx` (the 16-byte header + the two flat offsets).
#pragma once
#include <stdint.h>

#define BFT_MAX_W 4            // k <= 126 -> 252 bits
#define BFT_MODULO_HASH 1504   // include/default_param.h:41
#define BFT_NB_KMERS_PER_UC 255 // include/default_param.h:17-31
#define BFT_TRESH_SUF_PREF 3584 // include/default_param.h:42
#define BFT_F2_BITS_PER_WORD 48

struct BftNode {          // 16 B
    uint32_t cc_first;    // index of the node's first CC in ccs[]
    uint32_t bf_off;      // offset of the bit-sliced Bloom block in bfT, in 8-byte units
    uint32_t uc_first;    // first row of the node's UC in uck[] / ucrow[]
    uint16_t ncc;         // number of CCs
    uint8_t uc_n;         // UC rows (< 255, include/default_param.h:31)
    uint8_t bf_wb;        // bytes per Bloom bit position: 1, 2, 4 or 8*ceil(ncc/64)
};

struct BftCC {            // 16 B: one dwordx4 load
    uint32_t f2_off;      // into f2w[] (u64 units)
    uint32_t clus_off;    // into clus[] (u64 units); one entry per cluster (= per set filter2 bit)
    uint32_t child_off;   // into child[] (u64 units); prefix entries of the multi-prefix clusters
    uint16_t nb_elem;     // prefixes in this CC (include/CC.h:36)
    uint8_t s;            // length of p_v in bits: 8, or 4 once nb_elem >= 3584 (src/insertNode.c:134-135)
    uint8_t pad0;
};

#define BFT_F18_WORDS 5462u  // ceil(2^18 / 48) u64 words of a flat CC's prefix bitmap

struct BftCCX {           // 32 B: what the walk reads per CC
    uint32_t f2_off, clus_off, child_off;
    uint16_t nb_elem;
    uint8_t s;
    uint8_t flat;         // 1: f18_off / fent_off are valid and the walk uses them
    uint32_t f18_off;     // into f18[] (u64 units), BFT_F18_WORDS words
    uint32_t fent_off;    // into fent[] (u64 units), nb_elem entries in r order
    uint32_t pad[2];
};

#if defined(__cplusplus)
static_assert(sizeof(BftCC) == 16 && sizeof(BftCCX) == 32, "CC headers are 16 / 32 bytes");
static_assert(__builtin_offsetof(BftCCX, flat) == __builtin_offsetof(BftCC, pad0) && __builtin_offsetof(BftCCX, s) == __builtin_offsetof(BftCC, s),
              "the first half of BftCCX is laid out like BftCC");
#endif

#define BFT_CHILD_IDX_MASK 0xFFFFFFFFFFULL
#define BFT_CHILD_CNT_SHIFT 40
#define BFT_CHILD_PV_SHIFT 48
// Last level of a k % 9 != 0 index only ("remainder groups", up to 4^8 rows): count-1 on 16 bits = bits 40..47 (low 8),
// bits 56..62 (next 7; bit 63 stays free for BFT_CLUS_MULTI) and bit 39 (the top one; rows are < 2^31, so bit 39 of
// the row field is otherwise always 0).
#define BFT_REM_ENTRY(pv, cnt, row)                                                                                                  \
    (((uint64_t)(pv) << BFT_CHILD_PV_SHIFT) | ((((uint64_t)(cnt)-1) & 0xFFull) << BFT_CHILD_CNT_SHIFT) |                           \
     (((((uint64_t)(cnt)-1) >> 8) & 0x7Full) << 56) | (((((uint64_t)(cnt)-1) >> 15) & 1ull) << 39) | (uint64_t)(row))
#define BFT_REM_COUNT(e) ((uint32_t)((((e) >> BFT_CHILD_CNT_SHIFT) & 0xFFull) | ((((e) >> 56) & 0x7Full) << 8) | ((((e) >> 39) & 1ull) << 15)) + 1u)
#define BFT_REM_ROW(e) ((e) & 0x7FFFFFFFFFull)
#define BFT_CLUS_MULTI (1ull << 63)
// Root direct table (derived, optional): one u64 per 18-bit rotated prefix r of the ROOT node = the outcome of the root
// level's Bloom probe + CC lookup for r.  A prefix lives in the first CC whose Bloom filter holds its key, so
// the outcome is a function of r alone: BFT_RDIR_NO_CC (no Bloom-positive CC: search the node's UC), BFT_RDIR_ABSENT (a CC
// claims the key but does not hold r: absent, src/presenceNode.c:1546-1548), or the prefix entry of r with bit 63 set.
// 2 MiB for the root of any index; child nodes keep the container walk.
// Root range table (derived with it, 1 MiB): rstart[r] = first row of the sorted table whose root prefix is >= r (r = 0..2^18), so
// the k-mers under root prefix r are the rows [rstart[r], rstart[r+1]).  Bit 31 of rstart[r] = "special": the prefix is not a plain
// suffix group of the root (child Node, rows held by the root's UC, last level of a k <= 17 index) and the lookup takes its
// rdir entry instead.  For a plain prefix the two words replace the 8-byte entry: an empty range is an absent prefix, a non-empty one
// is the suffix group {first row, count} -- the same {row, count} its prefix entry holds.  Half the footprint of rdir, so it stays
// in the 4 MiB L2 next to the table stream (measured: 0.35 -> 0.05 L2 misses per query at the root).
#define BFT_RSTART_SPECIAL 0x80000000u
// Root quartile table (derived with the range table, 1 MiB, optional): for a plain suffix group [a, a + n) of the root, rq[r] holds the
// offsets (8 bits each, n <= 255) of the first row whose next two key bits -- the top two bits of the level-1 digit -- are >= 1, 2, 3.
// The suffixes of a group are searched from an interpolated guess (bft_group_search); a pan-genome group is a handful of clusters of
// SNP variants, not a uniform sample, and the guess over the whole group lands in the wrong cache line half the time.  The quarter
// the k-mer falls in is exact, four times smaller and interpolated on its own: an empty quarter is an absent k-mer without a single
// table line read (2.30 -> 2.04 L2 misses per query on the config-4 share, profiles/r04).  Same answers with or without it (the
// quarters are exact sub-ranges of the group).
#define BFT_RQ_OFF(q, j) (((q) >> (8 * ((j)-1))) & 0xFFu)
#define BFT_RDIR_NO_CC 0ull
#define BFT_RDIR_ABSENT 1ull
#define BFT_RDIR_VALID (1ull << 63)
#define BFT_CLUS_LEN_SHIFT 32

// k-mer hash (derived when the image is built, loaded or unpacked; optional; any k): EVERY stored k-mer, whatever container of the trie
// holds it, in one open-addressed table of 64-byte lines.  A k-mer lives in the first line at or after its HOME line that had a free slot
// when the table was laid out; a lookup reads lines from the home line on until it meets the key (present; its colour set sits in the
// same slot), a line with a free slot (absent), or has looked as far past home as any k-mer of the table is displaced -- then the
// OVERFLOW LIST decides: the handful of k-mers (none, on most indexes) whose run of full lines was longer than a slot's displacement bits
// hold, sorted, searched only by the one lookup in millions that meets such a run.  (In the canonical layout the slot such a k-mer would
// have taken is marked in use with value 0 -- a tombstone no lookup matches --: the lines before the k-mers behind it stay full.)  At the default occupancy of the
// home lines (55 %: "kmer_hash_load") a lookup reads 1.09 lines: ONE cache line beyond the L2 per query, where the container walk of
// src/presenceNode.c:1284-1921 costs a line per level and per suffix-group probe even in its fastest form here; on MI355X a kernel of
// random gathers is bound by the lines it misses on (tools/microbench/gather.hip: ~55 G lines/s beyond the L2, whether the lane reads 8
// or 128 bytes of the line), so lines per query is the whole cost.
//   key     The home line is computed from the top hb0 = min(32, 2k - 4) bits of the T-form WITHOUT the two bits of the k-mer's first
//           nucleotide (hi, hb = hb0 - 2 bits; the two bits join the rest, on top), XOR-ed with a hash of the other bits WITHOUT the two bits of the
//           last nucleotide: the four successors of a k-mer (src/branchingNode.c:16-112: its last k - 1 nucleotides + any fourth) share their
//           home line and differ in two stored key bits, and so do its four predecessors -- a branching query reads two lines, not eight.
//           hi' = perm(hi ^ mix(rest)) -- perm a fixed bijection that scatters: a bijection of hi for every rest --, split as (a | c), c the low t bits:
//           home = a m + floor(c m / 2^t)  (nl = 2^(hb - t) m home lines, m in [16, 32]: any table size within 6 %).
//           The home line thus KNOWS most of hi', and the slot stores only what it does not: q = c - ceil(sub 2^t / m) (qb bits) under the
//           rest bits -- 2k - 32 + qb key bits instead of 2k (k = 27, 100 genomes: 32 instead of 54; this quotienting is what lets a line
//           hold 8 k-mers of k = 27 or 31 where the table of round 3 held 5).  A k-mer displaced d lines from home stores d (3 bits at 6 slots
//           per line and more, up to 8 bits at one slot per line, k >= 97).
//   line    = header (16 bytes) + S slot bodies of wb = floor(48 / S) bytes, body s at byte 16 + s wb.  Header: S fields of
//           f = min(32, floor(128 / S) - 1) bits, field s at bits [s f, (s + 1) f) = the LOW f bits of slot s's stored key; bits
//           [128 - S, 128): slot s is in use.  Body: bits [0, CB) = colour-set id + 1 (CB = bits of the number of colour sets), [CB, CB + db)
//           = d, above them the other key bits.  S = the largest of 10..1 whose body holds all that.  A lookup that loads the HEADER
//           first (one 16-byte load) reads the body only of a slot whose field matches: an absent k-mer costs one load instruction on the
//           line, a stored one two (bft_kh_scan: branching, sequences, the walk); the presence kernel fetches whole lines by quads of lanes
//           (bft_kh.hip, k_query_kh).
// The layout is CANONICAL: the k-mers in (home line, T-form) order fill the lines by slot-level linear probing, so the table is a function
// of the stored set, the colour sets and the occupancy alone: the GPU build -- one device-wide sort by home line, one max-scan, one pass of
// atomic ORs -- and the sequential host restatement give the same bytes (tests/test_gpu_parity.py).  The table never changes an answer: it
// holds exactly the k-mers of the sorted table `tk` with their colour sets, `tk` can be rebuilt from it ("compact_table"), and the
// container walk is used whenever the table is absent ("kmer_hash" 0, allocation failure, an overflow list beyond 4096 k-mers)
// or rows are asked for.
#define BFT_KH_LINE_WORDS 8u
#define BFT_KH_MAX_SLOTS 10u
// displacement bits of a slot, by the slots of a line (the fewer slots, the longer the runs of full lines: one slot per line at 55 % makes
// runs of dozens): a k-mer that would land further from home than they hold goes to the OVERFLOW LIST instead
#define BFT_KH_DBITS_FOR(S) ((S) >= 6u ? 3u : ((S) >= 4u ? 4u : ((S) == 3u ? 5u : ((S) == 2u ? 6u : 8u))))
#define BFT_KH_TAIL_LINES 256u // lines behind the home lines: what the last home lines spill into (the largest displacement + 1)
#define BFT_KH_OVF_CAP 4096u   // k-mers the overflow list holds at most (more: no table at this occupancy)
struct BftKhGeo {
    uint32_t S, f, wb, cb;     // slots per line, bits of a header field, bytes of a slot body, value bits
    uint32_t db, maxd;         // displacement bits of a slot; the largest displacement in the table (a lookup looks no further)
    uint32_t kb, qb;           // bits of a stored key (rest bits + qb), bits of q
    uint32_t hb, restb;        // hashed bits (the top hb0 bits of the T-form without the first nucleotide's two), bits of the rest (2k - hb)
    uint32_t hb0, po;          // top bits of the T-form taken for hashing; where in them the first nucleotide sits (32: not in them -- k < 11)
    uint64_t mm;               // mask of the rest's low word for mixing: without the last nucleotide (and the first, when it lies there)
    uint32_t t, m, inv;        // home = a m + floor(c m / 2^t); inv = ceil(2^32 / m)
    uint64_t nl;               // home lines
};

// Node prefix hash (derived when the image is bound, optional): the prefix entries of every node BELOW the root in one hash table
// keyed by (node id, rotated prefix) -- 64-byte buckets of four {key, entry} pairs, sized for <= 1 key per bucket on average.  On a
// deep trie a level then costs one cache line (the bucket) instead of four dependent ones (node record, CC header, filter2 word,
// cluster entry).  A key that found its bucket full is simply not in the table: a lookup that meets a full bucket without its key
// takes the container path, so the table never changes an answer.  A bucket with a free slot and no such key means no CC of the
// node holds the prefix; that is "absent" outright when no node below the root holds UC rows (nph_no_uc), else the container
// path (-> the node's UC).
#define BFT_NPH_SLOTS 4
#define BFT_NPH_EMPTY (~0ull)

struct BftImage {
    int k, L, W;
    uint32_t nb_genomes;
    uint32_t debug_stop;      // read only by -DBFT_PERF_PROBE builds (libbft_gpu_probe.so, tools/perf_probe.py): truncates the
                              // walk after a stage; the shipped library compiles those checks out (BFT_DBG_STOP == 0)
    uint32_t probe_big;       // suffix-group search (bft_group_probe): 0 = 4-row blocks, step to the adjacent block;
                              // 1 = 8-row blocks, next guess re-interpolated (big groups); same answers either way
    uint64_t n_kmers;
    const uint32_t* hashmod;  // [16384] (hash_v[2i] % 1504) | (hash_v[2i+1] % 1504) << 16
    const BftNode* nodes;
    const uint8_t* bfT;
    const BftCC* ccs;
    const BftCCX* ccx;        // [n_ccs] extended headers (derived, see above)
    const uint64_t* f18;      // flat prefix bitmaps + ranks of the s = 4 CCs
    const uint64_t* fent;     // flat prefix entries of the s = 4 CCs
    const uint64_t* rdir;     // [2^18] root direct table (see BFT_RDIR_*), or NULL
    const uint32_t* rstart;   // [2^18 + 1] root range table (see BFT_RSTART_SPECIAL), or NULL; only with rdir
    const uint32_t* rq;       // [2^18] root quartile table (see BFT_RQ_*), or NULL; only with rstart
    const uint64_t* nph;      // node prefix hash (see BFT_NPH_*): (nph_mask + 1) buckets of 4 {key, entry}, or NULL
    uint64_t nph_mask;
    uint32_t nph_no_uc;       // 1: no node below the root holds UC rows
    const uint64_t* f2w;
    const uint64_t* clus;
    const uint64_t* child;
    const uint64_t* tk;       // [n_kmers * W] sorted T-form table
    const uint64_t* kh_lines; // [(kh.nl + BFT_KH_TAIL_LINES) * 8] k-mer hash (BFT_KH_*, above), or NULL
    BftKhGeo kh;
    const uint64_t* kh_ovf;   // [kh_ovf_n * W] its overflow list: sorted T-form k-mers, and
    const uint32_t* kh_ovf_val;  // their values
    uint32_t kh_ovf_n;
    const uint32_t* rspec;    // [2^18 / 32] one bit per root prefix: not a plain suffix group of the root (bit 31 of rstart), or NULL
    uint32_t walk_kh;         // 1: the container walk looks a plain root suffix group up in the k-mer hash (one line) instead of searching
                              // its rows of the sorted table -- when the caller wants presence or colour sets, not rows
    const uint32_t* tcol;     // [n_kmers] colour-set id per row
    uint32_t emit_cs;         // per launch: the query kernels write the colour set of a found k-mer where they otherwise write its row
    const uint64_t* uck;      // [n_uc_rows * W] node-UC rows (T-form)
    const uint32_t* ucrow;    // [n_uc_rows] row of that k-mer in tk
    const uint32_t* cs_off;   // [n_cs + 1] colour-set dictionary (sorted genome ids)
    const void* cs_ids;       // genome ids of all sets, cs_w bytes each (1 up to 256 genomes, 2 up to 65536, else 4): bft_cs_id()
    uint32_t cs_w;
};
