/*
 * oracle/bft_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the Bloom Filter Trie hot path of
 * GuillaumeHolley/BloomFilterTrie: k-mer insertion, presence query and
 * colour-set retrieval.  It is the checker for the HIP path and the "port"
 * CPU baseline of bench.py.  Nothing in bloomfiltertrie_amd/ may include,
 * link or call this file: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do.
 *
 * PARITY PINNING STATUS (read DESIGN.md "Oracle"):
 *   - the reference's trie code (presenceNode.c, insertNode.c, CC.c, UC.c,
 *     annotation.c) includes <Judy.h>, which this image lacks, and a
 *     stand-in header is not allowed, so the reference cannot be built here;
 *   - the reference ships no tests, fixtures or golden vectors for the path;
 *   - pinned: XXH64 + hash_v table (against the reference's own xxhash.c built
 *     into oracle/_ref, and the values recorded in SURVEY.md), the 2-bit
 *     codec (README.md:172 vector), the byte LUTs (reference popcnt.c built
 *     into oracle/_ref), annotation id sizes (reference log2.c in oracle/_ref);
 *   - trie-level behaviour (which container a k-mer lands in, query results)
 *     is "parity unpinned" against the reference binary; it is checked against
 *     the mathematical ground truth instead (the BFT is an exact set/colour
 *     index: presence == set membership, colours == set of inserting genomes).
 *
 * Each function cites the reference file:line it restates.
 */
#ifndef BFT_ORACLE_H
#define BFT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_bft orc_bft;

/* glibc rand() without srand(): the seeds every reference-built BFT carries
 * (include/CC.h:246-248). */
#define ORC_DEFAULT_R1 1804289383
#define ORC_DEFAULT_R2 846930886

/* XXH64 (xxHash 0.6.2 algorithm, restated from the published spec;
 * reference copy: src/xxhash.c). */
uint64_t orc_xxh64(const void *data, size_t len, uint64_t seed);

/* createBFT_Root(k, ...) include/CC.h:214-258.  k must be a multiple of 9,
 * 9 <= k <= 126 (src/main.c:61-63).  r1/r2 <= 0 selects the defaults. */
orc_bft *orc_create(int k, int r1, int r2);
void orc_free(orc_bft *t);

int orc_k(const orc_bft *t);
int orc_kmer_bytes(const orc_bft *t); /* CEIL(2k, 8) */

/* hash_v[2i], hash_v[2i+1] for i < 4^9 (include/Node.h:158-185). */
const uint64_t *orc_hash_v(const orc_bft *t);

/* insertKmers(root, array_kmers, nb_kmers, id_genome, size_id_genome)
 * src/insertNode.c:18-36.  kmers: n * CEIL(2k,8) bytes, layout of
 * parseKmerCount (src/fasta.c:3-53).  Genome ids must be inserted in
 * non-decreasing order per k-mer, as in the reference. */
int orc_insert_kmers(orc_bft *t, const uint8_t *kmers, long n, uint32_t id_genome);

/* Materialise the packed CC/UC arrays (BF_filter2, filter3, extra_filter3,
 * children_type, children) of include/CC.h:34-67 from the insertion model.
 * Called automatically by the query entry points when the trie is dirty. */
void orc_freeze(orc_bft *t);

/* Loop of src/file_io.c:726-768 around isKmerPresent (src/presenceNode.c:1823).
 * present_bits: CEIL(n,8) bytes, bit i%8 of byte i/8 (LSB first).
 * Returns the number of present k-mers. */
long orc_query_presence(orc_bft *t, const uint8_t *kmers, long n, uint8_t *present_bits);

/* Same, over nthreads disjoint slices sharing the read-only trie
 * (isKmerPresent is re-entrant: SURVEY.md 8b "Threading"). */
long orc_query_presence_mt(orc_bft *t, const uint8_t *kmers, long n, uint8_t *present_bits,
                           int nthreads);

/* Counting mode: also sums the trie bytes the restated algorithm dereferences
 * (SURVEY.md 8d "S").  bytes_out[0] = total bytes, [1] = CCs scanned,
 * [2] = levels visited. */
long orc_query_presence_count(orc_bft *t, const uint8_t *kmers, long n, uint8_t *present_bits,
                              uint64_t *bytes_out);

/* get_annotation + get_list_id_genomes (src/bft.c:363-387, 622-641;
 * src/annotation.c:2086-2250) for every k-mer of the batch.
 * offsets: n+1 entries; ids: capacity ids_cap.  Returns the total number of
 * ids (if > ids_cap nothing past ids_cap is written; call again). */
long orc_query_colors(orc_bft *t, const uint8_t *kmers, long n, uint8_t *present_bits,
                      uint64_t *offsets, uint32_t *ids, long ids_cap);

/* parseKmerCount (src/fasta.c:3-53): ASCII -> 2-bit.  out must be zeroed by
 * the caller.  Returns 1 if all k characters were ACGTU (any case), else 0
 * and the bytes written so far are cleared as the reference does. */
int orc_parse_kmer(const char *ascii, int k, uint8_t *out);
/* kmer_comp_to_ascii (src/fasta.c:55-83). */
void orc_kmer_to_ascii(const uint8_t *kmer, int k, char *out);

/* get_nb_bytes_power2_annot (include/log2.h:45-50). */
int orc_nb_bytes_id(uint32_t id);

/* Annotation codec (src/annotation.c:416-916 encode sizes/modes,
 * :2086-2250 decode).  encode: smallest of modes 0/1/2 for a sorted id
 * list; returns the size written (<= cap) or -1.  decode: returns n ids. */
int orc_annot_encode(const uint32_t *ids, int n, uint8_t *out, int cap);
int orc_annot_decode(const uint8_t *annot, int size, uint32_t *ids, int cap);

/* Trie shape (printMemory.c-style walk): out[0]=nodes, [1]=CCs,
 * [2]=distinct k-mers, [3]=root CCs, [4]=root UC rows, [5]=UC rows total,
 * [6]=child nodes, [7]=prefixes total, [8]=CCs in s=4 mode, [9]=max CCs/node. */
void orc_stats(orc_bft *t, long *out);
/* nb_elem of each root CC; returns the number of root CCs. */
int orc_root_cc_sizes(orc_bft *t, int *out, int cap);

/* iterate_over_kmers-style extraction (src/extract_kmers.c): writes every
 * stored k-mer (packed) and the id of its colour set; returns the count.
 * Pass NULLs to only count. */
long orc_extract(orc_bft *t, uint8_t *kmers_out, uint32_t *cs_out);
/* colour set by id: returns the number of genome ids. */
int orc_colorset(orc_bft *t, uint32_t cs, uint32_t *ids, int cap);

/* Loop of src/file_io.c:943-998 over isBranchingRight / isBranchingLeft (src/branchingNode.c:16-112, :240-340):
 * successors = present k-mers kmer[1..k-1]+N, predecessors = present N+kmer[0..k-2] (N in ACGT).  Restated at the
 * level of its definition -- four isKmerPresent calls per side -- not as the 4-way wildcard variants
 * presenceNeighborsRight/Left (src/presenceNode.c:15-1211), which compute the same counts in one walk.
 * counts (optional): (successors << 4) | predecessors.  Returns the number of branching k-mers. */
long orc_query_branching(orc_bft *t, const uint8_t *kmers, long n, uint8_t *branching_bits, uint8_t *counts);

/* query_sequence(bft, sequence, threshold, canonical_search) src/bft.c:1241-1351: genome ids that hold at least
 * ceil(nb_kmers * threshold) of the sequence's k-mers.  Returns the number of ids written (sorted). */
int orc_query_sequence(orc_bft *t, const char *sequence, double threshold, int canonical, uint32_t nb_genomes,
                       uint32_t *ids, int cap);

/* write_BFT_Root / read_BFT_Root (src/write_to_disk.c:21-258, :260-776): the .bft file format of
 * SURVEY.md A.6 (compressed == 0, no comp_set_colors, no extended annotations).  genome names are
 * "genome_<id>".  orc_load_bft returns NULL on a malformed file. */
int orc_write_bft(orc_bft *t, const char *path, int nb_genomes);
/* Test modes of the writer, to produce the shapes reference-built files have: comp_on = annotations stored as mode-3
 * indices into comp_set_colors (what compress_annotations_disk leaves, src/file_io.c:3-76; entries delta-coded by
 * comp_annotation, src/annotation.c:1777-1838); ext_on = the widest rows of each UC keep their last annotation byte
 * in the extended-annotation table (src/UC.c:321-521).  Call after the insertions, before orc_write_bft. */
void orc_set_annotation_modes(orc_bft *t, int comp_on, int ext_on);
orc_bft *orc_load_bft(const char *path);
int orc_nb_genomes_loaded(const orc_bft *t);

#ifdef __cplusplus
}
#endif
#endif
