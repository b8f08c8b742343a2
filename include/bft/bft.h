/*
 * <bft/bft.h> -- the public C API of GuillaumeHolley/BloomFilterTrie for the build / load / query path, served by the
 * MI355X library (libbft_gpu.so through the small C layer bloomfiltertrie_amd/csrc/bft_compat.c -> libbft.so).
 *
 * A program written against the reference's header (`#include <bft/bft.h>`, link `-lbft`, README.md:91-111) recompiles
 * against this one unchanged as long as it stays on the functions below: same names, same argument meaning, same
 * ownership rules (returned objects are malloc'd, the caller frees them with the same free_* functions), same error
 * behaviour (a message on stderr and exit(EXIT_FAILURE), include/useful_macros.h:33-43).  Each declaration cites the
 * reference declaration it stands for (include/bft.h) and the reference definition (src/bft.c).
 *
 * What differs, by construction:
 *   - the index lives in GPU memory.  BFT_Root keeps the documented public fields (filenames, k, nb_genomes,
 *     treshold_compression) and a handle instead of the host trie; resultPresence holds indexes instead of host
 *     pointers (its link_child is still non-NULL exactly when the k-mer is stored, which is what is_kmer_in_cdbg
 *     tests, src/bft.c:246-248);
 *   - every call is one (small) GPU batch, about 24 us; loops over many k-mers should use the batched entry points of
 *     bft_gpu.h on bft_device_index(bft) -- that is the point of the GPU path;
 *   - insertions are collected on the GPU and the containers are rebuilt in bulk by the first query after them;
 *   - there is no CPU fallback: without a usable GPU every function reports the error and exits.
 * Not provided (outside the path): marking / flags, annotation set operations, prefix_matching,
 * create_cdbg_from_bft_kmers, add_id_genomes, colour compression (write_BFT ignores compress_annotations and writes
 * uncompressed annotations, which the reference loads).
 */
#ifndef BFT_GPU_COMPAT_BFT_H
#define BFT_GPU_COMPAT_BFT_H

#include <stdarg.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "../bft_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

struct BFT_Root;

/* A trie vertex (reference: include/Node.h:55-58, {CC_array, UC_array}).  The index lives in HBM, so a Node is an opaque
 * placeholder here: the only Node a caller can name is &root->node, which is what the reference's harness passes to
 * isKmerPresent (src/file_io.c:732,810). */
typedef struct {
    void* CC_array; /* always NULL */
    uint64_t reserved[2];
} Node;

/* Location of a looked-up k-mer (reference: include/Node.h:60-92, host pointers and positions inside the trie). */
typedef struct {
    void* link_child;      /* non-NULL iff the k-mer is stored (the test of is_kmer_in_cdbg, src/bft.c:246-248) */
    struct BFT_Root* bft;  /* the index the k-mer was looked up in */
    uint32_t row;          /* its position in the stored k-mer table (order of iterate_over_kmers here) */
    uint32_t colorset;     /* id of its colour set in the index's dictionary */
} resultPresence;

/* Root of a BFT (reference: include/Node.h:96-122).  filenames, k, nb_genomes and treshold_compression are the fields
 * the reference documents as public; the others mirror its header values.  `gpu` replaces the host trie. */
typedef struct BFT_Root {
    char** filenames; /* names of the inserted genomes, nb_genomes of them */
    int k;            /* k-mer length */
    int r1;
    int r2;
    int nb_genomes;
    int treshold_compression;
    uint8_t compressed; /* always 0, as the reference's CLI and create_cdbg set it */
    uint8_t marked;     /* always 0 here (marking is not provided) */
    bft_gpu* gpu;       /* the index, resident in HBM */
    Node node;          /* last member as in the reference (include/Node.h:121): the root vertex handed to isKmerPresent */
} BFT_Root;

typedef BFT_Root BFT; /* include/bft.h:29 */

/* include/Node.h:129-133 */
typedef struct {
    char* kmer;          /* ASCII, NUL-terminated */
    uint8_t* kmer_comp;  /* 2 bits per nucleotide, parseKmerCount layout (src/fasta.c:3-53) */
    resultPresence* res;
} BFT_kmer;

/* include/bft.h:34-42.  annot holds the colour set in the reference's own byte encoding (modes 0/1/2,
 * src/annotation.c:2086-2250), size_annot its length; annot_ext / annot_cplx are never used here. */
typedef struct {
    uint8_t* annot;
    uint8_t* annot_ext;
    uint8_t* annot_cplx;
    int size_annot;
    int size_annot_cplx;
    uint8_t from_BFT;
} BFT_annotation;

/* include/bft.h:51: called on every k-mer by iterate_over_kmers; returning 0 stops the iteration. */
typedef size_t (*BFT_func_ptr)(BFT_kmer* bft_kmer, BFT* bft, va_list args);

/* ---- graph (include/bft.h:62-73, src/bft.c:12-118) ---- */
BFT* create_cdbg(int k, int treshold_compression);
void free_cdbg(BFT* bft);
/* k-mer files: one ASCII k-mer per line, optionally followed by a count; one genome per file, named by its basename */
void insert_genomes_from_files(int nb_files, char** paths, BFT* bft, char* prefix_bft_filename);
void insert_kmers_new_genome(int nb_kmers, char** kmers, char* genome_name, BFT* bft);
void insert_kmers_last_genome(int nb_kmers, char** kmers, BFT* bft);

/* ---- the per-k-mer seam the reference's own harness calls directly, not through bft.h: the loops of
 * src/file_io.c:726-768 / :810 (presence CSV) and :146,166,181 (build) relink against these unchanged ---- */
/* include/presenceNode.h:57, src/presenceNode.c:1823-1921.  `node` must be &root->node (or NULL), lvl_node the root
 * level k/9-1 and size_kmer == root->k: the whole-k-mer lookup, which is the only way the harness calls it.  The result
 * is malloc'd and the caller frees it (src/file_io.c:767). */
resultPresence* isKmerPresent(Node* node, BFT_Root* root, int lvl_node, uint8_t* kmer, int size_kmer);
/* include/insertNode.h:26, src/insertNode.c:18-36: nb_kmers packed k-mers (contiguous, parseKmerCount layout) of genome
 * id_genome; size_id_genome (= get_nb_bytes_power2_annot(id_genome)) is accepted and not needed. */
void insertKmers(BFT_Root* root, uint8_t* array_kmers, int nb_kmers, uint32_t id_genome, int size_id_genome);
/* include/CC.h:307-338: appends nb_files genome names (copied) to root->filenames, nb_genomes += nb_files */
void add_genomes_BFT_Root(int nb_files, char** filenames, BFT_Root* root);
/* src/fasta.c:3-53: ORs the 2-bit codes of the first size_kmer characters of line into tab[pos_tab..] (the caller
 * pre-zeroes); 1 when all of them are in ACGTU (either case), else 0 with the bytes written so far cleared. */
int parseKmerCount(const char* line, int size_kmer, uint8_t* tab, int pos_tab);
/* src/fasta.c:55-87 */
void kmer_comp_to_ascii(const uint8_t* kmer_comp, int k, char* kmer);
/* include/log2.h:45-50: bytes an annotation needs to hold genome id `pos` in its list encodings */
int get_nb_bytes_power2_annot(uint32_t pos);

/* ---- k-mers (include/bft.h:81-87, :125-126; src/bft.c:125-340) ---- */
BFT_kmer* create_kmer(const char* kmer, int k);
BFT_kmer* create_empty_kmer(void);
void free_BFT_kmer(BFT_kmer* bft_kmer, int nb_bft_kmer);
void free_BFT_kmer_content(BFT_kmer* bft_kmer, int nb_bft_kmer);
BFT_kmer* get_kmer(const char* kmer, BFT* bft);
bool is_kmer_in_cdbg(BFT_kmer* bft_kmer);
void extract_kmers_to_disk(BFT* bft, char* filename_output, bool compressed_output);
size_t write_kmer_ascii_to_disk(BFT_kmer* bft_kmer, BFT* bft, va_list args);
size_t write_kmer_comp_to_disk(BFT_kmer* bft_kmer, BFT* bft, va_list args);

/* ---- annotations = colour sets (include/bft.h:95-98, :115-117; src/bft.c:326-420, :622-688) ---- */
BFT_annotation* create_BFT_annotation(void);
void free_BFT_annotation(BFT_annotation* bft_annot);
BFT_annotation* get_annotation(BFT_kmer* bft_kmer);
bool presence_genome(uint32_t id_genome, BFT_annotation* bft_annot, BFT* bft);
uint32_t* get_list_id_genomes(BFT_annotation* bft_annot, BFT* bft); /* [0] = count, then the sorted ids */
uint32_t get_count_id_genomes(BFT_annotation* bft_annot, BFT* bft);
uint32_t* intersection_list_id_genomes(uint32_t* list_a, uint32_t* list_b);

/* ---- sequence query (include/bft.h:127, src/bft.c:1241-1351) ---- */
uint32_t* query_sequence(BFT* bft, char* sequence, double threshold, bool canonical_search);

/* ---- neighbours (include/bft.h:154-158, src/bft.c:795-1003).  Arrays of 8 / 4 / 4 BFT_kmer in A,C,G,T order
 * (predecessors first in get_neighbors); is_kmer_in_cdbg tells which exist.  set_/unset_neighbors_traversal only
 * prepare caches in the reference and are no-ops here. ---- */
void set_neighbors_traversal(BFT* bft);
void unset_neighbors_traversal(BFT* bft);
BFT_kmer* get_neighbors(BFT_kmer* bft_kmer, BFT* bft);
BFT_kmer* get_predecessors(BFT_kmer* bft_kmer, BFT* bft);
BFT_kmer* get_successors(BFT_kmer* bft_kmer, BFT* bft);

/* ---- iteration (include/bft.h:166-167, src/bft.c:1016-1085).  Same set of k-mers as the reference; the order is
 * the index's (ascending in its internal key), not the reference's container order. ---- */
void iterate_over_kmers(BFT* bft, BFT_func_ptr f, ...);
void v_iterate_over_kmers(BFT* bft, BFT_func_ptr f, va_list args);

/* ---- disk (include/bft.h:175-176, src/bft.c:1090-1110, src/write_to_disk.c) ---- */
void write_BFT(BFT* bft, char* filename, bool compress_annotations);
BFT* load_BFT(char* filename);

/* ---- additive: the handle behind a BFT, for the batched calls of bft_gpu.h ---- */
bft_gpu* bft_device_index(BFT* bft);

#ifdef __cplusplus
}
#endif
#endif
