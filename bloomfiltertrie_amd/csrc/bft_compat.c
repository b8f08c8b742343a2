/*
 * bft_compat.c -- host-side C layer that serves the reference's public API (<bft/bft.h>, include/bft/bft.h here) from
 * the C-ABI of include/bft_gpu.h.  Built as libbft.so so that a program of the reference links with `-lbft` as
 * README.md:91-111 says.  Nothing is computed here: every lookup, colour-set decode, neighbour test and sequence query
 * is a (small) batch handed to libbft_gpu.so; this file only converts between the reference's objects (BFT_kmer,
 * BFT_annotation, uint32_t id lists with the count in [0]) and the batch layouts, and keeps the reference's
 * exit-on-error behaviour (ERROR(), include/useful_macros.h:33-43).
 */
#define _GNU_SOURCE
#include <libgen.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/bft/bft.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); exit(EXIT_FAILURE); } while (0)
#define NOT_NULL(p, where) do { if ((p) == NULL) DIE("%s: NULL pointer\n", where); } while (0)
#define ABSENT 0xFFFFFFFFu

static void ck(int rc, const char* where) {
    if (rc != BFT_GPU_OK) DIE("%s: %s\n", where, bft_gpu_last_error());
}

static int bytes_of(int k) { return (2 * k + 7) / 8; }

/* parseKmerCount (src/fasta.c:3-53): 1 if the first k characters are all in ACGTU (either case) */
static int pack_kmer(const char* s, int k, uint8_t* out) {
    memset(out, 0, (size_t)bytes_of(k));
    for (int j = 0; j < k; j++) {
        unsigned c;
        switch (s[j]) {
        case 'a': case 'A': c = 0; break;
        case 'c': case 'C': c = 1; break;
        case 'g': case 'G': c = 2; break;
        case 't': case 'T': case 'u': case 'U': c = 3; break;
        default: memset(out, 0, (size_t)bytes_of(k)); return 0;
        }
        out[j >> 2] |= (uint8_t)(c << (2 * (j & 3)));
    }
    return 1;
}

static void unpack_kmer(const uint8_t* in, int k, char* s) { /* kmer_comp_to_ascii, src/fasta.c:55-87 */
    for (int j = 0; j < k; j++) s[j] = "ACGT"[(in[j >> 2] >> (2 * (j & 3))) & 3];
    s[k] = '\0';
}

/* ---------------------------------------------------------------- graph */

static int device_from_env(void) {
    const char* d = getenv("BFT_GPU_DEVICE");
    return d ? atoi(d) : 0;
}

static BFT* new_root(int k, int treshold_compression, bft_gpu* g) {
    BFT* bft = calloc(1, sizeof(BFT));
    NOT_NULL(bft, "createBFT_Root()");
    bft->k = k;
    bft->treshold_compression = treshold_compression;
    bft->gpu = g;
    return bft;
}

static void push_name(BFT* bft, const char* name) { /* add_genomes_BFT_Root, include/CC.h:307-338 */
    bft->filenames = realloc(bft->filenames, (size_t)(bft->nb_genomes + 1) * sizeof(char*));
    NOT_NULL(bft->filenames, "add_genomes_BFT_Root()");
    bft->filenames[bft->nb_genomes] = strdup(name);
    NOT_NULL(bft->filenames[bft->nb_genomes], "add_genomes_BFT_Root()");
    bft->nb_genomes++;
}

BFT* create_cdbg(int k, int treshold_compression) {
    if (k <= 0 || k % 9 != 0) DIE("create_cdbg(): k must be a positive multiple of 9.\n"); /* src/main.c:61-63 */
    bft_gpu* g = NULL;
    ck(bft_gpu_create(k, device_from_env(), &g), "create_cdbg()");
    return new_root(k, treshold_compression, g);
}

void free_cdbg(BFT* bft) {
    if (bft == NULL) return;
    for (int i = 0; i < bft->nb_genomes; i++) free(bft->filenames[i]);
    free(bft->filenames);
    bft_gpu_free(bft->gpu);
    free(bft);
}

bft_gpu* bft_device_index(BFT* bft) {
    NOT_NULL(bft, "bft_device_index()");
    return bft->gpu;
}

static uint32_t new_genome(BFT* bft, const char* name) {
    uint32_t gid = 0;
    ck(bft_gpu_add_genome(bft->gpu, name, &gid), "add_genomes_BFT_Root()");
    push_name(bft, name);
    return gid;
}

static void insert_strings(int nb_kmers, char** kmers, uint32_t gid, BFT* bft, const char* where) {
    if (nb_kmers <= 0) return;
    NOT_NULL(kmers, where);
    const int nb = bytes_of(bft->k);
    uint8_t* batch = malloc((size_t)nb_kmers * (size_t)nb);
    NOT_NULL(batch, where);
    for (int i = 0; i < nb_kmers; i++)
        if (kmers[i] == NULL || strlen(kmers[i]) < (size_t)bft->k || !pack_kmer(kmers[i], bft->k, batch + (size_t)i * nb))
            DIE("%s: could not insert k-mer in graph, it probably contains unvalid characters.\n", where); /* src/bft.c:67 */
    ck(bft_gpu_insert_kmers(bft->gpu, batch, (uint64_t)nb_kmers, gid), where);
    free(batch);
}

void insert_kmers_new_genome(int nb_kmers, char** kmers, char* genome_name, BFT* bft) {
    NOT_NULL(bft, "insert_kmers_new_genome()");
    NOT_NULL(genome_name, "insert_kmers_new_genome()");
    insert_strings(nb_kmers, kmers, new_genome(bft, genome_name), bft, "insert_kmers_new_genome()");
}

void insert_kmers_last_genome(int nb_kmers, char** kmers, BFT* bft) {
    NOT_NULL(bft, "insert_kmers_last_genome()");
    if (bft->nb_genomes <= 0) DIE("insert_kmers_last_genome(): the graph is empty, there is no last genome.\n");
    insert_strings(nb_kmers, kmers, (uint32_t)bft->nb_genomes - 1, bft, "insert_kmers_last_genome()");
}

/* insert_Genomes_from_KmerFiles with binary_files = 0 (src/bft.c:31-37, src/file_io.c:89-213): lines that are not a
 * k-mer are skipped; the whole file goes to the GPU as one batch. */
void insert_genomes_from_files(int nb_files, char** paths, BFT* bft, char* prefix_bft_filename) {
    (void)prefix_bft_filename; /* only used by the reference's colour compression */
    NOT_NULL(bft, "insert_genomes_from_files()");
    if (nb_files > 0) NOT_NULL(paths, "insert_genomes_from_files()");
    const int nb = bytes_of(bft->k);
    for (int i = 0; i < nb_files; i++) {
        NOT_NULL(paths[i], "insert_genomes_from_files()");
        char* tmp = strdup(paths[i]);
        const uint32_t gid = new_genome(bft, basename(tmp));
        free(tmp);
        FILE* f = fopen(paths[i], "r");
        if (f == NULL) DIE("insert_Genomes_from_KmerFiles(): cannot open %s\n", paths[i]);
        size_t cap = 1 << 16, n = 0;
        uint8_t* batch = malloc(cap * (size_t)nb);
        NOT_NULL(batch, "insert_genomes_from_files()");
        char* line = NULL;
        size_t lcap = 0;
        while (getline(&line, &lcap, f) != -1) {
            if (n == cap) {
                cap *= 2;
                batch = realloc(batch, cap * (size_t)nb);
                NOT_NULL(batch, "insert_genomes_from_files()");
            }
            if (strlen(line) >= (size_t)bft->k && pack_kmer(line, bft->k, batch + n * (size_t)nb)) n++;
        }
        free(line);
        fclose(f);
        ck(bft_gpu_insert_kmers(bft->gpu, batch, (uint64_t)n, gid), "insert_genomes_from_files()");
        free(batch);
    }
    ck(bft_gpu_build(bft->gpu), "insert_genomes_from_files()");
}

/* ---------------------------------------------------------------- k-mers */

static resultPresence* new_res(BFT* bft, int present, uint32_t row, uint32_t colorset) {
    resultPresence* r = malloc(sizeof(resultPresence));
    NOT_NULL(r, "create_resultPresence()");
    r->link_child = present ? (void*)bft : NULL;
    r->bft = bft;
    r->row = present ? row : ABSENT;
    r->colorset = present ? colorset : ABSENT;
    return r;
}

static void fill_kmer(BFT_kmer* km, const char* ascii, int k) {
    km->kmer = malloc((size_t)k + 1);
    km->kmer_comp = malloc((size_t)bytes_of(k));
    if (km->kmer == NULL || km->kmer_comp == NULL) DIE("create_kmer(): out of memory\n");
    memcpy(km->kmer, ascii, (size_t)k);
    km->kmer[k] = '\0';
    km->res = NULL;
}

BFT_kmer* create_kmer(const char* kmer, int k) {
    NOT_NULL(kmer, "create_kmer()");
    if (strlen(kmer) != (size_t)k) DIE("create_kmer(): k-mer length is not the one used in the graph.\n");
    BFT_kmer* km = malloc(sizeof(BFT_kmer));
    NOT_NULL(km, "create_kmer()");
    fill_kmer(km, kmer, k);
    if (!pack_kmer(km->kmer, k, km->kmer_comp)) DIE("create_kmer(): Unexpected character encountered in k-mer.\n");
    km->res = new_res(NULL, 0, ABSENT, ABSENT);
    return km;
}

BFT_kmer* create_empty_kmer(void) {
    BFT_kmer* km = malloc(sizeof(BFT_kmer));
    NOT_NULL(km, "create_empty_kmer()");
    km->kmer = NULL;
    km->kmer_comp = NULL;
    km->res = NULL;
    return km;
}

void free_BFT_kmer_content(BFT_kmer* bft_kmer, int nb_bft_kmer) {
    NOT_NULL(bft_kmer, "free_BFT_kmer_content()");
    for (int i = 0; i < nb_bft_kmer; i++) {
        free(bft_kmer[i].kmer);
        free(bft_kmer[i].kmer_comp);
        free(bft_kmer[i].res);
    }
}

void free_BFT_kmer(BFT_kmer* bft_kmer, int nb_bft_kmer) {
    NOT_NULL(bft_kmer, "free_BFT_kmer()");
    free_BFT_kmer_content(bft_kmer, nb_bft_kmer);
    free(bft_kmer);
}

/* One batch of packed k-mers -> their resultPresence (isKmerPresent for each, src/presenceNode.c:1823-1921). */
static void locate(BFT* bft, const uint8_t* packed, int n, BFT_kmer* out, const char* where) {
    uint8_t bits[8] = {0};
    uint32_t rows[64], sets[64];
    ck(bft_gpu_query_rows(bft->gpu, packed, (uint64_t)n, bits, rows, sets), where);
    for (int i = 0; i < n; i++) out[i].res = new_res(bft, (bits[i >> 3] >> (i & 7)) & 1, rows[i], sets[i]);
}

BFT_kmer* get_kmer(const char* kmer, BFT* bft) {
    NOT_NULL(kmer, "get_kmer()");
    NOT_NULL(bft, "get_kmer()");
    if (strlen(kmer) < (size_t)bft->k) DIE("get_kmer(): Unexpected character encountered in k-mer.\n");
    BFT_kmer* km = malloc(sizeof(BFT_kmer));
    NOT_NULL(km, "get_kmer()");
    fill_kmer(km, kmer, bft->k);
    if (!pack_kmer(km->kmer, bft->k, km->kmer_comp)) DIE("get_kmer(): Unexpected character encountered in k-mer.\n");
    locate(bft, km->kmer_comp, 1, km, "get_kmer()");
    return km;
}

bool is_kmer_in_cdbg(BFT_kmer* bft_kmer) {
    NOT_NULL(bft_kmer, "is_kmer_in_cdbg()");
    NOT_NULL(bft_kmer->res, "is_kmer_in_cdbg()");
    return bft_kmer->res->link_child != NULL;
}

/* ---------------------------------------------------------------- the harness seam (src/file_io.c loops) */

int parseKmerCount(const char* line, int size_kmer, uint8_t* tab, int pos_tab) { /* src/fasta.c:3-53 */
    NOT_NULL(line, "parseKmerCount()");
    NOT_NULL(tab, "parseKmerCount()");
    uint8_t* t = tab + pos_tab;
    int j = 0;
    for (; j < size_kmer; j++) {
        unsigned c;
        switch (line[j]) {
        case 'a': case 'A': c = 0; break;
        case 'c': case 'C': c = 1; break;
        case 'g': case 'G': c = 2; break;
        case 't': case 'T': case 'u': case 'U': c = 3; break;
        default: /* IUPAC codes, end of line, anything else: the bytes touched so far are cleared (:49) */
            memset(t, 0, (size_t)((j + 1) / 4));
            return 0;
        }
        t[j >> 2] |= (uint8_t)(c << (2 * (j & 3)));
    }
    return 1;
}

void kmer_comp_to_ascii(const uint8_t* kmer_comp, int k, char* kmer) {
    NOT_NULL(kmer_comp, "kmer_comp_to_ascii()");
    NOT_NULL(kmer, "kmer_comp_to_ascii()");
    unpack_kmer(kmer_comp, k, kmer);
}

int get_nb_bytes_power2_annot(uint32_t pos) { /* include/log2.h:45-50: CEIL(bits needed for pos, 6), 1 for pos = 0 */
    const int bits = pos ? 32 - __builtin_clz(pos) : 1;
    return (bits + 5) / 6;
}

void add_genomes_BFT_Root(int nb_files, char** filenames, BFT_Root* root) { /* include/CC.h:307-338 */
    NOT_NULL(root, "add_genomes_BFT_Root()");
    if (nb_files < 0) DIE("add_genomes_BFT_Root(): the number of genomes to insert cannot be less than 0.\n");
    if (nb_files > 0) NOT_NULL(filenames, "add_genomes_BFT_Root()");
    for (int i = 0; i < nb_files; i++) {
        NOT_NULL(filenames[i], "add_genomes_BFT_Root()");
        new_genome(root, filenames[i]);
    }
}

void insertKmers(BFT_Root* root, uint8_t* array_kmers, int nb_kmers, uint32_t id_genome, int size_id_genome) {
    (void)size_id_genome; /* width of the id inside the reference's annotation bytes: the colour sets here are id lists */
    NOT_NULL(root, "insertKmers()");
    if (nb_kmers <= 0) return;
    NOT_NULL(array_kmers, "insertKmers()");
    ck(bft_gpu_insert_kmers(root->gpu, array_kmers, (uint64_t)nb_kmers, id_genome), "insertKmers()");
}

resultPresence* isKmerPresent(Node* node, BFT_Root* root, int lvl_node, uint8_t* kmer, int size_kmer) {
    NOT_NULL(root, "isKmerPresent()");
    NOT_NULL(kmer, "isKmerPresent()");
    if ((node != NULL && node != &root->node) || size_kmer != root->k || lvl_node != root->k / 9 - 1)
        DIE("isKmerPresent(): only whole k-mers from the root vertex (&root->node, level k/9-1, size k) can be looked up.\n");
    BFT_kmer tmp;
    locate(root, kmer, 1, &tmp, "isKmerPresent()");
    return tmp.res;
}

/* ---------------------------------------------------------------- annotations */

BFT_annotation* create_BFT_annotation(void) {
    BFT_annotation* a = malloc(sizeof(BFT_annotation));
    NOT_NULL(a, "create_BFT_annotation()");
    a->annot = a->annot_ext = a->annot_cplx = NULL;
    a->size_annot = a->size_annot_cplx = -1;
    a->from_BFT = 0;
    return a;
}

void free_BFT_annotation(BFT_annotation* bft_annot) {
    NOT_NULL(bft_annot, "free_BFT_annotation()");
    /* the reference's from_BFT annotations alias the trie; here the bytes are always a private copy */
    free(bft_annot->annot);
    free(bft_annot->annot_ext);
    free(bft_annot->annot_cplx);
    free(bft_annot);
}

BFT_annotation* get_annotation(BFT_kmer* bft_kmer) {
    NOT_NULL(bft_kmer, "get_annotation()");
    if (!is_kmer_in_cdbg(bft_kmer)) DIE("get_annotation(): k-mer is not present in the graph.\n");
    BFT* bft = bft_kmer->res->bft;
    uint32_t n = 0;
    ck(bft_gpu_colorset_annot(bft->gpu, bft_kmer->res->colorset, NULL, 0, &n), "get_annotation()");
    BFT_annotation* a = create_BFT_annotation();
    a->annot = malloc(n ? n : 1);
    NOT_NULL(a->annot, "get_annotation()");
    ck(bft_gpu_colorset_annot(bft->gpu, bft_kmer->res->colorset, a->annot, n, &n), "get_annotation()");
    a->size_annot = (int)n;
    a->from_BFT = 1;
    return a;
}

/* get_id_genomes_from_annot, modes 0/1/2 (src/annotation.c:2086-2250): the byte codec of one annotation object, the
 * inverse of what bft_gpu_colorset_annot produced.  ids may be NULL to count only. */
static uint32_t decode_annot(const BFT_annotation* a, uint32_t* ids) {
    const uint8_t* b = a->annot;
    const int size = a->size_annot;
    uint32_t n = 0;
    if (b == NULL || size <= 0) return 0;
    const int mode = b[0] & 3;
    int i = 0;
    if (mode == 0) {
        for (int bit = 2; bit < size * 8; bit++)
            if (b[bit >> 3] & (1u << (bit & 7))) { if (ids) ids[n] = (uint32_t)bit - 2; n++; }
    } else if (mode == 1) { /* inclusive ranges: start byte flag 1, continuation flag 2 */
        while (i < size && (b[i] & 1)) {
            uint32_t lo = b[i++] >> 2, hi;
            while (i < size && (b[i] & 2)) lo = (lo << 6) | (b[i++] >> 2);
            if (i >= size || !(b[i] & 1)) break;
            hi = b[i++] >> 2;
            while (i < size && (b[i] & 2)) hi = (hi << 6) | (b[i++] >> 2);
            for (uint32_t v = lo; v <= hi; v++) { if (ids) ids[n] = v; n++; }
        }
    } else if (mode == 2) { /* id list: start byte flag 2, continuation flag 1 */
        while (i < size && (b[i] & 2)) {
            uint32_t v = b[i++] >> 2;
            while (i < size && (b[i] & 1)) v = (v << 6) | (b[i++] >> 2);
            if (ids) ids[n] = v;
            n++;
        }
    } else
        DIE("get_id_genomes_from_annot(): compressed annotations (mode 3) are not produced by this library.\n");
    return n;
}

uint32_t* get_list_id_genomes(BFT_annotation* bft_annot, BFT* bft) {
    NOT_NULL(bft_annot, "get_list_id_genomes()");
    NOT_NULL(bft, "get_list_id_genomes()");
    const uint32_t n = decode_annot(bft_annot, NULL);
    uint32_t* ids = malloc(((size_t)n + 1) * sizeof(uint32_t));
    NOT_NULL(ids, "get_list_id_genomes()");
    ids[0] = n;
    decode_annot(bft_annot, ids + 1);
    return ids;
}

uint32_t get_count_id_genomes(BFT_annotation* bft_annot, BFT* bft) {
    NOT_NULL(bft_annot, "get_count_id_genomes()");
    NOT_NULL(bft, "get_count_id_genomes()");
    return decode_annot(bft_annot, NULL);
}

bool presence_genome(uint32_t id_genome, BFT_annotation* bft_annot, BFT* bft) {
    NOT_NULL(bft_annot, "is_genome_present()");
    NOT_NULL(bft, "is_genome_present()");
    if (id_genome >= (uint32_t)bft->nb_genomes) return false;
    uint32_t* ids = get_list_id_genomes(bft_annot, bft);
    bool found = false;
    for (uint32_t i = 1; i <= ids[0] && !found; i++) found = ids[i] == id_genome;
    free(ids);
    return found;
}

uint32_t* intersection_list_id_genomes(uint32_t* list_a, uint32_t* list_b) { /* src/bft.c:659-688 */
    NOT_NULL(list_a, "intersection_list_id_genomes()");
    NOT_NULL(list_b, "intersection_list_id_genomes()");
    const uint32_t na = list_a[0], nb = list_b[0];
    uint32_t* out = malloc(((size_t)(na < nb ? na : nb) + 1) * sizeof(uint32_t));
    NOT_NULL(out, "intersection_list_id_genomes()");
    uint32_t i = 1, j = 1, n = 0;
    while (i <= na && j <= nb) {
        if (list_a[i] < list_b[j]) i++;
        else if (list_b[j] < list_a[i]) j++;
        else { out[++n] = list_a[i]; i++; j++; }
    }
    out[0] = n;
    return out;
}

/* ---------------------------------------------------------------- sequence query */

uint32_t* query_sequence(BFT* bft, char* sequence, double threshold, bool canonical_search) {
    NOT_NULL(bft, "query_sequence()");
    NOT_NULL(sequence, "query_sequence()");
    if (threshold <= 0) DIE("query_sequence(): the threshold must be superior to 0.\n");
    if (threshold > 1) DIE("query_sequence(): the threshold must be inferior or equal to 1.\n");
    const size_t len = strlen(sequence);
    if (len < (size_t)bft->k) printf("query_sequence(): query %s is too small and must be at least of length k.\n", sequence);
    const uint32_t G = (uint32_t)bft->nb_genomes, rowbytes = (G + 7) / 8;
    uint8_t* row = calloc(rowbytes ? rowbytes : 1, 1);
    NOT_NULL(row, "query_sequence()");
    const uint64_t off[2] = {0, (uint64_t)len};
    if (G) ck(bft_gpu_query_sequences(bft->gpu, sequence, off, 1, threshold, canonical_search ? 1 : 0, row), "query_sequence()");
    uint32_t n = 0;
    for (uint32_t g = 0; g < G; g++) n += (row[g >> 3] >> (g & 7)) & 1;
    uint32_t* ids = malloc(((size_t)n + 1) * sizeof(uint32_t));
    NOT_NULL(ids, "query_sequence()");
    ids[0] = n;
    for (uint32_t g = 0, j = 0; g < G; g++)
        if ((row[g >> 3] >> (g & 7)) & 1) ids[++j] = g;
    free(row);
    return ids;
}

/* ---------------------------------------------------------------- neighbours */

void set_neighbors_traversal(BFT* bft) { NOT_NULL(bft, "set_neighbors_traversal()"); }
void unset_neighbors_traversal(BFT* bft) { NOT_NULL(bft, "unset_neighbors_traversal()"); }

/* side 0: N + kmer[0..k-2] (predecessors), side 1: kmer[1..k-1] + N (successors); N = A, C, G, T */
static void neighbours_of(const BFT_kmer* km, BFT* bft, int side, BFT_kmer* out, uint8_t* packed) {
    const int k = bft->k, nb = bytes_of(k);
    for (int i = 0; i < 4; i++) {
        out[i].kmer = malloc((size_t)k + 1);
        out[i].kmer_comp = malloc((size_t)nb);
        if (out[i].kmer == NULL || out[i].kmer_comp == NULL) DIE("get_neighbors(): out of memory\n");
        if (side == 0) {
            out[i].kmer[0] = "ACGT"[i];
            memcpy(out[i].kmer + 1, km->kmer, (size_t)k - 1);
        } else {
            memcpy(out[i].kmer, km->kmer + 1, (size_t)k - 1);
            out[i].kmer[k - 1] = "ACGT"[i];
        }
        out[i].kmer[k] = '\0';
        pack_kmer(out[i].kmer, k, out[i].kmer_comp);
        memcpy(packed + (size_t)i * nb, out[i].kmer_comp, (size_t)nb);
    }
}

static BFT_kmer* neighbours(BFT_kmer* km, BFT* bft, int first_side, int n_sides, const char* where) {
    NOT_NULL(km, where);
    NOT_NULL(bft, where);
    if (!is_kmer_in_cdbg(km)) DIE("%s: k-mer is not present in the graph.\n", where);
    const int nb = bytes_of(bft->k), n = 4 * n_sides;
    BFT_kmer* out = malloc((size_t)n * sizeof(BFT_kmer));
    uint8_t* packed = malloc((size_t)n * (size_t)nb);
    if (out == NULL || packed == NULL) DIE("%s: out of memory\n", where);
    for (int s = 0; s < n_sides; s++) neighbours_of(km, bft, first_side + s, out + 4 * s, packed + (size_t)(4 * s) * nb);
    locate(bft, packed, n, out, where);
    free(packed);
    return out;
}

BFT_kmer* get_neighbors(BFT_kmer* bft_kmer, BFT* bft) { return neighbours(bft_kmer, bft, 0, 2, "get_neighbors()"); }
BFT_kmer* get_predecessors(BFT_kmer* bft_kmer, BFT* bft) { return neighbours(bft_kmer, bft, 0, 1, "get_predecessors()"); }
BFT_kmer* get_successors(BFT_kmer* bft_kmer, BFT* bft) { return neighbours(bft_kmer, bft, 1, 1, "get_successors()"); }

/* ---------------------------------------------------------------- iteration, extraction */

void v_iterate_over_kmers(BFT* bft, BFT_func_ptr f, va_list args) {
    NOT_NULL(bft, "v_iterate_over_kmers()");
    NOT_NULL(f, "v_iterate_over_kmers()");
    uint64_t n = 0;
    ck(bft_gpu_extract(bft->gpu, NULL, NULL, 0, &n), "iterate_over_kmers()");
    if (n == 0) return;
    const int k = bft->k, nb = bytes_of(k);
    uint8_t* packed = malloc((size_t)n * (size_t)nb);
    uint32_t* sets = malloc((size_t)n * sizeof(uint32_t));
    if (packed == NULL || sets == NULL) DIE("iterate_over_kmers(): out of memory\n");
    ck(bft_gpu_extract(bft->gpu, packed, sets, n, &n), "iterate_over_kmers()");
    BFT_kmer* km = create_empty_kmer();
    km->kmer = malloc((size_t)k + 1);
    km->kmer_comp = malloc((size_t)nb);
    km->res = new_res(bft, 1, 0, 0);
    if (km->kmer == NULL || km->kmer_comp == NULL) DIE("iterate_over_kmers(): out of memory\n");
    for (uint64_t i = 0; i < n; i++) {
        memcpy(km->kmer_comp, packed + i * (size_t)nb, (size_t)nb);
        unpack_kmer(km->kmer_comp, k, km->kmer);
        km->res->row = (uint32_t)i;
        km->res->colorset = sets[i];
        va_list copy; /* f consumes its arguments with va_arg on every call (src/extract_kmers.c does the same) */
        va_copy(copy, args);
        const size_t go_on = f(km, bft, copy);
        va_end(copy);
        if (go_on == 0) break;
    }
    free_BFT_kmer(km, 1);
    free(packed);
    free(sets);
}

void iterate_over_kmers(BFT* bft, BFT_func_ptr f, ...) {
    va_list args;
    va_start(args, f);
    v_iterate_over_kmers(bft, f, args);
    va_end(args);
}

size_t write_kmer_ascii_to_disk(BFT_kmer* bft_kmer, BFT* bft, va_list args) { /* src/bft.c:299-308 */
    FILE* file = va_arg(args, FILE*);
    bft_kmer->kmer[bft->k] = '\n';
    fwrite(bft_kmer->kmer, sizeof(char), (size_t)bft->k + 1, file);
    bft_kmer->kmer[bft->k] = '\0';
    return 1;
}

size_t write_kmer_comp_to_disk(BFT_kmer* bft_kmer, BFT* bft, va_list args) { /* src/bft.c:316-324 */
    (void)bft;
    const int nb_bytes_kmer_comp = va_arg(args, int);
    FILE* file = va_arg(args, FILE*);
    fwrite(bft_kmer->kmer_comp, sizeof(uint8_t), (size_t)nb_bytes_kmer_comp, file);
    return 1;
}

void extract_kmers_to_disk(BFT* bft, char* filename_output, bool compressed_output) { /* src/bft.c:255-290 */
    NOT_NULL(bft, "extract_kmers_to_disk()");
    NOT_NULL(filename_output, "extract_kmers_to_disk()");
    FILE* f = fopen(filename_output, "w");
    if (f == NULL) DIE("extract_kmers_to_disk(): failed to create/open output file.\n");
    if (compressed_output) {
        uint64_t n = 0;
        ck(bft_gpu_extract(bft->gpu, NULL, NULL, 0, &n), "extract_kmers_to_disk()");
        fprintf(f, "%d\n%llu\n", bft->k, (unsigned long long)n);
        iterate_over_kmers(bft, write_kmer_comp_to_disk, bytes_of(bft->k), f);
    } else
        iterate_over_kmers(bft, write_kmer_ascii_to_disk, f);
    fclose(f);
}

/* ---------------------------------------------------------------- disk */

void write_BFT(BFT* bft, char* filename, bool compress_annotations) {
    (void)compress_annotations;
    NOT_NULL(bft, "write_BFT()");
    NOT_NULL(filename, "write_BFT()");
    ck(bft_gpu_write_bft(bft->gpu, filename), "write_BFT()");
}

BFT* load_BFT(char* filename) {
    NOT_NULL(filename, "load_BFT()");
    bft_gpu* g = NULL;
    ck(bft_gpu_load_bft(filename, device_from_env(), &g), "load_BFT()");
    uint64_t info[16];
    ck(bft_gpu_info(g, info, 16), "load_BFT()");
    BFT* bft = new_root((int)info[0], 0, g);
    char name[4096];
    for (uint64_t i = 0; i < info[11]; i++) {
        ck(bft_gpu_genome_name(g, (uint32_t)i, name, sizeof name), "load_BFT()");
        push_name(bft, name);
    }
    return bft;
}
