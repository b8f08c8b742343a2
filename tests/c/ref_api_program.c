/*
 * ref_api_program.c -- a program written the way a user of GuillaumeHolley/BloomFilterTrie writes one: it includes
 * <bft/bft.h>, links -lbft and uses nothing but the reference's documented API (include/bft.h of the reference;
 * doc/doxygen "Graph functions", "K-mer functions", "Annotation functions", "Traversal", "Disk").  The test
 * tests/test_ref_api.py compiles it against include/bft/bft.h + libbft.so of this repository, runs it on the GPU box
 * and compares every printed line with the oracle and with ground truth.
 *
 *   ref_api_program k out.bft queries.txt seqs.txt extracted.txt genome_file... last_genome_file
 *
 * All genome files but the last go through insert_genomes_from_files; the last one is read here and inserted through
 * insert_kmers_new_genome (first half) and insert_kmers_last_genome (second half).
 */
#include <bft/bft.h>
#include <stdlib.h>
#include <string.h>

static char** read_lines(const char* path, int* n_out) {
    FILE* f = fopen(path, "r");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    int cap = 1024, n = 0;
    char** lines = malloc((size_t)cap * sizeof(char*));
    char* line = NULL;
    size_t lcap = 0;
    while (getline(&line, &lcap, f) != -1) {
        line[strcspn(line, "\r\n")] = 0;
        if (n == cap) { cap *= 2; lines = realloc(lines, (size_t)cap * sizeof(char*)); }
        lines[n++] = strdup(line);
    }
    free(line);
    fclose(f);
    *n_out = n;
    return lines;
}

static void print_ids(const uint32_t* ids) {
    for (uint32_t j = 1; j <= ids[0]; j++) printf("%s%u", j > 1 ? "," : "", ids[j]);
}

/* BFT_func_ptr: counts the k-mers and sums the sizes of their colour sets; stops after `limit` k-mers if limit > 0 */
static size_t count_kmers(BFT_kmer* km, BFT* bft, va_list args) {
    uint64_t* n = va_arg(args, uint64_t*);
    uint64_t* colours = va_arg(args, uint64_t*);
    const uint64_t limit = va_arg(args, uint64_t);
    BFT_annotation* a = get_annotation(km);
    *colours += get_count_id_genomes(a, bft);
    free_BFT_annotation(a);
    (*n)++;
    return (limit && *n >= limit) ? 0 : 1;
}

static void query_all(BFT* bft, char** q, int nq, const char* tag) {
    int nb_present = 0;
    for (int i = 0; i < nq; i++) {
        BFT_kmer* km = get_kmer(q[i], bft);
        if (is_kmer_in_cdbg(km)) {
            nb_present++;
            BFT_annotation* a = get_annotation(km);
            uint32_t* ids = get_list_id_genomes(a, bft);
            if (get_count_id_genomes(a, bft) != ids[0]) { printf("BAD count\n"); exit(3); }
            for (uint32_t g = 0; g < (uint32_t)bft->nb_genomes + 2; g++) { /* presence_genome agrees with the list */
                bool in_list = false;
                for (uint32_t j = 1; j <= ids[0]; j++) in_list |= ids[j] == g;
                if (presence_genome(g, a, bft) != in_list) { printf("BAD presence_genome\n"); exit(3); }
            }
            printf("%s %s 1 ", tag, q[i]);
            print_ids(ids);
            printf("\n");
            free(ids);
            free_BFT_annotation(a);
        } else
            printf("%s %s 0\n", tag, q[i]);
        free_BFT_kmer(km, 1);
    }
    printf("%s present %d\n", tag, nb_present);
}

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage\n"); return 2; }
    const int k = atoi(argv[1]);
    const char* out_bft = argv[2];
    int nq = 0, ns = 0, nl = 0;
    char** q = read_lines(argv[3], &nq);
    char** seqs = read_lines(argv[4], &ns);
    const char* extracted = argv[5];
    const int nb_files = argc - 7;

    BFT* bft = create_cdbg(k, 0);
    insert_genomes_from_files(nb_files, &argv[6], bft, NULL);
    char** last = read_lines(argv[argc - 1], &nl);
    insert_kmers_new_genome(nl / 2, last, "the_last_genome", bft);
    insert_kmers_last_genome(nl - nl / 2, last + nl / 2, bft);

    printf("GENOMES %d", bft->nb_genomes);
    for (int g = 0; g < bft->nb_genomes; g++) printf(" %s", bft->filenames[g]);
    printf("\n");

    query_all(bft, q, nq, "Q");

    /* neighbours of the first present queries */
    set_neighbors_traversal(bft);
    for (int i = 0, done = 0; i < nq && done < 300; i++) {
        BFT_kmer* km = get_kmer(q[i], bft);
        if (is_kmer_in_cdbg(km)) {
            BFT_kmer* pred = get_predecessors(km, bft);
            BFT_kmer* succ = get_successors(km, bft);
            BFT_kmer* both = get_neighbors(km, bft);
            printf("N %s ", q[i]);
            for (int j = 0; j < 4; j++) printf("%d", is_kmer_in_cdbg(&pred[j]) ? 1 : 0);
            printf(" ");
            for (int j = 0; j < 4; j++) printf("%d", is_kmer_in_cdbg(&succ[j]) ? 1 : 0);
            printf("\n");
            for (int j = 0; j < 4; j++) {
                if (is_kmer_in_cdbg(&both[j]) != is_kmer_in_cdbg(&pred[j]) || strcmp(both[j].kmer, pred[j].kmer)) { printf("BAD neighbors\n"); exit(3); }
                if (is_kmer_in_cdbg(&both[4 + j]) != is_kmer_in_cdbg(&succ[j]) || strcmp(both[4 + j].kmer, succ[j].kmer)) { printf("BAD neighbors\n"); exit(3); }
            }
            if (pred[1].kmer[0] != 'C' || succ[2].kmer[k - 1] != 'G') { printf("BAD neighbor order\n"); exit(3); }
            free_BFT_kmer(pred, 4);
            free_BFT_kmer(succ, 4);
            free_BFT_kmer(both, 8);
            done++;
        }
        free_BFT_kmer(km, 1);
    }
    unset_neighbors_traversal(bft);

    /* sequence queries */
    for (int i = 0; i < ns; i++) {
        uint32_t* a = query_sequence(bft, seqs[i], 0.7, false);
        uint32_t* b = query_sequence(bft, seqs[i], 1.0, true);
        uint32_t* c = intersection_list_id_genomes(a, b);
        printf("S %d ", i); print_ids(a); printf(" | "); print_ids(b); printf(" | "); print_ids(c); printf("\n");
        free(a); free(b); free(c);
    }

    /* iteration */
    uint64_t n_kmers = 0, n_colours = 0;
    iterate_over_kmers(bft, count_kmers, &n_kmers, &n_colours, (uint64_t)0);
    printf("ITER %llu %llu\n", (unsigned long long)n_kmers, (unsigned long long)n_colours);
    uint64_t n_some = 0, c_some = 0;
    iterate_over_kmers(bft, count_kmers, &n_some, &c_some, (uint64_t)10);
    printf("ITER_STOP %llu\n", (unsigned long long)n_some);
    extract_kmers_to_disk(bft, (char*)extracted, false);

    /* disk round trip */
    write_BFT(bft, (char*)out_bft, false);
    free_cdbg(bft);
    bft = load_BFT((char*)out_bft);
    printf("RELOADED %d %d", bft->k, bft->nb_genomes);
    for (int g = 0; g < bft->nb_genomes; g++) printf(" %s", bft->filenames[g]);
    printf("\n");
    query_all(bft, q, nq, "R");
    free_cdbg(bft);

    /* objects that never touch an index */
    BFT_kmer* loose = create_kmer(q[0], k);
    printf("LOOSE %s %d\n", loose->kmer, is_kmer_in_cdbg(loose) ? 1 : 0);
    free_BFT_kmer(loose, 1);
    BFT_kmer* empty = create_empty_kmer();
    free_BFT_kmer(empty, 1);
    BFT_annotation* ea = create_BFT_annotation();
    free_BFT_annotation(ea);
    return 0;
}
