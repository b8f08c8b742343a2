/*
 * ref_loop_program.c -- the two loops of the reference's own harness, written against the names that harness uses and
 * nothing else: the build loop of insert_Genomes_from_KmerFiles (src/file_io.c:89-213: add_genomes_BFT_Root,
 * get_nb_bytes_power2_annot, parseKmerCount into a 4096-byte buffer, insertKmers per full buffer and for the tail) and
 * the presence loop of queryBFT_kmerPresences_from_KmerFiles (src/file_io.c:651-895: parseKmerCount, isKmerPresent on
 * &(root->node) at the root level, is_kmer_in_cdbg, get_annotation, get_list_id_genomes, one CSV row per input line,
 * free(res)).  tests/test_ref_api.py links it with -lbft and compares the CSV with the oracle.
 *
 *   ref_loop_program k queries.txt out.csv genome_file...
 */
#include <bft/bft.h>
#include <libgen.h>
#include <stdlib.h>
#include <string.h>

#define SIZE_BUFFER 4096 /* include/default_param.h of the reference: the harness works in 4096-byte k-mer buffers */

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s k queries.txt out.csv genome_file...\n", argv[0]); return 2; }
    const int size_kmer = atoi(argv[1]);
    BFT_Root* root = create_cdbg(size_kmer, 0);
    const int nb_bytes_kmer = (size_kmer * 2 + 7) / 8;
    const int nb_kmer_in_buf = SIZE_BUFFER / nb_bytes_kmer;
    uint8_t* array_kmers = calloc(SIZE_BUFFER, sizeof(uint8_t));
    char* line = calloc(100, sizeof(char));
    if (array_kmers == NULL || line == NULL) return 2;

    /* ---- build: src/file_io.c:116-185 ---- */
    for (int i = 4; i < argc; i++) {
        int j = 0, k = 0;
        char* dup = strdup(argv[i]);
        char* str_tmp = basename(dup);
        add_genomes_BFT_Root(1, &str_tmp, root);
        free(dup);
        const int size_id_genome = get_nb_bytes_power2_annot((uint32_t)root->nb_genomes - 1);
        FILE* file = fopen(argv[i], "r");
        if (file == NULL) { fprintf(stderr, "cannot open %s\n", argv[i]); return 2; }
        while (fgets(line, 100, file) != NULL) {
            if (parseKmerCount(line, root->k, array_kmers, k) == 1) {
                k += nb_bytes_kmer;
                j++;
                if (j == nb_kmer_in_buf) {
                    insertKmers(root, array_kmers, nb_kmer_in_buf, (uint32_t)root->nb_genomes - 1, size_id_genome);
                    j = 0;
                    k = 0;
                    memset(array_kmers, 0, SIZE_BUFFER * sizeof(uint8_t));
                }
            }
        }
        insertKmers(root, array_kmers, j, (uint32_t)root->nb_genomes - 1, size_id_genome);
        memset(array_kmers, 0, SIZE_BUFFER * sizeof(uint8_t));
        fclose(file);
    }

    /* ---- presence CSV: src/file_io.c:700-895 (text queries) ---- */
    const int lvl_root = root->k / 9 - 1; /* NB_CHAR_SUF_PREF = 9 */
    const char csv_sep = ',', not_present = '0', present = '1', nl = '\n';
    FILE* file_query = fopen(argv[2], "r");
    FILE* file_output = fopen(argv[3], "w");
    if (file_query == NULL || file_output == NULL) { fprintf(stderr, "cannot open the query / output file\n"); return 2; }
    char* csv_line_res = malloc((size_t)root->nb_genomes * 2);
    if (csv_line_res == NULL) return 2;
    int i = 0;
    for (; i < root->nb_genomes - 1; i++) {
        fwrite(root->filenames[i], sizeof(char), strlen(root->filenames[i]), file_output);
        fwrite(&csv_sep, sizeof(char), 1, file_output);
        csv_line_res[i * 2 + 1] = csv_sep;
    }
    csv_line_res[root->nb_genomes * 2 - 1] = nl;
    fwrite(root->filenames[i], sizeof(char), strlen(root->filenames[i]), file_output);
    fwrite(&nl, sizeof(char), 1, file_output);

    BFT_kmer* bft_kmer = create_empty_kmer();
    uint64_t nb_kmers_present = 0;
    char* buffer_queries = NULL;
    size_t size_buffer_queries = 0;
    while (getline(&buffer_queries, &size_buffer_queries, file_query) != -1) {
        buffer_queries[strcspn(buffer_queries, "\r\n")] = '\0';
        memset(array_kmers, 0, (size_t)nb_bytes_kmer);
        int it_csv_line_res = 0;
        const int ok = strlen(buffer_queries) >= (size_t)root->k && parseKmerCount(buffer_queries, root->k, array_kmers, 0) == 1;
        bft_kmer->res = ok ? isKmerPresent(&(root->node), root, lvl_root, array_kmers, root->k) : NULL;
        if (ok && is_kmer_in_cdbg(bft_kmer)) {
            nb_kmers_present++;
            BFT_annotation* bft_annot = get_annotation(bft_kmer);
            uint32_t* ids_present = get_list_id_genomes(bft_annot, root);
            free_BFT_annotation(bft_annot);
            for (uint32_t it_annot = 1; it_annot <= ids_present[0]; it_annot++) {
                for (uint32_t z = 0; z < ids_present[it_annot] - (it_annot == 1 ? 0 : ids_present[it_annot - 1] + 1); z++, it_csv_line_res += 2)
                    csv_line_res[it_csv_line_res] = not_present;
                csv_line_res[it_csv_line_res] = present;
                it_csv_line_res += 2;
            }
            for (uint32_t it_annot = ids_present[ids_present[0]] + 1; it_annot < (uint32_t)root->nb_genomes; it_annot++, it_csv_line_res += 2)
                csv_line_res[it_csv_line_res] = not_present;
            free(ids_present);
        } else { /* absent, or a line with a character outside ACGT: an all-zero row (src/file_io.c:844-850) */
            for (int it_annot = 0; it_annot < root->nb_genomes; it_annot++, it_csv_line_res += 2) csv_line_res[it_csv_line_res] = not_present;
        }
        fwrite(csv_line_res, sizeof(char), (size_t)root->nb_genomes * 2, file_output);
        free(bft_kmer->res);
        bft_kmer->res = NULL;
    }
    free(buffer_queries);
    fclose(file_query);
    fclose(file_output);
    printf("Nb k-mers present = %llu\n", (unsigned long long)nb_kmers_present);
    free(csv_line_res);
    free(bft_kmer);
    free(array_kmers);
    free(line);
    free_cdbg(root);
    return 0;
}
