/*
 * bft_gpu_cli.c -- host-side C harness over the C-ABI of include/bft_gpu.h, with the command line of the reference's
 * `bft` binary for the hot path (src/main.c:40-47, :180-316):
 *
 *   bft_gpu build k {kmers|kmers_comp} list_genome_files output_file
 *   bft_gpu load file_bft [-add_genomes {kmers|kmers_comp} list_genome_files output_file]
 *                         [-query_kmers {kmers|kmers_comp} list_kmer_files]
 *                         [-query_branching {kmers|kmers_comp} list_kmer_files]
 *                         [-query_sequences threshold {canonical|non_canonical} list_sequence_files]
 *                         [-extract_kmers {kmers|kmers_comp} output_file]
 *
 * It is the per-k-mer loops of src/file_io.c:89-213 (build), :651-895 (presence CSV) and :897-1020 (branching)
 * rewired to one batched GPU call per file; file reading, ASCII parsing (parseKmerCount, src/fasta.c:3-53) and CSV
 * formatting stay host C exactly as in the reference, so the outputs are byte-identical:
 *   - CSV name = basename(query file) with its extension replaced by ".csv", in the current directory;
 *   - line 1 = genome names joined by ','; one "0,1,..." line per input line (an all-0 line for a line that is not
 *     a valid k-mer, src/file_io.c:844-850); the final '\n' is overwritten by '\0' (src/file_io.c:873-876);
 *   - stdout: "Nb k-mers present = <n>" (src/main.c:266), "Nb branching k-mers = <n>" (src/main.c:312).
 */
#define _GNU_SOURCE
#include <libgen.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/bft_gpu.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); exit(EXIT_FAILURE); } while (0) /* ERROR(), include/useful_macros.h:33-43 */
static void ck(int rc) { if (rc) DIE("%s\n", bft_gpu_last_error()); }

/* parseKmerCount, src/fasta.c:3-53 */
static int parse_kmer(const char* line, int k, uint8_t* out) {
    for (int j = 0; j < k; j++) {
        uint8_t c;
        switch (line[j]) {
        case 'a': case 'A': c = 0; break;
        case 'c': case 'C': c = 1; break;
        case 'g': case 'G': c = 2; break;
        case 'u': case 'U': case 't': case 'T': c = 3; break;
        default: memset(out, 0, (size_t)((j + 1) / 4)); return 0;
        }
        out[j / 4] |= (uint8_t)(c << (2 * (j % 4)));
    }
    return 1;
}

typedef struct { uint8_t* kmers; uint8_t* valid; uint64_t n_lines, n_kmers; } batch;

/* One k-mer file -> packed batch.  ASCII: one k-mer per line (invalid lines are remembered, they still get a CSV row).
 * kmers_comp: two header lines (k, count) then packed k-mers (src/file_io.c:134-156). */
static batch read_kmer_file(const char* path, int k, int binary) {
    const int nb = (2 * k + 7) / 8;
    batch b = {NULL, NULL, 0, 0};
    FILE* f = fopen(path, "r");
    if (!f) DIE("Cannot open %s\n", path);
    size_t cap = 1 << 16;
    b.kmers = calloc(cap, (size_t)nb);
    b.valid = calloc(cap, 1);
    if (binary) {
        char line[100];
        if (!fgets(line, 100, f) || !fgets(line, 100, f)) DIE("Cannot read header of the file\n");
        for (;;) {
            if (b.n_lines == cap) { cap *= 2; b.kmers = realloc(b.kmers, cap * nb); b.valid = realloc(b.valid, cap); }
            if (fread(b.kmers + b.n_lines * nb, (size_t)nb, 1, f) != 1) break;
            b.valid[b.n_lines++] = 1;
        }
    } else {
        char* line = NULL;
        size_t lcap = 0;
        while (getline(&line, &lcap, f) != -1) {
            if (b.n_lines == cap) {
                b.kmers = realloc(b.kmers, 2 * cap * nb);
                memset(b.kmers + cap * nb, 0, cap * nb);
                b.valid = realloc(b.valid, 2 * cap);
                cap *= 2;
            }
            memset(b.kmers + b.n_lines * nb, 0, (size_t)nb);
            b.valid[b.n_lines] = (uint8_t)(strlen(line) >= (size_t)k && parse_kmer(line, k, b.kmers + b.n_lines * nb));
            b.n_lines++;
        }
        free(line);
    }
    fclose(f);
    for (uint64_t i = 0; i < b.n_lines; i++) b.n_kmers += b.valid[i];
    return b;
}

/* BFT_GPU_DEVICES="0,1,2,...": the -query_kmers / -query_branching batches are sharded over these GPUs of the node, the index replicated
 * on each (bft_gpu_group_*; the reference's loops, src/file_io.c:651-895 and :897-1020, have one BFT_Root).  Unset: the one GPU of `h`. */
static bft_gpu_group* group_from_env(bft_gpu* h) {
    const char* e = getenv("BFT_GPU_DEVICES");
    if (!e || !*e) return NULL;
    int devs[64], n = 0;
    for (const char* p = e; *p && n < 64;) {
        devs[n++] = atoi(p);
        while (*p && *p != ',') p++;
        if (*p == ',') p++;
    }
    if (n < 2 && !(n == 1 && devs[0] != 0)) return NULL;
    bft_gpu_group* g = NULL;
    ck(bft_gpu_group_create(h, 0, devs, n, &g));
    return g;
}

static void query_kmers(bft_gpu* h, const char* path, int k, int binary, uint32_t nb_genomes, char** names) {
    batch b = read_kmer_file(path, k, binary);
    const uint32_t rowbytes = (nb_genomes + 7) / 8;
    uint8_t* present = calloc((b.n_lines + 7) / 8 + 1, 1);
    uint8_t* rows = calloc(b.n_lines ? b.n_lines : 1, rowbytes ? rowbytes : 1);
    bft_gpu_group* grp = group_from_env(h);
    if (grp) {
        ck(bft_gpu_group_query_color_rows(grp, b.kmers, b.n_lines, present, rows));
        bft_gpu_group_free(grp);
    } else
        ck(bft_gpu_query_color_rows(h, b.kmers, b.n_lines, present, rows)); /* invalid lines are all-zero k-mers: masked below */
    char* tmp = strdup(path);
    char* base = basename(tmp);
    char* outname = malloc(strlen(base) + 5);
    strcpy(outname, base);
    char* dot = strrchr(outname, '.');
    if (dot) strcpy(dot, ".csv"); else strcat(outname, ".csv");
    FILE* out = fopen(outname, "w");
    if (!out) DIE("Cannot create %s\n", outname);
    printf("\nQuerying BFT for k-mers in %s\n\n", path);
    for (uint32_t g = 0; g < nb_genomes; g++) fprintf(out, "%s%c", names[g], g + 1 < nb_genomes ? ',' : '\n');
    int nb_present = 0;
    char* line = malloc((size_t)nb_genomes * 2 + 1);
    for (uint64_t i = 0; i < b.n_lines; i++) {
        const int ok = b.valid[i] && ((present[i >> 3] >> (i & 7)) & 1);
        nb_present += ok;
        for (uint32_t g = 0; g < nb_genomes; g++) {
            line[2 * g] = (ok && ((rows[i * rowbytes + (g >> 3)] >> (g & 7)) & 1)) ? '1' : '0';
            line[2 * g + 1] = g + 1 < nb_genomes ? ',' : '\n';
        }
        fwrite(line, 1, (size_t)nb_genomes * 2, out);
    }
    fseek(out, -1L, SEEK_CUR); /* src/file_io.c:873-876 */
    fputc('\0', out);
    fclose(out);
    printf("\nNb k-mers present = %d\n", nb_present);
    free(line); free(outname); free(tmp); free(present); free(rows); free(b.kmers); free(b.valid);
}

static void query_branching(bft_gpu* h, const char* path, int k, int binary) {
    batch b = read_kmer_file(path, k, binary);
    const int nb = (2 * k + 7) / 8;
    /* only valid k-mers are queried (src/file_io.c:963: parseKmerCount(...) == 1) */
    uint64_t m = 0;
    for (uint64_t i = 0; i < b.n_lines; i++)
        if (b.valid[i]) { if (m != i) memmove(b.kmers + m * nb, b.kmers + i * nb, (size_t)nb); m++; }
    uint8_t* bits = calloc((m + 7) / 8 + 1, 1);
    printf("\nQuerying BFT for branching k-mers in %s\n\n", path);
    bft_gpu_group* grp = group_from_env(h);
    if (grp) {
        ck(bft_gpu_group_query_branching(grp, b.kmers, m, bits, NULL));
        bft_gpu_group_free(grp);
    } else
        ck(bft_gpu_query_branching(h, b.kmers, m, bits, NULL));
    int count = 0;
    for (uint64_t i = 0; i < m; i++) count += (bits[i >> 3] >> (i & 7)) & 1;
    printf("\nNb branching k-mers = %d\n", count);
    free(bits); free(b.kmers); free(b.valid);
}

/* query_sequences_outputCSV (src/file_io.c:1464-1574): one sequence per line, one CSV row per line */
static void query_sequences(bft_gpu* h, const char* path, double threshold, int canonical, uint32_t nb_genomes, char** names) {
    FILE* f = fopen(path, "r");
    if (!f) DIE("Cannot open %s\n", path);
    char* blob = NULL;
    size_t blen = 0, bcap = 0, ns = 0, ocap = 1024;
    uint64_t* off = malloc(ocap * sizeof(uint64_t));
    off[0] = 0;
    char* line = NULL;
    size_t lcap = 0;
    while (getline(&line, &lcap, f) != -1) {
        line[strcspn(line, "\r\n")] = 0;
        const size_t l = strlen(line);
        if (blen + l + 1 > bcap) { bcap = (blen + l + 1) * 2; blob = realloc(blob, bcap); }
        memcpy(blob + blen, line, l);
        blen += l;
        if (ns + 2 > ocap) { ocap *= 2; off = realloc(off, ocap * sizeof(uint64_t)); }
        off[++ns] = blen;
    }
    free(line);
    fclose(f);
    if (!blob) blob = calloc(1, 1);
    const uint32_t rowbytes = (nb_genomes + 7) / 8;
    uint8_t* rows = calloc(ns ? ns : 1, rowbytes ? rowbytes : 1);
    ck(bft_gpu_query_sequences(h, blob, off, ns, threshold, canonical, rows));
    char* tmp = strdup(path);
    char* base = basename(tmp);
    char* outname = malloc(strlen(base) + 5);
    strcpy(outname, base);
    char* dot = strrchr(outname, '.');
    if (dot) strcpy(dot, ".csv"); else strcat(outname, ".csv");
    FILE* out = fopen(outname, "w");
    if (!out) DIE("Cannot create %s\n", outname);
    for (uint32_t g = 0; g < nb_genomes; g++) fprintf(out, "%s%c", names[g], g + 1 < nb_genomes ? ',' : '\n');
    char* row = malloc((size_t)nb_genomes * 2 + 1);
    for (size_t i = 0; i < ns; i++) {
        for (uint32_t g = 0; g < nb_genomes; g++) {
            row[2 * g] = ((rows[i * rowbytes + (g >> 3)] >> (g & 7)) & 1) ? '1' : '0';
            row[2 * g + 1] = g + 1 < nb_genomes ? ',' : '\n';
        }
        fwrite(row, 1, (size_t)nb_genomes * 2, out);
    }
    fseek(out, -1L, SEEK_CUR);
    fputc('\0', out);
    fclose(out);
    printf("\nFile %s has been processed.\n", path);
    free(row); free(outname); free(tmp); free(rows); free(off); free(blob);
}

/* insert_Genomes_from_KmerFiles (src/file_io.c:89-213): one genome per listed file, ids in file order */
static void insert_genomes(bft_gpu* h, const char* list_path, int k, int binary) {
    char buffer[2048];
    FILE* lst = fopen(list_path, "r");
    if (!lst) DIE("Invalid list_genome_files.\n");
    while (fgets(buffer, sizeof buffer, lst)) {
        buffer[strcspn(buffer, "\r\n")] = 0;
        if (!buffer[0]) continue;
        uint32_t gid;
        char* tmp = strdup(buffer);
        ck(bft_gpu_add_genome(h, basename(tmp), &gid));
        free(tmp);
        printf("\nFile %u: %s\n\n", gid, buffer);
        batch b = read_kmer_file(buffer, k, binary);
        const int nb = (2 * k + 7) / 8;
        uint64_t m = 0; /* invalid lines are skipped on insertion (src/file_io.c:159) */
        for (uint64_t q = 0; q < b.n_lines; q++)
            if (b.valid[q]) { if (m != q) memmove(b.kmers + m * nb, b.kmers + q * nb, (size_t)nb); m++; }
        ck(bft_gpu_insert_kmers(h, b.kmers, m, gid));
        free(b.kmers); free(b.valid);
    }
    fclose(lst);
    ck(bft_gpu_build(h));
}

/* extract_kmers_to_disk (src/bft.c:255-290): every stored k-mer, ASCII one per line or "k\ncount\n" + packed k-mers.
 * The reference writes them in its trie iteration order; the set is the same, the order here is the image's. */
static void extract_kmers(bft_gpu* h, const char* path, int k, int compressed) {
    uint64_t n = 0;
    ck(bft_gpu_extract(h, NULL, NULL, 0, &n));
    const int nb = (2 * k + 7) / 8;
    uint8_t* km = malloc(n ? n * nb : 1);
    ck(bft_gpu_extract(h, km, NULL, n, &n));
    FILE* f = fopen(path, "w");
    if (!f) DIE("extract_kmers_to_disk(): failed to create/open output file.\n");
    if (compressed) {
        fprintf(f, "%d\n%llu\n", k, (unsigned long long)n);
        fwrite(km, (size_t)nb, n, f);
    } else {
        char* line = malloc((size_t)k + 2);
        for (uint64_t i = 0; i < n; i++) {
            for (int j = 0; j < k; j++) line[j] = "ACGT"[(km[i * nb + j / 4] >> (2 * (j % 4))) & 3];
            line[k] = '\n';
            fwrite(line, 1, (size_t)k + 1, f);
        }
        free(line);
    }
    fclose(f);
    free(km);
}

int main(int argc, char** argv) {
    if (argc >= 2 && (strcmp(argv[1], "--version") == 0 || strcmp(argv[1], "-v") == 0)) { /* src/main.c:51-55 */
        fprintf(stderr, "0.8 (%s)\n", bft_gpu_version());
        return 0;
    }
    if (argc < 3)
        DIE("\nUsage:\n"
            "bft_gpu build k {kmers|kmers_comp} list_genome_files output_file\n"
            "bft_gpu load file_bft [-add_genomes {kmers|kmers_comp} list_genome_files output_file] [Options]\n\nOptions:\n"
            "[-query_kmers {kmers|kmers_comp} list_kmer_files]\n"
            "[-query_branching {kmers|kmers_comp} list_kmer_files]\n"
            "[-query_sequences threshold {canonical|non_canonical} list_sequence_files]\n"
            "[-extract_kmers {kmers|kmers_comp} output_file]\n\n");
    bft_gpu* h = NULL;
    int k = 0, i = 0;
    char buffer[2048];
    if (strcmp(argv[1], "build") == 0) {
        if (argc < 6) DIE("bft_gpu build k {kmers|kmers_comp} list_genome_files output_file\n");
        k = atoi(argv[2]);
        if (k <= 0) DIE("Provided length k (for k-mers) is either <= 0 or not a number.\n");
        if (k % 9 != 0) DIE("Length k (for k-mers) must be a multiple of 9.\n"); /* src/main.c:63 */
        const int binary = strcmp(argv[3], "kmers_comp") == 0;
        ck(bft_gpu_create(k, 0, &h));
        insert_genomes(h, argv[4], k, binary);
        ck(bft_gpu_write_bft(h, argv[5]));
        i = 6;
    } else if (strcmp(argv[1], "load") == 0) {
        ck(bft_gpu_load_bft(argv[2], 0, &h));
        i = 3;
        if (i + 3 < argc && strcmp(argv[i], "-add_genomes") == 0) { /* src/main.c:217-246 */
            uint64_t inf[16];
            ck(bft_gpu_info(h, inf, 16));
            insert_genomes(h, argv[i + 2], (int)inf[0], strcmp(argv[i + 1], "kmers_comp") == 0);
            ck(bft_gpu_write_bft(h, argv[i + 3]));
            i += 4;
        }
    } else
        DIE("Unrecognized command %s.\n", argv[1]);

    uint64_t info[16];
    ck(bft_gpu_info(h, info, 16));
    k = (int)info[0];
    const uint32_t nb_genomes = (uint32_t)info[11];
    char** names = calloc(nb_genomes ? nb_genomes : 1, sizeof(char*));
    for (uint32_t g = 0; g < nb_genomes; g++) {
        names[g] = malloc(4096);
        ck(bft_gpu_genome_name(h, g, names[g], 4096));
    }
    for (; i + 2 < argc; i += 3) {
        if (strcmp(argv[i], "-query_sequences") == 0) { /* src/main.c:270-296: four arguments */
            if (i + 3 >= argc) DIE("-query_sequences threshold {canonical|non_canonical} list_sequence_files\n");
            const double threshold = atof(argv[i + 1]);
            if (threshold == 0) DIE("Could not parse threshold for command -query_sequences.\n");
            if (strcmp(argv[i + 2], "canonical") != 0 && strcmp(argv[i + 2], "non_canonical") != 0)
                DIE("Unrecognized type of k-mers to search for %s.\n", argv[i]);
            FILE* lst = fopen(argv[i + 3], "r");
            if (!lst) DIE("Invalid sequence query file list.\n");
            while (fgets(buffer, sizeof buffer, lst)) {
                buffer[strcspn(buffer, "\r\n")] = 0;
                if (buffer[0]) query_sequences(h, buffer, threshold, strcmp(argv[i + 2], "canonical") == 0, nb_genomes, names);
            }
            fclose(lst);
            i++;
            continue;
        }
        const int binary = strcmp(argv[i + 1], "kmers_comp") == 0;
        if (!binary && strcmp(argv[i + 1], "kmers") != 0) DIE("Unrecognized type of input files for %s.\n", argv[i]);
        if (strcmp(argv[i], "-extract_kmers") == 0) { extract_kmers(h, argv[i + 2], k, binary); continue; }
        FILE* lst = fopen(argv[i + 2], "r");
        if (!lst) DIE("Invalid k-mer queries files list.\n");
        while (fgets(buffer, sizeof buffer, lst)) {
            buffer[strcspn(buffer, "\r\n")] = 0;
            if (!buffer[0]) continue;
            if (strcmp(argv[i], "-query_kmers") == 0) query_kmers(h, buffer, k, binary, nb_genomes, names);
            else if (strcmp(argv[i], "-query_branching") == 0) query_branching(h, buffer, k, binary);
            else DIE("Unrecognized command %s.\n", argv[i]);
        }
        fclose(lst);
    }
    bft_gpu_free(h);
    return 0;
}
