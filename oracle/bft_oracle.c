/*
 * oracle/bft_oracle.c -- TEST INFRASTRUCTURE ONLY (see bft_oracle.h).
 *
 * CPU restatement of the Bloom Filter Trie insertion / presence / colour path.
 * Two parts:
 *   1. an insertion model that restates the container-selection logic of
 *      src/insertNode.c:38-226 and :241-423 (first-BF-positive CC, false
 *      positive recycling, last-CC-below-255 seeding, UC burst at 255 rows,
 *      suffix-group burst at 255 rows into a child Node) on sorted arrays;
 *   2. freeze(): the packed arrays of include/CC.h:34-67 (BF | filter2 |
 *      SkipFilter2 | SkipFilter3, filter3, extra_filter3, children_type,
 *      children UCs of 128 prefixes), on which presenceKmer / findCluster /
 *      isKmerPresent (src/presenceNode.c:1284-1921) are restated byte for byte
 *      of the algorithm (byte-LUT popcounts, skip cells, memcmp row search).
 *
 * level_min==0 levels (suffix length not 9 mod 36, not the root): the
 * reference has no extra_filter3 there -- the cluster-start bit of a prefix
 * hides in bit 7 of the last suffix byte of its first row, or in bit 0 of a
 * child Node's UC_array.nb_children -- and findCluster walks children_type
 * position by position from the SkipFilter3 cell (src/presenceNode.c:
 * 1690-1812), handing running child / node counts on to presenceKmer
 * (:1425-1448).  findCluster_lm0 below restates that walk and those counts;
 * the frozen model keeps the flag bits in one bitvector per CC (they are
 * moved into the rows when a .bft file is written, and back when one is
 * read), so the walk reads the flag there and ACCOUNTS (counting mode) the
 * byte the reference dereferences for it.
 *
 * PARITY: see the header -- primitives pinned, trie-level "parity unpinned"
 * against the reference binary (unbuildable here), checked against ground
 * truth set semantics.
 */
#define _GNU_SOURCE
#include "bft_oracle.h"

#include <limits.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* include/default_param.h:3-46 */
#define NB_CHAR_SUF_PREF 9
#define SIZE_BYTES_SUF_PREF 3
#define MODULO_HASH 1504
#define SIZE_BF_BYTES (MODULO_HASH / 8)
#define NB_KMERS_PER_UC 255
#define NB_UC_PER_SKP 128
#define TRESH_SUF_PREF 3584
#define CEIL(a, b) (((a) + (b)-1) / (b))

#ifdef ORC_COUNT
static __thread uint64_t g_touch, g_ccs, g_lvls;
#define TOUCH(n) (g_touch += (uint64_t)(n))
#define CC_SCANNED() (g_ccs++)
#define LVL_VISITED() (g_lvls++)
#else
#define TOUCH(n) ((void)0)
#define CC_SCANNED() ((void)0)
#define LVL_VISITED() ((void)0)
#endif

/* ------------------------------------------------------------------ */
/* primitives                                                         */
/* ------------------------------------------------------------------ */

static uint8_t REV[256];  /* src/popcnt.c:112-127: swaps the four 2-bit fields of a byte */
static uint8_t POP8[256]; /* src/popcnt.c:45-110 */
static int g_luts_ready = 0;

static void init_luts(void) {
    if (g_luts_ready) return;
    for (int b = 0; b < 256; b++) {
        REV[b] = (uint8_t)(((b & 0x3) << 6) | ((b & 0xc) << 2) | ((b & 0x30) >> 2) | ((b & 0xc0) >> 6));
        int c = 0;
        for (int j = 0; j < 8; j++) c += (b >> j) & 1;
        POP8[b] = (uint8_t)c;
    }
    g_luts_ready = 1;
}

/* popcnt_8_par (include/popcnt.h:27-37): popcount of bytes [start, end) */
static inline int popcnt_8_par(const uint8_t *v, int start, int end) {
    int c = 0;
    for (int i = start; i < end; i++) c += POP8[v[i]];
    TOUCH(end > start ? end - start : 0);
    return c;
}

#define XP1 11400714785074694791ULL
#define XP2 14029467366897019727ULL
#define XP3 1609587929392839161ULL
#define XP4 9650029242287828579ULL
#define XP5 2870177450012600261ULL

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xround(uint64_t acc, uint64_t in) {
    acc += in * XP2;
    acc = rotl64(acc, 31);
    return acc * XP1;
}
static inline uint64_t xmerge(uint64_t acc, uint64_t v) {
    acc ^= xround(0, v);
    return acc * XP1 + XP4;
}

uint64_t orc_xxh64(const void *data, size_t len, uint64_t seed) {
    const uint8_t *p = (const uint8_t *)data, *end = p + len;
    uint64_t h;
    if (len >= 32) {
        const uint8_t *lim = end - 32;
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        do {
            v1 = xround(v1, rd64(p));
            v2 = xround(v2, rd64(p + 8));
            v3 = xround(v3, rd64(p + 16));
            v4 = xround(v4, rd64(p + 24));
            p += 32;
        } while (p <= lim);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1);
        h = xmerge(h, v2);
        h = xmerge(h, v3);
        h = xmerge(h, v4);
    } else
        h = seed + XP5;
    h += (uint64_t)len;
    while (p + 8 <= end) {
        h ^= xround(0, rd64(p));
        h = rotl64(h, 27) * XP1 + XP4;
        p += 8;
    }
    if (p + 4 <= end) {
        h ^= (uint64_t)rd32(p) * XP1;
        h = rotl64(h, 23) * XP2 + XP3;
        p += 4;
    }
    while (p < end) {
        h ^= (*p) * XP5;
        h = rotl64(h, 11) * XP1;
        p++;
    }
    h ^= h >> 33;
    h *= XP2;
    h ^= h >> 29;
    h *= XP3;
    h ^= h >> 32;
    return h;
}

/* get_nb_bytes_power2_annot (include/log2.h:45-50): 6-bit chunks needed for id */
int orc_nb_bytes_id(uint32_t id) {
    int bits = id ? 32 - __builtin_clz(id) : 1;
    return CEIL(bits, 6);
}

/* parseKmerCount src/fasta.c:3-53 */
int orc_parse_kmer(const char *line, int k, uint8_t *tab) {
    int pos = 0;
    for (; pos < k; pos++) {
        uint8_t code;
        switch (line[pos]) {
        case 'a': case 'A': code = 0; break;
        case 'c': case 'C': code = 1; break;
        case 'g': case 'G': code = 2; break;
        case 'u': case 'U': case 't': case 'T': code = 3; break;
        default:
            memset(tab, 0, (size_t)((pos + 1) / 4));
            return 0;
        }
        tab[pos / 4] |= (uint8_t)(code << (2 * (pos % 4)));
    }
    return 1;
}

/* kmer_comp_to_ascii src/fasta.c:55-83 */
void orc_kmer_to_ascii(const uint8_t *kmer, int k, char *out) {
    static const char C2A[4] = {'A', 'C', 'G', 'T'};
    for (int j = 0; j < k; j++) out[j] = C2A[(kmer[j / 4] >> (2 * (j % 4))) & 3];
    out[k] = '\0';
}

/* ------------------------------------------------------------------ */
/* annotation codec  (src/annotation.c)                               */
/* ------------------------------------------------------------------ */

static int put_id(uint8_t *out, uint32_t id, uint8_t start_flag, uint8_t cont_flag) {
    int nb = orc_nb_bytes_id(id);
    for (int j = 0; j < nb; j++) {
        uint8_t chunk = (uint8_t)((id >> (6 * (nb - 1 - j))) & 0x3f);
        out[j] = (uint8_t)((chunk << 2) | (j == 0 ? start_flag : cont_flag));
    }
    return nb;
}

/* sizes as compute_best_mode (src/annotation.c:416-656) computes them; the
 * tie rules of :634-650 (mode 2 over 1 on equality, mode 0 when <=) kept. */
static void annot_sizes(const uint32_t *ids, int n, int *sz0, int *sz1, int *sz2) {
    int s1 = 0, s2 = 0;
    for (int a = 0; a < n;) {
        int b = a;
        while (b + 1 < n && ids[b + 1] == ids[b] + 1) b++;
        s1 += orc_nb_bytes_id(ids[a]) + orc_nb_bytes_id(ids[b]);
        a = b + 1;
    }
    for (int a = 0; a < n; a++) s2 += orc_nb_bytes_id(ids[a]);
    *sz0 = n ? CEIL(3 + (int)ids[n - 1], 8) : 1;
    *sz1 = s1;
    *sz2 = s2;
}

/* The mode an annotation ends up in depends on its history: the reference re-decides at EVERY insertion of a genome id
 * (modify_annotations -> compute_best_mode, src/retrieveAnnotation.c:232-314, src/annotation.c:416-656), and on a size tie
 * it keeps the mode the annotation is already in (:652-653).  Ids reach a k-mer in ascending order, so the history of a
 * colour set is its sorted id list: replay it.  Per step, with g the id being added (src/annotation.c:621-653):
 *   size0 = CEIL(3 + g, 8)                      (disabled_flags is never set anywhere in the reference: bit 0 is always clear)
 *   size2 = previous size2 + bytes(g)
 *   size1 = previous size1 + 2 bytes(g) when g opens a new range, else + bytes(g) - bytes(previous id)   (:628-633)
 *   min  = mode 2 if size2 <= size1 else mode 1; mode 0 if that minimum is >= size0            (:638-650)
 *   if the current mode's new size equals the minimum, the current mode stays                   (:652-653)
 * Not replayed: while an annotation is in mode 0 with ids >= 64 the reference prices the END of a run with the byte count of
 * the id that FOLLOWS it (:515-523), one byte too many when a run ends at id 4095 or 262143 -- an estimate quirk that needs
 * more than 4096 genomes in bitmap mode to show. */
static int annot_best(const uint32_t *ids, int n, int *mode) {
    int s1 = 0, s2 = 0, cur = -1, cur_sz = 1;
    /* While in bitmap mode the reference re-derives the list sizes from the bits and prices the end of a run with the byte count of
     * the id one past it (src/annotation.c:515-523): +1 for every run ending at 63, 4095, 262143, 16777215 (`over`). */
    int over63 = 0, over4095 = 0, over262143 = 0, over16m = 0;
    for (int a = 0; a < n; a++) {
        const int b = orc_nb_bytes_id(ids[a]);
        const int s0 = CEIL(3 + (int)ids[a], 8);
        s2 += b;
        if (a > 0 && ids[a] == ids[a - 1] + 1) s1 += b - orc_nb_bytes_id(ids[a - 1]);
        else s1 += 2 * b;
        const int s1e = s1 + (cur == 0 ? over63 + over4095 + over262143 + over16m : 0);
        int m, sz;
        if (s2 <= s1e) { m = 2; sz = s2; } else { m = 1; sz = s1e; }
        if (sz >= s0) { m = 0; sz = s0; }
        if (cur >= 0 && m != cur) {
            const int same = cur == 0 ? s0 : (cur == 1 ? s1e : s2);
            if (same == sz) m = cur;
        }
        cur = m;
        cur_sz = sz;
        const int ext = a > 0 && ids[a] == ids[a - 1] + 1;
        if (ids[a] == 63u) over63 = 1; else if (ids[a] == 64u && ext) over63 = 0;
        if (ids[a] == 4095u) over4095 = 1; else if (ids[a] == 4096u && ext) over4095 = 0;
        if (ids[a] == 262143u) over262143 = 1; else if (ids[a] == 262144u && ext) over262143 = 0;
        if (ids[a] == 16777215u) over16m = 1; else if (ids[a] == 16777216u && ext) over16m = 0;
    }
    if (n == 0) { int s0; annot_sizes(ids, n, &s0, &s1, &s2); cur = 0; cur_sz = s0; }
    *mode = cur;
    return cur_sz;
}

int orc_annot_encode(const uint32_t *ids, int n, uint8_t *out, int cap) {
    int mode, sz = annot_best(ids, n, &mode);
    if (n == 0) { if (cap < 1) return -1; out[0] = 0; return 1; }
    if (sz > cap) return -1;
    memset(out, 0, (size_t)sz);
    if (mode == 0) { /* genome g <-> bit g+2 (src/annotation.c:2134-2144) */
        for (int a = 0; a < n; a++) out[(ids[a] + 2) / 8] |= (uint8_t)(1u << ((ids[a] + 2) % 8));
    } else if (mode == 1) { /* ranges: start byte flag 1, continuation flag 2 (:2145-2178) */
        int o = 0;
        for (int a = 0; a < n;) {
            int b = a;
            while (b + 1 < n && ids[b + 1] == ids[b] + 1) b++;
            o += put_id(out + o, ids[a], 0x1, 0x2);
            o += put_id(out + o, ids[b], 0x1, 0x2);
            a = b + 1;
        }
    } else { /* id list: start flag 2, continuation flag 1 (:2228-2244) */
        int o = 0;
        for (int a = 0; a < n; a++) o += put_id(out + o, ids[a], 0x2, 0x1);
    }
    return sz;
}

/* get_id_genomes_from_annot, modes 0/1/2, comp_annot<=0 (src/annotation.c:2086-2250) */
int orc_annot_decode(const uint8_t *annot, int size, uint32_t *ids, int cap) {
    int n = 0, i = 0;
    if (size <= 0) return 0;
    TOUCH(size);
    int mode = annot[0] & 0x3;
    if (mode == 0) {
        for (i = 2; i < size * 8; i++)
            if (annot[i / 8] & (1u << (i % 8))) { if (n < cap) ids[n] = (uint32_t)(i - 2); n++; }
    } else if (mode == 1) {
        int it = 0;
        uint32_t prev = 0;
        while (i < size && (annot[i] & 0x1)) {
            uint32_t v = annot[i] >> 2;
            i++;
            while (i < size && (annot[i] & 0x2)) { v = (v << 6) | (annot[i] >> 2); i++; }
            if (it) {
                for (uint32_t j = prev + 1; j <= v && prev != v; j++) { if (n < cap) ids[n] = j; n++; }
            } else {
                if (n < cap) ids[n] = v;
                n++;
                prev = v;
            }
            it = !it;
        }
    } else if (mode == 2) {
        while (i < size && (annot[i] & 0x2)) {
            uint32_t v = annot[i] >> 2;
            i++;
            while (i < size && (annot[i] & 0x1)) { v = (v << 6) | (annot[i] >> 2); i++; }
            if (n < cap) ids[n] = v;
            n++;
        }
    } else
        return -1; /* mode 3 needs comp_set_colors: never produced without Judy compression */
    return n;
}

/* ------------------------------------------------------------------ */
/* structures                                                         */
/* ------------------------------------------------------------------ */

typedef struct orc_node orc_node;

typedef struct { /* insertion model: one prefix of a CC */
    uint32_t r;   /* rotated prefix n2..n9,n1 (src/presenceNode.c:1367-1371) */
    uint16_t cnt; /* children_type value: suffix rows, 0 => child Node */
    union {
        uint8_t *rows;  /* cnt rows of (nbm1 suffix bytes + 4-byte colour-set id) */
        orc_node *node; /* cnt == 0 */
        uint32_t cs;    /* leaf level (suffix length 9): annotation only */
    } c;
} orc_pref;

typedef struct { /* frozen UC (include/UC.h:13-20 subset) */
    uint8_t *suffixes;
    int size_annot;
    int nb_children;
} orc_uc;

typedef struct {
    uint8_t bf[SIZE_BF_BYTES];
    orc_pref *prefs;
    int n, cap;
    /* frozen, include/CC.h:34-67 */
    uint16_t type, nb_elem, nb_Node_children;
    uint8_t *BF_filter2, *filter3, *extra_filter3, *children_type;
    orc_uc *children;
    orc_node **children_nodes;
} orc_cc;

struct orc_node {
    orc_cc *ccs;
    int ncc;
    uint8_t *uc; /* model rows: nb(i) suffix bytes + 4-byte cs, sorted by memcmp */
    int uc_n;
    orc_uc fuc; /* frozen node UC */
};

/* comp_set_colors (src/write_to_disk.c:283-310): elements of equal-size entries */
typedef struct { int64_t last_index; int size_annot; uint8_t *bytes; } comp_elem;

struct orc_bft {
    int k, r1, r2;
    /* optional test mode: annotations written as mode-3 indices into comp_set_colors, and/or with the last
     * annotation byte of the widest rows moved to the extended-annotation table (what reference-built files hold) */
    int comp_on, ext_on;
    uint32_t *cs_pos; comp_elem *celems; int ncelems;
    uint64_t *hash_v;   /* include/Node.h:158-185 */
    uint16_t *hmod;     /* hash_v % 1504 for the 2^14 keys used when compressed==0 */
    orc_node root;
    int dirty;
    int nb_genomes_loaded;
    long nkmers;
    /* colour sets, interned; set 0 is empty */
    uint32_t *cs_ids; long cs_ids_n, cs_ids_cap;
    long *cs_off; long cs_n, cs_cap; /* cs_off[c]..cs_off[c+1] */
    /* memo (cs, gid) -> cs' open addressing */
    uint64_t *memo_key; uint32_t *memo_val; long memo_cap, memo_n;
    int *szmemo; long szmemo_n; /* annotation size per colour set (cs_size) */
};

static comp_elem *g_comp = NULL; /* only used while a file is being loaded (single-threaded test infrastructure) */
static int g_ncomp = 0;

/* decomp_annotation + get_id_genomes_from_annot with comp_annot > 0 (src/annotation.c:1840-1922, :2179-2226):
 * the stored ids are deltas to the previous stored id; mode 1 then expands (start, stop) pairs */
static int decode_comp_entry(const uint8_t *a, int size, uint32_t *ids, int cap) {
    int mode = a[0] & 3;
    if (mode == 0 || mode == 3) return orc_annot_decode(a, size, ids, cap);
    uint8_t flag1 = mode == 2 ? 2 : 1, flag2 = mode == 2 ? 1 : 2;
    uint32_t st[8192];
    int ns = 0, i = 0;
    while (i < size && (a[i] & flag1) && ns < 8192) {
        uint32_t v = a[i++] >> 2;
        while (i < size && (a[i] & flag2)) v = (v << 6) | (a[i++] >> 2);
        st[ns++] = v;
    }
    for (int q = 1; q < ns; q++) st[q] += st[q - 1];
    int n = 0;
    if (mode == 2) { for (int q = 0; q < ns; q++) { if (n < cap) ids[n] = st[q]; n++; } }
    else for (int q = 0; q + 1 < ns; q += 2) for (uint32_t v = st[q]; v <= st[q + 1]; v++) { if (n < cap) ids[n] = v; n++; }
    return n;
}

static int decode_any(const uint8_t *annot, int size, uint32_t *ids, int cap) {
    if (size > 0 && (annot[0] & 3) == 3) { /* index into comp_set_colors (src/annotation.c:2097-2119) */
        uint32_t pos = annot[0] >> 2;
        for (int i = 1; i < size && (annot[i] & 1); i++) pos |= ((uint32_t)(annot[i] >> 1)) << (6 + (i - 1) * 7);
        int e = 0;
        while (e < g_ncomp && (int64_t)pos > g_comp[e].last_index) e++;
        if (e >= g_ncomp) return -1;
        int64_t rel = e == 0 ? pos : (int64_t)pos - g_comp[e - 1].last_index - 1;
        return decode_comp_entry(g_comp[e].bytes + rel * g_comp[e].size_annot, g_comp[e].size_annot, ids, cap);
    }
    return orc_annot_decode(annot, size, ids, cap);
}


static inline int nb_bytes(int i) { return CEIL(i * 2, 8); }
static inline int nbm1_bytes(int i) { return i > 9 ? CEIL((i - 9) * 2, 8) : 0; }
/* mask_shift_kmer, src/CC.c:1913-1989 */
static inline uint8_t mask_shift(int i) {
    switch (i % 36) { case 9: return 0xff; case 18: return 0x3; case 27: return 0xf; default: return 0x3f; }
}

static void *xmalloc(size_t n) { void *p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(1); } return p; }
static void *xrealloc(void *q, size_t n) { void *p = realloc(q, n ? n : 1); if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(1); } return p; }
static void *xcalloc(size_t n, size_t m) { void *p = calloc(n ? n : 1, m ? m : 1); if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(1); } return p; }

/* ------------------------------------------------------------------ */
/* colour sets                                                        */
/* ------------------------------------------------------------------ */

static uint32_t cs_new(orc_bft *t, const uint32_t *ids, long n, uint32_t extra, int has_extra) {
    long need = n + (has_extra ? 1 : 0);
    if (t->cs_ids_n + need > t->cs_ids_cap) {
        t->cs_ids_cap = (t->cs_ids_n + need) * 2 + 64;
        t->cs_ids = xrealloc(t->cs_ids, (size_t)t->cs_ids_cap * 4);
    }
    if (t->cs_n + 2 > t->cs_cap) {
        t->cs_cap = t->cs_cap * 2 + 64;
        t->cs_off = xrealloc(t->cs_off, (size_t)t->cs_cap * sizeof(long));
    }
    if (n) memmove(t->cs_ids + t->cs_ids_n, ids, (size_t)n * 4);
    if (has_extra) t->cs_ids[t->cs_ids_n + n] = extra;
    t->cs_ids_n += need;
    t->cs_n++;
    t->cs_off[t->cs_n] = t->cs_ids_n;
    return (uint32_t)(t->cs_n - 1);
}

static void memo_grow(orc_bft *t) {
    long ncap = t->memo_cap ? t->memo_cap * 2 : 1024;
    uint64_t *nk = xmalloc((size_t)ncap * 8);
    uint32_t *nv = xmalloc((size_t)ncap * 4);
    memset(nk, 0xff, (size_t)ncap * 8);
    for (long a = 0; a < t->memo_cap; a++) {
        if (t->memo_key[a] == UINT64_MAX) continue;
        uint64_t h = t->memo_key[a] * 0x9E3779B97F4A7C15ULL;
        long p = (long)(h >> 20) & (ncap - 1);
        while (nk[p] != UINT64_MAX) p = (p + 1) & (ncap - 1);
        nk[p] = t->memo_key[a];
        nv[p] = t->memo_val[a];
    }
    free(t->memo_key);
    free(t->memo_val);
    t->memo_key = nk;
    t->memo_val = nv;
    t->memo_cap = ncap;
}

/* modify_annotations (src/retrieveAnnotation.c:232-314) at the set level:
 * add genome gid to colour set cs (ids arrive in non-decreasing order). */
static uint32_t cs_add(orc_bft *t, uint32_t cs, uint32_t gid) {
    long a = t->cs_off[cs], b = t->cs_off[cs + 1];
    if (b > a && t->cs_ids[b - 1] == gid) return cs;
    if (t->memo_n * 2 >= t->memo_cap) memo_grow(t);
    uint64_t key = ((uint64_t)cs << 32) | gid;
    uint64_t h = key * 0x9E3779B97F4A7C15ULL;
    long p = (long)(h >> 20) & (t->memo_cap - 1);
    while (t->memo_key[p] != UINT64_MAX) {
        if (t->memo_key[p] == key) return t->memo_val[p];
        p = (p + 1) & (t->memo_cap - 1);
    }
    uint32_t ncs;
    if (b > a && t->cs_ids[b - 1] > gid) {
        /* out-of-order id: keep the set sorted (the reference requires
         * non-decreasing ids; we stay well defined anyway) */
        uint32_t *tmp = xmalloc((size_t)(b - a + 1) * 4);
        long m = 0; int placed = 0, dup = 0;
        for (long q = a; q < b; q++) {
            if (!placed && t->cs_ids[q] >= gid) { if (t->cs_ids[q] == gid) dup = 1; else tmp[m++] = gid; placed = 1; }
            tmp[m++] = t->cs_ids[q];
        }
        if (dup) { free(tmp); ncs = cs; }
        else { ncs = cs_new(t, tmp, m, 0, 0); free(tmp); }
    } else {
        /* cs_new may realloc cs_ids: pass offsets through a temp copy */
        long n = b - a;
        uint32_t *tmp = xmalloc((size_t)(n + 1) * 4);
        if (n) memcpy(tmp, t->cs_ids + a, (size_t)n * 4);
        ncs = cs_new(t, tmp, n, gid, 1);
        free(tmp);
    }
    /* re-probe: memo may not have moved, but recompute slot for safety */
    p = (long)(h >> 20) & (t->memo_cap - 1);
    while (t->memo_key[p] != UINT64_MAX) p = (p + 1) & (t->memo_cap - 1);
    t->memo_key[p] = key;
    t->memo_val[p] = ncs;
    t->memo_n++;
    return ncs;
}

int orc_colorset(orc_bft *t, uint32_t cs, uint32_t *ids, int cap) {
    long a = t->cs_off[cs], b = t->cs_off[cs + 1];
    for (long q = a; q < b && q - a < cap; q++) ids[q - a] = t->cs_ids[q];
    return (int)(b - a);
}

/* ------------------------------------------------------------------ */
/* create / free                                                      */
/* ------------------------------------------------------------------ */

/* create_hash_v_array include/Node.h:158-185 */
static uint64_t *make_hash_v(int r1, int r2) {
    uint32_t nb = 1u << 18;
    uint64_t *hv = xmalloc((size_t)nb * 2 * 8);
    uint8_t g[SIZE_BYTES_SUF_PREF];
    for (uint32_t i = 0; i < nb; i++) {
        int nbits = NB_CHAR_SUF_PREF * 2;
        for (int j = 0; j < SIZE_BYTES_SUF_PREF; j++) {
            nbits -= 8;
            if (nbits >= 0) g[j] = (uint8_t)((i >> nbits) & 0xff);
            else g[j] = (uint8_t)((i << (-nbits)) & 0xff);
        }
        hv[i * 2] = orc_xxh64(g, SIZE_BYTES_SUF_PREF, (uint64_t)(long long)r1);
        hv[i * 2 + 1] = orc_xxh64(g, SIZE_BYTES_SUF_PREF, (uint64_t)(long long)r2);
    }
    return hv;
}

orc_bft *orc_create(int k, int r1, int r2) {
    if (k < 9 || k > 126 || k % 9) return NULL; /* src/main.c:61-63 */
    init_luts();
    orc_bft *t = xcalloc(1, sizeof(*t));
    t->k = k;
    t->r1 = r1 > 0 ? r1 : ORC_DEFAULT_R1;
    t->r2 = r2 > 0 ? r2 : ORC_DEFAULT_R2;
    t->hash_v = make_hash_v(t->r1, t->r2);
    t->hmod = xmalloc(16384 * 2 * 2);
    for (int i = 0; i < 16384; i++) {
        t->hmod[i * 2] = (uint16_t)(t->hash_v[i * 2] % MODULO_HASH);
        t->hmod[i * 2 + 1] = (uint16_t)(t->hash_v[i * 2 + 1] % MODULO_HASH);
    }
    t->cs_cap = 64;
    t->cs_off = xmalloc((size_t)t->cs_cap * sizeof(long));
    t->cs_off[0] = 0;
    t->cs_off[1] = 0;
    t->cs_n = 1; /* set 0 = empty */
    t->dirty = 1;
    return t;
}

static void free_frozen_cc(orc_cc *cc) {
    free(cc->BF_filter2); free(cc->filter3); free(cc->extra_filter3); free(cc->children_type);
    if (cc->children) {
        int nbk = CEIL((int)cc->nb_elem, NB_UC_PER_SKP);
        for (int b = 0; b < nbk; b++) free(cc->children[b].suffixes);
        free(cc->children);
    }
    free(cc->children_nodes);
    cc->BF_filter2 = cc->filter3 = cc->extra_filter3 = cc->children_type = NULL;
    cc->children = NULL;
    cc->children_nodes = NULL;
}

static void free_node(orc_node *nd, int i) {
    for (int c = 0; c < nd->ncc; c++) {
        orc_cc *cc = &nd->ccs[c];
        free_frozen_cc(cc);
        if (i != 9)
            for (int j = 0; j < cc->n; j++) {
                if (cc->prefs[j].cnt == 0) { free_node(cc->prefs[j].c.node, i - 9); free(cc->prefs[j].c.node); }
                else free(cc->prefs[j].c.rows);
            }
        free(cc->prefs);
    }
    free(nd->ccs);
    free(nd->uc);
    free(nd->fuc.suffixes);
}

void orc_free(orc_bft *t) {
    if (!t) return;
    free_node(&t->root, t->k);
    for (int e = 0; e < t->ncelems; e++) free(t->celems[e].bytes);
    free(t->celems); free(t->cs_pos);
    free(t->hash_v); free(t->hmod); free(t->cs_ids); free(t->cs_off); free(t->memo_key); free(t->memo_val); free(t->szmemo);
    free(t);
}

int orc_k(const orc_bft *t) { return t->k; }
int orc_kmer_bytes(const orc_bft *t) { return nb_bytes(t->k); }
const uint64_t *orc_hash_v(const orc_bft *t) { return t->hash_v; }

/* ------------------------------------------------------------------ */
/* insertion model                                                    */
/* ------------------------------------------------------------------ */

/* prefix extraction: src/presenceNode.c:1327-1343 (key) and :1367-1371 (rotation) */
static inline void prefix_of(const uint8_t *suf, uint32_t *key, uint32_t *r) {
    uint32_t s0 = REV[suf[0]], s1 = REV[suf[1]], s2 = REV[suf[2]] & 0xc0;
    uint32_t sp = (s0 << 8) | s1; /* n1..n8 */
    *key = sp & 0x3fff;           /* n2..n8 */
    uint32_t b0 = (sp >> 6) & 0xff, b1 = ((sp << 2) | (s2 >> 6)) & 0xff, b2 = (sp >> 8) & 0xc0;
    *r = (b0 << 10) | (b1 << 2) | (b2 >> 6); /* n2..n9,n1 */
}

/* drop the first 9 nt: src/presenceNode.c:1853-1861 == src/insertNode.c:96-104 */
static inline void strip9(uint8_t *s, int i) {
    int nb_cell = nb_bytes(i);
    int del = 2 + ((i == 45) || (i == 81) || (i == 117));
    int j;
    for (j = 0; j < nb_cell - del; j++) {
        s[j] = (uint8_t)(s[j + 2] >> 2);
        if (j + 3 < nb_cell) s[j] |= (uint8_t)(s[j + 3] << 6);
    }
    s[j - 1] &= mask_shift(i);
}

static inline int bf_test(const uint8_t *bf, uint16_t h) { return bf[h >> 3] & (1u << (h & 7)); }
static inline void bf_set(uint8_t *bf, uint16_t h) { bf[h >> 3] |= (uint8_t)(1u << (h & 7)); }

static int pref_lower_bound(const orc_cc *cc, uint32_t r) {
    int lo = 0, hi = cc->n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (cc->prefs[mid].r < r) lo = mid + 1; else hi = mid; }
    return lo;
}

/* lower bound over fixed-stride rows by memcmp (src/UC.c:81-124, mask 0xff) */
static int rows_lower_bound(const uint8_t *rows, int n, int stride, const uint8_t *suf, int nbs) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = lo + (hi - lo) / 2; if (memcmp(rows + (size_t)mid * stride, suf, (size_t)nbs) < 0) lo = mid + 1; else hi = mid; }
    return lo;
}

static inline uint32_t row_cs(const uint8_t *row, int nbs) { uint32_t v; memcpy(&v, row + nbs, 4); return v; }
static inline void row_set_cs(uint8_t *row, int nbs, uint32_t v) { memcpy(row + nbs, &v, 4); }

static orc_pref *cc_open_slot(orc_cc *cc, int pos) {
    if (cc->n == cc->cap) { cc->cap = cc->cap ? cc->cap * 2 : 16; cc->prefs = xrealloc(cc->prefs, (size_t)cc->cap * sizeof(orc_pref)); }
    memmove(&cc->prefs[pos + 1], &cc->prefs[pos], (size_t)(cc->n - pos) * sizeof(orc_pref));
    cc->n++;
    return &cc->prefs[pos];
}

/* insertSP_CC (src/CC.c:714-1474) at the model level: new prefix r at sorted
 * position pos with one suffix row (or one annotation at the leaf level). */
static void cc_insert_prefix(orc_bft *t, orc_cc *cc, int pos, uint32_t r, int i, uint8_t *suf, uint32_t gid) {
    orc_pref *p = cc_open_slot(cc, pos);
    p->r = r;
    p->cnt = 1;
    if (i == 9) p->c.cs = cs_add(t, 0, gid);
    else {
        int nbs = nbm1_bytes(i);
        strip9(suf, i);
        p->c.rows = xmalloc((size_t)(nbs + 4));
        memcpy(p->c.rows, suf, (size_t)nbs);
        row_set_cs(p->c.rows, nbs, cs_add(t, 0, gid));
    }
}

typedef struct { uint32_t r; int idx; } sort_ent;
static int cmp_sort_ent(const void *a, const void *b) {
    const sort_ent *x = a, *y = b;
    if (x->r != y->r) return x->r < y->r ? -1 : 1;
    return x->idx - y->idx;
}

/* transform2CC (src/CC.c:40-367) / transform2CC_from_arraySuffix (:381-700):
 * build a new last CC of `node` from n rows of suffix length i
 * (row = nb(i) bytes + 4-byte cs, sorted by memcmp). */
static void node_add_cc_from_rows(orc_bft *t, orc_node *node, uint8_t *rows, int n, int i) {
    int nbi = nb_bytes(i), stride = nbi + 4, nbs = nbm1_bytes(i);
    node->ccs = xrealloc(node->ccs, (size_t)(node->ncc + 1) * sizeof(orc_cc));
    orc_cc *cc = &node->ccs[node->ncc++];
    memset(cc, 0, sizeof(*cc));
    sort_ent *ord = xmalloc((size_t)n * sizeof(sort_ent));
    for (int a = 0; a < n; a++) {
        uint32_t key, r;
        prefix_of(rows + (size_t)a * stride, &key, &r);
        bf_set(cc->bf, t->hmod[key * 2]);      /* src/CC.c:114-133 */
        bf_set(cc->bf, t->hmod[key * 2 + 1]);
        ord[a].r = r;
        ord[a].idx = a;
    }
    qsort(ord, (size_t)n, sizeof(sort_ent), cmp_sort_ent); /* quicksort_init on rotated prefixes, :137 */
    for (int a = 0; a < n;) {
        int b = a;
        while (b + 1 < n && ord[b + 1].r == ord[a].r) b++;
        orc_pref *p = cc_open_slot(cc, cc->n);
        p->r = ord[a].r;
        int cnt = b - a + 1;
        p->cnt = (uint16_t)cnt;
        if (i == 9) p->c.cs = row_cs(rows + (size_t)ord[a].idx * stride, nbi);
        else {
            p->c.rows = xmalloc((size_t)cnt * (nbs + 4));
            for (int q = 0; q < cnt; q++) {
                uint8_t tmp[40];
                const uint8_t *src = rows + (size_t)ord[a + q].idx * stride;
                memcpy(tmp, src, (size_t)nbi);
                strip9(tmp, i);
                /* sorted insert among the q rows already placed (binary_search_UC_array, :300-326) */
                int z = rows_lower_bound(p->c.rows, q, nbs + 4, tmp, nbs);
                memmove(p->c.rows + (size_t)(z + 1) * (nbs + 4), p->c.rows + (size_t)z * (nbs + 4), (size_t)(q - z) * (nbs + 4));
                memcpy(p->c.rows + (size_t)z * (nbs + 4), tmp, (size_t)nbs);
                row_set_cs(p->c.rows + (size_t)z * (nbs + 4), nbs, row_cs(src, nbi));
            }
        }
        a = b + 1;
    }
    free(ord);
}

/* insertKmer_Node (src/insertNode.c:38-226) + insertKmer_Node_special (:241-423) */
static void node_insert(orc_bft *t, orc_node *node, int i, uint8_t *suf, uint32_t gid) {
    for (;;) {
        int nbi = nb_bytes(i);
        uint32_t key, r;
        prefix_of(suf, &key, &r);
        uint16_t h1 = t->hmod[key * 2], h2 = t->hmod[key * 2 + 1];
        orc_cc *cc = NULL;
        for (int c = 0; c < node->ncc; c++) /* presenceKmer BF scan, src/presenceNode.c:1353-1362 */
            if (bf_test(node->ccs[c].bf, h1) && bf_test(node->ccs[c].bf, h2)) { cc = &node->ccs[c]; break; }

        if (cc) {
            int pos = pref_lower_bound(cc, r);
            if (pos < cc->n && cc->prefs[pos].r == r) { /* prefix present: insertNode.c:60-124 */
                orc_pref *p = &cc->prefs[pos];
                if (i == 9) { p->c.cs = cs_add(t, p->c.cs, gid); return; } /* :64-86 */
                strip9(suf, i);                                             /* :94-104 */
                if (p->cnt == 0) { node = p->c.node; i -= 9; continue; }    /* :106-113 */
                /* insertKmer_Node_special: the child is a suffix group */
                int nbs = nbm1_bytes(i), stride = nbs + 4;
                int z = rows_lower_bound(p->c.rows, p->cnt, stride, suf, nbs);
                if (z < p->cnt && memcmp(p->c.rows + (size_t)z * stride, suf, (size_t)nbs) == 0) {
                    uint8_t *row = p->c.rows + (size_t)z * stride; /* :417-421 modify_annotations */
                    row_set_cs(row, nbs, cs_add(t, row_cs(row, nbs), gid));
                    return;
                }
                if (p->cnt == NB_KMERS_PER_UC) { /* :291-352 burst the group into a child Node */
                    orc_node *child = xcalloc(1, sizeof(orc_node));
                    node_add_cc_from_rows(t, child, p->c.rows, p->cnt, i - 9);
                    free(p->c.rows);
                    p->cnt = 0;
                    p->c.node = child;
                    /* :120-123 then insert the new suffix into the new node */
                    node = child;
                    i -= 9;
                    continue;
                }
                /* :354-414 sorted insert of the new suffix row */
                p->c.rows = xrealloc(p->c.rows, (size_t)(p->cnt + 1) * stride);
                memmove(p->c.rows + (size_t)(z + 1) * stride, p->c.rows + (size_t)z * stride, (size_t)(p->cnt - z) * stride);
                memcpy(p->c.rows + (size_t)z * stride, suf, (size_t)nbs);
                row_set_cs(p->c.rows + (size_t)z * stride, nbs, cs_add(t, 0, gid));
                p->cnt++;
                t->nkmers++;
                return;
            }
            /* BF says yes, filters say no: false-positive recycling, insertNode.c:126-136 */
            cc_insert_prefix(t, cc, pos, r, i, suf, gid);
            t->nkmers++;
            return;
        }

        /* no BF-positive CC: the node's UC, presenceNode.c:1554-1573 */
        int stride = nbi + 4;
        int z = rows_lower_bound(node->uc, node->uc_n, stride, suf, nbi);
        if (z < node->uc_n && memcmp(node->uc + (size_t)z * stride, suf, (size_t)nbi) == 0) {
            uint8_t *row = node->uc + (size_t)z * stride; /* insertNode.c:87-90 / :114-118 */
            row_set_cs(row, nbi, cs_add(t, row_cs(row, nbi), gid));
            return;
        }
        if (node->ncc > 0 && node->ccs[node->ncc - 1].n < NB_KMERS_PER_UC) {
            /* insertNode.c:146-180: the last CC still has < 255 prefixes */
            orc_cc *last = &node->ccs[node->ncc - 1];
            bf_set(last->bf, h1);
            bf_set(last->bf, h2);
            /* :170 re-runs presenceKmer from the last CC: it is now BF-positive there */
            cc_insert_prefix(t, last, pref_lower_bound(last, r), r, i, suf, gid);
            t->nkmers++;
            return;
        }
        /* insertKmer_UC (src/UC.c:13-79) */
        node->uc = xrealloc(node->uc, (size_t)(node->uc_n + 1) * stride);
        memmove(node->uc + (size_t)(z + 1) * stride, node->uc + (size_t)z * stride, (size_t)(node->uc_n - z) * stride);
        memcpy(node->uc + (size_t)z * stride, suf, (size_t)nbi);
        row_set_cs(node->uc + (size_t)z * stride, nbi, cs_add(t, 0, gid));
        node->uc_n++;
        t->nkmers++;
        if (node->uc_n == NB_KMERS_PER_UC) { /* insertNode.c:197-223: the UC is full -> new CC */
            node_add_cc_from_rows(t, node, node->uc, node->uc_n, i);
            free(node->uc);
            node->uc = NULL;
            node->uc_n = 0;
        }
        return;
    }
}

int orc_insert_kmers(orc_bft *t, const uint8_t *kmers, long n, uint32_t id_genome) {
    int nb = nb_bytes(t->k);
    uint8_t buf[40];
    for (long a = 0; a < n; a++) { /* src/insertNode.c:29-35 */
        memcpy(buf, kmers + (size_t)a * nb, (size_t)nb);
        node_insert(t, &t->root, t->k, buf, id_genome);
    }
    t->dirty = 1;
    return 0;
}

/* ------------------------------------------------------------------ */
/* freeze: packed arrays of include/CC.h:34-67                        */
/* ------------------------------------------------------------------ */

/* index of a comp_set_colors entry as annotation bytes (src/replaceAnnotation.c:380-386) */
static int put_comp_index(uint32_t id, uint8_t *out) {
    int n = 1;
    out[0] = (uint8_t)(((id & 0x3f) << 2) | 0x3);
    for (uint32_t rest = id >> 6; rest; rest >>= 7) out[n++] = (uint8_t)(((rest & 0x7f) << 1) | 0x1);
    return n;
}
static int cs_encode(orc_bft *t, uint32_t cs, uint8_t *out, int cap) {
    long a = t->cs_off[cs], b = t->cs_off[cs + 1];
    if (t->comp_on && b > a) { uint8_t tmp[8]; int n = put_comp_index(t->cs_pos[cs], tmp); if (n > cap) return -1; memcpy(out, tmp, (size_t)n); return n; }
    return orc_annot_encode(t->cs_ids + a, (int)(b - a), out, cap);
}
/* size of the annotation of colour set cs; the replay of compute_best_mode is O(ids) and freeze() asks once per row: remembered per set
 * (sets are immutable once created) */
static int cs_size(orc_bft *t, uint32_t cs) {
    long a = t->cs_off[cs], b = t->cs_off[cs + 1];
    int mode;
    if (t->comp_on && b > a) { uint8_t tmp[8]; return put_comp_index(t->cs_pos[cs], tmp); }
    if (b <= a) return 1;
    if ((long)cs >= t->szmemo_n) {
        const long ncap = t->cs_n + 1024;
        t->szmemo = xrealloc(t->szmemo, (size_t)ncap * sizeof(int));
        for (long i = t->szmemo_n; i < ncap; i++) t->szmemo[i] = -1;
        t->szmemo_n = ncap;
    }
    if (t->szmemo[cs] < 0) t->szmemo[cs] = annot_best(t->cs_ids + a, (int)(b - a), &mode);
    return t->szmemo[cs];
}

static void freeze_uc_rows(orc_bft *t, orc_uc *uc, const uint8_t *rows, int n, int nbs) {
    free(uc->suffixes);
    uc->suffixes = NULL;
    uc->nb_children = n;
    uc->size_annot = 0;
    if (!n) return;
    int sa = 1;
    for (int a = 0; a < n; a++) { int s = cs_size(t, row_cs(rows + (size_t)a * (nbs + 4), nbs)); if (s > sa) sa = s; }
    uc->size_annot = sa;
    uc->suffixes = xcalloc((size_t)n, (size_t)(nbs + sa));
    for (int a = 0; a < n; a++) {
        uint8_t *dst = uc->suffixes + (size_t)a * (nbs + sa);
        memcpy(dst, rows + (size_t)a * (nbs + 4), (size_t)nbs);
        cs_encode(t, row_cs(rows + (size_t)a * (nbs + 4), nbs), dst + nbs, sa);
    }
}

static void freeze_node(orc_bft *t, orc_node *nd, int i);

static void freeze_cc(orc_bft *t, orc_cc *cc, int i, int is_last) {
    free_frozen_cc(cc);
    int n = cc->n;
    int s = n >= TRESH_SUF_PREF ? 4 : 8, p = 18 - s; /* transform_Filter2n3 at 3584, insertNode.c:134-135 */
    int f2bytes = (1 << p) / 8, skip2 = n >= TRESH_SUF_PREF ? (1 << p) / 128 : 0, skip3 = n / 128;
    cc->nb_elem = (uint16_t)n;
    cc->BF_filter2 = xcalloc(1, (size_t)(SIZE_BF_BYTES + f2bytes + skip2 + skip3));
    memcpy(cc->BF_filter2, cc->bf, SIZE_BF_BYTES);
    uint8_t *f2 = cc->BF_filter2 + SIZE_BF_BYTES, *sk2 = f2 + f2bytes, *sk3 = sk2 + skip2;
    cc->filter3 = xcalloc(1, (size_t)(s == 8 ? n : CEIL(n, 2)));
    cc->extra_filter3 = xcalloc(1, (size_t)CEIL(n, 8));
    int type_byte = 0, nnodes = 0;
    if (i != 9)
        for (int j = 0; j < n; j++) { if (cc->prefs[j].cnt >= 16) type_byte = 1; if (cc->prefs[j].cnt == 0) nnodes++; }
    uint32_t prev_pu = UINT32_MAX;
    for (int j = 0; j < n; j++) {
        uint32_t r = cc->prefs[j].r, pu = r >> s, pv = r & ((1u << s) - 1);
        f2[pu >> 3] |= (uint8_t)(1u << (pu & 7));
        if (s == 8) cc->filter3[j] = (uint8_t)pv;
        else cc->filter3[j / 2] |= (uint8_t)((j & 1) ? (pv << 4) : pv);
        if (pu != prev_pu) {
            cc->extra_filter3[j >> 3] |= (uint8_t)(1u << (j & 7));
            if (j / 128 < skip3) sk3[j / 128]++;
            if (skip2) sk2[pu / 128]++;
            prev_pu = pu;
        }
    }
    int nbk = CEIL(n, NB_UC_PER_SKP);
    cc->children = xcalloc((size_t)nbk, sizeof(orc_uc));
    if (i != 9) {
        int nbs = nbm1_bytes(i);
        cc->children_type = xcalloc(1, (size_t)(type_byte ? n : CEIL(n, 2)));
        cc->children_nodes = xmalloc((size_t)nnodes * sizeof(orc_node *));
        cc->nb_Node_children = (uint16_t)nnodes;
        int kn = 0;
        for (int b = 0; b < nbk; b++) {
            int j0 = b * 128, j1 = j0 + 128 < n ? j0 + 128 : n, total = 0;
            for (int j = j0; j < j1; j++) total += cc->prefs[j].cnt;
            uint8_t *tmp = xmalloc((size_t)total * (nbs + 4));
            int o = 0;
            for (int j = j0; j < j1; j++) {
                orc_pref *pf = &cc->prefs[j];
                if (type_byte) cc->children_type[j] = (uint8_t)pf->cnt;
                else cc->children_type[j / 2] |= (uint8_t)((j & 1) ? (pf->cnt << 4) : pf->cnt);
                if (pf->cnt == 0) { cc->children_nodes[kn++] = pf->c.node; freeze_node(t, pf->c.node, i - 9); }
                else { memcpy(tmp + (size_t)o * (nbs + 4), pf->c.rows, (size_t)pf->cnt * (nbs + 4)); o += pf->cnt; }
            }
            freeze_uc_rows(t, &cc->children[b], tmp, total, nbs);
            if (total == 0) cc->children[b].size_annot = 1; /* insertNode.c:337-341 */
            free(tmp);
        }
    } else {
        for (int b = 0; b < nbk; b++) {
            int j0 = b * 128, j1 = j0 + 128 < n ? j0 + 128 : n, cnt = j1 - j0;
            uint8_t *tmp = xmalloc((size_t)cnt * 4);
            for (int j = j0; j < j1; j++) memcpy(tmp + (size_t)(j - j0) * 4, &cc->prefs[j].c.cs, 4);
            freeze_uc_rows(t, &cc->children[b], tmp, cnt, 0);
            free(tmp);
        }
    }
    cc->type = (uint16_t)((SIZE_BF_BYTES << 7) | (type_byte << 6) | (s << 1) | (is_last ? 1 : 0));
}

static void freeze_node(orc_bft *t, orc_node *nd, int i) {
    for (int c = 0; c < nd->ncc; c++) freeze_cc(t, &nd->ccs[c], i, c == nd->ncc - 1);
    freeze_uc_rows(t, &nd->fuc, nd->uc, nd->uc_n, nb_bytes(i));
}

void orc_freeze(orc_bft *t) {
    if (!t->dirty) return;
    freeze_node(t, &t->root, t->k);
    t->dirty = 0;
}

/* ------------------------------------------------------------------ */
/* query: presenceKmer / findCluster / isKmerPresent on packed arrays */
/* ------------------------------------------------------------------ */

/* include/CC.h:349-366 */
static inline int is_child(const orc_cc *cc, int pos, int type) {
    TOUCH(1);
    if (type) return cc->children_type[pos] != 0;
    if (pos & 1) return cc->children_type[pos / 2] > 0xf;
    return (cc->children_type[pos / 2] & 0xf) != 0;
}
static inline int getNbElts(const orc_cc *cc, int pos, int type) {
    if (type) return cc->children_type[pos];
    if (pos & 1) return cc->children_type[pos / 2] >> 4;
    return cc->children_type[pos / 2] & 0xf;
}
/* include/CC.h:471-489 */
static int count_nodes(const orc_cc *cc, int start, int end, int type) {
    int count = 0;
    const uint8_t *z;
    if (type) {
        for (z = cc->children_type + start; z < cc->children_type + end; z++) count += *z == 0;
        TOUCH(end > start ? end - start : 0);
    } else {
        z = cc->children_type + start / 2;
        if (start & 1) count -= (*z & 0xf) == 0;
        for (; z < cc->children_type + end / 2; z++) count += (*z < 0x10) + ((*z & 0xf) == 0);
        if (end & 1) count += (*z & 0xf) == 0;
        TOUCH(end / 2 - start / 2 + 1);
    }
    return count;
}
/* include/CC.h:532-550 */
static int count_children(const orc_cc *cc, int start, int end, int type) {
    int count = 0;
    const uint8_t *z;
    if (type) {
        for (z = cc->children_type + start; z < cc->children_type + end; z++) count += *z;
        TOUCH(end > start ? end - start : 0);
    } else {
        z = cc->children_type + start / 2;
        if (start & 1) count -= *z & 0xf;
        for (; z < cc->children_type + end / 2; z++) count += (*z >> 4) + (*z & 0xf);
        if (end & 1) count += *z & 0xf;
        TOUCH(end / 2 - start / 2 + 1);
    }
    return count;
}

static inline int level_min_of(const orc_bft *t, int i) { return i == t->k || i % 36 == 9; } /* src/CC.c:1906-1989 */

/* findCluster, level_min==1 branch: src/presenceNode.c:1578-1688 */
static void findCluster(const orc_cc *cc, int pos_filter2, int *pos_extra_filter3, int *hamming_weight_0) {
    int size_bf = cc->type >> 7;
    int s = (cc->type >> 1) & 0x1f, p = NB_CHAR_SUF_PREF * 2 - s;
    int size_filter2 = size_bf + (1 << p) / 8;
    int size_filter2_n_skip = size_filter2;
    int skip_filter2 = (1 << p) / NB_UC_PER_SKP;
    if (cc->nb_elem >= TRESH_SUF_PREF) size_filter2_n_skip += skip_filter2;
    int nb_cell_3rdlist = CEIL((int)cc->nb_elem, 8);
    int j = 0, m = 0, sum = 0, hamming_weight = 0;
    int k = size_bf + pos_filter2 / 8, cnt = 0;
    int pos_extra_tmp = INT_MAX, hw0 = 0, posFilter2;
    uint8_t word_tmp;

    if (cc->nb_elem >= TRESH_SUF_PREF) { /* SkipFilter2: :1619-1630 */
        int skip_posfilter2 = (pos_filter2 / NB_UC_PER_SKP < skip_filter2 ? pos_filter2 / NB_UC_PER_SKP : skip_filter2) + size_filter2;
        cnt = size_filter2;
        while (cnt < skip_posfilter2) { hamming_weight += cc->BF_filter2[cnt]; cnt++; }
        TOUCH(skip_posfilter2 - size_filter2);
        cnt -= size_filter2;
    }
    hamming_weight += popcnt_8_par(cc->BF_filter2, size_bf + cnt * (NB_UC_PER_SKP / 8), k); /* :1633 */
    word_tmp = cc->BF_filter2[k];
    for (k = 0; k <= pos_filter2 % 8; k++, word_tmp >>= 1) hamming_weight += word_tmp & 1; /* :1635-1636 */
    posFilter2 = hamming_weight;

    k = 0; hamming_weight = 0;
    /* SkipFilter3: :1648-1651 */
    while ((m < cc->nb_elem / NB_UC_PER_SKP) && ((hamming_weight += cc->BF_filter2[size_filter2_n_skip + m]) < posFilter2)) m++;
    TOUCH(m + 1);
    if (hamming_weight >= posFilter2) hamming_weight -= cc->BF_filter2[size_filter2_n_skip + m];

    for (k = m * (NB_UC_PER_SKP / 8); k < nb_cell_3rdlist; k++) { /* :1656-1676 */
        TOUCH(1);
        if ((sum = hamming_weight + POP8[cc->extra_filter3[k]]) >= posFilter2) {
            word_tmp = cc->extra_filter3[k];
            int size_word = 7;
            if (k == nb_cell_3rdlist - 1) size_word = (cc->nb_elem - 1) % 8;
            for (j = 0; j <= size_word; j++, word_tmp >>= 1) {
                if ((hamming_weight += (word_tmp & 1)) == posFilter2) { pos_extra_tmp = k * 8 + j; j++; break; }
            }
            goto MATCH;
        } else hamming_weight = sum;
    }
    if (k == nb_cell_3rdlist) k--;
MATCH:
    if (pos_extra_tmp == INT_MAX) pos_extra_tmp = cc->nb_elem;
    /* cluster length: following zeros, :1682-1688 */
    for (sum = k * 8 + j, word_tmp = (uint8_t)(cc->extra_filter3[k] >> j); sum < cc->nb_elem; sum++, word_tmp >>= 1) {
        if (sum % 8 == 0) { word_tmp = cc->extra_filter3[sum / 8]; TOUCH(1); }
        if (word_tmp & 1) break;
        hw0++;
    }
    *pos_extra_filter3 = pos_extra_tmp;
    *hamming_weight_0 = hw0;
}

/* findCluster, level_min==0 branch: src/presenceNode.c:1690-1812.  Same rank in filter2 / SkipFilter3 start as above, then the walk over
 * children_type.  Out: the reference's cpt_node_return, pos_extra_filter3, hamming_weight_0 and res->{pos_children, count_children,
 * count_nodes}. */
static void findCluster_lm0(const orc_cc *cc, int pos_filter2, int *cpt_node_return, int *pos_extra_filter3, int *hamming_weight_0,
                            int *r_pos_children, int *r_count_children, int *r_count_nodes) {
    int size_bf = cc->type >> 7, type = (cc->type >> 6) & 1;
    int s = (cc->type >> 1) & 0x1f, p = NB_CHAR_SUF_PREF * 2 - s;
    int size_filter2 = size_bf + (1 << p) / 8;
    int size_filter2_n_skip = size_filter2;
    int skip_filter2 = (1 << p) / NB_UC_PER_SKP;
    if (cc->nb_elem >= TRESH_SUF_PREF) size_filter2_n_skip += skip_filter2;
    int nb_skp = CEIL((int)cc->nb_elem, NB_UC_PER_SKP);
    int m = 0, hamming_weight = 0, k = size_bf + pos_filter2 / 8, cnt = 0;
    int pos_extra_tmp = INT_MAX, hw0 = 0, posFilter2;
    uint8_t word_tmp;
#define LM0_FLAG(pos) ((cc->extra_filter3[(pos) >> 3] >> ((pos) & 7)) & 1)

    if (cc->nb_elem >= TRESH_SUF_PREF) { /* SkipFilter2: :1619-1630 */
        int skip_posfilter2 = (pos_filter2 / NB_UC_PER_SKP < skip_filter2 ? pos_filter2 / NB_UC_PER_SKP : skip_filter2) + size_filter2;
        cnt = size_filter2;
        while (cnt < skip_posfilter2) { hamming_weight += cc->BF_filter2[cnt]; cnt++; }
        TOUCH(skip_posfilter2 - size_filter2);
        cnt -= size_filter2;
    }
    hamming_weight += popcnt_8_par(cc->BF_filter2, size_bf + cnt * (NB_UC_PER_SKP / 8), k);
    word_tmp = cc->BF_filter2[k];
    for (k = 0; k <= pos_filter2 % 8; k++, word_tmp >>= 1) hamming_weight += word_tmp & 1;
    posFilter2 = hamming_weight;
    hamming_weight = 0;
    while ((m < cc->nb_elem / NB_UC_PER_SKP) && ((hamming_weight += cc->BF_filter2[size_filter2_n_skip + m]) < posFilter2)) m++; /* :1648-1651 */
    TOUCH(m + 1);
    if (hamming_weight >= posFilter2) hamming_weight -= cc->BF_filter2[size_filter2_n_skip + m];

    /* :1700-1712 -- k is a multiple of 128, i.e. a bucket start: count_Nodes_Children over [nb_elem_in_pv, k) is empty */
    k = m * NB_UC_PER_SKP;
    int pos_children = k / NB_UC_PER_SKP, nb_elem_in_pv = pos_children * NB_UC_PER_SKP;
    int cpt_pv = 0, cpt_node = 0, end, it;
    if (nb_elem_in_pv > cc->nb_elem - nb_elem_in_pv) cpt_node += cc->nb_Node_children - count_nodes(cc, nb_elem_in_pv, cc->nb_elem, type);
    else cpt_node += count_nodes(cc, 0, nb_elem_in_pv, type);
    const int cpt_node_return_tmp = cpt_node;
    nb_elem_in_pv = 0;
    it = k - pos_children * NB_UC_PER_SKP;
    while (pos_children < nb_skp) { /* :1714-1752 */
        end = pos_children == nb_skp - 1 ? cc->nb_elem - pos_children * NB_UC_PER_SKP : NB_UC_PER_SKP;
        while (it < end) {
            TOUCH(1); /* getNbElts: one children_type byte */
            if ((nb_elem_in_pv = getNbElts(cc, k, type)) == 0) {
                TOUCH(2); /* children_Node_container[cpt_node].UC_array.nb_children (uint16) */
                hamming_weight += LM0_FLAG(k);
                cpt_node++;
            } else {
                TOUCH(1); /* bit 7 of the last suffix byte of the group's first row */
                hamming_weight += LM0_FLAG(k);
                cpt_pv += nb_elem_in_pv;
            }
            if (hamming_weight == posFilter2) { pos_extra_tmp = k; goto MATCH2; }
            k++;
            it++;
        }
        it = 0;
        cpt_pv = 0;
        pos_children++;
    }
    if (pos_extra_tmp == INT_MAX) pos_extra_tmp = cc->nb_elem;
MATCH2:
    *r_pos_children = pos_children;
    *r_count_children = cpt_pv - nb_elem_in_pv;
    *r_count_nodes = cpt_node - (nb_elem_in_pv == 0);
    if (pos_children < nb_skp) { /* cluster length: :1763-1810 */
        it++;
        k++;
        end = pos_children == nb_skp - 1 ? cc->nb_elem - pos_children * NB_UC_PER_SKP : NB_UC_PER_SKP;
        if (it >= end) { it = 0; cpt_pv = 0; pos_children++; }
        while (pos_children < nb_skp) {
            end = pos_children == nb_skp - 1 ? cc->nb_elem - pos_children * NB_UC_PER_SKP : NB_UC_PER_SKP;
            while (it < end) {
                TOUCH(1);
                if ((nb_elem_in_pv = getNbElts(cc, k, type)) == 0) {
                    TOUCH(2);
                    if (LM0_FLAG(k)) goto OUT_LOOP;
                    hw0 += 1;
                    cpt_node++;
                } else {
                    TOUCH(1);
                    if (LM0_FLAG(k)) goto OUT_LOOP;
                    hw0 += 1;
                    cpt_pv += nb_elem_in_pv;
                }
                k++;
                it++;
            }
            it = 0;
            cpt_pv = 0;
            pos_children++;
        }
    }
OUT_LOOP:
    *cpt_node_return = cpt_node_return_tmp;
    *pos_extra_filter3 = pos_extra_tmp;
    *hamming_weight_0 = hw0;
#undef LM0_FLAG
}

typedef struct {
    int found;
    const uint8_t *annot;
    int size_annot;
} orc_res;

/* binary_search_UC (src/UC.c:81-124) on frozen rows */
static int binary_search_rows(const uint8_t *rows, int size_annot, int pos_start, int pos_end, const uint8_t *suf, int nbs, uint8_t mask) {
    int imin = pos_start, imax = pos_end, size_line = nbs + size_annot;
    if (mask == 0xff) {
        while (imin < imax) {
            int imid = imin + (imax - imin) / 2;
            TOUCH(nbs);
            if (memcmp(rows + (size_t)imid * size_line, suf, (size_t)nbs) < 0) imin = imid + 1;
            else imax = imid;
        }
    } else {
        while (imin < imax) {
            int imid = imin + (imax - imin) / 2;
            TOUCH(nbs);
            int cmp = memcmp(rows + (size_t)imid * size_line, suf, (size_t)(nbs - 1));
            if (cmp < 0) imin = imid + 1;
            else if (cmp == 0 && (rows[(size_t)imid * size_line + nbs - 1] & mask) < suf[nbs - 1]) imin = imid + 1;
            else imax = imid;
        }
    }
    return imin;
}

/* isKmerPresent (src/presenceNode.c:1823-1921) with presenceKmer (:1284-1576) inlined per level */
static void is_kmer_present(const orc_bft *t, const uint8_t *kmer, orc_res *res) {
    uint8_t kt[40];
    const orc_node *node = &t->root;
    int i = t->k;
    memcpy(kt, kmer, (size_t)nb_bytes(i));
    res->found = 0;
    res->annot = NULL;
    res->size_annot = 0;

    for (;;) {
        LVL_VISITED();
        /* presenceKmer :1327-1350 */
        uint8_t sub0 = REV[kt[0]], sub1 = REV[kt[1]], sub2 = REV[kt[2]] & 0xc0;
        uint32_t substring_prefix = ((uint32_t)sub0 << 8) | sub1;
        uint16_t hash1_v = (uint16_t)(t->hash_v[(substring_prefix & 0x3fff) * 2] % MODULO_HASH);
        uint16_t hash2_v = (uint16_t)(t->hash_v[(substring_prefix & 0x3fff) * 2 + 1] % MODULO_HASH);
        TOUCH(16);
        uint16_t h1m = hash1_v % 8, h2m = hash2_v % 8;
        hash1_v /= 8;
        hash2_v /= 8;

        const orc_cc *cc = NULL;
        for (int c = 0; c < node->ncc; c++) { /* :1353-1362 */
            const orc_cc *q = &node->ccs[c];
            CC_SCANNED();
            TOUCH(1);
            if ((q->BF_filter2[hash1_v] & (1u << h1m)) == 0) continue;
            TOUCH(1);
            if ((q->BF_filter2[hash2_v] & (1u << h2m)) == 0) continue;
            cc = q;
            break;
        }

        if (cc) {
            /* rotation :1367-1371 */
            sub0 = (uint8_t)((substring_prefix >> 6) & 0xff);
            uint8_t nsub1 = (uint8_t)((substring_prefix << 2) | (sub2 >> 6));
            sub2 = (uint8_t)((substring_prefix >> 8) & 0xc0);
            sub1 = nsub1;
            int size_bf = cc->type >> 7, type = (cc->type >> 6) & 1;
            int s = (cc->type >> 1) & 0x1f, p = 18 - s;
            int posFilter2 = p == 10 ? (((int)sub0) << 2) | (sub1 >> 6) : (((int)sub0) << 6) | (sub1 >> 2);
            TOUCH(1);
            if ((cc->BF_filter2[size_bf + posFilter2 / 8] & (1u << (posFilter2 % 8))) == 0) return; /* :1548 */

            int pos_extra, hw0, cpt_node_tmp = -1, r_pos_children = 0, r_count_children = 0, r_count_nodes = 0;
            const int lm = level_min_of(t, i);
            if (lm) findCluster(cc, posFilter2, &pos_extra, &hw0);
            else findCluster_lm0(cc, posFilter2, &cpt_node_tmp, &pos_extra, &hw0, &r_pos_children, &r_count_children, &r_count_nodes);
            if (pos_extra >= cc->nb_elem) return;
            int imin = pos_extra, imax = pos_extra + hw0, hit = 0;
            if (s == 8) { /* :1399-1410 */
                uint8_t suffix = (uint8_t)((sub1 << 2) | (sub2 >> 6));
                while (imin < imax) { int imid = (imin + imax) / 2; TOUCH(1); if (cc->filter3[imid] < suffix) imin = imid + 1; else imax = imid; }
                TOUCH(1);
                hit = cc->filter3[imin] == suffix;
            } else { /* :1472-1489 */
                uint8_t suffix = (uint8_t)(((sub1 & 0x3) << 2) | (sub2 >> 6)), tmp;
                while (imin < imax) {
                    int imid = (imin + imax) / 2;
                    TOUCH(1);
                    tmp = (imid & 1) ? cc->filter3[imid / 2] >> 4 : cc->filter3[imid / 2] & 0xf;
                    if (tmp < suffix) imin = imid + 1; else imax = imid;
                }
                TOUCH(1);
                tmp = (imin & 1) ? cc->filter3[imin / 2] >> 4 : cc->filter3[imin / 2] & 0xf;
                hit = tmp == suffix;
            }
            if (!hit) return;

            if (i == NB_CHAR_SUF_PREF) { /* leaf: :1453-1463 */
                int bucket = imin / NB_UC_PER_SKP, psb = imin % NB_UC_PER_SKP;
                const orc_uc *uc = &cc->children[bucket];
                res->found = 1;
                res->annot = uc->suffixes + (size_t)psb * uc->size_annot;
                res->size_annot = uc->size_annot;
                return;
            }
            strip9(kt, i); /* isKmerPresent :1853-1861 */
            if (is_child(cc, imin, type)) { /* :1419-1440 */
                int bucket = imin / NB_UC_PER_SKP;
                const orc_uc *uc = &cc->children[bucket];
                int psb = bucket * NB_UC_PER_SKP;
                int nb_elem = cc->nb_elem - psb < NB_UC_PER_SKP ? cc->nb_elem - psb : NB_UC_PER_SKP;
                if (!lm && r_pos_children == bucket) { /* :1425-1430: the walk of findCluster already counted up to the cluster start */
                    if (imin - pos_extra > psb + nb_elem - imin) psb = uc->nb_children - count_children(cc, imin, psb + nb_elem, type);
                    else psb = r_count_children + count_children(cc, pos_extra, imin, type);
                } else if (imin - psb > psb + nb_elem - imin) psb = uc->nb_children - count_children(cc, imin, psb + nb_elem, type);
                else psb = count_children(cc, psb, imin, type);
                /* isKmerPresent :1874-1915 */
                int nb_elt = getNbElts(cc, imin, type);
                int nb_cell = nbm1_bytes(i), size_line = nb_cell + uc->size_annot;
                if (nb_elt == 0) return;
                if (i == 45 || i == 81 || i == 117) {
                    int j = binary_search_rows(uc->suffixes, uc->size_annot, psb, psb + nb_elt - 1, kt, nb_cell, 0xff);
                    TOUCH(nb_cell);
                    if (memcmp(uc->suffixes + (size_t)j * size_line, kt, (size_t)nb_cell) == 0) {
                        res->found = 1; res->annot = uc->suffixes + (size_t)j * size_line + nb_cell; res->size_annot = uc->size_annot;
                    }
                } else {
                    int j = binary_search_rows(uc->suffixes, uc->size_annot, psb, psb + nb_elt - 1, kt, nb_cell, 0x7f);
                    TOUCH(nb_cell);
                    if (memcmp(uc->suffixes + (size_t)j * size_line, kt, (size_t)(nb_cell - 1)) == 0 &&
                        (uc->suffixes[(size_t)j * size_line + nb_cell - 1] & 0x7f) == kt[nb_cell - 1]) {
                        res->found = 1; res->annot = uc->suffixes + (size_t)j * size_line + nb_cell; res->size_annot = uc->size_annot;
                    }
                }
                return;
            }
            /* child Node: :1443-1448 */
            int idx;
            if (cpt_node_tmp != -1) idx = r_count_nodes + count_nodes(cc, pos_extra, imin + 1, type) - 1;
            else if (imin < cc->nb_elem - imin) idx = count_nodes(cc, 0, imin, type);
            else idx = cc->nb_Node_children - count_nodes(cc, imin, cc->nb_elem, type);
            node = cc->children_nodes[idx];
            i -= NB_CHAR_SUF_PREF;
            continue; /* isKmerPresent :1867 recursion */
        }

        /* node UC: :1554-1573 */
        if (node->fuc.suffixes != NULL) {
            int nb_cell = nb_bytes(i), size_line = nb_cell + node->fuc.size_annot;
            int pos = binary_search_rows(node->fuc.suffixes, node->fuc.size_annot, 0, node->fuc.nb_children - 1, kt, nb_cell, 0xff);
            TOUCH(nb_cell);
            if (memcmp(node->fuc.suffixes + (size_t)pos * size_line, kt, (size_t)nb_cell) == 0) {
                res->found = 1; res->annot = node->fuc.suffixes + (size_t)pos * size_line + nb_cell; res->size_annot = node->fuc.size_annot;
            }
        }
        return;
    }
}

long orc_query_presence(orc_bft *t, const uint8_t *kmers, long n, uint8_t *present_bits) {
    orc_freeze(t);
    int nb = nb_bytes(t->k);
    long cnt = 0;
    memset(present_bits, 0, (size_t)CEIL(n, 8));
    orc_res res;
    for (long a = 0; a < n; a++) {
        is_kmer_present(t, kmers + (size_t)a * nb, &res);
        if (res.found) { present_bits[a >> 3] |= (uint8_t)(1u << (a & 7)); cnt++; }
    }
    return cnt;
}

typedef struct { orc_bft *t; const uint8_t *kmers; long a, b; uint8_t *bits; long cnt; } mt_job;
static void *mt_worker(void *arg) {
    mt_job *j = arg;
    int nb = nb_bytes(j->t->k);
    orc_res res;
    long cnt = 0;
    for (long a = j->a; a < j->b; a++) {
        is_kmer_present(j->t, j->kmers + (size_t)a * nb, &res);
        if (res.found) { j->bits[a >> 3] |= (uint8_t)(1u << (a & 7)); cnt++; }
    }
    j->cnt = cnt;
    return NULL;
}

long orc_query_presence_mt(orc_bft *t, const uint8_t *kmers, long n, uint8_t *present_bits, int nthreads) {
    orc_freeze(t);
    if (nthreads < 1) nthreads = 1;
    memset(present_bits, 0, (size_t)CEIL(n, 8));
    pthread_t *th = xmalloc((size_t)nthreads * sizeof(pthread_t));
    mt_job *jobs = xcalloc((size_t)nthreads, sizeof(mt_job));
    long per = CEIL(CEIL(n, nthreads), 8) * 8; /* slices are byte aligned in the bitmap */
    int used = 0;
    for (int q = 0; q < nthreads; q++) {
        long a = (long)q * per, b = a + per < n ? a + per : n;
        if (a >= n) break;
        jobs[q] = (mt_job){t, kmers, a, b, present_bits, 0};
        pthread_create(&th[q], NULL, mt_worker, &jobs[q]);
        used++;
    }
    long cnt = 0;
    for (int q = 0; q < used; q++) { pthread_join(th[q], NULL); cnt += jobs[q].cnt; }
    free(th);
    free(jobs);
    return cnt;
}

long orc_query_presence_count(orc_bft *t, const uint8_t *kmers, long n, uint8_t *present_bits, uint64_t *bytes_out) {
#ifdef ORC_COUNT
    g_touch = g_ccs = g_lvls = 0;
#endif
    long c = orc_query_presence(t, kmers, n, present_bits);
#ifdef ORC_COUNT
    bytes_out[0] = g_touch; bytes_out[1] = g_ccs; bytes_out[2] = g_lvls;
#else
    bytes_out[0] = bytes_out[1] = bytes_out[2] = 0;
#endif
    return c;
}

long orc_query_colors(orc_bft *t, const uint8_t *kmers, long n, uint8_t *present_bits, uint64_t *offsets, uint32_t *ids, long ids_cap) {
    orc_freeze(t);
    g_comp = t->celems;
    g_ncomp = t->ncelems;
    int nb = nb_bytes(t->k);
    long total = 0;
    memset(present_bits, 0, (size_t)CEIL(n, 8));
    orc_res res;
    for (long a = 0; a < n; a++) {
        offsets[a] = (uint64_t)total;
        is_kmer_present(t, kmers + (size_t)a * nb, &res);
        if (!res.found) continue;
        present_bits[a >> 3] |= (uint8_t)(1u << (a & 7));
        long room = ids_cap - total;
        int cnt = decode_any(res.annot, res.size_annot, room > 0 ? ids + total : NULL, room > 0 ? (int)(room > INT_MAX ? INT_MAX : room) : 0);
        total += cnt;
    }
    offsets[n] = (uint64_t)total;
    return total;
}

/* ------------------------------------------------------------------ */
/* stats / extraction                                                 */
/* ------------------------------------------------------------------ */

static void stats_node(const orc_node *nd, int i, long *o) {
    o[0]++;
    o[1] += nd->ncc;
    o[5] += nd->uc_n;
    o[2] += nd->uc_n;
    if (nd->ncc > o[9]) o[9] = nd->ncc;
    for (int c = 0; c < nd->ncc; c++) {
        const orc_cc *cc = &nd->ccs[c];
        o[7] += cc->n;
        if (cc->n >= TRESH_SUF_PREF) o[8]++;
        for (int j = 0; j < cc->n; j++) {
            if (i == 9) o[2]++;
            else if (cc->prefs[j].cnt == 0) { o[6]++; stats_node(cc->prefs[j].c.node, i - 9, o); }
            else o[2] += cc->prefs[j].cnt;
        }
    }
}

void orc_stats(orc_bft *t, long *out) {
    memset(out, 0, 10 * sizeof(long));
    stats_node(&t->root, t->k, out);
    out[3] = t->root.ncc;
    out[4] = t->root.uc_n;
}

int orc_root_cc_sizes(orc_bft *t, int *out, int cap) {
    for (int c = 0; c < t->root.ncc && c < cap; c++) out[c] = t->root.ccs[c].n;
    return t->root.ncc;
}

/* write nt codes of a packed suffix (len nt) at nt offset `at` of the k-mer being rebuilt */
static void put_nts(uint8_t *kmer, int at, const uint8_t *packed, int len) {
    for (int j = 0; j < len; j++) {
        uint8_t code = (packed[j / 4] >> (2 * (j % 4))) & 3;
        int q = at + j;
        kmer[q / 4] = (uint8_t)((kmer[q / 4] & ~(3u << (2 * (q % 4)))) | (code << (2 * (q % 4))));
    }
}
/* rotated prefix r = n2..n9,n1 -> nt codes n1..n9 at nt offset `at` */
static void put_prefix(uint8_t *kmer, int at, uint32_t r) {
    uint8_t nts[9];
    nts[0] = r & 3;
    for (int j = 1; j < 9; j++) nts[j] = (r >> (2 * (9 - j))) & 3;
    for (int j = 0; j < 9; j++) {
        int q = at + j;
        kmer[q / 4] = (uint8_t)((kmer[q / 4] & ~(3u << (2 * (q % 4)))) | (nts[j] << (2 * (q % 4))));
    }
}

static long extract_node(orc_bft *t, const orc_node *nd, int i, uint8_t *cur, uint8_t *kout, uint32_t *csout, long pos) {
    int nbk = nb_bytes(t->k), at = t->k - i;
    for (int a = 0; a < nd->uc_n; a++) {
        const uint8_t *row = nd->uc + (size_t)a * (nb_bytes(i) + 4);
        if (kout) { put_nts(cur, at, row, i); memcpy(kout + (size_t)pos * nbk, cur, (size_t)nbk); }
        if (csout) csout[pos] = row_cs(row, nb_bytes(i));
        pos++;
    }
    for (int c = 0; c < nd->ncc; c++) {
        const orc_cc *cc = &nd->ccs[c];
        for (int j = 0; j < cc->n; j++) {
            const orc_pref *p = &cc->prefs[j];
            put_prefix(cur, at, p->r);
            if (i == 9) {
                if (kout) memcpy(kout + (size_t)pos * nbk, cur, (size_t)nbk);
                if (csout) csout[pos] = p->c.cs;
                pos++;
            } else if (p->cnt == 0) pos = extract_node(t, p->c.node, i - 9, cur, kout, csout, pos);
            else {
                int nbs = nbm1_bytes(i);
                for (int q = 0; q < p->cnt; q++) {
                    const uint8_t *row = p->c.rows + (size_t)q * (nbs + 4);
                    if (kout) { put_nts(cur, at + 9, row, i - 9); memcpy(kout + (size_t)pos * nbk, cur, (size_t)nbk); }
                    if (csout) csout[pos] = row_cs(row, nbs);
                    pos++;
                }
            }
        }
    }
    return pos;
}

long orc_extract(orc_bft *t, uint8_t *kmers_out, uint32_t *cs_out) {
    uint8_t cur[40];
    memset(cur, 0, sizeof(cur));
    return extract_node(t, &t->root, t->k, cur, kmers_out, cs_out, 0);
}

/* ------------------------------------------------------------------ */
/* .bft writer / reader (src/write_to_disk.c)                         */
/* ------------------------------------------------------------------ */


static void wr(FILE *f, const void *p, size_t n) { if (n && fwrite(p, 1, n, f) != n) { fprintf(stderr, "oracle: write error\n"); exit(1); } }
static void wr_u16(FILE *f, uint16_t v) { wr(f, &v, 2); }
static void wr_u32(FILE *f, uint32_t v) { wr(f, &v, 4); }
static void wr_i32(FILE *f, int32_t v) { wr(f, &v, 4); }

static int g_write_ext = 0;
/* write_UC, uncompressed branch (src/write_to_disk.c:107-213).  With g_write_ext the last annotation byte of the rows
 * that use all size_annot bytes goes to the extended-annotation table (src/UC.c:321-521) and size_annot shrinks by
 * one: the shape reference-built files have after annotations grew by one byte. */
static void write_uc(FILE *f, const uint8_t *data, int size_annot, int nbs, int count, int header_field, int with_header) {
    if (with_header) wr_u16(f, (uint16_t)header_field);
    if (!count) return;
    int stride = nbs + size_annot, next = 0;
    if (g_write_ext && size_annot >= 2)
        for (int q = 0; q < count; q++) next += data[(size_t)q * stride + stride - 1] != 0;
    if (!next) {
        wr_u16(f, 0);          /* nb_extended_annot */
        wr_i32(f, size_annot); /* UC_SIZE_ANNOT_T */
        wr(f, data, (size_t)count * stride);
        return;
    }
    wr_u16(f, (uint16_t)next);
    wr_i32(f, size_annot - 1);
    for (int q = 0; q < count; q++) wr(f, data + (size_t)q * stride, (size_t)stride - 1);
    int prev = 0;
    for (int q = 0; q < count; q++) {
        uint8_t last = data[(size_t)q * stride + stride - 1];
        if (!last) continue;
        uint8_t e[3] = {(uint8_t)((q - prev) >> 8), (uint8_t)((q - prev) & 0xff), last};
        wr(f, e, 3);
        prev = q;
    }
}

static void write_node(FILE *f, orc_bft *t, orc_node *nd, int i, int flag);

/* write_CC (src/write_to_disk.c:215-258) */
static void write_cc(FILE *f, orc_bft *t, orc_cc *cc, int i) {
    int s = (cc->type >> 1) & 0x1f, p = 18 - s, n = cc->nb_elem, lm = level_min_of(t, i);
    wr_u16(f, cc->type);
    wr_u16(f, cc->nb_elem);
    wr_u16(f, cc->nb_Node_children);
    wr(f, cc->BF_filter2 + SIZE_BF_BYTES, (size_t)(1 << p) / 8);
    wr(f, cc->filter3, (size_t)(s == 8 ? n : CEIL(n, 2)));
    if (lm) wr(f, cc->extra_filter3, (size_t)CEIL(n, 8));
    int nbk = CEIL(n, NB_UC_PER_SKP);
    if (i != 9) {
        int type = (cc->type >> 6) & 1, nbs = nbm1_bytes(i);
        wr(f, cc->children_type, (size_t)(type ? n : CEIL(n, 2)));
        for (int b = 0; b < nbk; b++) {
            orc_uc *uc = &cc->children[b];
            int stride = nbs + uc->size_annot;
            uint8_t *tmp = xmalloc((size_t)uc->nb_children * stride + 1);
            if (uc->nb_children) memcpy(tmp, uc->suffixes, (size_t)uc->nb_children * stride);
            if (!lm) { /* cluster-start flags live in bit 7 of the first row's last suffix byte (src/CC.c:349-352) */
                int row = 0, j1 = b * 128 + 128 < n ? b * 128 + 128 : n;
                for (int j = b * 128; j < j1; j++) {
                    int cnt = getNbElts(cc, j, type);
                    if (cnt && (cc->extra_filter3[j >> 3] & (1u << (j & 7)))) tmp[(size_t)row * stride + nbs - 1] |= 0x80;
                    row += cnt;
                }
            }
            write_uc(f, tmp, uc->size_annot, nbs, uc->nb_children, uc->nb_children, 1);
            free(tmp);
        }
    } else {
        for (int b = 0; b < nbk; b++) {
            int cnt = b != nbk - 1 ? NB_UC_PER_SKP : n - b * NB_UC_PER_SKP;
            write_uc(f, cc->children[b].suffixes, cc->children[b].size_annot, 0, cnt, 0, 0);
        }
    }
    if (i != 9) {
        int type = (cc->type >> 6) & 1, kn = 0;
        for (int j = 0; j < n; j++)
            if (getNbElts(cc, j, type) == 0) {
                int flag = !lm && (cc->extra_filter3[j >> 3] & (1u << (j & 7))) ? 1 : 0; /* src/insertNode.c:308-311 */
                write_node(f, t, cc->children_nodes[kn++], i - 9, flag);
            }
    }
}

/* write_Node (src/write_to_disk.c:84-105) */
static void write_node(FILE *f, orc_bft *t, orc_node *nd, int i, int flag) {
    write_uc(f, nd->fuc.suffixes, nd->fuc.size_annot, nb_bytes(i), nd->uc_n, (nd->uc_n << 1) | flag, 1);
    wr_u32(f, (uint32_t)nd->ncc);
    for (int c = 0; c < nd->ncc; c++) write_cc(f, t, &nd->ccs[c], i);
}

int orc_write_bft(orc_bft *t, const char *path, int nb_genomes) {
    orc_freeze(t);
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    g_write_ext = t->ext_on;
    wr_i32(f, t->comp_on ? t->ncelems : 0); /* length_comp_set_colors */
    if (t->comp_on)
        for (int e = 0; e < t->ncelems; e++) { /* src/write_to_disk.c:36-61 */
            int64_t cnt = e ? t->celems[e].last_index - t->celems[e - 1].last_index : t->celems[e].last_index + 1;
            wr(f, &t->celems[e].last_index, 8);
            wr_i32(f, t->celems[e].size_annot);
            wr(f, t->celems[e].bytes, (size_t)(cnt * t->celems[e].size_annot));
        }
    wr_i32(f, t->r1); wr_i32(f, t->r2); wr_i32(f, 0 /* treshold_compression */); wr_i32(f, nb_genomes); wr_i32(f, t->k);
    uint8_t comp = 0;
    wr(f, &comp, 1);
    for (int g = 0; g < nb_genomes; g++) {
        char name[64];
        snprintf(name, sizeof(name), "genome_%d", g);
        wr_u16(f, (uint16_t)(strlen(name) + 1));
        wr(f, name, strlen(name) + 1);
    }
    for (int i = 9; i <= t->k; i += 9) { /* src/write_to_disk.c:76-84 */
        wr_i32(f, NB_UC_PER_SKP); wr_i32(f, NB_UC_PER_SKP); wr_i32(f, NB_UC_PER_SKP); wr_i32(f, NB_KMERS_PER_UC);
        wr_i32(f, level_min_of(t, i)); wr_i32(f, MODULO_HASH); wr_i32(f, TRESH_SUF_PREF);
    }
    write_node(f, t, &t->root, t->k, 0);
    fclose(f);
    g_write_ext = 0;
    return 0;
}

/* comp_annotation (src/annotation.c:1777-1838): delta-code the stored ids of a mode-1 / mode-2 annotation */
static int comp_entry(const uint32_t *ids, int n, uint8_t *out, int cap) {
    uint8_t plain[8192];
    int sz = orc_annot_encode(ids, n, plain, (int)sizeof(plain));
    if (sz < 0) return -1;
    int mode = plain[0] & 3;
    if (mode == 0) { if (sz > cap) return -1; memcpy(out, plain, (size_t)sz); return sz; }
    uint32_t st[8192];
    int ns = 0;
    if (mode == 2) for (int a = 0; a < n && ns < 8192; a++) st[ns++] = ids[a];
    else for (int a = 0; a < n && ns + 1 < 8192;) { int b = a; while (b + 1 < n && ids[b + 1] == ids[b] + 1) b++; st[ns++] = ids[a]; st[ns++] = ids[b]; a = b + 1; }
    for (int q = ns - 1; q > 0; q--) st[q] -= st[q - 1];
    int o = 0;
    for (int q = 0; q < ns; q++) { if (o + 6 > cap) return -1; o += put_id(out + o, st[q], mode == 2 ? 0x2 : 0x1, mode == 2 ? 0x1 : 0x2); }
    return o;
}

typedef struct { int size; uint32_t cs; } comp_ord;
static int cmp_comp_ord(const void *a, const void *b) {
    const comp_ord *x = a, *y = b;
    if (x->size != y->size) return x->size - y->size;
    return x->cs < y->cs ? -1 : x->cs > y->cs;
}

void orc_set_annotation_modes(orc_bft *t, int comp_on, int ext_on) {
    for (int e = 0; e < t->ncelems; e++) free(t->celems[e].bytes);
    free(t->celems); free(t->cs_pos);
    t->celems = NULL; t->cs_pos = NULL; t->ncelems = 0;
    t->comp_on = comp_on; t->ext_on = ext_on; t->dirty = 1;
    if (!comp_on || t->cs_n <= 1) { t->comp_on = 0; return; }
    long ncs = t->cs_n - 1; /* set 0 is the empty set */
    uint8_t **enc = xcalloc((size_t)ncs, sizeof(uint8_t *));
    comp_ord *ord = xmalloc((size_t)ncs * sizeof(comp_ord));
    for (long c = 0; c < ncs; c++) {
        long a = t->cs_off[c + 1], b = t->cs_off[c + 2];
        uint8_t tmp[16384];
        int sz = comp_entry(t->cs_ids + a, (int)(b - a), tmp, (int)sizeof(tmp));
        if (sz < 0) { fprintf(stderr, "oracle: colour set too large for the test compressor\n"); exit(1); }
        enc[c] = xmalloc((size_t)sz);
        memcpy(enc[c], tmp, (size_t)sz);
        ord[c].size = sz; ord[c].cs = (uint32_t)(c + 1);
    }
    qsort(ord, (size_t)ncs, sizeof(comp_ord), cmp_comp_ord);
    t->cs_pos = xcalloc((size_t)t->cs_n, 4);
    t->celems = xcalloc((size_t)ncs, sizeof(comp_elem));
    for (long q = 0; q < ncs;) {
        long q2 = q;
        while (q2 + 1 < ncs && ord[q2 + 1].size == ord[q].size) q2++;
        comp_elem *el = &t->celems[t->ncelems++];
        el->size_annot = ord[q].size;
        el->last_index = q2;
        el->bytes = xmalloc((size_t)(q2 - q + 1) * ord[q].size);
        for (long z = q; z <= q2; z++) { memcpy(el->bytes + (size_t)(z - q) * ord[q].size, enc[ord[z].cs - 1], (size_t)ord[q].size); t->cs_pos[ord[z].cs] = (uint32_t)z; }
        q = q2 + 1;
    }
    for (long c = 0; c < ncs; c++) free(enc[c]);
    free(enc); free(ord);
}

typedef struct { FILE *f; int err; int nb_genomes; } rd_ctx;
static void rd(rd_ctx *c, void *p, size_t n) { if (n && fread(p, 1, n, c->f) != n) c->err = 1; }
static uint16_t rd_u16(rd_ctx *c) { uint16_t v = 0; rd(c, &v, 2); return v; }
static uint32_t rd_u32(rd_ctx *c) { uint32_t v = 0; rd(c, &v, 4); return v; }
static int32_t rd_i32(rd_ctx *c) { int32_t v = 0; rd(c, &v, 4); return v; }

/* Loader only: annotation bytes -> colour set of the model.  The same few 10^6 byte strings label the 10^7..10^8 rows of a pan-genome
 * file, so the decode + cs_add walk runs once per distinct string (exact: a hit compares the bytes). */
typedef struct { uint64_t h; uint32_t cs; int len; uint8_t *bytes; } annot_memo;
static annot_memo *g_memo = NULL;
static size_t g_memo_cap = 0, g_memo_n = 0;
static void memo_clear(void) {
    for (size_t i = 0; i < g_memo_cap; i++) free(g_memo[i].bytes);
    free(g_memo);
    g_memo = NULL;
    g_memo_cap = g_memo_n = 0;
}
static uint32_t cs_from_annot_slow(orc_bft *t, const uint8_t *annot, int size) {
    uint32_t ids[4096];
    int n = decode_any(annot, size, ids, 4096);
    if (n > 4096) n = 4096;
    if (n == 0) return 0;
    /* the whole list at once (a chain of cs_add would create -- and copy -- every prefix of it: quadratic, 20 s of a 50 s load of the
     * 100-genome index); the byte memo above makes equal annotations share the set */
    return cs_new(t, ids, n, 0, 0);
}
static uint32_t cs_from_annot(orc_bft *t, const uint8_t *annot, int size) {
    while (size > 1 && annot[size - 1] == 0) size--; /* rows are padded to the UC's width with zero bytes */
    if (g_memo_n * 2 >= g_memo_cap) {
        const size_t ncap = g_memo_cap ? g_memo_cap * 2 : (1u << 16);
        annot_memo *nm = xcalloc(ncap, sizeof(annot_memo));
        for (size_t i = 0; i < g_memo_cap; i++)
            if (g_memo[i].bytes) {
                size_t j = (size_t)g_memo[i].h & (ncap - 1);
                while (nm[j].bytes) j = (j + 1) & (ncap - 1);
                nm[j] = g_memo[i];
            }
        free(g_memo);
        g_memo = nm;
        g_memo_cap = ncap;
    }
    const uint64_t h = orc_xxh64(annot, (size_t)size, 0x5bd1e995u);
    size_t j = (size_t)h & (g_memo_cap - 1);
    while (g_memo[j].bytes) {
        if (g_memo[j].h == h && g_memo[j].len == size && memcmp(g_memo[j].bytes, annot, (size_t)size) == 0) return g_memo[j].cs;
        j = (j + 1) & (g_memo_cap - 1);
    }
    g_memo[j].h = h;
    g_memo[j].len = size;
    g_memo[j].bytes = xmalloc((size_t)size + 1);
    memcpy(g_memo[j].bytes, annot, (size_t)size);
    g_memo[j].cs = cs_from_annot_slow(t, annot, size);
    g_memo_n++;
    return g_memo[j].cs;
}

/* read_UC uncompressed branch (src/write_to_disk.c:383-531): returns rows (nbs + size_annot each).  Rows that own an
 * entry of the extended-annotation table (3-byte entries: big-endian position delta + 1 byte, src/UC.c:501-521) get
 * that byte appended: the rows are returned one byte wider (*size_annot is the widened size). */
static uint8_t *read_uc(rd_ctx *c, int nbs, int count, int *size_annot) {
    *size_annot = 0;
    if (!count) return NULL;
    uint16_t next = rd_u16(c);
    int32_t sa = rd_i32(c);
    if (c->err || sa < 0 || sa > (1 << 20) || next == 0xffff) { c->err = 1; return NULL; }
    size_t n = (size_t)count * (nbs + sa);
    uint8_t *raw = xmalloc(n + 1);
    rd(c, raw, n);
    if (!next) { *size_annot = sa; return raw; }
    uint8_t *buf = xcalloc((size_t)count, (size_t)(nbs + sa + 1));
    for (int q = 0; q < count; q++) memcpy(buf + (size_t)q * (nbs + sa + 1), raw + (size_t)q * (nbs + sa), (size_t)(nbs + sa));
    int pos = 0;
    for (int e = 0; e < next && !c->err; e++) {
        uint8_t t3[3];
        rd(c, t3, 3);
        pos += (t3[0] << 8) | t3[1];
        if (pos >= count) { c->err = 1; break; }
        buf[(size_t)pos * (nbs + sa + 1) + nbs + sa] = t3[2];
    }
    free(raw);
    *size_annot = sa + 1;
    return buf;
}

static void load_node(rd_ctx *c, orc_bft *t, orc_node *nd, int i, int *flag_out);

static void load_cc(rd_ctx *c, orc_bft *t, orc_cc *cc, int i) {
    memset(cc, 0, sizeof(*cc));
    uint16_t type = rd_u16(c), n = rd_u16(c), nnodes = rd_u16(c);
    int s = (type >> 1) & 0x1f, p = 18 - s, tbyte = (type >> 6) & 1, lm = level_min_of(t, i);
    if (c->err || (s != 4 && s != 8)) { c->err = 1; return; }
    uint8_t *f2 = xmalloc((size_t)(1 << p) / 8), *f3 = xmalloc((size_t)(s == 8 ? n : CEIL(n, 2)) + 1), *ex = xcalloc(1, (size_t)CEIL(n, 8) + 1);
    rd(c, f2, (size_t)(1 << p) / 8);
    rd(c, f3, (size_t)(s == 8 ? n : CEIL(n, 2)));
    if (lm) rd(c, ex, (size_t)CEIL(n, 8));
    int nbk = CEIL(n, NB_UC_PER_SKP);
    uint16_t *cnts = xcalloc((size_t)n + 1, 2);
    cc->prefs = xcalloc((size_t)n + 1, sizeof(orc_pref));
    cc->n = cc->cap = n;
    if (i != 9) {
        int nbs = nbm1_bytes(i);
        uint8_t *ct = xmalloc((size_t)(tbyte ? n : CEIL(n, 2)) + 1);
        rd(c, ct, (size_t)(tbyte ? n : CEIL(n, 2)));
        for (int j = 0; j < n; j++) cnts[j] = tbyte ? ct[j] : ((j & 1) ? ct[j / 2] >> 4 : ct[j / 2] & 0xf);
        free(ct);
        for (int b = 0; b < nbk && !c->err; b++) {
            int cnt = rd_u16(c), sa;
            uint8_t *rows = read_uc(c, nbs, cnt, &sa);
            int row = 0, j1 = b * 128 + 128 < n ? b * 128 + 128 : n;
            for (int j = b * 128; j < j1 && !c->err; j++) {
                orc_pref *pf = &cc->prefs[j];
                pf->cnt = cnts[j];
                if (!cnts[j]) continue;
                if (row + cnts[j] > cnt) { c->err = 1; break; }
                pf->c.rows = xmalloc((size_t)cnts[j] * (nbs + 4));
                for (int q = 0; q < cnts[j]; q++) {
                    uint8_t *src = rows + (size_t)(row + q) * (nbs + sa), *dst = pf->c.rows + (size_t)q * (nbs + 4);
                    memcpy(dst, src, (size_t)nbs);
                    if (!lm && q == 0) { if (dst[nbs - 1] & 0x80) ex[j >> 3] |= (uint8_t)(1u << (j & 7)); }
                    if (!lm) dst[nbs - 1] &= 0x7f;
                    row_set_cs(dst, nbs, cs_from_annot(t, src + nbs, sa));
                }
                row += cnts[j];
            }
            free(rows);
        }
        for (int j = 0; j < n && !c->err; j++)
            if (!cnts[j]) {
                int flag = 0;
                cc->prefs[j].c.node = xcalloc(1, sizeof(orc_node));
                load_node(c, t, cc->prefs[j].c.node, i - 9, &flag);
                if (!lm && flag) ex[j >> 3] |= (uint8_t)(1u << (j & 7));
            }
    } else {
        for (int b = 0; b < nbk && !c->err; b++) {
            int cnt = b != nbk - 1 ? NB_UC_PER_SKP : n - b * NB_UC_PER_SKP, sa;
            uint8_t *rows = read_uc(c, 0, cnt, &sa);
            for (int q = 0; q < cnt && !c->err; q++) { cc->prefs[b * 128 + q].cnt = 1; cc->prefs[b * 128 + q].c.cs = cs_from_annot(t, rows + (size_t)q * sa, sa); }
            free(rows);
        }
    }
    (void)nnodes;
    /* prefixes: walk filter2 bits and the cluster starts, as read_CC does to rebuild the Bloom filter (:649-681) */
    int j = 0;
    for (int pu = 0; pu < (1 << p) && !c->err; pu++) {
        if (!(f2[pu >> 3] & (1u << (pu & 7)))) continue;
        int first = 1;
        while (j < n && (first || !(ex[j >> 3] & (1u << (j & 7))))) {
            uint32_t pv = s == 8 ? f3[j] : ((j & 1) ? f3[j / 2] >> 4 : f3[j / 2] & 0xf);
            uint32_t r = ((uint32_t)pu << s) | pv, key = r >> 4;
            cc->prefs[j].r = r;
            bf_set(cc->bf, t->hmod[key * 2]);
            bf_set(cc->bf, t->hmod[key * 2 + 1]);
            j++;
            first = 0;
        }
    }
    if (j != n) c->err = 1;
    free(f2); free(f3); free(ex); free(cnts);
}

static void load_node(rd_ctx *c, orc_bft *t, orc_node *nd, int i, int *flag_out) {
    memset(nd, 0, sizeof(*nd));
    uint16_t field = rd_u16(c);
    int cnt = field >> 1, sa, nbi = nb_bytes(i);
    if (flag_out) *flag_out = field & 1;
    uint8_t *rows = read_uc(c, nbi, cnt, &sa);
    if (cnt && !c->err) {
        nd->uc = xmalloc((size_t)cnt * (nbi + 4));
        nd->uc_n = cnt;
        for (int q = 0; q < cnt; q++) {
            memcpy(nd->uc + (size_t)q * (nbi + 4), rows + (size_t)q * (nbi + sa), (size_t)nbi);
            row_set_cs(nd->uc + (size_t)q * (nbi + 4), nbi, cs_from_annot(t, rows + (size_t)q * (nbi + sa) + nbi, sa));
        }
    }
    free(rows);
    uint32_t ncc = rd_u32(c);
    if (c->err || ncc > 1000000) { c->err = 1; return; }
    nd->ccs = xcalloc(ncc ? ncc : 1, sizeof(orc_cc));
    nd->ncc = (int)ncc;
    for (uint32_t q = 0; q < ncc && !c->err; q++) load_cc(c, t, &nd->ccs[q], i);
}

static long count_kmers(const orc_node *nd, int i) {
    long n = nd->uc_n;
    for (int c = 0; c < nd->ncc; c++)
        for (int j = 0; j < nd->ccs[c].n; j++) {
            const orc_pref *p = &nd->ccs[c].prefs[j];
            if (i == 9) n++;
            else if (p->cnt == 0) n += p->c.node ? count_kmers(p->c.node, i - 9) : 0;
            else n += p->cnt;
        }
    return n;
}

orc_bft *orc_load_bft(const char *path) {
    rd_ctx c = {fopen(path, "rb"), 0, 0};
    if (!c.f) return NULL;
    int lcs = rd_i32(&c);
    if (lcs < 0 || lcs > (1 << 24)) { fclose(c.f); return NULL; }
    comp_elem *celems = xcalloc((size_t)lcs + 1, sizeof(comp_elem));
    for (int e = 0; e < lcs && !c.err; e++) { /* src/write_to_disk.c:283-310 */
        rd(&c, &celems[e].last_index, 8);
        celems[e].size_annot = rd_i32(&c);
        int64_t cnt = e ? celems[e].last_index - celems[e - 1].last_index : celems[e].last_index + 1;
        if (c.err || cnt < 0 || celems[e].size_annot < 0 || cnt * celems[e].size_annot > (1LL << 32)) { c.err = 1; break; }
        celems[e].bytes = xmalloc((size_t)(cnt * celems[e].size_annot) + 1);
        rd(&c, celems[e].bytes, (size_t)(cnt * celems[e].size_annot));
    }
    g_comp = celems;
    g_ncomp = lcs;
    int r1 = rd_i32(&c), r2 = rd_i32(&c);
    (void)rd_i32(&c);
    int nbg = rd_i32(&c), k = rd_i32(&c);
    uint8_t comp = 0;
    rd(&c, &comp, 1);
    if (c.err || comp != 0 || nbg < 0) { fclose(c.f); g_comp = NULL; g_ncomp = 0; return NULL; }
    orc_bft *t = orc_create(k, r1, r2);
    if (!t) { fclose(c.f); g_comp = NULL; g_ncomp = 0; return NULL; }
    for (int g = 0; g < nbg && !c.err; g++) {
        uint16_t len = rd_u16(&c);
        char buf[70000];
        rd(&c, buf, len);
    }
    for (int i = 9; i <= k && !c.err; i += 9)
        for (int q = 0; q < 7; q++) (void)rd_i32(&c);
    if (!c.err) load_node(&c, t, &t->root, k, NULL);
    fclose(c.f);
    memo_clear();
    for (int e = 0; e < lcs; e++) free(celems[e].bytes);
    free(celems);
    g_comp = NULL;
    g_ncomp = 0;
    if (c.err) { orc_free(t); return NULL; }
    t->nkmers = count_kmers(&t->root, k);
    t->dirty = 1;
    t->nb_genomes_loaded = nbg;
    return t;
}

int orc_nb_genomes_loaded(const orc_bft *t) { return t->nb_genomes_loaded; }

/* ------------------------------------------------------------------ */
/* branching (src/branchingNode.c)                                    */
/* ------------------------------------------------------------------ */
long orc_query_branching(orc_bft *t, const uint8_t *kmers, long n, uint8_t *bits, uint8_t *counts) {
    orc_freeze(t);
    const int nb = nb_bytes(t->k), k = t->k;
    long nbr = 0;
    memset(bits, 0, (size_t)CEIL(n, 8));
    orc_res res;
    for (long a = 0; a < n; a++) {
        const uint8_t *km = kmers + (size_t)a * nb;
        uint8_t sh[40];
        int cr = 0, cl = 0;
        /* right_shifting, src/branchingNode.c:43-48: drop the first nucleotide */
        for (int i = 0; i < nb; i++) { sh[i] = (uint8_t)(km[i] >> 2); if (i + 1 < nb) sh[i] |= (uint8_t)(km[i + 1] << 6); }
        for (int x = 0; x < 4; x++) {
            sh[(k - 1) / 4] = (uint8_t)((sh[(k - 1) / 4] & ~(3u << (2 * ((k - 1) % 4)))) | ((unsigned)x << (2 * ((k - 1) % 4))));
            is_kmer_present(t, sh, &res);
            cr += res.found;
        }
        /* left: shift in a first nucleotide, drop the last (src/branchingNode.c:266-276) */
        for (int i = nb - 1; i >= 0; i--) { sh[i] = (uint8_t)(km[i] << 2); if (i > 0) sh[i] |= (uint8_t)(km[i - 1] >> 6); }
        if ((2 * k) % 8) sh[nb - 1] &= (uint8_t)((1u << ((2 * k) % 8)) - 1u);
        for (int x = 0; x < 4; x++) {
            sh[0] = (uint8_t)((sh[0] & ~3u) | (unsigned)x);
            is_kmer_present(t, sh, &res);
            cl += res.found;
        }
        if (counts) counts[a] = (uint8_t)((cr << 4) | cl);
        if (cr > 1 || cl > 1) { bits[a >> 3] |= (uint8_t)(1u << (a & 7)); nbr++; }
    }
    return nbr;
}

/* ------------------------------------------------------------------ */
/* query_sequence (src/bft.c:1241-1351)                               */
/* ------------------------------------------------------------------ */
#include <math.h>
static char rc_char(char c) { /* reverse_complement, src/fasta.c:387-440 (ACGTU only) */
    switch (c) { case 'A': return 'T'; case 'a': return 't'; case 'C': return 'G'; case 'c': return 'g'; case 'G': return 'C'; case 'g': return 'c';
                 case 'T': case 'U': return 'A'; case 't': case 'u': return 'a'; default: return c; }
}
int orc_query_sequence(orc_bft *t, const char *sequence, double threshold, int canonical, uint32_t nb_genomes, uint32_t *ids, int cap) {
    orc_freeze(t);
    const int k = t->k;
    long len = (long)strlen(sequence), nb = len - k + 1;
    if (nb < 0) nb = 0;
    long min_ = (long)ceil((double)nb * threshold); /* :1281 */
    uint32_t *count = xcalloc(nb_genomes ? nb_genomes : 1, 4);
    char fwd[130], rc[130];
    uint8_t packed[40];
    uint32_t tmp[4096];
    orc_res res;
    for (long i = 0; i < nb; i++) {
        memcpy(fwd, sequence + i, (size_t)k);
        fwd[k] = 0;
        const char *kmer = fwd;
        if (canonical) { /* :1287-1296 */
            for (int j = 0; j < k; j++) rc[j] = rc_char(fwd[k - 1 - j]);
            rc[k] = 0;
            /* compare on the upper-cased ACGT alphabet (strcmp of the reference on its usual upper-case input) */
            int cmp = 0;
            for (int j = 0; j < k && !cmp; j++) {
                char a = fwd[j] & ~0x20, b = rc[j] & ~0x20;
                if (a == 'U') a = 'T';
                if (b == 'U') b = 'T';
                cmp = (a > b) - (a < b);
            }
            if (cmp >= 0) kmer = rc;
        }
        memset(packed, 0, sizeof(packed));
        if (!orc_parse_kmer(kmer, k, packed)) continue; /* IUPAC / invalid k-mers are skipped (:1298) */
        is_kmer_present(t, packed, &res);
        if (!res.found) continue;
        g_comp = t->celems;
        g_ncomp = t->ncelems;
        int n = decode_any(res.annot, res.size_annot, tmp, 4096);
        for (int a = 0; a < n && a < 4096; a++)
            if (tmp[a] < nb_genomes) count[tmp[a]]++;
    }
    int out = 0;
    for (uint32_t g = 0; g < nb_genomes; g++)
        if (count[g] && (long)count[g] >= min_) { if (out < cap) ids[out] = g; out++; } /* :1322-1336 */
    free(count);
    return out;
}
