// bft_kh.h -- build, dump and launchers of the k-mer hash (bft_kh.hip); the table itself is described in bft_image.h (BFT_KH_*).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bft_claims.h"
#include "bft_dev.h"
#include "bft_image.h"

#define BFT_KH_BLOCK 256
bool bft_kh_has_kernels(int W, uint32_t S);
// Builds the table (geometry g: bft_kh_geometry, bft_walk.h) of the n rows of the sorted table d_tk (W words per row) with values d_vals on
// stream s, in the canonical layout, in two steps:
//   bft_kh_sort  the k-mers sorted by home line, their T-forms and values as payload;
//   bft_kh_lay   positions by one max-scan, then every line assembled and stored once: d_kh ((g.nl + BFT_KH_TAIL_LINES) lines) is
//                overwritten, k-mers displaced beyond the slots' displacement bits go to d_ovf_k / d_ovf_v (BFT_KH_OVF_CAP entries, unsorted:
//                the caller sorts the few there are); d_status: four words, [0] != 0 afterwards = no table (more overflow than the list
//                holds), [2] = the largest displacement in the table, [3] = k-mers in the overflow list.
// Nothing is synchronised; the transients live in `sc`, which the caller keeps until s has drained.
struct BftKhScratch { DevBuf b[7]; };
int bft_kh_sort(const uint64_t* d_tk, const uint32_t* d_vals, uint64_t n, int k, int W, const BftKhGeo& g, BftKhScratch& sc, hipStream_t s);
int bft_kh_lay(uint64_t n, int k, int W, const BftKhGeo& g, uint64_t* d_kh, uint64_t* d_ovf_k, uint32_t* d_ovf_v, uint32_t* d_status, BftKhScratch& sc, hipStream_t s);
// every (k-mer, value) of the table, unordered, word w of k-mer j at d_keys[w * stride + j]; *d_cnt (zeroed by the caller) = how many
int bft_kh_dump(const BftImage& im, uint64_t* d_keys, uint64_t stride, uint32_t* d_vals, unsigned long long* d_cnt, hipStream_t s);
// presence bits (+ colour-set id per k-mer when d_out32 != NULL) of n packed k-mers of `rec` bytes each
// d_ctr: {NULL}, or the stream's claim counter with this launch's base (bft_claims.h) -- the rounds of `chunk` blocks of 256 k-mers after the first
// are then claimed instead of dealt out by workgroup number (bft_claims.h), and the kernel leaves the words zeroed
int bft_kh_query(const BftImage& im, int grid_mult, const uint8_t* d_kmers, uint64_t n, int rec, uint64_t* d_bits64, uint32_t* d_out32, BftClaimCtr d_ctr, uint32_t chunk,
                 hipStream_t s);
// presence bits, offsets [n + 1] and genome ids of n packed k-mers in ONE launch; d_scratch: bft_kh_colors_scratch_bytes(n) bytes of the caller's
size_t bft_kh_colors_scratch_bytes(uint64_t n);
// presence bits and bitmap rows (rowbytes >= 16, d_out 16-byte aligned) of n packed k-mers in ONE launch
int bft_kh_color_rows(const BftImage& im, const uint8_t* d_kmers, uint64_t n, int rec, uint64_t* d_bits64, const uint8_t* bm, uint32_t stride, uint32_t rowbytes, uint8_t* d_out,
                      int device, hipStream_t s);
int bft_kh_colors(const BftImage& im, const uint8_t* d_kmers, uint64_t n, int rec, uint64_t* d_bits64, uint64_t* d_offsets, uint32_t* d_ids, uint64_t ids_cap, uint64_t* d_needed,
                  void* d_scratch, hipStream_t s);
int bft_kh_branching(const BftImage& im, const uint8_t* d_kmers, uint64_t n, int B, uint64_t* d_bits64, uint8_t* d_counts, BftClaimCtr d_ctr, uint32_t chunk, hipStream_t s);
// colour set of every k-mer position of a chunk of sequences (the arrays of query_sequences_core)
int bft_kh_seq(const BftImage& im, const uint64_t* d_codes, const uint32_t* d_bad, const uint64_t* d_seq_off, const uint64_t* d_pos_off, const uint32_t* d_tile_seq,
               uint32_t n_seqs, int canonical, uint32_t* d_csout, BftClaimCtr d_ctr, uint32_t chunk, hipStream_t s);

// bft_walkh.hip: the container walk with plain root groups looked up in the k-mer hash ("walk_hash"): presence bits (and colour sets when
// im.emit_cs) of n k-mers of `rec` bytes; d_ctr: the stream's claim counters or NULL
int bft_walkh_query(const BftImage& im, const uint8_t* d_kmers, uint64_t n, int rec, uint64_t* d_bits64, uint32_t* d_rows, BftClaimCtr d_ctr, uint32_t grid_mult, hipStream_t s);
