/*
 * bft_gpu.h -- C-ABI of the MI355X-native batched k-mer presence / colour / insertion path of the
 * Bloom Filter Trie.  This is the drop-in boundary: plain pointers and sizes, no C++ or torch
 * types.  Every entry point names the reference interface (GuillaumeHolley/BloomFilterTrie,
 * file:line) it replaces or batches.  Library: bloomfiltertrie_amd/csrc/libbft_gpu.so
 * (hipcc --offload-arch=gfx950).  INTEGRATION.md shows the binding a maintainer adds to the
 * reference's src/file_io.c / src/bft.c.
 *
 * k-mer batches use the reference's packed layout everywhere (parseKmerCount, src/fasta.c:3-53;
 * README.md:171-172): CEIL(2k/8) bytes per k-mer, nucleotide j in byte j/4 bits 2(j%4)..+1,
 * A=0 C=1 G=2 T=3, k-mers contiguous -- the `array_kmers` argument of insertKmers
 * (include/insertNode.h:26) and the 4096-byte chunks of src/file_io.c:726-730.
 *
 * Presence bitmaps: CEIL(n/8) bytes, k-mer i -> bit i%8 of byte i/8.
 *
 * All functions return BFT_GPU_OK (0) or a negative error code; bft_gpu_last_error() gives the
 * message for the calling thread.  (The reference has no error codes: ERROR() prints and exits,
 * include/useful_macros.h:33-43; the C wrapper of INTEGRATION.md keeps that behaviour.)
 * There is NO CPU fallback: without a usable HIP device every call fails with BFT_GPU_E_HIP.
 * A handle is not thread-safe (like BFT_Root, whose scratch fields make insertion and colour decoding non re-entrant,
 * include/Node.h:107-109): use one handle per thread, or serialise calls; distinct handles are independent.
 */
#ifndef BFT_GPU_H
#define BFT_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BFT_GPU_OK 0
#define BFT_GPU_E_ARG (-1)      /* bad argument (k not a multiple of 9, NULL pointer, ...) */
#define BFT_GPU_E_HIP (-2)      /* HIP runtime / device error */
#define BFT_GPU_E_LIMIT (-3)    /* a container limit of the format was exceeded */
#define BFT_GPU_E_STATE (-4)    /* call order (query before anything was built, ...) */
#define BFT_GPU_E_IO (-5)       /* file error / malformed .bft or image blob */
#define BFT_GPU_E_NOSPACE (-6)  /* caller-provided output buffer too small */

typedef struct bft_gpu bft_gpu; /* opaque: one BFT resident on one GPU (replaces BFT_Root, include/Node.h:96-122) */

const char* bft_gpu_last_error(void);
int bft_gpu_device_count(void);
const char* bft_gpu_version(void);

/* createBFT_Root(k, treshold_compression, compressed=0) (include/CC.h:214-258) / create_cdbg
 * (include/bft.h:62).  k in [9,126]; the reference requires a multiple of 9 (src/main.c:61-63) and only such
 * indexes can be written as .bft; other k (e.g. 31) are an extension verified against ground truth.  The Bloom seeds are
 * the reference's un-seeded rand() values (include/CC.h:246-248) unless r1/r2 > 0 are given. */
int bft_gpu_create(int k, int device, bft_gpu** out);
int bft_gpu_create_seeded(int k, int device, int r1, int r2, bft_gpu** out);
/* freeBFT_Root (include/CC.h:260-268) / free_cdbg (include/bft.h:63).  The handle's device blocks go to the library's cache of released
 * blocks (bounded: BFT_GPU_POOL_MAX_MB, by default an eighth of the device's memory), where the next handle's build finds them. */
void bft_gpu_free(bft_gpu* h);
/* Gives every cached device block back to the HIP runtime (for a process that has finished building and wants the memory for something else --
 * torch's allocator cannot see these blocks); returns the bytes released.  No reference counterpart: free() is the reference's only allocator. */
uint64_t bft_gpu_cache_release(void);

/* add_genomes_BFT_Root (include/CC.h:307-338): register a genome name; returns its id in *id_genome
 * (ids are 0-based and increase, as the reference's nb_genomes-1). */
int bft_gpu_add_genome(bft_gpu* h, const char* name, uint32_t* id_genome);

/* BFT_Root::filenames[id] (include/Node.h:97): NUL-terminated genome name ("genome_<id>" if none was registered). */
int bft_gpu_genome_name(bft_gpu* h, uint32_t id_genome, char* out, uint32_t cap);

/* insertKmers(root, array_kmers, nb_kmers, id_genome, size_id_genome) (include/insertNode.h:26,
 * src/insertNode.c:18-36).  The batch is converted on the GPU and appended to a device-side log;
 * the log is merged into the index by bft_gpu_build (or lazily by the first query).  No bound on the number of (k-mer, genome)
 * pairs an index holds (the reference has none): the log is merged on its own before it reaches 2^30 pairs ("flush_pairs"), a
 * larger batch is taken in pieces.  Limits that remain: < 2^31 distinct k-mers, < 2^32 genome ids in the colour-set dictionary.
 * `kmers` is a HOST pointer; the _dev variant takes a DEVICE pointer to the same layout. */
int bft_gpu_insert_kmers(bft_gpu* h, const uint8_t* kmers, uint64_t nb_kmers, uint32_t id_genome);
int bft_gpu_insert_kmers_dev(bft_gpu* h, const void* d_kmers, uint64_t nb_kmers, uint32_t id_genome);
/* The same, stream-ordered on the caller's stream (NULL = the null stream): the batch is converted by a kernel enqueued on hip_stream and the
 * call returns at once -- d_kmers follows stream semantics (it may be reused by later work on hip_stream, not before);
 * bft_gpu_build (or the first query) waits for hip_stream.  A series of insertions costs its kernels, not a host
 * round trip per call. */
int bft_gpu_insert_kmers_dev_async(bft_gpu* h, const void* d_kmers, uint64_t nb_kmers, uint32_t id_genome, void* hip_stream);

/* Bulk construction: what was inserted since the last build is sorted and de-duplicated on the GPU (root-prefix buckets, bucket
 * sorts in LDS), its colour sets interned, and that run is merged into the index (k-mers by position, colour sets by union: the index
 * itself -- sorted k-mers, colour set per k-mer, dictionary -- is the only store); then the containers are assembled level by level.
 * Replaces the per-k-mer work of insertKmer_Node / insertSP_CC / transform2CC / modify_annotations
 * (src/insertNode.c:38-423, src/CC.c:40-1474, src/retrieveAnnotation.c:232-314) with the invariants of SURVEY.md A.7. */
int bft_gpu_build(bft_gpu* h);

/* The loop of src/file_io.c:726-768 over isKmerPresent (include/presenceNode.h:57,
 * src/presenceNode.c:1823): one bit per k-mer.  Host buffers in, host bitmap out. */
int bft_gpu_query_presence(bft_gpu* h, const uint8_t* kmers, uint64_t nb_kmers, uint8_t* present_bits);
/* Same with inputs and outputs resident in HBM; runs on `hip_stream` (a hipStream_t, NULL = the
 * handle's own stream) and does not synchronise. d_present_bits: CEIL(n/64)*8 bytes.
 * Every *_dev entry point records where its work ends on a caller's stream; a later rebuild (insert + query), option
 * change or bft_gpu_free waits for that point before it releases or rewrites image arrays, so the caller need not
 * synchronise its stream first.  Every ABI call restores the calling thread's current HIP device before it returns. */
int bft_gpu_query_presence_dev(bft_gpu* h, const void* d_kmers, uint64_t nb_kmers, void* d_present_bits,
                               void* hip_stream);

/* get_annotation + get_list_id_genomes (include/bft.h:97,115; src/bft.c:363-387, 622-641;
 * src/annotation.c:2086-2250) for a batch: offsets[n+1] into ids (sorted genome ids per k-mer, none
 * for absent k-mers).  If ids_cap is too small nothing is written to ids, *ids_needed is set and
 * BFT_GPU_E_NOSPACE is returned. */
int bft_gpu_query_colors(bft_gpu* h, const uint8_t* kmers, uint64_t nb_kmers, uint8_t* present_bits,
                         uint64_t* offsets, uint32_t* ids, uint64_t ids_cap, uint64_t* ids_needed);
/* The same on a RESIDENT batch, without synchronisation (runs on hip_stream; NULL = the handle's stream): d_present_bits as in
 * bft_gpu_query_presence_dev, d_offsets = nb_kmers + 1 uint64 (offsets[nb_kmers] = the number of ids), d_ids = room for ids_cap uint32; *d_ids_needed
 * (device, may be NULL) receives the number of ids -- when it exceeds ids_cap, d_ids holds the first ids_cap ids only (never a byte beyond
 * ids_cap; offsets and bits are complete): size the buffer and call again, or pass d_ids = NULL / ids_cap = 0 first to learn the size.  (Until
 * round 6 nothing at all was written in that case; lookup, offsets and ids are ONE launch now -- k_colors_kh --, which knows the total only at its
 * end.)  The colour set of every found k-mer comes out of the line of the k-mer hash that answers presence (no row, no sorted table:
 * "compact_table" stays in force), its list's length out of the dictionary's offsets, a tile's place among all ids by a look-back over the tiles
 * before it, and the ids are streamed out wavefront by wavefront.  Scratch (8 bytes per 1024 k-mers) belongs to the handle: calls of one handle
 * on different streams are serialised by the library.  An image without the k-mer hash ("kmer_hash" 0, "walk_hash" 1) takes three launches --
 * the container walk, a scan, the fill -- and keeps the old rule (nothing written to a buffer that is too small). */
int bft_gpu_query_colors_dev(bft_gpu* h, const void* d_kmers, uint64_t nb_kmers, void* d_present_bits, void* d_offsets, void* d_ids, uint64_t ids_cap,
                             void* d_ids_needed, void* hip_stream);
/* Fixed-width variant = the CSV row of src/file_io.c:744-765 before formatting: row i is
 * CEIL(nb_genomes/8) bytes, genome g -> bit g%8 of byte g/8 (all zero for absent k-mers). */
int bft_gpu_query_color_rows(bft_gpu* h, const uint8_t* kmers, uint64_t nb_kmers, uint8_t* present_bits,
                             uint8_t* rows);

/* The loop of src/file_io.c:943-998 (-query_branching): isBranchingRight / isBranchingLeft
 * (src/branchingNode.c:16-112, :240-340) for a batch.  Bit i = k-mer i has more than one successor
 * (present k-mers kmer[1..k-1]+N) or more than one predecessor (present N+kmer[0..k-2]); the sum of the bits is
 * the CLI's "Nb branching k-mers".  counts (optional, n bytes): (successors << 4) | predecessors. */
int bft_gpu_query_branching(bft_gpu* h, const uint8_t* kmers, uint64_t nb_kmers, uint8_t* branching_bits, uint8_t* counts);
int bft_gpu_query_branching_dev(bft_gpu* h, const void* d_kmers, uint64_t nb_kmers, void* d_branching_bits, void* d_counts,
                                void* hip_stream);

/* query_sequence(bft, sequence, threshold, canonical_search) (include/bft.h:127, src/bft.c:1241-1351; CSV harness
 * src/file_io.c:1464-1574) for a batch: sequence i is seqs[seq_off[i] .. seq_off[i+1]) (ASCII, no terminator needed).
 * Every k-mer of a sequence is looked up (its reverse complement instead when canonical != 0 and it is not
 * lexicographically smaller); k-mers with a character outside ACGTU are skipped.  Row i (CEIL(nb_genomes/8) bytes)
 * has bit g set iff genome g holds at least ceil(nb_kmers(i) * threshold) of the sequence's k-mers. */
int bft_gpu_query_sequences(bft_gpu* h, const char* seqs, const uint64_t* seq_off, uint64_t nb_seqs, double threshold,
                            int canonical, uint8_t* rows);
/* The same on device-resident buffers, without synchronisation: d_seqs (total_chars ASCII bytes; any alignment, 16-byte aligned
 * blobs are read faster), d_seq_off (nb_seqs + 1 uint64 offsets into d_seqs, d_seq_off[nb_seqs] <= total_chars), d_rows
 * (nb_seqs x CEIL(nb_genomes/8) bytes).  The k-mer positions of the batch are counted on the device, so the call returns
 * as soon as its kernels are enqueued on hip_stream (NULL = the handle's stream); scratch belongs to the handle. */
int bft_gpu_query_sequences_dev(bft_gpu* h, const void* d_seqs, const void* d_seq_off, uint64_t nb_seqs, uint64_t total_chars,
                                double threshold, int canonical, void* d_rows, void* hip_stream);

/* load_BFT / read_BFT_Root (include/bft.h:176, src/write_to_disk.c:260-776): parse a reference .bft file
 * (compressed == 0; annotation modes 0/1/2 and extended-annotation bytes) and build the GPU image from its
 * k-mers and colour sets, with the file's Bloom seeds and genome names.  The file is mapped and decoded by a pool of host threads
 * (BFT_GPU_IO_THREADS); it must not be truncated by another process while the call runs (a mapped page that no longer exists is a SIGBUS, as
 * with any mmap reader).  Both calls give the host memory they used back on a detached thread after they have returned. */
int bft_gpu_load_bft(const char* path, int device, bft_gpu** out);
/* write_BFT / write_BFT_Root (include/bft.h:175, src/write_to_disk.c:21-258): serialise the image in the
 * reference's container layout so that the reference's `bft load` reads it back (invariants of SURVEY.md A.7). */
int bft_gpu_write_bft(bft_gpu* h, const char* path);

/* Device-resident variant: d_rows = n * CEIL(nb_genomes/8) bytes, d_scratch_rows_u32 = n * 4 bytes of scratch (the row
 * index of every k-mer), d_present_bits as in bft_gpu_query_presence_dev; runs on hip_stream, does not synchronise. */
int bft_gpu_query_color_rows_dev(bft_gpu* h, const void* d_kmers, uint64_t nb_kmers, void* d_present_bits, void* d_rows,
                                 void* d_scratch_rows_u32, void* hip_stream);

/* Shape / size counters (the walk of src/printMemory.c:255).  out[0]=k, [1]=distinct k-mers,
 * [2]=nodes, [3]=CCs, [4]=node-UC rows, [5]=child nodes, [6]=prefixes, [7]=CCs in s=4 mode,
 * [8]=max CCs per node, [9]=(k-mer,genome) pairs, [10]=distinct colour sets, [11]=genomes,
 * [12]=image bytes in HBM, [13]=root CCs, [14]=root UC rows, [15]=pending (unbuilt) pairs. */
int bft_gpu_info(bft_gpu* h, uint64_t* out, int n_out);
/* Bytes resident in HBM per part of the handle (the walk of src/printMemory.c:255 reports the reference's bytes per container kind):
 * out[0]=sorted k-mer table tk, [1]=colour-set id per k-mer, [2]=colour-set dictionary (offsets + genome ids in 1, 2 or 4 bytes, by the largest id inserted), [3]=containers (nodes, Bloom blocks, CC headers, filter2
 * words, cluster table, prefix entries, node UCs), [4]=flat form of the big CCs, [5]=root tables, [6]=node prefix hash, [7]=k-mer hash,
 * [8]=bitmap form of the dictionary (derived by the first colour-row query), [9]=hash table (hash_v % 1504), [10]=0 (rounds 1-2 kept a sorted
 * (k-mer, genome) pair store for later insertions; the index is its own store now), [11]=pending insertion log. */
int bft_gpu_footprint(bft_gpu* h, uint64_t* out, int n_out);

/* Options.
 * "kmer_hash" (1, default): besides the containers, every stored k-mer also sits in one open-addressed table of 64-byte lines (bft_image.h,
 *   BFT_KH_*: home line from a scattering bijection of the T-form's top 32 bits, which the line then stands for -- a slot stores 2k - 32 + ~9
 *   key bits and the colour-set id, 8 k-mers per line at k = 27 / 31 and 100 genomes); presence, colour, sequence and branching queries then cost
 *   ONE cache line per k-mer instead of a container walk (src/presenceNode.c:1284-1921 costs a line per level and per suffix-group probe).  Any
 *   k; derived when an image is built, loaded or unpacked; rows (bft_gpu_query_rows) always come from the walk.  0: no table, every query walks
 *   the containers and searches the sorted table.
 * "kmer_hash_load" (55): occupancy of the table's home lines in per cent, 10..80 (55: 1.09 lines read per lookup, 15 bytes per k-mer at k = 27).
 * "walk_hash" (0, default; 1: presence / colour queries are answered by the container walk, k_query6h, whose root level looks PLAIN suffix groups up in
 *   their hashed form -- the table above: one line -- and walks the containers for the rest: child Nodes, the root's UC).
 * "query_dynamic" (1, default): the query kernels deal their blocks of k-mers out in rounds: the first by workgroup (wavefront) number, the others
 *   claimed from a counter (one pair per stream that launches them, allocated when the first image is built) -- workgroups are bound to an XCD by
 *   their number, and a static split makes the launch as slow as the XCD that reaches the table slowest; "query_chunk" (4) blocks of 256 per round
 *   of the k-mer hash kernels; batches below "query_dynamic_min" (2^16) k-mers are split statically.  The counter of a (handle, stream) pair is one
 *   64-bit word that only grows: every launch gets its own range of it and raises it to the range's start itself, so a launch that never finished
 *   (a fault, a killed process sharing nothing) cannot make a later one skip blocks -- nothing is reset by anyone (round 5; "test_stale_claims" is
 *   the test hook that leaves the counters where such a launch would).  A handle keeps counters for 32 streams; queried on more, the least recently
 *   used slot moves to the new stream once its last launch has completed, else that launch runs static and is counted (bft_gpu_build_time entry 20).
 *   0: always static.  Launches of ONE handle on ONE stream must not run concurrently (two host threads): they would share a range -- use one stream
 *   per thread, as for any stream-ordered API.  A *_dev call recorded into a HIP graph (its stream is being captured) runs static rounds: a range's
 *   start is a kernel argument, which a replay cannot move (tests/test_gpu_parity.py::test_captured_queries_replay).  The id-list, colour-row and
 *   sequence calls may be recorded too, after one direct call of the same size (which sizes the handle's scratch: nothing may allocate while a stream
 *   is captured); what they zero they zero with kernels of the library's own -- a captured hipMemsetAsync replays correctly only once on ROCm 7.0.2
 *   (test_captured_colour_queries_replay, test_captured_sequence_queries_replay; tools/probe_graph_memset.py).
 * The container walk (k_query*): "query_wgs_per_cu" (how it sits on a CU: 1 = one 1024-thread workgroup, 4 wavefronts per SIMD; 2 = two of them, 8 per
 *   SIMD with 64 VGPRs each; 3 = two 768-thread workgroups, 6 per SIMD with 84 VGPRs each; 0, default = by rule: 3), "query_probe" (rows per probe of the
 *   suffix-group search: 4 = adjacent 32-byte blocks, 8 = 64-byte blocks with a re-interpolated guess, 0 = by rule from the mean group size),
 *   "query_grid_mult" (grid = resident workgroups x value), "node_hash" (the prefix entries of the nodes below the root also go into one hash table keyed
 *   by (node, prefix) -- one cache line per level of a deep trie instead of four: 1, default = when the image has no k-mer hash, i.e. when the walk answers
 *   every query ("kmer_hash" 0, "walk_hash" 1); 2 = always; 0 = never), "root_quartiles" (1, default: a 1 MiB table of the quarter boundaries of every plain root
 *   group: its sorted rows are searched from a guess interpolated inside the k-mer's quarter), "root_direct" (the root level
 *   goes through tables derived from the containers: 1 = a 2 MiB table with one entry per 18-bit prefix; 2 = a 1 MiB table of row ranges for the plain
 *   suffix groups, backed by the 2 MiB table; 3, default = 2 unless most root prefixes are child Nodes; 0 = the containers), "flat_min" (CCs with at
 *   least this many prefixes also get the two-load flat form; default 3584 = the CCs in s=4 mode; 65536 = none), "tune" (1: measure residency, probe
 *   mode and root tables of the walk on the current image with a batch drawn from the index -- the only call that times anything; it synchronises;
 *   nothing is ever tuned implicitly by a build or a query).
 * Footprint: "compact_table" (1, default: the sorted k-mer table and the colour set per k-mer -- 12 bytes per k-mer -- do not stay resident once the
 *   k-mer hash holds every (k-mer, colour set): presence, colour-row, branching and sequence queries never need them; rows, extraction, a merge
 *   of new insertions, .bft files, packed images, the container walk and "tune" bring them back first -- a dump of the table + one sort, milliseconds --
 *   and they stay until the next build or until the option is set again.  No effect on a handle without a k-mer hash ("kmer_hash" 0).  0: the table stays.)
 * Build: "build_composite" (1, default: one-word keys whose genome ids arrive ascending take the root-prefix front end -- as 8-byte composites
 *   k-mer << bits | genome where that fits 63 bits, as (k-mer, id) pairs whose composite is formed inside a bucket otherwise; 0: the general key + value
 *   sort -- same image either way, a test hook), "composite_log" (1, default: where a one-word key leaves 7 bits or more for the id, k <= 28, the
 *   insertion log itself holds those composites -- 8 bytes per pending pair instead of 12, and the root-prefix split reads them as they are;
 *   an id beyond the room, ids that do not ascend or "build_composite" 0 turn the log back into k-mers + ids; 0: always k-mers + ids --
 *   same image, a test hook; only while the log is empty), "build_msd"(1, default: root-prefix buckets + bucket sorts from 2^20 pairs on; 0: one
 *   device-wide sort; 2: buckets at any size -- same image, test hooks), "test_front_rank_mode" (process-wide test hook: how a bucket ranks its digits:
 *   0, default = LDS atomics + order check + ballot fallback, 1 = ballots only, 2 = the check always fails), "sort_ballots" (process-wide: how the library's own
 *   radix sort -- csrc/bft_sort.h -- ranks the keys of a wavefront: 0, default = one LDS atomic per key once the device has shown that it serves the lanes of such an
 *   instruction in lane order, 1 = wavefront ballots, stable by construction; same image), "reserve_pairs" (room in the insertion log for this many pending (k-mer, genome)
 *   pairs, so that a series of insert calls never re-allocates it), "flush_pairs" (the log is merged into the index before it holds this many pairs:
 *   2^30 by default, 1024..2^30).
 * "timing" (0/1: record HIP events around query kernels; off until this option or the first bft_gpu_kernel_time call turns it on).
 * "build_stages" (0/1: bft_gpu_build records GPU time and algorithmic bytes per stage, see bft_gpu_build_stages). */
int bft_gpu_set_option(bft_gpu* h, const char* name, int64_t value);

/* Test hook: raw device->host copy of one array of the image ("nodes", "bfT", "ccs", "f2w", "clus",
 * "child", "uck", "ucrow", "tk", "tcol", and the derived "ccx", "f18", "fent", "kh"); out may be NULL to query the size. */
int bft_gpu_debug_get_array(bft_gpu* h, const char* name, void* out, uint64_t cap_bytes, uint64_t* nbytes);

/* HIP-event timing of the query kernels launched through this handle since the last reset:
 * *ms = summed kernel time, *launches = number of launches.  Timing is off by default (no event on the launch path); the
 * first call of this function turns it on, so call it once (reset = 1) before the region to be timed. */
int bft_gpu_kernel_time(bft_gpu* h, double* ms, uint64_t* launches, int reset);
/* Same for the GPU part and the host part of bft_gpu_build (last call): ms[0]=sort+dedupe (GPU),
 * ms[1]=colour-set interning (GPU), ms[2]=container assembly (GPU), ms[3]=root-prefix buckets of the last sort whose order check failed and that were sorted a second time (expected 0: see k_bucket_sort), ms[4]=derived arrays (flat CC form, root tables, node prefix
 * hash, k-mer hash), ms[5]=resident k_query workgroups per CU in use (1, 2 or 3), ms[6..7]=time of the "tune" batch with 1 / 2 workgroups per CU (0 when not
 * tuned), ms[8]=rows per suffix-group probe in use (4 or 8), ms[9]=lines of the k-mer hash (0 = none), ms[10]=GPU time of its fill, ms[11]=largest root-prefix bucket of the last sort (0: one device-wide sort),
 * ms[12]=times the colour-set interning had to compare lists (signature collisions), ms[13]=ms this process has spent in hipMalloc so far,
 * ms[14]=root tables in use (0 / 1 / 2, see "root_direct"), ms[15..16]="tune": time with the direct table alone / with the range table, ms[17]=keys in the
 * node prefix hash, ms[18]=keys it dropped (full bucket: those lookups take the container path), ms[19]="tune": time with residency 3,
 * ms[20]=query launches that wanted a claim counter and ran static, ms[21..24]=k-mer hash: slots per line, displacement bits, largest displacement, overflow list. */
int bft_gpu_build_time(bft_gpu* h, double* ms, int n_out);
/* The last bft_gpu_build stage by stage, when bft_gpu_set_option(h, "build_stages", 1) was set before it: names = the stage names, one per
 * line (NUL-terminated; names_cap bytes), ms[i] = GPU time of stage i (HIP events on the build's stream: the time the stream spent between the end
 * of the previous stage and the end of this one, the host's waits for counts included), bytes[i] = the bytes the stage's algorithm reads + writes,
 * from its own array sizes (0: a chain of small kernels, not a streaming stage).  Names that start with '+' ran on the build's second stream beside
 * the main chain and are timed from the build's start.  *n_out = number of stages; any of names / ms / bytes may be NULL.  Replaces nothing in the
 * reference (insertKmers has no instrumentation, src/insertNode.c:18-36): this is what bench.py's `insert` block is made of. */
int bft_gpu_build_stages(bft_gpu* h, char* names, uint32_t names_cap, double* ms, double* bytes, int cap, int* n_out);

/* iterate_over_kmers-style dump (include/bft.h:166): copies every stored k-mer (packed layout,
 * ascending T-form order) and its colour-set id; either pointer may be NULL. */
int bft_gpu_extract(bft_gpu* h, uint8_t* kmers_out, uint32_t* colorset_out, uint64_t cap, uint64_t* n_out);
int bft_gpu_colorset(bft_gpu* h, uint32_t colorset, uint32_t* ids, uint32_t cap, uint32_t* n_out);

/* What the reference keeps in resultPresence for a found k-mer (include/Node.h:60-92, filled by isKmerPresent,
 * src/presenceNode.c:1823-1921), as indexes instead of host pointers: rows[i] = position of k-mer i in the stored k-mer
 * table (the order of bft_gpu_extract), colorsets[i] = id of its colour set (argument of bft_gpu_colorset);
 * 0xFFFFFFFF for an absent k-mer.  Any of present_bits / rows / colorsets may be NULL.  Host buffers. */
int bft_gpu_query_rows(bft_gpu* h, const uint8_t* kmers, uint64_t nb_kmers, uint8_t* present_bits, uint32_t* rows,
                       uint32_t* colorsets);

/* A colour set as the reference's annotation bytes -- BFT_annotation::annot as get_annotation returns it
 * (include/bft.h:97, src/bft.c:363-387): mode 0 (bitmap, genome g <-> bit g+2), 1 (ranges) or 2 (id list), chosen the way the
 * reference chooses it: compute_best_mode re-decides at every insertion of a genome id and keeps the current mode on a size tie
 * (src/annotation.c:621-653), so the bytes depend on the order the ids arrived in -- ascending -- and the rule is replayed over the
 * sorted id list (e.g. {6,7} stays the id list 1a 1e it started as, although a bitmap would be no longer), the run-end size estimate of the
 * bitmap mode included (:515-523: the end of a run is priced with the byte count of the id one past it).  disabled_flags (:622) is never set
 * anywhere in the reference: nothing to replay.  The same bytes go into the .bft files bft_gpu_write_bft writes.  annot may be NULL to
 * query the size. */
int bft_gpu_colorset_annot(bft_gpu* h, uint32_t colorset, uint8_t* annot, uint32_t cap, uint32_t* n_out);

/* Replication of a built index on another GPU (the query path shards over GPUs with the trie image
 * replicated in each GPU's HBM; the reference has one BFT_Root per process, include/Node.h:96-122).
 * bft_gpu_image_size: bytes of the self-describing device blob; bft_gpu_image_pack: writes it at d_blob (device
 * memory of the handle's GPU, cap >= size) on hip_stream (0 = the handle's stream) and synchronises that stream;
 * bft_gpu_image_unpack: new handle on `device` from a blob resident in that GPU's memory (e.g. the receive buffer
 * of one RCCL broadcast) -- same answers, same bft_gpu_write_bft bytes, and insertion can continue on it. */
int bft_gpu_image_size(bft_gpu* h, uint64_t* nbytes);
int bft_gpu_image_pack(bft_gpu* h, void* d_blob, uint64_t cap, void* hip_stream);
int bft_gpu_image_unpack(const void* d_blob, uint64_t nbytes, int device, bft_gpu** out);

/* One index on several GPUs of ONE process (the loops of src/file_io.c:651-895 and :897-1020 can only ever use one
 * BFT_Root).  bft_gpu_group_create: `src` (a handle on GPU src_device; built if need be) is replicated into the HBM of every device of
 * devices[0..n_devices) -- image blob packed on the source GPU, one peer copy per replica, unpacked there; the first slot naming
 * src_device is served by src itself, further slots (the same device may appear twice) get copies.  The group owns its replicas, not src.
 * The *_query_* calls cut a host batch into contiguous slices whose starts are multiples of 64 k-mers (bft_gpu_group_shard gives slice i
 * of `parts`: the same rule bloomfiltertrie_amd/dist.py applies across processes) and return when every slot has answered its slice into the
 * caller's buffers -- same layouts as the single-GPU calls.  Every slot has a host thread of its own for as long as the group lives, with a
 * stream on its device and two slots of pinned staging memory: a slice moves in chunks (at most 2^22 k-mers / ~64 MiB), chunk c + 1 copied into
 * pinned memory while the GPU answers chunk c through the *_dev entry points -- the caller's arrays may be pageable, the copies never are.
 * Insertion stays single-GPU: insert into src, then create the group again. */
typedef struct bft_gpu_group bft_gpu_group;
int bft_gpu_group_shard(uint64_t n, int parts, int i, uint64_t* begin, uint64_t* end);
int bft_gpu_group_create(bft_gpu* src, int src_device, const int* devices, int n_devices, bft_gpu_group** out);
void bft_gpu_group_free(bft_gpu_group* g);
int bft_gpu_group_size(bft_gpu_group* g);
int bft_gpu_group_query_presence(bft_gpu_group* g, const uint8_t* kmers, uint64_t nb_kmers, uint8_t* present_bits);
int bft_gpu_group_query_color_rows(bft_gpu_group* g, const uint8_t* kmers, uint64_t nb_kmers, uint8_t* present_bits, uint8_t* rows);
int bft_gpu_group_query_branching(bft_gpu_group* g, const uint8_t* kmers, uint64_t nb_kmers, uint8_t* branching_bits, uint8_t* counts);
/* The same on DEVICE-RESIDENT batches: arrays of bft_gpu_group_size(g) entries, entry i = the batch of slot i -- d_kmers[i] (n[i] packed k-mers)
 * and the outputs lie in the memory of GPU bft_gpu_group_member_device(g, i); hip_streams[i] (or NULL: the slot's own stream; hip_streams itself
 * may be NULL) orders the work.  Like the single-GPU *_dev calls these only enqueue: no host thread, no synchronisation, the slots run side by
 * side; the caller synchronises its streams (a caller that owns one shard per GPU -- the partition of bft_gpu_group_shard or any other -- keeps
 * every buffer where it is produced and consumed: the resident rate of every GPU, not the host link's).  n[i] == 0 skips slot i. */
int bft_gpu_group_member_device(bft_gpu_group* g, int i);
/* bft_gpu_footprint / bft_gpu_info of slot i's handle (the replicas belong to the group: this is how their residency is inspected) */
int bft_gpu_group_member_footprint(bft_gpu_group* g, int i, uint64_t* out, int n_out);
int bft_gpu_group_member_info(bft_gpu_group* g, int i, uint64_t* out, int n_out);
int bft_gpu_group_query_presence_dev(bft_gpu_group* g, const void* const* d_kmers, const uint64_t* n, void* const* d_present_bits, void* const* hip_streams);
int bft_gpu_group_query_color_rows_dev(bft_gpu_group* g, const void* const* d_kmers, const uint64_t* n, void* const* d_present_bits, void* const* d_rows,
                                       void* const* d_scratch_rows_u32, void* const* hip_streams);
int bft_gpu_group_query_branching_dev(bft_gpu_group* g, const void* const* d_kmers, const uint64_t* n, void* const* d_branching_bits, void* const* d_counts,
                                      void* const* hip_streams);

#ifdef __cplusplus
}
#endif
#endif
