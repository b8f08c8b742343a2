// bft_gpu.hip -- the C-ABI of include/bft_gpu.h: handle, device-memory cache, insertion log, bulk build, query entry points,
// residency / probe tuning, .bft files, image replication.  One translation unit with its device code:
//   bft_kernels_query.h  k_pack_to_tform (packed 2-bit k-mers -> T-form words: the per-level rev[]/rotation work of
//                        src/presenceNode.c:1327-1371 done once per k-mer), k_query_kh / k_branching_kh (the same queries through the
//                        k-mer hash: one cache line per k-mer), k_query / k_query8 / k_query6 (batched isKmerPresent as a container walk,
//                        src/presenceNode.c:1823-1921: one lane per k-mer, coalesced dword loads of the batch, hash table + root
//                        Bloom block + root CC headers staged in LDS, 64 presence bits per wavefront via __ballot),
//                        k_branching / k_branching8 (src/branchingNode.c)
//   bft_kernels_seq.h    query_sequence (src/bft.c:1241-1351) around k_query
//   bft_kernels_build.h  de-duplication of sorted (k-mer, genome) pairs for the bulk build
//   bft_kernels_color.h  batched get_annotation + get_list_id_genomes (src/bft.c:363-387, 622-641)
//   bft_walk.h           the per-k-mer walk itself (shared with the host-side test helper)
// The bulk build sorts and scans with the library's own kernels (bft_sort.h, bft_scan.h); colour-set interning and container
// assembly are the kernels of bft_assemble.hip -- see DESIGN.md "Insertion".
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bft_gpu.h"
#include "bft_dev.h"
#include "bft_file.h"
#include "bft_hash.h"
#include "bft_image.h"
#include "bft_index.h"
#include "bft_kh.h"
#include "bft_scan.h"
#include "bft_sort.h"
#include "bft_walk.h"

#define BFT_BLOCK 256
#define BFT_ABSENT_ROW 0xFFFFFFFFu
// Genome ids index colour rows of CEIL(nb_genomes/8) bytes and bit positions of annotations: ids from 2^24 on are refused
// (the reference's own annotation codec holds 6 bits per byte, include/log2.h:45-50: 2^24 ids already take 4-byte entries).
#define BFT_MAX_GENOME_ID (1u << 24)

int bft_rs::g_bft_rs_rank_mode = -1;  // how bft_sort.h ranks: 0 LDS atomics (lane order checked on the device), 1 ballots ("sort_ballots")
static thread_local std::string g_err;
int bft_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
static int fail(int code, const std::string& msg) { return bft_fail(code, msg); }

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// BFT_GPU_TRACE_BUILD=1: host-side timeline of bft_gpu_build on stderr (where the host waits, allocates, reads counts back)
static bool g_trace = getenv("BFT_GPU_TRACE_BUILD") != nullptr;
static double g_trace_t0 = 0, g_trace_last = 0;
bool bft_trace_on(void) { return g_trace; }
void bft_trace_mark(const char* what) {
    if (!g_trace) return;
    const double t = now_ms();
    if (!what) { g_trace_t0 = g_trace_last = t; return; }
    fprintf(stderr, "[bft_gpu build] %8.3f ms (+%.3f) %s\n", t - g_trace_t0, t - g_trace_last, what);
    g_trace_last = t;
}

// Stage record of a build ("build_stages" 1): bft_stage() drops an event on the build's stream where a stage ends and notes the
// bytes the stage's algorithm reads + writes (from its own array sizes); bft_gpu_build resolves the events into GPU time per stage
// when it is done (bft_gpu_build_stages).  The host's waits between stages are inside the figures: a stage is what the stream spent
// between two marks.  One build at a time per thread.
namespace {
struct StageMark { std::string name; double bytes; hipEvent_t ev; };
thread_local bool t_stages_on = false;
thread_local std::vector<StageMark> t_stage_marks;
thread_local std::vector<hipEvent_t> t_stage_pool;
}  // namespace
void bft_stage(const char* name, double bytes, hipStream_t s) {
    if (!t_stages_on) return;
    hipEvent_t ev = nullptr;
    if (!t_stage_pool.empty()) { ev = t_stage_pool.back(); t_stage_pool.pop_back(); }
    else if (hipEventCreate(&ev) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipEventRecord(ev, s) != hipSuccess) { (void)hipGetLastError(); t_stage_pool.push_back(ev); return; }
    t_stage_marks.push_back({name, bytes, ev});
}

// device code, by topic
#include "bft_kernels_query.h"
#include "bft_kernels_seq.h"
#include "bft_kernels_build.h"
#include "bft_kernels_color.h"
// ------------------------------------------------------------------------------------------------
// device-memory cache (see bft_dev.h)
// ------------------------------------------------------------------------------------------------
namespace {
struct PoolBlock {
    void* p;
    size_t cap;
    int device;
    hipStream_t stream;
};
std::mutex g_pool_mu;
std::vector<PoolBlock> g_pool;
size_t g_pool_bytes = 0;
// cached, unused memory kept at most (per process): an eighth of the device's memory (36 GB of an MI355X's 288: the transients of a
// 2 x 10^8-pair build are ~12 GB, and a hipFree of a gigabyte block costs a device synchronisation and ~0.2 ms), 8 GiB at least,
// unless BFT_GPU_POOL_MAX_MB says otherwise (0 = no cache); torch's allocator cannot see these blocks, so the cap bounds what the
// library withholds from it
size_t pool_max_bytes() {
    static const size_t v = [] {
        const char* e = getenv("BFT_GPU_POOL_MAX_MB");
        if (e) return (size_t)strtoull(e, nullptr, 10) << 20;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); tot = 0; }
        return std::max<size_t>((size_t)8 << 30, tot / 8);
    }();
    return v;
}
constexpr size_t POOL_MAX_BLOCKS = 256;
// tag of a block whose stream was synchronised and destroyed since (a closed handle's): nothing to wait for, whoever takes it
const hipStream_t POOL_DRAINED = (hipStream_t)(uintptr_t)1;
double g_malloc_ms = 0;          // time spent in hipMalloc by this process (diagnostic: bft_gpu_build_time)
uint64_t g_malloc_calls = 0;
double g_free_ms = 0;            // ... and in hipFree
uint64_t g_free_calls = 0;
thread_local int t_pool_device = -1;
thread_local hipStream_t t_pool_stream = nullptr;

void pool_free_block(const PoolBlock& b) {
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != b.device) (void)hipSetDevice(b.device);
    (void)hipFree(b.p);
    if (cur != b.device && cur >= 0) (void)hipSetDevice(cur);
}
}  // namespace

void bft_pool_set_stream(int device, hipStream_t s) {
    t_pool_device = device;
    t_pool_stream = s;
}

int bft_pool_alloc(void** p, size_t n, size_t* cap) {
    *p = nullptr;
    n += 256;  // slack behind every device array: aligned multi-row probes may read past the last row (bft_load_pair)
    PoolBlock take{nullptr, 0, 0, nullptr};
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < g_pool.size(); i++) {
            const PoolBlock& b = g_pool[i];
            if (b.device != t_pool_device || b.cap < n || b.cap > n + n / 2 + 4096) continue;  // similar size only
            if (best == (size_t)-1 || b.cap < g_pool[best].cap) best = i;
        }
        if (best != (size_t)-1) {
            take = g_pool[best];
            g_pool[best] = g_pool.back();
            g_pool.pop_back();
            g_pool_bytes -= take.cap;
        }
    }
    if (take.p) {
        // released under another stream: its work must have drained before the block is written again
        if (take.stream != t_pool_stream && take.stream != POOL_DRAINED && hipStreamSynchronize(take.stream) != hipSuccess) {
            pool_free_block(take);
            take.p = nullptr;
        }
    }
    if (take.p) {
        *p = take.p;
        *cap = take.cap;
        return 0;
    }
    const double t_m0 = now_ms();
    hipError_t e = hipMalloc(p, n);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_malloc_ms += now_ms() - t_m0;
        g_malloc_calls++;
    }
    if (e != hipSuccess) {
        // out of memory with blocks parked in the cache: give them back and retry once
        std::vector<PoolBlock> all;
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            all.swap(g_pool);
            g_pool_bytes = 0;
        }
        for (const PoolBlock& b : all) pool_free_block(b);
        (void)hipGetLastError();
        e = hipMalloc(p, n);
    }
    if (e != hipSuccess) return bft_fail(BFT_GPU_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    *cap = n;
    return 0;
}

void bft_pool_release(void* p, size_t cap) {
    if (!p) return;
    PoolBlock b{p, cap, t_pool_device, t_pool_stream};
    bool keep = t_pool_device >= 0;
    if (keep) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_pool.size() < POOL_MAX_BLOCKS && g_pool_bytes + cap <= pool_max_bytes()) {
            g_pool.push_back(b);
            g_pool_bytes += cap;
        } else
            keep = false;
    }
    if (!keep) {
        const double t0 = now_ms();
        (void)hipFree(p);
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_free_ms += now_ms() - t0;
        g_free_calls++;
    }
}

void bft_pool_drop_stream(hipStream_t s) {
    // (the caller has synchronised s and destroys it next: its blocks stay in the cache -- the next handle's build of the same size finds them,
    // where a hipMalloc of a gigabyte block costs from a millisecond to 0.4 s depending on the box -- but must never be waited for on s again)
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (PoolBlock& b : g_pool)
        if (b.stream == s) b.stream = POOL_DRAINED;
}

extern "C" uint64_t bft_gpu_cache_release(void) {
    std::vector<PoolBlock> all;
    uint64_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        all.swap(g_pool);
        bytes = g_pool_bytes;
        g_pool_bytes = 0;
    }
    for (const PoolBlock& b : all) {
        if (b.stream != POOL_DRAINED) {  // (released under a live handle's stream: what was enqueued before the release may still read it)
            int cur = -1;
            (void)hipGetDevice(&cur);
            if (cur != b.device) (void)hipSetDevice(b.device);
            (void)hipStreamSynchronize(b.stream);
            if (cur != b.device && cur >= 0) (void)hipSetDevice(cur);
        }
        pool_free_block(b);
    }
    return bytes;
}

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
struct bft_gpu {
    int k = 0, L = 0, W = 0, B = 0, device = 0, r1 = 0, r2 = 0;
    hipStream_t stream = nullptr;
    std::vector<std::string> genomes;
    uint32_t max_gid_seen = 0;
    bool any_insert = false;
    bool log_g_sorted = true;   // the log's genome ids are non-decreasing
    uint32_t log_last_gid = 0;
    uint64_t opt_flush_pairs = 1ull << 30;  // "flush_pairs": the log is merged into the index before it holds this many pairs

    // Small host batches (the per-k-mer calls of <bft/bft.h>, 4096-byte file chunks): a pinned, device-mapped staging
    // block the kernels read and write directly -- one launch + one stream wait instead of two staged copies around it.
    uint8_t* pin = nullptr;  // [PIN_IN bytes of k-mers | bits | rows | colour sets]
    // Host batches of insertKmers up to a megabyte (a genome of a many-colour collection: 2000 calls of 20 000 k-mers on config 5): a ring
    // of pinned, device-mapped slots -- the batch is copied into the next slot, the packing kernel reads it there over the link, an
    // event says when the slot is free again; nothing waits for the GPU (95 -> ~35 us per call).
    static constexpr size_t RING_SLOT = (size_t)1 << 20;
    static constexpr int RING_SLOTS = 8;
    uint8_t* ring = nullptr;
    hipEvent_t ring_ev[RING_SLOTS] = {};
    bool ring_busy[RING_SLOTS] = {};
    int ring_next = 0;
    ~bft_gpu() {
        if (pin) (void)hipHostFree(pin);
        for (int i = 0; i < RING_SLOTS; i++)
            if (ring_ev[i]) (void)hipEventDestroy(ring_ev[i]);
        if (ring) (void)hipHostFree(ring);
        if (kh_ctr) (void)hipFree(kh_ctr);
        for (int i = 0; i < KH_CTR_SLOTS; i++)
            if (kh_ctr_ev[i]) (void)hipEventDestroy(kh_ctr_ev[i]);
    }

    // pending insert log (SoA: W key arrays of log_cap entries, then genome ids)
    DevBuf log_k, log_g;
    uint64_t log_n = 0, log_cap = 0;
    // One-word keys with room for a genome id beside them (k <= 28: 63 - 2k >= 7 bits) are logged as the COMPOSITES T << log_gb | genome the build's
    // root-prefix split sorts: 8 bytes per pair written at insert time and read by the split's histogram and first pass instead of 12 (no id array:
    // log_g stays empty).  A genome id beyond 2^log_gb, ids that do not ascend, or the general sort ("build_composite" 0) turn the log back
    // into k-mers + ids first (k_log_decompose).
    bool log_comp = false;
    uint32_t log_gb = 0;
    // the insert calls behind the log: positions [lb_end[j - 1], lb_end[j]) carry genome lb_gid[j] -- the multi-word sort reads a pair's id out of this
    // table (a search in a few cached words) instead of gathering it from the log (a fabric request per pair)
    std::vector<uint64_t> lb_end;
    std::vector<uint32_t> lb_gid;
    int opt_comp_log = 1;  // "composite_log": 0 = always k-mers + ids (a test hook: same image)

    uint64_t n_pairs = 0;  // distinct (k-mer, genome) pairs the index holds = sum of the sizes of its k-mers' colour sets

    // image
    bool built = false;
    uint64_t n_kmers = 0;
    DevBuf d_hashmod, d_nodes, d_bfT, d_ccs, d_f2w, d_clus, d_child, d_tk, d_tcol, d_uck, d_ucrow, d_cs_off, d_cs_ids, d_cs_bm;
    DevBuf d_ccx, d_f18, d_fent;  // derived: flat form of the big CCs (bft_flatten_gpu)
    DevBuf d_rdir, d_rstart;      // derived: root direct table (BFT_RDIR_*, k_root_direct) and root range table (BFT_RSTART_*), optional
    DevBuf d_rq;                  // derived: root quartile table (BFT_RQ_*, k_root_quartiles), optional
    int opt_root_quartiles = 1;   // "root_quartiles"
    bool rq_ok = false;
    DevBuf d_nph;                 // derived: node prefix hash (BFT_NPH_*, k_nph_fill), optional
    int opt_node_hash = 1;        // "node_hash": 1 = derived when the image has no k-mer hash (the walk then answers every query), 2 = always, 0 = never
    uint64_t nph_inserted = 0, nph_dropped = 0;
    DevBuf d_kh, d_rspec;         // derived: k-mer hash (BFT_KH_*, bft_kh_build), optional; one "special" bit per root prefix for the walk (sync_walk_kh)
    DevBuf d_kh_ovf_k, d_kh_ovf_v; // its overflow list (sorted k-mers, values)
    uint32_t kh_ovf_n = 0;
    uint64_t kh_lines = 0;        // home lines
    bool opt_kmer_hash = true;    // "kmer_hash"
    bool opt_walk_hash = false;   // "walk_hash": presence / colour queries through the container walk, which looks plain root groups up in the table's regions
    bool opt_compact = true;      // "compact_table" (default on): the sorted table and the colour set per k-mer are dropped once the k-mer hash holds them (ensure_table)
    bool table_dropped = false;   // d_tk / d_tcol are not resident: the k-mer hash is the only copy
    uint32_t opt_kh_load = 55;    // "kmer_hash_load": per cent of the slots of the home lines in use (55: 47.9 G k-mers/s at 15.0 B per k-mer on the config-4 share;
                                  // 50: 48.4 / 16.6; 60: 45.7 / 13.5; 70: 39.3 / 11.7 -- profiles/r04/kh_forms.jsonl)
    double kh_ms = 0;             // GPU time of the last fill
    hipStream_t stream2 = nullptr; // bft_gpu_build fills the k-mer hash here while the containers are assembled on `stream`
    int opt_root_direct = 3;      // "root_direct": 0 = containers, 1 = direct table, 2 = direct table + range table, 3 = 1 or 2, whichever
                                  // measured faster on this image (tune_residency)
    bool rstart_ok = false;       // d_rstart holds the range table of the current image
    int tuned_rstart = -1;        // result of that measurement (-1 = none)
    double rstart_tune_ms[2] = {0, 0};
    uint64_t n_f18 = 0, n_fent = 0;
    uint32_t opt_flat_min = BFT_TRESH_SUF_PREF;  // CCs with at least this many prefixes get the flat form ("flat_min")
    bool has_cs_bm = false, cs_bm_tried = false;
    bool opt_no_composite = false;  // test hook ("build_composite" 0): the general sort + flag-array path also for ordered one-word keys
    uint32_t front_redone = 0;      // root-prefix buckets of the last build whose order check failed (bft_front.hip)
    int opt_msd = 1;                // "build_msd": root-prefix buckets + bucket sorts for 2^20 pairs and more (1), always (2: test hook), never (0)
    uint32_t msd_max_bucket = 0;    // largest root-prefix bucket of the last build's sort (0: one device-wide sort)
    BftImage im;
    std::vector<uint32_t> hashmod;
    std::vector<uint32_t> cs_off, cs_ids;  // host copy of the colour-set dictionary, fetched on first use (host_colorsets)
    uint64_t n_sets = 0, n_ids = 0;
    uint32_t cs_w = 4;  // bytes per genome id of the resident dictionary d_cs_ids (1 / 2 / 4: narrow_ids)
    bool cs_on_host = false;
    uint64_t info[16] = {0};
    double build_ms[5] = {0, 0, 0, 0, 0};

    // kernel timing: off until bft_gpu_kernel_time / set_option("timing", 1) asks for it; events are pooled per handle
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_ev;
    std::vector<hipEvent_t> free_ev;
    double kernel_ms = 0;
    uint64_t kernel_launches = 0;
    bool timing = false;
    // last work the *_dev entry points put on a caller's stream: image arrays are not released or rewritten before it is done
    struct ExtEv { hipStream_t stream; hipEvent_t ev; bool pending; };
    std::vector<ExtEv> ext;  // one event per caller stream seen (a double-buffered caller alternates between two: neither call blocks the host)
    uint32_t root_ncc = 0;
    uint64_t idx_sizes[9] = {0};
    // The container walk (k_query*): how it sits on a CU and how it probes suffix groups.  0 = by rule from the shape of the index
    // (default_launch_shape), or -- after bft_gpu_set_option("tune", 1) -- as measured on the image (tune_residency).
    int opt_wgs_per_cu = 0;   // 1 / 2 workgroups of 1024 threads per CU, 3 = two of 768
    int tuned_wgs = 0;
    int opt_probe = 0;        // suffix-group probe: 4 or 8 rows per block (BftImage::probe_big)
    int tuned_probe = 0;
    double tune_ms[3] = {0, 0, 0};  // best time of the tuning batch per residency 1 / 2 / 3
    int opt_grid_mult = 1;    // grid = resident workgroups x this
    // The k-mer hash kernels claim their blocks of k-mers from a counter instead of splitting them by workgroup number (k_query_kh,
    // bft_kh.hip): one counter per stream that launches them -- launches of one stream follow each other, so a counter has one user at
    // a time; it only grows, every launch with a range of its own (bft_claims.h).  Batches too small to matter take the static split.
    int opt_query_dynamic = 1;
    uint64_t opt_query_dynamic_min = (uint64_t)1 << 16;  // batches below this many k-mers (lines of work for the branching kernel) keep the static split
    uint64_t claims_static_launches = 0;  // launches that wanted a counter and found every slot taken by streams with work still in flight
    uint32_t opt_query_chunk = 4;  // largest claim, in blocks of 256 k-mers (4 = every claim: the smaller the window of the query stream the
                                   // resident workgroups read at a time, the better -- 2.61 / 2.62 / 2.65 / 2.70 ms at 4 / 16 / 32 / 64)
    static constexpr int KH_CTR_SLOTS = 32;
    unsigned long long* kh_ctr = nullptr;  // (its own hipMalloc, not a block of the cache: nothing that was released while still in flight may write here)
    hipStream_t kh_ctr_stream[KH_CTR_SLOTS] = {};
    unsigned long long kh_ctr_base[KH_CTR_SLOTS] = {};  // where the next launch's range of the slot's counter starts (bft_claims.h)
    uint64_t kh_ctr_tick[KH_CTR_SLOTS] = {};            // last use: a handle queried on more streams than slots recycles the least recently used
    uint64_t kh_ctr_clock = 0;
    hipEvent_t kh_ctr_ev[KH_CTR_SLOTS] = {};            // end of the slot's last launch (recorded once half of the slots are in use)
    int kh_ctr_pending = -1;
    int kh_ctr_used = 0;
    bool kh_ctr_failed = false;
    DevBuf sq_codes, sq_bad, sq_npos, sq_poff, sq_tmp, sq_cs, sq_tile;  // scratch of the sequence queries (grown, never shrunk)
    hipStream_t sq_stream = nullptr;
    bool sq_used = false;
    uint64_t sq_units = 0;
    DevBuf qc_cs, qc_tmp;            // scratch of the resident colour-list queries: colour-set id per k-mer, the scan's temporary (grown, never shrunk)
    hipStream_t qc_stream = nullptr;
    hipEvent_t qc_ev = nullptr;  // where the last use of the scratch ends (the stream it ran on is the caller's: it may be gone by the next call)
    bool qc_used = false;  // bound on the blocks of k-mer positions the sequence kernel deals out (claim_counters)
    bool inject_build_failure = false;  // test hook: the next bft_gpu_build fails right before its commit point (one shot)
    bool opt_build_stages = false;      // "build_stages": the next builds record GPU time and bytes per stage (bft_gpu_build_stages)
    struct Stage { std::string name; double ms, bytes; };
    std::vector<Stage> stages;          // of the last build
};

static int grid_for(uint64_t nblk) { return bft_grid_for(nblk); }
// rows per probe block of the suffix-group search ("query_probe": 4 or 8) -> BftImage::probe_big
static uint32_t probe_mode(int rows) { return rows == 8 ? 1u : 0u; }

static int set_device(bft_gpu* h) {
    HIPCK(hipSetDevice(h->device));
    bft_pool_set_stream(h->device, h->stream);
    return 0;
}

// Every ABI call makes the handle's GPU current; the caller's current device is put back when the call returns.
struct DeviceScope {
    int prev = -1;
    DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define ENTER(h)     \
    DeviceScope ds_; \
    CK(set_device(h))

// A *_dev entry point launched on a caller's stream: remember where that work ends.
#define BFT_MAX_EXT_STREAMS 8
static int note_foreign_stream(bft_gpu* h, hipStream_t s) {
    if (s == h->stream) return 0;
    bft_gpu::ExtEv* slot = nullptr;
    for (auto& e : h->ext)
        if (e.stream == s) slot = &e;
    if (!slot && h->ext.size() < BFT_MAX_EXT_STREAMS) {
        hipEvent_t ev = nullptr;
        HIPCK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        h->ext.push_back({s, ev, false});
        slot = &h->ext.back();
    }
    if (!slot) {  // more caller streams than slots: the oldest slot's work is drained and the slot moves on
        slot = &h->ext[0];
        if (slot->pending) HIPCK(hipEventSynchronize(slot->ev));
        slot->stream = s;
    }
    HIPCK(hipEventRecord(slot->ev, s));
    slot->pending = true;
    return 0;
}
// Before image arrays are released, rewritten or re-derived: the queries the caller still has in flight must have drained.
static int wait_foreign_stream(bft_gpu* h) {
    for (auto& e : h->ext)
        if (e.pending) {
            HIPCK(hipEventSynchronize(e.ev));
            e.pending = false;
        }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C-ABI: lifecycle
// ------------------------------------------------------------------------------------------------
extern "C" const char* bft_gpu_last_error(void) { return g_err.c_str(); }
extern "C" const char* bft_gpu_version(void) { return "bft-mi355x 0.1 (gfx950)"; }

extern "C" int bft_gpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int bft_gpu_create_seeded(int k, int device, int r1, int r2, bft_gpu** out) {
    if (!out) return fail(BFT_GPU_E_ARG, "out is NULL");
    *out = nullptr;
    DeviceScope ds_;
    if (!bft_valid_k(k)) return fail(BFT_GPU_E_ARG, "Length k (for k-mers) must be in [9,126] (a multiple of 9 for reference-compatible indexes, src/main.c:61-63)");
    int ndev = 0;
    HIPCK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(BFT_GPU_E_HIP, "no HIP device: this library has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(BFT_GPU_E_ARG, "bad device index");
    bft_gpu* h = new bft_gpu();
    h->k = k;
    h->L = k / 9;
    h->W = bft_words_for_k(k);
    h->B = bft_bytes_for_k(k);
    h->device = device;
    h->r1 = r1 > 0 ? r1 : BFT_DEFAULT_R1;
    h->r2 = r2 > 0 ? r2 : BFT_DEFAULT_R2;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) {
        delete h;
        return fail(BFT_GPU_E_HIP, "hipSetDevice/hipStreamCreate failed");
    }
    bft_pool_set_stream(device, h->stream);
    h->hashmod.resize(16384);
    bft_make_hashmod(h->r1, h->r2, h->hashmod.data());
    if (h->d_hashmod.alloc(16384 * 4) != 0 ||
        hipMemcpy(h->d_hashmod.p, h->hashmod.data(), 16384 * 4, hipMemcpyHostToDevice) != hipSuccess) {
        bft_gpu_free(h);
        return fail(BFT_GPU_E_HIP, "hash table upload failed");
    }
    memset(&h->im, 0, sizeof(h->im));
    *out = h;
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_create(int k, int device, bft_gpu** out) { return bft_gpu_create_seeded(k, device, 0, 0, out); }

static void drain_events(bft_gpu* h) {
    for (auto& pr : h->pending_ev) {
        float ms = 0;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            h->kernel_ms += ms;
            h->kernel_launches++;
        }
        h->free_ev.push_back(pr.first);  // back to the handle's pool
        h->free_ev.push_back(pr.second);
    }
    h->pending_ev.clear();
}

// Timed launches: a pair of pooled events around the kernel (no event is created on the launch path once the pool is warm).
static int timing_begin(bft_gpu* h, hipStream_t s, hipEvent_t* e0, hipEvent_t* e1) {
    *e0 = *e1 = nullptr;
    if (!h->timing) return 0;
    if (h->pending_ev.size() >= 4096) drain_events(h);
    for (hipEvent_t* e : {e0, e1}) {
        if (!h->free_ev.empty()) { *e = h->free_ev.back(); h->free_ev.pop_back(); }
        else HIPCK(hipEventCreate(e));
    }
    HIPCK(hipEventRecord(*e0, s));
    return 0;
}
static int timing_end(bft_gpu* h, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    if (!e0) return 0;
    HIPCK(hipEventRecord(e1, s));
    h->pending_ev.push_back({e0, e1});
    return 0;
}

extern "C" void bft_gpu_free(bft_gpu* h) {
    if (!h) return;
    DeviceScope ds_;
    (void)hipSetDevice(h->device);
    bft_pool_set_stream(h->device, h->stream);
    drain_events(h);
    (void)wait_foreign_stream(h);
    for (hipEvent_t e : h->free_ev) (void)hipEventDestroy(e);
    h->free_ev.clear();
    for (auto& e : h->ext) (void)hipEventDestroy(e.ev);
    h->ext.clear();
    if (h->qc_ev) { (void)hipEventDestroy(h->qc_ev); h->qc_ev = nullptr; }
    const hipStream_t s = h->stream;
    if (s) (void)hipStreamSynchronize(s);
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); h->stream2 = nullptr; }
    delete h;                      // its buffers go to the cache under this stream's tag ...
    bft_pool_drop_stream(s);       // ... which is replaced by "drained" here (bft_gpu_cache_release gives the cache back to the runtime)
    bft_pool_set_stream(-1, nullptr);
    if (s) (void)hipStreamDestroy(s);
}

extern "C" int bft_gpu_genome_name(bft_gpu* h, uint32_t id_genome, char* out, uint32_t cap) {
    if (!h || !out || cap == 0) return fail(BFT_GPU_E_ARG, "NULL argument");
    const std::string name = id_genome < h->genomes.size() ? h->genomes[id_genome] : "genome_" + std::to_string(id_genome);
    if (name.size() + 1 > cap) return fail(BFT_GPU_E_NOSPACE, "name buffer too small");
    memcpy(out, name.c_str(), name.size() + 1);
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_add_genome(bft_gpu* h, const char* name, uint32_t* id_genome) {
    if (!h || !name) return fail(BFT_GPU_E_ARG, "NULL argument");
    h->genomes.push_back(name);
    if (id_genome) *id_genome = (uint32_t)h->genomes.size() - 1;
    return BFT_GPU_OK;
}

// ------------------------------------------------------------------------------------------------
// insertion log
// ------------------------------------------------------------------------------------------------
static uint32_t log_comp_bits(const bft_gpu* h) {  // id bits of a composite log, 0: this handle logs k-mers + ids
    const int room = 63 - 2 * h->k;
    return (h->W == 1 && h->opt_comp_log && !h->opt_no_composite && room >= 7) ? (uint32_t)std::min(room, 24) : 0u;
}
// composites -> k-mers + ids, for whatever cannot take composites (the id array is allocated here)
static int log_decompose(bft_gpu* h) {
    if (!h->log_comp) return 0;
    CK(wait_foreign_stream(h));
    DevBuf ng;
    CK(ng.alloc(std::max<uint64_t>(h->log_cap, 1) * 4));
    if (h->log_n) {
        hipLaunchKernelGGL(k_log_decompose, dim3(grid_for((h->log_n + 255) / 256)), dim3(256), 0, h->stream, h->log_k.as<uint64_t>(), h->log_n, h->log_gb, ng.as<uint32_t>());
        HIPCK(hipGetLastError());
    }
    HIPCK(hipStreamSynchronize(h->stream));
    h->log_g.swap(ng);
    h->log_comp = false;
    return 0;
}
static int log_reserve(bft_gpu* h, uint64_t need) {
    if (h->log_n == 0 && h->log_cap == 0) {  // an empty log: its format is decided here
        h->log_gb = log_comp_bits(h);
        h->log_comp = h->log_gb != 0;
    }
    if (need <= h->log_cap) return 0;
    uint64_t ncap = std::max<uint64_t>(need, h->log_cap * 2);
    ncap = std::max<uint64_t>(ncap, 1 << 16);
    // The log is about to move: batches still being packed into it on a caller's stream (bft_gpu_insert_kmers_dev_async) finish
    // first -- the copy below would miss their rows and the old block would go back to the cache with writes in flight.
    CK(wait_foreign_stream(h));
    DevBuf nk, ng;
    CK(nk.alloc(ncap * h->W * 8));
    if (!h->log_comp) CK(ng.alloc(ncap * 4));
    if (h->log_n) {
        for (int w = 0; w < h->W; w++)
            HIPCK(hipMemcpyAsync(nk.as<uint64_t>() + (uint64_t)w * ncap, h->log_k.as<uint64_t>() + (uint64_t)w * h->log_cap,
                                 h->log_n * 8, hipMemcpyDeviceToDevice, h->stream));
        if (!h->log_comp) HIPCK(hipMemcpyAsync(ng.p, h->log_g.p, h->log_n * 4, hipMemcpyDeviceToDevice, h->stream));
    }
    // (also without a copy: a fresh block may come from the cache with work of this handle's stream still queued on it, and the
    // next writer may be a caller's stream, which is not ordered behind ours)
    HIPCK(hipStreamSynchronize(h->stream));
    h->log_k.swap(nk);
    h->log_g.swap(ng);
    h->log_cap = ncap;
    return 0;
}

template <int W>
static int launch_pack(bft_gpu* h, const uint8_t* d_packed, uint64_t n, uint32_t gid, hipStream_t s) {
    const uint64_t nblk = (n + BFT_BLOCK - 1) / BFT_BLOCK;
    hipLaunchKernelGGL(k_pack_to_tform<W>, dim3(grid_for(nblk)), dim3(BFT_BLOCK), 0, s, d_packed, n, h->B,
                       h->k, h->log_k.as<uint64_t>(), h->log_cap, h->log_n, h->log_g.as<uint32_t>(), gid, h->log_comp ? h->log_gb : 0u);
    HIPCK(hipGetLastError());
    return 0;
}

// insertKmers on a device-resident batch.  !ordered: on the handle's stream, synchronised before returning (the caller may
// reuse d_kmers at once).  ordered: stream-ordered on the caller's stream s (NULL = the null stream): nothing waits; the build waits for s.
// own_async: on the handle's stream, NOT synchronised (the caller keeps d_kmers alive until that stream has passed: the pinned ring).
static int insert_dev(bft_gpu* h, const void* d_kmers, uint64_t n, uint32_t id_genome, hipStream_t s, bool ordered, bool own_async = false) {
    if (!h || (!d_kmers && n)) return fail(BFT_GPU_E_ARG, "NULL argument");
    if (id_genome >= BFT_MAX_GENOME_ID) return fail(BFT_GPU_E_ARG, "id_genome out of range (must be below 2^24)");
    if (n == 0) return BFT_GPU_OK;
    ENTER(h);
    // No bound on the pairs an index holds (the reference has none, src/insertNode.c:18-36): a batch beyond what one sort takes is
    // inserted in pieces, and the log is merged into the index before it reaches "flush_pairs".
    if (n > h->opt_flush_pairs) {
        const uint64_t half = n / 2;
        CK(insert_dev(h, d_kmers, half, id_genome, s, ordered, own_async));
        return insert_dev(h, (const uint8_t*)d_kmers + half * (uint64_t)h->B, n - half, id_genome, s, ordered, own_async);
    }
    if (h->log_n && h->log_n + n > h->opt_flush_pairs) CK(bft_gpu_build(h));
    CK(log_reserve(h, h->log_n + n));
    if (h->log_comp && (id_genome >> h->log_gb) != 0) CK(log_decompose(h));  // (an id the composites have no room for)
    const hipStream_t run = ordered ? s : h->stream;
    const uint8_t* p = (const uint8_t*)d_kmers;
    switch (h->W) {
    case 1: CK(launch_pack<1>(h, p, n, id_genome, run)); break;
    case 2: CK(launch_pack<2>(h, p, n, id_genome, run)); break;
    case 3: CK(launch_pack<3>(h, p, n, id_genome, run)); break;
    default: CK(launch_pack<4>(h, p, n, id_genome, run)); break;
    }
    if (ordered) CK(note_foreign_stream(h, s));
    else if (!own_async) HIPCK(hipStreamSynchronize(h->stream));
    if (h->log_n > 0 && id_genome < h->log_last_gid) h->log_g_sorted = false;
    h->log_last_gid = id_genome;
    h->log_n += n;
    if (!h->lb_gid.empty() && h->lb_gid.back() == id_genome) h->lb_end.back() = h->log_n;
    else { h->lb_end.push_back(h->log_n); h->lb_gid.push_back(id_genome); }
    h->max_gid_seen = std::max(h->max_gid_seen, id_genome);
    h->any_insert = true;
    return BFT_GPU_OK;
}
extern "C" int bft_gpu_insert_kmers_dev(bft_gpu* h, const void* d_kmers, uint64_t n, uint32_t id_genome) { return insert_dev(h, d_kmers, n, id_genome, nullptr, false); }
extern "C" int bft_gpu_insert_kmers_dev_async(bft_gpu* h, const void* d_kmers, uint64_t n, uint32_t id_genome, void* hip_stream) {
    return insert_dev(h, d_kmers, n, id_genome, (hipStream_t)hip_stream, true);
}

extern "C" int bft_gpu_insert_kmers(bft_gpu* h, const uint8_t* kmers, uint64_t n, uint32_t id_genome) {
    if (!h || (!kmers && n)) return fail(BFT_GPU_E_ARG, "NULL argument");
    if (id_genome >= BFT_MAX_GENOME_ID) return fail(BFT_GPU_E_ARG, "id_genome out of range (must be below 2^24)");
    if (n == 0) return BFT_GPU_OK;
    ENTER(h);
    if (n * (uint64_t)h->B <= bft_gpu::RING_SLOT) {  // through the pinned ring (see struct bft_gpu)
        if (!h->ring && hipHostMalloc((void**)&h->ring, bft_gpu::RING_SLOT * bft_gpu::RING_SLOTS, hipHostMallocMapped) != hipSuccess) {
            h->ring = nullptr;
            (void)hipGetLastError();
        }
        if (h->ring) {
            const int slot = h->ring_next;
            if (!h->ring_ev[slot]) HIPCK(hipEventCreateWithFlags(&h->ring_ev[slot], hipEventDisableTiming));
            if (h->ring_busy[slot]) HIPCK(hipEventSynchronize(h->ring_ev[slot]));
            uint8_t* dst = h->ring + (size_t)slot * bft_gpu::RING_SLOT;
            memcpy(dst, kmers, n * (uint64_t)h->B);
            CK(insert_dev(h, dst, n, id_genome, nullptr, false, true));
            HIPCK(hipEventRecord(h->ring_ev[slot], h->stream));
            h->ring_busy[slot] = true;
            h->ring_next = (slot + 1) % bft_gpu::RING_SLOTS;
            return BFT_GPU_OK;
        }
    }
    const uint64_t chunk = 1ull << 26;
    DevBuf tmp;
    CK(tmp.alloc(std::min(n, chunk) * h->B));
    for (uint64_t a = 0; a < n; a += chunk) {
        const uint64_t m = std::min(chunk, n - a);
        HIPCK(hipMemcpy(tmp.p, kmers + a * h->B, m * h->B, hipMemcpyHostToDevice));
        CK(bft_gpu_insert_kmers_dev(h, tmp.p, m, id_genome));
    }
    return BFT_GPU_OK;
}

// ------------------------------------------------------------------------------------------------
// bulk build
// ------------------------------------------------------------------------------------------------
static int bits_for(uint64_t v) {
    int b = 1;
    while (b < 64 && (v >> b)) b++;
    return b;
}

// Stable LSD sort of `total` entries by (keys word 0..W-1 as one big integer, then g).
// keys: SoA with stride `stride`.  Result in okeys (stride ostride) / og.
static int sort_pairs(bft_gpu* h, const uint64_t* keys, uint64_t stride, const uint32_t* g, uint64_t total, uint64_t* okeys,
                      uint64_t ostride, uint32_t* og, bool g_already_ordered, bool is_log = false) {
    const int W = h->W;
    if (W == 1) {
        // one-word keys: sort the (key, genome) pairs themselves; the radix sort is stable, so a genome-id pass
        // first (skipped when the ids already ascend) followed by the key pass gives (T, genome) order
        DevBuf k2, g2;
        const uint64_t* kin = keys;
        const uint32_t* gin = g;
        if (!g_already_ordered) {
            CK(k2.alloc(total * 8));
            CK(g2.alloc(total * 4));
            const int gb = bits_for(h->max_gid_seen);
            CK((bft_rs::sort_pairs<uint32_t, uint64_t>(gin, kin, total, g2.as<uint32_t>(), k2.as<uint64_t>(), 0, gb, h->stream)));
            kin = k2.as<uint64_t>();
            gin = g2.as<uint32_t>();
        }
        CK((bft_rs::sort_pairs<uint64_t, uint32_t>(kin, gin, total, okeys, og, 0, 2 * h->k, h->stream)));
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(h->stream));
        return 0;
    }
    DevBuf perm, perm2, ku, ku2, kg2;
    CK(perm.alloc(total * 4));
    CK(perm2.alloc(total * 4));
    CK(ku.alloc(total * 8));
    CK(ku2.alloc(total * 8));
    const int grid = grid_for((total + 255) / 256);
    hipLaunchKernelGGL(k_iota, dim3(grid), dim3(256), 0, h->stream, perm.as<uint32_t>(), total);
    // pass 0: genome id (least significant); skipped when the input is already in genome-id order
    // (ids inserted in non-decreasing order, as the reference requires: the stable key passes keep it)
    if (!g_already_ordered) {
        CK(kg2.alloc(total * 4));
        const int gb = bits_for(h->max_gid_seen);
        CK((bft_rs::sort_pairs<uint32_t, uint32_t>(g, perm.as<uint32_t>(), total, kg2.as<uint32_t>(), perm2.as<uint32_t>(), 0, gb, h->stream)));
        perm.swap(perm2);
    }
    // passes over the key words, least significant word (W-1) first
    for (int w = W - 1; w >= 0; w--) {
        const int nbits = (w == 0) ? (2 * h->k - 64 * (W - 1)) : 64;
        const uint64_t* kin = keys + (uint64_t)w * stride;
        if (w != W - 1 || !g_already_ordered) {  // (the first pass of an id-ordered log: the permutation is still the identity, the word is sorted where it lies)
            hipLaunchKernelGGL(k_gather<uint64_t>, dim3(grid), dim3(256), 0, h->stream, kin, perm.as<uint32_t>(), ku.as<uint64_t>(), total);
            kin = ku.as<uint64_t>();
        }
        CK((bft_rs::sort_pairs<uint64_t, uint32_t>(kin, perm.as<uint32_t>(), total, ku2.as<uint64_t>(), perm2.as<uint32_t>(), 0, nbits, h->stream)));
        perm.swap(perm2);
    }
    // (one pass over the permutation for every word and the id, instead of a pass each; the ids of the LOG come out of the table of its insert calls)
    DevBuf lbe, lbg;
    uint32_t nlb = 0;
    if (is_log && !h->lb_end.empty() && h->lb_end.size() <= (1u << 16) && h->lb_end.back() == total) {
        nlb = (uint32_t)h->lb_end.size();
        CK(lbe.alloc(nlb * 8));
        CK(lbg.alloc(nlb * 4));
        HIPCK(hipMemcpyAsync(lbe.p, h->lb_end.data(), nlb * 8, hipMemcpyHostToDevice, h->stream));
        HIPCK(hipMemcpyAsync(lbg.p, h->lb_gid.data(), nlb * 4, hipMemcpyHostToDevice, h->stream));
    }
    hipLaunchKernelGGL(k_gather_pairs, dim3(grid), dim3(256), 0, h->stream, keys, stride, W, g, perm.as<uint32_t>(), okeys, ostride, og, total, lbe.as<uint64_t>(), lbg.as<uint32_t>(), nlb);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(h->stream));
    return 0;
}

template <class T>
static int upload(DevBuf& d, const std::vector<T>& v) {
    CK(d.alloc(v.size() * sizeof(T)));
    if (!v.empty()) HIPCK(hipMemcpy(d.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static uint64_t image_bytes(const bft_gpu* h) {
    return h->d_nodes.bytes + h->d_bfT.bytes + h->d_ccs.bytes + h->d_f2w.bytes + h->d_clus.bytes + h->d_child.bytes + h->d_tk.bytes + h->d_tcol.bytes +
           h->d_uck.bytes + h->d_ucrow.bytes + h->d_cs_off.bytes + h->d_cs_ids.bytes + h->d_hashmod.bytes + h->d_cs_bm.bytes + h->d_ccx.bytes +
           h->d_f18.bytes + h->d_fent.bytes + h->d_rdir.bytes + h->d_rstart.bytes + h->d_rq.bytes + h->d_nph.bytes + h->d_kh.bytes + h->d_rspec.bytes + h->d_kh_ovf_k.bytes + h->d_kh_ovf_v.bytes;
}

static int tune_residency(bft_gpu* h);
static void ensure_claim_counters(bft_gpu* h);

// Points h->im at the device arrays of the handle (cannot fail).
static void point_image(bft_gpu* h, uint32_t nb_genomes) {
    BftImage& im = h->im;
    im.k = h->k;
    im.L = h->L;
    im.W = h->W;
    im.nb_genomes = nb_genomes;
    im.n_kmers = h->n_kmers;
    im.hashmod = h->d_hashmod.as<uint32_t>();
    im.nodes = h->d_nodes.as<BftNode>();
    im.bfT = h->d_bfT.as<uint8_t>();
    im.ccs = h->d_ccs.as<BftCC>();
    im.f2w = h->d_f2w.as<uint64_t>();
    im.clus = h->d_clus.as<uint64_t>();
    im.child = h->d_child.as<uint64_t>();
    im.tk = h->d_tk.as<uint64_t>();
    im.kh_lines = nullptr;    // (derive_kmer_hash)
    memset(&im.kh, 0, sizeof(im.kh));
    im.kh_ovf = nullptr;
    im.kh_ovf_val = nullptr;
    im.kh_ovf_n = 0;
    im.rspec = nullptr;
    im.walk_kh = 0;
    im.tcol = h->d_tcol.as<uint32_t>();
    im.uck = h->d_uck.as<uint64_t>();
    im.ucrow = h->d_ucrow.as<uint32_t>();
    im.cs_off = h->d_cs_off.as<uint32_t>();
    im.cs_ids = h->d_cs_ids.p;
    im.cs_w = h->cs_w;
    im.ccx = h->d_ccx.as<BftCCX>();
    im.f18 = h->d_f18.as<uint64_t>();
    im.fent = h->d_fent.as<uint64_t>();
    im.rdir = nullptr;  // derived after this call (derive_root_direct)
    im.rstart = nullptr;
    im.rq = nullptr;
    im.nph = nullptr;   // (derive_node_hash)
    im.nph_mask = 0;
    im.nph_no_uc = 0;
    h->has_cs_bm = false;  // the bitmap form of the colour-set dictionary is derived by the first colour-row query (ensure_cs_bitmaps)
    h->cs_bm_tried = false;
    h->d_cs_bm.release();
    im.emit_cs = 0;
    h->tuned_wgs = 0;
    h->tuned_probe = 0;
    h->im.probe_big = probe_mode(h->opt_probe);
}

// Root direct table: one thread per 18-bit prefix evaluates the root level's Bloom probe + CC lookup on the bound image.
__global__ void k_root_direct(BftImage im, uint64_t* __restrict__ out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (1u << 18)) return;
    const BftRootGlobal root(im);
    out[r] = bft_root_direct_entry(im, root, im.nodes[0], r);
}

template <int W>
__global__ void k_root_ranges(BftImage im, uint32_t root_uc_n, uint32_t* __restrict__ out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > (1u << 18)) return;
    out[r] = bft_root_range_entry<W>(im, r, r < (1u << 18) ? im.rdir[r] : 0ull, root_uc_n);
}
__global__ void k_root_ranges_check(const uint64_t* __restrict__ rdir, uint32_t* __restrict__ rs) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (1u << 18)) return;
    const uint32_t a = rs[r];
    if (!(a & BFT_RSTART_SPECIAL) && !bft_root_range_ok(a, rs[r + 1], rdir[r])) rs[r] = a | BFT_RSTART_SPECIAL;  // (readers mask the flag)
}

// Root quartile table (BFT_RQ_*): one thread per root prefix; for a plain group three lower-bound searches on the top two of the
// key bits that follow the root digit.
template <int W>
__global__ void k_root_quartiles(BftImage im, const uint32_t* __restrict__ rs, uint32_t* __restrict__ out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (1u << 18)) return;
    out[r] = bft_root_quartile_entry<W>(im, rs[r], rs[r + 1]);
}

// Node prefix hash.  k_nph_ccnode: the node of every CC (one thread per node).  k_nph_fill: one WAVEFRONT per CC below the root,
// one lane per filter2 word (48 bits + the running rank of the bits before it, so a word knows its first cluster without
// looking at the others): set bits -> clusters -> entries, each entry claims a slot of its (node, prefix) bucket with atomicCAS;
// keys whose bucket is full are dropped (the lookup then takes the container path).  (One thread per node -- a few thousand
// threads, each a chain of ~10^3 dependent loads -- took 16 ms on config 3.)
// stats: [0] inserted, [1] dropped, [2] nodes below the root that hold UC rows.
__global__ void k_nph_ccnode(const BftNode* __restrict__ nodes, uint32_t n_nodes, uint32_t* __restrict__ cc_node, unsigned long long* __restrict__ stats) {
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < n_nodes; m += gridDim.x * blockDim.x) {
        const BftNode nd = nodes[m];
        if (m > 0 && nd.uc_n) atomicAdd(&stats[2], 1ull);
        for (uint32_t c = 0; c < nd.ncc; c++) cc_node[nd.cc_first + c] = m;
    }
}
__global__ __launch_bounds__(256) void k_nph_fill(BftImage im, const uint32_t* __restrict__ cc_node, uint32_t first_cc, uint32_t n_ccs, uint64_t* __restrict__ tab,
                                                  uint64_t mask, unsigned long long* __restrict__ stats) {
    const uint32_t lane = threadIdx.x & 63u, wpb = blockDim.x >> 6;
    for (uint32_t c = first_cc + blockIdx.x * wpb + (threadIdx.x >> 6); c < n_ccs; c += gridDim.x * wpb) {
        const BftCC cc = im.ccs[c];
        const uint32_t m = cc_node[c];
        const uint32_t nw = ((1u << (18 - cc.s)) + BFT_F2_BITS_PER_WORD - 1) / BFT_F2_BITS_PER_WORD;
        uint32_t placed_n = 0, dropped_n = 0;
        for (uint32_t w = lane; w < nw; w += 64) {
            const uint64_t fw = im.f2w[cc.f2_off + w];
            uint64_t bits = fw & ((1ull << BFT_F2_BITS_PER_WORD) - 1ull);
            uint32_t clu = (uint32_t)(fw >> BFT_F2_BITS_PER_WORD);  // clusters before this word
            while (bits) {
                const uint32_t b = (uint32_t)__builtin_ctzll(bits);
                bits &= bits - 1ull;
                const uint32_t pu = w * BFT_F2_BITS_PER_WORD + b;
                const uint64_t ce = im.clus[cc.clus_off + clu++];
                const uint32_t len = (ce & BFT_CLUS_MULTI) ? (uint32_t)((ce >> BFT_CLUS_LEN_SHIFT) & 0xFFFFu) : 1u;
                for (uint32_t j = 0; j < len; j++) {
                    const uint64_t ent = (ce & BFT_CLUS_MULTI) ? im.child[cc.child_off + (uint32_t)ce + j] : ce;
                    const uint32_t r = (pu << cc.s) | ((uint32_t)(ent >> BFT_CHILD_PV_SHIFT) & 0xFFu);
                    const uint64_t key = bft_nph_key(m, r);
                    unsigned long long* bk = (unsigned long long*)(tab + bft_nph_bucket(key, mask) * (2 * BFT_NPH_SLOTS));
                    bool placed = false;
                    for (int s = 0; s < BFT_NPH_SLOTS && !placed; s++)
                        if (atomicCAS(&bk[2 * s], (unsigned long long)BFT_NPH_EMPTY, (unsigned long long)key) == BFT_NPH_EMPTY) {
                            bk[2 * s + 1] = ent;
                            placed = true;
                        }
                    placed_n += placed;
                    dropped_n += !placed;
                }
            }
        }
        if (placed_n) atomicAdd(&stats[0], (unsigned long long)placed_n);
        if (dropped_n) atomicAdd(&stats[1], (unsigned long long)dropped_n);
    }
}

// Derives the node prefix hash of the image h->im points at.  An accelerator only: without it the walk keeps the container path.
static void derive_node_hash(bft_gpu* h) {
    h->im.nph = nullptr;
    h->im.nph_mask = 0;
    h->im.nph_no_uc = 0;
    h->nph_inserted = h->nph_dropped = 0;
    const uint64_t n_nodes = h->idx_sizes[0] / sizeof(BftNode);
    if (!h->opt_node_hash || (h->opt_node_hash == 1 && h->im.kh_lines != nullptr && !h->opt_walk_hash) || n_nodes <= 1 || h->info[6] == 0) {
        h->d_nph.release();
        return;
    }
    // keys = the prefixes of the nodes below the root = all prefixes (info[6]) minus the root's (nb_elem of its CCs)
    uint64_t root_prefixes = 0;
    {
        std::vector<BftCC> rc(h->root_ncc);
        if (h->root_ncc && hipMemcpy(rc.data(), h->d_ccs.p, rc.size() * sizeof(BftCC), hipMemcpyDeviceToHost) != hipSuccess) return;  // (the root's CCs come first)
        for (const BftCC& c : rc) root_prefixes += c.nb_elem;
    }
    const uint64_t keys = h->info[6] > root_prefixes ? h->info[6] - root_prefixes : 0;
    uint64_t nbk = 1024;
    while (nbk < keys) nbk <<= 1;  // <= 1 key per 4-slot bucket on average
    const size_t bytes = nbk * BFT_NPH_SLOTS * 16;
    DevBuf stats;
    if (h->d_nph.alloc(bytes) != 0 || stats.alloc_zero(24, h->stream) != 0) { h->d_nph.release(); return; }
    if (hipMemsetAsync(h->d_nph.p, 0xFF, bytes, h->stream) != hipSuccess) { h->d_nph.release(); return; }
    BftImage tmp = h->im;
    const uint64_t n_ccs = h->idx_sizes[2] / sizeof(BftCC);
    DevBuf cc_node;
    if (cc_node.alloc(n_ccs * 4) != 0) { h->d_nph.release(); return; }
    hipLaunchKernelGGL(k_nph_ccnode, dim3(grid_for((n_nodes + 255) / 256)), dim3(256), 0, h->stream, h->d_nodes.as<BftNode>(), (uint32_t)n_nodes, cc_node.as<uint32_t>(),
                       stats.as<unsigned long long>());
    hipLaunchKernelGGL(k_nph_fill, dim3(grid_for((n_ccs + 3) / 4)), dim3(256), 0, h->stream, tmp, cc_node.as<uint32_t>(), h->root_ncc, (uint32_t)n_ccs,
                       h->d_nph.as<uint64_t>(), nbk - 1, stats.as<unsigned long long>());
    unsigned long long st[3] = {0, 0, 0};
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(st, stats.p, 24, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) {
        h->d_nph.release();
        return;
    }
    h->nph_inserted = st[0];
    h->nph_dropped = st[1];
    h->im.nph = h->d_nph.as<uint64_t>();
    h->im.nph_mask = nbk - 1;
    h->im.nph_no_uc = st[2] == 0 ? 1u : 0u;
}

// The build of the k-mer hash for a table (tk, tcol) that is complete on the device, started on the handle's second stream: the build
// assembles the containers on `stream` meanwhile.  The build sorts and gathers and starves what runs beside it of memory bandwidth and
// latency (k_prefix_flags over the whole table: 0.2 ms alone, 2.8 ms beside it; the root's single-workgroup k_assign_cc: 0.8 -> 3.5 ms),
// so it starts behind those (`after`: an event of the assembly stream) and overlaps the chain of small kernels and read-back counts that
// follows.  kh_finish waits for it.  Any failure just leaves the image without the table.
struct KhFill {
    DevBuf buf, status, ovf_k, ovf_v;
    BftKhScratch scratch;
    BftKhGeo geo;
    uint64_t lines_cap = 0, lines_used = 0;
    uint32_t ovf_n = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr, ew = nullptr;
    hipStream_t s2 = nullptr;
    bool started = false, prepared = false;
    std::thread prep;  // kh_prepare_async
    KhFill() { memset(&geo, 0, sizeof(geo)); }
    ~KhFill() {  // (a build that fails half-way: the fill must be over before its buffers go back to the cache)
        if (prep.joinable()) prep.join();
        if (started && s2) (void)hipStreamSynchronize(s2);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (ew) (void)hipEventDestroy(ew);
    }
};
static bool kh_wanted(const bft_gpu* h, uint64_t nk) { return h->opt_kmer_hash && nk > 0 && nk < (1ull << 31); }
static bool kh_geo_ok(const bft_gpu* h, const BftKhGeo& g) { return bft_kh_has_kernels(h->W, g.S) && g.nl + BFT_KH_TAIL_LINES < (1ull << 32); }
// the host side of the build that needs no table yet: the second stream, the table's memory, the events (a millisecond of driver
// calls for a gigabyte table that is not in the cache).  n_values: the colour sets of the index, or -- before they are known -- a bound
// (sets <= k-mers): fewer slots per line at worst, i.e. a table allocated larger than needed, never smaller.
static void kh_prepare(bft_gpu* h, uint64_t nk, uint64_t n_values, KhFill& f) {
    f.prepared = false;
    if (!kh_wanted(h, nk)) return;
    if (!h->stream2) {
        int lo = 0, hi = 0;  // (numerically larger = lower priority)
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; (void)hipGetLastError(); }
        if (hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, lo) != hipSuccess) { h->stream2 = nullptr; (void)hipGetLastError(); return; }
    }
    f.geo = bft_kh_geometry(h->k, nk, std::max<uint64_t>(1, n_values), h->opt_kh_load);
    if (!kh_geo_ok(h, f.geo)) return;
    f.lines_cap = f.geo.nl + BFT_KH_TAIL_LINES;
    if (f.buf.alloc(f.lines_cap * BFT_KH_LINE_WORDS * 8) != 0 || f.status.alloc(16) != 0 || f.ovf_k.alloc((size_t)BFT_KH_OVF_CAP * h->W * 8) != 0 ||
        f.ovf_v.alloc((size_t)BFT_KH_OVF_CAP * 4) != 0)
        return;
    f.prepared = (f.e0 || hipEventCreate(&f.e0) == hipSuccess) && (f.e1 || hipEventCreate(&f.e1) == hipSuccess) &&
                 (f.ew || hipEventCreateWithFlags(&f.ew, hipEventDisableTiming) == hipSuccess);
    if (!f.prepared) { (void)hipGetLastError(); f.buf.release(); }
}
// ... on a thread of its own while the build's stream is busy with the colour sets
static void kh_prepare_async(bft_gpu* h, uint64_t nk, uint64_t n_values, KhFill& f) {
    KhFill* fp = &f;
    try {
        f.prep = std::thread([h, nk, n_values, fp] {
            if (hipSetDevice(h->device) != hipSuccess) { (void)hipGetLastError(); return; }
            bft_pool_set_stream(h->device, h->stream);
            kh_prepare(h, nk, n_values, *fp);
        });
    } catch (...) {  // (no thread to be had: kh_start prepares on the caller's)
    }
}
// run: the stream the build is enqueued on (the handle's second stream behind `after`, or the handle's own)
static void kh_start(bft_gpu* h, const uint64_t* d_tk, const uint32_t* d_tcol, uint64_t nk, uint64_t n_sets, KhFill& f, hipStream_t after, hipStream_t run) {
    if (f.prep.joinable()) f.prep.join();
    if (!kh_wanted(h, nk)) return;
    const BftKhGeo geo = bft_kh_geometry(h->k, nk, std::max<uint64_t>(1, n_sets), h->opt_kh_load);
    if (!kh_geo_ok(h, geo)) return;
    // (prepared with a bound on the colour sets: keep the block unless the real geometry needs a tenth less)
    const uint64_t need = geo.nl + BFT_KH_TAIL_LINES;
    if (!f.prepared || f.lines_cap < need || f.lines_cap > need + need / 10) kh_prepare(h, nk, n_sets, f);
    if (!f.prepared) return;
    f.geo = geo;
    if (!run) run = h->stream2;
    bool ok = true;
    if (after && after != run) ok = hipEventRecord(f.ew, after) == hipSuccess && hipStreamWaitEvent(run, f.ew, 0) == hipSuccess;
    ok = ok && hipEventRecord(f.e0, run) == hipSuccess;
    if (run != h->stream) bft_stage("+k-mer hash build starts (side stream)", 0, run);
    ok = ok && bft_kh_sort(d_tk, d_tcol, nk, h->k, h->W, f.geo, f.scratch, run) == 0 && bft_kh_lay(nk, h->k, h->W, f.geo, f.buf.as<uint64_t>(), f.ovf_k.as<uint64_t>(), f.ovf_v.as<uint32_t>(), f.status.as<uint32_t>(), f.scratch, run) == 0 &&
         hipEventRecord(f.e1, run) == hipSuccess;
    // (keys: table in, records out; the sort's passes over the records; the lines written once)
    bft_stage(run != h->stream ? "+k-mer hash build ends (side stream)" : "k-mer hash build", (double)nk * (8.0 * h->W + 4) * 8 + (double)f.geo.nl * 64, run);
    f.s2 = run;
    f.started = true;  // (whatever was enqueued is waited for before the buffers go anywhere)
    if (!ok) { (void)hipGetLastError(); (void)hipStreamSynchronize(run); f.started = false; f.buf.release(); }
}
static bool kh_finish(bft_gpu* h, KhFill& f, double* ms) {
    bool ok = f.started && hipStreamSynchronize(f.s2) == hipSuccess;
    f.started = false;
    float t = 0;
    if (ok && hipEventElapsedTime(&t, f.e0, f.e1) == hipSuccess && ms) *ms = t;
    if (f.e0) (void)hipEventDestroy(f.e0);
    if (f.e1) (void)hipEventDestroy(f.e1);
    f.e0 = f.e1 = nullptr;
    uint32_t st[4] = {2, 0, 0, 0};
    if (ok) ok = hipMemcpy(st, f.status.p, 16, hipMemcpyDeviceToHost) == hipSuccess && st[0] == 0 && st[3] <= BFT_KH_OVF_CAP;
    for (DevBuf& b : f.scratch.b) b.release();
    f.lines_used = st[1];
    f.geo.maxd = st[2];  // (a lookup looks no further than the table's largest displacement)
    f.ovf_n = ok ? st[3] : 0;
    if (ok && f.ovf_n) {  // the overflow list, sorted by k-mer (a handful: on the host)
        const int W = h->W;
        std::vector<uint64_t> kk((size_t)f.ovf_n * W), k2(kk.size());
        std::vector<uint32_t> vv(f.ovf_n), v2(f.ovf_n), ord(f.ovf_n);
        ok = hipMemcpy(kk.data(), f.ovf_k.p, kk.size() * 8, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(vv.data(), f.ovf_v.p, vv.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
        for (uint32_t i = 0; i < f.ovf_n; i++) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return std::lexicographical_compare(&kk[(size_t)a * W], &kk[(size_t)a * W] + W, &kk[(size_t)b * W], &kk[(size_t)b * W] + W); });
        for (uint32_t i = 0; i < f.ovf_n; i++) {
            for (int w = 0; w < W; w++) k2[(size_t)i * W + w] = kk[(size_t)ord[i] * W + w];
            v2[i] = vv[ord[i]];
        }
        ok = ok && hipMemcpy(f.ovf_k.p, k2.data(), k2.size() * 8, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(f.ovf_v.p, v2.data(), v2.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { (void)hipGetLastError(); f.buf.release(); }
    if (getenv("BFT_GPU_VERBOSE"))
        fprintf(stderr, "[bft_gpu] k-mer hash: %s, %llu home lines (2^%u x %u), %u slots (%u-bit header fields, %u-byte bodies: %u key bits (%u below the hashed %u, %u of q) + %u + %u value bits), largest displacement %u, %u in the overflow list, %.2f ms\n",
                ok ? "built" : "NOT built", (unsigned long long)f.geo.nl, f.geo.hb - f.geo.t, f.geo.m, f.geo.S, f.geo.f, f.geo.wb, f.geo.kb, f.geo.restb, f.geo.hb, f.geo.qb, f.geo.db, f.geo.cb,
                f.geo.maxd, f.ovf_n, t);
    return ok;
}
// the table of a finished build becomes the image's
static void kh_adopt(bft_gpu* h, KhFill& f, double ms) {
    h->d_kh.swap(f.buf);
    h->kh_lines = f.geo.nl;
    h->kh_ms = ms;
    h->im.kh_lines = h->d_kh.as<uint64_t>();
    h->im.kh = f.geo;
    h->d_kh_ovf_k.swap(f.ovf_k);
    h->d_kh_ovf_v.swap(f.ovf_v);
    h->kh_ovf_n = f.ovf_n;
    if (!f.ovf_n) { h->d_kh_ovf_k.release(); h->d_kh_ovf_v.release(); }
    h->im.kh_ovf = f.ovf_n ? h->d_kh_ovf_k.as<uint64_t>() : nullptr;
    h->im.kh_ovf_val = f.ovf_n ? h->d_kh_ovf_v.as<uint32_t>() : nullptr;
    h->im.kh_ovf_n = f.ovf_n;
}
static void kh_drop(bft_gpu* h) {
    h->d_kh.release();
    h->kh_lines = 0;
    h->kh_ms = 0;
    h->im.kh_lines = nullptr;
    memset(&h->im.kh, 0, sizeof(h->im.kh));
    h->d_kh_ovf_k.release();
    h->d_kh_ovf_v.release();
    h->kh_ovf_n = 0;
    h->im.kh_ovf = nullptr;
    h->im.kh_ovf_val = nullptr;
    h->im.kh_ovf_n = 0;
    h->im.walk_kh = 0;
}

// Derives the k-mer hash of the image h->im points at (BFT_KH_*), on the handle's own stream.  An accelerator only: without it every
// query walks the containers.
static void derive_kmer_hash(bft_gpu* h) {
    kh_drop(h);
    if (!kh_wanted(h, h->n_kmers)) return;
    KhFill f;
    kh_start(h, h->d_tk.as<uint64_t>(), h->d_tcol.as<uint32_t>(), h->n_kmers, h->n_sets, f, nullptr, h->stream);
    double ms = 0;
    if (kh_finish(h, f, &ms)) kh_adopt(h, f, ms);
}

// What the walk needs to look plain root groups up in the k-mer hash: one bit per root prefix, "not a plain suffix group of the root"
// (bit 31 of rstart[r]: child Node, UC rows) -- those keep the containers.
__global__ void k_rspec(const uint32_t* __restrict__ rstart, uint32_t* __restrict__ rspec) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= (1u << 18) / 32) return;
    uint32_t bits = 0;
    for (uint32_t b = 0; b < 32; b++) bits |= ((rstart[32 * w + b] & BFT_RSTART_SPECIAL) ? 1u : 0u) << b;
    rspec[w] = bits;
}
static void sync_walk_kh(bft_gpu* h) {
    h->im.walk_kh = 0;
    h->im.rspec = nullptr;
    if (!h->im.kh_lines || !h->rstart_ok) return;
    if (h->d_rspec.bytes < (1u << 18) / 8 && h->d_rspec.alloc((1u << 18) / 8) != 0) return;
    hipLaunchKernelGGL(k_rspec, dim3((1u << 18) / 32 / 256), dim3(256), 0, h->stream, h->d_rstart.as<uint32_t>(), h->d_rspec.as<uint32_t>());
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return;
    h->im.rspec = h->d_rspec.as<uint32_t>();
    h->im.walk_kh = 1;
}

// "compact_table": the k-mer hash holds every (k-mer, colour set) of the index, so the sorted table tk and tcol -- 12 of the image's
// 41 bytes per k-mer on the 100-genome index -- need not stay resident for presence, colour, branching and sequence queries.  They are
// dropped after a build and come back (a dump of the table's slots + one radix sort: milliseconds) when something asks for rows, an
// extraction, a merge, a .bft file, a packed image or the container walk.
__global__ void k_rows_from_words(const uint64_t* __restrict__ words, uint64_t n, int W, uint64_t* __restrict__ rows) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        for (int w = 0; w < W; w++) rows[i * W + w] = words[(uint64_t)w * n + i];
}
static bool compact_possible(const bft_gpu* h) { return h->im.kh_lines != nullptr && h->n_kmers > 0; }
static void drop_table(bft_gpu* h) {
    if (!h->opt_compact || h->table_dropped || !compact_possible(h)) return;
    h->d_tk.release();
    h->d_tcol.release();
    h->table_dropped = true;
    h->im.tk = nullptr;
    h->im.tcol = nullptr;
    h->info[12] = image_bytes(h);
}
static int ensure_table(bft_gpu* h) {
    if (!h->table_dropped) return 0;
    CK(wait_foreign_stream(h));
    const uint64_t n = h->n_kmers;
    const int W = h->W;
    DevBuf keys, vals, cnt, tmp, tk, tcol;
    CK(keys.alloc(n * W * 8));
    CK(vals.alloc(n * 4));
    CK(tk.alloc(n * W * 8));
    CK(tcol.alloc(n * 4));
    CK(cnt.alloc_zero(8, h->stream));
    CK(bft_kh_dump(h->im, keys.as<uint64_t>(), n, vals.as<uint32_t>(), cnt.as<unsigned long long>(), h->stream));
    unsigned long long got = 0;
    HIPCK(hipMemcpyAsync(&got, cnt.p, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    if (got != n) return fail(BFT_GPU_E_HIP, "k-mer hash does not hold the index (compact_table)");
    if (W == 1) {
        CK((bft_rs::sort_pairs<uint64_t, uint32_t>(keys.as<uint64_t>(), vals.as<uint32_t>(), n, tk.as<uint64_t>(), tcol.as<uint32_t>(), 0u, (unsigned)(2 * h->k), h->stream)));
    } else {  // two-word keys: the build's permutation sort (words apart, as the dump wrote them), then rows of two words
        DevBuf sorted;
        CK(sorted.alloc(n * W * 8));
        CK(sort_pairs(h, keys.as<uint64_t>(), n, vals.as<uint32_t>(), n, sorted.as<uint64_t>(), n, tcol.as<uint32_t>(), true));
        hipLaunchKernelGGL(k_rows_from_words, dim3(grid_for((n + 255) / 256)), dim3(256), 0, h->stream, sorted.as<uint64_t>(), n, W, tk.as<uint64_t>());
        HIPCK(hipGetLastError());
    }
    HIPCK(hipStreamSynchronize(h->stream));
    h->d_tk.swap(tk);
    h->d_tcol.swap(tcol);
    h->table_dropped = false;
    h->im.tk = h->d_tk.as<uint64_t>();
    h->im.tcol = h->d_tcol.as<uint32_t>();
    h->info[12] = image_bytes(h);
    return 0;
}

// Launch shape of the container walk when nothing was measured ("tune") or fixed by the caller: two 768-thread workgroups per CU (the
// arrangement that was best or within a few per cent of it on every index measured), 8-row probes once suffix groups
// hold dozens of rows, and the root range table unless most root prefixes are child Nodes (their lookups pay it for nothing).
static void default_launch_shape(bft_gpu* h) {
    h->tuned_wgs = 3;
    // (rows per ROOT prefix: the groups most queries end in hang off the root; deeper levels add prefixes, not rows)
    const uint64_t prefixes = std::max<uint64_t>(1, std::min<uint64_t>(h->info[6], 1ull << 18));
    h->tuned_probe = h->n_kmers / prefixes >= 48 ? 8 : 4;
    h->im.probe_big = probe_mode(h->opt_probe ? h->opt_probe : h->tuned_probe);
    if (h->opt_root_direct == 3 && h->rstart_ok) {
        h->tuned_rstart = h->info[5] * 4 < (1u << 18) ? 1 : 0;
        h->im.rstart = h->tuned_rstart ? h->d_rstart.as<uint32_t>() : nullptr;
        h->im.rq = h->tuned_rstart && h->rq_ok ? h->d_rq.as<uint32_t>() : nullptr;
    }
}

// Derives the table for the image h->im points at (after point_image).  An accelerator only: on any failure the walk simply
// keeps the container path (im.rdir == NULL).
static void derive_root_direct(bft_gpu* h) {
    h->im.rdir = nullptr;
    h->im.rstart = nullptr;
    h->im.rq = nullptr;
    h->rstart_ok = false;
    h->rq_ok = false;
    h->tuned_rstart = -1;
    if (!h->opt_root_direct || h->root_ncc == 0 || h->n_kmers == 0) return;
    if (h->d_rdir.bytes < (8u << 18) && h->d_rdir.alloc(8u << 18) != 0) return;
    BftImage tmp = h->im;
    tmp.rdir = nullptr;
    tmp.debug_stop = 0;
    hipLaunchKernelGGL(k_root_direct, dim3((1u << 18) / 256), dim3(256), 0, h->stream, tmp, h->d_rdir.as<uint64_t>());
    if (hipGetLastError() != hipSuccess) { (void)hipStreamSynchronize(h->stream); return; }
    // (ONE wait for the three tables, at whichever point this function is left: the kernels follow each other on the handle's stream, and a caller's
    // stream must not meet a table that is still being written)
    h->im.rdir = h->d_rdir.as<uint64_t>();
    if (h->opt_root_direct < 2) { if (hipStreamSynchronize(h->stream) != hipSuccess) h->im.rdir = nullptr; return; }
    const size_t rs_bytes = ((1u << 18) + 2) * 4;
    if (h->d_rstart.bytes < rs_bytes && h->d_rstart.alloc(rs_bytes) != 0) { if (hipStreamSynchronize(h->stream) != hipSuccess) h->im.rdir = nullptr; return; }
    tmp.rdir = h->im.rdir;
    const uint32_t root_uc_n = (uint32_t)h->info[14];  // rows of the root's UC (BftNode::uc_n of node 0: bft_gpu_info entry 14, set by the build / the blob)
    const dim3 g((1u << 18) / 256 + 1), b(256);
    switch (h->W) {
    case 1: hipLaunchKernelGGL(k_root_ranges<1>, g, b, 0, h->stream, tmp, root_uc_n, h->d_rstart.as<uint32_t>()); break;
    case 2: hipLaunchKernelGGL(k_root_ranges<2>, g, b, 0, h->stream, tmp, root_uc_n, h->d_rstart.as<uint32_t>()); break;
    case 3: hipLaunchKernelGGL(k_root_ranges<3>, g, b, 0, h->stream, tmp, root_uc_n, h->d_rstart.as<uint32_t>()); break;
    default: hipLaunchKernelGGL(k_root_ranges<4>, g, b, 0, h->stream, tmp, root_uc_n, h->d_rstart.as<uint32_t>()); break;
    }
    hipLaunchKernelGGL(k_root_ranges_check, g, b, 0, h->stream, h->im.rdir, h->d_rstart.as<uint32_t>());
    if (hipGetLastError() != hipSuccess) return;  // (no wait here: the stream is synchronised below, or by whoever uses the tables next)
    h->rstart_ok = true;
    h->im.rstart = h->d_rstart.as<uint32_t>();  // (mode 3: tune_residency decides whether it stays)
    h->rq_ok = false;
    if (!h->opt_root_quartiles || h->L < 2 || (h->d_rq.bytes < (4u << 18) && h->d_rq.alloc(4u << 18) != 0)) {
        if (hipStreamSynchronize(h->stream) != hipSuccess) { h->im.rdir = nullptr; h->im.rstart = nullptr; h->rstart_ok = false; }
        return;
    }
    tmp.rstart = h->im.rstart;
    const dim3 gq((1u << 18) / 256);
    switch (h->W) {
    case 1: hipLaunchKernelGGL(k_root_quartiles<1>, gq, b, 0, h->stream, tmp, h->d_rstart.as<uint32_t>(), h->d_rq.as<uint32_t>()); break;
    case 2: hipLaunchKernelGGL(k_root_quartiles<2>, gq, b, 0, h->stream, tmp, h->d_rstart.as<uint32_t>(), h->d_rq.as<uint32_t>()); break;
    case 3: hipLaunchKernelGGL(k_root_quartiles<3>, gq, b, 0, h->stream, tmp, h->d_rstart.as<uint32_t>(), h->d_rq.as<uint32_t>()); break;
    default: hipLaunchKernelGGL(k_root_quartiles<4>, gq, b, 0, h->stream, tmp, h->d_rstart.as<uint32_t>(), h->d_rq.as<uint32_t>()); break;
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) { h->im.rdir = nullptr; h->im.rstart = nullptr; h->rstart_ok = false; return; }
    h->rq_ok = true;
    h->im.rq = h->d_rq.as<uint32_t>();
}

// (Re)derives the flat form of the handle's current containers, points the image at everything and tunes the launch.
// The derived arrays are built aside and swapped in only once complete: a failure leaves the image as it was.
static int bind_image(bft_gpu* h, uint32_t nb_genomes) {
    CK(wait_foreign_stream(h));
    DevBuf ccx, f18, fent;
    uint64_t n_f18 = 0, n_fent = 0;
    CK(bft_flatten_gpu(h->d_ccs.as<BftCC>(), h->idx_sizes[2] / sizeof(BftCC), h->d_f2w.as<uint64_t>(), h->d_clus.as<uint64_t>(), h->d_child.as<uint64_t>(),
                       h->opt_flat_min, h->stream, ccx, f18, fent, n_f18, n_fent));
    h->d_ccx.swap(ccx);
    h->d_f18.swap(f18);
    h->d_fent.swap(fent);
    h->n_f18 = n_f18;
    h->n_fent = n_fent;
    point_image(h, nb_genomes);
    derive_root_direct(h);
    derive_kmer_hash(h);
    sync_walk_kh(h);
    derive_node_hash(h);
    default_launch_shape(h);
    return 0;
}

// The genome ids of the dictionary stay resident in the narrowest width that holds every id (a pan-genome of 100 genomes: one byte
// instead of four -- 484 MB -> 121 MB on config 3, 8 of the image's 49 bytes per k-mer); the build and the merge work on 32-bit ids.
template <class T>
__global__ void k_narrow_ids(const uint32_t* __restrict__ in, uint64_t n, T* __restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (T)in[i];
}
template <class T>
__global__ void k_widen_ids(const T* __restrict__ in, uint64_t n, uint32_t* __restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (uint32_t)in[i];
}
static uint32_t id_width(uint32_t max_gid) { return max_gid < 256u ? 1u : (max_gid < 65536u ? 2u : 4u); }
// ids32 (n u32) -> out in width w; w == 4: the buffers are swapped
static int narrow_ids(DevBuf& ids32, uint64_t n, uint32_t w, hipStream_t s, DevBuf& out) {
    if (w == 4) { out.swap(ids32); return 0; }
    CK(out.alloc(n * w));
    const dim3 grid(grid_for((n + 255) / 256)), block(256);
    if (n) {
        if (w == 1) hipLaunchKernelGGL(k_narrow_ids<uint8_t>, grid, block, 0, s, ids32.as<uint32_t>(), n, out.as<uint8_t>());
        else hipLaunchKernelGGL(k_narrow_ids<uint16_t>, grid, block, 0, s, ids32.as<uint32_t>(), n, out.as<uint16_t>());
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(s));
    }
    return 0;
}
// the resident dictionary as u32 (a transient of the merge); w == 4: `src` itself
static int widen_ids(const DevBuf& src, uint64_t n, uint32_t w, hipStream_t s, DevBuf& tmp, const uint32_t** out) {
    if (w == 4) { *out = src.as<uint32_t>(); return 0; }
    CK(tmp.alloc(n * 4));
    const dim3 grid(grid_for((n + 255) / 256)), block(256);
    if (n) {
        if (w == 1) hipLaunchKernelGGL(k_widen_ids<uint8_t>, grid, block, 0, s, src.as<uint8_t>(), n, tmp.as<uint32_t>());
        else hipLaunchKernelGGL(k_widen_ids<uint16_t>, grid, block, 0, s, src.as<uint16_t>(), n, tmp.as<uint32_t>());
        HIPCK(hipGetLastError());
    }
    *out = tmp.as<uint32_t>();
    return 0;
}

// The colour-set dictionary lives in HBM; only bft_gpu_colorset and bft_gpu_write_bft need it on the host.
static int host_colorsets(bft_gpu* h) {
    if (h->cs_on_host) return 0;
    h->cs_off.assign(h->n_sets + 1, 0);
    h->cs_ids.assign(h->n_ids, 0);
    HIPCK(hipMemcpy(h->cs_off.data(), h->d_cs_off.p, (h->n_sets + 1) * 4, hipMemcpyDeviceToHost));
    if (h->n_ids) {
        if (h->cs_w == 4) HIPCK(hipMemcpy(h->cs_ids.data(), h->d_cs_ids.p, h->n_ids * 4, hipMemcpyDeviceToHost));
        else {
            std::vector<uint8_t> raw(h->n_ids * h->cs_w);
            HIPCK(hipMemcpy(raw.data(), h->d_cs_ids.p, raw.size(), hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < h->n_ids; i++) h->cs_ids[i] = h->cs_w == 1 ? raw[i] : ((const uint16_t*)raw.data())[i];
        }
    }
    h->cs_on_host = true;
    return 0;
}

// One-word keys, sorted, with their genome ids (GT wide): flags on the fly, one 64-bit scan, scatter into the genome-id lists, the
// distinct-k-mer table and the segment offsets (bft_kernels_build.h).
template <class GT>
static int dedupe_w1(bft_gpu* h, const uint64_t* sk, const GT* sg, uint64_t total, DevBuf& tk, DevBuf& seg_off, DevBuf& npg, uint64_t& nk,
                     uint64_t& np) {
    DevBuf tmp, pos;
    CK(pos.alloc(total * 8));
    const BftPairFlags2<GT> pf{sk, sg};
    CK((bft_scan::exclusive_sum<uint64_t>(pf, pos.as<uint64_t>(), total, h->stream, tmp)));
    uint64_t last_pos = 0, last_k[2] = {0, 0};
    GT last_g[2] = {0, 0};
    HIPCK(hipMemcpyAsync(&last_pos, pos.as<uint64_t>() + total - 1, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipMemcpyAsync(&last_k[1], sk + total - 1, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipMemcpyAsync(&last_g[1], sg + total - 1, sizeof(GT), hipMemcpyDeviceToHost, h->stream));
    if (total > 1) {
        HIPCK(hipMemcpyAsync(&last_k[0], sk + total - 2, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipMemcpyAsync(&last_g[0], sg + total - 2, sizeof(GT), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCK(hipStreamSynchronize(h->stream));
    const bool last_head = total == 1 || last_k[1] != last_k[0], last_keep = last_head || last_g[1] != last_g[0];
    nk = (last_pos >> 32) + (last_head ? 1 : 0);
    np = (last_pos & 0xFFFFFFFFull) + (last_keep ? 1 : 0);
    CK(tk.alloc(nk * 8));
    CK(seg_off.alloc((nk + 1) * 4));
    CK(npg.alloc(np * 4));
    hipLaunchKernelGGL(k_scatter_2<GT>, dim3(grid_for((total + 255) / 256)), dim3(256), 0, h->stream, sk, sg, total, pos.as<uint64_t>(), npg.as<uint32_t>(),
                       tk.as<uint64_t>(), seg_off.as<uint32_t>());
    const uint32_t np32 = (uint32_t)np;
    HIPCK(hipMemcpyAsync(seg_off.as<uint32_t>() + nk, &np32, 4, hipMemcpyHostToDevice, h->stream));
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(h->stream));
    return 0;
}
// Stable sort of ordered one-word keys with their ids narrowed to GT (values through a transform iterator over the log), then dedupe_w1.
template <class GT>
static int sort_dedupe_w1_narrow(bft_gpu* h, const uint64_t* src_k, const uint32_t* src_g, uint64_t total, DevBuf& tk, DevBuf& seg_off, DevBuf& npg,
                                 uint64_t& nk, uint64_t& np) {
    DevBuf sk, sg;
    CK(sk.alloc(total * 8));
    CK(sg.alloc(total * sizeof(GT)));
    CK((bft_rs::sort_in<uint64_t, GT>(BftPairIn<GT>{src_k, src_g}, total, sk.as<uint64_t>(), sg.as<GT>(), 0u, (unsigned)(2 * h->k), h->stream)));
    HIPCK(hipGetLastError());
    return dedupe_w1<GT>(h, sk.as<uint64_t>(), sg.as<GT>(), total, tk, seg_off, npg, nk, np);
}

// The same front end for ordered one-word keys whose composite does not fit 63 bits (k = 31 beyond a couple of genomes, k = 27 beyond
// 512): a stable key + value sort on the 18 root-prefix bits -- the ids travel beside the keys, narrowed to VT --, then the buckets
// (bft_front.hip), where the bits a bucket's k-mers share make room for the id.  done = false: a bucket is too large, nothing was built.
template <class VT>
static int split_dedupe_w1(bft_gpu* h, const uint64_t* src_k, const uint32_t* src_g, uint64_t total, int gb, DevBuf& tk, DevBuf& seg_off, DevBuf& npg,
                           uint64_t& nk, uint64_t& np, bool& done) {
    done = false;
    const unsigned top = (unsigned)std::min(18, 2 * h->k), rest = (unsigned)(2 * h->k) - top;
    DevBuf sk, sv, boff, maxb;
    CK(sk.alloc(total * 8));
    CK(sv.alloc(total * sizeof(VT)));
    CK(boff.alloc(((1u << top) + 1) * 4));
    CK(maxb.alloc_zero(4, h->stream));
    {
        DevBuf tk2, tv2, scratch;
        const bft_rs::Plan spl = bft_rs::make_plan(rest, (unsigned)(2 * h->k));
        if (spl.P > 1) {
            CK(tk2.alloc(total * 8));
            CK(tv2.alloc(total * sizeof(VT)));
        }
        const uint32_t* dbase = nullptr;
        CK((bft_rs::sort<uint64_t, VT, BftPairIn<VT>>(BftPairIn<VT>{src_k, src_g}, total, sk.as<uint64_t>(), sv.as<VT>(), tk2.as<uint64_t>(), tv2.as<VT>(), rest, (unsigned)(2 * h->k),
                                                      h->stream, scratch, &dbase)));
        hipLaunchKernelGGL(k_msd_bounds, dim3(((1u << top) + 1 + 255) / 256), dim3(256), 0, h->stream, sk.as<uint64_t>(), total, (uint32_t)rest, 1u << top, boff.as<uint32_t>(), dbase,
                           spl.P ? spl.nbits[spl.P - 1] : 0u, top);
        hipLaunchKernelGGL(k_msd_max, dim3(((1u << top) + 255) / 256), dim3(256), 0, h->stream, boff.as<uint32_t>(), 1u << top, maxb.as<uint32_t>());
    }
    bft_stage("split (root-prefix, 2 x 9 bits, pairs)", (double)total * (12 + 12 + 2 * (8 + sizeof(VT)) + (8 + sizeof(VT))), h->stream);
    uint32_t mx = 0;
    CK(bft_front_buckets(sk.as<uint64_t>(), total, boff.as<uint32_t>(), 1u << top, (uint32_t)gb, (uint32_t)gb + rest, h->stream, tk, seg_off, npg, nk, np, maxb.as<uint32_t>(), &mx, &done,
                         &h->front_redone, sv.p, (uint32_t)sizeof(VT)));
    h->msd_max_bucket = mx;
    return 0;
}

// "build_stages": the marks of one bft_gpu_build (bft_stage) become the handle's stage table when the build returns, however it returns
struct StageScope {
    bft_gpu* h;
    explicit StageScope(bft_gpu* hh) : h(hh) {
        t_stages_on = h->opt_build_stages;
        t_stage_marks.clear();
        if (t_stages_on) bft_stage("start", 0, h->stream);
    }
    ~StageScope() {
        if (!t_stages_on) return;
        t_stages_on = false;
        h->stages.clear();
        (void)hipStreamSynchronize(h->stream);
        if (h->stream2) (void)hipStreamSynchronize(h->stream2);
        for (size_t i = 0; i < t_stage_marks.size(); i++) {
            float ms = 0;
            // a mark on the second stream ("+name": work that runs beside the main chain) is timed from the build's start
            const bool side = !t_stage_marks[i].name.empty() && t_stage_marks[i].name[0] == '+';
            size_t prev = 0;
            if (!side) for (size_t j = i; j-- > 0;) if (t_stage_marks[j].name[0] != '+') { prev = j; break; }
            if (i > 0 && hipEventSynchronize(t_stage_marks[i].ev) == hipSuccess &&
                hipEventElapsedTime(&ms, t_stage_marks[prev].ev, t_stage_marks[i].ev) == hipSuccess)
                h->stages.push_back({t_stage_marks[i].name, (double)ms, t_stage_marks[i].bytes});
            (void)hipGetLastError();
        }
        for (auto& m : t_stage_marks) t_stage_pool.push_back(m.ev);
        t_stage_marks.clear();
    }
};

extern "C" int bft_gpu_build(bft_gpu* h) {
    if (!h) return fail(BFT_GPU_E_ARG, "NULL handle");
    ENTER(h);
    if (h->built && h->log_n == 0) return BFT_GPU_OK;
    CK(wait_foreign_stream(h));  // batches still being packed into the log on a caller's stream (bft_gpu_insert_kmers_dev_async)
    if (h->log_comp && (!h->log_g_sorted || h->opt_no_composite)) CK(log_decompose(h));  // (only the composite path below takes a log of composites)
    if (h->built) CK(ensure_table(h));  // ("compact_table": the merge reads the index's sorted table)
    const int W = h->W;
    const uint64_t total = h->log_n;  // the run: what was inserted since the last build
    double t0 = now_ms();
    bft_trace_mark(nullptr);
    StageScope stage_scope(h);

    DevBuf tk, seg_off, npg;
    uint64_t nk = 0, np = 0;
    h->front_redone = 0;
    h->msd_max_bucket = 0;
    if (total > 0) {
        // 1. the pairs to sort: the insertion log
        DevBuf ck, cg, sk, sg;
        const uint64_t* src_k = h->log_k.as<uint64_t>();
        const uint32_t* src_g = h->log_g.as<uint32_t>();
        const uint64_t src_stride = h->log_cap;
        const int gb = h->log_comp ? (int)h->log_gb : bits_for(h->max_gid_seen);  // (a log of composites: their id field as logged)
        // (composites of up to 63 bits: a limit from the time of the library sort -- rocPRIM's radix sort mis-sorts the bit range [2, 64), found by
        // test_any_k_against_ground_truth[31-0] -- kept: the paths behind it are the tested ones for such k)
        if (W == 1 && h->log_g_sorted && 2 * h->k + gb <= 63 && !h->opt_no_composite) {
            // 2c + 3c. composite path (bft_kernels_build.h): sort (T << gb | genome) on the T bits, flags on the fly, one scan
            DevBuf cs, tmp, pos;
            CK(cs.alloc(total * 8));
            CK(pos.alloc(total * 8));
            const BftCompose comp{src_k, h->log_comp ? nullptr : src_g, (uint32_t)gb};
            // MSD first: a stable sort on the top 18 bits of T (the rotated root prefix: 2^18 buckets of ~10^3 composites on a
            // pan-genome index), then every bucket on its own on the remaining bits, in LDS, with the duplicates flagged and counted
            // on the way (bft_front.hip) -- one read and one write of the array instead of the four or five full passes those
            // bits cost a device-wide LSD sort, and no flag scan over the pairs.  Skewed inputs (a bucket beyond what one workgroup
            // sorts) keep the one-sort path.
            const unsigned top = (unsigned)std::min(18, 2 * h->k), rest = (unsigned)(2 * h->k) - top;
            const bool msd = h->opt_msd && (total >= (1u << 20) || h->opt_msd == 2);
            bool done = false;
            h->msd_max_bucket = 0;
            if (msd) {
                DevBuf boff, maxb;
                CK(boff.alloc(((1u << top) + 1) * 4));
                CK(maxb.alloc_zero(4, h->stream));
                // (the library's own partition, bft_sort.h: one histogram kernel, a first pass that needs no look-back, a second in 64 chains;
                // the composites are formed from the log by the kernels that read it; `pos` carries them between the two passes)
                const uint32_t* dbase = nullptr;  // (where the last pass's digit -- the top bits of the prefix -- changes: in `tmp`)
                CK((bft_rs::sort<uint64_t, bft_rs::NoVal, BftCompose>(comp, total, cs.as<uint64_t>(), (bft_rs::NoVal*)nullptr, pos.as<uint64_t>(), (bft_rs::NoVal*)nullptr,
                                                                      (unsigned)gb + rest, (unsigned)(gb + 2 * h->k), h->stream, tmp, &dbase)));
                const bft_rs::Plan spl = bft_rs::make_plan((unsigned)gb + rest, (unsigned)(gb + 2 * h->k));
                hipLaunchKernelGGL(k_msd_bounds, dim3(((1u << top) + 1 + 255) / 256), dim3(256), 0, h->stream, cs.as<uint64_t>(), total, (uint32_t)(gb + rest), 1u << top,
                                   boff.as<uint32_t>(), dbase, spl.P ? spl.nbits[spl.P - 1] : 0u, top);
                hipLaunchKernelGGL(k_msd_max, dim3(((1u << top) + 255) / 256), dim3(256), 0, h->stream, boff.as<uint32_t>(), 1u << top, maxb.as<uint32_t>());
                // (histogram: the log, 12 B per pair; pass 1: the log in, composites out; pass 2: composites in and out)
                bft_stage("split (root-prefix, 2 x 9 bits)", (double)total * ((h->log_comp ? 8 : 12) * 2 + 8 + 8 + 8), h->stream);
                uint32_t mx = 0;
                CK(bft_front_buckets(cs.as<uint64_t>(), total, boff.as<uint32_t>(), 1u << top, (uint32_t)gb, (uint32_t)gb + rest, h->stream, tk, seg_off, npg, nk, np, maxb.as<uint32_t>(), &mx,
                                     &done, &h->front_redone));
                h->msd_max_bucket = mx;
                if (done) pos.release();
            }
            if (!done) {
            if (!pos.p) CK(pos.alloc(total * 8));
            CK((bft_rs::sort<uint64_t, bft_rs::NoVal, BftCompose>(comp, total, cs.as<uint64_t>(), (bft_rs::NoVal*)nullptr, pos.as<uint64_t>(), (bft_rs::NoVal*)nullptr, (unsigned)gb,
                                                                  (unsigned)(gb + 2 * h->k), h->stream, tmp)));
            const BftPairFlags pf{cs.as<uint64_t>(), (uint32_t)gb};
            CK((bft_scan::exclusive_sum<uint64_t>(pf, pos.as<uint64_t>(), total, h->stream, tmp)));
            uint64_t last_pos = 0, last_c[2] = {0, 0};
            HIPCK(hipMemcpyAsync(&last_pos, pos.as<uint64_t>() + total - 1, 8, hipMemcpyDeviceToHost, h->stream));
            HIPCK(hipMemcpyAsync(&last_c[1], cs.as<uint64_t>() + total - 1, 8, hipMemcpyDeviceToHost, h->stream));
            if (total > 1) HIPCK(hipMemcpyAsync(&last_c[0], cs.as<uint64_t>() + total - 2, 8, hipMemcpyDeviceToHost, h->stream));
            HIPCK(hipStreamSynchronize(h->stream));
            const bool last_head = total == 1 || (last_c[1] >> gb) != (last_c[0] >> gb), last_keep = total == 1 || last_c[1] != last_c[0];
            nk = (last_pos >> 32) + (last_head ? 1 : 0);
            np = (last_pos & 0xFFFFFFFFull) + (last_keep ? 1 : 0);
            CK(tk.alloc(nk * 8));
            CK(seg_off.alloc((nk + 1) * 4));
            CK(npg.alloc(np * 4));
            hipLaunchKernelGGL(k_scatter_c, dim3(grid_for((total + 255) / 256)), dim3(256), 0, h->stream, cs.as<uint64_t>(), (uint32_t)gb, total, pos.as<uint64_t>(),
                               npg.as<uint32_t>(), tk.as<uint64_t>(), seg_off.as<uint32_t>());
            const uint32_t np32 = (uint32_t)np;
            HIPCK(hipMemcpyAsync(seg_off.as<uint32_t>() + nk, &np32, 4, hipMemcpyHostToDevice, h->stream));
            HIPCK(hipGetLastError());
            HIPCK(hipStreamSynchronize(h->stream));
            }
            ck.release();
            cg.release();
        } else if (W == 1 && h->log_g_sorted && !h->opt_no_composite && (h->max_gid_seen < 65536 || (2 * h->k - std::min(18, 2 * h->k)) + gb <= 64)) {
            // 2s. ordered one-word keys whose composite does not fit (k = 31 with more than a few genomes): the root-prefix split with
            // the ids beside the keys, then the buckets -- where the composite does fit, the bucket's number being implied
            bool done = false;
            h->msd_max_bucket = 0;
            // (not at k = 32: a limit from the time of the library sort, which mis-sorted the bit range [46, 64) the split would sort -- found by
            // tools/stress_parity.py, k = 32 with build_msd = 2; bft_rs sorts that very range for two-word keys, but k = 32 stays on its tested path)
            if (h->opt_msd && (total >= (1u << 20) || h->opt_msd == 2) && (2 * h->k - std::min(18, 2 * h->k)) + gb <= 64 && 2 * h->k < 64) {
                if (h->max_gid_seen < 256) CK(split_dedupe_w1<uint8_t>(h, src_k, src_g, total, gb, tk, seg_off, npg, nk, np, done));
                else if (h->max_gid_seen < 65536) CK(split_dedupe_w1<uint16_t>(h, src_k, src_g, total, gb, tk, seg_off, npg, nk, np, done));
                else CK(split_dedupe_w1<uint32_t>(h, src_k, src_g, total, gb, tk, seg_off, npg, nk, np, done));
            }
            // 2n + 3'. ... or one key + value sort over every bit with the ids narrowed to one or two bytes (the values are a third of
            // the sort's traffic at four), flags on the fly, one scan
            if (!done) {
                if (h->max_gid_seen < 256) CK(sort_dedupe_w1_narrow<uint8_t>(h, src_k, src_g, total, tk, seg_off, npg, nk, np));
                else if (h->max_gid_seen < 65536) CK(sort_dedupe_w1_narrow<uint16_t>(h, src_k, src_g, total, tk, seg_off, npg, nk, np));
                else CK(sort_dedupe_w1_narrow<uint32_t>(h, src_k, src_g, total, tk, seg_off, npg, nk, np));
            }
            ck.release();
            cg.release();
        } else {
        bool done2 = false;
        if (W == 2 && h->log_g_sorted && h->opt_msd && (total >= (1u << 20) || h->opt_msd == 2)) {
            // 2w. two-word keys (33 <= k <= 64) whose ids arrived ascending: the root-prefix split on the top 18 bits of the T-form -- moving (top 64
            // bits, the bits below, id) --, then every bucket on its own: grouped by a hash of its key bits, only the distinct k-mers ordered (bft_front.hip)
            DevBuf hk, items, hk2, items2, boff, maxb, scratch;
            CK(hk.alloc(total * 8));
            CK(items.alloc(total * sizeof(BftSplit2Val)));
            CK(hk2.alloc(total * 8));
            CK(items2.alloc(total * sizeof(BftSplit2Val)));
            CK(boff.alloc(((1u << 18) + 1) * 4));
            CK(maxb.alloc_zero(4, h->stream));
            const BftSplit2In in2{src_k, src_k + src_stride, src_g, (uint32_t)(2 * h->k - 64)};
            const uint32_t* dbase = nullptr;
            CK((bft_rs::sort<uint64_t, BftSplit2Val, BftSplit2In>(in2, total, hk.as<uint64_t>(), items.as<BftSplit2Val>(), hk2.as<uint64_t>(), items2.as<BftSplit2Val>(), 46u, 64u, h->stream,
                                                                  scratch, &dbase)));
            hipLaunchKernelGGL(k_msd_bounds, dim3(((1u << 18) + 1 + 255) / 256), dim3(256), 0, h->stream, hk.as<uint64_t>(), total, 46u, 1u << 18, boff.as<uint32_t>(), dbase, 9u, 18u);
            hipLaunchKernelGGL(k_msd_max, dim3(((1u << 18) + 255) / 256), dim3(256), 0, h->stream, boff.as<uint32_t>(), 1u << 18, maxb.as<uint32_t>());
            bft_stage("split (root-prefix, 2 x 9 bits, two-word keys)", (double)total * (20 + 8 + 20 * 4), h->stream);
            hk2.release();
            items2.release();
            uint32_t mx = 0;
            CK(bft_front2_buckets(hk.as<uint64_t>(), items.p, total, boff.as<uint32_t>(), 1u << 18, h->k, h->stream, tk, seg_off, npg, nk, np, maxb.as<uint32_t>(), &mx, &done2, &h->front_redone, h->max_gid_seen));
            h->msd_max_bucket = mx;
        }
        if (!done2) {
        // 2. sort by (T, genome)
        CK(sk.alloc(total * W * 8));
        CK(sg.alloc(total * 4));
        CK(sort_pairs(h, src_k, src_stride, src_g, total, sk.as<uint64_t>(), total, sg.as<uint32_t>(), h->log_g_sorted, true));
        ck.release();
        cg.release();
        // (the insertion log stays until the new image is committed below: a failed build loses nothing)
        if (W == 1) {
            // 3'. one-word keys: flags on the fly, one 64-bit scan (as on the composite path)
            CK(dedupe_w1<uint32_t>(h, sk.as<uint64_t>(), sg.as<uint32_t>(), total, tk, seg_off, npg, nk, np));
        } else {
        // 3. flags, scans, compaction
        DevBuf head, keep, posK, posP, tmp;
        CK(head.alloc(total * 4));
        CK(keep.alloc(total * 4));
        CK(posK.alloc(total * 4));
        CK(posP.alloc(total * 4));
        const int grid = grid_for((total + 255) / 256);
        hipLaunchKernelGGL(k_flags, dim3(grid), dim3(256), 0, h->stream, sk.as<uint64_t>(), total, W, sg.as<uint32_t>(), total, head.as<uint32_t>(), keep.as<uint32_t>());
        CK(bft_scan::exclusive_sum_ptr<uint32_t>(head.as<uint32_t>(), posK.as<uint32_t>(), total, h->stream, tmp));
        DevBuf tmp2;  // (a scratch of its own: the first scan may still be running on the stream)
        CK(bft_scan::exclusive_sum_ptr<uint32_t>(keep.as<uint32_t>(), posP.as<uint32_t>(), total, h->stream, tmp2));
        uint32_t lastK[2], lastP[2];
        HIPCK(hipMemcpyAsync(&lastK[0], posK.as<uint32_t>() + total - 1, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipMemcpyAsync(&lastK[1], head.as<uint32_t>() + total - 1, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipMemcpyAsync(&lastP[0], posP.as<uint32_t>() + total - 1, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipMemcpyAsync(&lastP[1], keep.as<uint32_t>() + total - 1, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
        nk = (uint64_t)lastK[0] + lastK[1];
        np = (uint64_t)lastP[0] + lastP[1];
        CK(tk.alloc(nk * W * 8));
        CK(seg_off.alloc((nk + 1) * 4));
        CK(npg.alloc(np * 4));
        hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(256), 0, h->stream, sk.as<uint64_t>(), total, W, sg.as<uint32_t>(), total,
                           head.as<uint32_t>(), keep.as<uint32_t>(), posK.as<uint32_t>(), posP.as<uint32_t>(), npg.as<uint32_t>(), tk.as<uint64_t>(),
                           seg_off.as<uint32_t>());
        const uint32_t np32 = (uint32_t)np;
        HIPCK(hipMemcpyAsync(seg_off.as<uint32_t>() + nk, &np32, 4, hipMemcpyHostToDevice, h->stream));
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(h->stream));
        }
        }
        }
    }
    bft_trace_mark("sort + dedupe done");
    bft_stage("sort + dedupe (rest)", 0, h->stream);
    double t1 = now_ms();

    // 4. colour sets: signature sort + exact run detection + verification, all on the GPU
    uint64_t n_sets = 0, n_ids = 0;
    if (nk == 0) {
        CK(seg_off.alloc_zero(4, h->stream));
        CK(npg.alloc(4));
    }
    DevBuf n_tcol, n_cs_off, n_cs_ids;  // built aside, like every array of the new image
    KhFill khf;  // the k-mer hash, filled beside the container assembly
    if (!h->stream2 && nk > 0) {  // (the second stream, made here: the helper thread below and the interning's tail both use it)
        int lo = 0, hi = 0;  // (numerically larger = lower priority)
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; (void)hipGetLastError(); }
        if (hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, lo) != hipSuccess) { h->stream2 = nullptr; (void)hipGetLastError(); }
    }
    if (!(h->built && h->n_kmers > 0) && nk > 0) {  // (a merge changes the k-mers: kh_start does everything then)
        // (the table's sort by home line started HERE, beside the colour-set interning, was measured: the interning's persistent grids and the
        // sort starve each other -- 4.1 -> 10.5 ms for the interning, 20 -> 25.8 ms for the build; it starts behind the assembly's table passes)
        kh_prepare_async(h, nk, nk, khf);
    }
    // (the interning's tail -- dictionary copy + verification of every list -- goes to the second stream, idle until the k-mer hash build starts
    // behind the first level's table passes; its verdict is collected before the commit.  A merge reads the dictionary at once: not deferred.)
    const bool merging = h->built && h->n_kmers > 0 && nk > 0;
    BftInternTail tail;
    tail.side = merging ? nullptr : h->stream2;
    tail.narrow_w = id_width(h->max_gid_seen);  // (the tail also leaves the dictionary in the width the image keeps)
    CK(bft_intern_colors_gpu(seg_off.as<uint32_t>(), npg.as<uint32_t>(), nk, np, h->stream, n_tcol, n_cs_off, n_cs_ids, n_sets, n_ids, 0, &tail));
    bft_trace_mark("colour sets interned");
    bft_stage("colour sets (rest)", 0, h->stream);
    if (!tail.pending) {
        seg_off.release();
        npg.release();
    }
    // 4b. an index exists already: the run is merged into it (bft_merge.hip) -- k-mers by position, colour sets by union
    uint64_t total_pairs = np;
    if (h->built && h->n_kmers > 0 && nk > 0) {
        DevBuf wide;
        const uint32_t* old_ids = nullptr;
        CK(widen_ids(h->d_cs_ids, h->n_ids, h->cs_w, h->stream, wide, &old_ids));
        const BftRun old_run{h->d_tk.as<uint64_t>(), h->d_tcol.as<uint32_t>(), h->n_kmers, h->d_cs_off.as<uint32_t>(), old_ids, h->n_sets};
        const BftRun new_run{tk.as<uint64_t>(), n_tcol.as<uint32_t>(), nk, n_cs_off.as<uint32_t>(), n_cs_ids.as<uint32_t>(), n_sets};
        BftRunOut mo;
        CK(bft_merge_runs(W, old_run, new_run, h->stream, mo));
        tk.swap(mo.tk);
        n_tcol.swap(mo.tcol);
        n_cs_off.swap(mo.cs_off);
        n_cs_ids.swap(mo.cs_ids);
        nk = mo.n;
        n_sets = mo.n_sets;
        n_ids = mo.n_ids;
        CK(bft_count_pairs(n_tcol.as<uint32_t>(), nk, n_cs_off.as<uint32_t>(), h->stream, &total_pairs));
    }
    np = total_pairs;
    struct KhStart {  // (from here on tk, n_tcol and the number of colour sets are final)
        bft_gpu* h; const uint64_t* tk; const uint32_t* tcol; uint64_t nk, n_sets; KhFill* f;
        static void run(void* c, hipStream_t s) { KhStart* k = (KhStart*)c; kh_start(k->h, k->tk, k->tcol, k->nk, k->n_sets, *k->f, s, nullptr); }
    } khs{h, tk.as<uint64_t>(), n_tcol.as<uint32_t>(), nk, n_sets, &khf};
    // (started HERE instead -- beside the root's table passes -- the sort-based table build was measured too: 17.2-18.4 ms against 17.0)
    // (and earlier still -- behind the root's prefix scans, or before them: 15.7 ms each way, round 5: the build is bound by the device's total work,
    // not by either stream's chain)
    const BftAssembleHook hook{tk.p ? &KhStart::run : nullptr, &khs};
    bft_trace_mark("merge / bookkeeping");
    bft_stage("merge into the index", 0, h->stream);
    double t2 = now_ms();

    // 5. containers, level by level, on the GPU
    if (!tk.p) CK(tk.alloc(8));
    BftDeviceIndex idx;
    CK(bft_assemble_gpu(tk.as<uint64_t>(), nk, h->k, h->d_hashmod.as<uint32_t>(), h->stream, idx, &hook));
    bft_trace_mark("containers assembled");
    bft_stage("containers: concatenation", 0, h->stream);
    double t3 = now_ms();
    DevBuf n_ccx, n_f18buf, n_fentbuf;
    uint64_t n_f18 = 0, n_fent = 0;
    CK(bft_flatten_gpu(idx.ccs.as<BftCC>(), idx.n_ccs, idx.f2w.as<uint64_t>(), idx.clus.as<uint64_t>(), idx.child.as<uint64_t>(), h->opt_flat_min, h->stream,
                       n_ccx, n_f18buf, n_fentbuf, n_f18, n_fent));
    bft_trace_mark("flat forms");
    bft_stage("flat forms of the big CCs", (double)(n_f18 + n_fent) * 8, h->stream);
    bool kh_redo = false;
    {   // the interning's deferred tail: done long ago; two lists with one signature (never seen outside the test hook) -> the exact interning
        uint32_t collisions = 0;
        CK(tail.wait(&collisions));
        if (collisions) {
            double ms_ = 0;
            (void)kh_finish(h, khf, &ms_);  // (the table under construction holds the wrong colour sets, and reads the arrays replaced below)
            khf.buf.release();
            kh_redo = true;
            CK(bft_intern_colors_gpu(seg_off.as<uint32_t>(), npg.as<uint32_t>(), nk, np, h->stream, n_tcol, n_cs_off, n_cs_ids, n_sets, n_ids, 0, nullptr, true));
        }
        seg_off.release();
        npg.release();
    }
    const uint32_t new_cs_w = id_width(h->max_gid_seen);
    DevBuf n_cs_ids_w;
    bool narrowed_here = false;
    if (!kh_redo && tail.narrow.p && tail.narrow_w == new_cs_w) {  // (done by the interning's tail on the side stream)
        n_cs_ids_w.swap(tail.narrow);
        n_cs_ids.release();
    } else {
        CK(narrow_ids(n_cs_ids, n_ids, new_cs_w, h->stream, n_cs_ids_w));
        narrowed_here = new_cs_w < 4;
    }
    CK(wait_foreign_stream(h));  // queries a caller still has in flight on its own stream read the arrays released below
    bft_trace_mark("ids narrowed, foreign stream waited");
    bft_stage("dictionary ids narrowed", narrowed_here ? (double)n_ids * (4 + new_cs_w) : 0.0, h->stream);
    double kh_ms = 0;
    const bool kh_ok = !kh_redo && kh_finish(h, khf, &kh_ms);
    bft_trace_mark("k-mer hash fill waited");
    bft_stage("wait for the k-mer hash build", 0, h->stream);
    if (h->inject_build_failure) {
        h->inject_build_failure = false;
        return fail(BFT_GPU_E_LIMIT, "injected build failure (test hook)");
    }

    // ---- commit: nothing above touched the handle; from here on nothing can fail before the image is whole ----
    h->d_tcol.swap(n_tcol);
    h->d_cs_off.swap(n_cs_off);
    h->d_cs_ids.swap(n_cs_ids_w);
    h->cs_w = new_cs_w;
    h->n_sets = n_sets;
    h->n_ids = n_ids;
    h->cs_on_host = false;
    h->cs_off.clear();
    h->cs_ids.clear();
    h->d_ccx.swap(n_ccx);
    h->d_f18.swap(n_f18buf);
    h->d_fent.swap(n_fentbuf);
    h->n_f18 = n_f18;
    h->n_fent = n_fent;
    h->log_k.release();
    h->log_g.release();
    h->log_cap = 0;
    h->d_nodes.swap(idx.nodes);
    h->d_bfT.swap(idx.bfT);
    h->d_ccs.swap(idx.ccs);
    h->d_f2w.swap(idx.f2w);
    h->d_clus.swap(idx.clus);
    h->d_child.swap(idx.child);
    h->d_uck.swap(idx.uck);
    h->d_ucrow.swap(idx.ucrow);
    h->d_tk.swap(tk);
    h->n_pairs = np;
    h->log_n = 0;
    h->lb_end.clear();
    h->lb_gid.clear();
    h->log_comp = false;
    h->log_g_sorted = true;
    h->n_kmers = nk;
    h->idx_sizes[0] = idx.n_nodes * sizeof(BftNode); h->idx_sizes[1] = idx.n_bf8 * 8; h->idx_sizes[2] = idx.n_ccs * sizeof(BftCC);
    h->idx_sizes[3] = idx.n_f2w * 8; h->idx_sizes[4] = idx.n_clus * 8; h->idx_sizes[5] = idx.n_child * 8;
    h->idx_sizes[6] = idx.n_uc * (uint64_t)W * 8; h->idx_sizes[7] = idx.n_uc * 4; h->idx_sizes[8] = nk * (uint64_t)W * 8;
    h->root_ncc = (uint32_t)idx.root_ncc;
    point_image(h, std::max<uint32_t>((uint32_t)h->genomes.size(), h->any_insert ? h->max_gid_seen + 1 : 0));
    BftImage& im = h->im;

    uint64_t* I = h->info;
    I[0] = h->k;
    I[1] = nk;
    I[2] = idx.n_nodes;
    I[3] = idx.n_ccs;
    I[4] = idx.n_uc;
    I[5] = idx.n_child_nodes;
    I[6] = idx.n_prefixes;
    I[7] = idx.n_ccs_s4;
    I[8] = idx.max_ccs_per_node;
    I[9] = np;
    I[10] = n_sets;
    I[11] = im.nb_genomes;
    I[12] = image_bytes(h);
    I[13] = idx.root_ncc;
    I[14] = idx.root_uc;
    h->root_ncc = (uint32_t)idx.root_ncc;
    h->build_ms[0] = t1 - t0;
    h->build_ms[1] = t2 - t1;
    h->build_ms[2] = t3 - t2;
    h->build_ms[3] = (double)h->front_redone;
    h->built = true;
    ensure_claim_counters(h);
    bft_trace_mark("committed (buffers released)");
    derive_root_direct(h);
    if (kh_ok) kh_adopt(h, khf, kh_ms);  // built during the assembly
    else kh_drop(h);
    if (kh_redo) derive_kmer_hash(h);  // (after the exact interning: from the committed table and colour sets)
    sync_walk_kh(h);
    bft_trace_mark("root tables");
    bft_stage("root tables", 0, h->stream);
    derive_node_hash(h);
    default_launch_shape(h);
    I[12] = image_bytes(h);
    h->table_dropped = false;
    drop_table(h);  // ("compact_table")
    bft_trace_mark("launch shape; done");
    bft_stage("node hash, launch shape, compact table", 0, h->stream);
    if (bft_trace_on()) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        fprintf(stderr, "[bft_gpu build] cache of released blocks: %zu blocks, %.1f MB; this process so far: %llu hipMalloc (%.2f ms), %llu hipFree on release (%.2f ms)\n", g_pool.size(),
                g_pool_bytes / 1048576.0, (unsigned long long)g_malloc_calls, g_malloc_ms, (unsigned long long)g_free_calls, g_free_ms);
    }
    h->build_ms[4] = now_ms() - t3;
    return BFT_GPU_OK;
}

static int ensure_table(bft_gpu* h);
// need_table = false: the caller is answered by the k-mer hash alone ("compact_table": the sorted table may be away)
static int ensure_built(bft_gpu* h, bool need_table = true) {
    if (!h->built || h->log_n) CK(bft_gpu_build(h));
    if (need_table) CK(ensure_table(h));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// queries
// ------------------------------------------------------------------------------------------------
// Resident k_query workgroups per CU.  Measured on MI355X (tools/perf_probe.py): a one-level index whose suffix-group
// table exceeds the L2 runs at the beyond-L2 gather rate with 4 waves per SIMD and loses 10 % with 8; deep walks and
// L2-resident indexes gain 10-40 % from 8; which side an index falls on is measured, not guessed (tune_residency).
static int query_residency(const bft_gpu* h) {
    if (h->opt_wgs_per_cu) return h->opt_wgs_per_cu;
    return h->tuned_wgs ? h->tuned_wgs : 2;
}

static BftClaimCtr claim_counters(bft_gpu* h, hipStream_t s, uint64_t n, uint64_t units);
static void claims_launched(bft_gpu* h, hipStream_t s);
template <int W, bool STAGED, int PROBE>
static int launch_query_k(bft_gpu* h, const uint8_t* d_kmers, uint64_t n, uint64_t* d_bits64, uint32_t* d_rows, hipStream_t s, int rec) {
    // LDS: hash table + root Bloom block (<= 64 CCs) + root CC headers: at most two workgroups fit a CU (160 KB).  With
    // one workgroup per CU the request is padded past half the LDS so that the dispatcher cannot pair two on a CU.
    const int res = query_residency(h);  // 1: one 1024-thread workgroup per CU, 2: two (8 wavefronts per SIMD), 3: two of 768 threads (6 per SIMD)
    const int wgs = res == 1 ? 1 : 2;
    const uint32_t block = res == 3 ? BFT_BLOCK6 : 1024;
    // hash table + the root area (the root's Bloom block and CC headers, or -- with the derived root tables -- k_query's queue of deferred lanes)
    size_t lds = BFT_LDS_HM_BYTES + ((size_t)BFT_MODULO_HASH * 8 + 15) / 16 * 16 + BFT_LDS_ROOT_MAX_CC * sizeof(BftCCX);
    if (wgs == 1) lds = std::max<size_t>(lds, 84u << 10);
    const uint64_t resident = 256ull * (uint64_t)wgs;  // 256 CUs x resident workgroups per CU
    // chunks of 1024 k-mers per wavefront (query_body): the first by wavefront number, the others claimed from the stream's counter
    const uint64_t n_chunks = (n + 1023) / 1024, wg_chunks = (n_chunks + block / 64 - 1) / (block / 64);
    const BftClaimCtr ctr = claim_counters(h, s, n, n_chunks);
    const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(wg_chunks, resident * h->opt_grid_mult)));
    static std::atomic<uint64_t> attr_devs{0};  // the attribute is per device: one bit per device it was set on
    const uint64_t dev_bit = 1ull << (h->device & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & dev_bit)) {
        HIPCK(hipFuncSetAttribute((const void*)k_query<W, 1024, STAGED, PROBE>, hipFuncAttributeMaxDynamicSharedMemorySize, 84 << 10));
        HIPCK(hipFuncSetAttribute((const void*)k_query8<W, 1024, STAGED, PROBE>, hipFuncAttributeMaxDynamicSharedMemorySize, 84 << 10));
        HIPCK(hipFuncSetAttribute((const void*)k_query6<W, STAGED, PROBE>, hipFuncAttributeMaxDynamicSharedMemorySize, 84 << 10));
        attr_devs.fetch_or(dev_bit, std::memory_order_release);
    }
    // rec: bytes per input record (B, or 8W for the zero-padded word records of the sequence path); load_x reads that many
    if (h->im.walk_kh && PROBE == 0)  // ("walk_hash": launch_query leaves im.walk_kh set only for this)
        return bft_walkh_query(h->im, d_kmers, n, rec, d_bits64, d_rows, ctr, (uint32_t)h->opt_grid_mult, s);
    if (res == 1) hipLaunchKernelGGL((k_query<W, 1024, STAGED, PROBE>), grid, dim3(1024), lds, s, h->im, d_kmers, n, rec, d_bits64, d_rows, ctr);
    else if (res == 3) hipLaunchKernelGGL((k_query6<W, STAGED, PROBE>), grid, dim3(BFT_BLOCK6), lds, s, h->im, d_kmers, n, rec, d_bits64, d_rows, ctr);
    else hipLaunchKernelGGL((k_query8<W, 1024, STAGED, PROBE>), grid, dim3(1024), lds, s, h->im, d_kmers, n, rec, d_bits64, d_rows, ctr);
    HIPCK(hipGetLastError());
    return 0;
}

template <int W, bool STAGED>
static int launch_query_ws(bft_gpu* h, const uint8_t* d_kmers, uint64_t n, uint64_t* d_bits64, uint32_t* d_rows, hipStream_t s, int rec) {
    if (W <= BFT_PROBE_MAX_W && h->im.probe_big && !h->im.walk_kh) return launch_query_k<W, STAGED, 1>(h, d_kmers, n, d_bits64, d_rows, s, rec);
    return launch_query_k<W, STAGED, 0>(h, d_kmers, n, d_bits64, d_rows, s, rec);
}

template <int W>
static int launch_query_w(bft_gpu* h, const uint8_t* d_kmers, uint64_t n, uint64_t* d_bits64, uint32_t* d_rows, hipStream_t s, int rec) {
    const bool staged = h->root_ncc >= 1 && h->root_ncc <= BFT_LDS_ROOT_MAX_CC;
    return staged ? launch_query_ws<W, true>(h, d_kmers, n, d_bits64, d_rows, s, rec) : launch_query_ws<W, false>(h, d_kmers, n, d_bits64, d_rows, s, rec);
}

// through_kh: plain root groups are looked up in the k-mer hash (k_query6h) -- the caller wants presence or colour sets, not rows
static int launch_query_walk(bft_gpu* h, const uint8_t* d_kmers, uint64_t n, uint64_t* d_bits64, uint32_t* d_rows, hipStream_t s, int rec_bytes = 0, bool through_kh = false) {
    if (n == 0) return 0;
    const int rec = rec_bytes ? rec_bytes : h->B;
    hipEvent_t e0, e1;
    CK(timing_begin(h, s, &e0, &e1));
    const uint32_t keep = h->im.walk_kh;
    if (!through_kh) h->im.walk_kh = 0;
    int rc = 0;
    switch (h->W) {
    case 1: rc = launch_query_w<1>(h, d_kmers, n, d_bits64, d_rows, s, rec); break;
    case 2: rc = launch_query_w<2>(h, d_kmers, n, d_bits64, d_rows, s, rec); break;
    case 3: rc = launch_query_w<3>(h, d_kmers, n, d_bits64, d_rows, s, rec); break;
    default: rc = launch_query_w<4>(h, d_kmers, n, d_bits64, d_rows, s, rec); break;
    }
    h->im.walk_kh = keep;
    CK(rc);
    claims_launched(h, s);
    CK(timing_end(h, s, e0, e1));
    return 0;
}

// The claim counters of the handle: allocated and zeroed once, when the first image is committed (bft_gpu_build, bft_gpu_image_unpack) --
// never on the path of a query call.
static void ensure_claim_counters(bft_gpu* h) {
    if (h->kh_ctr || h->kh_ctr_failed) return;
    if (hipMalloc((void**)&h->kh_ctr, bft_gpu::KH_CTR_SLOTS * 8) != hipSuccess || hipMemset(h->kh_ctr, 0, bft_gpu::KH_CTR_SLOTS * 8) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {  // (blocking: no launch of any stream can meet a counter that is not zero yet)
        (void)hipGetLastError();
        if (h->kh_ctr) (void)hipFree(h->kh_ctr);
        h->kh_ctr = nullptr;
        h->kh_ctr_failed = true;
    }
}
// The claim counter of stream s (see struct bft_gpu) with the range this launch may use, or {NULL}: every round of blocks is dealt out by
// workgroup number.  units: what the launch deals out (blocks of 256 k-mers, chunks of 1024) -- the slot's base moves on by that plus what the
// resident workgroups can claim beyond it, so the next launch starts above anything this one can leave behind, finished or not
// (bft_claims.h).  A handle keeps a counter for each of KH_CTR_SLOTS streams; queried on more, it hands the least recently used slot to the
// new stream once that slot's last launch is known to be over (an event per slot), else the launch runs static rounds and is counted
// (bft_gpu_build_time, entry 20: nothing is silent).
static BftClaimCtr claim_counters(bft_gpu* h, hipStream_t s, uint64_t n, uint64_t units) {
    static const bool env_off = getenv("BFT_GPU_QUERY_DYNAMIC") && atoi(getenv("BFT_GPU_QUERY_DYNAMIC")) == 0;
    h->kh_ctr_pending = -1;  // (a launch that failed between claim_counters and claims_launched must not leave its slot behind for the next one)
    if (!h->opt_query_dynamic || env_off || n < h->opt_query_dynamic_min || !h->kh_ctr) return BftClaimCtr{nullptr, 0};
    {   // a launch being recorded into a graph gets static rounds: `base` is a kernel argument, frozen at capture time, and from the second replay on the
        // counter stands above it -- every workgroup would answer its first round only (the counter cannot be reset on the device without giving up
        // the property that nothing an unfinished launch leaves behind can hurt the next one)
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess) (void)hipGetLastError();
        else if (st != hipStreamCaptureStatusNone) return BftClaimCtr{nullptr, 0};
    }
    int slot = -1;
    for (int i = 0; i < h->kh_ctr_used; i++)
        if (h->kh_ctr_stream[i] == s) slot = i;
    if (slot < 0 && h->kh_ctr_used < bft_gpu::KH_CTR_SLOTS) {
        slot = h->kh_ctr_used++;
        h->kh_ctr_stream[slot] = s;
    }
    if (slot < 0) {  // every slot taken: the least recently used one, if its stream has nothing of ours in flight any more
        int lru = 0;
        for (int i = 1; i < bft_gpu::KH_CTR_SLOTS; i++)
            if (h->kh_ctr_tick[i] < h->kh_ctr_tick[lru]) lru = i;
        // (a slot last used before the handle started recording events -- its stream may be gone by now, so there is nothing to ask but the device:
        // once per such slot at most)
        const bool over = h->kh_ctr_ev[lru] ? hipEventQuery(h->kh_ctr_ev[lru]) == hipSuccess : hipDeviceSynchronize() == hipSuccess;
        if (over) {
            slot = lru;
            h->kh_ctr_stream[slot] = s;
        } else {
            (void)hipGetLastError();
            h->claims_static_launches++;
            return BftClaimCtr{nullptr, 0};
        }
    }
    h->kh_ctr_tick[slot] = ++h->kh_ctr_clock;
    h->kh_ctr_pending = slot;  // (claims_launched records the slot's event behind the launch)
    const BftClaimCtr c{h->kh_ctr + slot, h->kh_ctr_base[slot]};
    h->kh_ctr_base[slot] += units + ((unsigned long long)1 << 24);  // (2^24: more than the resident workgroups x the largest claim)
    return c;
}
// behind a launch that was given a counter: where the slot's last use ends (only looked at when the slots run out)
static void claims_launched(bft_gpu* h, hipStream_t s) {
    const int slot = h->kh_ctr_pending;
    h->kh_ctr_pending = -1;
    if (slot < 0 || h->kh_ctr_used < bft_gpu::KH_CTR_SLOTS / 2) return;  // (no event traffic on a handle that is queried on a few streams)
    if (!h->kh_ctr_ev[slot] && hipEventCreateWithFlags(&h->kh_ctr_ev[slot], hipEventDisableTiming) != hipSuccess) { h->kh_ctr_ev[slot] = nullptr; (void)hipGetLastError(); return; }
    if (hipEventRecord(h->kh_ctr_ev[slot], s) != hipSuccess) (void)hipGetLastError();
}

// Presence (and, with im.emit_cs, the colour set of every found k-mer into d_out32) through the k-mer hash.
static int launch_query_kh(bft_gpu* h, const uint8_t* d_kmers, uint64_t n, uint64_t* d_bits64, uint32_t* d_out32, hipStream_t s, int rec) {
    hipEvent_t e0, e1;
    CK(timing_begin(h, s, &e0, &e1));
    CK(bft_kh_query(h->im, h->opt_grid_mult, d_kmers, n, rec, d_bits64, d_out32, claim_counters(h, s, n, (n + 255) / 256), h->opt_query_chunk, s));
    HIPCK(hipGetLastError());
    claims_launched(h, s);
    CK(timing_end(h, s, e0, e1));
    return 0;
}

// Which kernel a batch takes: the k-mer hash answers presence and colour sets (one cache line per k-mer); rows -- positions in the
// sorted table, what the reference keeps in resultPresence -- come from the container walk, as does everything on an image
// without the table ("kmer_hash" 0, k >= 64, 2k % 64 == 0).  Same answers either way (tests/test_gpu_parity.py).
static int launch_query(bft_gpu* h, const uint8_t* d_kmers, uint64_t n, uint64_t* d_bits64, uint32_t* d_rows, hipStream_t s, int rec_bytes = 0) {
    if (n == 0) return 0;
    const int rec = rec_bytes ? rec_bytes : h->B;
    const bool no_rows = d_rows == nullptr || h->im.emit_cs;  // presence or colour sets: what the k-mer hash holds
    if (h->im.kh_lines != nullptr && no_rows && !h->opt_walk_hash) return launch_query_kh(h, d_kmers, n, d_bits64, d_rows, s, rec);
    CK(ensure_table(h));
    // the walk looks plain root groups up in the k-mer hash when no row is asked for ("walk_hash")
    return launch_query_walk(h, d_kmers, n, d_bits64, d_rows, s, rec, no_rows && h->opt_walk_hash);
}

// Synthetic batch for tune_residency: k-mers of the index itself (pseudo-random rows of tk), every other one with a
// single-nucleotide change -- the mix of present k-mers and near misses that exercises the whole walk.
template <int W>
__global__ void k_tune_queries(const uint64_t* __restrict__ tk, uint64_t n_kmers, int k, int B, uint64_t m, uint8_t* __restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
        const uint64_t row = z % n_kmers;
        uint64_t t[W], x[W];
#pragma unroll
        for (int w = 0; w < W; w++) t[w] = tk[row * W + w];
        bft_x_from_tform<W>(t, k, x);
        if (i & 1) {
            const int pos = (int)((z >> 33) % (uint64_t)k);
            const uint64_t flip = 1 + ((z >> 20) % 3);
#pragma unroll
            for (int w = 0; w < W; w++)
                if (w == (2 * pos) >> 6) x[w] ^= flip << ((2 * pos) & 63);
        }
        for (int b = 0; b < B; b++) {
            uint64_t v = 0;
#pragma unroll
            for (int w = 0; w < W; w++)
                if (w == (b >> 3)) v = x[w];
            out[i * B + b] = (uint8_t)(v >> (8 * (b & 7)));
        }
    }
}

// bft_gpu_set_option("tune", 1): the launch shape of the container walk measured on THIS image, on a batch drawn from the index
// (synchronises; never called implicitly -- a build or a query only ever applies default_launch_shape).
static int tune_residency(bft_gpu* h) {
    if (h->n_kmers < (1u << 16)) {  // small (L2-resident) indexes: always two workgroups, 4-row probes
        h->tuned_wgs = 2;
        h->tuned_probe = 4;
        h->im.probe_big = probe_mode(h->opt_probe);
        return 0;
    }
    const uint64_t m = 1ull << 22;
    DevBuf q, bits;
    CK(q.alloc(m * h->B));
    CK(bits.alloc(((m + 63) / 64) * 8));
    const dim3 grid(grid_for((m + 255) / 256)), block(256);
    switch (h->W) {
    case 1: hipLaunchKernelGGL(k_tune_queries<1>, grid, block, 0, h->stream, h->im.tk, h->n_kmers, h->k, h->B, m, q.as<uint8_t>()); break;
    case 2: hipLaunchKernelGGL(k_tune_queries<2>, grid, block, 0, h->stream, h->im.tk, h->n_kmers, h->k, h->B, m, q.as<uint8_t>()); break;
    case 3: hipLaunchKernelGGL(k_tune_queries<3>, grid, block, 0, h->stream, h->im.tk, h->n_kmers, h->k, h->B, m, q.as<uint8_t>()); break;
    default: hipLaunchKernelGGL(k_tune_queries<4>, grid, block, 0, h->stream, h->im.tk, h->n_kmers, h->k, h->B, m, q.as<uint8_t>()); break;
    }
    HIPCK(hipGetLastError());
    const bool timing = h->timing;
    const uint32_t dbg = h->im.debug_stop;
    h->timing = false;
    h->im.debug_stop = 0;  // (only read by -DBFT_PERF_PROBE builds)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = fail(BFT_GPU_E_HIP, "hipEventCreate failed");
    // (residency 1 / 2 / 3) x (probe block 4 / 8 rows); options fixed by the caller are not varied
    float best[6] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
    for (int cfg = 0; cfg < 6 && rc == 0; cfg++) {
        const int wgs = 1 + cfg % 3, probe = cfg >= 3 ? 8 : 4;
        if ((h->opt_wgs_per_cu && h->opt_wgs_per_cu != wgs) || (h->opt_probe && h->opt_probe != probe) || (h->W > BFT_PROBE_MAX_W && probe == 8))
            continue;
        h->tuned_wgs = wgs;
        h->im.probe_big = probe_mode(probe);
        for (int rep = 0; rep < 2 && rc == 0; rep++) {  // the first repetition warms the caches, the second counts (large batches measure again: launch_query)
            if (hipEventRecord(e0, h->stream) != hipSuccess) rc = fail(BFT_GPU_E_HIP, "hipEventRecord failed");
            if (rc == 0) rc = launch_query_walk(h, q.as<uint8_t>(), m, bits.as<uint64_t>(), nullptr, h->stream);
            if (rc == 0 && (hipEventRecord(e1, h->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = fail(BFT_GPU_E_HIP, "k_query failed while tuning");
            float ms = 0;
            if (rc == 0 && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = fail(BFT_GPU_E_HIP, "hipEventElapsedTime failed");
            if (rep > 0 && ms < best[cfg]) best[cfg] = ms;
        }
    }
    // Residency 3 is the incumbent: on full batches and on the eight walks per k-mer of k_branching it is as fast as the better of
    // the other two or 5-15 % faster (k = 36..63, config 5), and the 2^22-query tuning batch resolves differences of a few per
    // cent poorly -- another arrangement has to win by more than 4 %.
    int win = -1;
    float win_score = 1e30f;
    for (int cfg = 0; cfg < 6; cfg++) {
        if (best[cfg] >= 1e29f) continue;
        const float score = best[cfg] * (cfg % 3 == 2 ? 0.96f : 1.0f);
        if (score < win_score) { win_score = score; win = cfg; }
    }
    if (win < 0) win = 1;
    const int win_wgs = 1 + win % 3, win_probe = win >= 3 ? 8 : 4;
    // root level: range table + direct table, or the direct table alone ("root_direct" 3).  The range table halves the L2
    // footprint of the root level; its special prefixes (child Nodes) pay one more dependent load -- which side wins depends
    // on how many there are, so both are timed with the residency / probe mode just chosen.
    if (rc == 0 && h->opt_root_direct == 3 && h->rstart_ok) {
        h->tuned_wgs = win_wgs;
        h->im.probe_big = probe_mode(h->opt_probe ? h->opt_probe : win_probe);
        float rs[2] = {1e30f, 1e30f};
        for (int mode = 0; mode < 2 && rc == 0; mode++) {
            h->im.rstart = mode ? h->d_rstart.as<uint32_t>() : nullptr;
            h->im.rq = mode && h->rq_ok ? h->d_rq.as<uint32_t>() : nullptr;
            for (int rep = 0; rep < 3 && rc == 0; rep++) {
                if (hipEventRecord(e0, h->stream) != hipSuccess) rc = fail(BFT_GPU_E_HIP, "hipEventRecord failed");
                if (rc == 0) rc = launch_query_walk(h, q.as<uint8_t>(), m, bits.as<uint64_t>(), nullptr, h->stream);
                if (rc == 0 && (hipEventRecord(e1, h->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = fail(BFT_GPU_E_HIP, "k_query failed while tuning");
                float ms = 0;
                if (rc == 0 && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = fail(BFT_GPU_E_HIP, "hipEventElapsedTime failed");
                if (rep > 0 && ms < rs[mode]) rs[mode] = ms;
            }
        }
        h->rstart_tune_ms[0] = rs[0];
        h->rstart_tune_ms[1] = rs[1];
        h->tuned_rstart = rs[1] < rs[0] ? 1 : 0;
        h->im.rstart = h->tuned_rstart ? h->d_rstart.as<uint32_t>() : nullptr;
        h->im.rq = h->tuned_rstart && h->rq_ok ? h->d_rq.as<uint32_t>() : nullptr;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    h->timing = timing;
    h->im.debug_stop = dbg;
    h->tuned_wgs = 0;
    h->im.probe_big = probe_mode(h->opt_probe);
    CK(rc);
    for (int r = 0; r < 3; r++) h->tune_ms[r] = std::min(best[r], best[r + 3]) < 1e29f ? std::min(best[r], best[r + 3]) : 0;
    h->tuned_wgs = win_wgs;
    h->tuned_probe = win_probe;
    h->im.probe_big = probe_mode(h->opt_probe ? h->opt_probe : h->tuned_probe);
    return 0;
}

template <int W, bool STAGED, int PROBE>
static int launch_branching_k(bft_gpu* h, const uint8_t* d_kmers, uint64_t n, uint64_t* d_bits64, uint8_t* d_counts, hipStream_t s) {
    const int res = query_residency(h);  // eight walks per k-mer: the residency measured for k_query applies
    const int wgs = res == 1 ? 1 : 2;
    const uint32_t block = res == 3 ? BFT_BLOCK6 : 1024;
    // hash table + the root area (the root's Bloom block and CC headers, or -- with the derived root tables -- k_query's queue of deferred lanes)
    size_t lds = BFT_LDS_HM_BYTES + ((size_t)BFT_MODULO_HASH * 8 + 15) / 16 * 16 + BFT_LDS_ROOT_MAX_CC * sizeof(BftCCX);
    if (wgs == 1) lds = std::max<size_t>(lds, 84u << 10);
    const uint64_t nblk = (n + block - 1) / block;
    const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nblk, 256ull * (uint64_t)wgs)));
    static std::atomic<uint64_t> attr_devs{0};
    const uint64_t dev_bit = 1ull << (h->device & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & dev_bit)) {
        HIPCK(hipFuncSetAttribute((const void*)k_branching<W, 1024, STAGED, PROBE>, hipFuncAttributeMaxDynamicSharedMemorySize, 84 << 10));
        HIPCK(hipFuncSetAttribute((const void*)k_branching8<W, 1024, STAGED, PROBE>, hipFuncAttributeMaxDynamicSharedMemorySize, 84 << 10));
        HIPCK(hipFuncSetAttribute((const void*)k_branching6<W, STAGED, PROBE>, hipFuncAttributeMaxDynamicSharedMemorySize, 84 << 10));
        attr_devs.fetch_or(dev_bit, std::memory_order_release);
    }
    if (res == 1) hipLaunchKernelGGL((k_branching<W, 1024, STAGED, PROBE>), grid, dim3(1024), lds, s, h->im, d_kmers, n, h->B, d_bits64, d_counts);
    else if (res == 3) hipLaunchKernelGGL((k_branching6<W, STAGED, PROBE>), grid, dim3(BFT_BLOCK6), lds, s, h->im, d_kmers, n, h->B, d_bits64, d_counts);
    else hipLaunchKernelGGL((k_branching8<W, 1024, STAGED, PROBE>), grid, dim3(1024), lds, s, h->im, d_kmers, n, h->B, d_bits64, d_counts);
    HIPCK(hipGetLastError());
    return 0;
}

static int launch_branching(bft_gpu* h, const uint8_t* d_kmers, uint64_t n, uint64_t* d_bits64, uint8_t* d_counts, hipStream_t s) {
    if (n == 0) return 0;
    const bool staged = h->root_ncc >= 1 && h->root_ncc <= BFT_LDS_ROOT_MAX_CC;
    hipEvent_t e0, e1;
    CK(timing_begin(h, s, &e0, &e1));
    if (h->im.kh_lines != nullptr) {  // eight candidates per k-mer, each one cache line of the k-mer hash, four in flight at a time
        CK(bft_kh_branching(h->im, d_kmers, n, h->B, d_bits64, d_counts, claim_counters(h, s, n * 8, (n + 255) / 256), h->opt_query_chunk, s));
        claims_launched(h, s);
        CK(timing_end(h, s, e0, e1));
        return 0;
    }
#define BR(WW, PP) (staged ? launch_branching_k<WW, true, PP>(h, d_kmers, n, d_bits64, d_counts, s) : launch_branching_k<WW, false, PP>(h, d_kmers, n, d_bits64, d_counts, s))
    switch (h->W) {
    case 1: CK(BR(1, 0)); break;
    case 2: CK(BR(2, 0)); break;
    case 3: CK(BR(3, 0)); break;
    default: CK(BR(4, 0)); break;
    }
#undef BR
    CK(timing_end(h, s, e0, e1));
    return 0;
}

extern "C" int bft_gpu_query_branching_dev(bft_gpu* h, const void* d_kmers, uint64_t n, void* d_branching_bits, void* d_counts, void* hip_stream) {
    if (!h || ((!d_kmers || !d_branching_bits) && n)) return fail(BFT_GPU_E_ARG, "NULL argument");
    ENTER(h);
    CK(ensure_built(h, false));  // (the k-mer hash answers; the walk fetches the table itself)
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    CK(launch_branching(h, (const uint8_t*)d_kmers, n, (uint64_t*)d_branching_bits, (uint8_t*)d_counts, s));
    return note_foreign_stream(h, s);
}

extern "C" int bft_gpu_query_branching(bft_gpu* h, const uint8_t* kmers, uint64_t n, uint8_t* branching_bits, uint8_t* counts) {
    if (!h || ((!kmers || !branching_bits) && n)) return fail(BFT_GPU_E_ARG, "NULL argument");
    ENTER(h);
    CK(ensure_built(h, false));  // (the k-mer hash answers; the walk fetches the table itself)
    const uint64_t chunk = 1ull << 26;
    const uint64_t mc = std::min(n, chunk);
    DevBuf dk, db, dc;
    CK(dk.alloc(mc * h->B));
    CK(db.alloc(((mc + 63) / 64) * 8));
    if (counts) CK(dc.alloc(mc));
    for (uint64_t a = 0; a < n; a += chunk) {
        const uint64_t m = std::min(chunk, n - a);
        HIPCK(hipMemcpyAsync(dk.p, kmers + a * h->B, m * h->B, hipMemcpyHostToDevice, h->stream));
        CK(launch_branching(h, dk.as<uint8_t>(), m, db.as<uint64_t>(), counts ? dc.as<uint8_t>() : nullptr, h->stream));
        HIPCK(hipMemcpyAsync(branching_bits + a / 8, db.p, (m + 7) / 8, hipMemcpyDeviceToHost, h->stream));
        if (counts) HIPCK(hipMemcpyAsync(counts + a, dc.p, m, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
    }
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_query_presence_dev(bft_gpu* h, const void* d_kmers, uint64_t n, void* d_present_bits, void* hip_stream) {
    if (!h || ((!d_kmers || !d_present_bits) && n)) return fail(BFT_GPU_E_ARG, "NULL argument");
    ENTER(h);
    CK(ensure_built(h, false));  // (the k-mer hash answers; the walk fetches the table itself)
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    CK(launch_query(h, (const uint8_t*)d_kmers, n, (uint64_t*)d_present_bits, nullptr, s));
    return note_foreign_stream(h, s);
}

// ---- small host batches through the pinned block -------------------------------------------------------------------------
#define BFT_PIN_MAX_N 4096u                       // k-mers per call served this way
#define BFT_PIN_IN (BFT_PIN_MAX_N * 32u)          // B <= 32 bytes per k-mer
#define BFT_PIN_BITS_OFF BFT_PIN_IN
#define BFT_PIN_ROWS_OFF (BFT_PIN_BITS_OFF + BFT_PIN_MAX_N / 8u)
#define BFT_PIN_SETS_OFF (BFT_PIN_ROWS_OFF + BFT_PIN_MAX_N * 4u)
#define BFT_PIN_BYTES (BFT_PIN_SETS_OFF + BFT_PIN_MAX_N * 4u)

static int pin_block(bft_gpu* h) {
    if (h->pin) return 0;
    void* p = nullptr;
    HIPCK(hipHostMalloc(&p, BFT_PIN_BYTES, hipHostMallocMapped));
    h->pin = (uint8_t*)p;
    return 0;
}

// presence (+ rows, + colour-set ids) of n <= BFT_PIN_MAX_N host k-mers; any output pointer may be NULL
static int query_small(bft_gpu* h, const uint8_t* kmers, uint64_t n, uint8_t* present_bits, uint32_t* rows, uint32_t* colorsets) {
    CK(pin_block(h));
    uint8_t* pin = h->pin;
    memcpy(pin, kmers, n * h->B);
    uint32_t* prow = (uint32_t*)(pin + BFT_PIN_ROWS_OFF);
    uint32_t* pset = (uint32_t*)(pin + BFT_PIN_SETS_OFF);
    const bool want_rows = rows || colorsets;
    CK(launch_query(h, pin, n, (uint64_t*)(pin + BFT_PIN_BITS_OFF), want_rows ? prow : nullptr, h->stream));
    if (colorsets) {
        hipLaunchKernelGGL(k_row_colorsets, dim3(grid_for((n + 255) / 256)), dim3(256), 0, h->stream, prow, h->im.tcol, n, pset);
        HIPCK(hipGetLastError());
    }
    HIPCK(hipStreamSynchronize(h->stream));
    if (present_bits) memcpy(present_bits, pin + BFT_PIN_BITS_OFF, (n + 7) / 8);
    if (rows) memcpy(rows, prow, n * 4);
    if (colorsets) memcpy(colorsets, pset, n * 4);
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_query_presence(bft_gpu* h, const uint8_t* kmers, uint64_t n, uint8_t* present_bits) {
    if (!h || ((!kmers || !present_bits) && n)) return fail(BFT_GPU_E_ARG, "NULL argument");
    ENTER(h);
    CK(ensure_built(h, false));  // (the k-mer hash answers; the walk fetches the table itself)
    if (n && n <= BFT_PIN_MAX_N) return query_small(h, kmers, n, present_bits, nullptr, nullptr);
    const uint64_t chunk = 1ull << 26;  // multiple of 64: chunks are byte aligned in the bitmap
    DevBuf dk, db;
    CK(dk.alloc(std::min(n, chunk) * h->B));
    CK(db.alloc(((std::min(n, chunk) + 63) / 64) * 8));
    for (uint64_t a = 0; a < n; a += chunk) {
        const uint64_t m = std::min(chunk, n - a);
        HIPCK(hipMemcpyAsync(dk.p, kmers + a * h->B, m * h->B, hipMemcpyHostToDevice, h->stream));
        CK(launch_query(h, dk.as<uint8_t>(), m, db.as<uint64_t>(), nullptr, h->stream));
        HIPCK(hipMemcpyAsync(present_bits + a / 8, db.p, (m + 7) / 8, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
    }
    return BFT_GPU_OK;
}

// shared front half of the colour queries: rows + presence bits for one chunk
static int query_rows(bft_gpu* h, const uint8_t* kmers, uint64_t m, DevBuf& dk, DevBuf& db, DevBuf& dr, uint8_t* present_bits) {
    HIPCK(hipMemcpyAsync(dk.p, kmers, m * h->B, hipMemcpyHostToDevice, h->stream));
    CK(launch_query(h, dk.as<uint8_t>(), m, db.as<uint64_t>(), dr.as<uint32_t>(), h->stream));
    if (present_bits) HIPCK(hipMemcpyAsync(present_bits, db.p, (m + 7) / 8, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

// get_annotation + get_list_id_genomes for a resident batch (src/bft.c:363-387, 622-641; src/annotation.c:2086-2250): the k-mer hash hands out
// the colour-set id of every found k-mer (emit_cs: the same line that answers presence), the lists' lengths are scanned into offsets straight
// from the dictionary, and the wavefronts stream the ids out.  No row, no sorted table ("compact_table" stays), no host round trip.
static int colors_core(bft_gpu* h, const uint8_t* d_kmers, uint64_t n, uint64_t* d_bits64, uint64_t* d_offsets, uint32_t* d_ids, uint64_t ids_cap, uint64_t* d_needed,
                       hipStream_t s, bool fill) {
    if (h->qc_used && h->qc_stream != s) {  // (the scratch belongs to the handle: one stream at a time -- behind an event of the handle's own, not the other stream)
        const hipError_t e = h->qc_ev ? hipEventSynchronize(h->qc_ev) : hipDeviceSynchronize();
        h->qc_stream = s;
        HIPCK(e);
    }
    if (fill && h->im.kh_lines != nullptr && !h->opt_walk_hash && bft_kh_has_kernels(h->W, h->im.kh.S) && h->im.nb_genomes < 65536u) {  // (k_colors_kh keeps list lengths in 16 bits)
        // through the k-mer hash: lookup, offsets and ids in ONE launch (k_colors_kh; the host entry point counts first and fills per chunk: the three steps below)
        const size_t sb = bft_kh_colors_scratch_bytes(n);
        if (h->qc_tmp.bytes < sb) {
            if (h->qc_used) HIPCK(hipStreamSynchronize(s));
            CK(h->qc_tmp.alloc(sb + sb / 2));
        }
        h->qc_used = true;
        h->qc_stream = s;
        hipEvent_t e0, e1;
        CK(timing_begin(h, s, &e0, &e1));
        CK(bft_kh_colors(h->im, d_kmers, n, h->B, d_bits64, d_offsets, d_ids, d_ids ? ids_cap : 0, d_needed, h->qc_tmp.p, s));
        CK(timing_end(h, s, e0, e1));
        if (!h->qc_ev && hipEventCreateWithFlags(&h->qc_ev, hipEventDisableTiming) != hipSuccess) { h->qc_ev = nullptr; (void)hipGetLastError(); }
        if (h->qc_ev && hipEventRecord(h->qc_ev, s) != hipSuccess) (void)hipGetLastError();
        return 0;
    }
    const size_t tb = bft_scan::scratch_bytes(n + 1);  // (the scan's tile states)
    if (h->qc_cs.bytes < n * 4 || h->qc_tmp.bytes < tb) {
        if (h->qc_used) HIPCK(hipStreamSynchronize(s));
        if (h->qc_cs.bytes < n * 4) CK(h->qc_cs.alloc(n * 4 + n / 2));
        if (h->qc_tmp.bytes < tb) CK(h->qc_tmp.alloc(tb + tb / 2));
    }
    h->qc_used = true;
    h->qc_stream = s;
    uint32_t* d_cs = h->qc_cs.as<uint32_t>();
    {   // (the k-mer hash has the colour set in the line that answers presence; the walk, on an image without the table, takes it from tcol[row])
        h->im.emit_cs = 1;
        const int rc = launch_query(h, d_kmers, n, d_bits64, d_cs, s);
        h->im.emit_cs = 0;
        CK(rc);
    }
    const BftCsLen len{d_cs, h->im.cs_off, n};
    CK((bft_scan::exclusive_sum<uint64_t>(len, d_offsets, n + 1, s, h->qc_tmp)));
    if (fill) {
        hipLaunchKernelGGL(k_color_fill_cs, dim3(grid_for((n + 255) / 256)), dim3(256), 0, s, d_cs, h->im.cs_off, h->im.cs_ids, h->im.cs_w, d_offsets, n, ids_cap, d_ids, d_needed);
        HIPCK(hipGetLastError());
    }
    if (!h->qc_ev && hipEventCreateWithFlags(&h->qc_ev, hipEventDisableTiming) != hipSuccess) { h->qc_ev = nullptr; (void)hipGetLastError(); }
    if (h->qc_ev && hipEventRecord(h->qc_ev, s) != hipSuccess) (void)hipGetLastError();
    return 0;
}

extern "C" int bft_gpu_query_colors_dev(bft_gpu* h, const void* d_kmers, uint64_t n, void* d_present_bits, void* d_offsets, void* d_ids, uint64_t ids_cap,
                                        void* d_ids_needed, void* hip_stream) {
    if (!h || !d_offsets || ((!d_kmers || !d_present_bits) && n)) return fail(BFT_GPU_E_ARG, "NULL argument");
    ENTER(h);
    CK(ensure_built(h, false));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    if (n == 0) {
        CK(bft_zero_async(d_offsets, 8, s));  // (kernels, not memsets, wherever a caller may be capturing: bft_dev.h)
        if (d_ids_needed) CK(bft_zero_async(d_ids_needed, 8, s));
        return note_foreign_stream(h, s);
    }
    CK(colors_core(h, (const uint8_t*)d_kmers, n, (uint64_t*)d_present_bits, (uint64_t*)d_offsets, (uint32_t*)d_ids, ids_cap, (uint64_t*)d_ids_needed, s, true));
    return note_foreign_stream(h, s);
}

// The host-buffer form: the resident one on staged chunks.
extern "C" int bft_gpu_query_colors(bft_gpu* h, const uint8_t* kmers, uint64_t n, uint8_t* present_bits, uint64_t* offsets,
                                    uint32_t* ids, uint64_t ids_cap, uint64_t* ids_needed) {
    if (!h || !offsets || ((!kmers) && n)) return fail(BFT_GPU_E_ARG, "NULL argument");
    ENTER(h);
    CK(ensure_built(h, false));
    const uint64_t chunk = 1ull << 24;
    const uint64_t mc = std::min(n, chunk);
    DevBuf dk, db, doff, dids;
    CK(dk.alloc(mc * h->B));
    CK(db.alloc(((mc + 63) / 64) * 8));
    CK(doff.alloc((mc + 1) * 8));
    uint64_t total = 0;
    bool overflow = false;
    for (uint64_t a = 0; a < n; a += chunk) {
        const uint64_t m = std::min(chunk, n - a);
        HIPCK(hipMemcpyAsync(dk.p, kmers + a * h->B, m * h->B, hipMemcpyHostToDevice, h->stream));
        CK(colors_core(h, dk.as<uint8_t>(), m, db.as<uint64_t>(), doff.as<uint64_t>(), nullptr, 0, nullptr, h->stream, false));
        if (present_bits) HIPCK(hipMemcpyAsync(present_bits + a / 8, db.p, (m + 7) / 8, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipMemcpyAsync(offsets + a, doff.p, (m + 1) * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
        const uint64_t cnt = offsets[a + m];
        if (total)
            for (uint64_t i = 0; i <= m; i++) offsets[a + i] += total;  // chunk-local -> global
        if (!overflow && ids && total + cnt <= ids_cap) {
            if (cnt) {
                CK(dids.alloc(cnt * 4));
                hipLaunchKernelGGL(k_color_fill_cs, dim3(grid_for((m + 255) / 256)), dim3(256), 0, h->stream, h->qc_cs.as<uint32_t>(), h->im.cs_off, h->im.cs_ids, h->im.cs_w,
                                   doff.as<uint64_t>(), m, cnt, dids.as<uint32_t>(), (uint64_t*)nullptr);
                HIPCK(hipMemcpyAsync(ids + total, dids.p, cnt * 4, hipMemcpyDeviceToHost, h->stream));
                HIPCK(hipStreamSynchronize(h->stream));
            }
        } else
            overflow = true;
        total += cnt;
    }
    if (n == 0) offsets[0] = 0;
    if (ids_needed) *ids_needed = total;
    if (overflow) return fail(BFT_GPU_E_NOSPACE, "ids buffer too small");
    return BFT_GPU_OK;
}

// Bitmap form of the colour-set dictionary (one dword-aligned row per set), used by the colour-row queries when it stays
// below 4 GiB; derived on the first such query of an image, on the handle's stream.
static int ensure_cs_bitmaps(bft_gpu* h) {
    if (h->cs_bm_tried) return 0;
    h->cs_bm_tried = true;
    const uint64_t rowbytes = (h->im.nb_genomes + 7) / 8, nsets = h->n_sets;
    if (rowbytes && nsets && nsets * rowbytes <= (4ull << 30)) {
        const uint64_t stride = (rowbytes + 3) & ~3ull;  // dword-aligned dictionary rows (k_color_rows_bm)
        // (CS_BM_SLACK zero bytes in front of the first row and behind the last: k_color_rows_bm16 loads 16 bytes from up to 15 bytes before a row
        // and up to its last byte)
        CK(h->d_cs_bm.alloc_zero(nsets * stride + 2 * CS_BM_SLACK, h->stream));
        hipLaunchKernelGGL(k_cs_bitmaps, dim3(grid_for((nsets + 255) / 256)), dim3(256), 0, h->stream, h->im.cs_off, h->im.cs_ids, h->im.cs_w, nsets, (uint32_t)stride,
                           h->d_cs_bm.as<uint8_t>() + CS_BM_SLACK);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(h->stream));  // the row kernel may run on a caller's stream
        h->has_cs_bm = true;
    }
    return 0;
}

// d_rowidx: the row of every k-mer (scratch: overwritten with the colour-set ids when the bitmap dictionary is used)
static int launch_color_rows(bft_gpu* h, uint32_t* d_rowidx, uint64_t n, uint32_t rowbytes, uint8_t* d_out, hipStream_t s, bool are_colorsets = false) {
    CK(ensure_cs_bitmaps(h));
    if (h->has_cs_bm) {
        if (!are_colorsets)
            hipLaunchKernelGGL(k_row_colorsets, dim3(grid_for((n + 255) / 256)), dim3(256), 0, s, d_rowidx, h->im.tcol, n, d_rowidx);  // row -> colour set, in place
        // tiles of ~32 KiB of output (a multiple of 4 k-mers: tiles start dword aligned); magic number of the division by rowbytes
        // (round-up method, exact on u32)
        // (16-byte rows and up, 16-byte aligned output: the 16-bytes-per-lane kernel, whose tiles are a multiple of 16 k-mers)
        const bool wide16 = rowbytes >= 16 && ((uintptr_t)d_out & 15u) == 0;
        uint32_t tile_rows = std::min<uint32_t>(CR_MAX_TILE_ROWS, std::max<uint32_t>(4u, ((32768u / rowbytes) + 3u) & ~3u));
        if (wide16) {
            // tiles of the 16-byte kernel belong to wavefronts, which answer 64 x CR16_UNROLL chunks of 16 bytes per turn: among the tiles of
            // 16..64 KiB of output (a multiple of 16 k-mers, at most CR16_WAVE_ROWS) the one whose last turn is the fullest
            const uint32_t per_turn = 64u * CR16_UNROLL;
            double best = -1.0;
            tile_rows = 16u;
            for (uint32_t tr = 16u; tr <= CR16_WAVE_ROWS; tr += 16u) {
                const uint64_t bytes = (uint64_t)tr * rowbytes;
                // (tiles of 2-4, 4-8, 8-16 and 128-256 KiB were measured on config 5: 0.29-0.32 ms per GB written, no better than these)
                if (bytes > (64u << 10) && best >= 0.0) break;
                if (bytes < (16u << 10) && tr + 16u <= CR16_WAVE_ROWS && (uint64_t)(tr + 16u) * rowbytes <= (64u << 10)) continue;
                const uint64_t nch = (bytes + 15u) / 16u, turns = (nch + per_turn - 1) / per_turn;
                const double eff = (double)nch / (double)(turns * per_turn);
                if (eff >= best) { best = eff; tile_rows = tr; }
            }
        }
        uint32_t div_l = 0;
        while ((1ull << div_l) < rowbytes) div_l++;
        const uint32_t div_m = div_l ? (uint32_t)(((1ull << 32) * ((1ull << div_l) - rowbytes)) / rowbytes + 1ull) : 0u;
        const uint64_t tiles = (n + tile_rows - 1) / tile_rows;
        dim3 cgrid((unsigned)std::min<uint64_t>(tiles, 256ull * 8));
        if (wide16) {
            // the tiles are dealt out by workgroup number: a grid larger than what is resident would run its last workgroups -- with all
            // their tiles -- behind the others (82 registers: five workgroups per CU, not eight)
            static std::atomic<int> resident16_dev[64];  // per device: CU count and partition mode may differ between the GPUs of a process
            std::atomic<int>& resident16 = resident16_dev[h->device & 63];
            int r = resident16.load(std::memory_order_relaxed);
            if (!r) {
                int per_cu = 0, cus = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_color_rows_bm16, 256, 0) != hipSuccess || per_cu < 1) { per_cu = 4; (void)hipGetLastError(); }
                if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus < 1) { cus = 256; (void)hipGetLastError(); }
                r = per_cu * cus;
                resident16.store(r, std::memory_order_relaxed);
            }
            cgrid = dim3((unsigned)std::min<uint64_t>((tiles + 3) / 4, (uint64_t)r));
        }
        if (wide16)
            hipLaunchKernelGGL(k_color_rows_bm16, cgrid, dim3(256), 0, s, d_rowidx, h->d_cs_bm.as<uint8_t>() + CS_BM_SLACK, (rowbytes + 3) & ~3u, n, rowbytes, tile_rows, div_m,
                               div_l, d_out);
        else if (rowbytes >= 4)
            hipLaunchKernelGGL(k_color_rows_bm<true>, cgrid, dim3(256), 0, s, d_rowidx, h->d_cs_bm.as<uint8_t>() + CS_BM_SLACK, (rowbytes + 3) & ~3u, n, rowbytes, tile_rows, div_m,
                               div_l, d_out);
        else
            hipLaunchKernelGGL(k_color_rows_bm<false>, cgrid, dim3(256), 0, s, d_rowidx, h->d_cs_bm.as<uint8_t>() + CS_BM_SLACK, (rowbytes + 3) & ~3u, n, rowbytes, tile_rows, div_m,
                               div_l, d_out);
    }
    else
        hipLaunchKernelGGL(k_color_rows, dim3(grid_for((n + 255) / 256)), dim3(256), 0, s, d_rowidx, h->im.tcol, h->im.cs_off, h->im.cs_ids, h->im.cs_w, n, rowbytes, d_out);
    HIPCK(hipGetLastError());
    return 0;
}

// device-resident colour rows: presence bits + CEIL(nb_genomes/8)-byte bitmap row per k-mer, no synchronisation
extern "C" int bft_gpu_query_color_rows_dev(bft_gpu* h, const void* d_kmers, uint64_t n, void* d_present_bits, void* d_rows, void* d_scratch_rows_u32,
                                            void* hip_stream) {
    if (!h || ((!d_kmers || !d_present_bits || !d_rows || !d_scratch_rows_u32) && n)) return fail(BFT_GPU_E_ARG, "NULL argument");
    ENTER(h);
    CK(ensure_built(h, false));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    const uint32_t rowbytes = (h->im.nb_genomes + 7) / 8;
    if (n == 0 || rowbytes == 0) return BFT_GPU_OK;
    // with the bitmap dictionary the query kernel writes colour sets straight away (the k-mer hash holds them; bft_hit_out in the
    // walk): no row -> colour set pass
    CK(ensure_cs_bitmaps(h));
    const bool direct = h->has_cs_bm;
    // rows of 16 bytes and up through the k-mer hash: lookup and rows in one launch (the scratch array stays unused)
    if (direct && rowbytes >= 16 && ((uintptr_t)d_rows & 15u) == 0 && h->im.kh_lines != nullptr && !h->opt_walk_hash && bft_kh_has_kernels(h->W, h->im.kh.S)) {
        CK(bft_kh_color_rows(h->im, (const uint8_t*)d_kmers, n, h->B, (uint64_t*)d_present_bits, h->d_cs_bm.as<uint8_t>() + CS_BM_SLACK, (rowbytes + 3) & ~3u, rowbytes,
                             (uint8_t*)d_rows, h->device, s));
        return note_foreign_stream(h, s);
    }
    if (!direct) CK(ensure_table(h));
    if (direct) h->im.emit_cs = 1;
    const int rc = launch_query(h, (const uint8_t*)d_kmers, n, (uint64_t*)d_present_bits, (uint32_t*)d_scratch_rows_u32, s);
    h->im.emit_cs = 0;
    CK(rc);
    CK(launch_color_rows(h, (uint32_t*)d_scratch_rows_u32, n, rowbytes, (uint8_t*)d_rows, s, direct));
    return note_foreign_stream(h, s);
}

extern "C" int bft_gpu_query_color_rows(bft_gpu* h, const uint8_t* kmers, uint64_t n, uint8_t* present_bits, uint8_t* rows) {
    if (!h || !rows || (!kmers && n)) return fail(BFT_GPU_E_ARG, "NULL argument");
    ENTER(h);
    CK(ensure_built(h));
    const uint32_t rowbytes = (h->im.nb_genomes + 7) / 8;
    if (rowbytes == 0) return BFT_GPU_OK;
    const uint64_t chunk = 1ull << 22;
    const uint64_t mc = std::min(n, chunk);
    DevBuf dk, db, dr, dout;
    CK(dk.alloc(mc * h->B));
    CK(db.alloc(((mc + 63) / 64) * 8));
    CK(dr.alloc(mc * 4));
    CK(dout.alloc(mc * rowbytes));
    for (uint64_t a = 0; a < n; a += chunk) {
        const uint64_t m = std::min(chunk, n - a);
        CK(query_rows(h, kmers + a * h->B, m, dk, db, dr, present_bits ? present_bits + a / 8 : nullptr));
        CK(launch_color_rows(h, dr.as<uint32_t>(), m, rowbytes, dout.as<uint8_t>(), h->stream));
        HIPCK(hipMemcpyAsync(rows + a * rowbytes, dout.p, m * rowbytes, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
    }
    return BFT_GPU_OK;
}

// ------------------------------------------------------------------------------------------------
// info / timing / extraction
// ------------------------------------------------------------------------------------------------
// query_sequence (src/bft.c:1241-1351) for a batch of ASCII sequences
// Sequence queries on device-resident input.  Scratch (codes, plan, colour set per position) belongs to the handle and only grows; calls on
// different streams take turns on it.  Nothing here waits for the GPU: the positions of a chunk are counted on the device
// (k_seq_plan + scan) and the kernels read the total from there.
template <int W, bool STAGED, int PROBE>
static int launch_seq_walk_k(bft_gpu* h, uint32_t ns, int canonical, const uint64_t* d_soff, hipStream_t s) {
    size_t lds = BFT_LDS_HM_BYTES + ((size_t)BFT_MODULO_HASH * 8 + 15) / 16 * 16 + BFT_LDS_ROOT_MAX_CC * sizeof(BftCCX);
    static std::atomic<uint64_t> attr_devs{0};
    const uint64_t dev_bit = 1ull << (h->device & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & dev_bit)) {
        HIPCK(hipFuncSetAttribute((const void*)k_seq_walk8<W, STAGED, PROBE>, hipFuncAttributeMaxDynamicSharedMemorySize, 84 << 10));
        HIPCK(hipFuncSetAttribute((const void*)k_seq_walk6<W, STAGED, PROBE>, hipFuncAttributeMaxDynamicSharedMemorySize, 84 << 10));
        attr_devs.fetch_or(dev_bit, std::memory_order_release);
    }
    BftImage im = h->im;
    im.emit_cs = 1;
    if (W != 2)
        hipLaunchKernelGGL((k_seq_walk8<W, STAGED, PROBE>), dim3(512), dim3(1024), lds, s, im, h->sq_codes.as<uint64_t>(), h->sq_bad.as<uint32_t>(), d_soff,
                           h->sq_poff.as<uint64_t>(), h->sq_tile.as<uint32_t>(), ns, canonical, h->sq_cs.as<uint32_t>());
    else
        hipLaunchKernelGGL((k_seq_walk6<W, STAGED, PROBE>), dim3(512), dim3(BFT_BLOCK6), lds, s, im, h->sq_codes.as<uint64_t>(), h->sq_bad.as<uint32_t>(), d_soff,
                           h->sq_poff.as<uint64_t>(), h->sq_tile.as<uint32_t>(), ns, canonical, h->sq_cs.as<uint32_t>());
    HIPCK(hipGetLastError());
    return 0;
}

template <int W>
static int launch_seq_walk_w(bft_gpu* h, uint32_t ns, int canonical, const uint64_t* d_soff, hipStream_t s) {
    const bool staged = h->root_ncc >= 1 && h->root_ncc <= BFT_LDS_ROOT_MAX_CC;
    if (h->im.kh_lines != nullptr) {
        return bft_kh_seq(h->im, h->sq_codes.as<uint64_t>(), h->sq_bad.as<uint32_t>(), d_soff, h->sq_poff.as<uint64_t>(), h->sq_tile.as<uint32_t>(), ns, canonical,
                          h->sq_cs.as<uint32_t>(), claim_counters(h, s, h->opt_query_dynamic_min, h->sq_units), h->opt_query_chunk, s);
        // (claims_launched: the caller, behind the launch)
    }
    return staged ? launch_seq_walk_k<W, true, 0>(h, ns, canonical, d_soff, s) : launch_seq_walk_k<W, false, 0>(h, ns, canonical, d_soff, s);
}

static int query_sequences_core(bft_gpu* h, const char* d_seqs, const uint64_t* d_seq_off, uint64_t n_seqs, uint64_t total_chars, double threshold,
                                int canonical, uint8_t* d_rows, hipStream_t s) {
    const uint32_t G = h->im.nb_genomes, rowbytes = (G + 7) / 8;
    if (rowbytes == 0 || n_seqs == 0) return 0;
    if (h->sq_used && h->sq_stream != s) HIPCK(hipStreamSynchronize(h->sq_stream));
    auto need = [&](DevBuf& b, size_t bytes) -> int {
        if (b.bytes >= bytes) return 0;
        if (h->sq_used) HIPCK(hipStreamSynchronize(s));  // the block being replaced may still be read by the previous call
        return b.alloc(bytes + bytes / 8);
    };
    // sequences per chunk: 32-bit sequence numbers
    const uint64_t chunk = 1ull << 30;
    const uint64_t cmax = std::min(chunk, n_seqs);
    const uint64_t n_cw = (total_chars + 31) / 32;  // code words of the blob (32 characters each)
    const size_t scan_bytes = bft_scan::scratch_bytes(cmax + 1);  // (the scan's tile states)
    CK(need(h->sq_codes, (n_cw + BFT_MAX_W + 2) * 8));
    CK(need(h->sq_bad, (n_cw + BFT_MAX_W + 2) * 4));
    CK(need(h->sq_npos, (cmax + 1) * 8));
    CK(need(h->sq_poff, (cmax + 1) * 8));
    CK(need(h->sq_tmp, scan_bytes));
    CK(need(h->sq_tile, (total_chars / 64 + 2) * 4));
    CK(need(h->sq_cs, (total_chars + 64) * 4));  // colour set of every k-mer position of a chunk (positions <= characters)
    h->sq_used = true;
    h->sq_stream = s;
    // (the slack words behind the codes are read by windows at the very end of the blob: keep them defined)
    CK(bft_zero_async(h->sq_codes.as<uint64_t>() + n_cw, (BFT_MAX_W + 2) * 8, s));
    CK(bft_zero_async(h->sq_bad.as<uint32_t>() + n_cw, (BFT_MAX_W + 2) * 4, s));
    if (n_cw)
        hipLaunchKernelGGL(k_seq_encode, dim3(grid_for((n_cw + 255) / 256)), dim3(256), 0, s, d_seqs, total_chars, n_cw, h->sq_codes.as<uint64_t>(),
                           h->sq_bad.as<uint32_t>());
    for (uint64_t a = 0; a < n_seqs; a += chunk) {
        const uint64_t ns = std::min(chunk, n_seqs - a);
        const uint64_t* soff = d_seq_off + a;
        hipLaunchKernelGGL(k_seq_plan, dim3(grid_for((ns + 256) / 256)), dim3(256), 0, s, soff, ns, h->k, h->sq_npos.as<uint64_t>());
        CK(bft_scan::exclusive_sum_ptr<uint64_t>(h->sq_npos.as<uint64_t>(), h->sq_poff.as<uint64_t>(), ns + 1, s, h->sq_tmp));
        hipLaunchKernelGGL(k_seq_tiles, dim3(256 * 4), dim3(256), 0, s, h->sq_poff.as<uint64_t>(), (uint32_t)ns, h->sq_tile.as<uint32_t>());
        h->sq_units = total_chars / 256 + 2;  // (k-mer positions <= characters: the blocks the kernel can deal out)
        switch (h->W) {
        case 1: CK(launch_seq_walk_w<1>(h, (uint32_t)ns, canonical, soff, s)); break;
        case 2: CK(launch_seq_walk_w<2>(h, (uint32_t)ns, canonical, soff, s)); break;
        case 3: CK(launch_seq_walk_w<3>(h, (uint32_t)ns, canonical, soff, s)); break;
        default: CK(launch_seq_walk_w<4>(h, (uint32_t)ns, canonical, soff, s)); break;
        }
        claims_launched(h, s);
        const uint32_t win = std::min<uint32_t>(SEQ_TALLY_G, (G + 63u) & ~63u);  // counters per wavefront: all genomes up to 2048
        hipLaunchKernelGGL(k_seq_tally, dim3((unsigned)std::min<uint64_t>((ns + SEQ_TALLY_WAVES - 1) / SEQ_TALLY_WAVES, 256ull * 16)), dim3(64 * SEQ_TALLY_WAVES),
                           (size_t)SEQ_TALLY_WAVES * win * 4, s, h->sq_cs.as<uint32_t>(), h->sq_poff.as<uint64_t>(), (uint32_t)ns, h->im.cs_off, h->im.cs_ids, h->im.cs_w, G, rowbytes,
                           threshold, win, d_rows + a * rowbytes);
        HIPCK(hipGetLastError());
    }
    return 0;
}

extern "C" int bft_gpu_query_sequences_dev(bft_gpu* h, const void* d_seqs, const void* d_seq_off, uint64_t n_seqs, uint64_t total_chars, double threshold,
                                           int canonical, void* d_rows, void* hip_stream) {
    if (!h || ((!d_seqs || !d_seq_off || !d_rows) && n_seqs)) return fail(BFT_GPU_E_ARG, "NULL argument");
    if (!(threshold > 0) || threshold > 1) return fail(BFT_GPU_E_ARG, "the threshold must be in (0, 1] (reference src/bft.c:1246-1247)");
    ENTER(h);
    CK(ensure_built(h, false));  // (the k-mer hash answers; the walk fetches the table itself)
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    CK(query_sequences_core(h, (const char*)d_seqs, (const uint64_t*)d_seq_off, n_seqs, total_chars, threshold, canonical, (uint8_t*)d_rows, s));
    return note_foreign_stream(h, s);
}

extern "C" int bft_gpu_query_sequences(bft_gpu* h, const char* seqs, const uint64_t* seq_off, uint64_t n_seqs, double threshold, int canonical,
                                       uint8_t* rows) {
    if (!h || ((!seqs || !seq_off || !rows) && n_seqs)) return fail(BFT_GPU_E_ARG, "NULL argument");
    if (!(threshold > 0) || threshold > 1) return fail(BFT_GPU_E_ARG, "the threshold must be in (0, 1] (reference src/bft.c:1246-1247)");
    ENTER(h);
    CK(ensure_built(h, false));  // (the k-mer hash answers; the walk fetches the table itself)
    const uint32_t G = h->im.nb_genomes, rowbytes = (G + 7) / 8;
    if (rowbytes == 0 || n_seqs == 0) return BFT_GPU_OK;
    // host buffers: the blob and its offsets (rebased to the first sequence) go up in pieces of at most 2^30 characters
    uint64_t a = 0;
    while (a < n_seqs) {
        uint64_t b = a + 1;
        while (b < n_seqs && seq_off[b + 1] - seq_off[a] <= (1ull << 30)) b++;
        const uint64_t ns = b - a, nchars = seq_off[b] - seq_off[a];
        std::vector<uint64_t> soff(ns + 1);
        for (uint64_t i = 0; i <= ns; i++) soff[i] = seq_off[a + i] - seq_off[a];
        DevBuf d_seq, d_soff, d_out;
        CK(d_seq.alloc(nchars + 16));
        CK(d_soff.alloc((ns + 1) * 8));
        CK(d_out.alloc(ns * rowbytes));
        HIPCK(hipMemcpyAsync(d_seq.p, seqs + seq_off[a], nchars, hipMemcpyHostToDevice, h->stream));
        HIPCK(hipMemcpyAsync(d_soff.p, soff.data(), (ns + 1) * 8, hipMemcpyHostToDevice, h->stream));
        CK(query_sequences_core(h, d_seq.as<char>(), d_soff.as<uint64_t>(), ns, nchars, threshold, canonical, d_out.as<uint8_t>(), h->stream));
        HIPCK(hipMemcpyAsync(rows + a * rowbytes, d_out.p, ns * rowbytes, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
        a = b;
    }
    return BFT_GPU_OK;
}

// ------------------------------------------------------------------------------------------------
// .bft files
// ------------------------------------------------------------------------------------------------
extern "C" int bft_gpu_load_bft(const char* path, int device, bft_gpu** out) {
    if (!path || !out) return fail(BFT_GPU_E_ARG, "NULL argument");
    *out = nullptr;
    DeviceScope ds_;
    const double t_load0 = now_ms();
    BftFileContent fc;
    std::string err;
    try {  // sizes come from an untrusted file: an allocation failure is an I/O error of this call, not the end of the process
        if (!bft_file_read(path, fc, err)) return fail(BFT_GPU_E_IO, err);
    } catch (const std::bad_alloc&) {
        return fail(BFT_GPU_E_IO, "out of host memory while reading the .bft file (corrupt size field?)");
    } catch (const std::exception& e) {
        return fail(BFT_GPU_E_IO, std::string("malformed .bft file: ") + e.what());
    }
    bft_gpu* h = nullptr;
    CK(bft_gpu_create_seeded(fc.k, device, fc.r1, fc.r2, &h));
    int rc = 0;
    for (const std::string& g : fc.genomes) h->genomes.push_back(g);
    const size_t B = (size_t)h->B;
    {   // the log for every pair of the file at once (it would otherwise grow by doubling: a copy and a synchronisation per step)
        uint64_t pairs = 0;
        for (const auto& v : fc.per_genome) pairs += v.size() / B;
        if (pairs && pairs <= h->opt_flush_pairs && hipSetDevice(device) == hipSuccess) {
            bft_pool_set_stream(device, h->stream);
            (void)log_reserve(h, pairs);  // (a failure here only means the log grows as usual)
        }
    }
    static const bool io_trace = getenv("BFT_GPU_TRACE_IO") != nullptr;
    const double t_io0 = now_ms();
    if (io_trace) fprintf(stderr, "[bft_gpu io] %8.1f ms load: file decoded, handle created, log reserved\n", t_io0 - t_load0);
    for (size_t g = 0; g < fc.per_genome.size() && rc == 0; g++)
        if (!fc.per_genome[g].empty()) rc = bft_gpu_insert_kmers(h, fc.per_genome[g].data(), fc.per_genome[g].size() / B, (uint32_t)g);
    if (io_trace) fprintf(stderr, "[bft_gpu io] %8.1f ms load: every genome's k-mers inserted (host batches)\n", now_ms() - t_io0);
    if (rc == 0) rc = bft_gpu_build(h);
    if (io_trace) fprintf(stderr, "[bft_gpu io] %8.1f ms load: index built\n", now_ms() - t_io0);
    bft_dispose_async(fc);  // (the decoded k-mers of every genome)
    if (rc != 0) {
        const std::string keep = g_err;
        bft_gpu_free(h);
        return fail(rc, keep);
    }
    *out = h;
    return BFT_GPU_OK;
}

template <class T, class A>
static int download(const DevBuf& d, uint64_t bytes, std::vector<T, A>& v) {
    v.resize(bytes / sizeof(T));
    if (bytes) HIPCK(hipMemcpy(v.data(), d.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int bft_gpu_write_bft(bft_gpu* h, const char* path) {
    if (!h || !path) return fail(BFT_GPU_E_ARG, "NULL argument");
    if (!bft_reference_k(h->k)) return fail(BFT_GPU_E_ARG, "the .bft format requires k % 9 == 0 (reference src/main.c:61-63)");
    ENTER(h);
    static const bool io_trace = getenv("BFT_GPU_TRACE_IO") != nullptr;
    const double t_io0 = now_ms();
    CK(ensure_built(h));
    if (io_trace) fprintf(stderr, "[bft_gpu io] %8.1f ms write: index built, sorted table resident\n", now_ms() - t_io0);
    BftHostImage hi;
    hi.k = h->k;
    hi.r1 = h->r1;
    hi.r2 = h->r2;
    hi.genomes = h->genomes;
    while (hi.genomes.size() < h->im.nb_genomes) hi.genomes.push_back("genome_" + std::to_string(hi.genomes.size()));
    CK(download(h->d_nodes, h->idx_sizes[0], hi.nodes));
    CK(download(h->d_ccs, h->idx_sizes[2], hi.ccs));
    CK(download(h->d_f2w, h->idx_sizes[3], hi.f2w));
    CK(download(h->d_clus, h->idx_sizes[4], hi.clus));
    CK(download(h->d_child, h->idx_sizes[5], hi.child));
    CK(download(h->d_ucrow, h->idx_sizes[7], hi.ucrow));
    CK(download(h->d_tk, h->idx_sizes[8], hi.tk));
    CK(download(h->d_tcol, h->n_kmers * 4, hi.tcol));
    {   // the dictionary: offsets as they are, the ids widened to 32 bits (resident in 1, 2 or 4 bytes) by a few host threads
        CK(download(h->d_cs_off, (h->n_sets + 1) * 4, hi.cs_off));
        hi.cs_ids.resize(h->n_ids);
        if (h->cs_w == 4) {
            if (h->n_ids) HIPCK(hipMemcpy(hi.cs_ids.data(), h->d_cs_ids.p, h->n_ids * 4, hipMemcpyDeviceToHost));
        } else if (h->n_ids) {
            BftBigVec<uint8_t> raw;
            raw.resize(h->n_ids * h->cs_w);
            HIPCK(hipMemcpy(raw.data(), h->d_cs_ids.p, raw.size(), hipMemcpyDeviceToHost));
            const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
            const uint64_t per = (h->n_ids + nt - 1) / nt;
            const uint32_t w = h->cs_w;
            uint32_t* dst = hi.cs_ids.data();
            auto widen = [&raw, dst, w](uint64_t a, uint64_t b) {
                if (w == 1) for (uint64_t i = a; i < b; i++) dst[i] = raw[i];
                else { const uint16_t* r16 = (const uint16_t*)raw.data(); for (uint64_t i = a; i < b; i++) dst[i] = r16[i]; }
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < nt; t++) {
                const uint64_t a = std::min<uint64_t>(h->n_ids, t * per), b = std::min<uint64_t>(h->n_ids, a + per);
                if (a >= b) break;
                try { th.emplace_back(widen, a, b); } catch (...) { widen(a, b); }
            }
            widen(0, std::min<uint64_t>(h->n_ids, per));
            for (std::thread& x : th) x.join();
        }
    }
    if (io_trace) fprintf(stderr, "[bft_gpu io] %8.1f ms write: image arrays on the host\n", now_ms() - t_io0);
    std::string err;
    try {
        if (!bft_file_write(path, hi, err)) return fail(BFT_GPU_E_IO, err);
    } catch (const std::bad_alloc&) {
        return fail(BFT_GPU_E_LIMIT, "out of host memory while serialising the index");
    } catch (const std::exception& e) {
        return fail(BFT_GPU_E_IO, std::string("write_BFT: ") + e.what());
    }
    if (io_trace) fprintf(stderr, "[bft_gpu io] %8.1f ms write: file written and closed\n", now_ms() - t_io0);
    bft_dispose_async(hi);  // (the host copy of the image: gigabytes of vectors)
    return BFT_GPU_OK;
}

// ------------------------------------------------------------------------------------------------
// image replication (one contiguous device blob: header, then 256-byte aligned sections)
// ------------------------------------------------------------------------------------------------
namespace {
constexpr uint64_t BLOB_MAGIC = 0x3430555047544642ull;  // "BFTGPU04"
constexpr int BLOB_HDR_WORDS = 64, BLOB_SECTIONS = 13;
enum { H_MAGIC, H_TOTAL, H_K, H_R1, H_R2, H_NKMERS, H_NPAIRS, H_NSETS, H_NIDS, H_ROOTNCC, H_MAXGID, H_ANYINS, H_NBGEN, H_NNAMES,
       H_CSW, H_STOREANY, H_INFO = 16, H_IDX = 32, H_SEC = 41 };

struct BlobPlan {
    uint64_t hdr[BLOB_HDR_WORDS];
    uint64_t off[BLOB_SECTIONS];
    const void* src[BLOB_SECTIONS];
    std::string names;
};

inline uint64_t align256(uint64_t v) { return (v + 255ull) & ~255ull; }

void plan_blob(bft_gpu* h, BlobPlan& p) {
    memset(p.hdr, 0, sizeof(p.hdr));
    p.names.clear();
    for (const std::string& g : h->genomes) { p.names += g; p.names.push_back('\0'); }
    const uint64_t sz[BLOB_SECTIONS] = {h->idx_sizes[0], h->idx_sizes[1], h->idx_sizes[2], h->idx_sizes[3], h->idx_sizes[4], h->idx_sizes[5], h->idx_sizes[6],
                                        h->idx_sizes[7], h->idx_sizes[8], h->n_kmers * 4, (h->n_sets + 1) * 4, h->n_ids * (uint64_t)h->cs_w,
                                        p.names.size()};
    const void* src[BLOB_SECTIONS] = {h->d_nodes.p, h->d_bfT.p, h->d_ccs.p, h->d_f2w.p, h->d_clus.p, h->d_child.p, h->d_uck.p, h->d_ucrow.p,
                                      h->d_tk.p, h->d_tcol.p, h->d_cs_off.p, h->d_cs_ids.p, nullptr};
    uint64_t o = BLOB_HDR_WORDS * 8;
    for (int i = 0; i < BLOB_SECTIONS; i++) {
        p.off[i] = o;
        p.src[i] = src[i];
        p.hdr[H_SEC + i] = sz[i];
        o = align256(o + sz[i]);
    }
    uint64_t* H = p.hdr;
    H[H_MAGIC] = BLOB_MAGIC; H[H_TOTAL] = o; H[H_K] = h->k; H[H_R1] = h->r1; H[H_R2] = h->r2; H[H_NKMERS] = h->n_kmers; H[H_NPAIRS] = h->n_pairs;
    H[H_NSETS] = h->n_sets; H[H_NIDS] = h->n_ids; H[H_ROOTNCC] = h->root_ncc; H[H_MAXGID] = h->max_gid_seen; H[H_ANYINS] = h->any_insert;
    H[H_NBGEN] = h->im.nb_genomes; H[H_NNAMES] = h->genomes.size(); H[H_CSW] = h->cs_w;  // (word 14: bytes per dictionary id)
    for (int i = 0; i < 16; i++) H[H_INFO + i] = h->info[i];
    for (int i = 0; i < 9; i++) H[H_IDX + i] = h->idx_sizes[i];
}
}  // namespace

extern "C" int bft_gpu_image_size(bft_gpu* h, uint64_t* nbytes) {
    if (!h || !nbytes) return fail(BFT_GPU_E_ARG, "NULL argument");
    ENTER(h);
    CK(ensure_built(h, false));
    BlobPlan p;
    plan_blob(h, p);
    *nbytes = p.hdr[H_TOTAL];
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_image_pack(bft_gpu* h, void* d_blob, uint64_t cap, void* hip_stream) {
    if (!h || !d_blob) return fail(BFT_GPU_E_ARG, "NULL argument");
    ENTER(h);
    CK(ensure_built(h));
    BlobPlan p;
    plan_blob(h, p);
    if (cap < p.hdr[H_TOTAL]) return fail(BFT_GPU_E_NOSPACE, "blob buffer too small");
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    uint8_t* d = (uint8_t*)d_blob;
    HIPCK(hipMemcpyAsync(d, p.hdr, sizeof(p.hdr), hipMemcpyHostToDevice, s));
    for (int i = 0; i < BLOB_SECTIONS; i++) {
        const uint64_t n = p.hdr[H_SEC + i];
        if (!n) continue;
        if (i != 12) HIPCK(hipMemcpyAsync(d + p.off[i], p.src[i], n, hipMemcpyDeviceToDevice, s));
        else HIPCK(hipMemcpyAsync(d + p.off[i], p.names.data(), n, hipMemcpyHostToDevice, s));  // section 12: the genome names (host)
    }
    HIPCK(hipStreamSynchronize(s));  // the header and the names are host temporaries
    drop_table(h);  // ("compact_table": ensure_built brought the sorted table back for the blob; the source is a group member like the others)
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_image_unpack(const void* d_blob, uint64_t nbytes, int device, bft_gpu** out) {
    if (!d_blob || !out) return fail(BFT_GPU_E_ARG, "NULL argument");
    *out = nullptr;
    DeviceScope ds_;
    if (nbytes < BLOB_HDR_WORDS * 8) return fail(BFT_GPU_E_ARG, "blob shorter than its header");
    int ndev = 0;
    HIPCK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(BFT_GPU_E_ARG, "bad device index");
    HIPCK(hipSetDevice(device));
    uint64_t H[BLOB_HDR_WORDS];
    HIPCK(hipMemcpy(H, d_blob, sizeof(H), hipMemcpyDeviceToHost));
    if (H[H_MAGIC] != BLOB_MAGIC) return fail(BFT_GPU_E_IO, "not a bft_gpu image blob");
    if (H[H_TOTAL] > nbytes) return fail(BFT_GPU_E_IO, "truncated image blob");
    uint64_t off[BLOB_SECTIONS], o = BLOB_HDR_WORDS * 8;
    for (int i = 0; i < BLOB_SECTIONS; i++) {
        off[i] = o;
        if (H[H_SEC + i] > H[H_TOTAL] - o) return fail(BFT_GPU_E_IO, "image blob section out of range");
        o = align256(o + H[H_SEC + i]);
    }
    bft_gpu* h = nullptr;
    CK(bft_gpu_create_seeded((int)H[H_K], device, (int)H[H_R1], (int)H[H_R2], &h));
    const uint8_t* d = (const uint8_t*)d_blob;
    DevBuf* dst[BLOB_SECTIONS] = {&h->d_nodes, &h->d_bfT, &h->d_ccs, &h->d_f2w, &h->d_clus, &h->d_child, &h->d_uck, &h->d_ucrow,
                                  &h->d_tk, &h->d_tcol, &h->d_cs_off, &h->d_cs_ids, nullptr};
    int rc = 0;
    for (int i = 0; i < BLOB_SECTIONS && rc == 0; i++) {
        if (i == 12) continue;  // names: below
        rc = dst[i]->alloc(H[H_SEC + i]);
        if (rc == 0 && H[H_SEC + i] &&
            hipMemcpyAsync(dst[i]->p, d + off[i], H[H_SEC + i], hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
            rc = fail(BFT_GPU_E_HIP, "image blob copy failed");
    }
    std::string names(H[H_SEC + 12], '\0');
    if (rc == 0 && !names.empty() && hipMemcpyAsync(&names[0], d + off[12], names.size(), hipMemcpyDeviceToHost, h->stream) != hipSuccess)
        rc = fail(BFT_GPU_E_HIP, "image blob copy failed");
    if (rc == 0 && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(BFT_GPU_E_HIP, "image blob copy failed");
    if (rc == 0) {
        size_t a = 0;
        for (uint64_t i = 0; i < H[H_NNAMES] && a < names.size(); i++) {
            const size_t e = names.find('\0', a);
            h->genomes.push_back(names.substr(a, e == std::string::npos ? std::string::npos : e - a));
            a = e == std::string::npos ? names.size() : e + 1;
        }
        h->n_kmers = H[H_NKMERS]; h->n_pairs = H[H_NPAIRS]; h->n_sets = H[H_NSETS]; h->n_ids = H[H_NIDS];
        h->root_ncc = (uint32_t)H[H_ROOTNCC]; h->max_gid_seen = (uint32_t)H[H_MAXGID]; h->any_insert = H[H_ANYINS] != 0;
        h->cs_w = (uint32_t)H[H_CSW];
        if (h->cs_w != 1 && h->cs_w != 2 && h->cs_w != 4) rc = fail(BFT_GPU_E_IO, "image blob: bad dictionary id width");
        for (int i = 0; i < 16; i++) h->info[i] = H[H_INFO + i];
        for (int i = 0; i < 9; i++) h->idx_sizes[i] = H[H_IDX + i];
        h->cs_on_host = false;
        if (rc == 0) rc = bind_image(h, (uint32_t)H[H_NBGEN]);
    }
    if (rc != 0) {
        const std::string keep = g_err;
        bft_gpu_free(h);
        return fail(rc, keep);
    }
    h->info[12] = image_bytes(h);
    h->built = true;
    ensure_claim_counters(h);
    h->table_dropped = false;
    drop_table(h);  // ("compact_table", the default: the replica's k-mer hash was derived from the blob's sorted table, which need not stay)
    *out = h;
    return BFT_GPU_OK;
}

// Test hook (tests/test_gpu_build.py): raw copy of one index array of the image.
extern "C" int bft_gpu_debug_get_array(bft_gpu* h, const char* name, void* out, uint64_t cap_bytes, uint64_t* nbytes) {
    if (!h || !name) return fail(BFT_GPU_E_ARG, "NULL argument");
    ENTER(h);
    CK(ensure_built(h));
    static const char* names[16] = {"nodes", "bfT", "ccs", "f2w", "clus", "child", "uck", "ucrow", "tk", "ccx", "f18", "fent", "kh", "tcol", "kh_ovf_k", "kh_ovf_v"};
    const DevBuf* bufs[16] = {&h->d_nodes, &h->d_bfT, &h->d_ccs, &h->d_f2w, &h->d_clus, &h->d_child, &h->d_uck, &h->d_ucrow, &h->d_tk,
                              &h->d_ccx, &h->d_f18, &h->d_fent, &h->d_kh, &h->d_tcol, &h->d_kh_ovf_k, &h->d_kh_ovf_v};
    const uint64_t derived[7] = {h->idx_sizes[2] / sizeof(BftCC) * sizeof(BftCCX), h->n_f18 * 8, h->n_fent * 8,
                                 h->kh_lines ? (h->kh_lines + BFT_KH_TAIL_LINES) * BFT_KH_LINE_WORDS * 8 : 0ull, h->n_kmers * 4,
                                 (uint64_t)h->kh_ovf_n * h->W * 8, (uint64_t)h->kh_ovf_n * 4};
    for (int i = 0; i < 16; i++)
        if (std::string(name) == names[i]) {
            const uint64_t sz = i < 9 ? h->idx_sizes[i] : derived[i - 9];
            if (nbytes) *nbytes = sz;
            if (!out) return BFT_GPU_OK;
            if (cap_bytes < sz) return fail(BFT_GPU_E_NOSPACE, "buffer too small");
            if (sz) HIPCK(hipMemcpy(out, bufs[i]->p, sz, hipMemcpyDeviceToHost));
            return BFT_GPU_OK;
        }
    return fail(BFT_GPU_E_ARG, "unknown array");
}

// ------------------------------------------------------------------------------------------------
// test hooks (tests/test_gpu_sort_scan.py; not part of include/bft_gpu.h): the library's own sort and scan on the caller's device arrays
// ------------------------------------------------------------------------------------------------
// kind 0: u64 keys; 1: u64 keys + u32 values; 2: u32 keys + u32 values.  shape: bft_rs::SHAPE_*.  Stable LSD over the bits [begin_bit, end_bit).
extern "C" int bft_gpu_test_sort(int kind, int shape, const void* d_keys, const void* d_vals, uint64_t n, unsigned begin_bit, unsigned end_bit, void* d_out_keys, void* d_out_vals,
                                 void* hip_stream) {
    if ((!d_keys || !d_out_keys) && n) return fail(BFT_GPU_E_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)hip_stream;
    int dev = 0;
    HIPCK(hipGetDevice(&dev));
    bft_pool_set_stream(dev, s);
    int rc = BFT_GPU_E_ARG;
#define BFT_TS(K, V, SH) rc = bft_rs::sort_pairs<K, V, SH>((const K*)d_keys, (const V*)d_vals, n, (K*)d_out_keys, (V*)d_out_vals, begin_bit, end_bit, s)
    if (kind == 0) {
        rc = bft_rs::sort_keys<uint64_t>((const uint64_t*)d_keys, n, (uint64_t*)d_out_keys, begin_bit, end_bit, s);
    } else if (kind == 1) {
        if (shape == bft_rs::SHAPE_LIGHT) BFT_TS(uint64_t, uint32_t, bft_rs::SHAPE_LIGHT);
        else if (shape == bft_rs::SHAPE_BACK) BFT_TS(uint64_t, uint32_t, bft_rs::SHAPE_BACK);
        else BFT_TS(uint64_t, uint32_t, bft_rs::SHAPE_BIG);
    } else if (kind == 2) {
        if (shape == bft_rs::SHAPE_LIGHT) BFT_TS(uint32_t, uint32_t, bft_rs::SHAPE_LIGHT);
        else if (shape == bft_rs::SHAPE_BACK) BFT_TS(uint32_t, uint32_t, bft_rs::SHAPE_BACK);
        else BFT_TS(uint32_t, uint32_t, bft_rs::SHAPE_BIG);
    }
#undef BFT_TS
    if (rc) return rc;
    HIPCK(hipStreamSynchronize(s));
    return BFT_GPU_OK;
}
// kind 0: exclusive sum of u32 (out[n] = the total as well); 1: exclusive sum of u64 (likewise); 2: inclusive max of u64, init 5.  *d_total: the grand total.
// reps scans in a row share one scratch block (nothing is zeroed between them: bft_scan.h).
extern "C" int bft_gpu_test_scan(int kind, const void* d_in, uint64_t n, void* d_out, void* d_total, int reps, void* hip_stream) {
    if ((!d_in || !d_out) && n) return fail(BFT_GPU_E_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)hip_stream;
    int dev = 0;
    HIPCK(hipGetDevice(&dev));
    bft_pool_set_stream(dev, s);
    DevBuf scratch;
    for (int r = 0; r < std::max(1, reps); r++) {
        if (kind == 0) CK(bft_scan::exclusive_sum_ptr<uint32_t>((const uint32_t*)d_in, (uint32_t*)d_out, n, s, scratch, (unsigned long long*)d_total, true));
        else if (kind == 1) CK(bft_scan::exclusive_sum_ptr<uint64_t>((const uint64_t*)d_in, (uint64_t*)d_out, n, s, scratch, (unsigned long long*)d_total, true));
        else if (kind == 2)
            CK((bft_scan::scan<uint64_t, bft_scan::PtrIn<uint64_t>, bft_scan::Max, true>(bft_scan::PtrIn<uint64_t>{(const uint64_t*)d_in}, (uint64_t*)d_out, n, 5ull, bft_scan::Max(), s, scratch,
                                                                                          (unsigned long long*)d_total)));
        else return fail(BFT_GPU_E_ARG, "bad scan kind");
    }
    HIPCK(hipStreamSynchronize(s));
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_set_option(bft_gpu* h, const char* name, int64_t value) {
    if (!h || !name) return fail(BFT_GPU_E_ARG, "NULL argument");
    const std::string nm(name);
    if (nm == "query_wgs_per_cu") {
        if (value < 0 || value > 3) return fail(BFT_GPU_E_ARG, "query_wgs_per_cu must be 0 (automatic), 1, 2 or 3");
        h->opt_wgs_per_cu = (int)value;
    } else if (nm == "build_composite") {
        if (value != 0 && value != 1) return fail(BFT_GPU_E_ARG, "build_composite must be 0 or 1");
        h->opt_no_composite = value == 0;
    } else if (nm == "build_msd") {
        if (value < 0 || value > 2) return fail(BFT_GPU_E_ARG, "build_msd must be 0, 1 or 2");
        h->opt_msd = (int)value;
    } else if (nm == "sort_ballots") {  // how the library's radix sort ranks the keys of a wavefront (bft_sort.h); process-wide
        if (value != 0 && value != 1) return fail(BFT_GPU_E_ARG, "sort_ballots must be 0 (LDS atomics if the device serves them in lane order: checked once) or 1 (wavefront ballots)");
        bft_rs::g_bft_rs_rank_mode = value ? 1 : -1;
    } else if (nm == "reserve_pairs") {
        // room for this many not-yet-built (k-mer, genome) pairs in the insertion log, so that a long series of insertKmers
        // batches never re-allocates it (a caller usually knows the total: line 2 of a kmers_comp file, README.md:166-170)
        if (value < 0 || (uint64_t)value > h->opt_flush_pairs) return fail(BFT_GPU_E_ARG, "reserve_pairs must be in [0, flush_pairs]");
        ENTER(h);
        CK(log_reserve(h, (uint64_t)value));
    } else if (nm == "flush_pairs") {  // the insertion log is merged into the index before it holds this many pairs (default 2^30; a test hook below that)
        if (value < 1024 || value > (1ll << 30)) return fail(BFT_GPU_E_ARG, "flush_pairs must be in [1024, 2^30]");
        h->opt_flush_pairs = (uint64_t)value;
    } else if (nm == "query_probe") {
        if (value != 0 && value != 4 && value != 8) return fail(BFT_GPU_E_ARG, "query_probe must be 0 (automatic), 4 or 8");
        h->opt_probe = (int)value;
        h->im.probe_big = probe_mode(h->opt_probe ? h->opt_probe : h->tuned_probe);
    } else if (nm == "node_hash") {  // 1 (default): levels below the root through the node prefix hash; 0: through the containers
        if (value < 0 || value > 2) return fail(BFT_GPU_E_ARG, "node_hash must be 0, 1 or 2");
        h->opt_node_hash = (int)value;
        if (h->built) {
            ENTER(h);
            CK(wait_foreign_stream(h));
            HIPCK(hipStreamSynchronize(h->stream));
            derive_node_hash(h);
            h->info[12] = image_bytes(h);
        }
    } else if (nm == "kmer_hash" || nm == "kmer_hash_load") {  // the k-mer hash: on / off, and the occupancy of its home lines in per cent
        if (nm == "kmer_hash_load") {
            if (value < 10 || value > 80) return fail(BFT_GPU_E_ARG, "kmer_hash_load must be in [10, 80] (per cent)");
            h->opt_kh_load = (uint32_t)value;
        } else
            h->opt_kmer_hash = value != 0;
        if (h->built) {
            ENTER(h);
            CK(wait_foreign_stream(h));
            HIPCK(hipStreamSynchronize(h->stream));
            CK(ensure_table(h));
            derive_kmer_hash(h);
            sync_walk_kh(h);
            derive_node_hash(h);  // (by default the node prefix hash exists exactly when the container walk answers queries)
            h->info[12] = image_bytes(h);
            drop_table(h);
        }
    } else if (nm == "walk_hash") {  // 1: presence / colour queries through the container walk, plain root groups looked up in their regions of the k-mer hash
        h->opt_walk_hash = value != 0;
        if (h->built) {
            ENTER(h);
            CK(wait_foreign_stream(h));
            HIPCK(hipStreamSynchronize(h->stream));
            derive_node_hash(h);
            h->info[12] = image_bytes(h);
        }
    } else if (nm == "compact_table") {  // 1: the sorted k-mer table and the colour set per k-mer do not stay resident beside the k-mer hash (ensure_table)
        h->opt_compact = value != 0;
        if (h->built) {
            ENTER(h);
            CK(wait_foreign_stream(h));
            HIPCK(hipStreamSynchronize(h->stream));
            if (h->opt_compact) drop_table(h);
            else CK(ensure_table(h));
        }
    } else if (nm == "root_direct") {  // 2 (default): root level through the derived range + direct tables; 1: direct table only; 0: containers
        if (value < 0 || value > 3) return fail(BFT_GPU_E_ARG, "root_direct must be 0, 1, 2 or 3");
        h->opt_root_direct = (int)value;
        if (h->built) {
            ENTER(h);
            CK(wait_foreign_stream(h));
            HIPCK(hipStreamSynchronize(h->stream));
            CK(ensure_table(h));  // ("compact_table": k_root_ranges reads the sorted table)
            derive_root_direct(h);
            sync_walk_kh(h);
            default_launch_shape(h);
            h->info[12] = image_bytes(h);
            drop_table(h);
        }
    } else if (nm == "root_quartiles") {  // 1 (default): plain root groups are searched quarter by quarter (BFT_RQ_*)
        h->opt_root_quartiles = value != 0;
        if (h->built) {
            ENTER(h);
            CK(wait_foreign_stream(h));
            HIPCK(hipStreamSynchronize(h->stream));
            CK(ensure_table(h));
            derive_root_direct(h);
            sync_walk_kh(h);
            default_launch_shape(h);
            h->info[12] = image_bytes(h);
            drop_table(h);
        }
    } else if (nm == "tune") {  // measure the launch shape of the container walk on the current image (synchronises)
        if (value != 0 && h->built) {
            ENTER(h);
            CK(wait_foreign_stream(h));
            HIPCK(hipStreamSynchronize(h->stream));
            CK(ensure_table(h));
            CK(tune_residency(h));
        }
    } else if (nm == "query_dynamic") {  // 0: the k-mer hash kernels split their batch by workgroup number (what they did before the claims)
        h->opt_query_dynamic = value != 0;
    } else if (nm == "query_dynamic_min") {
        h->opt_query_dynamic_min = value;
    } else if (nm == "query_chunk") {
        if (value < 1 || value > 64) return fail(BFT_GPU_E_ARG, "query_chunk must be in [1,64]");
        h->opt_query_chunk = (uint32_t)value;
    } else if (nm == "query_grid_mult") {
        if (value < 1 || value > 64) return fail(BFT_GPU_E_ARG, "query_grid_mult must be in [1,64]");
        h->opt_grid_mult = (int)value;
#if defined(BFT_PERF_PROBE)
    } else if (nm == "debug_stop") {  // libbft_gpu_probe.so only (make probe): truncates the walk, results are wrong
        h->im.debug_stop = (uint32_t)value;
#endif
    } else if (nm == "test_front_rank_mode") {  // test hook (tests/test_gpu_build.py): how the root-prefix buckets rank their digits, see k_bucket_sort
        if (value > 2) return fail(BFT_GPU_E_ARG, "test_front_rank_mode must be 0, 1 or 2");
        bft_test_front_rank_mode((int)value);
    } else if (nm == "test_weak_signature") {  // test hook (tests/test_gpu_build.py): colour-set signatures that collide, see bft_intern_colors_gpu
        bft_test_weak_signature(value != 0);
    } else if (nm == "inject_build_failure") {  // test hook (tests/test_gpu_build.py): exercises the all-or-nothing build
        h->inject_build_failure = value != 0;
    } else if (nm == "timing") {
        h->timing = value != 0;
    } else if (nm == "test_stale_claims") {  // test hook (tests/test_gpu_parity.py): every claim counter as a launch that never finished can leave it --
        // anywhere up to the end of the range it was given, i.e. the next launch's base (1: one below it, 2: exactly there, 3: where it started)
        if (value < 1 || value > 3) return fail(BFT_GPU_E_ARG, "test_stale_claims must be 1, 2 or 3");
        ENTER(h);
        if (h->kh_ctr) {
            HIPCK(hipDeviceSynchronize());
            unsigned long long v[bft_gpu::KH_CTR_SLOTS];
            for (int i = 0; i < bft_gpu::KH_CTR_SLOTS; i++) v[i] = value == 3 ? 0ull : h->kh_ctr_base[i] - (value == 1 && h->kh_ctr_base[i] ? 1ull : 0ull);
            HIPCK(hipMemcpy(h->kh_ctr, v, sizeof(v), hipMemcpyHostToDevice));
        }
    } else if (nm == "composite_log") {  // 1 (default): one-word keys with room for an id are logged as composites; 0: always k-mers + ids (test hook: same image)
        if (h->log_n) return fail(BFT_GPU_E_STATE, "composite_log: the insertion log is not empty");
        h->opt_comp_log = value != 0;
        if (h->log_cap) { h->log_k.release(); h->log_g.release(); h->log_cap = 0; }  // (a reserved, empty log: its format is decided again)
    } else if (nm == "build_stages") {
        h->opt_build_stages = value != 0;
    } else if (nm == "flat_min") {
        if (value < 1 || value > 65536) return fail(BFT_GPU_E_ARG, "flat_min must be in [1,65536]");
        h->opt_flat_min = (uint32_t)value;
        if (h->built) {  // re-derive the flat arrays of the current image
            ENTER(h);
            CK(wait_foreign_stream(h));
            HIPCK(hipStreamSynchronize(h->stream));
            CK(ensure_table(h));  // ("compact_table": the derived tables are rebuilt from the sorted table)
            CK(bind_image(h, h->im.nb_genomes));
            h->info[12] = image_bytes(h);
            drop_table(h);
        }
    } else
        return fail(BFT_GPU_E_ARG, "unknown option");
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_info(bft_gpu* h, uint64_t* out, int n_out) {
    if (!h || !out) return fail(BFT_GPU_E_ARG, "NULL argument");
    h->info[0] = h->k;
    h->info[15] = h->log_n;
    for (int i = 0; i < n_out && i < 16; i++) out[i] = h->info[i];
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_footprint(bft_gpu* h, uint64_t* out, int n_out) {
    if (!h || !out) return fail(BFT_GPU_E_ARG, "NULL argument");
    const uint64_t v[12] = {h->d_tk.bytes, h->d_tcol.bytes, h->d_cs_off.bytes + h->d_cs_ids.bytes,
                            h->d_nodes.bytes + h->d_bfT.bytes + h->d_ccs.bytes + h->d_f2w.bytes + h->d_clus.bytes + h->d_child.bytes + h->d_uck.bytes + h->d_ucrow.bytes,
                            h->d_ccx.bytes + h->d_f18.bytes + h->d_fent.bytes, h->d_rdir.bytes + h->d_rstart.bytes + h->d_rq.bytes, h->d_nph.bytes, h->d_kh.bytes + h->d_rspec.bytes + h->d_kh_ovf_k.bytes + h->d_kh_ovf_v.bytes, h->d_cs_bm.bytes,
                            h->d_hashmod.bytes, 0ull, h->log_k.bytes + h->log_g.bytes};
    for (int i = 0; i < n_out && i < 12; i++) out[i] = v[i];
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_kernel_time(bft_gpu* h, double* ms, uint64_t* launches, int reset) {
    if (!h) return fail(BFT_GPU_E_ARG, "NULL handle");
    ENTER(h);
    drain_events(h);
    h->timing = true;  // from the first call on, query launches are bracketed by (pooled) events
    if (ms) *ms = h->kernel_ms;
    if (launches) *launches = h->kernel_launches;
    if (reset) {
        h->kernel_ms = 0;
        h->kernel_launches = 0;
    }
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_build_time(bft_gpu* h, double* ms, int n_out) {
    if (!h || !ms) return fail(BFT_GPU_E_ARG, "NULL argument");
    const double v[25] = {h->build_ms[0], h->build_ms[1], h->build_ms[2], h->build_ms[3], h->build_ms[4], (double)query_residency(h), h->tune_ms[0], h->tune_ms[1],
                          h->im.probe_big ? 8.0 : 4.0, (double)h->kh_lines, h->kh_ms, (double)h->msd_max_bucket, (double)bft_test_exact_passes(), g_malloc_ms,
                          (double)(h->im.rdir ? (h->im.rstart ? 2 : 1) : 0), h->rstart_tune_ms[0], h->rstart_tune_ms[1],
                          (double)h->nph_inserted, (double)h->nph_dropped, h->tune_ms[2], (double)h->claims_static_launches,
                          (double)h->im.kh.S, (double)h->im.kh.db, (double)h->im.kh.maxd, (double)h->kh_ovf_n};
    for (int i = 0; i < n_out && i < 25; i++) ms[i] = v[i];
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_build_stages(bft_gpu* h, char* names, uint32_t names_cap, double* ms, double* bytes, int cap, int* n_out) {
    if (!h || !n_out) return fail(BFT_GPU_E_ARG, "NULL argument");
    *n_out = (int)h->stages.size();
    std::string all;
    for (const auto& st : h->stages) { all += st.name; all += '\n'; }
    if (names) {
        if (all.size() + 1 > names_cap) return fail(BFT_GPU_E_NOSPACE, "names buffer too small");
        memcpy(names, all.c_str(), all.size() + 1);
    }
    for (int i = 0; i < cap && i < (int)h->stages.size(); i++) {
        if (ms) ms[i] = h->stages[i].ms;
        if (bytes) bytes[i] = h->stages[i].bytes;
    }
    return BFT_GPU_OK;
}

// stored T-form rows -> packed k-mers (the reference's layout), on the GPU: bft_gpu_extract copies bytes, not keys
template <int W>
__global__ void k_tform_to_packed(const uint64_t* __restrict__ tk, uint64_t n, int k, int B, uint8_t* __restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t t[W], x[W];
#pragma unroll
        for (int w = 0; w < W; w++) t[w] = tk[i * W + w];
        bft_x_from_tform<W>(t, k, x);
        for (int b = 0; b < B; b++) {
            uint64_t v = 0;
#pragma unroll
            for (int w = 0; w < W; w++)
                if (w == (b >> 3)) v = x[w];
            out[i * B + b] = (uint8_t)(v >> (8 * (b & 7)));
        }
    }
}

extern "C" int bft_gpu_extract(bft_gpu* h, uint8_t* kmers_out, uint32_t* colorset_out, uint64_t cap, uint64_t* n_out) {
    if (!h) return fail(BFT_GPU_E_ARG, "NULL handle");
    ENTER(h);
    CK(ensure_built(h));
    if (n_out) *n_out = h->n_kmers;
    if (!kmers_out && !colorset_out) return BFT_GPU_OK;
    if (cap < h->n_kmers) return fail(BFT_GPU_E_NOSPACE, "extract buffer too small");
    const uint64_t n = h->n_kmers;
    if (kmers_out && n) {
        DevBuf packed;
        CK(packed.alloc(n * h->B));
        const dim3 grid(grid_for((n + 255) / 256)), block(256);
        switch (h->W) {
        case 1: hipLaunchKernelGGL(k_tform_to_packed<1>, grid, block, 0, h->stream, h->d_tk.as<uint64_t>(), n, h->k, h->B, packed.as<uint8_t>()); break;
        case 2: hipLaunchKernelGGL(k_tform_to_packed<2>, grid, block, 0, h->stream, h->d_tk.as<uint64_t>(), n, h->k, h->B, packed.as<uint8_t>()); break;
        case 3: hipLaunchKernelGGL(k_tform_to_packed<3>, grid, block, 0, h->stream, h->d_tk.as<uint64_t>(), n, h->k, h->B, packed.as<uint8_t>()); break;
        default: hipLaunchKernelGGL(k_tform_to_packed<4>, grid, block, 0, h->stream, h->d_tk.as<uint64_t>(), n, h->k, h->B, packed.as<uint8_t>()); break;
        }
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(kmers_out, packed.p, n * h->B, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
    }
    if (colorset_out && n) HIPCK(hipMemcpy(colorset_out, h->d_tcol.p, n * 4, hipMemcpyDeviceToHost));
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_colorset(bft_gpu* h, uint32_t cs, uint32_t* ids, uint32_t cap, uint32_t* n_out) {
    if (!h) return fail(BFT_GPU_E_ARG, "NULL handle");
    ENTER(h);
    CK(ensure_built(h, false));
    CK(host_colorsets(h));
    if ((uint64_t)cs + 1 >= h->cs_off.size()) return fail(BFT_GPU_E_ARG, "unknown colour set");
    const uint32_t a = h->cs_off[cs], b = h->cs_off[cs + 1];
    if (n_out) *n_out = b - a;
    if (ids) {
        if (cap < b - a) return fail(BFT_GPU_E_NOSPACE, "ids buffer too small");
        memcpy(ids, &h->cs_ids[a], (size_t)(b - a) * 4);
    }
    return BFT_GPU_OK;
}

// What the reference keeps in resultPresence (include/Node.h:60-92) for a found k-mer, as indexes instead of host pointers:
// the row of the k-mer in the sorted table (the order of bft_gpu_extract) and the id of its colour set.
extern "C" int bft_gpu_query_rows(bft_gpu* h, const uint8_t* kmers, uint64_t n, uint8_t* present_bits, uint32_t* rows, uint32_t* colorsets) {
    if (!h || (!kmers && n)) return fail(BFT_GPU_E_ARG, "NULL argument");
    ENTER(h);
    CK(ensure_built(h));
    if (n && n <= BFT_PIN_MAX_N) return query_small(h, kmers, n, present_bits, rows, colorsets);
    const uint64_t chunk = 1ull << 24;
    const uint64_t mc = std::min(n, chunk);
    DevBuf dk, db, dr, dc;
    CK(dk.alloc(mc * h->B));
    CK(db.alloc(((mc + 63) / 64) * 8));
    CK(dr.alloc(mc * 4));
    if (colorsets) CK(dc.alloc(mc * 4));
    for (uint64_t a = 0; a < n; a += chunk) {
        const uint64_t m = std::min(chunk, n - a);
        CK(query_rows(h, kmers + a * h->B, m, dk, db, dr, present_bits ? present_bits + a / 8 : nullptr));
        if (rows) HIPCK(hipMemcpyAsync(rows + a, dr.p, m * 4, hipMemcpyDeviceToHost, h->stream));
        if (colorsets) {
            hipLaunchKernelGGL(k_row_colorsets, dim3(grid_for((m + 255) / 256)), dim3(256), 0, h->stream, dr.as<uint32_t>(), h->im.tcol, m, dc.as<uint32_t>());
            HIPCK(hipGetLastError());
            HIPCK(hipMemcpyAsync(colorsets + a, dc.p, m * 4, hipMemcpyDeviceToHost, h->stream));
        }
        HIPCK(hipStreamSynchronize(h->stream));
    }
    return BFT_GPU_OK;
}

// The colour set `cs` in the reference's annotation bytes (smallest of modes 0/1/2, compute_best_mode,
// src/annotation.c:634-650) -- what get_annotation hands out as BFT_annotation::annot (src/bft.c:363-387).
extern "C" int bft_gpu_colorset_annot(bft_gpu* h, uint32_t cs, uint8_t* annot, uint32_t cap, uint32_t* n_out) {
    if (!h) return fail(BFT_GPU_E_ARG, "NULL handle");
    ENTER(h);
    CK(ensure_built(h, false));
    CK(host_colorsets(h));
    if ((uint64_t)cs + 1 >= h->cs_off.size()) return fail(BFT_GPU_E_ARG, "unknown colour set");
    const uint32_t a = h->cs_off[cs], b = h->cs_off[cs + 1];
    std::vector<uint8_t> enc;
    bft_annot_encode(b > a ? &h->cs_ids[a] : nullptr, b - a, enc);
    if (n_out) *n_out = (uint32_t)enc.size();
    if (annot) {
        if (cap < enc.size()) return fail(BFT_GPU_E_NOSPACE, "annotation buffer too small");
        memcpy(annot, enc.data(), enc.size());
    }
    return BFT_GPU_OK;
}
