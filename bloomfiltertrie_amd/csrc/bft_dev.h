// bft_dev.h -- small HIP host-side helpers shared by the translation units of libbft_gpu.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bft_gpu.h"

int bft_fail(int code, const std::string& msg);  // records the thread's last error, returns code

#define HIPCK(expr)                                                                                           \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return bft_fail(BFT_GPU_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define CK(expr)                  \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != 0) return rc_; \
    } while (0)

// Device-memory cache behind DevBuf (bft_gpu.hip).  hipFree synchronises the device and costs ~0.1 ms per call on
// large blocks; a bulk build releases dozens of temporaries.  Released blocks are kept (per device, tagged with the
// stream of the ABI call that released them) and handed out again to requests of a similar size.  A block released
// under another stream is only reused after that stream has drained; bft_gpu_free drops the blocks of its handle.
int bft_pool_alloc(void** p, size_t n, size_t* cap);
void bft_pool_release(void* p, size_t cap);
void bft_pool_set_stream(int device, hipStream_t s);  // the stream of the current ABI call (thread-local)
void bft_pool_drop_stream(hipStream_t s);             // the stream was synchronised and is about to be destroyed: its blocks stay cached, tagged as drained

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;  // requested size
    size_t cap = 0;    // size of the block behind it
    uint32_t tag = 0, tag2 = 0;  // the user's (bft_scan.h: launches so far, states the last one used); zero after every alloc
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), cap(o.cap), tag(o.tag), tag2(o.tag2) { o.p = nullptr; o.bytes = 0; o.cap = 0; o.tag = 0; o.tag2 = 0; }
    ~DevBuf() { release(); }
    void release() {
        if (p) bft_pool_release(p, cap);
        p = nullptr;
        bytes = 0;
        cap = 0;
        tag = 0;
        tag2 = 0;
    }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 8;
        CK(bft_pool_alloc(&p, n, &cap));
        bytes = n;
        return 0;
    }
    int alloc_zero(size_t n, hipStream_t s) {
        CK(alloc(n));
        // (whole 16-byte words: the runtime fills an odd tail with a second kernel)
        HIPCK(hipMemsetAsync(p, 0, std::min(cap, (bytes + 15) & ~(size_t)15), s));
        return 0;
    }
    template <class T>
    T* as() const { return (T*)p; }
    void swap(DevBuf& o) {
        std::swap(p, o.p);
        std::swap(bytes, o.bytes);
        std::swap(cap, o.cap);
        std::swap(tag, o.tag);
        std::swap(tag2, o.tag2);
    }
};

// Counts the host needs (array sizes, format limits) come back through a pinned block that kernels write: the scans and checks
// of a stage are enqueued together and share ONE stream synchronisation (a total fetched by hipMemcpyAsync into pageable memory is
// two staged copies and a synchronisation of its own: ~60 us each, sixteen per trie level).
constexpr int PIN_SLOTS = 32;  // (+ one word behind them: the ticket of bft_pin_wait)
struct PinBlock;
// Waits until everything enqueued on `s` so far has run, by POLLING: a one-thread kernel behind it all writes a ticket into the pinned block and the
// host spins on that word.  hipStreamSynchronize / hipEventSynchronize sleep on an interrupt and come back ~0.1 ms after the stream is through --
// eleven such waits a build (array sizes the host needs) were a millisecond of idle GPU.  The stream's state is looked at every few thousand polls:
// an error there ends the wait with that error.  (bft_assemble.hip)
int bft_pin_wait(PinBlock& pin, hipStream_t s);
// the same in two steps: the ticket enqueued here, waited for later (what is enqueued in between is not waited for; one ticket in flight per block)
int bft_pin_post(PinBlock& pin, hipStream_t s, uint64_t* ticket);
int bft_pin_wait_for(PinBlock& pin, hipStream_t s, uint64_t ticket);
// Zeroes `bytes` bytes (a multiple of 4, 4-byte aligned) at p with a kernel of the library's own.  For every path a caller may record into a HIP graph:
// a hipMemsetAsync node replays correctly ONCE on this runtime (ROCm 7.0.2: the second replay of a captured memset writes garbage --
// tools/probe_graph_memset.py), a kernel node every time.  (bft_assemble.hip)
int bft_zero_async(void* p, size_t bytes, hipStream_t s);
uint64_t bft_pin_next_ticket();  // (for a kernel of the caller's that writes the ticket itself, behind its own results: pin.p[PIN_SLOTS], after __threadfence_system())
struct PinBlock {
    uint64_t* p = nullptr;
    PinBlock() {
        {
            std::lock_guard<std::mutex> lk(mu());
            if (!cache().empty()) { p = cache().back(); cache().pop_back(); }
        }
        if (!p && hipHostMalloc((void**)&p, (PIN_SLOTS + 1) * 8, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { p = nullptr; (void)hipGetLastError(); }
    }
    ~PinBlock() {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu());
        cache().push_back(p);  // (a handful of 256-byte blocks per process, kept; portable: any device of the process may write them)
    }
    static std::mutex& mu() { static std::mutex m; return m; }
    static std::vector<uint64_t*>& cache() { static std::vector<uint64_t*> c; return c; }
};


static inline int bft_grid_for(uint64_t nblk) {
    const uint64_t cap = 256ull * 8ull;  // 256 CUs x 8 resident workgroups of 256 threads
    return (int)std::max<uint64_t>(1, std::min<uint64_t>(nblk, cap));
}

// GPU container assembly and colour interning (bft_assemble.hip)
struct BftDeviceIndex {
    DevBuf nodes, bfT, ccs, f2w, clus, child, uck, ucrow;
    uint64_t n_nodes = 0, n_ccs = 0, n_f2w = 0, n_clus = 0, n_child = 0, n_bf8 = 0, n_uc = 0;
    uint64_t n_child_nodes = 0, n_prefixes = 0, n_ccs_s4 = 0, max_ccs_per_node = 0, root_ncc = 0, root_uc = 0;
};
bool bft_trace_on(void);
void bft_trace_mark(const char* what);  // BFT_GPU_TRACE_BUILD=1 (nullptr: start of a build)
// "build_stages" 1: a stage of the running build ends here on stream s; bytes = what its algorithm reads + writes (0: not a streaming
// stage).  A name that starts with '+' marks work on a side stream, timed from the build's start.  No-op unless the option is on.
void bft_stage(const char* name, double bytes, hipStream_t s);
// after_table_passes: called once, when the first level's passes over the whole sorted table and its CC assignment are done on `s`
struct BftAssembleHook {
    void (*after_table_passes)(void* ctx, hipStream_t s);
    void* ctx;
};
int bft_assemble_gpu(const uint64_t* d_tk, uint64_t n, int k, const uint32_t* d_hashmod, hipStream_t s, BftDeviceIndex& out,
                     const BftAssembleHook* hook = nullptr);
// flat form of the CCs with >= flat_min prefixes (bft_image.h): extended headers, prefix bitmaps + ranks, entries
struct BftCC;
int bft_flatten_gpu(const BftCC* d_ccs, uint64_t n_ccs, const uint64_t* d_f2w, const uint64_t* d_clus, const uint64_t* d_child, uint32_t flat_min,
                    hipStream_t s, DevBuf& ccx, DevBuf& f18, DevBuf& fent, uint64_t& n_f18, uint64_t& n_fent);
void bft_test_weak_signature(bool on);            // test hook: every list's signature = its length (collisions galore)
unsigned long long bft_test_exact_passes(void);   // how many times the interning had to fall back to comparing the lists
// The interning's tail -- the dictionary entries copied out of their representatives' lists, then EVERY k-mer's list compared with the entry it was
// given -- is a third of its time (1.2 of 3.4 ms on config 3) and nothing of the build needs its result before the commit: with `side` set it is
// enqueued on that stream (behind the interning's kernels) and bft_intern_colors_gpu returns at once; the caller keeps d_seg_off / d_pg alive, makes
// whatever reads the dictionary wait for `done`, and calls wait() before it commits -- *collisions != 0: two different lists shared a signature and
// the interning must be run again with exact = true (lists compared; on the caller's stream, nothing deferred).
struct BftInternTail {
    hipStream_t side = nullptr;
    hipEvent_t done = nullptr, ready = nullptr;
    bool pending = false;
    DevBuf rep, bad;  // (what the deferred kernels read and write besides the outputs)
    PinBlock pin;
    uint32_t narrow_w = 4;  // in: bytes per genome id of the resident dictionary (1 / 2: the deferred tail also writes `narrow`, the dictionary's ids in that width)
    DevBuf narrow;
    ~BftInternTail() {
        if (pending && side) (void)hipStreamSynchronize(side);
        if (done) (void)hipEventDestroy(done);
        if (ready) (void)hipEventDestroy(ready);
    }
    int wait(uint32_t* collisions) {
        *collisions = 0;
        if (!pending) return 0;
        HIPCK(hipEventSynchronize(done));
        pending = false;
        *collisions = (uint32_t)pin.p[0];
        return 0;
    }
};
int bft_intern_colors_gpu(const uint32_t* d_seg_off, const uint32_t* d_pg, uint64_t nk, uint64_t np, hipStream_t s, DevBuf& d_tcol,
                          DevBuf& d_cs_off, DevBuf& d_cs_ids, uint64_t& n_sets, uint64_t& n_ids, uint64_t distinct_hint = 0, BftInternTail* tail = nullptr,
                          bool exact = false);

// Merging a sorted run of newly inserted k-mers into the built index (bft_merge.hip).  A "run" = sorted distinct T-form k-mers, a
// colour-set id per k-mer and the dictionary those ids refer to -- what the index itself is made of.
struct BftRun {
    const uint64_t* tk;
    const uint32_t* tcol;
    uint64_t n;
    const uint32_t* cs_off;
    const uint32_t* cs_ids;
    uint64_t n_sets;
};
struct BftRunOut {
    DevBuf tk, tcol, cs_off, cs_ids;
    uint64_t n = 0, n_sets = 0, n_ids = 0;
};
int bft_merge_runs(int W, const BftRun& a, const BftRun& b, hipStream_t s, BftRunOut& out);
int bft_count_pairs(const uint32_t* d_tcol, uint64_t n, const uint32_t* d_cs_off, hipStream_t s, uint64_t* total);

// The build's front end behind the root-prefix split (bft_front.hip): bucket-wise sort of the composites c = T << gb | genome on the
// bits [gb, split_bit) and the de-duplicated outputs: sorted distinct k-mers, offsets of their genome-id lists, the genome ids.
uint32_t bft_front_bucket_capacity(void);
void bft_test_front_rank_mode(int mode);   // test hook: k_bucket_sort's mode (0 atomics + check, 1 ballots only, 2 the check always fails)
// d_vals (vw bytes per id): d_c holds whole T-form k-mers grouped by the bits from split_bit - gb up, the ids beside them
int bft_front_buckets(uint64_t* d_c, uint64_t n, const uint32_t* d_boff, uint32_t nb, uint32_t gb, uint32_t split_bit, hipStream_t s, DevBuf& tk, DevBuf& seg_off,
                      DevBuf& pg, uint64_t& nk, uint64_t& np, const uint32_t* d_max_bucket, uint32_t* max_bucket, bool* done, uint32_t* n_redone,
                      const void* d_vals = nullptr, uint32_t vw = 0);
// (d_max_bucket: the size of the largest bucket, on the device -- it reaches the host, *max_bucket, while the first sort kernel runs;
// *done = false: that bucket is beyond bft_front_bucket_capacity() and nothing was produced (d_c may have been reordered inside its
// buckets); *n_redone: buckets whose order check failed and that were sorted again)
// two-word keys (33 <= k <= 64): d_hk = the top 64 bits of the T-form, left-aligned, grouped by their top 18 bits (d_boff); d_items = {lo: the 2k - 64
// bits below, id: the genome} (12 bytes each) beside them.  *done = false: a bucket is beyond bft_front2_bucket_capacity(): nothing was produced
// (the arrays may have been reordered inside their buckets); *n_redone as bft_front_buckets'.
uint32_t bft_front2_bucket_capacity(void);
int bft_front2_buckets(uint64_t* d_hk, void* d_items, uint64_t n, const uint32_t* d_boff, uint32_t nb, int k, hipStream_t s, DevBuf& tk, DevBuf& seg_off, DevBuf& pg, uint64_t& nk, uint64_t& np,
                       const uint32_t* d_max_bucket, uint32_t* max_bucket, bool* done, uint32_t* n_redone, uint32_t max_gid);
