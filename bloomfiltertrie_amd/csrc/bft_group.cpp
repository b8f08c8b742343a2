// bft_group.cpp -- one index on several GPUs of one process, behind the C-ABI (include/bft_gpu.h, bft_gpu_group_*).
//
// The reference answers -query_kmers / -query_branching with one loop over one BFT_Root (src/file_io.c:651-895, :897-1020).  Queries
// are independent and read-only, so the batched path shards them: the built index is replicated into the HBM of every
// device of the group (bft_gpu_image_pack on the source GPU, one peer copy over xGMI per replica, bft_gpu_image_unpack there), a
// host batch is cut into contiguous slices whose starts are multiples of 64 k-mers (so the per-device presence bitmaps are
// byte ranges of the caller's bitmap), and every member answers its slice on a host thread of its own.  The thread lives as long as the
// group (a batch costs a wake-up, not a thread start) and owns what its member's host batches go through: a stream on the member's device,
// two slots of pinned staging memory and of device buffers.  A slice moves in chunks: the thread copies chunk c + 1 into its pinned slot
// while the GPU answers chunk c (copy in, *_dev entry point, copy out: all on the member's stream) -- eight members do not meet in the
// runtime's one staging path for pageable memory.  Nothing collective is involved: the answers land in the caller's buffers.  (Processes that hold one GPU each -- bench.py under
// torchrun -- replicate with one RCCL broadcast of the same blob instead: bloomfiltertrie_amd/dist.py.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bft_gpu.h"
#include "bft_dev.h"

namespace {

// what one query kind moves per k-mer, besides the packed k-mer itself
struct StageShape {
    uint64_t in_bytes;        // packed k-mer
    bool bits;                // a presence / branching bitmap (one bit per k-mer)
    uint64_t out_bytes;       // bytes per k-mer of the second output (colour row, neighbour counts; 0: none)
    bool scratch;             // a u32 per k-mer of device scratch (colour rows)
};

struct Member {
    bft_gpu* h = nullptr;
    int device = 0;
    // the member's thread: one job at a time, posted by run_sharded
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, finished = false, quit = false;
    int rc = 0;
    std::string msg;
    // staging (touched by the member's thread only, or by the caller's when no thread could be started)
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    void *pin_in[2] = {nullptr, nullptr}, *pin_out[2] = {nullptr, nullptr}, *dev_in[2] = {nullptr, nullptr}, *dev_out[2] = {nullptr, nullptr}, *dev_scr[2] = {nullptr, nullptr};
    size_t pin_in_cap = 0, pin_out_cap = 0, dev_in_cap = 0, dev_out_cap = 0, dev_scr_cap = 0;

    void loop() {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit) return;
            std::function<int()> f = std::move(job);
            has_job = false;
            lk.unlock();
            int r = 0;
            std::string m;
            try {
                r = f();
                if (r != 0) m = bft_gpu_last_error();
            } catch (...) {  // (nothing may leave the thread)
                r = BFT_GPU_E_HIP;
                m = "exception on a group member's thread";
            }
            lk.lock();
            rc = r;
            msg = std::move(m);
            finished = true;
            cv.notify_all();
        }
    }
    void free_staging() {
        (void)hipSetDevice(device);
        if (st) (void)hipStreamSynchronize(st);
        for (int i = 0; i < 2; i++) {
            if (pin_in[i]) (void)hipHostFree(pin_in[i]);
            if (pin_out[i]) (void)hipHostFree(pin_out[i]);
            if (dev_in[i]) (void)hipFree(dev_in[i]);
            if (dev_out[i]) (void)hipFree(dev_out[i]);
            if (dev_scr[i]) (void)hipFree(dev_scr[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            pin_in[i] = pin_out[i] = dev_in[i] = dev_out[i] = dev_scr[i] = nullptr;
            ev[i] = nullptr;
        }
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
        pin_in_cap = pin_out_cap = dev_in_cap = dev_out_cap = dev_scr_cap = 0;
        (void)hipGetLastError();
    }
    // both slots hold at least these many bytes
    int reserve(size_t in_b, size_t out_b, size_t scr_b) {
        HIPCK(hipSetDevice(device));
        if (!st) HIPCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++)
            if (!ev[i]) HIPCK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        auto grow = [&](void** p, size_t& cap, size_t want, bool host) -> int {
            if (want <= cap) return 0;
            HIPCK(hipStreamSynchronize(st));
            for (int i = 0; i < 2; i++) {
                if (p[i]) { if (host) (void)hipHostFree(p[i]); else (void)hipFree(p[i]); p[i] = nullptr; }
            }
            cap = 0;
            for (int i = 0; i < 2; i++) {
                if (host) HIPCK(hipHostMalloc(&p[i], want, hipHostMallocDefault));
                else HIPCK(hipMalloc(&p[i], want));
            }
            cap = want;
            return 0;
        };
        CK(grow(pin_in, pin_in_cap, in_b, true));
        CK(grow(pin_out, pin_out_cap, out_b, true));
        CK(grow(dev_in, dev_in_cap, in_b, false));
        CK(grow(dev_out, dev_out_cap, out_b, false));
        CK(grow(dev_scr, dev_scr_cap, scr_b, false));
        return 0;
    }
};

}  // namespace

struct bft_gpu_group {
    std::vector<bft_gpu*> members;  // one handle per device slot
    std::vector<bool> owned;        // replicas made here (freed with the group); the source handle is the caller's
    std::vector<int> devices;       // the device of every slot
    std::vector<Member*> workers;   // one per slot (its thread may be missing: the caller's thread stands in)
    int k = 0, B = 0;
    uint32_t nb_genomes = 0;
};

extern "C" int bft_gpu_group_shard(uint64_t n, int parts, int i, uint64_t* begin, uint64_t* end) {
    if (parts <= 0 || i < 0 || i >= parts || !begin || !end) return bft_fail(BFT_GPU_E_ARG, "bad shard arguments");
    uint64_t per = (n + (uint64_t)parts - 1) / (uint64_t)parts;
    per = (per + 63) / 64 * 64;  // slice starts are multiples of 64 queries = whole bytes (and u64 words) of the presence bitmap
    const uint64_t a = std::min<uint64_t>(n, (uint64_t)i * per);
    *begin = a;
    *end = std::min<uint64_t>(n, a + per);
    return BFT_GPU_OK;
}

static int replicate(bft_gpu* src, int src_dev, int dst_dev, bft_gpu** out) {
    uint64_t nbytes = 0;
    CK(bft_gpu_image_size(src, &nbytes));
    void *a = nullptr, *b = nullptr;
    int rc = 0;
    HIPCK(hipSetDevice(src_dev));
    if (hipMalloc(&a, nbytes) != hipSuccess) return bft_fail(BFT_GPU_E_HIP, "hipMalloc (image blob) failed");
    rc = bft_gpu_image_pack(src, a, nbytes, nullptr);
    if (rc == 0 && dst_dev != src_dev) {
        if (hipSetDevice(dst_dev) != hipSuccess || hipMalloc(&b, nbytes) != hipSuccess) rc = bft_fail(BFT_GPU_E_HIP, "hipMalloc on the replica's device failed");
        if (rc == 0 && hipMemcpyPeer(b, dst_dev, a, src_dev, nbytes) != hipSuccess) rc = bft_fail(BFT_GPU_E_HIP, "hipMemcpyPeer failed");
        if (rc == 0 && hipDeviceSynchronize() != hipSuccess) rc = bft_fail(BFT_GPU_E_HIP, "peer copy failed");
    }
    if (rc == 0) rc = bft_gpu_image_unpack(b ? b : a, nbytes, dst_dev, out);
    if (b) { (void)hipSetDevice(dst_dev); (void)hipFree(b); }
    (void)hipSetDevice(src_dev);
    (void)hipFree(a);
    return rc;
}

extern "C" int bft_gpu_group_create(bft_gpu* src, int src_device, const int* devices, int n_devices, bft_gpu_group** out) {
    if (!src || !devices || n_devices <= 0 || !out) return bft_fail(BFT_GPU_E_ARG, "NULL argument");
    *out = nullptr;
    int prev = -1, ndev = 0;
    (void)hipGetDevice(&prev);
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    for (int i = 0; i < n_devices; i++)
        if (devices[i] < 0 || devices[i] >= ndev) return bft_fail(BFT_GPU_E_ARG, "bad device index in the group");
    if (src_device < 0 || src_device >= ndev) return bft_fail(BFT_GPU_E_ARG, "bad source device index");
    uint64_t info[16] = {0};
    CK(bft_gpu_build(src));
    CK(bft_gpu_info(src, info, 16));
    bft_gpu_group* g = new bft_gpu_group();
    g->k = (int)info[0];
    g->B = (2 * g->k + 7) / 8;
    g->nb_genomes = (uint32_t)info[11];
    bool src_used = false;
    int rc = 0;
    for (int i = 0; i < n_devices && rc == 0; i++) {
        if (devices[i] == src_device && !src_used) {  // the source serves its own device (first slot that names it)
            g->members.push_back(src);
            g->owned.push_back(false);
            g->devices.push_back(devices[i]);
            src_used = true;
            continue;
        }
        bft_gpu* r = nullptr;
        rc = replicate(src, src_device, devices[i], &r);
        if (rc == 0) { g->members.push_back(r); g->owned.push_back(true); g->devices.push_back(devices[i]); }
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (rc != 0) {
        (void)hipGetLastError();  // (a failed runtime call leaves its error behind: the next launch check must not find it)
        const std::string keep = bft_gpu_last_error();
        for (size_t i = 0; i < g->members.size(); i++)
            if (g->owned[i]) bft_gpu_free(g->members[i]);
        delete g;
        return bft_fail(rc, keep);
    }
    for (size_t i = 0; i < g->members.size(); i++) {
        Member* m = new (std::nothrow) Member();
        if (!m) break;  // (the slots without a worker are answered by the single-GPU host entry points)
        m->h = g->members[i];
        m->device = g->devices[i];
        try {
            m->th = std::thread([m] { m->loop(); });
        } catch (...) {  // (no thread to be had: the calling thread will run this member's jobs)
        }
        g->workers.push_back(m);
    }
    *out = g;
    return BFT_GPU_OK;
}

extern "C" void bft_gpu_group_free(bft_gpu_group* g) {
    if (!g) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (Member* m : g->workers) {
        if (m->th.joinable()) {
            { std::lock_guard<std::mutex> lk(m->mu); m->quit = true; }
            m->cv.notify_all();
            m->th.join();
        }
        m->free_staging();
        delete m;
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    for (size_t i = 0; i < g->members.size(); i++)
        if (g->owned[i]) bft_gpu_free(g->members[i]);
    delete g;
}

extern "C" int bft_gpu_group_size(bft_gpu_group* g) { return g ? (int)g->members.size() : 0; }

extern "C" int bft_gpu_group_member_device(bft_gpu_group* g, int i) {
    if (!g || i < 0 || i >= (int)g->members.size()) return -1;
    return g->devices[(size_t)i];
}

extern "C" int bft_gpu_group_member_footprint(bft_gpu_group* g, int i, uint64_t* out, int n_out) {
    if (!g || i < 0 || i >= (int)g->members.size()) return bft_fail(BFT_GPU_E_ARG, "bad group slot");
    return bft_gpu_footprint(g->members[(size_t)i], out, n_out);
}
extern "C" int bft_gpu_group_member_info(bft_gpu_group* g, int i, uint64_t* out, int n_out) {
    if (!g || i < 0 || i >= (int)g->members.size()) return bft_fail(BFT_GPU_E_ARG, "bad group slot");
    return bft_gpu_info(g->members[(size_t)i], out, n_out);
}

// ---- device-resident batches: member i answers the batch that lies in ITS GPU's memory, on ITS stream.  Nothing here waits for a GPU and no
// host thread is started: every member's call only enqueues (the single-GPU *_dev entry points are stream-ordered), so the members of the
// group run side by side.  streams may be NULL (every member's own stream) and so may streams[i].
extern "C" int bft_gpu_group_query_presence_dev(bft_gpu_group* g, const void* const* d_kmers, const uint64_t* n, void* const* d_present_bits, void* const* hip_streams) {
    if (!g || !d_kmers || !n || !d_present_bits) return bft_fail(BFT_GPU_E_ARG, "NULL argument");
    for (size_t i = 0; i < g->members.size(); i++)  // every slot is checked before anything is enqueued on any member
        if (n[i] && (!d_kmers[i] || !d_present_bits[i])) return bft_fail(BFT_GPU_E_ARG, "NULL device pointer for a slot with k-mers");
    for (size_t i = 0; i < g->members.size(); i++)
        if (n[i]) CK(bft_gpu_query_presence_dev(g->members[i], d_kmers[i], n[i], d_present_bits[i], hip_streams ? hip_streams[i] : nullptr));
    return BFT_GPU_OK;
}
extern "C" int bft_gpu_group_query_color_rows_dev(bft_gpu_group* g, const void* const* d_kmers, const uint64_t* n, void* const* d_present_bits, void* const* d_rows,
                                                  void* const* d_scratch_rows_u32, void* const* hip_streams) {
    if (!g || !d_kmers || !n || !d_present_bits || !d_rows || !d_scratch_rows_u32) return bft_fail(BFT_GPU_E_ARG, "NULL argument");
    for (size_t i = 0; i < g->members.size(); i++)
        if (n[i] && (!d_kmers[i] || !d_present_bits[i] || !d_rows[i] || !d_scratch_rows_u32[i])) return bft_fail(BFT_GPU_E_ARG, "NULL device pointer for a slot with k-mers");
    for (size_t i = 0; i < g->members.size(); i++)
        if (n[i]) CK(bft_gpu_query_color_rows_dev(g->members[i], d_kmers[i], n[i], d_present_bits[i], d_rows[i], d_scratch_rows_u32[i], hip_streams ? hip_streams[i] : nullptr));
    return BFT_GPU_OK;
}
extern "C" int bft_gpu_group_query_branching_dev(bft_gpu_group* g, const void* const* d_kmers, const uint64_t* n, void* const* d_branching_bits, void* const* d_counts,
                                                 void* const* hip_streams) {
    if (!g || !d_kmers || !n || !d_branching_bits) return bft_fail(BFT_GPU_E_ARG, "NULL argument");
    for (size_t i = 0; i < g->members.size(); i++)
        if (n[i] && (!d_kmers[i] || !d_branching_bits[i])) return bft_fail(BFT_GPU_E_ARG, "NULL device pointer for a slot with k-mers");
    for (size_t i = 0; i < g->members.size(); i++)
        if (n[i]) CK(bft_gpu_query_branching_dev(g->members[i], d_kmers[i], n[i], d_branching_bits[i], d_counts ? d_counts[i] : nullptr, hip_streams ? hip_streams[i] : nullptr));
    return BFT_GPU_OK;
}

// Member i answers its slice on its own thread; the first failure (code and message) is reported on the calling thread.
template <class F>
static int run_sharded(bft_gpu_group* g, uint64_t n, F f) {
    const int parts = (int)g->members.size();
    std::vector<int> rc(parts, 0);
    std::vector<std::string> msg(parts);
    std::vector<uint64_t> lo(parts, 0), hi(parts, 0);
    for (int i = 0; i < parts; i++) CK(bft_gpu_group_shard(n, parts, i, &lo[i], &hi[i]));  // (before any job is posted: an early return must not leave one behind)
    std::vector<int> posted;
    for (int i = 0; i < parts; i++) {
        const uint64_t a = lo[i], b = hi[i];
        if (b <= a) continue;
        Member* m = i < (int)g->workers.size() ? g->workers[(size_t)i] : nullptr;
        if (m && m->th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(m->mu);
                m->job = [=] { return f(m, g->members[(size_t)i], a, b - a); };
                m->has_job = true;
                m->finished = false;
            }
            m->cv.notify_all();
            posted.push_back(i);
        } else {  // no thread for this slot: here and now
            int prev = -1;
            (void)hipGetDevice(&prev);
            rc[i] = f(m, g->members[(size_t)i], a, b - a);
            if (rc[i] != 0) msg[i] = bft_gpu_last_error();
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    }
    for (int i : posted) {
        Member* m = g->workers[(size_t)i];
        std::unique_lock<std::mutex> lk(m->mu);
        m->cv.wait(lk, [&] { return m->finished; });
        rc[i] = m->rc;
        msg[i] = m->msg;
    }
    for (int i = 0; i < parts; i++)
        if (rc[i] != 0) return bft_fail(rc[i], msg[i]);
    return BFT_GPU_OK;
}

// One member's slice through its pinned slots.  call(d_in, m, d_bits, d_out, d_scratch, stream) enqueues the *_dev entry point.
// kmers / bits / out: the slice's first byte in the caller's arrays (bits, out may be NULL: not wanted -- the device still produces them).
template <class Call>
static int staged(Member* w, const StageShape& sh, const uint8_t* kmers, uint64_t n, uint8_t* bits, uint8_t* out, Call call) {
    // chunks of at most 2^22 k-mers and ~64 MiB per buffer, whole presence words
    const uint64_t per = std::max<uint64_t>(sh.in_bytes, std::max<uint64_t>(sh.out_bytes, 1));
    const uint64_t chunk = std::max<uint64_t>(64, std::min<uint64_t>(1ull << 22, ((64ull << 20) / per) & ~63ull));
    const uint64_t mc0 = std::min(n, chunk);
    const size_t bits_b = (((mc0 + 63) / 64) * 8 + 255) & ~(size_t)255, out_b = bits_b + mc0 * sh.out_bytes;  // (the second output starts 256-byte aligned: the row kernels store 16 bytes per lane)
    CK(w->reserve(mc0 * sh.in_bytes, out_b, sh.scratch ? mc0 * 4 : 0));
    struct Pending { uint64_t at = 0, m = 0; bool live = false; } pend[2];
    auto retire = [&](int slot) -> int {  // the slot's answers are in its pinned buffer: into the caller's arrays
        if (!pend[slot].live) return 0;
        HIPCK(hipEventSynchronize(w->ev[slot]));
        const uint64_t at = pend[slot].at, m = pend[slot].m;
        const uint8_t* p = (const uint8_t*)w->pin_out[slot];
        if (bits) std::memcpy(bits + at / 8, p, (m + 7) / 8);
        if (out && sh.out_bytes) std::memcpy(out + at * sh.out_bytes, p + bits_b, m * sh.out_bytes);
        pend[slot].live = false;
        return 0;
    };
    int rc = 0;
    uint64_t c = 0;
    for (uint64_t at = 0; at < n && rc == 0; at += chunk, c++) {
        const int slot = (int)(c & 1);
        const uint64_t m = std::min(chunk, n - at);
        rc = retire(slot);
        if (rc) break;
        std::memcpy(w->pin_in[slot], kmers + at * sh.in_bytes, m * sh.in_bytes);
        uint8_t* d_out = (uint8_t*)w->dev_out[slot];
        if (hipMemcpyAsync(w->dev_in[slot], w->pin_in[slot], m * sh.in_bytes, hipMemcpyHostToDevice, w->st) != hipSuccess) { rc = bft_fail(BFT_GPU_E_HIP, "copy to the device failed"); break; }
        rc = call(w->dev_in[slot], m, d_out, d_out + bits_b, w->dev_scr[slot], (void*)w->st);
        if (rc) break;
        if (hipMemcpyAsync(w->pin_out[slot], d_out, ((m + 63) / 64) * 8, hipMemcpyDeviceToHost, w->st) != hipSuccess ||
            (sh.out_bytes && hipMemcpyAsync((uint8_t*)w->pin_out[slot] + bits_b, d_out + bits_b, m * sh.out_bytes, hipMemcpyDeviceToHost, w->st) != hipSuccess) ||
            hipEventRecord(w->ev[slot], w->st) != hipSuccess) {
            rc = bft_fail(BFT_GPU_E_HIP, "copy from the device failed");
            break;
        }
        pend[slot] = Pending{at, m, true};
    }
    if (rc == 0) rc = retire((int)(c & 1));
    if (rc == 0) rc = retire((int)((c + 1) & 1));
    if (rc != 0) {
        const std::string keep = bft_gpu_last_error();
        (void)hipStreamSynchronize(w->st);  // (nothing of this call stays in flight over the caller's arrays or the slots)
        (void)hipGetLastError();
        return bft_fail(rc, keep);
    }
    return 0;
}

extern "C" int bft_gpu_group_query_presence(bft_gpu_group* g, const uint8_t* kmers, uint64_t n, uint8_t* present_bits) {
    if (!g || ((!kmers || !present_bits) && n)) return bft_fail(BFT_GPU_E_ARG, "NULL argument");
    const uint64_t B = (uint64_t)g->B;
    return run_sharded(g, n, [=](Member* w, bft_gpu* h, uint64_t a, uint64_t m) -> int {
        if (!w) return bft_gpu_query_presence(h, kmers + a * B, m, present_bits + a / 8);
        const StageShape sh{B, true, 0, false};
        return staged(w, sh, kmers + a * B, m, present_bits + a / 8, nullptr,
                      [&](void* d_in, uint64_t mc, void* d_bits, void*, void*, void* st) { return bft_gpu_query_presence_dev(h, d_in, mc, d_bits, st); });
    });
}

extern "C" int bft_gpu_group_query_color_rows(bft_gpu_group* g, const uint8_t* kmers, uint64_t n, uint8_t* present_bits, uint8_t* rows) {
    if (!g || ((!rows || !kmers) && n)) return bft_fail(BFT_GPU_E_ARG, "NULL argument");
    const uint64_t B = (uint64_t)g->B, rowbytes = (g->nb_genomes + 7) / 8;
    return run_sharded(g, n, [=](Member* w, bft_gpu* h, uint64_t a, uint64_t m) -> int {
        if (!w || rowbytes == 0) return bft_gpu_query_color_rows(h, kmers + a * B, m, present_bits ? present_bits + a / 8 : nullptr, rows + a * rowbytes);
        const StageShape sh{B, true, rowbytes, true};
        return staged(w, sh, kmers + a * B, m, present_bits ? present_bits + a / 8 : nullptr, rows + a * rowbytes,
                      [&](void* d_in, uint64_t mc, void* d_bits, void* d_rows, void* d_scr, void* st) { return bft_gpu_query_color_rows_dev(h, d_in, mc, d_bits, d_rows, d_scr, st); });
    });
}

extern "C" int bft_gpu_group_query_branching(bft_gpu_group* g, const uint8_t* kmers, uint64_t n, uint8_t* branching_bits, uint8_t* counts) {
    if (!g || ((!kmers || !branching_bits) && n)) return bft_fail(BFT_GPU_E_ARG, "NULL argument");
    const uint64_t B = (uint64_t)g->B;
    return run_sharded(g, n, [=](Member* w, bft_gpu* h, uint64_t a, uint64_t m) -> int {
        if (!w) return bft_gpu_query_branching(h, kmers + a * B, m, branching_bits + a / 8, counts ? counts + a : nullptr);
        const StageShape sh{B, true, counts ? 1u : 0u, false};
        return staged(w, sh, kmers + a * B, m, branching_bits + a / 8, counts ? counts + a : nullptr, [&](void* d_in, uint64_t mc, void* d_bits, void* d_counts, void*, void* st) {
            return bft_gpu_query_branching_dev(h, d_in, mc, d_bits, counts ? d_counts : nullptr, st);
        });
    });
}
