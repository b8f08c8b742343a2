// bft_group.cpp -- one index on several GPUs of one process, behind the C-ABI (include/bft_gpu.h, bft_gpu_group_*).
//
// The reference answers -query_kmers / -query_branching with one loop over one BFT_Root (src/file_io.c:651-895, :897-1020).  Queries
// are independent and read-only, so the batched path shards them: the built index is replicated into the HBM of every
// device of the group (bft_gpu_image_pack on the source GPU, one peer copy over xGMI per replica, bft_gpu_image_unpack there), a
// host batch is cut into contiguous slices whose starts are multiples of 64 k-mers (so the per-device presence bitmaps are
// byte ranges of the caller's bitmap), and one host thread per device runs the ordinary single-GPU entry point on its slice.
// Nothing collective is involved: the answers land in the caller's buffers.  (Processes that hold one GPU each -- bench.py under
// torchrun -- replicate with one RCCL broadcast of the same blob instead: bloomfiltertrie_amd/dist.py.)
#include <hip/hip_runtime.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/bft_gpu.h"
#include "bft_dev.h"

struct bft_gpu_group {
    std::vector<bft_gpu*> members;  // one handle per device slot
    std::vector<bool> owned;        // replicas made here (freed with the group); the source handle is the caller's
    std::vector<int> devices;       // the device of every slot
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
    *out = g;
    return BFT_GPU_OK;
}

extern "C" void bft_gpu_group_free(bft_gpu_group* g) {
    if (!g) return;
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

// One host thread per member on its slice; the first failure (code and message) is reported on the calling thread.
template <class F>
static int run_sharded(bft_gpu_group* g, uint64_t n, F f) {
    const int parts = (int)g->members.size();
    std::vector<int> rc(parts, 0);
    std::vector<std::string> msg(parts);
    std::vector<uint64_t> lo(parts, 0), hi(parts, 0);
    for (int i = 0; i < parts; i++) CK(bft_gpu_group_shard(n, parts, i, &lo[i], &hi[i]));  // (before any thread exists: an early return must not leave one unjoined)
    std::vector<std::thread> th;
    for (int i = 0; i < parts; i++) {
        const uint64_t a = lo[i], b = hi[i];
        if (b <= a) continue;
        auto work = [&, i, a, b] {
            rc[i] = f(g->members[i], a, b - a);
            if (rc[i] != 0) msg[i] = bft_gpu_last_error();
        };
        try {
            th.emplace_back(work);
        } catch (...) {  // (no thread to be had: this slice is answered on the calling thread -- nothing may cross the extern "C" boundary)
            work();
        }
    }
    for (std::thread& t : th) t.join();
    for (int i = 0; i < parts; i++)
        if (rc[i] != 0) return bft_fail(rc[i], msg[i]);
    return BFT_GPU_OK;
}

extern "C" int bft_gpu_group_query_presence(bft_gpu_group* g, const uint8_t* kmers, uint64_t n, uint8_t* present_bits) {
    if (!g || ((!kmers || !present_bits) && n)) return bft_fail(BFT_GPU_E_ARG, "NULL argument");
    const uint64_t B = (uint64_t)g->B;
    return run_sharded(g, n, [&](bft_gpu* h, uint64_t a, uint64_t m) { return bft_gpu_query_presence(h, kmers + a * B, m, present_bits + a / 8); });
}

extern "C" int bft_gpu_group_query_color_rows(bft_gpu_group* g, const uint8_t* kmers, uint64_t n, uint8_t* present_bits, uint8_t* rows) {
    if (!g || ((!rows || !kmers) && n)) return bft_fail(BFT_GPU_E_ARG, "NULL argument");
    const uint64_t B = (uint64_t)g->B, rowbytes = (g->nb_genomes + 7) / 8;
    return run_sharded(g, n, [&](bft_gpu* h, uint64_t a, uint64_t m) {
        return bft_gpu_query_color_rows(h, kmers + a * B, m, present_bits ? present_bits + a / 8 : nullptr, rows + a * rowbytes);
    });
}

extern "C" int bft_gpu_group_query_branching(bft_gpu_group* g, const uint8_t* kmers, uint64_t n, uint8_t* branching_bits, uint8_t* counts) {
    if (!g || ((!kmers || !branching_bits) && n)) return bft_fail(BFT_GPU_E_ARG, "NULL argument");
    const uint64_t B = (uint64_t)g->B;
    return run_sharded(g, n, [&](bft_gpu* h, uint64_t a, uint64_t m) {
        return bft_gpu_query_branching(h, kmers + a * B, m, branching_bits + a / 8, counts ? counts + a : nullptr);
    });
}
