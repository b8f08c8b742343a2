// bft_file.cpp -- the reference's serialised .bft format, host side.
//
//   bft_file_read : read_BFT_Root / read_Node / read_UC / read_CC (src/write_to_disk.c:260-776) reduced to what
//                   the GPU build needs: every stored k-mer with its sorted genome-id list, decoded from the
//                   reference's annotation encodings (modes 0/1/2, src/annotation.c:2086-2250; mode 3 = index
//                   into the file's comp_set_colors with delta-coded entries, :1840-1922; the one extra byte of
//                   the extended-annotation table, src/UC.c:501-521).  Bloom filters and skip tables
//                   are not in the file; the image rebuilds its own.
//   bft_file_write: write_BFT_Root / write_Node / write_UC / write_CC (src/write_to_disk.c:21-258) from a host copy
//                   of the image arrays, in the reference's container layout (filter2 | filter3 |
//                   extra_filter3 or in-row cluster flags | children_type | 128-prefix UC buckets | child Nodes).
// File (de)serialisation is host work in the reference too; nothing here is on the query path.
#include "bft_file.h"

#include <stdio.h>
#include <string.h>

#include <stdlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <exception>
#include <deque>
#include <functional>
#include <memory>
#include <thread>

#include "bft_index.h"
#include "bft_walk.h"

// the disposer threads in flight (bft_file.h)
static std::mutex g_disp_mu;
static std::condition_variable g_disp_cv;
static int g_disp_n = 0;
void bft_dispose_begin(void) {
    std::lock_guard<std::mutex> lk(g_disp_mu);
    g_disp_n++;
}
void bft_dispose_end(void) {
    std::lock_guard<std::mutex> lk(g_disp_mu);
    if (--g_disp_n == 0) g_disp_cv.notify_all();
}
void bft_dispose_drain(void) {
    std::unique_lock<std::mutex> lk(g_disp_mu);
    g_disp_cv.wait(lk, [] { return g_disp_n == 0; });
}
__attribute__((destructor)) static void bft_dispose_at_unload(void) { bft_dispose_drain(); }

namespace {

// BFT_GPU_TRACE_IO=1: where the writer and the loader spend their time (stderr)
struct IoTrace {
    bool on = getenv("BFT_GPU_TRACE_IO") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    void mark(const char* what) {
        if (!on) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[bft_gpu io] %8.1f ms (+%.1f) %s\n", std::chrono::duration<double, std::milli>(t - t0).count(), std::chrono::duration<double, std::milli>(t - last).count(), what);
        last = t;
    }
};

// How many host threads the writer and the loader use: BFT_GPU_IO_THREADS, else the hardware's (at most 32; a container's CPU quota may grant
// fewer cores than it shows -- oversubscribing them costs little here).
unsigned io_threads() {
    if (const char* e = getenv("BFT_GPU_IO_THREADS")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 1) return (unsigned)std::min<long>(v, 256);
    }
    const unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(hc ? hc : 8u, 32u));
}
// joins what it holds when it goes out of scope, however that happens (an exception on the calling thread must not meet joinable threads)
struct ThreadJoiner {
    std::vector<std::thread> th;
    ~ThreadJoiner() {
        for (std::thread& x : th)
            if (x.joinable()) x.join();
    }
};
// First thing a worker does: touch the C++ runtime's per-thread exception state.  libstdc++ keeps it in thread-local storage that the dynamic
// loader allocates on a thread's FIRST access when the library came in through dlopen (ctypes, JNI, cgo all do that) -- and the loader aborts
// the process if that allocation fails.  A worker whose first throw is the std::bad_alloc of an exhausted heap would ask for it at the worst
// moment; asked for at the start, while there is memory, the failure stays an exception.
inline void touch_exception_state() { (void)std::uncaught_exceptions(); }
// jobs 0 .. n-1 over the threads (the calling thread takes part).  An exception inside a job -- an allocation a hostile file asks for -- stops the
// hand-out of jobs and is thrown again, as std::bad_alloc, on the CALLING thread once every worker is back: the caller's try / catch sees it, and
// nothing reaches std::terminate.
template <class F>
void parallel_jobs(size_t n, F f) {
    const unsigned nt = (unsigned)std::min<size_t>(io_threads(), std::max<size_t>(1, n));
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    auto work = [&](unsigned t) {
        touch_exception_state();
        try {
            for (;;) {
                const size_t j = next.fetch_add(1);
                if (j >= n || failed.load(std::memory_order_relaxed)) break;
                f(j, t);
            }
        } catch (...) {
            failed = true;
        }
    };
    {
        ThreadJoiner pool;
        for (unsigned t = 1; t < nt; t++) {
            try { pool.th.emplace_back(work, t); } catch (...) { break; }  // (no thread to be had: the others do its share)
        }
        work(0);
    }
    if (failed) throw std::bad_alloc();
}


// ------------------------------------------------------------------------------------------------
// annotation codec (src/annotation.c)
// ------------------------------------------------------------------------------------------------
int nb_bytes_id(uint32_t id) {  // get_nb_bytes_power2_annot, include/log2.h:45-50
    int bits = id ? 32 - __builtin_clz(id) : 1;
    return (bits + 5) / 6;
}

int put_id(uint8_t* out, uint32_t id, uint8_t start_flag, uint8_t cont_flag) {
    const int nb = nb_bytes_id(id);
    for (int j = 0; j < nb; j++) out[j] = (uint8_t)((((id >> (6 * (nb - 1 - j))) & 0x3f) << 2) | (j == 0 ? start_flag : cont_flag));
    return nb;
}

// The bytes the reference holds for a colour set.  It picks the encoding anew at every insertion of a genome id
// (compute_best_mode, src/annotation.c:416-656, called from modify_annotations, src/retrieveAnnotation.c:232-314): the smallest
// of modes 0 (bitmap) / 1 (ranges) / 2 (id list) for the set as it then is -- mode 2 over 1 on equality, mode 0 when it is no
// larger (:638-650) -- EXCEPT that on a tie with the mode the annotation is already in, it stays in that mode (:652-653).  The
// result therefore depends on the order the ids arrived in; they arrive in ascending order (a k-mer meets genome ids in
// increasing order), so the sorted id list IS the history and the rule is replayed over it, one id at a time, with the sizes of
// :621-633 (mode 1: a new range costs two ids, extending the last one swaps its end).  While the annotation is a bitmap the reference
// does not know its id list: it re-derives the two list sizes by scanning the bits (:476-535), and prices the END of a run with the
// byte count of the id one past it (:515-523) -- one byte too many when the run ends at 63, 4095, 262143 or 16777215 (the ids after
// which an id takes one more byte).  That estimate, not the exact size, is what the bitmap is compared with, so it is replayed too
// (`over`); an annotation that leaves the bitmap on such an estimate is given the estimated size (a trailing zero byte, which
// the decoders stop at).  disabled_flags (:622) is never set anywhere in the reference (grep: only tested), so there is nothing to
// replay.  E.g. {6,7}: 6 enters in mode 2 (1 byte < 2 bytes of bitmap), at 7 all three modes cost 2 bytes and the annotation STAYS
// in mode 2 (a fresh decision would pick the bitmap).
void annot_encode(const uint32_t* ids, uint32_t n, std::vector<uint8_t>& out) {
    out.clear();
    if (n == 0) { out.push_back(0); return; }
    size_t s0 = 0, s1 = 0, s2 = 0, sz = 0;
    int mode = -1;
    uint32_t pend = 0;  // runs of the set so far that end at an id of the form 64^j - 1 (bit j-1): over-priced while in bitmap mode
    for (uint32_t a = 0; a < n; a++) {
        const size_t b = (size_t)nb_bytes_id(ids[a]);
        s0 = (3 + (size_t)ids[a] + 7) / 8;
        s2 += b;
        if (a > 0 && ids[a] == ids[a - 1] + 1) s1 = s1 + b - (size_t)nb_bytes_id(ids[a - 1]);
        else s1 += 2 * b;
        const size_t s1e = s1 + (mode == 0 ? (size_t)__builtin_popcount(pend) : 0);  // the reference's estimate of the ranges size
        int m;
        if (s2 <= s1e) { m = 2; sz = s2; } else { m = 1; sz = s1e; }
        if (sz >= s0) { m = 0; sz = s0; }
        if (mode >= 0 && m != mode && (mode == 0 ? s0 : (mode == 1 ? s1e : s2)) == sz) m = mode;  // tie: the current mode stays
        mode = m;
        for (int j = 1; j <= 4; j++) {
            const uint32_t edge = (1u << (6 * j)) - 1u;
            if (ids[a] == edge) pend |= 1u << (j - 1);
            if (ids[a] == edge + 1 && a > 0 && ids[a - 1] == edge) pend &= ~(1u << (j - 1));
        }
    }
    out.assign(sz, 0);
    if (mode == 0) {
        for (uint32_t a = 0; a < n; a++) out[(ids[a] + 2) / 8] |= (uint8_t)(1u << ((ids[a] + 2) % 8));
    } else if (mode == 1) {
        size_t o = 0;
        for (uint32_t a = 0; a < n;) {
            uint32_t b = a;
            while (b + 1 < n && ids[b + 1] == ids[b] + 1) b++;
            o += put_id(&out[o], ids[a], 0x1, 0x2);
            o += put_id(&out[o], ids[b], 0x1, 0x2);
            a = b + 1;
        }
    } else {
        size_t o = 0;
        for (uint32_t a = 0; a < n; a++) o += put_id(&out[o], ids[a], 0x2, 0x1);
    }
}

// get_id_genomes_from_annot, modes 0/1/2 (src/annotation.c:2086-2250)
bool annot_decode(const uint8_t* a, size_t size, std::vector<uint32_t>& ids) {
    ids.clear();
    if (size == 0) return true;
    const int mode = a[0] & 3;
    size_t i = 0;
    if (mode == 0) {
        for (size_t b = 2; b < size * 8; b++)
            if (a[b / 8] & (1u << (b % 8))) ids.push_back((uint32_t)(b - 2));
    } else if (mode == 1) {
        bool second = false;
        uint32_t prev = 0;
        while (i < size && (a[i] & 0x1)) {
            uint32_t v = a[i++] >> 2;
            while (i < size && (a[i] & 0x2)) v = (v << 6) | (a[i++] >> 2);
            if (second) { for (uint32_t j = prev + 1; j <= v && prev != v; j++) ids.push_back(j); }
            else { ids.push_back(v); prev = v; }
            second = !second;
        }
    } else if (mode == 2) {
        while (i < size && (a[i] & 0x2)) {
            uint32_t v = a[i++] >> 2;
            while (i < size && (a[i] & 0x1)) v = (v << 6) | (a[i++] >> 2);
            ids.push_back(v);
        }
    } else
        return false;  // mode 3: index into comp_set_colors, resolved by the reader (needs the file's dictionary)
    return true;
}

// comp_set_colors (src/write_to_disk.c:283-310): elements of equal-size entries; an entry in mode 1/2 is delta-coded
// by comp_annotation (src/annotation.c:1777-1838) and decoded as decomp_annotation + get_id_genomes_from_annot with
// comp_annot > 0 do (:1840-1922, :2179-2226)
struct CompElem {
    int64_t last_index = 0;
    int size_annot = 0;
    std::vector<uint8_t> bytes;
};

bool decode_comp_entry(const uint8_t* a, size_t size, std::vector<uint32_t>& ids) {
    if (size == 0) { ids.clear(); return true; }
    const int mode = a[0] & 3;
    if (mode == 0) return annot_decode(a, size, ids);
    if (mode == 3) return false;
    ids.clear();
    const uint8_t flag1 = mode == 2 ? 2 : 1, flag2 = mode == 2 ? 1 : 2;
    std::vector<uint32_t> st;
    size_t i = 0;
    while (i < size && (a[i] & flag1)) {
        uint32_t v = a[i++] >> 2;
        while (i < size && (a[i] & flag2)) v = (v << 6) | (a[i++] >> 2);
        st.push_back(v);
    }
    for (size_t q = 1; q < st.size(); q++) st[q] += st[q - 1];
    if (mode == 2) ids = st;
    else
        for (size_t q = 0; q + 1 < st.size(); q += 2)
            for (uint32_t v = st[q]; v <= st[q + 1]; v++) ids.push_back(v);
    return true;
}

inline int nb_bytes(int i) { return (2 * i + 7) / 8; }
inline bool level_min_of(int k, int i) { return i == k || i % 36 == 9; }  // src/CC.c:1906-1989

// ------------------------------------------------------------------------------------------------
// reader: parse the stream into a tree of raw containers, then walk it with the path known
// ------------------------------------------------------------------------------------------------
struct Rows {
    const uint8_t* data = nullptr;  // a view into the file's bytes (Reader::file)
    int size_annot = 0, nbs = 0, count = 0;
    std::vector<int> ext_pos;
    std::vector<uint8_t> ext_byte;
};
struct PNode;
struct PCC {
    int s = 8, n = 0;
    std::vector<uint8_t> f2, f3, ex;
    std::vector<uint16_t> cnts;
    std::vector<Rows> buckets;
    std::vector<PNode> children;  // child Nodes in prefix order
};
struct PNode {
    int flag = 0;
    Rows uc;
    std::vector<PCC> ccs;
};

// Two passes.  The first walks the file's bytes (read into memory at once) into a tree of raw containers -- headers and filters copied, the rows of
// the UC blocks left where they are (views) --: sequential, a few per cent of the time.  The second rebuilds every k-mer with its genome ids:
// the work is in the rows (2 x 10^8 (k-mer, genome) pairs on the 100-genome index), so the 128-prefix blocks of the ROOT's CCs -- each with the
// subtrees of its child Nodes -- are dealt out to a pool of threads, every thread with an Emitter of its own (path buffer, id list, per-genome output).
struct Emitter {  // what the second pass writes through: one per thread
    std::vector<uint8_t> cur;  // nucleotide codes of the k-mer being rebuilt
    std::vector<uint32_t> ids;
    std::vector<uint8_t> packed;
    std::vector<std::vector<uint8_t>> per_genome;
    uint64_t n_kmers = 0;
    bool err = false;
    std::string msg;
    void fail(const char* m) { if (!err) { err = true; msg = m; } }
};

// the whole .bft, mapped: the rows are left where they lie until the second pass decodes them, so the file's pages are faulted in by the pool's
// threads as they get there (read into a buffer first: 0.21 s for config 3's 0.8 GB before anything was parsed)
struct MappedFile {
    const uint8_t* base = nullptr;
    size_t size = 0;
    bool open(const char* path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); return false; }
        size = (size_t)st.st_size;
        if (size) {
            void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); size = 0; return false; }
            (void)madvise(m, size, MADV_WILLNEED);
            base = (const uint8_t*)m;
        }
        ::close(fd);
        return true;
    }
    ~MappedFile() { if (base) munmap((void*)base, size); }
    MappedFile() = default;
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
};

struct Reader {
    MappedFile file;
    size_t pos = 0;
    bool err = false;
    std::string msg;
    int k = 0;
    BftFileContent* out = nullptr;
    std::vector<CompElem> comp;  // the file's comp_set_colors

    // any annotation of the file -> ids; mode 3 = index into comp_set_colors (src/annotation.c:2097-2119)
    bool decode_annot(const uint8_t* a, size_t size, std::vector<uint32_t>& ids) const {
        if (size && (a[0] & 3) == 3) {
            uint32_t pos = a[0] >> 2;
            for (size_t i = 1; i < size && (a[i] & 1); i++) pos |= ((uint32_t)(a[i] >> 1)) << (6 + (i - 1) * 7);
            size_t e = 0;
            while (e < comp.size() && (int64_t)pos > comp[e].last_index) e++;
            if (e >= comp.size()) return false;
            const int64_t rel = e == 0 ? (int64_t)pos : (int64_t)pos - comp[e - 1].last_index - 1;
            return decode_comp_entry(comp[e].bytes.data() + rel * comp[e].size_annot, (size_t)comp[e].size_annot, ids);
        }
        return annot_decode(a, size, ids);
    }

    uint64_t left = 0;  // bytes of the file not read yet: every size taken from the file is checked against it before allocating
    int nbg = 0;        // genomes declared by the header: no annotation may name an id beyond them
    void fail(const char* m) { if (!err) { err = true; msg = m; } }
    bool fits(uint64_t n) { if (!err && n > left) fail("truncated file (a size field exceeds what is left of the file)"); return !err; }
    void rd(void* p, size_t n) {
        if (!err && n > left) fail("truncated file");
        if (!err && n) { memcpy(p, file.base + pos, n); pos += n; left -= n; }
    }
    const uint8_t* view(size_t n) {  // n bytes of the file, left in place
        if (!err && n > left) fail("truncated file");
        if (err) return nullptr;
        const uint8_t* p = file.base + pos;
        pos += n;
        left -= n;
        return p;
    }
    uint16_t u16() { uint16_t v = 0; rd(&v, 2); return v; }
    uint32_t u32() { uint32_t v = 0; rd(&v, 4); return v; }
    int32_t i32() { int32_t v = 0; rd(&v, 4); return v; }

    // read_UC, uncompressed branch (src/write_to_disk.c:383-531)
    void read_rows(Rows& r, int nbs, int count) {
        r.nbs = nbs; r.count = count;
        if (!count) return;
        const uint16_t next = u16();
        const int32_t sa = i32();
        if (err) return;
        if (next == 0xffff || sa < 0 || sa > (1 << 24)) { fail("compressed UC or bad size_annot"); return; }
        r.size_annot = sa;
        if (!fits((uint64_t)count * (uint64_t)(nbs + sa) + 3ull * next)) return;
        r.data = view((size_t)count * (nbs + sa));
        int pos = 0;
        for (int e = 0; e < next && !err; e++) {  // extended annotations: 2-byte big-endian position delta + 1 byte (src/UC.c:501-521)
            uint8_t t[3];
            rd(t, 3);
            pos += (t[0] << 8) | t[1];
            r.ext_pos.push_back(pos);
            r.ext_byte.push_back(t[2]);
        }
    }
    void parse_node(PNode& nd, int i) {  // read_Node (src/write_to_disk.c:353-381)
        const uint16_t field = u16();
        if (err) return;
        nd.flag = field & 1;
        read_rows(nd.uc, nb_bytes(i), field >> 1);
        const uint32_t ncc = u32();
        if (err || ncc > (1u << 24) || !fits(6ull * ncc)) { fail("bad CC count"); return; }  // >= 6 header bytes per CC
        nd.ccs.resize(ncc);
        for (uint32_t c = 0; c < ncc && !err; c++) parse_cc(nd.ccs[c], i);
    }
    void parse_cc(PCC& cc, int i) {  // read_CC (src/write_to_disk.c:533-646)
        const uint16_t type = u16(), n = u16(), nnodes = u16();
        if (err) return;
        cc.s = (type >> 1) & 0x1f;
        cc.n = n;
        const int p = 18 - cc.s, tbyte = (type >> 6) & 1;
        if (cc.s != 4 && cc.s != 8) { fail("bad CC type"); return; }
        if (!fits(((uint64_t)1 << p) / 8 + (cc.s == 8 ? n : (n + 1) / 2))) return;
        cc.f2.resize((size_t(1) << p) / 8);
        cc.f3.resize(cc.s == 8 ? n : (n + 1) / 2);
        cc.ex.assign((n + 7) / 8 + 1, 0);
        rd(cc.f2.data(), cc.f2.size());
        rd(cc.f3.data(), cc.f3.size());
        if (level_min_of(k, i)) rd(cc.ex.data(), (n + 7) / 8);
        const int nbk = (n + 127) / 128;
        cc.cnts.assign(n, 1);
        cc.buckets.resize(nbk);
        if (i != 9) {
            std::vector<uint8_t> ct(tbyte ? n : (n + 1) / 2);
            rd(ct.data(), ct.size());
            uint32_t zeros = 0;
            for (int j = 0; j < n; j++) {
                cc.cnts[j] = tbyte ? ct[j] : ((j & 1) ? ct[j / 2] >> 4 : ct[j / 2] & 0xf);
                zeros += cc.cnts[j] == 0;
            }
            if (zeros != nnodes) { fail("children_type / nb_Node_children mismatch"); return; }
            for (int b = 0; b < nbk && !err; b++) read_rows(cc.buckets[b], nb_bytes(i - 9), u16());
            cc.children.resize(nnodes);
            for (uint32_t c = 0; c < nnodes && !err; c++) parse_node(cc.children[c], i - 9);
        } else {
            for (int b = 0; b < nbk && !err; b++) read_rows(cc.buckets[b], 0, b != nbk - 1 ? 128 : n - b * 128);
        }
    }

    // ---- second pass: k-mers ----
    void row_ids(Emitter& e, const Rows& r, int q) const {  // inline annotation bytes + the extended byte if the row has one
        const uint8_t* a = r.data + (size_t)q * (r.nbs + r.size_annot) + r.nbs;
        auto it = r.ext_pos.empty() ? r.ext_pos.end() : std::lower_bound(r.ext_pos.begin(), r.ext_pos.end(), q);
        bool ok;
        if (it != r.ext_pos.end() && *it == q) {
            std::vector<uint8_t> tmp(a, a + r.size_annot);
            tmp.push_back(r.ext_byte[it - r.ext_pos.begin()]);
            ok = decode_annot(tmp.data(), tmp.size(), e.ids);
        } else
            ok = decode_annot(a, (size_t)r.size_annot, e.ids);
        if (!ok) e.fail("undecodable annotation (bad comp_set_colors index or nested mode 3)");
    }
    void put_suffix(Emitter& e, int at, const uint8_t* bytes, int len_nt, bool mask_flag) const {
        const int last = nb_bytes(len_nt) - 1;
        for (int j = 0; j < len_nt; j++) {
            uint8_t b = bytes[j / 4];
            if (mask_flag && j / 4 == last) b &= 0x7f;
            e.cur[at + j] = (b >> (2 * (j % 4))) & 3;
        }
    }
    void put_prefix(Emitter& e, int at, uint32_t r) const {  // r = n2..n9,n1
        e.cur[at] = r & 3;
        for (int j = 1; j < 9; j++) e.cur[at + j] = (r >> (2 * (9 - j))) & 3;
    }
    void emit(Emitter& e) const {
        const int B = nb_bytes(k);
        e.packed.assign(B, 0);
        for (int j = 0; j < k; j++) e.packed[j / 4] |= (uint8_t)(e.cur[j] << (2 * (j % 4)));
        for (uint32_t g : e.ids) {
            if (g >= (uint32_t)nbg) { e.fail("an annotation names a genome id beyond nb_genomes"); return; }
            if (g >= e.per_genome.size()) e.per_genome.resize((size_t)g + 1);
            e.per_genome[g].insert(e.per_genome[g].end(), e.packed.begin(), e.packed.end());
        }
        e.n_kmers++;
    }
    void emit_node(Emitter& e, const PNode& nd, int i) const {
        const int at = k - i;
        for (int q = 0; q < nd.uc.count && !e.err; q++) {
            put_suffix(e, at, nd.uc.data + (size_t)q * (nd.uc.nbs + nd.uc.size_annot), i, false);
            row_ids(e, nd.uc, q);
            emit(e);
        }
        for (const PCC& cc : nd.ccs) {
            CCPrep pr;
            if (!prepare_cc(e, cc, i, pr)) return;
            for (int b = 0; b < (cc.n + 127) / 128 && !e.err; b++) emit_cc_block(e, cc, pr, i, b);
        }
    }
    // what the blocks of a CC share: the 18-bit prefix of every position, and how many child Nodes lie before each block
    struct CCPrep {
        std::vector<uint32_t> rs;
        std::vector<uint32_t> child_before;  // per block
    };
    bool prepare_cc(Emitter& e, const PCC& cc, int i, CCPrep& pr) const {
        if (e.err) return false;
        const int n = cc.n, s = cc.s, p = 18 - s, nbk = (n + 127) / 128;
        const bool lm = level_min_of(k, i);
        // cluster starts: extra_filter3, or on level_min == 0 levels bit 7 of the group's first row / bit 0 of the child
        // Node's UC.nb_children (src/presenceNode.c:1690-1812)
        std::vector<uint8_t> starts(n, 0);
        pr.child_before.assign(nbk + 1, 0);
        if (lm || i == 9) {
            for (int j = 0; j < n; j++) starts[j] = (cc.ex[j >> 3] >> (j & 7)) & 1;
            uint32_t cn = 0;
            for (int j = 0; j < n; j++) {
                if (j % 128 == 0) pr.child_before[j / 128] = cn;
                if (i != 9 && cc.cnts[j] == 0) cn++;
            }
            pr.child_before[nbk] = cn;
        } else {
            std::vector<int> row_at(nbk, 0);
            size_t cn = 0;
            for (int j = 0; j < n; j++) {
                const int b = j / 128;
                if (j % 128 == 0) pr.child_before[b] = (uint32_t)cn;
                if (cc.cnts[j] == 0) { starts[j] = (uint8_t)cc.children[cn++].flag; continue; }
                const Rows& r = cc.buckets[b];
                if (row_at[b] + cc.cnts[j] > r.count) { e.fail("children_type / bucket mismatch"); return false; }
                starts[j] = r.data[(size_t)row_at[b] * (r.nbs + r.size_annot) + r.nbs - 1] >> 7;
                row_at[b] += cc.cnts[j];
            }
            pr.child_before[nbk] = (uint32_t)cn;
        }
        pr.rs.assign(n, 0);
        int j = 0;
        for (int pu = 0; pu < (1 << p); pu++) {
            if (!(cc.f2[pu >> 3] & (1u << (pu & 7)))) continue;
            bool first = true;
            while (j < n && (first || !starts[j])) {
                const uint32_t pv = s == 8 ? cc.f3[j] : ((j & 1) ? cc.f3[j / 2] >> 4 : cc.f3[j / 2] & 0xf);
                pr.rs[j++] = ((uint32_t)pu << s) | pv;
                first = false;
            }
        }
        if (j != n) { e.fail("filter2 / cluster starts mismatch"); return false; }
        return true;
    }
    void emit_cc_block(Emitter& e, const PCC& cc, const CCPrep& pr, int i, int b) const {
        const int n = cc.n, at = k - i;
        const bool lm = level_min_of(k, i);
        int row_at = 0;
        size_t cn = pr.child_before[b];
        for (int j = b * 128; j < std::min(n, b * 128 + 128) && !e.err; j++) {
            put_prefix(e, at, pr.rs[j]);
            if (i == 9) {
                row_ids(e, cc.buckets[b], j % 128);
                emit(e);
            } else if (cc.cnts[j] == 0) {
                emit_node(e, cc.children[cn++], i - 9);
            } else {
                const Rows& r = cc.buckets[b];
                if (row_at + cc.cnts[j] > r.count) { e.fail("children_type / bucket mismatch"); return; }
                for (int q = 0; q < cc.cnts[j]; q++) {
                    const int row = row_at + q;
                    put_suffix(e, at + 9, r.data + (size_t)row * (r.nbs + r.size_annot), i - 9, !lm);
                    row_ids(e, r, row);
                    emit(e);
                }
                row_at += cc.cnts[j];
            }
        }
    }
};

}  // namespace

namespace {
// what a load leaves behind -- the mapped file, the parsed container tree, the threads' emitters: given back by a detached thread (bft_dispose_async)
struct ReadState {
    Reader R;
    PNode root;
    std::vector<Emitter> em;
    std::vector<Reader::CCPrep> prep;
};
struct ReadStateGuard {
    ReadState* p;
    ~ReadStateGuard() {
        ReadState* q = p;
        bft_dispose_begin();
        try {
            std::thread([q] { delete q; bft_dispose_end(); }).detach();
        } catch (...) {
            delete q;
            bft_dispose_end();
        }
    }
};
}  // namespace

static bool file_read_impl(const char* path, BftFileContent& out, std::string& err);
// (an allocation an untrusted file asks for is an I/O error of the call, on whichever thread it fails: parallel_jobs hands the workers' to this one)
bool bft_file_read(const char* path, BftFileContent& out, std::string& err) {
    try {
        return file_read_impl(path, out, err);
    } catch (const std::bad_alloc&) {
        err = "out of memory while reading the file";
    } catch (const std::exception& e) {
        err = e.what();
    }
    return false;
}
static bool file_read_impl(const char* path, BftFileContent& out, std::string& err) {
    IoTrace tr;
    out = BftFileContent();
    ReadStateGuard guard{new ReadState};
    Reader& R = guard.p->R;
    if (!R.file.open(path)) { err = std::string("cannot open ") + path; return false; }
    R.left = R.file.size;
    const int lcs = R.i32();
    if (R.err || lcs < 0 || lcs > (1 << 24) || !R.fits(12ull * (uint64_t)lcs)) { err = "bad .bft header"; return false; }
    R.comp.resize(lcs);
    for (int e = 0; e < lcs && !R.err; e++) {  // src/write_to_disk.c:283-310
        R.rd(&R.comp[e].last_index, 8);
        R.comp[e].size_annot = R.i32();
        const int64_t cnt = e ? R.comp[e].last_index - R.comp[e - 1].last_index : R.comp[e].last_index + 1;
        if (R.err || cnt < 0 || cnt > (int64_t(1) << 32) || R.comp[e].size_annot < 0 || cnt * R.comp[e].size_annot > (int64_t(1) << 32) ||
            !R.fits((uint64_t)(cnt * R.comp[e].size_annot))) { R.fail("bad comp_set_colors"); break; }
        R.comp[e].bytes.resize((size_t)(cnt * R.comp[e].size_annot));
        R.rd(R.comp[e].bytes.data(), R.comp[e].bytes.size());
    }
    out.r1 = R.i32();
    out.r2 = R.i32();
    (void)R.i32();
    const int nbg = R.i32();
    out.k = R.i32();
    uint8_t comp = 0;
    R.rd(&comp, 1);
    if (R.err || comp != 0 || nbg < 0 || nbg > 100000000 || !bft_reference_k(out.k) || !R.fits(2ull * (uint64_t)nbg)) { err = "bad .bft header"; return false; }
    R.nbg = nbg;
    for (int g = 0; g < nbg && !R.err; g++) {
        const uint16_t len = R.u16();
        std::string name(len, '\0');
        R.rd(&name[0], len);
        if (!name.empty() && name.back() == '\0') name.pop_back();
        out.genomes.push_back(name);
    }
    for (int i = 9; i <= out.k && !R.err; i += 9)
        for (int q = 0; q < 7; q++) (void)R.i32();
    R.k = out.k;
    R.out = &out;
    PNode& root = guard.p->root;
    tr.mark("load: file mapped, header read");
    if (!R.err) R.parse_node(root, out.k);
    if (R.err) { err = R.msg; return false; }
    tr.mark("load: container tree parsed (rows left in place)");
    // second pass: the root's UC on this thread, the 128-prefix blocks of its CCs over the pool
    const unsigned nt = io_threads();
    std::vector<Emitter>& em = guard.p->em;
    em.resize(nt);
    for (Emitter& e : em) e.cur.assign(out.k, 0);
    {
        Emitter& e = em[0];
        for (int q = 0; q < root.uc.count && !e.err; q++) {
            R.put_suffix(e, 0, root.uc.data + (size_t)q * (root.uc.nbs + root.uc.size_annot), out.k, false);
            R.row_ids(e, root.uc, q);
            R.emit(e);
        }
    }
    std::vector<Reader::CCPrep>& prep = guard.p->prep;
    prep.resize(root.ccs.size());
    struct Blk { uint32_t cc, b; };
    std::vector<Blk> blocks;
    for (size_t c = 0; c < root.ccs.size(); c++) {
        if (!R.prepare_cc(em[0], root.ccs[c], out.k, prep[c])) break;
        for (int b = 0; b < (root.ccs[c].n + 127) / 128; b++) blocks.push_back(Blk{(uint32_t)c, (uint32_t)b});
    }
    if (!em[0].err)
        parallel_jobs(blocks.size(), [&](size_t j, unsigned t) {
            if (!em[t].err) R.emit_cc_block(em[t], root.ccs[blocks[j].cc], prep[blocks[j].cc], out.k, (int)blocks[j].b);
        });
    for (const Emitter& e : em)
        if (e.err) { err = e.msg; return false; }
    tr.mark("load: k-mers and genome ids rebuilt");
    // the threads' outputs, genome by genome (the order of a genome's k-mers does not matter: insertKmers takes a set)
    out.per_genome.assign((size_t)nbg, std::vector<uint8_t>());
    for (const Emitter& e : em) out.n_kmers += e.n_kmers;
    parallel_jobs((size_t)nbg, [&](size_t g, unsigned) {
        size_t total = 0;
        for (const Emitter& e : em) total += g < e.per_genome.size() ? e.per_genome[g].size() : 0;
        std::vector<uint8_t>& dst = out.per_genome[g];
        dst.reserve(total);
        for (Emitter& e : em)
            if (g < e.per_genome.size()) {
                dst.insert(dst.end(), e.per_genome[g].begin(), e.per_genome[g].end());
                std::vector<uint8_t>().swap(e.per_genome[g]);
            }
    });
    tr.mark("load: per-genome batches gathered");
    return true;
}

// ------------------------------------------------------------------------------------------------
// writer
// ------------------------------------------------------------------------------------------------
namespace {

// The annotation bytes of every colour set, encoded once (many rows share a set), in parallel: cs -> [off[cs], off[cs + 1]) of `bytes`.
struct AnnotCache {
    std::vector<uint64_t> off;
    std::vector<uint8_t> bytes;
    void build(const BftHostImage& im) {
        const size_t n = im.cs_off.empty() ? 0 : im.cs_off.size() - 1;
        off.assign(n + 1, 0);
        const size_t chunk = 4096, nchunks = (n + chunk - 1) / chunk;
        std::vector<std::vector<uint8_t>> part(nchunks);
        std::vector<std::vector<uint32_t>> len(nchunks);
        parallel_jobs(nchunks, [&](size_t c, unsigned) {
            std::vector<uint8_t> enc;
            for (size_t cs = c * chunk; cs < std::min(n, (c + 1) * chunk); cs++) {
                annot_encode(&im.cs_ids[im.cs_off[cs]], im.cs_off[cs + 1] - im.cs_off[cs], enc);
                len[c].push_back((uint32_t)enc.size());
                part[c].insert(part[c].end(), enc.begin(), enc.end());
            }
        });
        uint64_t o = 0;
        for (size_t c = 0; c < nchunks; c++)
            for (size_t q = 0; q < len[c].size(); q++) { off[c * chunk + q] = o; o += len[c][q]; }
        off[n] = o;
        bytes.resize(o);
        parallel_jobs(nchunks, [&](size_t c, unsigned) {
            if (!part[c].empty()) memcpy(&bytes[off[c * chunk]], part[c].data(), part[c].size());
        });
    }
};

// The file is a depth-first walk (write_Node -> write_UC, write_CC -> 128-prefix UC blocks -> child Nodes, src/write_to_disk.c:84-258).  The walk
// itself is cheap; the bytes are in the UC blocks (every stored k-mer's suffix and annotation, sorted per block) and in the subtrees of the child
// Nodes.  The writer therefore walks the ROOT's CCs on one thread and emits the file as an ordered list of parts: what it writes itself
// (headers, filters, children_type), and one part per UC block / per child Node of a root CC, filled by a pool of threads afterwards -- each
// with a Writer of its own (scratch buffers) over the shared, read-only image and annotation cache.  Round 4 wrote through fwrite on one
// thread: 4.9 s for the 0.80 GB file of the 100-genome index.
struct Writer;
struct Parts {
    std::deque<std::vector<uint8_t>> bufs;                 // in file order (a deque: growing it does not move the buffers)
    std::vector<std::function<void(Writer&)>> jobs;        // job j fills bufs[job_buf[j]]
    std::vector<size_t> job_buf;
};

struct Writer {
    const BftHostImage& im;
    const AnnotCache& ann;
    int k, L, W;
    Parts* par = nullptr;           // set on the root's walker: blocks and child Nodes of the root's CCs become jobs
    std::vector<uint8_t>* out = nullptr;
    bool err = false;

    Writer(const BftHostImage& image, const AnnotCache& a) : im(image), ann(a), k(image.k), L(image.k / 9), W(bft_words_for_k(image.k)) {}

    void wr(const void* p, size_t n) {
        if (!n) return;
        const uint8_t* b = (const uint8_t*)p;
        out->insert(out->end(), b, b + n);
    }
    void u16(uint16_t v) { wr(&v, 2); }
    void u32(uint32_t v) { wr(&v, 4); }
    void i32(int32_t v) { wr(&v, 4); }
    void literal_part() {  // what the walker writes from here on goes into a fresh part
        par->bufs.emplace_back();
        out = &par->bufs.back();
    }
    template <class F>
    void defer(F f) {  // a part of its own, filled later
        par->bufs.emplace_back();
        par->job_buf.push_back(par->bufs.size() - 1);
        par->jobs.emplace_back(std::move(f));
        literal_part();
    }

    // nucleotides [from, k) of the k-mer at T-form row -> packed suffix bytes.  In the packed layout nucleotide j sits at bits
    // 2j, so the suffix is the k-mer shifted right by 2*from bits, little-endian bytes (the bits above 2k are zero).
    template <int WW>
    void suffix_bytes_w(const uint64_t* t, int from, uint8_t* o, int nbytes) const {
        uint64_t x[WW + 1];
        bft_x_from_tform<WW>(t, k, x);
        x[WW] = 0;
        const int ws = (2 * from) >> 6, bs = (2 * from) & 63;
        uint64_t y[WW];
        for (int w = 0; w < WW; w++) {
            const uint64_t lo = w + ws < WW ? x[w + ws] : 0ull, hi = w + ws + 1 < WW ? x[w + ws + 1] : 0ull;
            y[w] = bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
        }
        memcpy(o, y, (size_t)nbytes);  // little-endian host, as the .bft format itself assumes (native ints)
    }
    void suffix_bytes(const uint64_t* t, int from, uint8_t* o, int nbytes) const {
        switch (W) {
        case 1: suffix_bytes_w<1>(t, from, o, nbytes); break;
        case 2: suffix_bytes_w<2>(t, from, o, nbytes); break;
        case 3: suffix_bytes_w<3>(t, from, o, nbytes); break;
        default: suffix_bytes_w<4>(t, from, o, nbytes); break;
        }
    }
    struct Ann { const uint8_t* p; uint32_t n; };
    Ann annot_of_row(uint64_t row) const {
        const uint32_t cs = im.tcol[row];
        return Ann{ann.bytes.data() + ann.off[cs], (uint32_t)(ann.off[cs + 1] - ann.off[cs])};
    }

    // rows of one UC block: suffixes in one flat buffer (nbs bytes each), annotations as pointers into the per-colour-set cache
    struct RowSet {
        int nbs = 0;
        std::vector<uint8_t> suf;
        std::vector<Ann> ann;
        size_t size() const { return ann.size(); }
        void clear() { suf.clear(); ann.clear(); }
    };
    std::vector<uint8_t> tmp_suf;
    std::vector<uint32_t> tmp_order;
    // appends the rows [row0, row0 + cnt) of the table (or the listed rows) to `rs`, sorted by memcmp of the suffix bytes
    // (write_UC layout); returns the index of the first appended row
    size_t append_rows(const uint64_t* rows, uint64_t row0, size_t cnt, int from_nt, int nbs, RowSet& rs) {
        rs.nbs = nbs;
        const size_t first = rs.size();
        tmp_suf.resize(cnt * (size_t)nbs);
        tmp_order.resize(cnt);
        for (size_t q = 0; q < cnt; q++) {
            const uint64_t r = rows ? rows[q] : row0 + q;
            if (nbs) suffix_bytes(&im.tk[r * W], from_nt, &tmp_suf[q * (size_t)nbs], nbs);
            tmp_order[q] = (uint32_t)q;
        }
        if (nbs) std::sort(tmp_order.begin(), tmp_order.end(), [&](uint32_t a, uint32_t b) { return memcmp(&tmp_suf[a * (size_t)nbs], &tmp_suf[b * (size_t)nbs], (size_t)nbs) < 0; });
        rs.suf.resize((first + cnt) * (size_t)nbs);
        for (size_t q = 0; q < cnt; q++) {
            const uint32_t o = tmp_order[q];
            if (nbs) memcpy(&rs.suf[(first + q) * (size_t)nbs], &tmp_suf[o * (size_t)nbs], (size_t)nbs);
            rs.ann.push_back(annot_of_row(rows ? rows[o] : row0 + o));
        }
        return first;
    }
    void write_block(const RowSet& rs, int header_field, bool with_header) {
        if (with_header) u16((uint16_t)header_field);
        if (rs.size() == 0) return;
        const int nbs = rs.nbs;
        size_t sa = 1;
        for (const Ann& a : rs.ann) sa = std::max<size_t>(sa, a.n);
        u16(0);  // nb_extended_annot
        i32((int32_t)sa);
        const size_t line = (size_t)nbs + sa, at = out->size();
        out->resize(at + rs.size() * line, 0);  // (zero-filled: the bytes behind a shorter annotation)
        uint8_t* o = out->data() + at;
        for (size_t q = 0; q < rs.size(); q++, o += line) {
            if (nbs) memcpy(o, &rs.suf[q * (size_t)nbs], (size_t)nbs);
            memcpy(o + nbs, rs.ann[q].p, rs.ann[q].n);
        }
    }

    void write_node(uint32_t node, int d, int flag) {
        const BftNode& nd = im.nodes[node];
        const int i = k - 9 * d;
        {   // node UC (write_Node -> write_UC, src/write_to_disk.c:84-105)
            std::vector<uint64_t> rows;
            for (uint32_t q = 0; q < nd.uc_n; q++) rows.push_back(im.ucrow[nd.uc_first + q]);
            RowSet rs;
            append_rows(rows.data(), 0, rows.size(), 9 * d, nb_bytes(i), rs);
            write_block(rs, (nd.uc_n << 1) | flag, true);
        }
        u32(nd.ncc);
        for (uint32_t c = 0; c < nd.ncc && !err; c++) write_cc(im.ccs[nd.cc_first + c], d, c + 1 == nd.ncc);
    }

    struct P { uint32_t r; uint32_t cnt; uint64_t idx; bool start; };
    // UC block b of a CC (prefixes [128 b, 128 b + 128)): the suffix rows of its groups, or on the leaf level its annotations
    void write_cc_block(const std::vector<P>& prefs, int b, int d, bool lm, bool leaf) {
        const int n = (int)prefs.size(), i = k - 9 * d;
        RowSet rs;
        if (!leaf) {
            const int nbs = nb_bytes(i - 9);
            rs.nbs = nbs;
            for (int j = b * 128; j < std::min(n, b * 128 + 128); j++) {
                if (!prefs[j].cnt) continue;
                const size_t first = append_rows(nullptr, prefs[j].idx, prefs[j].cnt, 9 * (d + 1), nbs, rs);
                if (!lm && prefs[j].start) rs.suf[first * (size_t)nbs + nbs - 1] |= 0x80;  // cluster-start flag, src/CC.c:349-352
            }
            write_block(rs, (int)rs.size(), true);
        } else {
            for (int j = b * 128; j < std::min(n, b * 128 + 128); j++) rs.ann.push_back(annot_of_row(prefs[j].idx));
            write_block(rs, 0, false);
        }
    }

    void write_cc(const BftCC& cc, int d, bool last) {
        const int i = k - 9 * d, s = cc.s, p = 18 - s, n = cc.nb_elem;
        const bool lm = level_min_of(k, i), leaf = i == 9;
        // prefixes in filter3 order, from the filter2 words and the cluster table
        std::shared_ptr<std::vector<P>> prefs_p = std::make_shared<std::vector<P>>();
        std::vector<P>& prefs = *prefs_p;
        prefs.reserve(n);
        std::vector<uint8_t> f2((size_t(1) << p) / 8, 0);
        const size_t nwords = ((size_t(1) << p) + BFT_F2_BITS_PER_WORD - 1) / BFT_F2_BITS_PER_WORD;
        uint32_t clu = 0;
        for (size_t w = 0; w < nwords; w++) {
            const uint64_t fw = im.f2w[cc.f2_off + w];
            for (int b = 0; b < BFT_F2_BITS_PER_WORD; b++) {
                if (!((fw >> b) & 1ull)) continue;
                const uint32_t pu = (uint32_t)(w * BFT_F2_BITS_PER_WORD + b);
                f2[pu >> 3] |= (uint8_t)(1u << (pu & 7));
                uint64_t e = im.clus[cc.clus_off + clu++];
                if (e & BFT_CLUS_MULTI) {
                    const uint32_t st = (uint32_t)e, len = (uint32_t)((e >> BFT_CLUS_LEN_SHIFT) & 0xFFFFu);
                    for (uint32_t q = 0; q < len; q++) {
                        const uint64_t m = im.child[cc.child_off + st + q];
                        prefs.push_back(P{(pu << s) | ((uint32_t)(m >> BFT_CHILD_PV_SHIFT) & 0xFFu), (uint32_t)(m >> BFT_CHILD_CNT_SHIFT) & 0xFFu, m & BFT_CHILD_IDX_MASK, q == 0});
                    }
                } else
                    prefs.push_back(P{(pu << s) | ((uint32_t)(e >> BFT_CHILD_PV_SHIFT) & 0xFFu), (uint32_t)(e >> BFT_CHILD_CNT_SHIFT) & 0xFFu, e & BFT_CHILD_IDX_MASK, true});
            }
        }
        if ((int)prefs.size() != n) { err = true; return; }
        bool tbyte = false;
        uint16_t nnodes = 0;
        if (!leaf)
            for (auto& pf : prefs) { if (pf.cnt >= 16) tbyte = true; if (pf.cnt == 0) nnodes++; }
        u16((uint16_t)((188u << 7) | ((tbyte ? 1u : 0u) << 6) | ((uint32_t)s << 1) | (last ? 1u : 0u)));
        u16((uint16_t)n);
        u16(nnodes);
        wr(f2.data(), f2.size());
        std::vector<uint8_t> f3(s == 8 ? n : (n + 1) / 2, 0), ex((n + 7) / 8, 0);
        for (int j = 0; j < n; j++) {
            const uint32_t pv = prefs[j].r & ((1u << s) - 1u);
            if (s == 8) f3[j] = (uint8_t)pv;
            else f3[j / 2] |= (uint8_t)((j & 1) ? (pv << 4) : pv);
            if (prefs[j].start) ex[j >> 3] |= (uint8_t)(1u << (j & 7));
        }
        wr(f3.data(), f3.size());
        if (lm) wr(ex.data(), ex.size());
        const int nbk = (n + 127) / 128;
        if (!leaf) {
            std::vector<uint8_t> ct(tbyte ? n : (n + 1) / 2, 0);
            for (int j = 0; j < n; j++) {
                if (tbyte) ct[j] = (uint8_t)prefs[j].cnt;
                else ct[j / 2] |= (uint8_t)((j & 1) ? (prefs[j].cnt << 4) : prefs[j].cnt);
            }
            wr(ct.data(), ct.size());
        }
        const bool jobs = par != nullptr && d == 0;  // the root's CCs: blocks and child Nodes are filled by the pool
        for (int b = 0; b < nbk && !err; b++) {
            if (jobs) defer([prefs_p, b, d, lm, leaf](Writer& w) { w.write_cc_block(*prefs_p, b, d, lm, leaf); });
            else write_cc_block(prefs, b, d, lm, leaf);
        }
        if (!leaf)
            for (int j = 0; j < n && !err; j++)
                if (prefs[j].cnt == 0) {  // src/insertNode.c:308-311
                    const uint32_t child = (uint32_t)prefs[j].idx;
                    const int flag = (!lm && prefs[j].start) ? 1 : 0;
                    if (jobs) defer([child, d, flag](Writer& w) { w.write_node(child, d + 1, flag); });
                    else write_node(child, d + 1, flag);
                }
    }
};

}  // namespace

void bft_annot_encode(const uint32_t* ids, uint32_t n, std::vector<uint8_t>& out) { annot_encode(ids, n, out); }

static bool file_write_impl(const char* path, const BftHostImage& im, std::string& err);
bool bft_file_write(const char* path, const BftHostImage& im, std::string& err) {
    try {
        return file_write_impl(path, im, err);
    } catch (const std::bad_alloc&) {
        err = "out of memory while writing the file";
    } catch (const std::exception& e) {
        err = e.what();
    }
    return false;
}
static bool file_write_impl(const char* path, const BftHostImage& im, std::string& err) {
    IoTrace tr;
    AnnotCache ann;
    ann.build(im);
    tr.mark("write: annotations of the colour sets encoded");
    Parts parts;
    Writer w(im, ann);
    w.par = &parts;
    w.literal_part();
    w.i32(0);  // length_comp_set_colors
    w.i32(im.r1);
    w.i32(im.r2);
    w.i32(0);  // treshold_compression
    w.i32((int32_t)im.genomes.size());
    w.i32(im.k);
    const uint8_t comp = 0;
    w.wr(&comp, 1);
    for (const std::string& g : im.genomes) {
        w.u16((uint16_t)(g.size() + 1));
        w.wr(g.c_str(), g.size() + 1);
    }
    for (int i = 9; i <= im.k; i += 9) {  // src/write_to_disk.c:76-84
        w.i32(128); w.i32(128); w.i32(128); w.i32(255);
        w.i32(level_min_of(im.k, i) ? 1 : 0);
        w.i32(BFT_MODULO_HASH);
        w.i32(BFT_TRESH_SUF_PREF);
    }
    w.write_node(0, 0, 0);
    if (w.err) { err = "inconsistent image"; return false; }
    tr.mark("write: the root's CCs walked (headers, filters; blocks and child Nodes deferred)");
    // the deferred parts: UC blocks and child-Node subtrees of the root's CCs, every thread with a Writer of its own
    // The pool fills the deferred parts (jobs are taken in file order); this thread streams the parts to the file in order as they complete --
    // the file's pages are written by ONE stream (tmpfs and most file systems serialise page allocation: 32 threads writing side by side
    // took longer than one) while the others still fill what comes behind.
    const unsigned nt = io_threads();
    std::vector<std::unique_ptr<Writer>> ws;
    for (unsigned t = 0; t < nt; t++) ws.emplace_back(new Writer(im, ann));
    std::atomic<bool> bad{false};
    const size_t nparts = parts.bufs.size(), njobs = parts.jobs.size();
    std::unique_ptr<std::atomic<uint8_t>[]> ready(new std::atomic<uint8_t>[nparts]);
    for (size_t i = 0; i < nparts; i++) ready[i].store(1);
    std::vector<uint8_t> parts_is_job(nparts, 0);
    for (size_t j = 0; j < njobs; j++) { ready[parts.job_buf[j]].store(0); parts_is_job[parts.job_buf[j]] = 1; }
    std::atomic<size_t> next{0};
    // (a job that throws -- an allocation failure -- marks its part ready and the file bad: the streaming thread never waits for it, and nothing
    // leaves a worker as an exception)
    // Back-pressure: the pool may run at most WRITE_AHEAD bytes of filled parts ahead of the one stream that writes them -- a slow file system
    // must not make the whole file pile up in memory beside the host image.  The writing thread is exempt (it fills the part it waits for
    // itself when nobody else has taken it), so the cap cannot stall the file.
    constexpr size_t WRITE_AHEAD = (size_t)1 << 30;
    std::atomic<size_t> unwritten{0};
    auto one_job = [&](Writer& x, size_t j) {
        try {
            x.out = &parts.bufs[parts.job_buf[j]];
            parts.jobs[j](x);
            if (x.err) bad = true;
        } catch (...) {
            bad = true;
        }
        unwritten.fetch_add(parts.bufs[parts.job_buf[j]].size(), std::memory_order_relaxed);
        ready[parts.job_buf[j]].store(1, std::memory_order_release);
    };
    auto work = [&](unsigned t) {
        touch_exception_state();
        for (;;) {
            while (t != 0 && unwritten.load(std::memory_order_relaxed) > WRITE_AHEAD && !bad.load() && next.load() < njobs) std::this_thread::sleep_for(std::chrono::microseconds(200));
            const size_t j = next.fetch_add(1);
            if (j >= njobs) break;
            one_job(*ws[t], j);
        }
    };
    FILE* f = nullptr;
    bool ok = false;
    {
        ThreadJoiner pool;
        for (unsigned t = 1; t < nt; t++) {
            try { pool.th.emplace_back(work, t); } catch (...) { break; }
        }
        f = fopen(path, "wb");
        ok = f != nullptr;
        if (pool.th.empty()) work(0);  // (no thread to be had: fill first, then write)
        for (size_t i = 0; i < nparts && ok && !bad; i++) {
            while (!ready[i].load(std::memory_order_acquire)) {
                // (help with a job instead of spinning, if any is left)
                const size_t j = next.fetch_add(1);
                if (j < njobs) one_job(*ws[0], j);
                else std::this_thread::yield();
            }
            std::vector<uint8_t>& b = parts.bufs[i];
            if (!b.empty() && fwrite(b.data(), 1, b.size(), f) != b.size()) ok = false;
            if (parts_is_job[i]) unwritten.fetch_sub(b.size(), std::memory_order_relaxed);
            std::vector<uint8_t>().swap(b);  // (written: its memory goes back while the rest is still being filled)
        }
        if (!ok || bad) next.store(njobs);  // (nothing more to fill)
    }
    if (f && fclose(f) != 0) ok = false;
    tr.mark("write: parts filled by the pool and streamed to the file in order");
    if (!f) { err = std::string("cannot create ") + path; return false; }
    if (bad) { err = "inconsistent image"; return false; }
    if (!ok) { err = "write error"; return false; }
    return true;
}
