// bft_merge.hip -- merging a sorted run of newly inserted k-mers into the built index.
//
// The reference inserts without bound, one k-mer at a time, into the containers themselves (insertKmers, src/insertNode.c:18-36;
// modify_annotations appends the genome id to the k-mer's annotation, src/retrieveAnnotation.c:232-314).  The bulk path keeps
// no list of (k-mer, genome) pairs around for that: the index IS the store -- the sorted distinct k-mers `tk`, a colour-set id
// per k-mer and the dictionary of colour sets -- and a build sorts only what was inserted since the last one (a "run": its own
// sorted k-mers, colour-set ids and dictionary, made by the same front end) and merges the two:
//   k_merge_search    every run k-mer finds its place among the index's k-mers (lower bound) and whether it is already there
//   k_merge_old/new   both sides scatter into the merged table; a row carries (old colour set | none, run colour set | none)
//   colour sets       the distinct (old, run) combinations that occur are listed -- old sets still in use as they are, plus the
//                     unique combinations on the run's rows (a sort of the RUN's rows only) --, each combination's genome-id list is
//                     the union of its two sorted lists, and the lists are interned by the same exact-comparison pass the first
//                     build uses (bft_intern_colors_gpu), so equal sets share one id however they came about.
// An insertion into a 2x10^8-pair index then costs a merge of its k-mers, not a re-sort of every pair, and the number of pairs an
// index can hold is not bounded by what one sort can take (the log is flushed into the index before it reaches 2^30 pairs).

#include "bft_dev.h"
#include "bft_image.h"
#include "bft_scan.h"
#include "bft_sort.h"
#include "bft_walk.h"

#define MBLK 256
// (tracing only: the stream is drained so that the mark shows the GPU time of the stage before it)
#define TRACE_STAGE(s, what) do { if (bft_trace_on()) { (void)hipStreamSynchronize(s); bft_trace_mark(what); } } while (0)
#define NONE32 0xFFFFFFFFu

namespace {

struct Scan32 {
    DevBuf tmp;
    hipStream_t s;
    explicit Scan32(hipStream_t st) : s(st) {}
    int run(const uint32_t* in, uint32_t* out, uint64_t n, uint64_t* total) {
        if (n == 0) { if (total) *total = 0; return 0; }
        CK(bft_scan::exclusive_sum_ptr<uint32_t>(in, out, n, s, tmp));
        if (total) {
            uint32_t a = 0, b = 0;
            HIPCK(hipMemcpyAsync(&a, in + n - 1, 4, hipMemcpyDeviceToHost, s));
            HIPCK(hipMemcpyAsync(&b, out + n - 1, 4, hipMemcpyDeviceToHost, s));
            HIPCK(hipStreamSynchronize(s));
            *total = (uint64_t)a + b;
        }
        return 0;
    }
};

// pos[j] = rows of the index below run row j; ins[j] = 1 when the index does not hold it
template <int W>
__global__ void k_merge_search(const uint64_t* __restrict__ tk_a, uint64_t n_a, const uint64_t* __restrict__ tk_b, uint64_t n_b, uint32_t* __restrict__ pos,
                               uint32_t* __restrict__ ins) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < n_b; j += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t t[W];
        bft_load_row<W>(tk_b + j * W, t);
        const uint32_t z = bft_rows_lower_bound<W>(tk_a, (uint32_t)n_a, t);
        bool eq = false;
        if (z < n_a) {
            uint64_t q[W];
            bft_load_row<W>(tk_a + (uint64_t)z * W, q);
            eq = bft_cmp<W>(q, t) == 0;
        }
        pos[j] = z;
        ins[j] = eq ? 0u : 1u;
    }
}
// cnt[i] = run rows inserted right before index row i (i = n_a: after the last one)
__global__ void k_merge_count(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ ins, uint64_t n_b, uint32_t* __restrict__ cnt) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < n_b; j += (uint64_t)gridDim.x * blockDim.x)
        if (ins[j]) atomicAdd(&cnt[pos[j]], 1u);
}
// index rows: row i moves to i + (insertions at or before it) = i + qex[i] + cnt[i]  (qex = exclusive scan of cnt)
template <int W>
__global__ void k_merge_old(const uint64_t* __restrict__ tk_a, const uint32_t* __restrict__ tcol_a, uint64_t n_a, const uint32_t* __restrict__ qex,
                            const uint32_t* __restrict__ cnt, uint64_t* __restrict__ tk_o, uint32_t* __restrict__ pa) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_a; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t o = i + qex[i] + cnt[i];
#pragma unroll
        for (int w = 0; w < W; w++) tk_o[o * W + w] = tk_a[i * W + w];
        pa[o] = tcol_a[i];
    }
}
// run rows: an inserted row goes to pos + (insertions before it); a row the index holds lands on that index row's new place
template <int W>
__global__ void k_merge_new(const uint64_t* __restrict__ tk_b, const uint32_t* __restrict__ tcol_b, uint64_t n_b, const uint32_t* __restrict__ pos,
                            const uint32_t* __restrict__ ins, const uint32_t* __restrict__ pex, const uint32_t* __restrict__ qex, const uint32_t* __restrict__ cnt,
                            uint64_t* __restrict__ tk_o, uint32_t* __restrict__ pa, uint32_t* __restrict__ pb, uint32_t* __restrict__ orow) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < n_b; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t z = pos[j];
        uint64_t o;
        if (ins[j]) {
            o = (uint64_t)z + pex[j];
#pragma unroll
            for (int w = 0; w < W; w++) tk_o[o * W + w] = tk_b[j * W + w];
            pa[o] = NONE32;
        } else
            o = (uint64_t)z + qex[z] + cnt[z];
        pb[o] = tcol_b[j];
        orow[j] = (uint32_t)o;
    }
}
// old colour sets that still own a row of their own (no run colour set on it)
__global__ void k_used_old(const uint32_t* __restrict__ pa, const uint32_t* __restrict__ pb, uint64_t n, uint32_t* __restrict__ used) {
    for (uint64_t o = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; o < n; o += (uint64_t)gridDim.x * blockDim.x)
        if (pb[o] == NONE32 && pa[o] != NONE32) used[pa[o]] = 1u;
}
// the combination on every run row as one sortable key
__global__ void k_pair_keys(const uint32_t* __restrict__ pa, const uint32_t* __restrict__ pb, const uint32_t* __restrict__ orow, uint64_t n_b, uint64_t* __restrict__ key,
                            uint32_t* __restrict__ iota) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < n_b; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t o = orow[j];
        key[j] = ((uint64_t)(pa[o] + 1u) << 32) | (uint64_t)pb[o];  // (none -> 0)
        iota[j] = (uint32_t)j;
    }
}
__global__ void k_pair_heads(const uint64_t* __restrict__ key_s, uint64_t n_b, uint32_t* __restrict__ head) {
    for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < n_b; q += (uint64_t)gridDim.x * blockDim.x) head[q] = (q == 0 || key_s[q] != key_s[q - 1]) ? 1u : 0u;
}
// Length of every list of U: the used old sets first (in id order), then the unique combinations (in key order).
__device__ __forceinline__ uint32_t union_len(const uint32_t* __restrict__ x, uint32_t nx, const uint32_t* __restrict__ y, uint32_t ny, uint32_t* __restrict__ out) {
    uint32_t i = 0, j = 0, n = 0;
    while (i < nx || j < ny) {
        uint32_t v;
        if (j >= ny || (i < nx && x[i] < y[j])) v = x[i++];
        else if (i >= nx || y[j] < x[i]) v = y[j++];
        else { v = x[i]; i++; j++; }
        if (out) out[n] = v;
        n++;
    }
    return n;
}
__global__ void k_u_len_old(const uint32_t* __restrict__ used, const uint32_t* __restrict__ uidx, const uint32_t* __restrict__ cs_off_a, uint64_t n_sets_a,
                            uint32_t* __restrict__ len) {
    for (uint64_t a = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; a < n_sets_a; a += (uint64_t)gridDim.x * blockDim.x)
        if (used[a]) len[uidx[a]] = cs_off_a[a + 1] - cs_off_a[a];
}
// Union of two sorted id lists by the 64 lanes of a wavefront, the shorter one (`small`, at most 64 ids: a run is a few genomes) held one
// id per lane: every small id finds its place in `big` by binary search (and whether it is there already), every big id the number of
// NEW small ids below it by a search over the lanes.  out == nullptr: only the length.  Reads and writes of `big` are coalesced.
__device__ __forceinline__ uint32_t wave_union(const uint32_t* __restrict__ big, uint32_t n_big, const uint32_t* __restrict__ small, uint32_t n_small,
                                               uint32_t* __restrict__ out, uint32_t lane) {
    const bool has = lane < n_small;
    const uint32_t sv = has ? small[lane] : 0xFFFFFFFFu;
    uint32_t lb = 0;
    bool there = false;
    if (has) {
        uint32_t lo = 0, hi = n_big;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (big[mid] < sv) lo = mid + 1; else hi = mid;
        }
        lb = lo;
        there = lo < n_big && big[lo] == sv;
    }
    const uint64_t fresh = __ballot(has && !there);
    if (!out) return n_big + (uint32_t)__builtin_popcountll(fresh);
    if (has && !there) out[lb + (uint32_t)__builtin_popcountll(fresh & ((1ull << lane) - 1ull))] = sv;
    for (uint32_t j0 = 0; j0 < n_big; j0 += 64) {  // (every lane takes part in every shuffle)
        const uint32_t j = j0 + lane;
        const uint32_t v = j < n_big ? big[j] : 0u;
        uint32_t lo = 0, hi = n_small;  // small ids below v
#pragma unroll
        for (int it = 0; it < 7; it++) {
            const uint32_t mid = min((lo + hi) >> 1, 63u);
            const uint32_t sm = __shfl(sv, mid);
            if (lo < hi) { if (sm < v) lo = mid + 1; else hi = mid; }
        }
        if (j < n_big) out[j + (uint32_t)__builtin_popcountll(lo >= 64 ? fresh : (fresh & ((1ull << lo) - 1ull)))] = v;
    }
    return n_big + (uint32_t)__builtin_popcountll(fresh);
}

// fill != nullptr: second pass, writes the union at off[u]; else the lengths.  One pair at a time per wavefront when one of its lists
// fits the lanes; a pair of two long lists is merged by its own lane.
__global__ __launch_bounds__(MBLK) void k_u_pairs(const uint64_t* __restrict__ key_s, const uint32_t* __restrict__ head, const uint32_t* __restrict__ hidx, uint64_t n_b,
                                                  uint32_t n_used, const uint32_t* __restrict__ cs_off_a, const uint32_t* __restrict__ cs_ids_a,
                                                  const uint32_t* __restrict__ cs_off_b, const uint32_t* __restrict__ cs_ids_b, uint32_t* __restrict__ len,
                                                  const uint32_t* __restrict__ off, uint32_t* __restrict__ fill) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nblk = (n_b + MBLK - 1) / MBLK;
    for (uint64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {  // whole wavefronts stay in the loop together
        const uint64_t q = blk * MBLK + threadIdx.x;
        uint32_t xs = 0, nx = 0, ys = 0, ny = 0, u = 0;
        bool hd = false;
        if (q < n_b && head[q]) {
            hd = true;
            const uint64_t key = key_s[q];
            const uint32_t a1 = (uint32_t)(key >> 32), b = (uint32_t)key;
            u = n_used + hidx[q];
            if (a1) { xs = cs_off_a[a1 - 1]; nx = cs_off_a[a1] - xs; }
            ys = cs_off_b[b];
            ny = cs_off_b[b + 1] - ys;
        }
        const bool coop = hd && (nx <= 64u || ny <= 64u);
        if (hd && !coop) {  // two long lists
            if (fill) union_len(cs_ids_a + xs, nx, cs_ids_b + ys, ny, fill + off[u]);
            else len[u] = union_len(cs_ids_a + xs, nx, cs_ids_b + ys, ny, nullptr);
        }
        uint64_t todo = __ballot(coop);
        while (todo) {
            const int t = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t xs0 = __shfl(xs, t), nx0 = __shfl(nx, t), ys0 = __shfl(ys, t), ny0 = __shfl(ny, t), u0 = __shfl(u, t);
            const bool y_small = ny0 <= 64u;
            const uint32_t* big = y_small ? cs_ids_a + xs0 : cs_ids_b + ys0;
            const uint32_t* sml = y_small ? cs_ids_b + ys0 : cs_ids_a + xs0;
            const uint32_t n = wave_union(big, y_small ? nx0 : ny0, sml, y_small ? ny0 : nx0, fill ? fill + off[u0] : nullptr, lane);
            if (!fill && lane == 0) len[u0] = n;
        }
    }
}
__global__ void k_u_fill_old(const uint32_t* __restrict__ used, const uint32_t* __restrict__ uidx, const uint32_t* __restrict__ cs_off_a,
                             const uint32_t* __restrict__ cs_ids_a, uint64_t n_sets_a, const uint32_t* __restrict__ off, uint32_t* __restrict__ fill) {
    // one wavefront per old set: the copy is coalesced
    const uint32_t lane = threadIdx.x & 63u, wpb = blockDim.x >> 6;
    for (uint64_t a = (uint64_t)blockIdx.x * wpb + (threadIdx.x >> 6); a < n_sets_a; a += (uint64_t)gridDim.x * wpb) {
        if (!used[a]) continue;
        const uint32_t s0 = cs_off_a[a], n = cs_off_a[a + 1] - s0, d0 = off[uidx[a]];
        for (uint32_t i = lane; i < n; i += 64) fill[d0 + i] = cs_ids_a[s0 + i];
    }
}
// the run rows' entry of U, per run row (through the sorted order)
__global__ void k_pair_of_row(const uint32_t* __restrict__ order, const uint32_t* __restrict__ head, const uint32_t* __restrict__ hidx, uint64_t n_b, uint32_t n_used,
                              uint32_t* __restrict__ urow) {
    for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < n_b; q += (uint64_t)gridDim.x * blockDim.x) urow[order[q]] = n_used + hidx[q] + head[q] - 1u;
}
// final colour-set id of every merged row
__global__ void k_tcol_old(const uint32_t* __restrict__ pa, const uint32_t* __restrict__ pb, uint64_t n, const uint32_t* __restrict__ uidx, const uint32_t* __restrict__ tcol_u,
                           uint32_t* __restrict__ tcol_o) {
    for (uint64_t o = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; o < n; o += (uint64_t)gridDim.x * blockDim.x)
        if (pb[o] == NONE32) tcol_o[o] = tcol_u[uidx[pa[o]]];
}
__global__ void k_tcol_new(const uint32_t* __restrict__ orow, const uint32_t* __restrict__ urow, uint64_t n_b, const uint32_t* __restrict__ tcol_u, uint32_t* __restrict__ tcol_o) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < n_b; j += (uint64_t)gridDim.x * blockDim.x) tcol_o[orow[j]] = tcol_u[urow[j]];
}
__global__ void k_count_pairs(const uint32_t* __restrict__ tcol, uint64_t n, const uint32_t* __restrict__ cs_off, unsigned long long* __restrict__ total) {
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = tcol[i];
        acc += cs_off[c + 1] - cs_off[c];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63u) == 0 && acc) atomicAdd(total, acc);
}

__global__ void k_sum32(const uint32_t* __restrict__ v, uint64_t n, unsigned long long* __restrict__ total) {
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc += v[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63u) == 0 && acc) atomicAdd(total, acc);
}

template <int W>
int merge_w(const BftRun& a, const BftRun& b, hipStream_t s, BftRunOut& out) {
    const uint64_t n_a = a.n, n_b = b.n;
#define G(n) dim3(bft_grid_for(((uint64_t)(n) + MBLK - 1) / MBLK)), dim3(MBLK), 0, s
    Scan32 scan(s);
    DevBuf pos, ins, pex, cnt, qex;
    CK(pos.alloc(n_b * 4));
    CK(ins.alloc(n_b * 4));
    CK(pex.alloc(n_b * 4));
    CK(cnt.alloc_zero((n_a + 1) * 4, s));
    CK(qex.alloc((n_a + 1) * 4));
    hipLaunchKernelGGL(k_merge_search<W>, G(n_b), a.tk, n_a, b.tk, n_b, pos.as<uint32_t>(), ins.as<uint32_t>());
    TRACE_STAGE(s, "  merge: search");
    uint64_t n_ins = 0, chk = 0;
    CK(scan.run(ins.as<uint32_t>(), pex.as<uint32_t>(), n_b, &n_ins));
    hipLaunchKernelGGL(k_merge_count, G(n_b), pos.as<uint32_t>(), ins.as<uint32_t>(), n_b, cnt.as<uint32_t>());
    CK(scan.run(cnt.as<uint32_t>(), qex.as<uint32_t>(), n_a + 1, &chk));
    if (chk != n_ins) return bft_fail(BFT_GPU_E_LIMIT, "merge self-check failed (insertion counts disagree)");
    TRACE_STAGE(s, "  merge: counts + scans");
    const uint64_t n_o = n_a + n_ins;
    if (n_o >= 0x7FFFFFFFull) return bft_fail(BFT_GPU_E_LIMIT, "more than 2^31-1 distinct k-mers");
    DevBuf pa, pb, orow;
    CK(out.tk.alloc(n_o * W * 8));
    CK(out.tcol.alloc(n_o * 4));
    CK(pa.alloc(n_o * 4));
    CK(pb.alloc(n_o * 4));
    CK(orow.alloc(n_b * 4));
    HIPCK(hipMemsetAsync(pb.p, 0xFF, n_o * 4, s));
    TRACE_STAGE(s, "  merge: outputs allocated");
    hipLaunchKernelGGL(k_merge_old<W>, G(n_a), a.tk, a.tcol, n_a, qex.as<uint32_t>(), cnt.as<uint32_t>(), out.tk.as<uint64_t>(), pa.as<uint32_t>());
    hipLaunchKernelGGL(k_merge_new<W>, G(n_b), b.tk, b.tcol, n_b, pos.as<uint32_t>(), ins.as<uint32_t>(), pex.as<uint32_t>(), qex.as<uint32_t>(), cnt.as<uint32_t>(),
                       out.tk.as<uint64_t>(), pa.as<uint32_t>(), pb.as<uint32_t>(), orow.as<uint32_t>());
    HIPCK(hipGetLastError());
    TRACE_STAGE(s, "  merge: scatter of both sides");
    pos.release(); ins.release(); pex.release(); cnt.release(); qex.release();
    bft_trace_mark("  merge: k-mers placed");

    // ---- colour sets: U = old sets still on a row of their own + the unique (old | none, run) combinations ----
    DevBuf used, uidx, key, key_s, iota, order, head, hidx, urow;
    CK(used.alloc_zero(std::max<uint64_t>(1, a.n_sets) * 4, s));
    CK(uidx.alloc(std::max<uint64_t>(1, a.n_sets) * 4));
    hipLaunchKernelGGL(k_used_old, G(n_o), pa.as<uint32_t>(), pb.as<uint32_t>(), n_o, used.as<uint32_t>());
    uint64_t n_used = 0, n_comb = 0;
    CK(scan.run(used.as<uint32_t>(), uidx.as<uint32_t>(), a.n_sets, &n_used));
    CK(key.alloc(n_b * 8));
    CK(key_s.alloc(n_b * 8));
    CK(iota.alloc(n_b * 4));
    CK(order.alloc(n_b * 4));
    CK(head.alloc(n_b * 4));
    CK(hidx.alloc(n_b * 4));
    CK(urow.alloc(n_b * 4));
    hipLaunchKernelGGL(k_pair_keys, G(n_b), pa.as<uint32_t>(), pb.as<uint32_t>(), orow.as<uint32_t>(), n_b, key.as<uint64_t>(), iota.as<uint32_t>());
    {
        int abits = 1;  // (old set + 1) sits above the run's 32 bits
        while (abits < 32 && ((a.n_sets + 1) >> abits)) abits++;
        CK((bft_rs::sort_pairs<uint64_t, uint32_t, bft_rs::SHAPE_LIGHT>(key.as<uint64_t>(), iota.as<uint32_t>(), n_b, key_s.as<uint64_t>(), order.as<uint32_t>(), 0, 32 + abits, s)));
        HIPCK(hipStreamSynchronize(s));
    }
    key.release(); iota.release();
    bft_trace_mark("  merge: (old, run) pairs sorted");
    hipLaunchKernelGGL(k_pair_heads, G(n_b), key_s.as<uint64_t>(), n_b, head.as<uint32_t>());
    CK(scan.run(head.as<uint32_t>(), hidx.as<uint32_t>(), n_b, &n_comb));
    const uint64_t n_u = n_used + n_comb;
    if (n_u >= 0x7FFFFFFFull) return bft_fail(BFT_GPU_E_LIMIT, "too many colour-set combinations in one merge");
    DevBuf ulen, uoff, uids;
    CK(ulen.alloc((n_u + 1) * 4));
    CK(uoff.alloc((n_u + 1) * 4));
    hipLaunchKernelGGL(k_u_len_old, G(a.n_sets), used.as<uint32_t>(), uidx.as<uint32_t>(), a.cs_off, a.n_sets, ulen.as<uint32_t>());
    hipLaunchKernelGGL(k_u_pairs, G(n_b), key_s.as<uint64_t>(), head.as<uint32_t>(), hidx.as<uint32_t>(), n_b, (uint32_t)n_used, a.cs_off, a.cs_ids, b.cs_off, b.cs_ids,
                       ulen.as<uint32_t>(), (const uint32_t*)nullptr, (uint32_t*)nullptr);
    HIPCK(hipMemsetAsync(ulen.as<uint32_t>() + n_u, 0, 4, s));
    uint64_t n_uids = 0, n_uids64 = 0;
    CK(scan.run(ulen.as<uint32_t>(), uoff.as<uint32_t>(), n_u + 1, &n_uids));
    {   // the scan is 32-bit like the dictionary's offsets: a total that wrapped shows against the 64-bit sum of the lengths
        DevBuf acc;
        CK(acc.alloc_zero(8, s));
        hipLaunchKernelGGL(k_sum32, G(n_u), ulen.as<uint32_t>(), n_u, acc.as<unsigned long long>());
        unsigned long long v = 0;
        HIPCK(hipMemcpyAsync(&v, acc.p, 8, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
        n_uids64 = v;
    }
    if (n_uids64 != n_uids) return bft_fail(BFT_GPU_E_LIMIT, "colour-set dictionary beyond 2^32 genome ids");
    bft_trace_mark("  merge: union lengths");
    CK(uids.alloc(std::max<uint64_t>(1, n_uids) * 4));
    {
        const uint64_t nw = (a.n_sets + 3) / 4;
        hipLaunchKernelGGL(k_u_fill_old, dim3(bft_grid_for(nw)), dim3(MBLK), 0, s, used.as<uint32_t>(), uidx.as<uint32_t>(), a.cs_off, a.cs_ids, a.n_sets, uoff.as<uint32_t>(),
                           uids.as<uint32_t>());
    }
    hipLaunchKernelGGL(k_u_pairs, G(n_b), key_s.as<uint64_t>(), head.as<uint32_t>(), hidx.as<uint32_t>(), n_b, (uint32_t)n_used, a.cs_off, a.cs_ids, b.cs_off, b.cs_ids,
                       (uint32_t*)nullptr, uoff.as<uint32_t>(), uids.as<uint32_t>());
    hipLaunchKernelGGL(k_pair_of_row, G(n_b), order.as<uint32_t>(), head.as<uint32_t>(), hidx.as<uint32_t>(), n_b, (uint32_t)n_used, urow.as<uint32_t>());
    HIPCK(hipGetLastError());
    bft_trace_mark("  merge: union lists enqueued");
    DevBuf tcol_u;
    CK(bft_intern_colors_gpu(uoff.as<uint32_t>(), uids.as<uint32_t>(), n_u, n_uids, s, tcol_u, out.cs_off, out.cs_ids, out.n_sets, out.n_ids, n_used));
    bft_trace_mark("  merge: lists interned");
    hipLaunchKernelGGL(k_tcol_old, G(n_o), pa.as<uint32_t>(), pb.as<uint32_t>(), n_o, uidx.as<uint32_t>(), tcol_u.as<uint32_t>(), out.tcol.as<uint32_t>());
    hipLaunchKernelGGL(k_tcol_new, G(n_b), orow.as<uint32_t>(), urow.as<uint32_t>(), n_b, tcol_u.as<uint32_t>(), out.tcol.as<uint32_t>());
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(s));
    out.n = n_o;
#undef G
    return 0;
}

}  // namespace

int bft_merge_runs(int W, const BftRun& a, const BftRun& b, hipStream_t s, BftRunOut& out) {
    switch (W) {
    case 1: return merge_w<1>(a, b, s, out);
    case 2: return merge_w<2>(a, b, s, out);
    case 3: return merge_w<3>(a, b, s, out);
    default: return merge_w<4>(a, b, s, out);
    }
}

// sum over the k-mers of the size of their colour set = the number of distinct (k-mer, genome) pairs the index holds
int bft_count_pairs(const uint32_t* d_tcol, uint64_t n, const uint32_t* d_cs_off, hipStream_t s, uint64_t* total) {
    *total = 0;
    if (n == 0) return 0;
    DevBuf acc;
    CK(acc.alloc_zero(8, s));
    hipLaunchKernelGGL(k_count_pairs, dim3(bft_grid_for((n + MBLK - 1) / MBLK)), dim3(MBLK), 0, s, d_tcol, n, d_cs_off, acc.as<unsigned long long>());
    unsigned long long v = 0;
    HIPCK(hipMemcpyAsync(&v, acc.p, 8, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    *total = v;
    return 0;
}
