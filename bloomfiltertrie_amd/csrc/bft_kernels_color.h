// bft_kernels_color.h -- colour-set retrieval: id lists (k_color_counts / k_color_fill), bitmap dictionary and fixed-width rows (k_cs_bitmaps, k_color_rows_bm, k_color_rows), k_row_colorsets
// Device code of libbft_gpu.so, included by bft_gpu.hip only (one translation unit: the kernels are templates launched from
// the host code there).
#pragma once
#include "bft_rows16.h"
#define CS_BM_SLACK 32u  // zero bytes in front of and behind the bitmap dictionary (bft_gpu.hip, ensure_cs_bitmaps)
__global__ void k_color_counts(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ tcol, const uint32_t* __restrict__ cs_off,
                               uint64_t n, uint64_t* __restrict__ counts) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = rows[i];
        uint64_t c = 0;
        if (r != BFT_ABSENT_ROW) {
            const uint32_t cs = tcol[r];
            c = cs_off[cs + 1] - cs_off[cs];
        }
        counts[i] = c;
    }
}

__global__ void k_color_fill(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ tcol, const uint32_t* __restrict__ cs_off,
                             const void* __restrict__ cs_ids, uint32_t cs_w, const uint64_t* __restrict__ offsets, uint64_t n, uint32_t* __restrict__ ids) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = rows[i];
        if (r == BFT_ABSENT_ROW) continue;
        const uint32_t cs = tcol[r];
        const uint32_t a = cs_off[cs], b = cs_off[cs + 1];
        uint64_t o = offsets[i];
        for (uint32_t q = a; q < b; q++) ids[o++] = bft_cs_id_at(cs_ids, cs_w, q);
    }
}

// ---- id lists out of colour-set ids (what the k-mer hash hands out with emit_cs: no row, no sorted table) ----
// length of k-mer i's id list as the input "iterator" of the offsets' scan (entry n: 0, so that offsets[n] = the total)
struct BftCsLen {
    const uint32_t* cs;  // colour-set id per k-mer, BFT_ABSENT_ROW for an absent one
    const uint32_t* cs_off;
    uint64_t n;
    __host__ __device__ uint64_t operator()(uint64_t i) const {
        if (i >= n) return 0;
        const uint32_t c = cs[i];
        return c == BFT_ABSENT_ROW ? 0ull : (uint64_t)(cs_off[c + 1] - cs_off[c]);
    }
};
// The lists of 64 consecutive k-mers are one contiguous range of `ids`: the wavefront streams that range -- every output element finds its k-mer
// by a search over the lanes' start offsets (6 shuffles) and copies its id out of the dictionary: coalesced stores, whatever the lists' lengths.
// offsets[n] beyond ids_cap: nothing is written (the caller reads *needed).
__global__ __launch_bounds__(256) void k_color_fill_cs(const uint32_t* __restrict__ cs, const uint32_t* __restrict__ cs_off, const void* __restrict__ cs_ids, uint32_t cs_w,
                                                        const uint64_t* __restrict__ offsets, uint64_t n, uint64_t ids_cap, uint32_t* __restrict__ ids,
                                                        uint64_t* __restrict__ needed) {
    const uint64_t total = offsets[n];
    if (blockIdx.x == 0 && threadIdx.x == 0 && needed) *needed = total;
    if (total > ids_cap || !ids) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nblk = (n + 255) / 256;
    for (uint64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const uint64_t i = blk * 256 + threadIdx.x;
        const bool valid = i < n;
        const uint64_t a = offsets[valid ? i : n];
        uint64_t b = __shfl_down(a, 1);  // (the next lane's start is this lane's end: one load of the offsets per k-mer, not two)
        if (lane == 63u) b = offsets[valid ? i + 1 : n];
        uint32_t src = 0;
        if (valid && b > a) src = cs_off[cs[i]];
        const uint64_t base = __shfl(a, 0), end = __shfl(b, 63);
        // (positions relative to the wavefront's first: 64 lists of < 2^24 ids each fit 32 bits -- one shuffle per step of the search, not two)
        const uint32_t ar = (uint32_t)(a - base), span = (uint32_t)(end - base);
        for (uint32_t r0 = 0; r0 < span; r0 += 64) {
            const uint32_t r = r0 + lane;
            const bool in = r < span;
            const uint32_t rr = in ? r : span - 1;
            uint32_t lo = 0, hi = 63;  // the last lane whose list starts at or before rr (starts are non-decreasing over the lanes)
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                const uint32_t am = __shfl(ar, mid);
                if (am <= rr) lo = mid; else hi = mid - 1;
            }
            const uint32_t as = __shfl(ar, lo);
            const uint32_t ss = __shfl(src, lo);
            if (in) ids[base + r] = bft_cs_id_at(cs_ids, cs_w, ss + (r - as));
        }
    }
}

// colour-set dictionary as bitmaps, built once per image: one row per set, CEIL(G/8) bytes padded to a multiple of 4
// (`stride`) so that the row kernel reads it with aligned dword loads
__global__ void k_cs_bitmaps(const uint32_t* __restrict__ cs_off, const void* __restrict__ cs_ids, uint32_t cs_w, uint64_t n_sets, uint32_t stride,
                             uint8_t* __restrict__ bm) {
    // one thread per set (a wavefront-cooperative fill with atomic ORs on the row dwords measured 2x slower)
    for (uint64_t c = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; c < n_sets; c += (uint64_t)gridDim.x * blockDim.x) {
        uint8_t* o = bm + c * stride;
        for (uint32_t q = cs_off[c]; q < cs_off[c + 1]; q++) {
            const uint32_t g = bft_cs_id_at(cs_ids, cs_w, q);
            o[g >> 3] |= (uint8_t)(1u << (g & 7));
        }
    }
}

// bytes [b, b+4) of a dictionary row (dword-aligned base; the bytes past the row's end are whatever follows: callers mask)
__device__ __forceinline__ uint32_t bm_dword_at(const uint32_t* __restrict__ row, uint32_t b) {
    const uint32_t lo = row[b >> 2];
    const uint32_t sh = 8u * (b & 3u);
    if (sh == 0) return lo;
    return (lo >> sh) | (row[(b >> 2) + 1] << (32u - sh));
}

// One output dword at tile-relative byte offset `byte` (a multiple of 4), which starts at byte b of k-mer q's row and may
// straddle two or more rows: the bytes come from the bitmap row of each present k-mer (colour sets of the tile: cs[]).
__device__ __forceinline__ uint32_t color_dword_cs(const uint32_t* cs, const uint8_t* __restrict__ bm, uint32_t stride, uint32_t rowbytes, uint32_t nt,
                                                   uint32_t q, uint32_t b) {
    uint32_t v = 0, have = 0;  // bytes of the output dword filled so far
    while (have < 4u && q < nt) {
        const uint32_t take = min(4u - have, rowbytes - b);
        const uint32_t c = cs[q];
        if (c != BFT_ABSENT_ROW) {
            uint32_t w = bm_dword_at((const uint32_t*)(bm + (uint64_t)c * stride), b);
            if (take < 4u) w &= (1u << (8u * take)) - 1u;
            v |= w << (8u * have);
        }
        have += take;
        b = 0;
        q++;
    }
    return v;
}

// Colour rows from the bitmap dictionary, row-cooperative (retrieveAnnotation for a batch: src/annotation.c:2134-2144 decodes one
// bitmap annotation, the loop of src/file_io.c:744-765 one row per k-mer).  The rows of consecutive k-mers are contiguous
// (CEIL(G/8) bytes each), so a workgroup owns a TILE of `tile_rows` consecutive k-mers = one contiguous, dword-aligned stretch of
// the output:
//   (before) k_row_colorsets turns the row of every k-mer into its colour-set id, in place (one gather per k-mer, its own launch);
//   step A  the tile's colour-set ids: one coalesced load into LDS;
//   step B  every thread produces aligned output dwords of the stretch: k-mer = byte / rowbytes (multiply-high by the host's
//           magic number), dictionary row from LDS, one or two aligned source dwords + a funnel shift, one coalesced 4-byte store;
//           CR_UNROLL independent dwords per thread are in flight at a time (a 32 KiB tile = two rounds of a 256-thread workgroup) --
//           no dependent global load is left in the loop.
// The first version resolved row -> colour set -> dictionary row per output DWORD (three dependent loads per 4 bytes) and was
// latency-bound at 1.2 TB/s written; dwords that straddle two rows (rowbytes % 4 != 0) go through color_dword.
#define CR_UNROLL 8
#define CR_MAX_TILE_ROWS 2048
// WIDE: rowbytes >= 4 -- an output dword touches at most two rows, handled without a branch (the second row's first dword is
// loaded only by the lanes that straddle).  Rows of 1-3 bytes (<= 24 genomes) take the generic per-byte path.
// The straddling dwords matter: with 250-byte rows one lane in 62 straddles, i.e. nearly every wavefront holds one, and a
// branchy slow path there was executed by every wavefront for every dword (measured: ~250 VALU instructions per output dword,
// 1.2 TB/s written whatever was done to the loads and stores).
template <bool WIDE>
__global__ __launch_bounds__(256) void k_color_rows_bm(const uint32_t* __restrict__ csid, const uint8_t* __restrict__ bm, uint32_t stride, uint64_t n,
                                                       uint32_t rowbytes, uint32_t tile_rows, uint32_t div_m, uint32_t div_l, uint8_t* __restrict__ out) {
    __shared__ uint32_t s_cs[CR_MAX_TILE_ROWS + 1];  // colour set of each k-mer of the tile (BFT_ABSENT_ROW: absent)
    const uint64_t ntiles = (n + tile_rows - 1) / tile_rows;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t q0 = tile * tile_rows;
        const uint32_t nt = (uint32_t)min((uint64_t)tile_rows, n - q0);  // k-mers of this tile
        // (WIDE: the dictionary row as a dword offset -- the dictionary stays below 4 GiB -- so that the loop needs no 64-bit multiply)
        for (uint32_t j = threadIdx.x; j <= nt; j += blockDim.x) {
            const uint32_t c = j < nt ? csid[q0 + j] : BFT_ABSENT_ROW;
            s_cs[j] = (WIDE && c != BFT_ABSENT_ROW) ? c * (stride >> 2) : c;
        }
        __syncthreads();
        const uint32_t* bmw = (const uint32_t*)bm;
        const uint32_t total = nt * rowbytes, ndw = (total + 3u) / 4u;
        uint8_t* tout = out + q0 * rowbytes;  // dword aligned: tile_rows is a multiple of 4
        for (uint32_t i0 = threadIdx.x; i0 < ndw; i0 += blockDim.x * CR_UNROLL) {
            if (WIDE) {
                uint32_t b[CR_UNROLL], c0[CR_UNROLL], lo[CR_UNROLL], hi[CR_UNROLL], nx[CR_UNROLL];
#pragma unroll
                for (int u = 0; u < CR_UNROLL; u++) {
                    const uint32_t i = min(i0 + (uint32_t)u * blockDim.x, ndw - 1u), byte = i * 4u;
                    const uint32_t t = __umulhi(byte, div_m);
                    const uint32_t q = div_l ? (t + ((byte - t) >> 1)) >> (div_l - 1u) : byte;
                    b[u] = byte - q * rowbytes;
                    c0[u] = s_cs[q];
                    const uint32_t c1 = b[u] + 4u > rowbytes ? s_cs[q + 1] : BFT_ABSENT_ROW;  // straddles into the next k-mer's row
                    const uint32_t* src = bmw + (c0[u] != BFT_ABSENT_ROW ? c0[u] : 0u) + (b[u] >> 2);
                    struct __attribute__((packed, aligned(4))) Pair { uint32_t a, b; };  // ONE 8-byte load at 4-byte alignment
                    const Pair pr = *reinterpret_cast<const Pair*>(src);                  // (the dictionary carries slack behind its last row)
                    lo[u] = pr.a;
                    hi[u] = pr.b;
                    nx[u] = c1 != BFT_ABSENT_ROW ? bmw[c1] : 0u;
                }
#pragma unroll
                for (int u = 0; u < CR_UNROLL; u++) {
                    const uint32_t i = i0 + (uint32_t)u * blockDim.x;
                    if (i >= ndw) continue;
                    const uint32_t byte = i * 4u, sh = 8u * (b[u] & 3u), take = rowbytes - b[u];
                    uint32_t v = sh ? (lo[u] >> sh) | (hi[u] << (32u - sh)) : lo[u];
                    if (c0[u] == BFT_ABSENT_ROW) v = 0u;
                    if (take < 4u) v = (v & ((1u << (8u * take)) - 1u)) | (nx[u] << (8u * take));
                    if (byte + 4u <= total) __builtin_nontemporal_store(v, (uint32_t*)(tout + byte));  // written once, read by nobody here
                    else
                        for (uint32_t x = 0; byte + x < total; x++) tout[byte + x] = (uint8_t)(v >> (8u * x));  // the last dword of the whole batch
                }
            } else {
#pragma unroll 2
                for (int u = 0; u < CR_UNROLL; u++) {
                    const uint32_t i = i0 + (uint32_t)u * blockDim.x;
                    if (i >= ndw) continue;
                    const uint32_t byte = i * 4u;
                    uint32_t q = byte;
                    if (div_l) {
                        const uint32_t t = __umulhi(byte, div_m);
                        q = (t + ((byte - t) >> 1)) >> (div_l - 1u);
                    }
                    const uint32_t w = color_dword_cs(s_cs, bm, stride, rowbytes, nt, q, byte - q * rowbytes);
                    if (byte + 4u <= total) *(uint32_t*)(tout + byte) = w;
                    else
                        for (uint32_t x = 0; byte + x < total; x++) tout[byte + x] = (uint8_t)(w >> (8u * x));
                }
            }
        }
        __syncthreads();
    }
}

// The same rows, 16 output bytes per lane (rowbytes >= 16, `out` 16-byte aligned, tile_rows a multiple of 16: tiles start 16-byte
// aligned): 16-byte non-temporal stores at aligned addresses.  The chunk starts at byte b of k-mer q's row: ONE unaligned 16-byte load at
// that byte of the dictionary row (the hardware takes unaligned loads, and loads are not what this kernel waits for); a chunk that straddles
// into the next k-mer's row (one in rowbytes / 16) takes that row's first bytes by a second unaligned load that starts `take` bytes BEFORE
// the row -- the bytes land where they belong -- and the two are merged under a byte mask: no funnel shifts, no splicing by selects
// (round 3's form: five aligned source dwords, four funnel shifts, a 128-bit byte shift by selects and 64-bit shifts: 0.18 ms of arithmetic per
// GB written beside 0.23 ms of stores).
// Measured on config 5 (250-byte rows out of a 75 MB dictionary, 1.016 GB written per 4x10^6 k-mers; rocprofv3 kernel trace,
// profiles/r04/config5_kernel_stats.txt): 0.259 ms = 3.9 TB/s written = 0.85 of a plain fill (4.55-4.66 TB/s, tools/microbench/stream.hip).
// The kernel's data comes out of a table that does not fit the L2: `expand_nt` of the same microbenchmark -- every 16 bytes written are read
// from a uniformly random 256-byte row of a 75 MiB source -- writes 3.49 TB/s (5.0 from a 4 MiB source); the real batch does better than
// that because popular colour sets repeat.  On the way here (per GB written): workgroup tiles with barriers 0.36 ms (a barrier waits for
// the wavefront's stores); loads and stores share one in-order counter, so the next turn's loads are issued ahead of this turn's stores
// (0.33); one chunk per lane and turn = eight wavefronts per SIMD instead of five (0.28); the funnel shifts and 128-bit splices of round 3's
// form replaced by unaligned loads, 49 registers (0.255).  Row-by-row unaligned STORES: 0.43.  Tiles of 2 ... 256 KB: no difference.
__global__ __launch_bounds__(256) void k_color_rows_bm16(const uint32_t* __restrict__ csid, const uint8_t* __restrict__ bm, uint32_t stride, uint64_t n,
                                                         uint32_t rowbytes, uint32_t tile_rows, uint32_t div_m, uint32_t div_l, uint8_t* __restrict__ out) {
    // The tiles belong to WAVEFRONTS (tile_rows k-mers each, at most CR16_WAVE_ROWS): a wavefront fills its own slice of LDS with the
    // dictionary rows of its tile and writes the tile's bytes, 64 x CR16_UNROLL chunks per turn, without ever meeting the other wavefronts of
    // its workgroup.  (Tiles of a workgroup, with a barrier on either side, made every tile wait for its own stores to land -- a barrier
    // waits for all of a wavefront's memory operations -- and for the next tile's colour-set ids to arrive: the kernel took the time of its
    // arithmetic PLUS the time of its stores, 0.35 ms per GB written where either alone takes 0.18 / 0.23.)
    __shared__ uint32_t s_cs_all[4][CR16_WAVE_ROWS + 1];  // dictionary row (dword offset) of each k-mer of the wavefront's tile (BFT_ABSENT_ROW: absent)
    const uint64_t ntiles = (n + tile_rows - 1) / tile_rows;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* const s_cs = s_cs_all[wave];
    for (uint64_t tile = (uint64_t)blockIdx.x * 4u + wave; tile < ntiles; tile += (uint64_t)gridDim.x * 4u) {
        const uint64_t q0 = tile * tile_rows;
        const uint32_t nt = (uint32_t)min((uint64_t)tile_rows, n - q0);
        __builtin_amdgcn_wave_barrier();  // (the last turn's reads of s_cs come before these writes: same wavefront, in order)
        for (uint32_t j = lane; j <= nt; j += 64u) {
            const uint32_t c = j < nt ? csid[q0 + j] : BFT_ABSENT_ROW;
            s_cs[j] = c != BFT_ABSENT_ROW ? c * (stride >> 2) : c;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        cr16_stream_tile(s_cs, nt, rowbytes, div_m, div_l, bm, out + q0 * rowbytes, lane);
    }
}

// colour-set id of every located k-mer (BFT_ABSENT_ROW stays BFT_ABSENT_ROW)
__global__ void k_row_colorsets(const uint32_t* rows, const uint32_t* __restrict__ tcol, uint64_t n, uint32_t* out) {  // out may be rows (in place)
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = rows[i];
        out[i] = r == BFT_ABSENT_ROW ? BFT_ABSENT_ROW : tcol[r];
    }
}

__global__ void k_color_rows(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ tcol, const uint32_t* __restrict__ cs_off,
                             const void* __restrict__ cs_ids, uint32_t cs_w, uint64_t n, uint32_t rowbytes, uint8_t* __restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint8_t* o = out + i * rowbytes;
        for (uint32_t b = 0; b < rowbytes; b++) o[b] = 0;
        const uint32_t r = rows[i];
        if (r == BFT_ABSENT_ROW) continue;
        const uint32_t cs = tcol[r];
        for (uint32_t q = cs_off[cs]; q < cs_off[cs + 1]; q++) {
            const uint32_t gid = bft_cs_id_at(cs_ids, cs_w, q);
            o[gid >> 3] |= (uint8_t)(1u << (gid & 7));
        }
    }
}
