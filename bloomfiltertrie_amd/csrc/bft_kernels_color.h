// bft_kernels_color.h -- colour-set retrieval: id lists (k_color_counts / k_color_fill), bitmap dictionary and fixed-width rows (k_cs_bitmaps, k_color_rows_bm, k_color_rows), k_row_colorsets
// Device code of libbft_gpu.so, included by bft_gpu.hip only (one translation unit: the kernels are templates launched from
// the host code there).
#pragma once
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
                             const uint32_t* __restrict__ cs_ids, const uint64_t* __restrict__ offsets, uint64_t n, uint32_t* __restrict__ ids) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = rows[i];
        if (r == BFT_ABSENT_ROW) continue;
        const uint32_t cs = tcol[r];
        const uint32_t a = cs_off[cs], b = cs_off[cs + 1];
        uint64_t o = offsets[i];
        for (uint32_t q = a; q < b; q++) ids[o++] = cs_ids[q];
    }
}

// colour-set dictionary as bitmaps, built once per image: one row per set, CEIL(G/8) bytes padded to a multiple of 4
// (`stride`) so that the row kernel reads it with aligned dword loads
__global__ void k_cs_bitmaps(const uint32_t* __restrict__ cs_off, const uint32_t* __restrict__ cs_ids, uint64_t n_sets, uint32_t stride,
                             uint8_t* __restrict__ bm) {
    // one thread per set (a wavefront-cooperative fill with atomic ORs on the row dwords measured 2x slower)
    for (uint64_t c = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; c < n_sets; c += (uint64_t)gridDim.x * blockDim.x) {
        uint8_t* o = bm + c * stride;
        for (uint32_t q = cs_off[c]; q < cs_off[c + 1]; q++) o[cs_ids[q] >> 3] |= (uint8_t)(1u << (cs_ids[q] & 7));
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
// straddle two or more rows: the bytes come from the bitmap row of each present k-mer.
__device__ __forceinline__ uint32_t color_dword(const uint32_t* __restrict__ trow, const uint32_t* __restrict__ tcol, const uint8_t* __restrict__ bm,
                                                uint32_t stride, uint32_t rowbytes, uint32_t nt, uint32_t q, uint32_t b) {
    uint32_t v = 0, have = 0;  // bytes of the output dword filled so far
    while (have < 4u && q < nt) {
        const uint32_t take = min(4u - have, rowbytes - b);
        const uint32_t r = trow[q];
        if (r != BFT_ABSENT_ROW) {
            uint32_t w = bm_dword_at((const uint32_t*)(bm + (uint64_t)tcol[r] * stride), b);
            if (take < 4u) w &= (1u << (8u * take)) - 1u;
            v |= w << (8u * have);
        }
        have += take;
        b = 0;
        q++;
    }
    return v;
}

// Colour rows from the bitmap dictionary.  The rows of consecutive k-mers are contiguous (CEIL(G/8) bytes each); one thread
// writes aligned dwords of that stream (coalesced 4-byte stores), a wavefront covering 256 consecutive bytes, i.e. mostly one
// row: its lanes read consecutive dwords of the same dictionary row (one or two aligned loads + a funnel shift each).
// Every thread works on CR_UNROLL dwords (one per grid stride) at a time, stage by stage (row index -> colour set -> bitmap
// dwords).  Dwords that straddle rows go through color_dword.  Measured (config 5, 250-byte rows, 10^9 bytes out): 0.83 ms
// = 1.2 TB/s written; byte gathers from unpadded dictionary rows took 1.2 ms; 16-byte chunks per thread were slower (a
// wavefront then touches four dictionary rows per load instruction), more chains in flight per thread changed nothing,
// non-temporal stores neither; without the dictionary reads or without the stores the kernel is only 17 % faster either
// way; 8 bytes per thread (three source dwords, one 8-byte store) was 30 % slower again, like the 16-byte variant.  blockIdx.y selects a tile of `tile_rows` k-mers (a multiple of 4, tile bytes < 2^31) so that offsets inside
// a tile are 32-bit and byte / rowbytes is a multiply-high by the host's magic number (div_m, div_l; exact on u32).
#define CR_UNROLL 4
__global__ void k_color_rows_bm(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ tcol, const uint8_t* __restrict__ bm, uint32_t stride,
                                uint64_t n, uint32_t rowbytes, uint32_t tile_rows, uint32_t div_m, uint32_t div_l, uint8_t* __restrict__ out) {
    const uint64_t q0 = (uint64_t)blockIdx.y * tile_rows;
    const uint32_t nt = (uint32_t)min((uint64_t)tile_rows, n - q0);  // k-mers of this tile
    const uint32_t total = nt * rowbytes, ndw = (total + 3u) / 4u;
    const uint32_t* trow = rows + q0;
    uint8_t* tout = out + q0 * rowbytes;
    const uint32_t G = gridDim.x * blockDim.x;
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < ndw; i0 += G * CR_UNROLL) {
        uint32_t q[CR_UNROLL], b[CR_UNROLL], r[CR_UNROLL], lo[CR_UNROLL], hi[CR_UNROLL];
        const uint32_t* src[CR_UNROLL];
        bool ok[CR_UNROLL], fast[CR_UNROLL];
#pragma unroll
        for (int u = 0; u < CR_UNROLL; u++) {
            const uint32_t i = i0 + (uint32_t)u * G;
            ok[u] = i < ndw;
            const uint32_t byte = i * 4u;
            q[u] = byte;
            if (div_l) {
                const uint32_t t = __umulhi(byte, div_m);
                q[u] = (t + ((byte - t) >> 1)) >> (div_l - 1u);
            }
            b[u] = byte - q[u] * rowbytes;
            fast[u] = ok[u] && b[u] + 4u <= rowbytes && byte + 4u <= total;
        }
        // unconditional loads on clamped indices (every array has slack behind it): the compiler issues each stage's
        // CR_UNROLL loads back to back
#pragma unroll
        for (int u = 0; u < CR_UNROLL; u++) r[u] = trow[min(q[u], nt - 1u)];
#pragma unroll
        for (int u = 0; u < CR_UNROLL; u++) src[u] = (const uint32_t*)(bm + (uint64_t)tcol[r[u] != BFT_ABSENT_ROW ? r[u] : 0u] * stride) + (b[u] >> 2);
#pragma unroll
        for (int u = 0; u < CR_UNROLL; u++) {
            lo[u] = src[u][0];
            hi[u] = src[u][1];
        }
#pragma unroll
        for (int u = 0; u < CR_UNROLL; u++)
            if (r[u] == BFT_ABSENT_ROW) lo[u] = hi[u] = 0u;
#pragma unroll
        for (int u = 0; u < CR_UNROLL; u++) {
            if (!ok[u]) continue;
            const uint32_t byte = (i0 + (uint32_t)u * G) * 4u;
            if (fast[u]) {
                const uint32_t sh = 8u * (b[u] & 3u);
                *(uint32_t*)(tout + byte) = sh ? (lo[u] >> sh) | (hi[u] << (32u - sh)) : lo[u];
            } else {
                const uint32_t w = color_dword(trow, tcol, bm, stride, rowbytes, nt, q[u], b[u]);
                if (byte + 4u <= total) *(uint32_t*)(tout + byte) = w;
                else
                    for (uint32_t x = 0; byte + x < total; x++) tout[byte + x] = (uint8_t)(w >> (8u * x));
            }
        }
    }
}

// colour-set id of every located k-mer (BFT_ABSENT_ROW stays BFT_ABSENT_ROW)
__global__ void k_row_colorsets(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ tcol, uint64_t n, uint32_t* __restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = rows[i];
        out[i] = r == BFT_ABSENT_ROW ? BFT_ABSENT_ROW : tcol[r];
    }
}

__global__ void k_color_rows(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ tcol, const uint32_t* __restrict__ cs_off,
                             const uint32_t* __restrict__ cs_ids, uint64_t n, uint32_t rowbytes, uint8_t* __restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint8_t* o = out + i * rowbytes;
        for (uint32_t b = 0; b < rowbytes; b++) o[b] = 0;
        const uint32_t r = rows[i];
        if (r == BFT_ABSENT_ROW) continue;
        const uint32_t cs = tcol[r];
        for (uint32_t q = cs_off[cs]; q < cs_off[cs + 1]; q++) {
            const uint32_t gid = cs_ids[q];
            o[gid >> 3] |= (uint8_t)(1u << (gid & 7));
        }
    }
}
