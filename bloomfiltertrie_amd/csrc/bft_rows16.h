// bft_rows16.h -- a wavefront streams the colour rows of its tile, 16 bytes per lane and store (device code; shared by k_color_rows_bm16,
// bft_kernels_color.h, which reads the colour sets from an array, and k_color_rows_kh, bft_kh.hip, which looks them up itself).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CR16_UNROLL 1
#define CR16_WAVE_ROWS 1024u
#define CR16_ABSENT 0xFFFFFFFFu

// s_cs[j], j <= nt: dword offset of the dictionary row of k-mer j of the tile (CR16_ABSENT: absent; entry nt: CR16_ABSENT); tout: the tile's
// first output byte (16-byte aligned); div_m, div_l: magic number of the division by rowbytes (>= 16)
__device__ __forceinline__ void cr16_stream_tile(const uint32_t* s_cs, uint32_t nt, uint32_t rowbytes, uint32_t div_m, uint32_t div_l, const uint8_t* __restrict__ bm,
                                                 uint8_t* __restrict__ tout, uint32_t lane) {
    struct __attribute__((packed, aligned(1))) QuadU { uint32_t a, b, c, d; };
    const uint32_t total = nt * rowbytes, nch = (total + 15u) / 16u;
    // A turn = 64 x CR16_UNROLL chunks.  The loads of turn t + 1 are issued BEFORE the stores of turn t: loads and stores count on one
    // in-order counter (vmcnt), so a wavefront that stores and then loads cannot use what it loaded before its stores are acknowledged --
    // the kernel took the time of its arithmetic plus the time of its stores (0.35 ms per GB where either alone takes 0.18 / 0.23).
    struct Turn {
        uint32_t take[CR16_UNROLL];
        QuadU a[CR16_UNROLL], nx[CR16_UNROLL];
    };
    auto fetch = [&](uint32_t i0, Turn& T) {
#pragma unroll
        for (int u = 0; u < CR16_UNROLL; u++) {
            const uint32_t i = min(i0 + (uint32_t)u * 64u, nch - 1u), byte = i * 16u;
            const uint32_t t = __umulhi(byte, div_m);
            const uint32_t q = (t + ((byte - t) >> 1)) >> (div_l - 1u);  // rowbytes >= 16: div_l >= 4
            const uint32_t b = byte - q * rowbytes;
            T.take[u] = rowbytes - b;  // bytes of the chunk that belong to k-mer q (>= 16: all of it)
            const uint32_t c0 = s_cs[q];
            const uint32_t c1 = T.take[u] < 16u ? s_cs[q + 1] : CR16_ABSENT;
            T.a[u] = QuadU{0u, 0u, 0u, 0u};
            T.nx[u] = QuadU{0u, 0u, 0u, 0u};
            if (c0 != CR16_ABSENT) T.a[u] = *reinterpret_cast<const QuadU*>(bm + 4ull * c0 + b);          // (slack behind the last row)
            if (c1 != CR16_ABSENT) T.nx[u] = *reinterpret_cast<const QuadU*>(bm + 4ull * c1 - T.take[u]);  // (slack in front of the first)
        }
    };
    auto emit = [&](uint32_t i0, const Turn& T) {
#pragma unroll
        for (int u = 0; u < CR16_UNROLL; u++) {
            const uint32_t i = i0 + (uint32_t)u * 64u;
            if (i >= nch) continue;
            const uint32_t byte = i * 16u;
            const uint32_t tk = min(T.take[u], 16u);
            // bytes [0, tk) from this row, [tk, 16) from the next: per-dword masks
            const uint64_t keep_lo = tk >= 8u ? ~0ull : (1ull << (8u * tk)) - 1ull;                  // bytes 0..7
            const uint64_t keep_hi = tk >= 16u ? ~0ull : (tk <= 8u ? 0ull : (1ull << (8u * (tk - 8u))) - 1ull);  // bytes 8..15
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 o;
            o.x = (T.a[u].a & (uint32_t)keep_lo) | (T.nx[u].a & ~(uint32_t)keep_lo);
            o.y = (T.a[u].b & (uint32_t)(keep_lo >> 32)) | (T.nx[u].b & ~(uint32_t)(keep_lo >> 32));
            o.z = (T.a[u].c & (uint32_t)keep_hi) | (T.nx[u].c & ~(uint32_t)keep_hi);
            o.w = (T.a[u].d & (uint32_t)(keep_hi >> 32)) | (T.nx[u].d & ~(uint32_t)(keep_hi >> 32));
            if (byte + 16u <= total) __builtin_nontemporal_store(o, (u32x4*)(tout + byte));
            else {
                const uint32_t w[4] = {o.x, o.y, o.z, o.w};  // the last chunk of the whole batch
                for (uint32_t x = 0; byte + x < total; x++) tout[byte + x] = (uint8_t)(w[x >> 2] >> (8u * (x & 3u)));
            }
        }
    };
    constexpr uint32_t STEP = 64u * CR16_UNROLL;
    Turn A, B;
    if (lane < nch) fetch(lane, A);
    for (uint32_t i0 = lane; i0 < nch; i0 += 2u * STEP) {
        if (i0 + STEP < nch) fetch(i0 + STEP, B);
        emit(i0, A);
        if (i0 + STEP < nch) {
            if (i0 + 2u * STEP < nch) fetch(i0 + 2u * STEP, A);
            emit(i0 + STEP, B);
        }
    }
}
