// bft_kh_dev.h -- device-side pieces of the k-mer hash lookup shared by the kernels of bft_kh.hip and the walk that looks plain root groups
// up in the table (bft_walkh.hip): the quad's line fetch into LDS and the scan of a line there.  Device code only.
#pragma once
#include <hip/hip_runtime.h>

#include "bft_image.h"
#include "bft_walk.h"

__device__ __forceinline__ uint32_t quad_bcast(uint32_t v, int j) {  // the value of lane j of the quad, on all four of its lanes
    switch (j) {
    case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xF, 0xF, true);
    case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xF, 0xF, true);
    case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xF, 0xF, true);
    default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xF, 0xF, true);
    }
}

// The lines `home` of the four lanes of a quad, fetched together: one after the other, every lane loads ITS quarter (16 bytes) of the line of
// the quad's lane r and puts it where it belongs in the wavefront's LDS block (64 lines, 80 bytes apart: headers and quarters then spread
// over the banks); every lane then reads its own line's header from there, and the body of the slot whose header field matches.  Four load
// instructions of 16 bytes a lane bring four whole lines -- the instructions a wavefront issues per line are what a kernel of random gathers
// pays for (bft_image.h) -- and the line never passes through sixty-four selects on its way (round 4 had it gathered by DPP broadcasts into
// registers and scanned slot by slot there: 430 vector instructions per k-mer of a loop of 800, which had become the limit).
// Every lane of the wavefront calls this; a lane that is not `live` asks for nothing and finds zeros.
#define BFT_KH_LDS_LINE 5u  // uint4s a line takes in LDS (4 + 1 of padding)
__device__ __forceinline__ void kh_fetch_quad(const BftImage& im, uint64_t home, bool live, uint4* wave_lines) {
    const uint32_t lane = threadIdx.x & 63u, ql = lane & 3u, q0 = lane & ~3u;
    uint4 v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t lo = quad_bcast((uint32_t)home, r), hi = quad_bcast((uint32_t)(home >> 32), r), lv = quad_bcast(live ? 1u : 0u, r);
        v[r] = make_uint4(0, 0, 0, 0);
        if (lv) v[r] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(im.kh_lines) + ((((uint64_t)hi << 32) | lo) * 64ull) + 16u * ql);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) wave_lines[(q0 + (uint32_t)r) * BFT_KH_LDS_LINE + ql] = v[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// body of slot s of the line at `mine` in LDS: W words from byte 16 + s wb on (aligned 8-byte reads, shifted)
template <int W, int S>
__device__ __forceinline__ void kh_lds_body(const uint4* mine, uint32_t s, uint64_t* body) {
    constexpr uint32_t wb = 48u / (uint32_t)S;
    static_assert(((16u + ((uint32_t)S - 1u) * wb) & ~7u) + 8u * ((uint32_t)W + 1u) <= 16u * BFT_KH_LDS_LINE, "the body reads of the last slot stay inside the line's LDS bytes");
    const uint32_t off = 16u + s * wb, sh = (off & 7u) * 8u;
    const uint64_t* p = reinterpret_cast<const uint64_t*>(reinterpret_cast<const uint8_t*>(mine) + (off & ~7u));
    uint64_t w[W + 1];
#pragma unroll
    for (int i = 0; i <= W; i++) w[i] = p[i];  // (never beyond the line's 80 bytes: bft_kh_has_kernels)
#pragma unroll
    for (int i = 0; i < W; i++) body[i] = sh ? (w[i] >> sh) | (w[i + 1] << (64u - sh)) : w[i];
}
// the slots of the line whose header field agrees with `field` under `keep`, and that are in use; *full = no slot is free
template <int S>
__device__ __forceinline__ uint32_t kh_lds_candidates(const uint4* mine, uint32_t field, uint32_t keep, bool* full) {
    constexpr uint32_t f = bft_kh_field_bits((uint32_t)S), fmask = f < 32u ? (1u << f) - 1u : 0xFFFFFFFFu;
    const uint4 h = mine[0];
    const uint32_t hw[5] = {h.x, h.y, h.z, h.w, 0u};
    const uint32_t occ = hw[3] >> (32u - (uint32_t)S);
    uint32_t cand = 0;
#pragma unroll
    for (uint32_t s = 0; s < (uint32_t)S; s++) {
        constexpr uint32_t dummy = 0; (void)dummy;
        const uint32_t o = s * f, wi = o >> 5, sh = o & 31u;
        const uint32_t fld = (uint32_t)((((uint64_t)hw[wi + 1] << 32) | hw[wi]) >> sh) & fmask;
        cand |= (((fld ^ field) & keep) == 0 ? 1u : 0u) << s;
    }
    *full = occ == (1u << (uint32_t)S) - 1u;
    return cand & occ;
}
// the k-mer `key` in the line at `mine` (LDS), d lines past its home: 1 = found (*val), 0 = not here and a slot is free (absent), -1 = not here, line full
template <int W, int S>
__device__ __forceinline__ int kh_lds_scan(const BftImage& im, const uint4* mine, const BftKhKey<W>& key, uint32_t d, uint32_t* val) {
    const uint32_t cb = im.kh.cb;
    const uint64_t vmask = (1ull << cb) - 1ull;
    bool full;
    uint32_t cand = kh_lds_candidates<S>(mine, (uint32_t)key.field, 0xFFFFFFFFu, &full);
    int found = 0;
    while (cand) {  // (a second candidate: two slots whose keys share their low f bits -- once in thousands of lines)
        const uint32_t s = (uint32_t)__builtin_ctz(cand);
        cand &= cand - 1u;
        uint64_t body[W];
        kh_lds_body<W, S>(mine, s, body);
        bool same = ((body[0] ^ (key.body[0] | ((uint64_t)d << cb))) & key.bmask[0]) == 0 && (body[0] & vmask) != 0;  // (value 0: a tombstone)
#pragma unroll
        for (int i = 1; i < W; i++) same = same && ((body[i] ^ key.body[i]) & key.bmask[i]) == 0;
        if (same) { *val = (uint32_t)(body[0] & vmask) - 1u; found = 1; }
    }
    return found ? 1 : (full ? -1 : 0);
}
// how many k-mers of the family (bft_kh_family) the line holds; -1: the line is full (the family may go on behind it), else 0
template <int W, int S>
__device__ __forceinline__ int kh_lds_count(const BftImage& im, const uint4* mine, const BftKhKey<W>& key, const BftKhFamily<W>& fam, uint32_t d, int* count) {
    const uint32_t cb = im.kh.cb;
    const uint64_t vmask = (1ull << cb) - 1ull;
    bool full;
    uint32_t cand = kh_lds_candidates<S>(mine, (uint32_t)key.field, (uint32_t)fam.fkeep, &full);
    while (cand) {
        const uint32_t s = (uint32_t)__builtin_ctz(cand);
        cand &= cand - 1u;
        uint64_t body[W];
        kh_lds_body<W, S>(mine, s, body);
        bool same = ((body[0] ^ (key.body[0] | ((uint64_t)d << cb))) & fam.bkeep[0]) == 0 && (body[0] & vmask) != 0;
#pragma unroll
        for (int i = 1; i < W; i++) same = same && ((body[i] ^ key.body[i]) & fam.bkeep[i]) == 0;
        *count += same ? 1 : 0;
    }
    return full ? -1 : 0;
}

// The kernels exist per key width W and slots per line S (bft_kh_geometry picks S from the key and value bits of an index): one-word keys
// (k <= 32) 10..4 slots, two-word keys 10..3, three-word keys 4..2, four-word keys 2..1.  CALL sees constexpr int KW, KS.
#define KH_DISPATCH(W_, S_, CALL)                                                                                                                   \
    switch ((W_) * 16 + (S_)) {                                                                                                                      \
    case 1 * 16 + 10: { constexpr int KW = 1, KS = 10; CALL; } break;                                                                                \
    case 1 * 16 + 9: { constexpr int KW = 1, KS = 9; CALL; } break;                                                                                  \
    case 1 * 16 + 8: { constexpr int KW = 1, KS = 8; CALL; } break;                                                                                  \
    case 1 * 16 + 7: { constexpr int KW = 1, KS = 7; CALL; } break;                                                                                  \
    case 1 * 16 + 6: { constexpr int KW = 1, KS = 6; CALL; } break;                                                                                  \
    case 1 * 16 + 5: { constexpr int KW = 1, KS = 5; CALL; } break;                                                                                  \
    case 1 * 16 + 4: { constexpr int KW = 1, KS = 4; CALL; } break;                                                                                  \
    case 2 * 16 + 10: { constexpr int KW = 2, KS = 10; CALL; } break;                                                                                \
    case 2 * 16 + 9: { constexpr int KW = 2, KS = 9; CALL; } break;                                                                                  \
    case 2 * 16 + 8: { constexpr int KW = 2, KS = 8; CALL; } break;                                                                                  \
    case 2 * 16 + 7: { constexpr int KW = 2, KS = 7; CALL; } break;                                                                                  \
    case 2 * 16 + 6: { constexpr int KW = 2, KS = 6; CALL; } break;                                                                                  \
    case 2 * 16 + 5: { constexpr int KW = 2, KS = 5; CALL; } break;                                                                                  \
    case 2 * 16 + 4: { constexpr int KW = 2, KS = 4; CALL; } break;                                                                                  \
    case 2 * 16 + 3: { constexpr int KW = 2, KS = 3; CALL; } break;                                                                                  \
    case 3 * 16 + 4: { constexpr int KW = 3, KS = 4; CALL; } break;                                                                                  \
    case 3 * 16 + 3: { constexpr int KW = 3, KS = 3; CALL; } break;                                                                                  \
    case 3 * 16 + 2: { constexpr int KW = 3, KS = 2; CALL; } break;                                                                                  \
    case 4 * 16 + 2: { constexpr int KW = 4, KS = 2; CALL; } break;                                                                                  \
    case 4 * 16 + 1: { constexpr int KW = 4, KS = 1; CALL; } break;                                                                                  \
    default: return bft_fail(BFT_GPU_E_LIMIT, "k-mer hash: no kernel for this key width / slots per line");                                            \
    }

