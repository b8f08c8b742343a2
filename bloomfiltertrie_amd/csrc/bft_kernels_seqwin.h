// bft_kernels_seqwin.h -- the k-mer window of a sequence position (device code shared by the translation units of libbft_gpu.so)
#pragma once
// reverse the 32 two-bit fields of a word
__device__ __forceinline__ uint64_t rev2_64(uint64_t x) {
    x = __brevll(x);
    return ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);
}

// The window of one k-mer position: 2k bits of the code stream at a bit offset (W+1 word loads + funnel shifts, O(1) instead of a
// scan over k characters) -> packed words x[W]; false when one of its k "bad" bits is set (windows with a character outside ACGTU
// are skipped by the reference, src/bft.c:1298).  canonical: the reverse complement = complement, reverse the 2-bit fields of the
// 2k-bit string; strcmp(kmer, revcomp) >= 0 -> the reverse complement is searched (src/bft.c:1290-1296) = comparison of the
// lowest differing field.  c0 = index of the window's first character in the blob.
template <int W>
__device__ __forceinline__ bool seq_window(const uint64_t* __restrict__ codes, const uint32_t* __restrict__ bad, uint64_t c0, int k, int canonical, uint64_t* x) {
    const uint64_t w0 = c0 >> 5;
    const uint32_t sh = (uint32_t)(c0 & 31u) * 2u;
    uint64_t cw[W + 1], xf[W];
#pragma unroll
    for (int q = 0; q <= W; q++) cw[q] = codes[w0 + q];  // the code array has W+1 words of slack
#pragma unroll
    for (int q = 0; q < W; q++) xf[q] = sh ? (cw[q] >> sh) | (cw[q + 1] << (64u - sh)) : cw[q];
    const int top = 2 * k - 64 * (W - 1);  // bits used in the last word
    if (top < 64) xf[W - 1] &= (1ull << top) - 1ull;
    bool ok = true;
    {
        const uint32_t bs = (uint32_t)(c0 & 31u);
        int left = k;
        uint32_t first = bad[w0] >> bs;
        if (left < 32 - (int)bs) first &= (1u << left) - 1u;
        ok = first == 0;
        left -= 32 - (int)bs;
        for (uint64_t j = w0 + 1; left > 0; j++, left -= 32) {
            uint32_t m = bad[j];
            if (left < 32) m &= (1u << left) - 1u;
            ok = ok && m == 0;
        }
    }
    bool use_rc = false;
    uint64_t xr[W];
    if (canonical) {
        uint64_t rv[W + 1];
#pragma unroll
        for (int q = 0; q < W; q++) rv[q] = rev2_64(~xf[W - 1 - q]);
        rv[W] = 0;
        const uint32_t dn = (uint32_t)(64 * W - 2 * k);  // < 64
#pragma unroll
        for (int q = 0; q < W; q++) xr[q] = dn ? (rv[q] >> dn) | (rv[q + 1] << (64u - dn)) : rv[q];
        if (top < 64) xr[W - 1] &= (1ull << top) - 1ull;
        use_rc = true;
#pragma unroll
        for (int q = W - 1; q >= 0; q--) {  // the lowest differing field decides: word 0 last
            const uint64_t df = xf[q] ^ xr[q];
            if (df) {
                const int fs = __builtin_ctzll(df) & ~1;
                use_rc = ((xf[q] >> fs) & 3ull) > ((xr[q] >> fs) & 3ull);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < W; q++) x[q] = (canonical && use_rc) ? xr[q] : xf[q];
    return ok;
}

