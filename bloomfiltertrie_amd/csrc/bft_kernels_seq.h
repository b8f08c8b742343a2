// bft_kernels_seq.h -- sequence queries around k_query: k_seq_encode / k_seq_pack / k_seq_count / k_seq_threshold
// Device code of libbft_gpu.so, included by bft_gpu.hip only (one translation unit: the kernels are templates launched from
// the host code there).
#pragma once
// ---- query_sequence (src/bft.c:1241-1351, harness src/file_io.c:1464-1574): every k-mer of every sequence ----
__device__ __forceinline__ int nt_code(char c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return -1;
    }
}

// Sequence queries, step 0.  The ASCII blob -> 2 bits per character (32 characters per u64, character c at bits 2(c%32) of
// word c/32: the packed layout of src/fasta.c:11-23 continued over the whole blob) + one "not ACGTU" bit per character.
// One thread per 32 characters; the blob is padded to a multiple of 32 bytes.
__global__ void k_seq_encode(const char* __restrict__ seqs, uint64_t n_words, uint64_t* __restrict__ codes, uint32_t* __restrict__ bad) {
    for (uint64_t wi = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; wi < n_words; wi += (uint64_t)gridDim.x * blockDim.x) {
        const uint4* src = (const uint4*)(seqs + wi * 32);
        const uint4 a = src[0], b = src[1];
        const uint32_t d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint64_t cw = 0;
        uint32_t bw = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int code = nt_code((char)((d[j] >> (8 * c)) & 0xFFu));
                const int i = 4 * j + c;
                cw |= (uint64_t)(code & 3) << (2 * i);
                bw |= (code < 0 ? 1u : 0u) << i;
            }
        }
        codes[wi] = cw;
        bad[wi] = bw;
    }
}

// reverse the 32 two-bit fields of a word
__device__ __forceinline__ uint64_t rev2_64(uint64_t x) {
    x = __brevll(x);
    return ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);
}

// Step 1.  One thread per k-mer position of the batch: its window is 2k bits of the code stream at a bit offset (W+1 word
// loads + funnel shifts, O(1) instead of a scan over k characters), valid unless one of its k "bad" bits is set (windows
// with a character outside ACGTU are skipped by the reference, src/bft.c:1298).  canonical: the reverse complement =
// complement, reverse the 2-bit fields of the 2k-bit string; strcmp(kmer, revcomp) >= 0 -> the reverse complement is
// searched (src/bft.c:1290-1296) = comparison of the lowest differing field.  Output: W zero-padded words per position (the
// record layout k_query reads with a record size of 8W bytes), valid[p], seq_of[p] = the sequence of position p.
template <int W>
__global__ void k_seq_pack(const uint64_t* __restrict__ codes, const uint32_t* __restrict__ bad, const uint64_t* __restrict__ seq_off,
                           const uint64_t* __restrict__ pos_off, uint32_t n_seqs, uint64_t P, int k, int canonical, uint64_t* __restrict__ words,
                           uint8_t* __restrict__ valid, uint32_t* __restrict__ seq_of) {
    for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < P; p += (uint64_t)gridDim.x * blockDim.x) {
        // the sequence of position p: last s with pos_off[s] <= p.  The 64 positions of a wavefront are consecutive, so the
        // binary search runs once per wavefront on its first position (uniform values: scalar loads) and every lane
        // walks forward from there (sequences shorter than k own no position and are stepped over).
        const uint64_t p0 = p - (threadIdx.x & 63u);
        const uint64_t p0u = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(p0 >> 32)) << 32) | (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)p0);
        uint32_t lo = 0, hi = n_seqs;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pos_off[mid] <= p0u) lo = mid; else hi = mid;
        }
        while (lo + 1 < n_seqs && pos_off[lo + 1] <= p) lo++;
        const uint64_t c0 = seq_off[lo] + (p - pos_off[lo]);  // first character of the window, as an index into the blob
        // 2k bits of the code stream from bit 2*c0
        const uint64_t w0 = c0 >> 5;
        const uint32_t sh = (uint32_t)(c0 & 31u) * 2u;
        uint64_t cw[W + 1], xf[W], xr[W];
#pragma unroll
        for (int q = 0; q <= W; q++) cw[q] = codes[w0 + q];  // the code array has W+1 words of slack
#pragma unroll
        for (int q = 0; q < W; q++) xf[q] = sh ? (cw[q] >> sh) | (cw[q + 1] << (64u - sh)) : cw[q];
        const int top = 2 * k - 64 * (W - 1);  // bits used in the last word
        if (top < 64) xf[W - 1] &= (1ull << top) - 1ull;
        // any bad character in [c0, c0 + k)?
        bool ok = true;
        {
            const uint64_t b0 = c0 >> 5;
            const uint32_t bs = (uint32_t)(c0 & 31u);
            int left = k;
            uint32_t first = bad[b0] >> bs;
            if (left < 32 - (int)bs) first &= (1u << left) - 1u;
            ok = first == 0;
            left -= 32 - (int)bs;
            for (uint64_t j = b0 + 1; left > 0; j++, left -= 32) {
                uint32_t m = bad[j];
                if (left < 32) m &= (1u << left) - 1u;
                ok = ok && m == 0;
            }
        }
        bool use_rc = false;
        if (canonical) {
            // complement, then reverse the fields of the 64W-bit string and shift the 2k bits of interest back down
            uint64_t rv[W + 1];
#pragma unroll
            for (int q = 0; q < W; q++) rv[q] = rev2_64(~xf[W - 1 - q]);
            rv[W] = 0;
            const uint32_t dn = (uint32_t)(64 * W - 2 * k);  // < 64
#pragma unroll
            for (int q = 0; q < W; q++) xr[q] = dn ? (rv[q] >> dn) | (rv[q + 1] << (64u - dn)) : rv[q];
            if (top < 64) xr[W - 1] &= (1ull << top) - 1ull;
            use_rc = true;  // equal strings: the (identical) reverse complement
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
        for (int q = 0; q < W; q++) words[p * W + q] = ok ? (use_rc ? xr[q] : xf[q]) : 0ull;
        valid[p] = ok ? 1 : 0;
        seq_of[p] = lo;
    }
}

// Step 3 (step 2 is k_query on the word records): per-(sequence, genome) counters.  Consecutive k-mers of a read mostly
// carry the same colour set, so counting works on runs: the 64 lanes of a wavefront hold 64 consecutive positions, run
// boundaries come from a shuffle + __ballot, and the first lane of every run of equal (sequence, colour set) adds the run
// length (up to the end of the wavefront) once per genome of the set -- instead of one atomic per k-mer and genome.
__global__ void k_seq_count(const uint32_t* __restrict__ rows, const uint8_t* __restrict__ valid, const uint32_t* __restrict__ seq_of,
                            const uint32_t* __restrict__ tcol, const uint32_t* __restrict__ cs_off, const uint32_t* __restrict__ cs_ids, uint64_t P, uint32_t G,
                            uint32_t* __restrict__ counts) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nblk = (P + blockDim.x - 1) / blockDim.x;
    for (uint64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {  // whole wavefronts stay in the loop together
        const uint64_t p = blk * blockDim.x + threadIdx.x;
        uint32_t cs = 0xFFFFFFFFu, sq = 0xFFFFFFFFu;
        if (p < P) {
            sq = seq_of[p];
            const uint32_t r = rows[p];
            if (r != BFT_ABSENT_ROW && valid[p]) cs = tcol[r];
        }
        const uint32_t pcs = __shfl_up(cs, 1), psq = __shfl_up(sq, 1);
        const bool boundary = lane == 0 || pcs != cs || psq != sq;
        const uint64_t bmask = __ballot(boundary);
        if (boundary && cs != 0xFFFFFFFFu) {
            const uint64_t above = lane == 63 ? 0ull : bmask >> (lane + 1);
            const uint32_t len = above ? (uint32_t)__builtin_ctzll(above) + 1u : 64u - lane;
            uint32_t* c = counts + (size_t)sq * G;
            for (uint32_t q = cs_off[cs]; q < cs_off[cs + 1]; q++) atomicAdd(&c[cs_ids[q]], len);
        }
    }
}

__global__ void k_seq_threshold(const uint32_t* __restrict__ counts, const uint64_t* __restrict__ minv, uint32_t n_seqs, uint32_t G, uint32_t rowbytes,
                                uint8_t* __restrict__ out) {
    const uint64_t total = (uint64_t)n_seqs * rowbytes;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = (uint32_t)(i / rowbytes), b = (uint32_t)(i % rowbytes);
        uint32_t v = 0;
        for (uint32_t j = 0; j < 8 && b * 8 + j < G; j++) {
            const uint32_t c = counts[(size_t)s * G + b * 8 + j];
            if (c && c >= minv[s]) v |= 1u << j;
        }
        out[i] = (uint8_t)v;
    }
}
