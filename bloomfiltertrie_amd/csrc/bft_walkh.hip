// bft_walkh.hip -- k_query6h, the kernel of "walk_hash": the container walk whose plain root groups are looked up in their hashed form (the
// k-mer hash, bft_image.h BFT_KH_*) and whose special prefixes -- child Nodes, UC rows at the root -- walk the containers (bft_walk.h).  Its own
// translation unit: one instance per key width and slots per line.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/bft_gpu.h"
#include "bft_dev.h"
#include "bft_image.h"
#include "bft_kh.h"
#include "bft_walk.h"

#define BFT_BLOCK 256
#define BFT_ABSENT_ROW 0xFFFFFFFFu
#include "bft_kernels_query.h"

template <int W, int S>
__global__ __launch_bounds__(BFT_BLOCK6) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_query6h(BftImage im, const uint8_t* __restrict__ packed, uint64_t n, int B,
                                                                                                  uint64_t* __restrict__ bits64, uint32_t* __restrict__ rows,
                                                                                                  BftClaimCtr ctr) {
    query_body<W, BFT_BLOCK6, false, 0, true, S>(im, packed, n, B, bits64, rows, ctr);
}

// n k-mers of `rec` bytes each; d_ctr: the stream's claim counters (NULL: chunks by wavefront number)
int bft_walkh_query(const BftImage& im, const uint8_t* d_kmers, uint64_t n, int rec, uint64_t* d_bits64, uint32_t* d_rows, BftClaimCtr d_ctr, uint32_t grid_mult, hipStream_t s) {
    if (!im.walk_kh || !im.rspec || !im.rdir) return bft_fail(BFT_GPU_E_STATE, "walk_hash: no hashed root groups in this image");
    const uint64_t n_chunks = (n + 64ull * BFT_WALK_PASSES - 1) / (64ull * BFT_WALK_PASSES);
    const uint64_t wgc = (n_chunks + BFT_BLOCK6 / 64 - 1) / (BFT_BLOCK6 / 64);
    const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(wgc, 512ull * std::max(1u, grid_mult))));
    // dynamic LDS: the wavefronts' queues of parked k-mers (where the other walk kernels keep the root's Bloom block) and room for CC headers
    const size_t lds = ((size_t)BFT_MODULO_HASH * 8 + 15) / 16 * 16 + BFT_LDS_ROOT_MAX_CC * sizeof(BftCCX);
    KH_DISPATCH(im.W, (int)im.kh.S, hipLaunchKernelGGL((k_query6h<KW, KS>), grid, dim3(BFT_BLOCK6), lds, s, im, d_kmers, n, rec, d_bits64, d_rows, d_ctr));
    HIPCK(hipGetLastError());
    return 0;
}
