// bft_claims.h -- how the persistent query kernels (k_query_kh, k_branching_kh, k_seq_kh, k_query*) deal the blocks of a batch out to their
// resident workgroups (device code).
//
// Workgroups are bound to an XCD by their number, so a split of the batch by workgroup number gives every XCD one eighth of it however
// fast it gets through it -- and how fast an XCD reaches a randomly read table depends on where the table lies: 2.70 / 2.77 / 3.03 ms for
// the same table, batch and kernel (same misses, same latency per request, fewer requests in flight: XCDs idle at the end), 2.63 ms
// wherever it lies once the blocks are claimed from a counter.  Round 4: the FIRST round is static -- workgroup b
// starts on the blocks [b * chunk, (b + 1) * chunk) without asking anyone -- and only the rounds after it are claimed (counter value c
// stands for block G * chunk + c).  2048 workgroups asking one counter for their first blocks at the same instant cost a launch ~50 us
// (a claim is served every ~12 ns), which is why batches under 2^25 k-mers kept the static split in round 3; now the first claim is made
// while the first round is being answered and every batch size can take part.
// A claim takes remaining / (2 x workgroups) blocks, at most `chunk`, at least MINC: for the one-line lookups of the k-mer hash (a block of
// 256 k-mers is answered in ~11 us by 2048 workgroups) fewer than four blocks per claim saturate the counter (a block per claim: 5.9 ms per
// launch, two: 3.4, against 2.6), and more than four widen the window of the query stream the resident workgroups read at a time (2.61 /
// 2.62 / 2.65 / 2.70 ms at 4 / 16 / 32 / 64).
// ctr.p == NULL: static rounds only (round r of workgroup b = blocks [(b + r G) chunk, +chunk)), same loop, same whole-line stores.
// The counter is ONE 64-bit word per (handle, stream) that only ever grows, and every launch is given its own range of it (round 5; rounds 3-4
// had the last workgroup of a launch zero a 32-bit pair for the next one -- a launch that never finished left it non-zero and every later launch
// on the stream silently skipped blocks): the host hands the kernel `base`, the value the counter stands at when the launch's first claim is
// made, and moves its own copy on by a bound on what the launch can claim (blocks + one claim per workgroup beyond them).  A workgroup raises
// the counter to `base` (atomicMax) before its first claim, so whatever an earlier launch left behind -- it can only be BELOW this launch's base
// -- is irrelevant; a claim's value minus `base` = blocks claimed so far beyond the first round.  Nothing is reset, nothing counts the workgroups
// that are done.  One launch at a time per counter (the host keeps one per stream and handle: launches of one handle on one stream from two host
// threads at once, or a captured graph replayed beside a live launch, would share a range -- not supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct BftClaimCtr {
    unsigned long long* p;    // the stream's counter, or NULL: static rounds
    unsigned long long base;  // where this launch's range of it starts
};

template <uint32_t MINC>
struct BftClaims {
    unsigned long long* ctr;
    unsigned long long base;
    uint32_t chunk;
    uint64_t nblk;
    uint32_t* s_next;  // two words of LDS: first block and size of the claim in flight
    uint64_t blk, blk_end, start;  // the block being answered; end and first block of the round it belongs to

    // (a batch the first round covers needs no counter at all: it runs exactly like the static split)
    __device__ __forceinline__ BftClaims(BftClaimCtr c, uint32_t ch, uint64_t nb, uint32_t* lds2)
        : ctr((uint64_t)gridDim.x * ch >= nb ? nullptr : c.p), base(c.base), chunk(ch), nblk(nb), s_next(lds2), blk(0), blk_end(0), start(0) {}

    __device__ __forceinline__ void claim(uint64_t from) {
        const uint64_t rem = nblk - min(nblk, from);
        const uint32_t want = (uint32_t)max((uint64_t)min(chunk, MINC), min((uint64_t)chunk, rem / (2ull * gridDim.x)));
        s_next[1] = want;
        s_next[0] = (uint32_t)(atomicAdd(ctr, (unsigned long long)want) - base);
    }
    // the first round: static (and the counter enters this launch's range before the thread's first claim: same thread, same address --
    // the two atomics are ordered)
    __device__ __forceinline__ void first() {
        if (ctr && threadIdx.x == 0) atomicMax(ctr, base);
        blk = start = (uint64_t)blockIdx.x * chunk;
        blk_end = min(nblk, blk + chunk);
    }
    // the next round (every thread of the workgroup calls it: the barriers also stand between a caller's reads of its per-round LDS
    // and the next round's writes)
    __device__ __forceinline__ void take() {
        __syncthreads();
        if (ctr) {
            blk = start = (uint64_t)gridDim.x * chunk + s_next[0];
            blk_end = min(nblk, blk + s_next[1]);
        } else {
            blk = start = start + (uint64_t)gridDim.x * chunk;
            blk_end = min(nblk, blk + chunk);
        }
        __syncthreads();
    }
    __device__ __forceinline__ bool last_of_round() const { return blk + 1 >= blk_end; }
    // The claim for the next round is sent off when the FIRST block of a round has been answered and travels while the others are
    // (sent at the start of the round, the thread's wavefront would wait for the counter's answer before it could use any of its own
    // loads -- vector memory results return in order --, and at the start of a launch every workgroup asks at the same instant).  A
    // round of one block sends it right before it is needed: use rounds of two or more with a counter.
    __device__ __forceinline__ void advance() {
        if (ctr && threadIdx.x == 0 && blk == start) claim(blk_end);
        if (++blk >= blk_end) take();
    }
    // the same for kernels that answer `nb` blocks of a round at a time
    __device__ __forceinline__ void advance_by(uint32_t nb) {
        if (ctr && threadIdx.x == 0 && blk == start) claim(blk_end);
        blk += nb;
        if (blk >= blk_end) take();
    }
    // after the loop: nothing to do (the counter is never reset, see above)
    __device__ __forceinline__ void done() {}
};
