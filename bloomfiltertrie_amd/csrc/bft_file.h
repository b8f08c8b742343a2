// bft_file.h -- reader / writer of the reference's .bft files; see bft_file.cpp.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <sys/mman.h>

#include <memory>
#include <string>
#include <utility>
#include <new>
#include <thread>
#include <vector>

#include "bft_image.h"

struct BftFileContent {  // what a .bft holds, as the GPU build wants it
    int k = 0, r1 = 0, r2 = 0;
    std::vector<std::string> genomes;
    std::vector<std::vector<uint8_t>> per_genome;  // packed k-mers (reference layout) of each genome id
    uint64_t n_kmers = 0;
};
bool bft_file_read(const char* path, BftFileContent& out, std::string& err);

// Gigabytes of host vectors are given back by a thread of their own: unmapping them costs the caller 0.1 s per gigabyte (config 3: a third of the
// load, a fifth of the write) and nothing depends on it.  `obj` is left empty (moved from).  The threads are counted: bft_dispose_drain() -- called
// when the library is unloaded or the process exits (a destructor in bft_file.cpp) and available to a caller that wants the memory back now -- waits
// for them, so that none is still running in code that is being unmapped or beside the static destructors.
void bft_dispose_begin(void);
void bft_dispose_end(void);
void bft_dispose_drain(void);
template <class T>
void bft_dispose_async(T& obj) {
    T* p = new (std::nothrow) T(std::move(obj));
    if (!p) return;  // (obj keeps its content and is destroyed by its owner)
    bft_dispose_begin();
    try {
        std::thread([p] { delete p; bft_dispose_end(); }).detach();
    } catch (...) {
        delete p;
        bft_dispose_end();
    }
}

// a vector whose resize() leaves trivially constructible elements uninitialised: the big arrays of the host image are filled by a copy from
// the device right away, and zeroing a gigabyte first is a page fault per 4 KB for nothing
// ... and blocks of 8 MB and more are 2 MB-aligned and advised to be backed by huge pages (where the kernel grants them on madvise): the copy from
// the device then faults 512 times fewer pages in, and giving the block back unmaps as many fewer.
template <class T>
struct BftDefaultInit : std::allocator<T> {
    template <class U> struct rebind { using other = BftDefaultInit<U>; };
    using std::allocator<T>::allocator;
    static constexpr size_t HUGE_FROM = (size_t)8 << 20, HUGE_ALIGN = (size_t)2 << 20;
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes >= HUGE_FROM) {
            void* p = nullptr;
            const size_t padded = (bytes + HUGE_ALIGN - 1) / HUGE_ALIGN * HUGE_ALIGN;
            if (posix_memalign(&p, HUGE_ALIGN, padded) != 0 || !p) throw std::bad_alloc();
            (void)madvise(p, padded, MADV_HUGEPAGE);
            return static_cast<T*>(p);
        }
        void* p = malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t) noexcept { free(p); }
    template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <class T>
using BftBigVec = std::vector<T, BftDefaultInit<T>>;

struct BftHostImage {  // host copy of the device image, for serialisation
    int k = 0, r1 = 0, r2 = 0;
    std::vector<std::string> genomes;
    std::vector<BftNode> nodes;
    std::vector<BftCC> ccs;
    std::vector<uint64_t> f2w, clus, child;
    std::vector<uint32_t> ucrow, cs_off;
    BftBigVec<uint64_t> tk;            // sorted T-form k-mers
    BftBigVec<uint32_t> tcol, cs_ids;  // colour set per k-mer; the dictionary's genome ids
};
bool bft_file_write(const char* path, const BftHostImage& im, std::string& err);

// annotation bytes of a sorted genome-id list: the smallest of the reference's modes 0/1/2 (src/annotation.c:634-650)
void bft_annot_encode(const uint32_t* ids, uint32_t n, std::vector<uint8_t>& out);
