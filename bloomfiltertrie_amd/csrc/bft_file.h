// bft_file.h -- reader / writer of the reference's .bft files; see bft_file.cpp.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "bft_image.h"

struct BftFileContent {  // what a .bft holds, as the GPU build wants it
    int k = 0, r1 = 0, r2 = 0;
    std::vector<std::string> genomes;
    std::vector<std::vector<uint8_t>> per_genome;  // packed k-mers (reference layout) of each genome id
    uint64_t n_kmers = 0;
};
bool bft_file_read(const char* path, BftFileContent& out, std::string& err);

struct BftHostImage {  // host copy of the device image, for serialisation
    int k = 0, r1 = 0, r2 = 0;
    std::vector<std::string> genomes;
    std::vector<BftNode> nodes;
    std::vector<BftCC> ccs;
    std::vector<uint64_t> f2w, clus, child, tk;
    std::vector<uint32_t> ucrow, tcol, cs_off, cs_ids;
};
bool bft_file_write(const char* path, const BftHostImage& im, std::string& err);

// annotation bytes of a sorted genome-id list: the smallest of the reference's modes 0/1/2 (src/annotation.c:634-650)
void bft_annot_encode(const uint32_t* ids, uint32_t n, std::vector<uint8_t>& out);
