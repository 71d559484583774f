// Shared by the one-lane-per-node frontier kernels (lean_fwd_kernel.h, lean_bwd_kernel.h): an LDS
// hash node -> lane for the <= 64 nodes resident on a wave, rebuilt once per read position.
#pragma once

#include "sparse_dev.h"

namespace phmm {

static constexpr uint32_t LN_EMPTY = 0xffffffffu;
static constexpr int LN_HASH = 256;
static constexpr uint64_t LN_SLAB = 32768;  // record-pool bytes claimed per atomic

struct LeanShared {
    uint32_t hkey[LN_HASH];
    uint8_t hval[LN_HASH];
    uint32_t winkey[64];
    uint16_t winh[64];
    unsigned long long mark;
};

__device__ __forceinline__ uint32_t ln_hash(uint32_t id) { return (id * 2654435761u) >> 24; }

// lane of node `id` or -1
__device__ __forceinline__ int ln_find(const LeanShared &sh, uint32_t id) {
    uint32_t h = ln_hash(id);
    for (;;) {
        const uint32_t k = sh.hkey[h];
        if (k == id) return (int)sh.hval[h];
        if (k == LN_EMPTY) return -1;
        h = (h + 1) & (LN_HASH - 1);
    }
}

__device__ __forceinline__ double ln_shfl(double v, int src) {
    // value of lane `src` (any lane when src < 0: the caller masks the result)
    return __shfl(v, src < 0 ? 0 : src);
}


// (re)build the map from the nodes currently on the lanes
__device__ __forceinline__ void ln_rebuild(LeanShared &sh, uint32_t id) {
    for (int h = threadIdx.x; h < LN_HASH; h += 64) sh.hkey[h] = LN_EMPTY;
    __syncthreads();
    if (id != LN_EMPTY) {
        uint32_t h = ln_hash(id);
        for (;;) {
            const uint32_t old = atomicCAS(&sh.hkey[h], LN_EMPTY, id);
            if (old == LN_EMPTY) break;
            h = (h + 1) & (LN_HASH - 1);
        }
        sh.hval[h] = (uint8_t)threadIdx.x;
    }
    __syncthreads();
}

}  // namespace phmm
