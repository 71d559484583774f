// Shared by the one-lane-per-node frontier kernels (lean_fwd_kernel.h, lean_bwd_kernel.h): constants, and the LDS
// hash node -> lane of the backward kernel (rebuilt once per read position; the forward kernel keeps lane links
// from position to position instead).
#pragma once

#include "sparse_dev.h"

namespace phmm {

static constexpr uint32_t LN_EMPTY = 0xffffffffu;
static constexpr int LN_HASH = 256;
static constexpr uint64_t LN_SLAB = 32768;  // record-pool bytes claimed per atomic

struct LeanShared {
    uint2 ent[LN_HASH];  // {node, lane}: one 8-byte LDS read answers a lookup
};

__device__ __forceinline__ void ln_sync() { wave_sync(); }

__device__ __forceinline__ uint32_t ln_hash(uint32_t id) { return (id * 2654435761u) >> 24; }

// lane of node `id` or -1
__device__ __forceinline__ int ln_find(const LeanShared &sh, uint32_t id) {
    uint32_t h = ln_hash(id);
    for (;;) {
        const uint2 e = sh.ent[h];
        if (e.x == id) return (int)e.y;
        if (e.x == LN_EMPTY) return -1;
        h = (h + 1) & (LN_HASH - 1);
    }
}

// Lanes of K nodes at once (valid[q] false: -1).  The first probes of all K keys are in flight together -- at a
// load of at most 64 keys in 256 cells nearly every lookup ends there; the rest walks on one by one.
template <int K> __device__ __forceinline__ void ln_find_many(const LeanShared &sh, const uint32_t (&key)[K], const bool (&valid)[K],
                                                              int (&out)[K]) {
    uint2 e[K];
#pragma unroll
    for (int q = 0; q < K; q++) e[q] = sh.ent[ln_hash(key[q])];
#pragma unroll
    for (int q = 0; q < K; q++) {
        int r = -1;
        if (valid[q]) {
            if (e[q].x == key[q]) r = (int)e[q].y;
            else if (e[q].x != LN_EMPTY) {
                uint32_t h = (ln_hash(key[q]) + 1) & (LN_HASH - 1);
                for (;;) {
                    const uint2 f = sh.ent[h];
                    if (f.x == key[q]) {
                        r = (int)f.y;
                        break;
                    }
                    if (f.x == LN_EMPTY) break;
                    h = (h + 1) & (LN_HASH - 1);
                }
            }
        }
        out[q] = r;
    }
}

__device__ __forceinline__ double ln_shfl(double v, int src) {
    // value of lane `src` (any lane when src < 0: the caller masks the result)
    return __shfl(v, src < 0 ? 0 : src);
}


// (re)build the map from the nodes currently on the lanes
__device__ __forceinline__ void ln_rebuild(LeanShared &sh, uint32_t id) {
    for (int h = threadIdx.x; h < LN_HASH; h += 64) sh.ent[h].x = LN_EMPTY;
    ln_sync();
    if (id != LN_EMPTY) {
        uint32_t h = ln_hash(id);
        for (;;) {
            const uint32_t old = atomicCAS(&sh.ent[h].x, LN_EMPTY, id);
            if (old == LN_EMPTY) break;
            h = (h + 1) & (LN_HASH - 1);
        }
        sh.ent[h].y = threadIdx.x;
    }
    ln_sync();
}

}  // namespace phmm
