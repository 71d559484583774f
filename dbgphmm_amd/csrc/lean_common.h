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

// ---- vector-memory waits of the frontier kernels.
// gfx950 counts loads, stores and LDS-DMA on ONE in-order counter (vmcnt): waiting for a load also waits for every
// older store, and hipcc, which cannot know how many stores a loop iteration issued behind a load that is still in
// flight at the back edge, waits with vmcnt(0) -- a wave that walks a read alone then sits out one store round trip
// to HBM (~1 us) per read position.  The frontier kernels therefore issue the memory operations of their
// position loop from inline asm (one 16-byte store instruction per record, requests for the next records straight
// into LDS), count them in a wave-uniform integer, and wait with the exact vmcnt(N) for the one operation they
// need.  Operations the compiler issues on rare paths are not counted: N is then smaller than the true distance
// and the wait is merely conservative.
__device__ __forceinline__ void vm_wait_upto(int n) {
    // n: operations issued AFTER the one waited for (wave-uniform: told to the compiler, or the choice below
    // becomes nine exec-masked branches)
    n = __builtin_amdgcn_readfirstlane(n);
    if (n < 4) {
        if (n < 2) {
            if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else if (n < 6) {
        if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
// 16 bytes per lane to global memory, invisible to the compiler's counter model (see above); exec-masked by the caller
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// (a store of more than 8 bytes reads its data registers over two passes: a VALU write to them in the next slot can
// overtake the second -- hipcc pads that hazard for its own stores, not for inline asm.  Found the hard way: one
// low dword in 10^7 mapping entries changed from run to run.)
__device__ __forceinline__ void vm_store16(void *dst, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void vm_store8(void *dst, unsigned long long v) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
}
// every vector-memory operation of the wave has completed, and the compiler knows it (a real S_WAITCNT: vmcnt 0,
// expcnt / lgkmcnt untouched) -- behind a load on a rare path, so that the registers it wrote are not waited for
// again, with vmcnt(0), at every later use on EVERY path
__device__ __forceinline__ void vm_drain() { __builtin_amdgcn_s_waitcnt(0x0f70); }
// a value that a (compiler-visible) load has just delivered: consumed HERE, so that the load's wait is placed here
// and not at a later join, where it would be a vmcnt(0) on every path
__device__ __forceinline__ int vm_settle(int v) {
    int o;
    asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(v));
    return o;
}

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
