// LDS-DMA: global memory straight into LDS (gfx950 global_load_lds_dwordx4), shared by the dense forward kernel
// (row prefetch) and the one-lane-per-node frontier kernel (adjacency record fetched a position ahead).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace phmm {

// global_load_lds_dwordx4: 16 bytes per lane straight into LDS (wave-uniform base in M0 + lane*16), no
// VGPR held while the load is in flight -- the only way to keep several rows per wave in flight at 128
// VGPRs.  Issued from inline asm so that hipcc does not count it; the matching s_waitcnt vmcnt(N) is placed
// by hand (loads, stores and LDS-DMA retire in issue order on one counter).
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// the same with the address as a wave-uniform base (SGPR pair) plus a 32-bit per-lane byte offset: no 64-bit
// per-lane address to keep in (or spill from) vector registers
__device__ __forceinline__ void glds16_sv(const void *sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

}  // namespace phmm
