// In-place bitonic sort of up to 512 (key, tie) pairs in LDS by a block of at least 256 threads: descending by key,
// equal keys ascending by tie -- a total order when the ties are distinct (slot numbers, node ids), so the result is
// the stable rank-by-counting sort of the one-wave kernels (sort_desc, frontier_dev.h; emit_mapping, mapping_flow.hip)
// at n log^2 n instead of n^2 LDS reads: on a 400-element column the counting sort's 160 000 broadcast reads were the
// LDS pipe's whole time (two blocks per CU share it).
//
// Thread t < NP/2 owns one compare-exchange per step.  The pairs of a step with distance j <= 64 lie inside the 128
// elements of the thread's own wave: those steps are ordered by the wave's own instruction order (wave_sync); only the
// steps with j = 128 / 256 (three of the 45 at NP = 512) and their neighbours take a block barrier.
#pragma once

#include "sparse_dev.h"

namespace phmm {

// NP: a power of two in [128, 512]; key[n..NP) / tie[n..NP) hold the caller's padding (smaller than every real key).
// Every thread of the block calls this (barriers inside); on return the arrays are visible to the whole block.
template <typename TieT> __device__ __forceinline__ void block_bitonic_desc(double *key, TieT *tie, int NP) {
    const int t = threadIdx.x;
    const bool active = t < (NP >> 1);
    bool wide_prev = true;  // (the caller's writes: a block barrier before the first step)
    for (int k = 2; k <= NP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool wide = j >= 128;
            if (wide || wide_prev) __syncthreads();
            else wave_sync();
            wide_prev = wide;
            if (active) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const double ki = key[i], kl = key[l];
                const TieT ti = tie[i], tl = tie[l];
                const bool l_first = (kl > ki) || (kl == ki && tl < ti);  // l belongs before i in descending order
                const bool desc = (i & k) == 0;
                if (desc ? l_first : !l_first) {
                    key[i] = kl;
                    key[l] = ki;
                    tie[i] = tl;
                    tie[l] = ti;
                }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int bitonic_size(int n) {
    int np = 128;
    while (np < n) np <<= 1;
    return np;
}

}  // namespace phmm
