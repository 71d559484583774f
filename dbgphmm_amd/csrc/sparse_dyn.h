// Shared between sparse_dyn.hip (adaptive sparse forward) and mapping_flow.hip
// (backward_by_forward + mapping extraction).
#pragma once

#include <mutex>

#include "frontier_dev.h"

namespace phmm {

static constexpr uint32_t SP_STOP_SLICE = 64u;  // not an error: the launch's step budget ended (SparseBwdArgs::max_steps)
static constexpr uint32_t SP_ERR_POOL = 16u;  // record pool exhausted: the host grows it and reruns

// Bump-allocated records in HBM, one per (read, position).
//   forward table record : [n u32][na u32][E i32][pad] ids[n|1] m[na] i[na] d[n]
//   mapping record       : [n u32][pad u32]            ids[n|1] logp[n]
struct RecPool {
    uint8_t *base;
    unsigned long long *top;  // bytes used
    uint64_t cap;
    uint64_t *off;  // [positions] byte offset of the record + 8 (0 = none); the bias keeps every derived address 8-byte aligned: hipcc folds a "+1 -1" bias into a misaligned scalar-load base, whose low bits the hardware drops
};

__device__ __forceinline__ uint64_t pool_alloc(const RecPool &p, uint64_t bytes) {
    // wave-uniform: lane 0 allocates, everyone gets the offset
    unsigned long long o = 0;
    if (threadIdx.x == 0) o = atomicAdd(p.top, (unsigned long long)bytes);
    o = __shfl(o, 0);
    return o;
}

template <int CAP> __device__ bool store_record(const RecPool &p, uint64_t pos_index, const FVec<CAP> &c) {
    const int n = c.n, na = c.na;
    const uint64_t idb = (uint64_t)((n + 1) & ~1) * 4;
    const uint64_t bytes = (16 + idb + (uint64_t)(2 * na + n) * 8 + 15) & ~15ull;  // records start 16-byte aligned (LDS-DMA reads)
    const uint64_t o = pool_alloc(p, bytes);
    if (o + bytes > p.cap) return false;
    uint8_t *rec = p.base + o;
    if (threadIdx.x == 0) {
        ((uint32_t *)rec)[0] = (uint32_t)n;
        ((uint32_t *)rec)[1] = (uint32_t)na;
        ((int *)rec)[2] = c.E;
        ((uint32_t *)rec)[3] = 0;
        p.off[pos_index] = o + 8;
    }
    uint32_t *ids = (uint32_t *)(rec + 16);
    double *m = (double *)(rec + 16 + idb), *i = m + na, *d = i + na;
    for (int j = threadIdx.x; j < n; j += 64) {
        ids[j] = c.id[j];
        d[j] = c.d[j];
        if (j < na) {
            m[j] = c.m[j];
            i[j] = c.i[j];
        }
    }
    return true;
}

// a Col (list column of the hinted flow: every entry carries m, i and d) as a forward record
template <int CAP> __device__ bool store_record_col(const RecPool &p, uint64_t pos_index, const Col<CAP> &c) {
    const int n = c.n;
    const uint64_t idb = (uint64_t)((n + 1) & ~1) * 4;
    const uint64_t bytes = (16 + idb + (uint64_t)(3 * n) * 8 + 15) & ~15ull;
    const uint64_t o = pool_alloc(p, bytes);
    if (o + bytes > p.cap) return false;
    uint8_t *rec = p.base + o;
    if (threadIdx.x == 0) {
        ((uint32_t *)rec)[0] = (uint32_t)n;
        ((uint32_t *)rec)[1] = (uint32_t)n;
        ((int *)rec)[2] = c.E;
        ((uint32_t *)rec)[3] = 0;
        p.off[pos_index] = o + 8;
    }
    uint32_t *ids = (uint32_t *)(rec + 16);
    double *m = (double *)(rec + 16 + idb), *i = m + n, *d = i + n;
    for (int j = threadIdx.x; j < n; j += 64) {
        ids[j] = c.id[j];
        m[j] = c.m[j];
        i[j] = c.i[j];
        d[j] = c.d[j];
    }
    return true;
}

inline SparseModel sparse_model_of(const phmm_model *m) {
    const ModelDev &d = m->dev;
    SparseModel s{};
    s.N = (int)m->N;
    s.emis = d.emis.as<uint8_t>();
    s.init = d.init.as<double>();
    s.par_off = d.par_off.as<uint32_t>();
    s.par_node = d.par_node.as<uint32_t>();
    s.par_edge = d.par_edge.as<uint32_t>();
    s.chi_off = d.chi_off.as<uint32_t>();
    s.chi_node = d.chi_node.as<uint32_t>();
    s.chi_edge = d.chi_edge.as<uint32_t>();
    s.trans = d.trans_lin.as<double>();
    s.fadj = d.fadj.as<FwdAdj>();
    s.badj = d.badj.as<BwdAdj>();
    s.prec = d.prec.as<ParRec>();
    s.par_w = d.par_w.as<double>();
    s.chi_w = d.chi_w.as<double>();
    s.lp = m->lin;
    s.logib = d.logib.as<double>();
    s.packed = d.max_degree <= (uint32_t)ADJ_DEG ? 1 : 0;
    return s;
}

}  // namespace phmm

#include "dense_internal.h"

namespace phmm {

// Capacity classes as in the forward kernel (sparse_fwd_kernel.h): the tail of a read is walked
// by a <64>-slot kernel (many reads per CU), the few positions next to the dense/sparse switch
// (forward records of up to 400 entries) by the <400>-slot kernel, which also hands the column
// over to the dense backward kernel.  Between two phases the last B column travels through a
// per-read hand-off slot in HBM.
static constexpr int HANDOFF_CAP = 128;
struct BHandoff {
    int n, E;
    uint32_t id[HANDOFF_CAP];
    double m[HANDOFF_CAP], i[HANDOFF_CAP], d[HANDOFF_CAP];
};

struct SparseBwdArgs {
    SparseModel M;
    DenseArgs d;
    int W, Lb;
    const int *sw;
    const uint8_t *bases;
    RecPool fpool, mpool;
    const uint64_t *lane_pos0;  // forward-record position base of each lane (chunk local)
    const uint64_t *map_pos0;   // global read position base of each lane (mapping records)
    const uint32_t *lanes;
    double ratio_lin;
    uint32_t *err;
    // list mode (backward_with_mapping, backward.rs:59-93): B.tables[i] over mapping.nodes(i); no dense head
    const uint64_t *list_off;   // [total_pos+1] (global positions) or null
    const uint32_t *list_nodes;
    int topk;        // > 0: to_mapping(topk) instead of to_mapping_by_score_ratio
    int mode;        // 0: start at the last position from b_init; 1: resume from the hand-off slot
    int max_steps;   // > 0: positions to walk at most in this launch (the column is parked in the hand-off slot: <64>
                     // kernels anywhere -- their err carries SP_STOP_SLICE --, the 400-slot kernel where it fits 64 nodes)
    int *stop;       // [lanes] in (mode 1): position to compute next; out: see below
    BHandoff *hand;  // [lanes]
};
// stop[gi] on exit: s0      -> finished (column s0+1 handed to the dense kernel)
//                   len     -> nothing done (the record of the last position does not fit the class)
//                   other p -> positions > p are done, B.tables[p+1] is in the hand-off slot



// Device-side collector of the mapping lists of all reads: one record pool for the whole call,
// record offsets indexed by the GLOBAL read position (reads->off[read] + i), so the final CSR is
// produced on the device (counts -> scan -> compaction) without a host pass over positions.
struct SinkOverflow {};  // thrown by a chunk when the shared mapping pool is full: the call restarts with a bigger one
struct MappingSink {
    RecPool mp;
    uint64_t cap;
    uint64_t total_pos;
    const phmm_reads *reads;
};

// Everything the backward/mapping pass needs to know about one chunk of read groups after
// its (dense warm-up + sparse) forward pass.
struct MapChunk {
    phmm_model *m;
    int W, Lc, Lfull, ngc, lanes;
    DenseArgs a;
    SparseModel fa_M;
    const int *d_sw;
    const uint8_t *d_bases_full;
    RecPool fpool;
    const uint64_t *d_lane_pos0;
    const std::vector<int> *hl, *hsw;
    const std::vector<uint64_t> *lane_pos0;
    double ratio_lin;
    double *d_logp_sparse;
    uint32_t *cand_node;  // [lanes][400] scratch
    double *cand_tot;
    int topk;  // > 0: to_mapping(topk) instead of to_mapping_by_score_ratio
    std::mutex *dense_token;  // held while the chunk runs its HBM-bound dense backward (sparse_dyn.hip)
    bool main_plan;  // per-launch statistics (bench.py's roofline) cover the main plan's full-width launches only
};

// top_nodes(K) / the 400 best of a dense column (sparse_dyn.hip: select_top400).  The column is read through
// d.Fm/Fi/Fd at column sw[lane]-1 of a [ng][d.Lc][N][W] table; d.tmaxF bounds its totals from above.
struct Top400Args {
    DenseArgs d;
    int W;
    const int *sw;
    const uint32_t *need;  // lanes (g*W + r) that need the selection
    uint32_t *sc_node;     // [n_need][N]
    double *sc_tot;        // [n_need][N]
    int *sc_n;             // [n_need]
    uint32_t *cand_node;
    double *cand_tot;
    int *cand_n;
    double ratio_lin;
    int K;  // entries kept: 400 (ArrayVec capacity) or n_active_nodes (top_nodes, table.rs:127-131)
};
void launch_select_top(const Top400Args &ta, unsigned n, hipStream_t s);

void mapping_backward_chunk(MapChunk &mc, const std::vector<uint32_t> &sparse_lanes, MappingSink *sink,
                            const Plan &plan, int g0, uint64_t R);

}  // namespace phmm
