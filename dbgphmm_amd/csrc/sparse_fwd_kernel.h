// The per-read adaptive sparse forward kernel (one wave64 per read), in capacity classes.
//
// The frontier right after the dense->sparse switch holds up to 400 nodes, a few positions later
// a few dozen.  LDS per wave decides how many reads a CU can interleave (the kernel is bound by
// dependent L2 accesses, not by arithmetic), so a read is walked in phases:
//   phase A  <400>  positions [s0, s0 + PA)            from the dense warm-up column
//   phase B  <64>   positions [s0 + PA, len)           resumed from the stored record
//   phase C  <400>  from the position where phase B ran out of slots (rare), to the end
// Every column is stored as a record (sparse_dyn.h); a phase resumes from the record of the
// position before its first one.  A <64> phase that would drop an insert (which the 400-slot
// vector of the reference would keep) stops BEFORE storing that column and reports the position.
#pragma once

#include "dense_internal.h"
#include "sparse_dyn.h"

namespace phmm {

struct SparseFwdArgs {
    SparseModel M;
    DenseArgs d;  // dense warm-up tables of the chunk
    int W;
    const int *sw;
    const uint32_t *cand_node;
    const double *cand_tot;
    const int *cand_n;
    const uint32_t *lanes;  // [n] flattened (g*W + r) of the reads handled by this launch
    const uint8_t *bases;   // chunk-transposed full-length bases [ng][Lb][W]
    int Lb;
    double ratio_lin;
    double *out_logp;  // [ng*W]
    uint32_t *err;     // [ng*W]
    RecPool pool;      // one record per sparse position
    const uint64_t *lane_pos0;  // [ng*W] first position index of each lane
    int mode;          // 0: start at the switch column from the dense tables; 1: resume
    int max_steps;     // mode 0: positions to walk at most (0 = to the end of the read)
    int topk;          // > 0: top_nodes(topk) instead of top_nodes_by_score_ratio (use_max_ratio = false)
    int *stop;         // [ng*W] in (mode 1): first position to compute; out: first position NOT done
};

template <int CAP> __device__ bool load_record_fvec(const RecPool &p, uint64_t pos_index, FVec<CAP> &f) {
    const uint64_t o1 = p.off[pos_index];
    if (o1 == 0) return false;
    const uint8_t *rec = p.base + (o1 - 8);
    const int *hw = (const int *)rec;
    const int n = hw[0], na = hw[1], E = hw[2];
    if (n > CAP) return false;
    fv_clear(f);
    wave_sync();
    const uint64_t idb = (uint64_t)((n + 1) & ~1) * 4;
    const uint32_t *ids = (const uint32_t *)(rec + 16);
    const double *m = (const double *)(rec + 16 + idb), *i = m + na, *d = i + na;
    for (int j = threadIdx.x; j < n; j += 64) {
        const uint32_t id = ids[j];
        f.id[j] = id;
        f.d[j] = d[j];
        f.m[j] = j < na ? m[j] : 0.0;
        f.i[j] = j < na ? i[j] : 0.0;
        const uint32_t cell = fv_cell(f, id);
        f.hslot[cell] = (uint16_t)j;
    }
    if (threadIdx.x == 0) {
        f.n = n;
        f.na = na;
        f.E = E;
    }
    wave_sync();
    return true;
}

template <int CAP>
__global__ void __launch_bounds__(64) sparse_forward_kernel(const SparseFwdArgs a) {
    __shared__ FVec<CAP> cols[2];
    __shared__ FScratch<CAP> sc;
    const int lane = threadIdx.x;
    const uint32_t gi = a.lanes[blockIdx.x];
    const int g = (int)(gi / a.W), r = (int)(gi % a.W);
    const int len = a.d.len[gi];
    const int s0 = a.sw[gi];
    const size_t NW = (size_t)a.d.N * a.W;
    const uint64_t p0 = a.lane_pos0[gi];
    constexpr bool SMALL = CAP < PHMM_MAX_ACTIVE_NODES;
    if (lane == 0) sc.dropped = 0;
    wave_sync();
    uint32_t err = 0;
    int pos;        // next position to compute
    int end = len;  // first position this launch does not compute
    int done_to;    // first position NOT done when the kernel leaves
    if (a.mode == 0) {
        // ---- first sparse column: top list = candidates of dense column s0-1 sorted by
        // (total desc, node asc) = the reference's stable sort over the dense nodevec
        pos = s0;
        if (a.max_steps > 0 && s0 + a.max_steps < len) end = s0 + a.max_steps;
        const int nc = a.cand_n[gi];
        FVec<CAP> &c0 = cols[s0 & 1];
        fv_clear(c0);
        wave_sync();
        if (nc > CAP) {
            err |= SP_ERR_CAPACITY;
            done_to = s0;
        } else {
            const uint32_t *cn = a.cand_node + (size_t)gi * PHMM_MAX_ACTIVE_NODES;
            const double *ct = a.cand_tot + (size_t)gi * PHMM_MAX_ACTIVE_NODES;
            for (int j = lane; j < nc; j += 64) {
                const double v = ct[j];
                const uint32_t id = cn[j];
                int rank = 0;
                for (int q = 0; q < nc; q++) {
                    const double u = ct[q];
                    rank += (u > v) || (u == v && cn[q] < id);
                }
                c0.id[rank] = id;
                c0.m[rank] = c0.i[rank] = c0.d[rank] = 0.0;
            }
            wave_sync();
            for (int j = lane; j < nc; j += 64) {
                const uint32_t cell = fv_cell(c0, c0.id[j]);
                c0.hslot[cell] = (uint16_t)j;
            }
            if (lane == 0) c0.n = nc;
            wave_sync();
            PrevRef<CAP> pr{};
            pr.vec = nullptr;
            pr.gm = a.d.Fm + ((size_t)g * a.d.Lc + (s0 - 1)) * NW;
            pr.gi = a.d.Fi + ((size_t)g * a.d.Lc + (s0 - 1)) * NW;
            pr.gd = a.d.Fd + ((size_t)g * a.d.Lc + (s0 - 1)) * NW;
            pr.W = a.W;
            pr.lane = r;
            pr.sc = 1.0;
            pr.E = a.d.FE[((size_t)g * (a.d.Lc + 1) + (s0 - 1)) * a.W + r];
            pr.is_init = false;
            fwd_adaptive_step<CAP>(a.M, pr, c0, sc, a.bases[((size_t)g * a.Lb + s0) * a.W + r], s0);
            if (!store_record<CAP>(a.pool, p0 + s0, c0)) err |= SP_ERR_POOL;
            pos = s0 + 1;
            done_to = pos;
        }
    } else {
        pos = a.stop[gi];
        done_to = pos;
        if (a.max_steps > 0 && pos + a.max_steps < len) end = pos + a.max_steps;
        if (!load_record_fvec<CAP>(a.pool, p0 + (uint64_t)(pos - 1), cols[(pos - 1) & 1])) {
            err |= SP_ERR_CAPACITY;  // the column to resume from does not fit this class
            end = pos;
        }
    }
    for (; pos < end && !err; pos++) {
        FVec<CAP> &prev = cols[(pos + 1) & 1];
        FVec<CAP> &cur = cols[pos & 1];
        select_top<CAP>(prev, cur, sc, a.topk == 0, a.ratio_lin, a.topk);
        PrevRef<CAP> p2{};
        p2.vec = &prev;
        p2.E = prev.E;
        p2.is_init = false;
        fwd_adaptive_step<CAP>(a.M, p2, cur, sc, a.bases[((size_t)g * a.Lb + pos) * a.W + r], pos);
        if (SMALL && sc.dropped) {
            err |= SP_ERR_CAPACITY;  // a 400-slot vector would have kept the insert: redo in a bigger class
            break;
        }
        if (!store_record<CAP>(a.pool, p0 + pos, cur)) {
            err |= SP_ERR_POOL;
            break;
        }
        done_to = pos + 1;
    }
    const bool finished = !err && done_to >= len;
    const double lp = finished ? fv_log_end(a.M, cols[(len - 1) & 1]) : NAN;
    if (lane == 0) {
        if (finished) a.out_logp[gi] = lp;
        a.stop[gi] = done_to;
        a.err[gi] = err;
    }
}

}  // namespace phmm
