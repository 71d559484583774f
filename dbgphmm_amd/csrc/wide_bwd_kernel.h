// The 400-slot class of backward_by_forward + to_mapping_by_score_ratio on ONE BLOCK of 448 threads per read
// (7 waves; thread = slot), the counterpart of wide_fwd_kernel.h.
//
// sparse_backward_kernel<400> (mapping_flow.hip) walks a read position with one wave: the forward record's totals are
// sorted (400 x 400 comparisons on 64 lanes), the column B.tables[pos] is computed over the record's nodes (backward.rs:
// 122-129, 216-261: bd0 + n_max_gaps x bdt restricted to the list, bm, bi), the posterior S = F (.) B / P
// (table.rs:500-505) is cut at the score ratio and sorted again (hint.rs:135-142).  Every one of these passes is a loop
// over the slots: seven iterations of a wave on a 400-node column, one here.  Same statements, same per-node summation
// order (the packed records keep the CSR order of a node's children); a slot's adjacency record and the slots of its
// in-list children are found once per position and stay in the thread's registers through the Del sweeps.
// Score-ratio lists only (topk == 0), no list mode, packed records: the host keeps the generic kernel otherwise.
#pragma once

#include "block_sort.h"
#include "sparse_dyn.h"

namespace phmm {

static constexpr int WBK_T = 448;
static constexpr int WBK_WAVES = WBK_T / 64;
static constexpr int WBK_CAP = PHMM_MAX_ACTIVE_NODES;
static constexpr int WBK_HASH = 1024;

struct WBCol {
    double m[WBK_T], i[WBK_T], d[WBK_T];
    uint32_t id[WBK_T];
    uint32_t hkey[WBK_HASH];
    uint16_t hslot[WBK_HASH];
    int n, E;
};
struct WideBwdShared {
    WBCol col[2];
    double fm[WBK_T], fi[WBK_T], fd[WBK_T];  // the forward record
    uint32_t fid[WBK_T];
    double dA[512], dB[512];      // level buffers; outside the column step: val = dA, order = dB, the list entries being
                                  // sorted = (dB, list)
    uint32_t list[512];
    double red[2][8];
    int wsum[2][8];
    unsigned long long bc;
    LinParams lp;  // (read from here: the scalar registers are short)
};

__device__ __forceinline__ uint32_t wb_hash(uint32_t id) { return (id * 2654435761u) >> 22; }
__device__ __forceinline__ void wb_insert(WBCol &c, uint32_t id, int slot) {
    uint32_t h = wb_hash(id);
    for (;;) {
        const uint32_t old = atomicCAS(&c.hkey[h], H_EMPTY, id);
        if (old == H_EMPTY) {
            c.hslot[h] = (uint16_t)slot;
            return;
        }
        if (old == id) return;
        h = (h + 1) & (WBK_HASH - 1);
    }
}
__device__ __forceinline__ int wb_find(const WBCol &c, uint32_t id) {
    uint32_t h = wb_hash(id);
    for (;;) {
        const uint32_t k = c.hkey[h];
        if (k == id) return (int)c.hslot[h];
        if (k == H_EMPTY) return -1;
        h = (h + 1) & (WBK_HASH - 1);
    }
}
// (`par` alternates between two sets of wave results: one barrier per reduction)
__device__ __forceinline__ double wb_block_max(WideBwdShared &sh, int &par, double v) {  // non-negative values
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh.red[par][threadIdx.x >> 6] = v;
    __syncthreads();
    double r = sh.red[par][0];
#pragma unroll
    for (int w = 1; w < WBK_WAVES; w++) r = fmax(r, sh.red[par][w]);
    par ^= 1;
    return r;
}
__device__ __forceinline__ int wb_block_isum(WideBwdShared &sh, int &par, int v) {
    v = wave_isum(v);
    if ((threadIdx.x & 63) == 0) sh.wsum[par][threadIdx.x >> 6] = v;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int w = 0; w < WBK_WAVES; w++) r += sh.wsum[par][w];
    par ^= 1;
    return r;
}

// forward record of a position into sh.f*; false: missing (or larger than the class)
__device__ __forceinline__ bool wb_load_record(const RecPool &p, uint64_t pos_index, WideBwdShared &sh, int &n, int &na,
                                               int &E) {
    const uint64_t o1 = p.off[pos_index];
    if (o1 == 0) return false;
    const uint8_t *rec = p.base + (o1 - 8);
    const int *hw = (const int *)rec;
    n = hw[0];
    na = hw[1];
    E = hw[2];
    if (n > WBK_CAP) return false;
    const uint64_t idb = (uint64_t)((n + 1) & ~1) * 4;
    const uint32_t *ids = (const uint32_t *)(rec + 16);
    const double *m = (const double *)(rec + 16 + idb), *i = m + na, *d = i + na;
    const int t = threadIdx.x;
    if (t < n) {
        sh.fid[t] = ids[t];
        sh.fd[t] = d[t];
        sh.fm[t] = t < na ? m[t] : 0.0;
        sh.fi[t] = t < na ? i[t] : 0.0;
    }
    __syncthreads();
    return true;
}

__device__ __forceinline__ int wb_excl_scan(WideBwdShared &sh, int &par, int v, int &total) {
    const int inc = wave_iscan(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) sh.wsum[par][w] = inc;
    __syncthreads();
    int pre = 0, tt = 0;
#pragma unroll
    for (int k = 0; k < WBK_WAVES; k++) {
        const int x = sh.wsum[par][k];
        tt += x;
        pre += k < w ? x : 0;
    }
    total = tt;
    par ^= 1;
    return pre + inc - v;
}

// emit_mapping (mapping_flow.hip) with by_node = true, topk = 0: val[0..n) -> the list record of a position.  Only the
// entries that stay are sorted (a column next to a wide spot has 400 entries of which a few dozen stay).
__device__ __forceinline__ bool wb_emit(const RecPool &mp, uint64_t pos_index, WideBwdShared &sh, int &par, int n,
                                        double ratio_lin) {
    const int t = threadIdx.x;
    const double v = t < n ? sh.dA[t] : 0.0;
    double *val = sh.dB;       // the list being built: log values ...
    uint32_t *ids = sh.list;   // ... and node ids
    const double thr = wb_block_max(sh, par, v) * ratio_lin;
    const bool stay = t < n && v > 0.0 && v > thr;
    int keep;
    const int at = wb_excl_scan(sh, par, stay ? 1 : 0, keep);
    // the list is ordered by the values it holds -- the LOGS -- and equal logs by node id (emit_mapping)
    const int NP = bitonic_size(keep);
    if (stay) {
        val[at] = log(v);
        ids[at] = sh.fid[t];
    }
    for (int k = t; k < NP; k += WBK_T)
        if (k >= keep) {
            val[k] = -INFINITY;
            ids[k] = 0xffffffffu;
        }
    block_bitonic_desc(val, ids, NP);
    const uint64_t idb = (uint64_t)((keep + 1) & ~1) * 4;
    const uint64_t bytes = 8 + idb + (uint64_t)keep * 8;
    if (t == 0) sh.bc = atomicAdd(mp.top, (unsigned long long)bytes);
    __syncthreads();
    const uint64_t o = sh.bc;
    if (o + bytes > mp.cap) return false;
    uint8_t *rec = mp.base + o;
    if (t == 0) {
        ((uint32_t *)rec)[0] = (uint32_t)keep;
        ((uint32_t *)rec)[1] = 0;
        mp.off[pos_index] = o + 8;
    }
    uint32_t *oid = (uint32_t *)(rec + 8);
    double *olp = (double *)(rec + 8 + idb);
    if (t < keep) {
        oid[t] = ids[t];
        olp[t] = val[t];
    }
    __syncthreads();
    return true;
}

__global__ void __launch_bounds__(WBK_T) wide_backward_kernel(const SparseBwdArgs a) {
    __shared__ WideBwdShared sh;
    const int t = threadIdx.x;
    const uint32_t gi = a.lanes[blockIdx.x];
    const int g = (int)(gi / a.W), r = (int)(gi % a.W);
    const int len = a.d.len[gi];
    const int s0 = a.sw[gi];
    const uint64_t p0 = a.lane_pos0[gi];
    const uint64_t q0 = a.map_pos0[gi];
    const double logP = a.d.logPf[gi];
    if (t == 0) sh.lp = a.M.lp;
    __syncthreads();
    const LinParams &lp = sh.lp;
    uint32_t err = 0;
    int par = 0;
    const bool ok = logP > -INFINITY;
    int pos;            // next position to compute
    int have_cols = 0;  // col[(pos+1)&1] holds B.tables[pos+1]
    bool stopped = false;
    int stop_at = 0;
    int fn = 0, fna = 0, fE = 0;
    if (a.mode == 0) {
        pos = len - 1;
        // merged index len: F.tables[len-1] (.) b_init / P   (table.rs:414-434, backward.rs:197-211)
        if (!wb_load_record(a.fpool, p0 + (uint64_t)(len - 1), sh, fn, fna, fE)) {
            stopped = true;  // missing: nothing done
            stop_at = len;
        } else {
            const double w = ok ? exp((double)fE * SP_LN2 - logP) * lp.p_end : 0.0;
            if (t < fn) sh.dA[t] = w * (sh.fm[t] + sh.fi[t] + sh.fd[t]);
            __syncthreads();
            if (!wb_emit(a.mpool, q0 + (uint64_t)(len - 1), sh, par, fn, a.ratio_lin)) err |= SP_ERR_POOL;
        }
    } else {
        pos = a.stop[gi];
        if (pos < len - 1) {
            // B.tables[pos+1] from the hand-off slot
            const BHandoff &h = a.hand[gi];
            WBCol &c = sh.col[(pos + 1) & 1];
            const int n = h.n;
            for (int k = t; k < WBK_HASH; k += WBK_T) c.hkey[k] = H_EMPTY;
            if (t == 0) {
                c.n = n;
                c.E = h.E;
            }
            __syncthreads();
            if (t < n) {
                c.id[t] = h.id[t];
                c.m[t] = h.m[t];
                c.i[t] = h.i[t];
                c.d[t] = h.d[t];
                wb_insert(c, h.id[t], t);
            }
            __syncthreads();
            have_cols = 1;
        }
    }
    int steps_done = 0;
    for (; !stopped && pos >= s0 + 1 && !err; pos--) {
        // B.tables[pos] over filled_nodes(F.tables[pos-1]) (backward.rs:122-129)
        if (!wb_load_record(a.fpool, p0 + (uint64_t)(pos - 1), sh, fn, fna, fE)) {
            stopped = true;
            stop_at = pos;
            break;
        }
        double *val = sh.dA;
        uint16_t *order = (uint16_t *)sh.dB;
        // filled_nodes (table.rs:117-123): the record's elements by total, stable (sort_desc, frontier_dev.h)
        {
            const int NP = bitonic_size(fn);
            for (int k = t; k < NP; k += WBK_T) {
                val[k] = k < fn ? sh.fm[k] + sh.fi[k] + sh.fd[k] : -1.0;
                order[k] = (uint16_t)k;
            }
            block_bitonic_desc(val, order, NP);
        }
        const int nl = fna < fn ? fna : fn;
        if (t < nl) sh.list[t] = sh.fid[order[t]];
        __syncthreads();
        // ---- one backward column over the list (bwd_list_step, sparse_dev.h)
        const WBCol &prev = sh.col[(pos + 1) & 1];
        WBCol &cur = sh.col[pos & 1];
        const bool prev_is_init = pos == len - 1;
        const uint8_t x = a.bases[((size_t)g * a.Lb + pos) * a.W + r];
        for (int k = t; k < WBK_HASH; k += WBK_T) cur.hkey[k] = H_EMPTY;
        if (t == 0) cur.n = nl;
        __syncthreads();
        BwdAdj rec;
        rec.nchi = 0;
        if (t < nl) {
            const uint32_t k = sh.list[t];
            cur.id[t] = k;
            wb_insert(cur, k, t);
            rec = a.M.badj[k];
        }
        __syncthreads();
        const double pend = lp.p_end;
        // bd0 (backward.rs:354-377); keep A1 = sum_w t e_w m'[w] and q0 = p_r i'[v] for bm/bi
        int cs[ADJ_DEG];
        double a1 = 0.0, qq = 0.0;
        if (t < nl) {
#pragma unroll
            for (int q = 0; q < ADJ_DEG; q++) {
                cs[q] = -1;
                if (q >= (int)rec.nchi) continue;
                const double w = rec.chi_w[q];
                if (w == 0.0) continue;
                const uint32_t u = rec.chi[q];
                double mu = 0.0;
                if (prev_is_init) mu = pend;
                else {
                    const int ps = wb_find(prev, u);
                    if (ps >= 0) mu = prev.m[ps];
                }
                a1 += w * (rec.chi_emis[q] == x ? lp.p_match : lp.p_mismatch) * mu;
                cs[q] = wb_find(cur, u);
            }
            double iv = 0.0;
            if (prev_is_init) iv = pend;
            else {
                const int os = wb_find(prev, cur.id[t]);
                if (os >= 0) iv = prev.i[os];
            }
            qq = lp.p_random * iv;
            const double d0 = lp.p_DM * a1 + lp.p_DI * qq;
            cur.d[t] = d0;
            sh.dA[t] = d0;
        }
        __syncthreads();
        // bdt (backward.rs:387-404), restricted to the list
        double *src = sh.dA, *dst = sh.dB;
        for (int lv = 1; lv <= lp.n_max_gaps; lv++) {
            if (t < nl) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < ADJ_DEG; q++)
                    if (cs[q] >= 0) s += rec.chi_w[q] * src[cs[q]];
                s *= lp.p_DD;
                dst[t] = s;
                cur.d[t] += s;
            }
            __syncthreads();
            double *tmp = src;
            src = dst;
            dst = tmp;
        }
        // bm, bi: sum_w t (p_XM e_w m'[w] + p_XD d[w]) + p_XI p_r i'[v]
        double nm = 0.0, ni = 0.0, nd = 0.0;
        if (t < nl) {
            double td = 0.0;
#pragma unroll
            for (int q = 0; q < ADJ_DEG; q++)
                if (cs[q] >= 0) td += rec.chi_w[q] * cur.d[cs[q]];
            nm = lp.p_MM * a1 + lp.p_MD * td + lp.p_MI * qq;
            ni = lp.p_IM * a1 + lp.p_ID * td + lp.p_II * qq;
            nd = cur.d[t];
        }
        // rescale (col_rescale): the column maximum into [0.5, 1)
        const double mx = wb_block_max(sh, par, fmax(fmax(nm, ni), nd));
        const int e = sp_exp_of(mx);
        const double sc = sp_pow2(-e);
        const int Ecur = (prev_is_init ? 0 : prev.E) + e;
        if (t < nl) {
            cur.m[t] = nm * sc;
            cur.i[t] = ni * sc;
            cur.d[t] = nd * sc;
        }
        if (t == 0) cur.E = Ecur;
        __syncthreads();
        have_cols = 1;
        // S = F.tables[pos-1] (.) B.tables[pos] / P over F's elements (table.rs:320-345, 500-505)
        const double w = ok ? exp((double)(fE + Ecur) * SP_LN2 - logP) : 0.0;
        if (t < fn) {
            const int bs = wb_find(cur, sh.fid[t]);
            val[t] = bs >= 0 ? w * (sh.fm[t] * cur.m[bs] + sh.fi[t] * cur.i[bs] + sh.fd[t] * cur.d[bs]) : 0.0;
        }
        __syncthreads();
        if (!wb_emit(a.mpool, q0 + (uint64_t)(pos - 1), sh, par, fn, a.ratio_lin)) err |= SP_ERR_POOL;
        // a burst (max_steps) ends where the column fits the one-lane-per-node class again
        if (a.max_steps > 0 && !err && ++steps_done >= a.max_steps && nl <= 64 && pos - 1 >= s0 + 1) {
            stopped = true;
            stop_at = pos - 1;
            break;
        }
    }
    if (stopped && !err) {
        // park B.tables[stop_at + 1] for the next phase
        if (have_cols && stop_at < len) {
            const WBCol &c = sh.col[(stop_at + 1) & 1];
            BHandoff &h = a.hand[gi];
            if (c.n > HANDOFF_CAP) err |= SP_ERR_CAPACITY;
            else {
                if (t == 0) {
                    h.n = c.n;
                    h.E = c.E;
                }
                if (t < c.n) {
                    h.id[t] = c.id[t];
                    h.m[t] = c.m[t];
                    h.i[t] = c.i[t];
                    h.d[t] = c.d[t];
                }
            }
        }
        if (t == 0) a.stop[gi] = stop_at;
    } else if (!err) {
        // hand B.tables[s0+1] to the dense backward kernel: dense column (zeros elsewhere), its exponent and maximum
        if (have_cols) {
            const WBCol &c = sh.col[(s0 + 1) & 1];
            const size_t NW = (size_t)a.d.N * a.W;
            const int pc = (s0 + 1) & 1;
            double *bm = a.d.Bm + ((size_t)g * a.d.bcols + pc) * NW;
            double *bi = a.d.Bi + ((size_t)g * a.d.bcols + pc) * NW;
            double mx = 0.0;
            if (t < c.n) {
                bm[(size_t)c.id[t] * a.W + r] = c.m[t];
                bi[(size_t)c.id[t] * a.W + r] = c.i[t];
                mx = fmax(c.m[t], c.i[t]);
            }
            mx = wb_block_max(sh, par, mx);
            if (t == 0) {
                a.d.cmaxB[((size_t)g * a.d.Lc + (s0 + 1)) * a.W + r] = (unsigned long long)__double_as_longlong(mx);
                a.d.BE[((size_t)g * (a.d.Lc + 1) + (s0 + 1)) * a.W + r] = c.E;
            }
        }
        if (t == 0) a.stop[gi] = s0;
    }
    if (t == 0) a.err[gi] = err;
}

}  // namespace phmm
