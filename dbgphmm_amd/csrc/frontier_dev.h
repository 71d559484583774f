// Adaptive active-node frontier on one wave64: the device restatement of
//   PHMMTable::{to_nodevec, top_nodes_by_score_ratio, top_nodes, filled_nodes}
//       src/hmmv2/table.rs:117-149, 199-211
//   PHMMModel::{to_childs, to_childs_and_us, to_parents_and_us}
//       src/hmmv2/active_nodes.rs:15-56   (chain -> first-occurrence unique -> take(400))
//   f_step with is_adaptive = true      src/hmmv2/forward.rs:276-306, 423-466
//
// A column is ONE insertion-ordered vector of (node, m, i, d): the first `na` entries are
// the active list handed to f_step (they carry m and i), the rest are the nodes that only
// received a Del value, in the order the reference's `t0.d += ...` inserts them.  That is
// exactly the element order of the reference's `to_nodevec()`; the capacity of 400 is
// applied to the union (entries past it are dropped, which is what the reference's
// nodevec construction does to them anyway).
#pragma once

#include "sparse_dev.h"

namespace phmm {

template <int CAP> struct FVec {
    double m[CAP], i[CAP], d[CAP];
    uint32_t id[CAP];
    uint32_t hkey[HashSize<CAP>::N];
    uint16_t hslot[HashSize<CAP>::N];
    int n, na, E;
};
static constexpr uint16_t SLOT_NONE = 0xffff;

template <int CAP> __device__ __forceinline__ void fv_clear(FVec<CAP> &v) {
    for (int h = threadIdx.x; h < HashSize<CAP>::N; h += 64) {
        v.hkey[h] = H_EMPTY;
        v.hslot[h] = SLOT_NONE;
    }
    if (threadIdx.x == 0) v.n = v.na = 0;
}
template <int CAP> __device__ __forceinline__ int fv_find(const FVec<CAP> &v, uint32_t id) {
    uint32_t h = hash_of<CAP>(id);
    for (;;) {
        const uint32_t k = v.hkey[h];
        if (k == id) {
            const uint16_t s = v.hslot[h];
            return s == SLOT_NONE ? -1 : (int)s;
        }
        if (k == H_EMPTY) return -1;
        h = (h + 1) & (HashSize<CAP>::N - 1);
    }
}
// claim (or find) the hash cell of id; returns the cell index
template <int CAP> __device__ __forceinline__ uint32_t fv_cell(FVec<CAP> &v, uint32_t id) {
    uint32_t h = hash_of<CAP>(id);
    for (;;) {
        const uint32_t old = atomicCAS(&v.hkey[h], H_EMPTY, id);
        if (old == H_EMPTY || old == id) return h;
        h = (h + 1) & (HashSize<CAP>::N - 1);
    }
}

// find the hash cell of id without claiming one; -1 if absent
template <int CAP> __device__ __forceinline__ int fv_probe(const FVec<CAP> &v, uint32_t id) {
    uint32_t h = hash_of<CAP>(id);
    for (;;) {
        const uint32_t k = v.hkey[h];
        if (k == id) return (int)h;
        if (k == H_EMPTY) return -1;
        h = (h + 1) & (HashSize<CAP>::N - 1);
    }
}

// Scratch shared by the frontier primitives (LDS).
template <int CAP> struct FScratch {
    double tot[CAP];               // nodevec totals / level values A
    double lvb[CAP];               // level values B
    uint32_t arb[HashSize<CAP>::N];  // per hash cell: lowest candidate lane of the current chunk
    uint32_t pref[CAP + 1];        // exclusive prefix of degrees
    uint16_t order[CAP];           // sorted order (slots)
    uint16_t la[CAP], lb[CAP];     // level lists (slots into the column vector)
    uint8_t sta[CAP], stb[CAP];    // level stamps
    int nla, nlb, dropped;
};

// rank-by-counting stable descending sort of tot[0..n): order[rank] = slot
template <int CAP> __device__ __forceinline__ void sort_desc(const double *tot, int n, uint16_t *order) {
    for (int j = threadIdx.x; j < n; j += 64) {
        const double v = tot[j];
        int rank = 0;
        for (int q0 = 0; q0 < n; q0 += 8) {  // (eight LDS reads in flight instead of one at a time)
            double u[8];
#pragma unroll
            for (int k = 0; k < 8; k++) u[k] = tot[q0 + k < n ? q0 + k : n - 1];
#pragma unroll
            for (int k = 0; k < 8; k++) rank += (q0 + k < n) && ((u[k] > v) || (u[k] == v && q0 + k < j));
        }
        order[rank] = (uint16_t)j;
    }
}

// Append, in order and without duplicates, the children (or parents) of the nodes
// src_id[src_slot[0..nsrc)] to the vector v (capacity CAP; further new nodes are dropped).
// If lvl_list != nullptr the slots of ALL first occurrences at this level (already present
// or newly inserted) are written to it in order and stamped with `level` in `stamp`.
// `stamp[slot] == level` on entry marks slots that must not be listed again.
// The caller guarantees that v's hash cells carry SLOT_NONE for unassigned keys.
template <int CAP>
__device__ void append_neighbours(const SparseModel &M, bool children, FVec<CAP> &v, FScratch<CAP> &sc,
                                  const uint16_t *src_slot, int nsrc, uint16_t *lvl_list, int *lvl_n, uint8_t *stamp,
                                  uint8_t level) {
    const uint32_t *off = children ? M.chi_off : M.par_off;
    const uint32_t *nb = children ? M.chi_node : M.par_node;
    const int lane = threadIdx.x;
    // exclusive prefix of degrees in source order
    int running = 0;
    for (int base = 0; base < nsrc; base += 64) {
        const int j = base + lane;
        int deg = 0;
        if (j < nsrc) {
            const uint32_t k = v.id[src_slot[j]];
            if (M.packed) deg = children ? (int)M.fadj[k].nchi : (int)M.fadj[k].npar;
            else deg = (int)(off[k + 1] - off[k]);
        }
        const int inc = wave_iscan(deg);
        if (j < nsrc) sc.pref[j] = (uint32_t)(running + inc - deg);
        running += __shfl(inc, 63);
    }
    if (lane == 0) sc.pref[nsrc] = (uint32_t)running;
    wave_sync();
    const int total = running;
    int nl = lvl_list ? *lvl_n : 0;
    for (int cbase = 0; cbase < total; cbase += 64) {
        const int c = cbase + lane;
        bool valid = c < total;
        uint32_t key = 0, cell = 0;
        if (valid) {
            // source j with pref[j] <= c < pref[j+1]
            int lo = 0, hi = nsrc - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if ((int)sc.pref[mid] <= c) lo = mid;
                else hi = mid - 1;
            }
            const uint32_t k = v.id[src_slot[lo]];
            const uint32_t q = (uint32_t)(c - (int)sc.pref[lo]);
            if (M.packed) key = children ? M.fadj[k].chi[q] : M.fadj[k].par[q];  // (CSR order kept in the records)
            else key = nb[off[k] + q];
            if (v.n < CAP) {
                cell = fv_cell(v, key);
            } else {
                // the vector is full: unknown nodes are dropped; do not let them fill the hash
                const int pc = fv_probe(v, key);
                valid = pc >= 0;
                cell = valid ? (uint32_t)pc : 0u;
                if (!valid) sc.dropped = 1;
            }
            if (valid) sc.arb[cell] = 0xffffffffu;
        }
        wave_sync();
        if (valid) atomicMin(&sc.arb[cell], (uint32_t)lane);
        wave_sync();
        const bool winner = valid && sc.arb[cell] == (uint32_t)lane;
        int slot = winner ? (int)v.hslot[cell] : -2;
        const bool is_new = winner && slot == (int)SLOT_NONE;
        // new nodes get consecutive slots in candidate order
        const unsigned long long newmask = __ballot(is_new);
        const int nbefore = __popcll(newmask & ((1ull << lane) - 1ull));
        const int n0 = v.n;
        if (is_new) {
            const int s = n0 + nbefore;
            if (s < CAP) {
                v.id[s] = key;
                v.m[s] = 0.0;
                v.i[s] = 0.0;
                v.d[s] = 0.0;
                v.hslot[cell] = (uint16_t)s;
                if (stamp) stamp[s] = 0xff;
                slot = s;
            } else {
                slot = -1;  // dropped (the cell keeps SLOT_NONE: later lookups miss)
            }
        }
        wave_sync();
        if (lane == 0) {
            const int add = __popcll(newmask);
            v.n = n0 + add > CAP ? CAP : n0 + add;
            if (n0 + add > CAP) sc.dropped = 1;
        }
        if (lvl_list) {
            const bool list_it = winner && slot >= 0 && stamp[slot] != level;
            const unsigned long long lm = __ballot(list_it);
            if (list_it) {
                const int p = nl + __popcll(lm & ((1ull << lane) - 1ull));
                lvl_list[p] = (uint16_t)slot;
                stamp[slot] = level;
            }
            nl += __popcll(lm);
        }
        wave_sync();
    }
    if (lvl_list && lane == 0) *lvl_n = nl;
    wave_sync();
}

// Where the previous column lives for the step being computed.
template <int CAP> struct PrevRef {
    const FVec<CAP> *vec;       // sparse previous column (LDS) or nullptr
    const double *gm, *gi, *gd;  // dense previous column (global, [node][W]) when vec == nullptr
    int W, lane;                // read-group width and this read's lane in the group
    double sc;                  // rescale of the dense column (2^-e of its maximum)
    int E;                      // exponent of the previous column after `sc`
    bool is_init;               // f_init (forward.rs:255-266)
};

template <int CAP>
__device__ __forceinline__ void prev_get(const PrevRef<CAP> &p, uint32_t node, double &m, double &i, double &d) {
    m = i = d = 0.0;
    if (p.is_init) return;
    if (p.vec) {
        const int s = fv_find(*p.vec, node);
        if (s >= 0) {
            m = p.vec->m[s];
            i = p.vec->i[s];
            d = p.vec->d[s];
        }
    } else {
        const size_t ix = (size_t)node * p.W + p.lane;
        m = p.gm[ix] * p.sc;
        i = p.gi[ix] * p.sc;
        d = p.gd[ix] * p.sc;
    }
}

// One adaptive forward column (f_step, is_dense = false, is_adaptive = true).
// On entry cur holds the `ntop` top nodes of the previous column in sorted order
// (cur.n == ntop, hash filled, m/i/d zero).  Implements to_childs_and_us, fm, fi, fib,
// the adaptive fd and the rescale; cur.na is the active list length afterwards.
template <int CAP>
__device__ void fwd_adaptive_step(const SparseModel &M, const PrevRef<CAP> &prev, FVec<CAP> &cur, FScratch<CAP> &sc,
                                  uint8_t x, int pos) {
    const LinParams &lp = M.lp;
    const int lane = threadIdx.x;
    const int ntop = cur.n;
    // active = to_childs_and_us(top)  (forward.rs:147; active_nodes.rs:23-35)
    for (int j = lane; j < ntop; j += 64) sc.order[j] = (uint16_t)j;
    wave_sync();
    append_neighbours<CAP>(M, true, cur, sc, sc.order, ntop, nullptr, nullptr, nullptr, 0);
    const int na = cur.n;
    if (lane == 0) cur.na = na;
    wave_sync();
    // fm (forward.rs:337-359), fi (378-388), fib (541-545)
    const bool first = prev.is_init;
    const double ibs = first ? 0.0 : exp(M.logib[pos - 1] - (double)prev.E * SP_LN2);
    const double c_begin = first ? lp.p_MM : lp.p_IM * ibs;
    const double ib_cur = first ? lp.p_random * lp.p_MI : lp.p_random * lp.p_II * ibs;
    const double c_del = lp.p_ID * ib_cur;
    for (int j = lane; j < na; j += 64) {
        const uint32_t k = cur.id[j];
        double acc = 0.0, pe, ini;
        if (M.packed) {
            const FwdAdj r = M.fadj[k];
            pe = r.emis == x ? lp.p_match : lp.p_mismatch;
            ini = r.init;
#pragma unroll
            for (int q = 0; q < ADJ_DEG; q++) {
                if (q >= (int)r.npar || r.par_w[q] == 0.0) continue;
                double pm, pi, pd;
                prev_get(prev, r.par[q], pm, pi, pd);
                acc += r.par_w[q] * (lp.p_MM * pm + lp.p_IM * pi + lp.p_DM * pd);
            }
        } else {
            pe = M.emis[k] == x ? lp.p_match : lp.p_mismatch;
            ini = M.init[k];
            for (uint32_t a = M.par_off[k]; a < M.par_off[k + 1]; a++) {
                const double w = M.par_w[a];
                if (w == 0.0) continue;
                double pm, pi, pd;
                prev_get(prev, M.par_node[a], pm, pi, pd);
                acc += w * (lp.p_MM * pm + lp.p_IM * pi + lp.p_DM * pd);
            }
        }
        double om, oi, od;
        prev_get(prev, k, om, oi, od);
        cur.m[j] = pe * (acc + ini * c_begin);
        cur.i[j] = lp.p_random * (lp.p_MI * om + lp.p_II * oi + lp.p_DI * od);
    }
    for (int j = lane; j < CAP; j += 64) {
        sc.sta[j] = 0xff;
        sc.stb[j] = 0xff;
    }
    if (lane == 0) sc.nla = sc.nlb = 0;
    wave_sync();
    // adaptive fd (forward.rs:423-466): S0 = to_childs(active), S_t = to_childs(S_{t-1})
    for (int j = lane; j < na; j += 64) sc.order[j] = (uint16_t)j;
    wave_sync();
    uint16_t *src = sc.order;
    int nsrc = na;
    for (int t = 0; t <= lp.n_max_gaps; t++) {
        uint16_t *lst = (t & 1) ? sc.lb : sc.la;
        int *ln = (t & 1) ? &sc.nlb : &sc.nla;
        uint8_t *st_cur = (t & 1) ? sc.stb : sc.sta;
        const uint8_t *st_prev = (t & 1) ? sc.sta : sc.stb;
        double *lv_cur = (t & 1) ? sc.lvb : sc.tot;
        const double *lv_prev = (t & 1) ? sc.tot : sc.lvb;
        if (lane == 0) *ln = 0;
        wave_sync();
        append_neighbours<CAP>(M, true, cur, sc, src, nsrc, lst, ln, st_cur, (uint8_t)t);
        const int nl = *ln;
        for (int j = lane; j < nl; j += 64) {
            const int s = lst[j];
            const uint32_t k = cur.id[s];
            double acc = 0.0, ini = 0.0;
            auto term = [&](double w, uint32_t parent) {
                if (w == 0.0) return;
                const int ps = fv_find(cur, parent);
                if (ps < 0) return;
                if (t == 0) {
                    if (ps < na) acc += w * (lp.p_MD * cur.m[ps] + lp.p_ID * cur.i[ps]);  // fd0, forward.rs:480-501
                } else if (st_prev[ps] == (uint8_t)(t - 1)) {
                    acc += w * lv_prev[ps];  // fdt, forward.rs:510-524
                }
            };
            if (M.packed) {
                const FwdAdj r = M.fadj[k];
                ini = r.init;
#pragma unroll
                for (int q = 0; q < ADJ_DEG; q++)
                    if (q < (int)r.npar) term(r.par_w[q], r.par[q]);
            } else {
                ini = M.init[k];
                for (uint32_t a = M.par_off[k]; a < M.par_off[k + 1]; a++) term(M.par_w[a], M.par_node[a]);
            }
            const double val = t == 0 ? acc + ini * c_del : lp.p_DD * acc;
            lv_cur[s] = val;
            cur.d[s] += val;
        }
        wave_sync();
        src = lst;
        nsrc = nl;
    }
    // rescale so that the column maximum is in [0.5, 1)
    double mx = ib_cur;
    for (int j = lane; j < cur.n; j += 64) mx = fmax(mx, fmax(fmax(cur.m[j], cur.i[j]), cur.d[j]));
    mx = wave_max(mx);
    const int e = sp_exp_of(mx);
    const double s = sp_pow2(-e);
    for (int j = lane; j < cur.n; j += 64) {
        cur.m[j] *= s;
        cur.i[j] *= s;
        cur.d[j] *= s;
    }
    if (lane == 0) cur.E = prev.E + e;
    wave_sync();
}

// top_nodes_by_score_ratio / top_nodes of a sparse column (table.rs:127-149) written as the
// first entries of `cur` (cleared here).  by_ratio: keep while ln p0 - ln p < max_ratio;
// otherwise keep the first k.
template <int CAP>
__device__ void select_top(const FVec<CAP> &prev, FVec<CAP> &cur, FScratch<CAP> &sc, bool by_ratio, double ratio_lin,
                           int k) {
    const int lane = threadIdx.x;
    const int n = prev.n;
    fv_clear(cur);
    for (int j = lane; j < n; j += 64) sc.tot[j] = prev.m[j] + prev.i[j] + prev.d[j];
    wave_sync();
    sort_desc<CAP>(sc.tot, n, sc.order);
    wave_sync();
    int ntop = 0;
    if (n > 0) {
        const double t0 = sc.tot[sc.order[0]];
        if (by_ratio) {
            int cnt = 0;
            for (int j = lane; j < n; j += 64) cnt += (sc.tot[j] > 0.0 && sc.tot[j] > t0 * ratio_lin) ? 1 : 0;
            ntop = wave_isum(cnt);
        } else {
            ntop = k < n ? k : n;
        }
    }
    for (int j = lane; j < ntop; j += 64) {
        const uint32_t id = prev.id[sc.order[j]];
        cur.id[j] = id;
        cur.m[j] = cur.i[j] = cur.d[j] = 0.0;
        const uint32_t cell = fv_cell(cur, id);
        cur.hslot[cell] = (uint16_t)j;
    }
    if (lane == 0) cur.n = ntop;
    wave_sync();
}

template <int CAP> __device__ __forceinline__ double fv_log_end(const SparseModel &M, const FVec<CAP> &c) {
    double s = 0.0;
    for (int j = threadIdx.x; j < c.na; j += 64) s += c.m[j] + c.i[j] + c.d[j];
    s = wave_sum(s);
    return log(M.lp.p_end * s) + (double)c.E * SP_LN2;
}

}  // namespace phmm
