// The 400-slot class of the adaptive sparse forward on ONE BLOCK of 448 threads per read (7 waves; thread = slot).
//
// sparse_forward_kernel<400> (sparse_fwd_kernel.h + frontier_dev.h) restates the reference's insertion-ordered
// 400-element vectors --
//   PHMMTable::{to_nodevec, top_nodes_by_score_ratio}     src/hmmv2/table.rs:117-149
//   PHMMModel::{to_childs, to_childs_and_us}              src/hmmv2/active_nodes.rs:15-56 (chain -> unique -> take(400))
//   f_step, is_adaptive = true                            src/hmmv2/forward.rs:276-306, 337-388, 423-524
// -- with one wave: a frontier of 100-400 nodes is walked 64 candidates at a time, with a binary search, a hash claim,
// an arbitration and four wave-level sync points per batch, six expansions per read position: 0.15-0.35 ms per read
// position, the whole time of a read set on a short-unit tandem repeat (bench `rep20`).  This kernel is the SAME
// algorithm, statement by statement, on seven waves: a batch is 448 candidates, a slot's values are computed by the
// thread of that index out of its own registers (the slot's packed adjacency record, fetched once when the slot is
// given out), and a sync point is a block barrier.  Element order, truncation at 400 elements (first occurrences in
// candidate order are kept), per-node summation order and the final sum's order are the generic kernel's: the two
// produce the same bits, which is what tests/test_gpu_repeats.py holds this one to (PHMM_NO_WIDE_CLASS=1 runs the
// generic one).  Needs the packed records (every node at most ADJ_DEG parents and children) and the score-ratio list
// (use_max_ratio = true); the host keeps the generic kernel for everything else.
#pragma once

#include "block_sort.h"
#include "sparse_fwd_kernel.h"

namespace phmm {

static constexpr int WFK_T = 448;  // threads = slots (7 waves)
static constexpr int WFK_WAVES = WFK_T / 64;
static constexpr int WFK_CAP = PHMM_MAX_ACTIVE_NODES;
static constexpr int WFK_HASH = 2048;  // 400 elements + the 448 candidates of the batch that filled the vector, at a load under 0.42
static_assert(WFK_CAP <= WFK_T && WFK_CAP + WFK_T < WFK_HASH, "slot / hash sizes");

struct WCol {
    double m[WFK_CAP], i[WFK_CAP], d[WFK_CAP];  // (slots: a column never holds more than 400 elements)
    uint32_t id[WFK_CAP];
    uint32_t hkey[WFK_HASH];
    uint16_t hslot[WFK_HASH];
    int n, na, E;
};
struct WideFwdShared {
    WCol col[2];
    double tot[WFK_T];             // level values A (also: the first column's candidate totals)
    double lvb[512];               // level values B / the top elements' totals while they are sorted
    uint32_t arb[WFK_HASH];        // per hash cell: (batch tag | first candidate of that batch); tags count DOWN, so a
                                   // cell never needs clearing between batches (atomicMin)
    uint32_t cand[WFK_T];          // the candidate keys of a batch, in candidate order (also: the first column's candidate ids)
    uint32_t chi[ADJ_DEG][WFK_T];  // children of the element in a slot
    uint16_t order[512], la[WFK_T], lb[WFK_T];
    uint8_t sta[WFK_T], stb[WFK_T], nchi[WFK_T];
    int wsum[2][8];
    double red[2][8];
    unsigned long long bc;
    LinParams lp;  // (read from here: thirteen doubles fewer in scalar registers, which the kernel was spilling)
};

__device__ __forceinline__ uint32_t wf_hash(uint32_t id) { return (id * 2654435761u) >> 21; }
__device__ __forceinline__ void wf_clear(WCol &c) {
    for (int h = threadIdx.x; h < WFK_HASH; h += WFK_T) {
        c.hkey[h] = H_EMPTY;
        c.hslot[h] = SLOT_NONE;
    }
    if (threadIdx.x == 0) c.n = c.na = 0;
}
__device__ __forceinline__ int wf_find(const WCol &c, uint32_t id) {
    uint32_t h = wf_hash(id);
    for (;;) {
        const uint32_t k = c.hkey[h];
        if (k == id) {
            const uint16_t s = c.hslot[h];
            return s == SLOT_NONE ? -1 : (int)s;
        }
        if (k == H_EMPTY) return -1;
        h = (h + 1) & (WFK_HASH - 1);
    }
}
__device__ __forceinline__ uint32_t wf_cell(WCol &c, uint32_t id) {
    uint32_t h = wf_hash(id);
    for (;;) {
        const uint32_t old = atomicCAS(&c.hkey[h], H_EMPTY, id);
        if (old == H_EMPTY || old == id) return h;
        h = (h + 1) & (WFK_HASH - 1);
    }
}
__device__ __forceinline__ int wf_probe(const WCol &c, uint32_t id) {
    uint32_t h = wf_hash(id);
    for (;;) {
        const uint32_t k = c.hkey[h];
        if (k == id) return (int)h;
        if (k == H_EMPTY) return -1;
        h = (h + 1) & (WFK_HASH - 1);
    }
}

// exclusive prefix sum over the block's threads (and the total); `par` alternates between two sets of wave totals so
// that one barrier per scan is enough
__device__ __forceinline__ int wf_excl_scan(WideFwdShared &sh, int &par, int v, int &total) {
    const int inc = wave_iscan(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) sh.wsum[par][w] = inc;
    __syncthreads();
    int pre = 0, tt = 0;
#pragma unroll
    for (int k = 0; k < WFK_WAVES; k++) {
        const int x = sh.wsum[par][k];
        tt += x;
        pre += k < w ? x : 0;
    }
    total = tt;
    par ^= 1;
    return pre + inc - v;
}
__device__ __forceinline__ double wf_block_max(WideFwdShared &sh, int &par, double v) {  // non-negative values
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh.red[par][threadIdx.x >> 6] = v;
    __syncthreads();
    double r = sh.red[par][0];
#pragma unroll
    for (int w = 1; w < WFK_WAVES; w++) r = fmax(r, sh.red[par][w]);
    par ^= 1;
    return r;
}

// The record of the element this thread's slot has just been given: the thread keeps it, its children go to LDS
// (the expansions read them by SOURCE order, not by slot).
__device__ __forceinline__ void wf_fetch(const SparseModel &M, WideFwdShared &sh, const WCol &c, FwdAdj &rec) {
    const int t = threadIdx.x;
    rec = M.fadj[c.id[t]];
    sh.nchi[t] = rec.nchi;
#pragma unroll
    for (int q = 0; q < ADJ_DEG; q++) sh.chi[q][t] = rec.chi[q];
}

// append_neighbours (frontier_dev.h) for children: the children of the elements src_slot[0..nsrc) are appended to v in
// source order without duplicates, up to 400 elements; with a level list, ALL first occurrences at this level are
// listed in order and stamped.
__device__ __forceinline__ int wf_append(const SparseModel &M, WideFwdShared &sh, WCol &v, int &par, uint32_t &tag,
                                         FwdAdj &rec, const uint16_t *src_slot, int nsrc, uint16_t *lvl_list,
                                         uint8_t *stamp, uint8_t level) {
    const int t = threadIdx.x;
    // thread t < nsrc speaks for source t: its children are the candidates ex .. ex + deg - 1
    int total;
    const int sslot = t < nsrc ? (int)src_slot[t] : 0;
    const int deg = t < nsrc ? (int)sh.nchi[sslot] : 0;
    const int ex = wf_excl_scan(sh, par, deg, total);
    int nl = 0;      // elements listed at this level so far (uniform)
    int n0 = v.n;    // elements of the vector (uniform; v.n is written back at the end)
    for (int cbase = 0; cbase < total; cbase += WFK_T) {
        // the batch's candidate keys, in candidate order
#pragma unroll
        for (int q = 0; q < ADJ_DEG; q++) {
            const int c = ex + q - cbase;
            if (q < deg && c >= 0 && c < WFK_T) sh.cand[c] = sh.chi[q][sslot];
        }
        __syncthreads();
        bool valid = cbase + t < total;
        uint32_t key = 0, cell = 0;
        const uint32_t mine = tag | (uint32_t)t;
        if (valid) {
            key = sh.cand[t];
            if (n0 < WFK_CAP) {
                cell = wf_cell(v, key);
            } else {
                // the vector is full: unknown nodes are dropped; they do not go into the hash
                const int pc = wf_probe(v, key);
                valid = pc >= 0;
                cell = valid ? (uint32_t)pc : 0u;
            }
            if (valid) atomicMin(&sh.arb[cell], mine);
        }
        __syncthreads();
        const bool winner = valid && sh.arb[cell] == mine;
        int slot = winner ? (int)v.hslot[cell] : -2;
        const bool is_new = winner && slot == (int)SLOT_NONE;
        const bool old_listed = lvl_list && winner && !is_new && stamp[slot] != level;
        // new elements get consecutive slots in candidate order; the level list takes old and (kept) new first
        // occurrences in candidate order: one scan of both counts
        int tot2;
        const int before = wf_excl_scan(sh, par, (is_new ? 1 : 0) | (old_listed ? 0x10000 : 0), tot2);
        const int new_before = before & 0xffff, old_before = before >> 16;
        const int add = tot2 & 0xffff, old_total = tot2 >> 16;
        const int room = WFK_CAP - n0;
        if (is_new) {
            const int s = n0 + new_before;
            if (s < WFK_CAP) {
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
        if (lvl_list && ((is_new && slot >= 0) || old_listed)) {
            const int p = nl + old_before + (new_before < room ? new_before : room);
            lvl_list[p] = (uint16_t)slot;
            stamp[slot] = level;
        }
        nl += old_total + (add < room ? add : room);
        const int n1 = n0 + add > WFK_CAP ? WFK_CAP : n0 + add;
        __syncthreads();
        // (what the owners of the new slots write -- their children -- is read by the NEXT expansion, behind the
        // barrier that ends this one)
        if (t >= n0 && t < n1) wf_fetch(M, sh, v, rec);
        n0 = n1;
        tag -= 512u;
    }
    if (t == 0) v.n = n0;
    __syncthreads();
    return nl;
}

__global__ void __launch_bounds__(WFK_T) wide_forward_kernel(const SparseFwdArgs a) {
    __shared__ WideFwdShared sh;
    const int t = threadIdx.x;
    const uint32_t gi = a.lanes[blockIdx.x];
    const int g = (int)(gi / a.W), r = (int)(gi % a.W);
    const int len = a.d.len[gi];
    const int s0 = a.sw[gi];
    const size_t NW = (size_t)a.d.N * a.W;
    const uint64_t p0 = a.lane_pos0[gi];
    if (t == 0) sh.lp = a.M.lp;
    const LinParams &lp = sh.lp;
    int par = 0;
    uint32_t tag = 0xfffffe00u;  // batch tag of the arbitration cells (wf_append)
    for (int h = t; h < WFK_HASH; h += WFK_T) sh.arb[h] = 0xffffffffu;
    uint32_t err = 0;
    int pos;        // next position to compute
    int end = len;  // first position this launch does not compute
    int done_to;    // first position NOT done when the kernel leaves
    FwdAdj rec;     // the record of the element in slot t of the column being built
    rec.npar = rec.nchi = 0;
    bool dense_prev = false;
    const double *gm = nullptr, *gi_ = nullptr, *gd = nullptr;
    int E_dense = 0;
    if (a.mode == 0) {
        // ---- first sparse column: top list = candidates of dense column s0-1 sorted by (total desc, node asc)
        pos = s0;
        done_to = s0;
        if (a.max_steps > 0 && s0 + a.max_steps < len) end = s0 + a.max_steps;
        const int nc = a.cand_n[gi];
        WCol &c0 = sh.col[s0 & 1];
        wf_clear(c0);
        const uint32_t *cn = a.cand_node + (size_t)gi * PHMM_MAX_ACTIVE_NODES;
        const double *ct = a.cand_tot + (size_t)gi * PHMM_MAX_ACTIVE_NODES;
        if (t < nc) {
            sh.tot[t] = ct[t];
            sh.cand[t] = cn[t];
        }
        __syncthreads();
        if (t < nc) {
            const double v = sh.tot[t];
            const uint32_t id = sh.cand[t];
            int rank = 0;
            for (int q = 0; q < nc; q++) {
                const double u = sh.tot[q];
                rank += (u > v) || (u == v && sh.cand[q] < id);
            }
            c0.id[rank] = id;
            c0.m[rank] = c0.i[rank] = c0.d[rank] = 0.0;
        }
        __syncthreads();
        if (t < nc) {
            const uint32_t cell = wf_cell(c0, c0.id[t]);
            c0.hslot[cell] = (uint16_t)t;
            wf_fetch(a.M, sh, c0, rec);
        }
        if (t == 0) c0.n = nc;
        __syncthreads();
        dense_prev = true;
        gm = a.d.Fm + ((size_t)g * a.d.Lc + (s0 - 1)) * NW;
        gi_ = a.d.Fi + ((size_t)g * a.d.Lc + (s0 - 1)) * NW;
        gd = a.d.Fd + ((size_t)g * a.d.Lc + (s0 - 1)) * NW;
        E_dense = a.d.FE[((size_t)g * (a.d.Lc + 1) + (s0 - 1)) * a.W + r];
    } else {
        pos = a.stop[gi];
        done_to = pos;
        if (a.max_steps > 0 && pos + a.max_steps < len) end = pos + a.max_steps;
        // resume from the stored column pos-1
        WCol &P = sh.col[(pos - 1) & 1];
        wf_clear(P);
        __syncthreads();
        const uint64_t o1 = a.pool.off[p0 + (uint64_t)(pos - 1)];
        const uint8_t *rc = a.pool.base + (o1 ? o1 - 8 : 0);
        const int *hw = (const int *)rc;
        const int n = o1 ? hw[0] : WFK_CAP + 1, na = o1 ? hw[1] : 0;
        if (n > WFK_CAP) {
            err |= SP_ERR_CAPACITY;
            end = pos;
        } else {
            const uint64_t idb = (uint64_t)((n + 1) & ~1) * 4;
            const uint32_t *ids = (const uint32_t *)(rc + 16);
            const double *rm = (const double *)(rc + 16 + idb), *ri = rm + na, *rd = ri + na;
            if (t < n) {
                const uint32_t id = ids[t];
                P.id[t] = id;
                P.d[t] = rd[t];
                P.m[t] = t < na ? rm[t] : 0.0;
                P.i[t] = t < na ? ri[t] : 0.0;
                const uint32_t cell = wf_cell(P, id);
                P.hslot[cell] = (uint16_t)t;
            }
            if (t == 0) {
                P.n = n;
                P.na = na;
                P.E = hw[2];
            }
        }
        __syncthreads();
    }
    for (; pos < end && !err; pos++) {
        WCol &prev = sh.col[(pos + 1) & 1];
        WCol &cur = sh.col[pos & 1];
        if (!dense_prev) {
            // ---- top_nodes_by_score_ratio of the previous column (table.rs:134-149) as the first elements of cur
            const int n = prev.n;
            wf_clear(cur);
            // The elements inside the ratio are the largest ones: their rank in the stable descending sort of all
            // totals (sort_desc, frontier_dev.h) is their rank among themselves -- only they are sorted, as
            // (total, slot) pairs
            const double mytot = t < n ? prev.m[t] + prev.i[t] + prev.d[t] : 0.0;
            const double t0 = wf_block_max(sh, par, mytot);
            const int mine = (t < n && mytot > 0.0 && mytot > t0 * a.ratio_lin) ? 1 : 0;
            int ntop;
            const int at = wf_excl_scan(sh, par, mine, ntop);
            const int NP = bitonic_size(ntop);
            if (mine) {
                sh.lvb[at] = mytot;
                sh.order[at] = (uint16_t)t;
            }
            for (int k = t; k < NP; k += WFK_T)
                if (k >= ntop) {
                    sh.lvb[k] = -1.0;
                    sh.order[k] = (uint16_t)k;
                }
            block_bitonic_desc(sh.lvb, sh.order, NP);
            if (t < ntop) {
                const uint32_t id = prev.id[sh.order[t]];
                cur.id[t] = id;
                cur.m[t] = cur.i[t] = cur.d[t] = 0.0;
                const uint32_t cell = wf_cell(cur, id);
                cur.hslot[cell] = (uint16_t)t;
                wf_fetch(a.M, sh, cur, rec);
            }
            if (t == 0) cur.n = ntop;
            __syncthreads();
        }
        // ---- one adaptive forward column (fwd_adaptive_step, frontier_dev.h)
        const uint8_t x = a.bases[((size_t)g * a.Lb + pos) * a.W + r];
        const int ntop = cur.n;
        const int Eprev = dense_prev ? E_dense : prev.E;
        if (t < ntop) sh.order[t] = (uint16_t)t;
        __syncthreads();
        wf_append(a.M, sh, cur, par, tag, rec, sh.order, ntop, nullptr, nullptr, 0);
        const int na = cur.n;
        if (t == 0) cur.na = na;
        // fm (forward.rs:337-359), fi (378-388), fib (541-545)
        const double ibs = exp(a.M.logib[pos - 1] - (double)Eprev * SP_LN2);
        const double c_begin = lp.p_IM * ibs;
        const double ib_cur = lp.p_random * lp.p_II * ibs;
        const double c_del = lp.p_ID * ib_cur;
        auto prev_get = [&](uint32_t node, double &m, double &i, double &d) {
            m = i = d = 0.0;
            if (dense_prev) {
                const size_t ix = (size_t)node * a.W + r;
                m = gm[ix];
                i = gi_[ix];
                d = gd[ix];
            } else {
                const int s = wf_find(prev, node);
                if (s >= 0) {
                    m = prev.m[s];
                    i = prev.i[s];
                    d = prev.d[s];
                }
            }
        };
        if (t < na) {
            double acc = 0.0;
            const double pe = rec.emis == x ? lp.p_match : lp.p_mismatch;
#pragma unroll
            for (int q = 0; q < ADJ_DEG; q++) {
                if (q >= (int)rec.npar || rec.par_w[q] == 0.0) continue;
                double pm, pi, pd;
                prev_get(rec.par[q], pm, pi, pd);
                acc += rec.par_w[q] * (lp.p_MM * pm + lp.p_IM * pi + lp.p_DM * pd);
            }
            double om, oi, od;
            prev_get(cur.id[t], om, oi, od);
            cur.m[t] = pe * (acc + rec.init * c_begin);
            cur.i[t] = lp.p_random * (lp.p_MI * om + lp.p_II * oi + lp.p_DI * od);
        }
        sh.sta[t] = 0xff;
        sh.stb[t] = 0xff;
        if (t < na) sh.order[t] = (uint16_t)t;
        __syncthreads();
        // adaptive fd (forward.rs:423-466): S0 = to_childs(active), S_t = to_childs(S_{t-1})
        const uint16_t *src = sh.order;
        int nsrc = na;
        int pslot[ADJ_DEG];  // slots of this element's parents in the column, once found (elements never move)
#pragma unroll
        for (int q = 0; q < ADJ_DEG; q++) pslot[q] = -1;
        for (int lvl = 0; lvl <= lp.n_max_gaps; lvl++) {
            uint16_t *lst = (lvl & 1) ? sh.lb : sh.la;
            uint8_t *st_cur = (lvl & 1) ? sh.stb : sh.sta;
            const uint8_t *st_prev = (lvl & 1) ? sh.sta : sh.stb;
            double *lv_cur = (lvl & 1) ? sh.lvb : sh.tot;
            const double *lv_prev = (lvl & 1) ? sh.tot : sh.lvb;
            const int nl = wf_append(a.M, sh, cur, par, tag, rec, src, nsrc, lst, st_cur, (uint8_t)lvl);
            // the elements listed at this level are the slots stamped with it: each by its own thread
            if (t < cur.n && st_cur[t] == (uint8_t)lvl) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < ADJ_DEG; q++) {
                    if (q >= (int)rec.npar) continue;
                    const double w = rec.par_w[q];
                    if (w == 0.0) continue;
                    if (pslot[q] < 0) pslot[q] = wf_find(cur, rec.par[q]);
                    const int ps = pslot[q];
                    if (ps < 0) continue;
                    if (lvl == 0) {
                        if (ps < na) acc += w * (lp.p_MD * cur.m[ps] + lp.p_ID * cur.i[ps]);  // fd0, forward.rs:480-501
                    } else if (st_prev[ps] == (uint8_t)(lvl - 1)) {
                        acc += w * lv_prev[ps];  // fdt, forward.rs:510-524
                    }
                }
                const double val = lvl == 0 ? acc + rec.init * c_del : lp.p_DD * acc;
                lv_cur[t] = val;
                cur.d[t] += val;
            }
            __syncthreads();
            src = lst;
            nsrc = nl;
        }
        // rescale so that the column maximum is in [0.5, 1)
        const int n = cur.n;
        const double mx = wf_block_max(sh, par, t < n ? fmax(ib_cur, fmax(fmax(cur.m[t], cur.i[t]), cur.d[t])) : ib_cur);
        const int e = sp_exp_of(mx);
        const double sc = sp_pow2(-e);
        if (t < n) {
            cur.m[t] *= sc;
            cur.i[t] *= sc;
            cur.d[t] *= sc;
        }
        const int Ecur = Eprev + e;
        // store the column (store_record, sparse_dyn.h)
        const uint64_t idb = (uint64_t)((n + 1) & ~1) * 4;
        const uint64_t bytes = (16 + idb + (uint64_t)(2 * na + n) * 8 + 15) & ~15ull;
        if (t == 0) {
            cur.E = Ecur;
            sh.bc = atomicAdd(a.pool.top, (unsigned long long)bytes);
        }
        __syncthreads();
        const uint64_t o = sh.bc;
        if (o + bytes > a.pool.cap) {
            err |= SP_ERR_POOL;
            break;
        }
        uint8_t *rc = a.pool.base + o;
        if (t == 0) {
            ((uint32_t *)rc)[0] = (uint32_t)n;
            ((uint32_t *)rc)[1] = (uint32_t)na;
            ((int *)rc)[2] = Ecur;
            ((uint32_t *)rc)[3] = 0;
            a.pool.off[p0 + (uint64_t)pos] = o + 8;
        }
        if (t < n) {
            uint32_t *ids = (uint32_t *)(rc + 16);
            double *om = (double *)(rc + 16 + idb), *oi = om + na, *od = oi + na;
            ids[t] = cur.id[t];
            od[t] = cur.d[t];
            if (t < na) {
                om[t] = cur.m[t];
                oi[t] = cur.i[t];
            }
        }
        dense_prev = false;
        done_to = pos + 1;
        __syncthreads();
    }
    const bool finished = !err && done_to >= len;
    if (finished && t < 64) {
        // fv_log_end (frontier_dev.h; forward.rs:554-558): the generic kernel's summation order
        const WCol &c = sh.col[(len - 1) & 1];
        double s = 0.0;
        for (int j = t; j < c.na; j += 64) s += c.m[j] + c.i[j] + c.d[j];
        s = wave_sum(s);
        if (t == 0) a.out_logp[gi] = log(lp.p_end * s) + (double)c.E * SP_LN2;
    }
    if (t == 0) {
        a.stop[gi] = done_to;
        a.err[gi] = err;
    }
}

}  // namespace phmm
