// The <= 64-node class of the adaptive sparse forward (phase B of sparse_fwd_kernel.h), one
// wave64 per read, ONE LANE PER NODE.
//
// Same recursion as fwd_adaptive_step (frontier_dev.h):
//   top     = nodes of the previous column within the score ratio       table.rs:134-149
//   active  = top ++ children(top)                                      active_nodes.rs:23-35
//   m, i    over the active list                                        forward.rs:337-388
//   d       = fd0 + n_max_gaps x fdt over S0 = children(active), S_t = children(S_{t-1})
//                                                                       forward.rs:423-524
// but laid out for the common case -- a frontier of 10-30 nodes walking along unitigs:
//   * a node owns a lane for as long as it stays in the frontier; its adjacency record (FwdAdj,
//     96 B) and its previous-column values live in that lane's registers, so a step only touches
//     memory for nodes that are new to the frontier (normally the deepest Del level: one round
//     trip per step instead of ~25 dependent ones);
//   * parents are reached by lane index (an LDS hash node -> lane rebuilt per step, resolved once
//     per step) and their values by ds_bpermute, no loops over vector entries, no sorting:
//     selection by ratio needs the set, not the order.
// What the lane order changes: only the order of equal-probability entries inside a stored record
// (the reference's own tie order is unpinned).  The class never drops an insert:
// when previous + current nodes need more than 64 lanes it stops BEFORE storing the column and the
// host continues in the 400-slot class, exactly like the generic <64> kernel.
#pragma once

#include "lean_common.h"
#include "sparse_fwd_kernel.h"

namespace phmm {

struct LeanLane {
    uint32_t id;          // node on this lane (LN_EMPTY: free)
    FwdAdj r;             // its adjacency record
    double pm, pi, pd;    // previous column (scaled), 0 for a node new to the frontier
    double m, i, d;       // current column
    int pl[ADJ_DEG];      // lanes of the parents (-1: not in the frontier)
    bool miss;            // some child is not in the frontier
};

// Children of the nodes on the lanes of `src`: lanes that hold them are returned as a mask; children
// that are not in the frontier yet take free lanes (their records are fetched).  Returns false when
// the free lanes do not suffice (nothing is modified in that case except hash cells of the keys that
// could not be placed -- the caller abandons the column).
__device__ __forceinline__ void ln_links(const LeanShared &sh, LeanLane &L) {
    // parents and children in one batch of lookups
    uint32_t key[2 * ADJ_DEG];
    bool valid[2 * ADJ_DEG];
    int res[2 * ADJ_DEG];
    const bool live = L.id != LN_EMPTY;
#pragma unroll
    for (int q = 0; q < ADJ_DEG; q++) {
        key[q] = L.r.par[q];
        valid[q] = live && q < (int)L.r.npar;
        key[ADJ_DEG + q] = L.r.chi[q];
        valid[ADJ_DEG + q] = live && q < (int)L.r.nchi;
    }
    ln_find_many<2 * ADJ_DEG>(sh, key, valid, res);
    bool miss = false;
#pragma unroll
    for (int q = 0; q < ADJ_DEG; q++) {
        L.pl[q] = res[q];
        miss |= valid[ADJ_DEG + q] && res[ADJ_DEG + q] < 0;
    }
    L.miss = miss;
}

// lanes that hold a child of a lane of `src` = resident nodes with a parent on a lane of `src`
__device__ __forceinline__ unsigned long long ln_children_of(const LeanLane &L, unsigned long long src) {
    bool hit = false;
#pragma unroll
    for (int q = 0; q < ADJ_DEG; q++) hit |= L.pl[q] >= 0 && ((src >> L.pl[q]) & 1ull);
    return __ballot(hit);
}

// The insertion half of an expansion, kept out of line (inlined next to the fast path it doubles the
// kernel's registers): children of the `insrc` lanes that are not in the frontier yet are queued and the
// free lanes take them in order.  Returns the number of new nodes (-1: the free lanes do not suffice; only
// hash cells of keys that could not be placed are modified then) and, per lane, the node it has to adopt.
struct LnAdopt {
    int total;
    uint32_t key;  // LN_EMPTY: this lane adopts nothing
};
__device__ __noinline__ LnAdopt ln_insert_children(LeanShared &sh, bool insrc, bool occupied, uint32_t c0, uint32_t c1,
                                                   uint32_t c2, uint32_t c3, uint32_t c4, int nchi) {
    const int lane = threadIdx.x;
    const uint32_t chi[ADJ_DEG] = {c0, c1, c2, c3, c4};
    uint16_t hc[ADJ_DEG];
    uint32_t wins = 0;
#pragma unroll
    for (int q = 0; q < ADJ_DEG; q++) {
        hc[q] = 0;
        if (insrc && q < nchi) {
            const uint32_t key = chi[q];
            uint32_t h = ln_hash(key);
            for (;;) {
                const uint32_t old = atomicCAS(&sh.ent[h].x, LN_EMPTY, key);
                if (old == LN_EMPTY) {
                    wins |= 1u << q;
                    break;
                }
                if (old == key) break;
                h = (h + 1) & (LN_HASH - 1);
            }
            hc[q] = (uint16_t)h;
        }
    }
    // winners of new keys queue them; free lanes pick them up in order
    const int nw = __popc(wins);
    const int incl = wave_iscan(nw);
    const int total = __shfl(incl, 63);
    const unsigned long long freemask = ~__ballot(occupied);
    LnAdopt r{total, LN_EMPTY};
    if (total > __popcll(freemask)) {
        r.total = -1;
        return r;
    }
    if (total > 0) {
        int w = incl - nw;
#pragma unroll
        for (int q = 0; q < ADJ_DEG; q++)
            if (wins & (1u << q)) {
                sh.winkey[w] = chi[q];
                sh.winh[w] = hc[q];
                w++;
            }
        ln_sync();
        const bool isfree = (freemask >> lane) & 1ull;
        const int frank = __popcll(freemask & ((1ull << lane) - 1ull));
        if (isfree && frank < total) {
            r.key = sh.winkey[frank];
            sh.ent[sh.winh[frank]].y = (uint32_t)lane;
        }
    }
    ln_sync();
    return r;
}

// Children of the nodes on the lanes of `src`: lanes that hold them are returned as a mask; children
// that are not in the frontier yet take free lanes (their records are fetched).  Returns false when
// the free lanes do not suffice (the caller abandons the column).
__device__ __forceinline__ bool ln_expand(const SparseModel &M, LeanShared &sh, LeanLane &L, unsigned long long src,
                                           unsigned long long &out, int &inserted) {
    const int lane = threadIdx.x;
    const bool insrc = (src >> lane) & 1ull;
    inserted = 0;
    // the common case on a unitig: every child is in the frontier already (links resolved once per step)
    if (__ballot(insrc && L.miss) != 0ull) {
        const LnAdopt ad = ln_insert_children(sh, insrc, L.id != LN_EMPTY, L.r.chi[0], L.r.chi[1], L.r.chi[2], L.r.chi[3],
                                              L.r.chi[4], (int)L.r.nchi);
        if (ad.total < 0) return false;
        inserted = ad.total;
        if (ad.key != LN_EMPTY) {
            L.id = ad.key;
            L.r = M.fadj[ad.key];
            L.pm = L.pi = L.pd = 0.0;
            L.m = L.i = L.d = 0.0;
        }
        ln_links(sh, L);  // new nodes: their own links and everybody's links to them
    }
    out = ln_children_of(L, src);
    return true;
}

__global__ void __launch_bounds__(64) lean_forward_kernel(const SparseFwdArgs a) {
    __shared__ LeanShared sh;
    const int lane = threadIdx.x;
    const uint32_t gi = a.lanes[blockIdx.x];
    const int g = (int)(gi / a.W), r = (int)(gi % a.W);
    const int len = a.d.len[gi];
    const uint64_t p0 = a.lane_pos0[gi];
    const LinParams &lp = a.M.lp;
    uint32_t err = 0;
    int pos = a.stop[gi];  // first position to compute
    int done_to = pos;
    int end = len;
    if (a.max_steps > 0 && pos + a.max_steps < len) end = pos + a.max_steps;

    LeanLane L;
    L.id = LN_EMPTY;
    L.pm = L.pi = L.pd = L.m = L.i = L.d = 0.0;
    int E = 0;
    unsigned long long act = 0ull;  // lanes of the active list of the last finished column
    {
        // resume from the stored column pos-1
        const uint64_t o1 = a.pool.off[p0 + (uint64_t)(pos - 1)];
        int n = 0, na = 0;
        const uint8_t *rec = nullptr;
        if (o1 == 0) err |= SP_ERR_CAPACITY;
        else {
            rec = a.pool.base + (o1 - 8);
            const int *hw = (const int *)rec;
            n = hw[0];
            na = hw[1];
            E = hw[2];
            if (n > 64) err |= SP_ERR_CAPACITY;  // does not fit this class
        }
        if (!err) {
            const uint64_t idb = (uint64_t)((n + 1) & ~1) * 4;
            const uint32_t *ids = (const uint32_t *)(rec + 16);
            const double *rm = (const double *)(rec + 16 + idb), *ri = rm + na, *rd = ri + na;
            if (lane < n) {
                L.id = ids[lane];
                L.pd = rd[lane];
                L.pm = lane < na ? rm[lane] : 0.0;
                L.pi = lane < na ? ri[lane] : 0.0;
                L.r = a.M.fadj[L.id];
            }
            act = na >= 64 ? ~0ull : ((1ull << na) - 1ull);
        }
    }
    // record pool slab of this wave (wave-uniform)
    unsigned long long slab = 0ull, slab_end = 0ull;
    // InsBegin of the previous column in that column's scale (fib, forward.rs:541-545); afterwards it is
    // carried along with the exact power-of-two rescales
    double ibs = (!err && pos < end) ? exp(a.M.logib[pos - 1] - (double)E * SP_LN2) : 0.0;
    uint8_t xn = (!err && pos < end) ? a.bases[((size_t)g * a.Lb + pos) * a.W + r] : (uint8_t)0;

#ifdef PHMM_LEAN_PROF
    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc0 = 0;
    int psteps = 0, plev = 0;
#define PROF_T(k)                         \
    {                                     \
        const long long now_ = clock64(); \
        pt[k] += now_ - pc0;              \
        pc0 = now_;                       \
    }
#else
#define PROF_T(k)
#endif
    for (; pos < end && !err; pos++) {
#ifdef PHMM_LEAN_PROF
        pc0 = clock64();
        psteps++;
#endif
        const uint8_t x = xn;
        if (pos + 1 < end) xn = a.bases[((size_t)g * a.Lb + pos + 1) * a.W + r];
        // ---- node -> lane map of the resident nodes (previous column)
        ln_rebuild(sh, L.id);
        ln_links(sh, L);
        // ---- top = previous nodes within the ratio of the best total (table.rs:134-149)
        const double t = L.id != LN_EMPTY ? L.pm + L.pi + L.pd : 0.0;
        const double tmax = wave_max(t);
        const unsigned long long top = __ballot(t > 0.0 && t > tmax * a.ratio_lin);
        PROF_T(0)
        // ---- one loop over the expansions of the step (a single copy of the insertion path keeps the kernel
        // at 4 waves per SIMD): h = 0 gives active = top ++ children(top) and m, i; h = 1 .. n_max_gaps + 1 the
        // adaptive fd levels S_0 = children(active), S_t = children(S_{t-1}) (forward.rs:423-524)
        const double c_begin = lp.p_IM * ibs;                 // p_MM*mb' + p_IM*ib' with mb' = 0
        const double ib_cur = lp.p_random * lp.p_II * ibs;    // fib
        const double c_del = lp.p_ID * ib_cur;                // fd0 from_begin with mb = 0
        unsigned long long members = 0ull, srcm = top;
        double lv = 0.0;  // level value handed to the next level
        bool overflow = false;
        int ins = 0;
        L.m = L.i = L.d = 0.0;
        for (int h = 0; h <= lp.n_max_gaps + 1; h++) {
            unsigned long long S = 0ull;
            if (!ln_expand(a.M, sh, L, srcm, S, ins)) {
                overflow = true;
                break;
            }
#ifdef PHMM_LEAN_PROF
            if (ins > 0) plev++;
#endif
            PROF_T(1)
            if (h == 0) {
                act = top | S;
                const bool is_act = (act >> lane) & 1ull;
                // fm, fi (forward.rs:337-388)
                const double G = lp.p_MM * L.pm + lp.p_IM * L.pi + lp.p_DM * L.pd;
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < ADJ_DEG; q++) {
                    const double v = ln_shfl(G, L.pl[q]);
                    if (is_act && q < (int)L.r.npar && L.pl[q] >= 0) acc += L.r.par_w[q] * v;
                }
                if (is_act) {
                    const double pe = L.r.emis == x ? lp.p_match : lp.p_mismatch;
                    L.m = pe * (acc + L.r.init * c_begin);
                    L.i = lp.p_random * (lp.p_MI * L.pm + lp.p_II * L.pi + lp.p_DI * L.pd);
                }
                // The previous column's values are not needed any more.  When lanes are short, the previous-only
                // nodes leave now instead of at the end of the step (one that comes back as a Del-level node is
                // fetched again), so that a wide frontier still fits the 64 lanes.
                const unsigned long long resident = __ballot(L.id != LN_EMPTY);
                if (64 - __popcll(resident) < 16 && (resident & ~act) != 0ull) {
                    if (!is_act) {
                        L.id = LN_EMPTY;
                        L.pm = L.pi = L.pd = 0.0;
                    }
                    ln_rebuild(sh, L.id);
                    ln_links(sh, L);
                }
                members = act;
                srcm = act;
                lv = lp.p_MD * L.m + lp.p_ID * L.i;
                PROF_T(2)
                continue;
            }
            const bool inS = (S >> lane) & 1ull;
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < ADJ_DEG; q++) {
                const double v = ln_shfl(lv, L.pl[q]);
                if (inS && q < (int)L.r.npar && L.pl[q] >= 0 && ((srcm >> L.pl[q]) & 1ull)) s += L.r.par_w[q] * v;
            }
            const double val = h == 1 ? s + L.r.init * c_del : lp.p_DD * s;
            if (inS) L.d += val;
            lv = inS ? val : 0.0;
            srcm = S;
            members |= S;
            PROF_T(4)
            if (S == 0ull) break;
        }
        if (overflow) {
            err |= SP_ERR_CAPACITY;
            break;
        }
        // ---- rescale so that the column maximum is in [0.5, 1)
        const bool member = (members >> lane) & 1ull;
        double mx = member ? fmax(fmax(L.m, L.i), L.d) : 0.0;
        mx = wave_max(fmax(mx, ib_cur));
        const int e = sp_exp_of(mx);
        const double sc = sp_pow2(-e);
        L.m *= sc;
        L.i *= sc;
        L.d *= sc;
        E += e;
        ibs = ib_cur * sc;
        PROF_T(5)
        // ---- store the column: active entries first, then the Del-only ones
        {
            const int na = __popcll(act);
            const int n = __popcll(members);
            const uint64_t idb = (uint64_t)((n + 1) & ~1) * 4;
            const uint64_t bytes = (16 + idb + (uint64_t)(2 * na + n) * 8 + 15) & ~15ull;
            if (slab + bytes > slab_end) {
                unsigned long long o = 0;
                if (lane == 0) o = atomicAdd(a.pool.top, (unsigned long long)LN_SLAB);
                slab = __shfl(o, 0);
                slab_end = slab + LN_SLAB;
            }
            if (slab_end > a.pool.cap) {
                err |= SP_ERR_POOL;
                break;
            }
            uint8_t *rec = a.pool.base + slab;
            if (lane == 0) {
                ((uint32_t *)rec)[0] = (uint32_t)n;
                ((uint32_t *)rec)[1] = (uint32_t)na;
                ((int *)rec)[2] = E;
                ((uint32_t *)rec)[3] = 0;
                a.pool.off[p0 + (uint64_t)pos] = slab + 8;
            }
            slab += bytes;
            if (member) {
                const unsigned long long below = (1ull << lane) - 1ull;
                const bool isa = (act >> lane) & 1ull;
                const int slot = isa ? __popcll(act & below) : na + __popcll(members & ~act & below);
                uint32_t *ids = (uint32_t *)(rec + 16);
                double *om = (double *)(rec + 16 + idb), *oi = om + na, *od = oi + na;
                ids[slot] = L.id;
                od[slot] = L.d;
                if (isa) {
                    om[slot] = L.m;
                    oi[slot] = L.i;
                }
            }
        }
        // ---- the column becomes the previous one; nodes that left the frontier free their lanes
        if (member) {
            L.pm = L.m;
            L.pi = L.i;
            L.pd = L.d;
        } else {
            L.id = LN_EMPTY;
            L.pm = L.pi = L.pd = 0.0;
        }
        done_to = pos + 1;
        PROF_T(6)
    }
#ifdef PHMM_LEAN_PROF
    if (blockIdx.x == 0 && lane == 0 && psteps > 0)
        printf("lean_fwd prof: steps %d insertions %d | rebuild+top %lld expand %lld links+fm %lld compact %lld del %lld rescale %lld store %lld (cycles/step)\n",
               psteps, plev, pt[0] / psteps, pt[1] / psteps, pt[2] / psteps, pt[3] / psteps, pt[4] / psteps, pt[5] / psteps,
               pt[6] / psteps);
#endif
    for (int off = 32; off >= 1; off >>= 1) err |= (uint32_t)__shfl_xor((int)err, off);
    const bool finished = !err && done_to >= len;
    double lpv = NAN;
    if (finished) {
        // fe (forward.rs:554-558): the active list of the last column
        double s = ((act >> lane) & 1ull) ? L.pm + L.pi + L.pd : 0.0;
        s = wave_sum(s);
        lpv = log(lp.p_end * s) + (double)E * SP_LN2;
    }
    if (lane == 0) {
        if (finished) a.out_logp[gi] = lpv;
        a.stop[gi] = done_to;
        a.err[gi] = err;
    }
}

}  // namespace phmm
