// The <= 64-node class of backward_by_forward + mapping extraction (the <64> phase of
// mapping_flow.hip), one wave64 per read, ONE LANE PER NODE -- the backward twin of
// lean_fwd_kernel.h.
//
//   B.tables[pos] over the `na` best nodes of F.tables[pos-1]         backward.rs:122-129, table.rs:117-123
//   b_step restricted to that list: bd0 + n_max_gaps x bdt, bm, bi    backward.rs:216-261, 299-483
//   S[pos-1] = F.tables[pos-1] (.) B.tables[pos] / P                  table.rs:320-345, 500-505
//   to_mapping_by_score_ratio                                          hint.rs:135-142, table.rs:134-149
//
// A node keeps its lane while it stays in the B column; its record (BwdAdj: children, their weights
// and emissions) and its previous-column values live in that lane's registers.  The entries of the
// forward record of the position are routed to the lanes of their nodes through LDS; the forward
// records are prefetched two positions ahead (offset -> header -> arrays), so a step waits for memory
// only when a node is new to the column (one record fetch).
// Stops, like the generic <64> kernel, at the first position whose nodes do not fit 64 lanes and parks
// the column in the read's hand-off slot for the 400-slot kernel.
#pragma once

#include "lean_common.h"
#include "sparse_dyn.h"

namespace phmm {

struct LeanBwdShared {
    LeanShared h;
    // routing of forward-record entries (record slot order) to lanes
    uint32_t eid[64];
    double ef[3][64];
    double etot[64];
    uint8_t tgt[64];      // lane that received entry j
    uint8_t slot_of[64];  // per lane: record slot of its entry (0xff: none)
};

__device__ __forceinline__ void lb_park(const SparseBwdArgs &a, uint32_t gi, bool inprev, uint32_t id, double m, double i, double d,
                                        int E) {
    // B column -> hand-off slot (list form)
    BHandoff &h = a.hand[gi];
    const unsigned long long mask = __ballot(inprev);
    const int n = __popcll(mask);
    if (threadIdx.x == 0) {
        h.n = n;
        h.E = E;
    }
    if (inprev) {
        const int s = __popcll(mask & ((1ull << threadIdx.x) - 1ull));
        h.id[s] = id;
        h.m[s] = m;
        h.i[s] = i;
        h.d[s] = d;
    }
}

__global__ void __launch_bounds__(64, 4) lean_backward_kernel(const SparseBwdArgs a) {
    __shared__ LeanBwdShared sh;
    const int lane = threadIdx.x;
    const uint32_t gi = a.lanes[blockIdx.x];
    const int g = (int)(gi / a.W), r = (int)(gi % a.W);
    const int len = a.d.len[gi];
    const int s0 = a.sw[gi];
    const uint64_t p0 = a.lane_pos0[gi];
    const uint64_t q0 = a.map_pos0[gi];
    const double logP = a.d.logPf[gi];
    const LinParams &lp = a.M.lp;
    const bool ok = logP > -INFINITY;
    const int kP = ok ? (int)rint(-logP / SP_LN2) : 0;
    const double cP = ok ? exp(-logP - (double)kP * SP_LN2) : 0.0;
    uint32_t err = 0;

    // ---- lane state: node, record, B values of the column of position pos+1
    uint32_t id = LN_EMPTY;
    BwdAdj R;
    R.nchi = 0;
    double pm = 0.0, pi = 0.0, pd = 0.0;
    bool inprev = false;
    bool have_col = false;  // the lanes hold a B column (computed here or taken from the hand-off slot)
    int Eprev = 0;
    bool prev_is_init = false;
    int pos;  // next position to compute
    bool stopped = false;
    int stop_at = 0;

    // mapping-pool slab of this wave
    unsigned long long slab = 0ull, slab_end = 0ull;
    auto map_alloc = [&](uint64_t bytes) -> uint8_t * {
        if (slab + bytes > slab_end) {
            unsigned long long o = 0;
            if (lane == 0) o = atomicAdd(a.mpool.top, (unsigned long long)LN_SLAB);
            slab = __shfl(o, 0);
            slab_end = slab + LN_SLAB;
        }
        if (slab_end > a.mpool.cap) return nullptr;
        uint8_t *p = a.mpool.base + slab;
        slab += bytes;
        return p;
    };
    // to_mapping_by_score_ratio of the values on the lanes (has: lane carries an entry; slot: its record
    // slot): kept = val > 0 and within the ratio of the best, sorted descending, equal values by node id
    auto emit = [&](uint64_t pos_index, bool has, uint32_t nid, double val, int slot) -> bool {
        (void)slot;
        const double v = has ? val : 0.0;
        const double p0v = wave_max(v);
        const bool keep = has && v > 0.0 && v > p0v * a.ratio_lin;
        const unsigned long long km = __ballot(keep);
        const int k = __popcll(km);
        // ordered by the value the list holds (the log) and equal logs by node id: the same list whatever lanes the
        // read's nodes sit on and however its reads were grouped (see emit_mapping, mapping_flow.hip)
        const double lv = keep ? log(v) : 0.0;
        int rank = 0;
        unsigned long long mm = km;
        const long long vb = __double_as_longlong(lv);
        while (mm) {
            // (l is wave-uniform: scalar lane reads instead of ds_bpermute round trips)
            const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)mm) - 1);
            mm &= mm - 1ull;
            const int ulo = __builtin_amdgcn_readlane((int)(vb & 0xffffffffll), l);
            const int uhi = __builtin_amdgcn_readlane((int)(vb >> 32), l);
            const double u = __longlong_as_double(((long long)uhi << 32) | (long long)(unsigned int)ulo);
            const uint32_t un = (uint32_t)__builtin_amdgcn_readlane((int)nid, l);
            rank += (u > lv) || (u == lv && un < nid);
        }
        const uint64_t idb = (uint64_t)((k + 1) & ~1) * 4;
        const uint64_t bytes = (8 + idb + (uint64_t)k * 8 + 15) & ~15ull;
        uint8_t *rec = map_alloc(bytes);
        if (!rec) return false;
        if (lane == 0) {
            ((uint32_t *)rec)[0] = (uint32_t)k;
            ((uint32_t *)rec)[1] = 0;
            a.mpool.off[pos_index] = (uint64_t)(rec - a.mpool.base) + 8;
        }
        if (keep) {
            ((uint32_t *)(rec + 8))[rank] = nid;
            ((double *)(rec + 8 + idb))[rank] = lv;
        }
        return true;
    };

    // ---- forward records, prefetched: (header of pos-2, arrays of pos-1) are in flight while pos is computed
    auto rec_ptr = [&](int p) -> const uint8_t * {  // record of position p (nullptr: none)
        if (p < 0) return nullptr;
        const uint64_t o1 = a.fpool.off[p0 + (uint64_t)p];
        return o1 ? a.fpool.base + (o1 - 8) : nullptr;
    };
    struct Hdr {
        const uint8_t *rec;
        int n, na, E;
    };
    auto load_hdr = [&](int p) -> Hdr {
        Hdr h{rec_ptr(p), 0, 0, 0};
        if (h.rec) {
            const int *hw = (const int *)h.rec;
            h.n = hw[0];
            h.na = hw[1];
            h.E = hw[2];
        }
        return h;
    };
    struct Ent {
        uint32_t id;
        double m, i, d;
    };
    auto load_ent = [&](const Hdr &h) -> Ent {
        Ent e{LN_EMPTY, 0.0, 0.0, 0.0};
        if (h.rec && h.n <= 64 && lane < h.n) {
            const uint64_t idb = (uint64_t)((h.n + 1) & ~1) * 4;
            const uint32_t *ids = (const uint32_t *)(h.rec + 16);
            const double *fm = (const double *)(h.rec + 16 + idb), *fi = fm + h.na, *fd = fi + h.na;
            e.id = ids[lane];
            e.d = fd[lane];
            e.m = lane < h.na ? fm[lane] : 0.0;
            e.i = lane < h.na ? fi[lane] : 0.0;
        }
        return e;
    };

    Hdr hcur{}, hnext{};
    Ent ecur{}, enext{};
    if (a.mode == 0) {
        pos = len - 1;
        prev_is_init = true;
        // merged index len: F.tables[len-1] (.) b_init / P   (table.rs:414-434, backward.rs:197-211)
        hcur = load_hdr(len - 1);
        if (!hcur.rec || hcur.n > 64) {
            stopped = true;  // does not fit this class (or missing): nothing done
            stop_at = len;
        } else {
            ecur = load_ent(hcur);
            const double w = ok ? exp((double)hcur.E * SP_LN2 - logP) * lp.p_end : 0.0;
            if (!emit(q0 + (uint64_t)(len - 1), lane < hcur.n, ecur.id, w * (ecur.m + ecur.i + ecur.d), lane)) err |= SP_ERR_POOL;
        }
    } else {
        pos = a.stop[gi];
        if (pos < len - 1) {
            const BHandoff &h = a.hand[gi];
            if (h.n > 64) {
                stopped = true;  // the parked column itself does not fit: leave it to the 400-slot kernel
                stop_at = pos;
            } else {
                if (lane < h.n) {
                    id = h.id[lane];
                    pm = h.m[lane];
                    pi = h.i[lane];
                    pd = h.d[lane];
                    inprev = true;
                    R = a.M.badj[id];
                }
                Eprev = h.E;
                have_col = true;
            }
        } else {
            prev_is_init = true;
        }
    }
    if (!stopped && !err && pos >= s0 + 1) {
        hcur = load_hdr(pos - 1);
        ecur = load_ent(hcur);
        hnext = load_hdr(pos - 2 >= s0 ? pos - 2 : -1);
    }
    uint8_t xn = (!stopped && !err && pos >= s0 + 1) ? a.bases[((size_t)g * a.Lb + pos) * a.W + r] : (uint8_t)0;

#ifdef PHMM_LEAN_PROF
    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc0 = 0;
    int psteps = 0;
#define PROFB_T(k)                        \
    {                                     \
        const long long now_ = clock64(); \
        pt[k] += now_ - pc0;              \
        pc0 = now_;                       \
    }
#else
#define PROFB_T(k)
#endif
    for (; !stopped && pos >= s0 + 1 && !err; pos--) {
#ifdef PHMM_LEAN_PROF
        pc0 = clock64();
        psteps++;
#endif
        const uint8_t x = xn;
        if (pos - 1 >= s0 + 1) xn = a.bases[((size_t)g * a.Lb + pos - 1) * a.W + r];
        // this position's record is (hcur, ecur); start the next one's arrays and the header after it
        enext = load_ent(hnext);
        const Hdr hnn = load_hdr(pos - 3 >= s0 ? pos - 3 : -1);
        if (!hcur.rec || hcur.n > 64) {
            stopped = true;  // the forward record is larger than this class
            stop_at = pos;
            break;
        }
        const int n = hcur.n, na = hcur.na < hcur.n ? hcur.na : hcur.n;
        PROFB_T(0)
        // ---- route the entries to the lanes of their nodes
        ln_rebuild(sh.h, id);
        const bool has_e = lane < n;
        int tl = has_e ? ln_find(sh.h, ecur.id) : -1;
        const bool miss = has_e && tl < 0;
        const unsigned long long missm = __ballot(miss);
        const unsigned long long freem = ~__ballot(id != LN_EMPTY);
        if (__popcll(missm) > __popcll(freem)) {
            stopped = true;  // previous + current nodes need more than 64 lanes
            stop_at = pos;
            break;
        }
        if (miss) {
            // the k-th missing entry takes the k-th free lane
            const int k = __popcll(missm & ((1ull << lane) - 1ull));
            unsigned long long f = freem;
            for (int t = 0; t < k; t++) f &= f - 1ull;
            tl = __ffsll((long long)f) - 1;
        }
        if (has_e) {
            sh.eid[lane] = ecur.id;
            sh.ef[0][lane] = ecur.m;
            sh.ef[1][lane] = ecur.i;
            sh.ef[2][lane] = ecur.d;
            sh.etot[lane] = ecur.m + ecur.i + ecur.d;
            sh.tgt[lane] = (uint8_t)tl;
        }
        sh.slot_of[lane] = 0xff;
        ln_sync();
        if (has_e) sh.slot_of[tl] = (uint8_t)lane;
        ln_sync();
        const int slot = sh.slot_of[lane] == 0xff ? -1 : (int)sh.slot_of[lane];
        PROFB_T(1)
        double fm = 0.0, fi = 0.0, fd = 0.0;
        bool sel = false;  // member of the B list: one of the `na` largest totals (ties: record order)
        if (slot >= 0) {
            fm = sh.ef[0][slot];
            fi = sh.ef[1][slot];
            fd = sh.ef[2][slot];
            if (id == LN_EMPTY) {
                // new to the column: take the node and fetch its record
                id = sh.eid[slot];
                R = a.M.badj[id];
                pm = pi = pd = 0.0;
                inprev = false;
                // make it findable for the child links below
                uint32_t h = ln_hash(id);
                for (;;) {
                    const uint32_t old = atomicCAS(&sh.h.ent[h].x, LN_EMPTY, id);
                    if (old == LN_EMPTY) break;
                    h = (h + 1) & (LN_HASH - 1);
                }
                sh.h.ent[h].y = (uint32_t)lane;
            }
            if (na >= n) sel = true;
            else {
                const double t = sh.etot[slot];
                int rank = 0;
                for (int j0 = 0; j0 < n; j0 += 8) {  // (eight LDS reads in flight; etot has 64 entries)
                    double u[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) u[q] = sh.etot[(j0 + q) & 63];
#pragma unroll
                    for (int q = 0; q < 8; q++) rank += (j0 + q < n) && ((u[q] > t) || (u[q] == t && j0 + q < slot));
                }
                sel = rank < na;
            }
        }
        ln_sync();
        PROFB_T(2)
        // ---- child links.  cmax: a wave-uniform bound on the child counts of the nodes on the lanes (2 on most of a
        // diploid DBG) -- the slot loops below skip the slots nobody uses
        int cmax = 1;
        {
            const int nch = id != LN_EMPTY ? (int)R.nchi : 0;
            for (int q = 1; q < ADJ_DEG; q++)
                if (__ballot(nch > q) != 0ull) cmax = q + 1;
        }
        int cl[ADJ_DEG];
        if (cmax <= 2) {
            const uint32_t ck[2] = {R.chi[0], R.chi[1]};
            const bool cv[2] = {sel && 0 < (int)R.nchi, sel && 1 < (int)R.nchi};
            int c2[2];
            ln_find_many<2>(sh.h, ck, cv, c2);
            cl[0] = c2[0];
            cl[1] = c2[1];
            cl[2] = cl[3] = cl[4] = -1;
        } else {
            uint32_t ck[ADJ_DEG];
            bool cv[ADJ_DEG];
#pragma unroll
            for (int q = 0; q < ADJ_DEG; q++) {
                ck[q] = R.chi[q];
                cv[q] = sel && q < (int)R.nchi;
            }
            ln_find_many<ADJ_DEG>(sh.h, ck, cv, cl);
        }
        const unsigned long long selm = __ballot(sel);
        const unsigned long long prevm = __ballot(inprev);
        PROFB_T(3)
        // ---- bd0 (backward.rs:354-377)
        const double pend = lp.p_end;
        double a1 = 0.0;
#pragma unroll
        for (int q = 0; q < ADJ_DEG; q++) {
            if (q >= cmax) continue;
            const double v = ln_shfl(pm, cl[q]);
            if (sel && q < (int)R.nchi && R.chi_w[q] != 0.0) {
                double mu = 0.0;
                if (prev_is_init) mu = pend;
                else if (cl[q] >= 0 && ((prevm >> cl[q]) & 1ull)) mu = v;
                a1 += R.chi_w[q] * (R.chi_emis[q] == x ? lp.p_match : lp.p_mismatch) * mu;
            }
        }
        const double iv = prev_is_init ? pend : (inprev ? pi : 0.0);
        const double qq = lp.p_random * iv;
        double dsum = sel ? lp.p_DM * a1 + lp.p_DI * qq : 0.0;
        double lv = dsum;
        // ---- bdt (backward.rs:387-404), restricted to the list
        for (int t = 1; t <= lp.n_max_gaps; t++) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < ADJ_DEG; q++) {
                if (q >= cmax) continue;
                const double v = ln_shfl(lv, cl[q]);
                if (sel && q < (int)R.nchi && cl[q] >= 0 && ((selm >> cl[q]) & 1ull)) s += R.chi_w[q] * v;
            }
            s *= lp.p_DD;
            lv = sel ? s : 0.0;
            dsum += lv;
        }
        // ---- bm, bi (backward.rs:423-483)
        double td = 0.0;
#pragma unroll
        for (int q = 0; q < ADJ_DEG; q++) {
            if (q >= cmax) continue;
            const double v = ln_shfl(dsum, cl[q]);
            if (sel && q < (int)R.nchi && cl[q] >= 0 && ((selm >> cl[q]) & 1ull)) td += R.chi_w[q] * v;
        }
        double bm = sel ? lp.p_MM * a1 + lp.p_MD * td + lp.p_MI * qq : 0.0;
        double bi = sel ? lp.p_IM * a1 + lp.p_ID * td + lp.p_II * qq : 0.0;
        double bd = dsum;
        PROFB_T(4)
        // ---- rescale
        double mx = wave_max(fmax(fmax(bm, bi), bd));
        const int e = sp_exp_of(mx);
        const double sc = sp_pow2(-e);
        bm *= sc;
        bi *= sc;
        bd *= sc;
        const int Ecur = (prev_is_init ? 0 : Eprev) + e;
        // ---- S = F.tables[pos-1] (.) B.tables[pos] / P over F's elements, then the mapping of pos-1
        // 2^(E_F + E_B) / P with P = 2^-kP e^-rP split once per read: a power of two per position, no exp
        const int ew = hcur.E + Ecur + kP;
        const double w = !ok ? 0.0 : ((ew > -1000 && ew < 1000) ? cP * sp_pow2(ew) : exp((double)(hcur.E + Ecur) * SP_LN2 - logP));
        const double val = (slot >= 0 && sel) ? w * (fm * bm + fi * bi + fd * bd) : 0.0;
        PROFB_T(5)
        if (!emit(q0 + (uint64_t)(pos - 1), slot >= 0, id, val, slot)) {
            err |= SP_ERR_POOL;
            break;
        }
        // ---- the column becomes the previous one; nodes outside the list free their lanes
        if (sel) {
            pm = bm;
            pi = bi;
            pd = bd;
            inprev = true;
        } else {
            id = LN_EMPTY;
            pm = pi = pd = 0.0;
            inprev = false;
        }
        Eprev = Ecur;
        prev_is_init = false;
        have_col = true;
        hcur = hnext;
        ecur = enext;
        hnext = hnn;
        PROFB_T(6)
    }
#ifdef PHMM_LEAN_PROF
    if (blockIdx.x == 0 && lane == 0 && psteps > 0)
        printf("lean_bwd prof: steps %d | loads %lld rebuild+route %lld select+insert %lld links %lld bd+bm %lld rescale %lld emit %lld (cycles/step)\n",
               psteps, pt[0] / psteps, pt[1] / psteps, pt[2] / psteps, pt[3] / psteps, pt[4] / psteps, pt[5] / psteps,
               pt[6] / psteps);
#endif
    // ---- leave: park the column for the next phase, or hand it to the dense backward kernel
    if (stopped && !err) {
        if (have_col && stop_at < len) lb_park(a, gi, inprev, id, pm, pi, pd, Eprev);
        if (lane == 0) a.stop[gi] = stop_at;
    } else if (!err) {
        if (have_col) {
            // dense column of position s0+1 (zeros elsewhere: the host cleared the B buffers), its exponent
            // and maximum
            const size_t NW = (size_t)a.d.N * a.W;
            const int pc = (s0 + 1) & 1;
            double *bmp = a.d.Bm + ((size_t)g * a.d.bcols + pc) * NW;
            double *bip = a.d.Bi + ((size_t)g * a.d.bcols + pc) * NW;
            double mx = 0.0;
            if (inprev) {
                bmp[(size_t)id * a.W + r] = pm;
                bip[(size_t)id * a.W + r] = pi;
                mx = fmax(pm, pi);
            }
            mx = wave_max(mx);
            if (lane == 0) {
                a.d.cmaxB[((size_t)g * a.d.Lc + (s0 + 1)) * a.W + r] = (unsigned long long)__double_as_longlong(mx);
                a.d.BE[((size_t)g * (a.d.Lc + 1) + (s0 + 1)) * a.W + r] = Eprev;
            }
        }
        if (lane == 0) a.stop[gi] = s0;
    }
    for (int off = 32; off >= 1; off >>= 1) err |= (uint32_t)__shfl_xor((int)err, off);
    if (lane == 0) a.err[gi] = err;
}

}  // namespace phmm
