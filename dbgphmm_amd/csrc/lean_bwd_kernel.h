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
// forward record of the position are routed to the lanes of their nodes through LDS.
//
// Memory side (round 3; lean_common.h "vector-memory waits"): a wave that walks a read alone must never wait on
// the one in-order memory counter with a store in front of the load it needs.  So
//   * the forward records come in by LDS-DMA, the first KB of the record of position pos-3 while pos is computed
//     (a ring of four 1 KB slots; the offsets of 64 positions are fetched by one load);
//   * the record of the node the column takes in next -- on a unitig the first parent of the one it took in last
//     (BwdAdj.par0) -- is requested a position ahead, also into LDS;
//   * the mapping list of a position is assembled in LDS and leaves as ONE 16-byte-per-lane store; its offset
//     is one more store; bases come 64 positions per load;
//   * every one of these operations is issued from inline asm and counted, every wait is the exact vmcnt(N).
// Stops, like the generic <64> kernel, at the first position whose nodes do not fit 64 lanes and parks
// the column in the read's hand-off slot for the 400-slot kernel.
#pragma once

#include "lds_dma.h"
#include "lean_common.h"
#include "sparse_dyn.h"

namespace phmm {

static constexpr int LB_RING = 4;         // forward records in flight / in use
static constexpr int LB_SLOT = 1024;      // bytes of a record fetched ahead (one 16-byte-per-lane request)

struct LeanBwdShared {
    LeanShared h;
    alignas(16) uint8_t ring[LB_RING][LB_SLOT];  // forward records (record layout, sparse_dyn.h)
    alignas(16) uint32_t stage[20];              // BwdAdj fetched ahead (80 bytes)
    alignas(16) uint8_t out[8 + 256 + 512 + 16]; // the mapping record of the position
    double etot[64];      // totals of the record's entries (B-list selection)
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
    const int len = __builtin_amdgcn_readfirstlane(a.d.len[gi]);
    const int s0 = __builtin_amdgcn_readfirstlane(a.sw[gi]);
    const uint64_t p0 = a.lane_pos0[gi];
    const uint64_t q0 = a.map_pos0[gi];
    const double logP = a.d.logPf[gi];
    const LinParams &lp = a.M.lp;
    const bool ok = logP > -INFINITY;
    const int kP = ok ? (int)rint(-logP / SP_LN2) : 0;
    const double cP = ok ? exp(-logP - (double)kP * SP_LN2) : 0.0;
    uint32_t err = 0;
    int vm_issued = 0;  // vector-memory operations issued from inline asm so far (wave-uniform)

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

    // ---- mapping records: slab of this wave
    unsigned long long slab = 0ull, slab_end = 0ull;
    // to_mapping_by_score_ratio of the values on the lanes (has: lane carries an entry): kept = val > 0 and within
    // the ratio of the best, sorted descending, equal values by node id
    auto emit = [&](int p, bool has, uint32_t nid, double val) -> bool {
        const double v = has ? val : 0.0;
        const double p0v = wave_max_pos(v);
        const bool keep = has && v > 0.0 && v > p0v * a.ratio_lin;
        const unsigned long long km = __ballot(keep);
        const int k = __popcll(km);
        // The list is written in lane order with flag 1 in its header; the builder of the CSR (map_compact) puts it
        // in the order a list has -- by the value it holds (the log), equal logs by node id, the same list whatever
        // lanes the read's nodes sit on and however its reads were grouped -- with the whole chip instead of one wave
        const double lv = keep ? log(v) : 0.0;
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
        const uint32_t idb = (uint32_t)((k + 1) & ~1) * 4;
        const uint32_t raw = 8 + idb + (uint32_t)k * 8;
        const uint32_t bytes = (raw + 15) & ~15u;
        if (slab + bytes > slab_end) {
            unsigned long long o = 0;
            if (lane == 0) o = atomicAdd(a.mpool.top, (unsigned long long)LN_SLAB);
            o = __shfl(o, 0);
            slab = (o + 15ull) & ~15ull;  // (other writers of this pool allocate multiples of 8)
            slab_end = o + LN_SLAB;
        }
        if (slab_end > a.mpool.cap) return false;
        uint8_t *rec = a.mpool.base + slab;
        if (lane == 0) vm_store8(&a.mpool.off[q0 + (uint64_t)p], slab + 8);
        vm_issued++;
        slab += bytes;
        if (lane == 0) {
            *(unsigned long long *)sh.out = (unsigned long long)(uint32_t)k | (1ull << 32);
            if (k & 1) *(uint32_t *)(sh.out + 8 + 4 * k) = 0u;
            if (raw != bytes) *(unsigned long long *)(sh.out + raw) = 0ull;
        }
        if (keep) {
            *(uint32_t *)(sh.out + 8 + 4 * rank) = nid;
            *(double *)(sh.out + 8 + idb + 8 * rank) = lv;
        }
        wave_sync();
        if (lane < (int)(bytes >> 4)) vm_store16(rec + lane * 16, *(const u32x4 *)(sh.out + lane * 16));
        vm_issued++;
        wave_sync();
        return true;
    };

    // ---- forward records.  Offsets of 64 positions per load (lane j: position fo_hi - j); the first KB of a record
    // by LDS-DMA into ring slot (position & 3); rqv: vm_issued right after the request of a slot
    unsigned long long fofv = 0ull;
    int fo_hi = -1, fo_lo = 0;  // positions [fo_lo, fo_hi] are on the lanes
    int rqv = 0;        // lane (p & 3): vm_issued right after the request of the record of position p
    int next_req = -1;  // highest position whose record has not been requested
    auto request = [&](int p) {  // p >= s0, wave-uniform
        if (p > fo_hi || p < fo_lo) {
            const int q = p - lane;
            unsigned long long o = q >= s0 ? a.fpool.off[p0 + (uint64_t)q] : 0ull;
            fofv = (unsigned long long)(uint32_t)vm_settle((int)(uint32_t)o) |
                   ((unsigned long long)(uint32_t)vm_settle((int)(uint32_t)(o >> 32)) << 32);
            fo_hi = p;
            fo_lo = p - 63 > s0 ? p - 63 : s0;
        }
        const int j = fo_hi - p;
        const unsigned long long o1 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(fofv >> 32), j) << 32) |
                                      (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)fofv, j);
        if (o1 != 0ull) {
            glds16_sv(a.fpool.base + (o1 - 8), (uint32_t)lane * 16u,
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)sh.ring[p & (LB_RING - 1)]));
            vm_issued++;
        } else if (lane == 0) {
            *(uint32_t *)sh.ring[p & (LB_RING - 1)] = 0xffffffffu;  // no record: reads as "does not fit"
        }
        if (lane == (p & (LB_RING - 1))) rqv = vm_issued;
        next_req = p - 1;
    };
    struct Hdr {
        bool present;
        int n, na, E;
    };
    // the record of position p is in its ring slot (waits for its request)
    auto header = [&](int p) -> Hdr {
        vm_wait_upto(vm_issued - __builtin_amdgcn_readlane(rqv, p & (LB_RING - 1)));
        wave_sync();
        const int *hw = (const int *)sh.ring[p & (LB_RING - 1)];
        Hdr h;
        h.n = __builtin_amdgcn_readfirstlane(hw[0]);
        h.na = __builtin_amdgcn_readfirstlane(hw[1]);
        h.E = __builtin_amdgcn_readfirstlane(hw[2]);
        h.present = h.n != -1;
        return h;
    };
    struct Ent {
        uint32_t id;
        double m, i, d;
    };
    // entry `slot` of the record of position p (header h): from the ring, or from memory where the record is longer
    // than a ring slot (more than ~40 nodes: rare)
    auto entry = [&](int p, const Hdr &h, int slot, bool want) -> Ent {
        Ent e{LN_EMPTY, 0.0, 0.0, 0.0};
        const uint32_t idb = (uint32_t)((h.n + 1) & ~1) * 4;
        const uint32_t om = 16 + idb, oi = om + 8 * (uint32_t)h.na, od = oi + 8 * (uint32_t)h.na;
        const uint32_t bytes = od + 8 * (uint32_t)h.n;
        if (bytes <= (uint32_t)LB_SLOT) {
            if (want) {
                const uint8_t *rec = sh.ring[p & (LB_RING - 1)];
                e.id = *(const uint32_t *)(rec + 16 + 4 * slot);
                e.d = *(const double *)(rec + od + 8 * slot);
                if (slot < h.na) {
                    e.m = *(const double *)(rec + om + 8 * slot);
                    e.i = *(const double *)(rec + oi + 8 * slot);
                }
            }
        } else {
            const unsigned long long o1 = a.fpool.off[p0 + (uint64_t)p];
            const uint8_t *rec = a.fpool.base + (o1 - 8);
            if (want) {
                e.id = *(const uint32_t *)(rec + 16 + 4 * slot);
                e.d = *(const double *)(rec + od + 8 * slot);
                if (slot < h.na) {
                    e.m = *(const double *)(rec + om + 8 * slot);
                    e.i = *(const double *)(rec + oi + 8 * slot);
                }
            }
            vm_drain();
        }
        return e;
    };

    // ---- bases, 64 positions per load (lane j: position xb_hi - j)
    int xb = 0, xb_hi = -1;
    auto load_bases = [&](int from) {
        const int p = from - lane;
        const int b = p >= 0 ? (int)a.bases[((size_t)g * a.Lb + p) * a.W + r] : 0;
        xb = vm_settle(b);
        xb_hi = from;
    };
    // ---- the BwdAdj fetched ahead
    uint32_t ahead = LN_EMPTY;
    int rq_adj = 0;
    auto request_adj = [&](uint32_t node) {
        ahead = node;
        if (node != LN_EMPTY) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (a record just read out of sh.stage is in registers)
            if (lane < 5)
                glds16_sv(&a.M.badj[node], (uint32_t)lane * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)sh.stage));
            vm_issued++;
            rq_adj = vm_issued;
        }
    };

    Hdr hcur{};
    if (a.mode == 0) {
        pos = len - 1;
        prev_is_init = true;
        // merged index len: F.tables[len-1] (.) b_init / P   (table.rs:414-434, backward.rs:197-211)
        request(len - 1);
        if (len - 2 >= s0) request(len - 2);
        if (len - 3 >= s0) request(len - 3);
        hcur = header(len - 1);
        if (!hcur.present || hcur.n > 64) {
            stopped = true;  // does not fit this class (or missing): nothing done
            stop_at = len;
        } else {
            const Ent e = entry(len - 1, hcur, lane, lane < hcur.n);
            const double w = ok ? exp((double)hcur.E * SP_LN2 - logP) * lp.p_end : 0.0;
            if (!emit(len - 1, lane < hcur.n, e.id, w * (e.m + e.i + e.d))) err |= SP_ERR_POOL;
        }
    } else {
        pos = __builtin_amdgcn_readfirstlane(a.stop[gi]);
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
        if (!stopped && pos >= s0 + 1) {
            request(pos - 1);
            if (pos - 2 >= s0) request(pos - 2);
        }
    }
    if (!stopped && !err && pos >= s0 + 1) load_bases(pos);
    vm_drain();  // everything the prologue loaded has arrived: no compiler-placed vmcnt wait inside the position loop

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
    int steps_left = a.max_steps;
    bool sliced = false;
    for (; !stopped && pos >= s0 + 1 && !err; pos--) {
#ifdef PHMM_LEAN_PROF
        pc0 = clock64();
        psteps++;
#endif
        if (a.max_steps > 0 && steps_left-- == 0) {
            stopped = true;  // this launch's share of the read is done: the column goes to the hand-off slot
            sliced = true;
            stop_at = pos;
            break;
        }
        if (xb_hi - pos >= 64) load_bases(pos);
        const uint8_t x = (uint8_t)__builtin_amdgcn_readlane(xb, xb_hi - pos);
        // this position's record is that of pos-1 (requested two positions ago); start the one of pos-3
        if (next_req >= s0 && next_req >= pos - 3) request(next_req);
        hcur = header(pos - 1);
        if (!hcur.present || hcur.n > 64) {
            stopped = true;  // the forward record is larger than this class
            stop_at = pos;
            break;
        }
        const int n = hcur.n, na = hcur.na < hcur.n ? hcur.na : hcur.n;
        const bool has_e = lane < n;
        const Ent ecur = entry(pos - 1, hcur, lane, has_e);
        PROFB_T(0)
        // ---- route the entries to the lanes of their nodes
        ln_rebuild(sh.h, id);
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
            sh.etot[lane] = ecur.m + ecur.i + ecur.d;
            sh.tgt[lane] = (uint8_t)tl;
        }
        sh.slot_of[lane] = 0xff;
        ln_sync();
        if (has_e) sh.slot_of[tl] = (uint8_t)lane;
        ln_sync();
        const int slot = sh.slot_of[lane] == 0xff ? -1 : (int)sh.slot_of[lane];
        PROFB_T(1)
        const Ent mine = entry(pos - 1, hcur, slot < 0 ? 0 : slot, slot >= 0);
        const double fm = mine.m, fi = mine.i, fd = mine.d;
        bool sel = false;  // member of the B list: one of the `na` largest totals (ties: record order)
        // ---- nodes new to the column take their lanes; the record of ONE new node that was foreseen comes out of LDS,
        // anything else from memory (then every memory operation of the wave is waited for)
        {
            const bool isnew = slot >= 0 && id == LN_EMPTY;
            const unsigned long long newm = __ballot(isnew);
            if (newm != 0ull) {
                const int l1 = __builtin_amdgcn_readfirstlane(__ffsll((long long)newm) - 1);
                const uint32_t key1 = (uint32_t)__builtin_amdgcn_readlane((int)mine.id, l1);
                const bool foreseen = (newm & (newm - 1ull)) == 0ull && key1 == ahead;
                if (foreseen) {
                    vm_wait_upto(vm_issued - rq_adj);
                    if (isnew) R = *(const BwdAdj *)sh.stage;
                } else {
                    if (isnew) R = a.M.badj[mine.id];
                    vm_drain();
                }
                if (isnew) {
                    id = mine.id;
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
                // the node expected next: the first parent of the (first) one that came in
                request_adj((uint32_t)__builtin_amdgcn_readlane((int)R.par0, l1));
            }
        }
        if (slot >= 0) {
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
        int e;
        if (!wave_exp_of_max_hi(max(max((uint32_t)__double2hiint(bm), (uint32_t)__double2hiint(bi)), (uint32_t)__double2hiint(bd)), e))
            e = sp_exp_of(wave_max(fmax(fmax(bm, bi), bd)));
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
        if (!emit(pos - 1, slot >= 0, id, val)) {
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
        PROFB_T(6)
    }
#ifdef PHMM_LEAN_PROF
    if (blockIdx.x == 0 && lane == 0 && psteps > 0)
        printf("lean_bwd prof: steps %d | loads %lld rebuild+route %lld select+insert %lld links %lld bd+bm %lld rescale %lld emit %lld (cycles/step)\n",
               psteps, pt[0] / psteps, pt[1] / psteps, pt[2] / psteps, pt[3] / psteps, pt[4] / psteps, pt[5] / psteps,
               pt[6] / psteps);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no fetch-ahead may outlive the wave's use of LDS
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
    if (lane == 0) a.err[gi] = err | ((sliced && !err) ? SP_STOP_SLICE : 0u);
}

}  // namespace phmm
