// The <= 64-node class of the adaptive sparse forward (phase B of sparse_fwd_kernel.h), one
// wave64 per read, ONE LANE PER NODE, links kept across positions.
//
// Same recursion as fwd_adaptive_step (frontier_dev.h):
//   top     = nodes of the previous column within the score ratio       table.rs:134-149
//   active  = top ++ children(top)                                      active_nodes.rs:23-35
//   m, i    over the active list                                        forward.rs:337-388
//   d       = fd0 + n_max_gaps x fdt over S0 = children(active), S_t = children(S_{t-1})
//                                                                       forward.rs:423-524
// laid out for the common case -- a frontier of 10-30 nodes walking along unitigs:
//   * a node owns a lane for as long as it stays in the frontier; its adjacency record (FwdAdj,
//     96 B) and its previous-column values live in that lane's registers;
//   * every lane knows the LANES of its node's parents and children (pl, cl).  The links are kept
//     from position to position and only touched when a node enters or leaves the frontier: a node
//     that enters announces its id by a lane read and every lane compares it with its own parent
//     and child ids; the new lane finds its own neighbours by reading its record's ids lane by
//     lane against the resident ids; a node that leaves is cut out of the links by its lane bit.
//     A position normally takes in one node (the deepest Del level) and drops one, so this is a
//     few dozen scalar-broadcast compares -- no LDS hash, no barriers, no scans;
//   * values of parents are read by ds_bpermute; selection by ratio needs the set, not the order.
// What the lane order changes: only the order of equal-probability entries inside a stored record
// (the reference's own tie order is unpinned).  The class never drops an insert:
// when previous + current nodes need more than 64 lanes it stops BEFORE storing the column and the
// host continues in the 400-slot class, exactly like the generic <64> kernel.
#pragma once

#include "lds_dma.h"
#include "lean_common.h"
#include "sparse_fwd_kernel.h"

namespace phmm {

// Five lane numbers, one byte each (L2_NONE: no lane).
struct L2Links {
    uint32_t lo, hi;  // slots 0..3, slot 4
};
static constexpr uint32_t L2_NONE = 0xffu;
template <int Q> __device__ __forceinline__ uint32_t l2_byte(const L2Links &k) {
    return Q < 4 ? (k.lo >> (8 * Q)) & 0xffu : k.hi & 0xffu;
}
template <int Q> __device__ __forceinline__ void l2_put(L2Links &k, uint32_t v) {
    if (Q < 4) k.lo = (k.lo & ~(0xffu << (8 * (Q & 3)))) | (v << (8 * (Q & 3)));
    else k.hi = v;
}

struct LeanLane {
    uint32_t id;          // node on this lane (LN_EMPTY: free)
    FwdAdj r;             // its adjacency record
    double pm, pi, pd;    // previous column (scaled), 0 on a free lane and for a node new to the frontier
    double m, i, d;       // current column (0 on a free lane)
    L2Links pl, cl;       // lanes of the parents / children
    unsigned long long pmask;  // the parents' lanes as a bit set
    int nres;             // children that are in the frontier (of r.nchi)
};

static_assert(sizeof(FwdAdj) == 96, "six 16-byte pieces are fetched ahead");

// The record of the node the frontier is expected to take in next is requested a position ahead, straight into
// LDS (no register, no wait until it is read).  On a unitig the node that enters is the first child of the one
// that entered before it.
struct LeanFwdShared {
    alignas(16) uint32_t stage[4][24];  // adjacency records fetched ahead (LF_STAGES slots of one FwdAdj, 96 bytes)
    // the column's record, assembled here and written out 16 bytes per lane (one store instruction up to 1 KB):
    // header 16 + ids 256 + m, i, d 3 x 512
    alignas(16) uint8_t rec[16 + 256 + 3 * 512];
};
static constexpr int LEAN_REC_MAX = 16 + 256 + 3 * 512;

// a wave-uniform double, told to the compiler (kept in scalar registers)
__device__ __forceinline__ double l2_uniform(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// The lanes in `gone` leave the frontier: cut them out of everybody's links.
// (`dmax`, here and below: a wave-uniform bound on the parent and child counts of every node that has been on a lane --
// 2 on most of a diploid DBG -- so that the slot loops skip the slots nobody uses)
__device__ __forceinline__ void l2_evict(LeanLane &L, unsigned long long gone, int dmax) {
    if ((gone >> threadIdx.x) & 1ull) {
        L.id = LN_EMPTY;
        L.pm = L.pi = L.pd = 0.0;
    }
    L.pmask &= ~gone;
    for (unsigned long long gm = gone; gm != 0ull; gm &= gm - 1ull) {
        const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane(__ffsll((long long)gm) - 1);
#define L2_CUT(Q)                                           \
    if (Q < dmax) {                                         \
        if (l2_byte<Q>(L.pl) == g) l2_put<Q>(L.pl, L2_NONE); \
        if (l2_byte<Q>(L.cl) == g) {                         \
            l2_put<Q>(L.cl, L2_NONE);                        \
            L.nres--;                                        \
        }                                                   \
    }
        L2_CUT(0) L2_CUT(1) L2_CUT(2) L2_CUT(3) L2_CUT(4)
#undef L2_CUT
    }
}

// Records fetched ahead (LDS-DMA): LF_STAGES slots; lane j < LF_STAGES of `ids` / `stamp` holds the node of slot j
// and the wave's count of vector-memory operations right after its request (lean_common.h: vector-memory waits).
// One slot would do on a haploid unitig -- the node that enters next is the first child of the one that entered
// last -- but a diploid bubble walks two chains that take turns, and each would evict the other's prediction.
static constexpr int LF_STAGES = 4;
struct L2Ahead {
    uint32_t ids;  // per lane
    int stamp;     // per lane
    int rr;        // next slot to replace (wave-uniform)
    int issued;    // vector-memory operations issued from inline asm so far (wave-uniform)
};

// Node `key` (wave-uniform), a child (slot `sq`) of the node on lane `l`, takes the free lane `f`: record, links
// in both directions.
#ifdef PHMM_LEAN_PROF
#define L2_PROF_PARAMS , long long *pt, long long &pc0
#define L2_PROF_ARGS , pt, pc0
#define L2_PROF_T(k)                      \
    {                                     \
        const long long now_ = clock64(); \
        pt[k] += now_ - pc0;              \
        pc0 = now_;                       \
    }
#else
#define L2_PROF_PARAMS
#define L2_PROF_ARGS
#define L2_PROF_T(k)
#endif
__device__ __forceinline__ void l2_adopt(const SparseModel &M, LeanFwdShared &sh, LeanLane &L, L2Ahead &ah, uint32_t key,
                                         int f, int l, int sq, int &dmax L2_PROF_PARAMS) {
    const int lane = threadIdx.x;
    const bool me = lane == f;
    L2_PROF_T(8)
    // its record: requested ahead, or fetched now
    const unsigned long long hitm = __ballot(lane < LF_STAGES && ah.ids == key);
    int slot;
    if (hitm != 0ull) {
        slot = __builtin_amdgcn_readfirstlane(__ffsll((long long)hitm) - 1);
        vm_wait_upto(ah.issued - __builtin_amdgcn_readlane(ah.stamp, slot));
        if (me) L.r = *(const FwdAdj *)sh.stage[slot];
    } else {
        slot = ah.rr;
        ah.rr = (ah.rr + 1) & (LF_STAGES - 1);
        if (me) L.r = M.fadj[key];
        vm_drain();
        L2_PROF_T(12)
    }
    L2_PROF_T(9)
    // (a free lane holds zeros: pm, pi, pd since it was freed, m, i, d since the step began)
    const int np = __builtin_amdgcn_readlane((int)L.r.npar, f), nc = __builtin_amdgcn_readlane((int)L.r.nchi, f);
    // children of the new node that are on a lane already (a loop of the graph: rare)
    L2Links ncl{0xffffffffu, L2_NONE};
    int nr = 0;
    unsigned long long anyc = 0ull;
#define L2_OWNC(Q)                                                                             \
    if (Q < nc) {                                                                              \
        const uint32_t ck = (uint32_t)__builtin_amdgcn_readlane((int)L.r.chi[Q], f);           \
        const unsigned long long cmk = __ballot(L.id == ck || (me && ck == key));              \
        if (cmk != 0ull) {                                                                     \
            l2_put<Q>(ncl, (uint32_t)(__ffsll((long long)cmk) - 1));                           \
            nr++;                                                                              \
            anyc |= cmk;                                                                       \
        }                                                                                      \
    }
    L2_OWNC(0) L2_OWNC(1) L2_OWNC(2) L2_OWNC(3) L2_OWNC(4)
#undef L2_OWNC
    if (np == 1 && anyc == 0ull) {
        // the common case: its one parent is the node that asked for it, nobody else knows it
        if (me) {
            L.id = key;
            L.pl = L2Links{0xffffff00u | (uint32_t)l, L2_NONE};
            L.cl = ncl;
            L.pmask = 1ull << l;
            L.nres = 0;
        }
        if (lane == l) {
            if (sq == 0) l2_put<0>(L.cl, (uint32_t)f);
            else if (sq == 1) l2_put<1>(L.cl, (uint32_t)f);
            else if (sq == 2) l2_put<2>(L.cl, (uint32_t)f);
            else if (sq == 3) l2_put<3>(L.cl, (uint32_t)f);
            else l2_put<4>(L.cl, (uint32_t)f);
            L.nres++;
        }
    } else {
        const bool live = L.id != LN_EMPTY;
        // everybody's links TO the new node
#define L2_TONEW(Q)                                                       \
    if (Q < dmax || Q < np || Q < nc) {                                   \
        if (live && Q < (int)L.r.npar && L.r.par[Q] == key) {             \
            l2_put<Q>(L.pl, (uint32_t)f);                                 \
            L.pmask |= 1ull << f;                                         \
        }                                                                 \
        if (live && Q < (int)L.r.nchi && L.r.chi[Q] == key) {             \
            l2_put<Q>(L.cl, (uint32_t)f);                                 \
            L.nres++;                                                     \
        }                                                                 \
    }
        L2_TONEW(0) L2_TONEW(1) L2_TONEW(2) L2_TONEW(3) L2_TONEW(4)
#undef L2_TONEW
        if (me) L.id = key;
        // the new node's own parent links among the resident nodes (itself included: a self loop)
        L2Links npl{0xffffffffu, L2_NONE};
        unsigned long long npm = 0ull;
#define L2_OWNP(Q)                                                                             \
    if (Q < np) {                                                                              \
        const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)L.r.par[Q], f);           \
        const unsigned long long pmk = __ballot(L.id == pk);                                   \
        if (pmk != 0ull) {                                                                     \
            const int pp = __ffsll((long long)pmk) - 1;                                        \
            l2_put<Q>(npl, (uint32_t)pp);                                                      \
            npm |= 1ull << pp;                                                                 \
        }                                                                                      \
    }
        L2_OWNP(0) L2_OWNP(1) L2_OWNP(2) L2_OWNP(3) L2_OWNP(4)
#undef L2_OWNP
        if (me) {
            L.pl = npl;
            L.cl = ncl;
            L.pmask = npm;
            L.nres = nr;
        }
    }
    dmax = max(dmax, max(np, nc));
    L2_PROF_T(10)
    // request the record of its first child for one of the next positions, into the slot just used
    if (nc > 0) {
        const uint32_t nxt = (uint32_t)__builtin_amdgcn_readlane((int)L.r.chi[0], f);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // a record just read out of that slot is in registers
        if (lane < 6)
            glds16_sv(&M.fadj[nxt], (uint32_t)lane * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)sh.stage[slot]));
        ah.issued++;
        if (lane == slot) {
            ah.ids = nxt;
            ah.stamp = ah.issued;
        }
    } else if (lane == slot) {
        ah.ids = LN_EMPTY;
    }
    L2_PROF_T(11)
}

// Lanes of `src` with a child that is not in the frontier yet.
__device__ __forceinline__ unsigned long long l2_need(const LeanLane &L, unsigned long long src) {
    return __ballot(((src >> threadIdx.x) & 1ull) && L.id != LN_EMPTY && L.nres < (int)L.r.nchi);
}

// Takes in ONE missing child: the first missing child slot of the lowest lane of `need` goes to the lowest free
// lane (so new nodes take lanes in the order (parent lane, child slot)).  Returns false when no lane is free (the
// caller abandons the column).  One adoption per call, and the caller loops: with the loop in here the compiler
// keeps the lane state in scratch memory.
__device__ __forceinline__ bool l2_take_one(const SparseModel &M, LeanFwdShared &sh, LeanLane &L, L2Ahead &ah,
                                            unsigned long long need, int &dmax L2_PROF_PARAMS) {
    const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)need) - 1);
    uint32_t mykey = L.r.chi[4];
    int mysq = 4;
    if (3 < (int)L.r.nchi && l2_byte<3>(L.cl) == L2_NONE) mykey = L.r.chi[3], mysq = 3;
    if (2 < (int)L.r.nchi && l2_byte<2>(L.cl) == L2_NONE) mykey = L.r.chi[2], mysq = 2;
    if (1 < (int)L.r.nchi && l2_byte<1>(L.cl) == L2_NONE) mykey = L.r.chi[1], mysq = 1;
    if (0 < (int)L.r.nchi && l2_byte<0>(L.cl) == L2_NONE) mykey = L.r.chi[0], mysq = 0;
    const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)mykey, l);
    const int sq = __builtin_amdgcn_readlane(mysq, l);
    const unsigned long long freem = ~__ballot(L.id != LN_EMPTY);
    if (freem == 0ull) return false;
    const int f = __builtin_amdgcn_readfirstlane(__ffsll((long long)freem) - 1);
    l2_adopt(M, sh, L, ah, key, f, l, sq, dmax L2_PROF_ARGS);
    return true;
}

// Lanes that hold a child of a node on a lane of `src` (all children are resident: l2_need(src) == 0).
__device__ __forceinline__ unsigned long long l2_children(const LeanLane &L, unsigned long long src) {
    return __ballot(L.id != LN_EMPTY && (L.pmask & src) != 0ull);
}

// sum over the parents of w[q] * value on the parent's lane (the value is 0 on lanes that are not a source).
// The lanes and weights of the first two parent slots -- all there are on most of a DBG -- are unpacked once per
// position (L2Psum); a parent that is not in the frontier enters with weight 0.
struct L2Psum {
    int a0, a1;     // ds_bpermute addresses of the parents' lanes
    double w0, w1;  // weight, 0 where the slot is empty or its node is not on a lane
};
__device__ __forceinline__ L2Psum l2_psum_prepare(const LeanLane &L) {
    const uint32_t p0 = l2_byte<0>(L.pl), p1 = l2_byte<1>(L.pl);
    L2Psum s;
    s.a0 = p0 == L2_NONE ? 0 : (int)(p0 << 2);
    s.a1 = p1 == L2_NONE ? 0 : (int)(p1 << 2);
    s.w0 = p0 == L2_NONE ? 0.0 : L.r.par_w[0];
    s.w1 = p1 == L2_NONE ? 0.0 : L.r.par_w[1];
    return s;
}
__device__ __forceinline__ double l2_bperm(int addr, double v) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
}
__device__ __forceinline__ double l2_parent_sum(const LeanLane &L, const L2Psum &ps, double v, int dmax) {
    double acc = ps.w0 * l2_bperm(ps.a0, v);
    if (dmax > 1) acc += ps.w1 * l2_bperm(ps.a1, v);
    if (dmax > 2) {
#define L2_PAR(Q)                                                \
    if (Q < dmax) {                                              \
        const uint32_t p = l2_byte<Q>(L.pl);                     \
        const double u = __shfl(v, p == L2_NONE ? 0 : (int)p);   \
        if (p != L2_NONE) acc += L.r.par_w[Q] * u;               \
    }
        L2_PAR(2) L2_PAR(3) L2_PAR(4)
#undef L2_PAR
    }
    return acc;
}

__global__ void __launch_bounds__(64, 4) lean_forward_kernel(const SparseFwdArgs a) {
    __shared__ LeanFwdShared sh;
    const int lane = threadIdx.x;
    const uint32_t gi = a.lanes[blockIdx.x];
    const int g = (int)(gi / a.W), r = (int)(gi % a.W);
    const int len = __builtin_amdgcn_readfirstlane(a.d.len[gi]);
    const uint64_t p0 = a.lane_pos0[gi];
    const LinParams &lp = a.M.lp;
    uint32_t err = 0;
    int pos = __builtin_amdgcn_readfirstlane(a.stop[gi]);  // first position to compute
    int done_to = pos;
    int end = len;
    if (a.max_steps > 0 && pos + a.max_steps < len) end = pos + a.max_steps;

    LeanLane L;
    L.id = LN_EMPTY;
    L.pm = L.pi = L.pd = L.m = L.i = L.d = 0.0;
    L.pl = L.cl = L2Links{0xffffffffu, L2_NONE};
    L.pmask = 0ull;
    L.nres = 0;
    L2Ahead ah{LN_EMPTY, 0, 0, 0};  // records requested into sh.stage
    int dmax = 1;               // bound on the degrees seen so far (wave-uniform)
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
            n = __builtin_amdgcn_readfirstlane(hw[0]);
            na = __builtin_amdgcn_readfirstlane(hw[1]);
            E = __builtin_amdgcn_readfirstlane(hw[2]);
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
            // links among the nodes of the column, once: every node announces itself
            for (int j = 0; j < n; j++) {
                const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)L.id, j);
#define L2_INIT(Q)                                                          \
    if (lane < n && Q < (int)L.r.npar && L.r.par[Q] == key) {               \
        l2_put<Q>(L.pl, (uint32_t)j);                                       \
        L.pmask |= 1ull << j;                                               \
    }                                                                       \
    if (lane < n && Q < (int)L.r.nchi && L.r.chi[Q] == key) {               \
        l2_put<Q>(L.cl, (uint32_t)j);                                       \
        L.nres++;                                                           \
    }
                L2_INIT(0) L2_INIT(1) L2_INIT(2) L2_INIT(3) L2_INIT(4)
#undef L2_INIT
            }
            const int deg = lane < n ? max((int)L.r.npar, (int)L.r.nchi) : 0;
            for (int q = 1; q < ADJ_DEG; q++)
                if (__ballot(deg > q) != 0ull) dmax = q + 1;
        }
    }
    // record pool slab of this wave (wave-uniform)
    unsigned long long slab = 0ull, slab_end = 0ull;
    // InsBegin of the previous column in that column's scale (fib, forward.rs:541-545); afterwards it is
    // carried along with the exact power-of-two rescales
    double ibs = l2_uniform((!err && pos < end) ? exp(a.M.logib[pos - 1] - (double)E * SP_LN2) : 0.0);
    // The read's bases, 64 positions per load: lane j holds the base of position xb0 + j.  (One load per position
    // would be one more wait per position, and every wait on a load is a wait on the stores before it.)
    int xb = 0, xb0 = pos;
    auto load_bases = [&](int from) {
        const int p = from + lane;
        const int b = p < a.Lb ? (int)a.bases[((size_t)g * a.Lb + p) * a.W + r] : 0;
        xb = vm_settle(b);
        xb0 = from;
    };
    if (!err && pos < end) load_bases(pos);
    vm_drain();  // everything the prologue loaded has arrived: no compiler-placed vmcnt wait inside the position loop
    // record offsets of the positions done, lane j: position (ob0 + j); written 64 at a time
    unsigned long long offv = 0ull, offm = 0ull;
    int ob0 = pos & ~63;
    auto flush_offsets = [&]() {
        if ((offm >> lane) & 1ull) vm_store8(&a.pool.off[p0 + (uint64_t)(ob0 + lane)], offv);
        ah.issued++;
        offm = 0ull;
    };

#ifdef PHMM_LEAN_PROF
    long long pt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pc0 = 0;
    int psteps = 0;
#define PROF_T(k)                        \
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
        if (pos - xb0 >= 64) load_bases(pos);
        const uint8_t x = (uint8_t)__builtin_amdgcn_readlane(xb, pos - xb0);
        // ---- top = previous nodes within the ratio of the best total (table.rs:134-149)
        const double t = L.id != LN_EMPTY ? L.pm + L.pi + L.pd : 0.0;
        const double tmax = l2_uniform(wave_max_pos(t));
        const unsigned long long top = __ballot(t > 0.0 && t > tmax * a.ratio_lin);
        PROF_T(0)
        // ---- the expansions of the step: h = 0 gives active = top ++ children(top) and m, i; h = 1 .. n_max_gaps + 1
        // the adaptive fd levels S_0 = children(active), S_t = children(S_{t-1}) (forward.rs:423-524)
        const double c_begin = l2_uniform(lp.p_IM * ibs);                 // p_MM*mb' + p_IM*ib' with mb' = 0
        const double ib_cur = l2_uniform(lp.p_random * lp.p_II * ibs);    // fib
        const double c_del = l2_uniform(lp.p_ID * ib_cur);                // fd0 from_begin with mb = 0
        // every child of the lanes of `src` gets a lane (one per turn: l2_take_one); false when no lane is left
        bool overflow = false;
#define L2_BRING_IN(src)                                                     \
    for (;;) {                                                               \
        const unsigned long long need_ = l2_need(L, (src));                  \
        if (need_ == 0ull) break;                                            \
        if (!l2_take_one(a.M, sh, L, ah, need_, dmax L2_PROF_ARGS)) {        \
            overflow = true;                                                 \
            break;                                                           \
        }                                                                    \
    }
        // ---- active = top ++ children(top); m, i (forward.rs:337-388)
        L2_BRING_IN(top)
        PROF_T(1)
        if (overflow) {
            err |= SP_ERR_CAPACITY;
            break;
        }
        act = top | l2_children(L, top);
        const bool is_act = (act >> lane) & 1ull;
        {
            const L2Psum ps = l2_psum_prepare(L);
            const double G = lp.p_MM * L.pm + lp.p_IM * L.pi + lp.p_DM * L.pd;
            const double acc = l2_parent_sum(L, ps, G, dmax);
            L.m = L.i = L.d = 0.0;
            if (is_act) {
                const double pe = L.r.emis == x ? lp.p_match : lp.p_mismatch;
                L.m = pe * (acc + L.r.init * c_begin);
                L.i = lp.p_random * (lp.p_MI * L.pm + lp.p_II * L.pi + lp.p_DI * L.pd);
            }
        }
        {
            // The previous column's values are not needed any more.  When lanes are short, the previous-only
            // nodes leave now instead of at the end of the step (one that comes back as a Del-level node is
            // fetched again), so that a wide frontier still fits the 64 lanes.
            const unsigned long long resident = __ballot(L.id != LN_EMPTY);
            if (64 - __popcll(resident) < 16 && (resident & ~act) != 0ull) l2_evict(L, resident & ~act, dmax);
        }
        PROF_T(2)
        // ---- the adaptive fd levels S_0 = children(active), S_t = children(S_{t-1}) (forward.rs:423-524): a level
        // takes in the children it is missing, then every lane of the level sums its parents' level values
        unsigned long long members = act, srcm = act;
        double lv = lp.p_MD * L.m + lp.p_ID * L.i;  // level value handed to the next level (0 outside the level's set)
        for (int h = 1; h <= lp.n_max_gaps + 1; h++) {
            L2_BRING_IN(srcm)
            PROF_T(3)
            if (overflow) break;
            const unsigned long long S = l2_children(L, srcm);
            if (S == 0ull) break;
            const bool inS = (S >> lane) & 1ull;
            const L2Psum ps = l2_psum_prepare(L);
            const double s = l2_parent_sum(L, ps, lv, dmax);
            const double val = h == 1 ? s + L.r.init * c_del : lp.p_DD * s;
            if (inS) L.d += val;
            lv = inS ? val : 0.0;
            srcm = S;
            members |= S;
            PROF_T(4)
        }
#undef L2_BRING_IN
        if (overflow) {
            err |= SP_ERR_CAPACITY;
            break;
        }
        // ---- rescale so that the column maximum is in [0.5, 1)
        const bool member = (members >> lane) & 1ull;
        int e;
        {
            // (only the exponent of the maximum is needed: it sits in the high word)
            uint32_t hw = member ? max(max((uint32_t)__double2hiint(L.m), (uint32_t)__double2hiint(L.i)), (uint32_t)__double2hiint(L.d)) : 0u;
            hw = max(hw, (uint32_t)__double2hiint(ib_cur));
            if (!wave_exp_of_max_hi(hw, e)) {
                double mx = member ? fmax(fmax(L.m, L.i), L.d) : 0.0;
                mx = wave_max(fmax(mx, ib_cur));
                e = sp_exp_of(mx);
            }
        }
        const double sc = l2_uniform(sp_pow2(-e));
        L.m *= sc;
        L.i *= sc;
        L.d *= sc;
        E += e;
        ibs = l2_uniform(ib_cur * sc);
        PROF_T(5)
        // ---- store the column: active entries first, then the Del-only ones.  The record is assembled in LDS and
        // leaves as 16 bytes per lane: one store instruction (two beyond 1 KB), counted in vm_after
        {
            const int na = __popcll(act);
            const int n = __popcll(members);
            const uint32_t idb = (uint32_t)((n + 1) & ~1) * 4;
            const uint32_t raw = 16 + idb + (uint32_t)(2 * na + n) * 8;
            const uint32_t bytes = (raw + 15) & ~15u;
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
            if ((pos & 63) == 0 && offm != 0ull) flush_offsets();
            if (offm == 0ull) ob0 = pos & ~63;
            if (lane == (pos & 63)) offv = slab + 8;
            offm |= 1ull << (pos & 63);
            slab += bytes;
            if (lane == 0) {
                *(u32x4 *)sh.rec = u32x4{(uint32_t)n, (uint32_t)na, (uint32_t)E, 0u};
                if (n & 1) *(uint32_t *)(sh.rec + 16 + 4 * n) = 0u;    // the pads are zero, not leftovers
                if (raw != bytes) *(unsigned long long *)(sh.rec + raw) = 0ull;
            }
            if (member) {
                const unsigned long long below = (1ull << lane) - 1ull;
                const bool isa = (act >> lane) & 1ull;
                const int slot = isa ? __popcll(act & below) : na + __popcll(members & ~act & below);
                uint8_t *vals = sh.rec + 16 + idb;
                *(uint32_t *)(sh.rec + 16 + 4 * slot) = L.id;
                *(double *)(vals + (size_t)(2 * na + slot) * 8) = L.d;
                if (isa) {
                    *(double *)(vals + (size_t)slot * 8) = L.m;
                    *(double *)(vals + (size_t)(na + slot) * 8) = L.i;
                }
            }
            wave_sync();
            const int nchunk = (int)(bytes >> 4);
            if (lane < nchunk) vm_store16(rec + lane * 16, *(const u32x4 *)(sh.rec + lane * 16));
            ah.issued++;
            if (nchunk > 64) {
                if (lane + 64 < nchunk) vm_store16(rec + (lane + 64) * 16, *(const u32x4 *)(sh.rec + (lane + 64) * 16));
                ah.issued++;
            }
            wave_sync();  // (the next column is assembled over this one)
        }
        PROF_T(6)
        // ---- the column becomes the previous one; nodes that left the frontier free their lanes
        {
            const unsigned long long gone = __ballot(L.id != LN_EMPTY) & ~members;
            if (gone != 0ull) l2_evict(L, gone, dmax);
        }
        if (member) {
            L.pm = L.m;
            L.pi = L.i;
            L.pd = L.d;
        }
        done_to = pos + 1;
        PROF_T(7)
    }
#ifdef PHMM_LEAN_PROF
    if (blockIdx.x == 0 && lane == 0 && psteps > 0)
        printf("lean_fwd prof: steps %d | top %lld take-in(active) %lld fm %lld take-in(levels) %lld levels %lld rescale %lld store %lld evict %lld (cycles/step)\n", psteps,
               pt[0] / psteps, pt[1] / psteps, pt[2] / psteps, pt[3] / psteps, pt[4] / psteps, pt[5] / psteps, pt[6] / psteps, pt[7] / psteps);
    if (blockIdx.x == 0 && lane == 0 && psteps > 0)
        printf("lean_fwd prof adopt: select %lld record %lld (miss %lld) links %lld request %lld\n", pt[8] / psteps, pt[9] / psteps, pt[12] / psteps, pt[10] / psteps, pt[11] / psteps);
#endif
    if (offm != 0ull) flush_offsets();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no fetch-ahead may outlive the wave's use of LDS
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
