// Sparse (active-node frontier) kernels: one wave64 per (read[, candidate]).
//
// hinted_score_kernel = PHMMModel::forward_with_mapping_score_only
//   (src/hmmv2/forward.rs:79-89): f_step over mapping.nodes(i) for every read position,
//   non-adaptive, returning table.e of the last position.  It is the inner loop of
//   `infer`: to_full_prob_reads (src/hmmv2/freq.rs:175-192) called once per candidate
//   copy-number vector per iteration (src/multi_dbg/posterior.rs:483-515).  The grid is
//   (reads x candidates): topology, reads and mappings are shared, only init/trans differ.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "sparse_dyn.h"

namespace phmm {

// a rescale exponent below this in ONE position: see hinted_exact_kernel
static constexpr int HINT_COLLAPSE_EXP = -512;

struct HintedArgs {
    SparseModel M;
    const double *init_c;   // [C][N] linear
    const double *trans_c;  // [C][E] linear
    uint32_t E;
    const uint8_t *bases;
    const uint64_t *read_off;
    const uint64_t *map_pos_off;
    const uint32_t *map_nodes;
    const uint32_t *read_ids;  // reads of this capacity class
    uint64_t R;
    double *out_logp;  // [C][R]
    uint32_t *err;     // [C][R]
    RecPool pool;      // optional (forward_with_mapping, forward.rs:51-75): one record per position, candidate 0 only
};

template <int CAP, int LPN>
__global__ void __launch_bounds__(64) hinted_score_kernel(const HintedArgs a) {
    __shared__ Col<CAP> cols[2];
    __shared__ int16_t lnk_slot[CAP * LPN];
    __shared__ double lnk_w[CAP * LPN];
    __shared__ double dA[CAP], dB[CAP];
    const uint32_t rd = a.read_ids[blockIdx.x];
    const uint32_t cand = blockIdx.y;
    SparseModel M = a.M;
    M.init = a.init_c + (size_t)cand * M.N;
    M.trans = a.trans_c + (size_t)cand * a.E;
    const uint64_t b0 = a.read_off[rd];
    const int len = (int)(a.read_off[rd + 1] - b0);
    uint32_t err = 0;
    if (threadIdx.x == 0) {
        cols[0].n = cols[0].na = 0;
        cols[0].E = 0;
        cols[1].n = cols[1].na = 0;
        cols[1].E = 0;
    }
    wave_sync();
    int pos = 0;
    bool collapse = false;
    for (; pos < len; pos++) {
        const uint64_t o0 = a.map_pos_off[b0 + pos], o1 = a.map_pos_off[b0 + pos + 1];
        const int n = (int)(o1 - o0);
        if (n > CAP) {
            err |= SP_ERR_CAPACITY;
            break;
        }
        err |= fwd_list_step<CAP, LPN>(M, cols[(pos + 1) & 1], cols[pos & 1], a.map_nodes + o0, n, a.bases[b0 + pos],
                                       pos == 0, pos, lnk_slot, lnk_w, dA, dB);
        // a column 2^-512 below the one before it: every listed node died and what is left restarts from an InsBegin
        // chain that is down among the denormals (or below) -- not trusted, see hinted_exact_kernel
        if (pos > 0 && cols[pos & 1].E - cols[(pos + 1) & 1].E < HINT_COLLAPSE_EXP) collapse = true;
        if (a.pool.base && cand == 0 && !store_record_col<CAP>(a.pool, b0 + pos, cols[pos & 1])) err |= SP_ERR_POOL;
    }
    for (int off = 32; off >= 1; off >>= 1) err |= (uint32_t)__shfl_xor((int)err, off);
    const double lp = err ? NAN : (collapse ? -INFINITY : col_log_end(M, cols[(len - 1) & 1]));
    if (threadIdx.x == 0) {
        a.out_logp[(size_t)cand * a.R + rd] = lp;
        a.err[(size_t)cand * a.R + rd] = err;
    }
}

// The same recursion for the <= 64-node class on graphs of degree <= 5 (every DBG), ONE LANE PER LIST ENTRY:
//   * a column lives in registers (m, i, d of the entry on its lane); the LDS only holds two small hashes
//     node -> lane (previous and current list).  Parents' values come by ds_bpermute: the previous column enters
//     through G = p_MM m + p_IM i + p_DM d (one shuffle per parent), the Del levels through the level value;
//   * software-pipelined over the positions: list offsets, node ids, their parent records (ParRec) and init
//     values and the base are requested one or two positions ahead, trans[edge] as soon as the record is there.
// The generic kernel (fwd_list_step) keeps the column in LDS and walks offsets -> ids -> CSR -> edges -> trans:
// ~25 dependent LDS and 5 dependent global round trips per position.  Same sums in the same order.
static constexpr int HL_HASH = 128;
struct HintedLeanShared {
    uint2 ent[2][HL_HASH];  // {node, lane} of the list of position parity
};
__device__ __forceinline__ uint32_t hl_hash(uint32_t id) { return (id * 2654435761u) >> 25; }
__device__ __forceinline__ int hl_find(const uint2 *ent, uint32_t id) {
    uint32_t h = hl_hash(id);
    for (;;) {
        const uint2 e = ent[h];
        if (e.x == id) return (int)e.y;
        if (e.x == 0xffffffffu) return -1;
        h = (h + 1) & (HL_HASH - 1);
    }
}
__device__ __forceinline__ double hl_shfl(double v, int src) { return __shfl(v, src < 0 ? 0 : src); }

// The arithmetic of one list entry, written with explicit roundings (no operand order or fma contraction is left to the
// compiler): hinted_lean_kernel and hinted_packed_kernel<WG, CPL> call these and nothing else on values, so a
// candidate's score has the same bits whichever kernel, group or lane slot evaluated it.
struct HStep {
    // G = p_MM m + p_IM i + p_DM d of the previous column (what flows into Match), H likewise into Ins
    static __device__ __forceinline__ double lin3(double a, double x, double b, double y, double c, double z) {
        return __fma_rn(c, z, __fma_rn(b, y, __dmul_rn(a, x)));
    }
    static __device__ __forceinline__ double acc(double w, double v, double s) { return __fma_rn(w, v, s); }
    // fm: e_k(x) (sum + init c_begin)
    static __device__ __forceinline__ double match(double pe, double sum, double init, double c_begin) {
        return __dmul_rn(pe, __fma_rn(init, c_begin, sum));
    }
    // Del level input p_MD m + p_ID i
    static __device__ __forceinline__ double lv(const LinParams &lp, double m, double i) {
        return __fma_rn(lp.p_ID, i, __dmul_rn(lp.p_MD, m));
    }
    static __device__ __forceinline__ double c_begin(const LinParams &lp, bool first, double ibs) {
        return first ? lp.p_MM : __dmul_rn(lp.p_IM, ibs);
    }
    static __device__ __forceinline__ double ib_cur(const LinParams &lp, bool first, double ibs) {
        return first ? __dmul_rn(lp.p_random, lp.p_MI) : __dmul_rn(__dmul_rn(lp.p_random, lp.p_II), ibs);
    }
    static __device__ __forceinline__ double log_end(const LinParams &lp, double stot, int E) {
        return __fma_rn((double)E, SP_LN2, log(__dmul_rn(lp.p_end, stot)));
    }
};

__global__ void __launch_bounds__(64) hinted_lean_kernel(const HintedArgs a) {
    constexpr int CAP = 64;
    __shared__ HintedLeanShared sh;
    const int lane = threadIdx.x;
    const uint32_t rd = a.read_ids[blockIdx.x];
    const uint32_t cand = blockIdx.y;
    const double *init = a.init_c + (size_t)cand * a.M.N;
    const double *trans = a.trans_c + (size_t)cand * a.E;
    const ParRec *prec = a.M.prec;
    const LinParams &lp = a.M.lp;
    const uint64_t b0 = a.read_off[rd];
    const int len = (int)(a.read_off[rd + 1] - b0);
    const uint64_t *po = a.map_pos_off + b0;
    uint32_t err = 0;
    // pipeline: (o, n, id, x) of the current and the next position, records and weights of the current one
    uint64_t o_cur = po[0], o_nx = po[1], o_n2 = po[len >= 2 ? 2 : 1];
    int n_cur = (int)(o_nx - o_cur), n_nx = len >= 2 ? (int)(o_n2 - o_nx) : 0;
    // (inactive lanes read node 0: every load below is unconditional)
    uint32_t id_cur = (lane < n_cur && n_cur <= CAP) ? a.map_nodes[o_cur + lane] : 0u;
    uint32_t id_nx = (lane < n_nx && n_nx <= CAP) ? a.map_nodes[o_nx + lane] : 0u;
    ParRec rc_cur = prec[id_cur];
    double in_cur = init[id_cur];
    double w_cur[ADJ_DEG];
#pragma unroll
    for (int q = 0; q < ADJ_DEG; q++) w_cur[q] = q < (int)rc_cur.npar ? trans[rc_cur.pedge[q]] : 0.0;
    uint8_t x_cur = a.bases[b0], x_nx = len >= 2 ? a.bases[b0 + 1] : (uint8_t)0;
    // previous column on the lanes of ITS list order
    double pm = 0.0, pi = 0.0, pd = 0.0, m = 0.0, ii = 0.0, d = 0.0, ibs = 0.0;
    int Eprev = 0, n_prev = 0;
    bool collapse = false;
#ifdef PHMM_LEAN_PROF
    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc0 = 0;
#define HPROF(k)                          \
    {                                     \
        const long long now_ = clock64(); \
        pt[k] += now_ - pc0;              \
        pc0 = now_;                       \
    }
#else
#define HPROF(k)
#endif
    for (int pos = 0; pos < len; pos++) {
        if (n_cur > CAP) {
            err |= SP_ERR_CAPACITY;
            break;
        }
#ifdef PHMM_LEAN_PROF
        pc0 = clock64();
#endif
        // ---- requests for the positions ahead
        const ParRec rc_nx = prec[id_nx];
        const double in_nx = init[id_nx];
        const uint64_t o_n3 = po[pos + 3 <= len ? pos + 3 : len];
        const int n_n2 = pos + 2 < len ? (int)(o_n3 - o_n2) : 0;
        const uint32_t id_n2 = (lane < n_n2 && n_n2 <= CAP) ? a.map_nodes[o_n2 + lane] : 0u;
        const uint8_t x_n2 = pos + 2 < len ? a.bases[b0 + pos + 2] : (uint8_t)0;
        HPROF(0)
        // ---- hash of this position's list
        const bool first = pos == 0;
        const int n = n_cur;
        const bool has = lane < n;
        uint2 *hc = sh.ent[pos & 1];
        const uint2 *hp = sh.ent[(pos + 1) & 1];
        for (int h = lane; h < HL_HASH; h += 64) hc[h].x = 0xffffffffu;
        wave_sync();
        if (has) {
            uint32_t h = hl_hash(id_cur);
            for (;;) {
                const uint32_t old = atomicCAS(&hc[h].x, 0xffffffffu, id_cur);
                if (old == 0xffffffffu) break;
                if (old == id_cur) {
                    err |= SP_ERR_DUPLICATE;
                    break;
                }
                h = (h + 1) & (HL_HASH - 1);
            }
            hc[h].y = (uint32_t)lane;
        }
        wave_sync();
        HPROF(1)
        // ---- fm, fi (forward.rs:337-388), fib (541-545)
        // (InsBegin of the previous column in that column's scale, fib forward.rs:541-545: carried along with the
        // exact power-of-two rescales instead of exp(logib[pos-1] - E ln 2) per position)
        const double c_begin = HStep::c_begin(lp, first, ibs);
        const double ib_cur = HStep::ib_cur(lp, first, ibs);
        const double c_del = __dmul_rn(lp.p_ID, ib_cur);
        const bool hadp = lane < n_prev;
        const double G = hadp ? HStep::lin3(lp.p_MM, pm, lp.p_IM, pi, lp.p_DM, pd) : 0.0;
        const double H = hadp ? HStep::lin3(lp.p_MI, pm, lp.p_II, pi, lp.p_DI, pd) : 0.0;
        int ps[ADJ_DEG], cs[ADJ_DEG];
        uint32_t anyq = 0;  // wave-uniform: some lane has a q-th parent (on a DBG mostly q = 0 only)
#pragma unroll
        for (int q = 0; q < ADJ_DEG; q++) {
            const bool use = has && q < (int)rc_cur.npar && w_cur[q] != 0.0;
            ps[q] = cs[q] = -1;
            if (__ballot(use) != 0ull) {
                anyq |= 1u << q;
                ps[q] = (use && !first) ? hl_find(hp, rc_cur.par[q]) : -1;
                cs[q] = use ? hl_find(hc, rc_cur.par[q]) : -1;
            }
        }
        const int os = (has && !first) ? hl_find(hp, id_cur) : -1;
        HPROF(2)
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < ADJ_DEG; q++) {
            if (!(anyq & (1u << q))) continue;
            const double v = hl_shfl(G, ps[q]);
            if (ps[q] >= 0) acc = HStep::acc(w_cur[q], v, acc);
        }
        const double hv = hl_shfl(H, os);
        m = ii = d = 0.0;
        if (has) {
            const double pe = rc_cur.emis == x_cur ? lp.p_match : lp.p_mismatch;
            m = HStep::match(pe, acc, in_cur, c_begin);
            ii = os >= 0 ? __dmul_rn(lp.p_random, hv) : 0.0;
        }
        HPROF(3)
        // ---- fd0 + n_max_gaps x fdt restricted to the list (forward.rs:423-524)
        double lv = HStep::lv(lp, m, ii);
        for (int t = 0; t <= lp.n_max_gaps; t++) {
            double sacc = 0.0;
#pragma unroll
            for (int q = 0; q < ADJ_DEG; q++) {
                if (!(anyq & (1u << q))) continue;
                const double v = hl_shfl(lv, cs[q]);
                if (cs[q] >= 0) sacc = HStep::acc(w_cur[q], v, sacc);
            }
            if (t == 0) sacc = __fma_rn(in_cur, c_del, sacc);
            else sacc = __dmul_rn(sacc, lp.p_DD);
            sacc = has ? sacc : 0.0;
            d = __dadd_rn(d, sacc);
            lv = sacc;
        }
        HPROF(4)
        // ---- rescale so that the column maximum is in [0.5, 1)
        const double mx = wave_max(fmax(has ? fmax(fmax(m, ii), d) : 0.0, ib_cur));
        const int e = sp_exp_of(mx);
        collapse |= !first && e < HINT_COLLAPSE_EXP;
        const double sc = sp_pow2(-e);
        m = __dmul_rn(m, sc);
        ii = __dmul_rn(ii, sc);
        d = __dmul_rn(d, sc);
        const int Ecur = (first ? 0 : Eprev) + e;
        ibs = __dmul_rn(ib_cur, sc);
        if (a.pool.base && cand == 0) {
            // forward record of the position (every entry carries m, i and d)
            const uint64_t idb = (uint64_t)((n + 1) & ~1) * 4;
            const uint64_t bytes = 16 + idb + (uint64_t)(3 * n) * 8;
            const uint64_t o = pool_alloc(a.pool, bytes);
            if (o + bytes > a.pool.cap) err |= SP_ERR_POOL;
            else {
                uint8_t *rec = a.pool.base + o;
                if (lane == 0) {
                    ((uint32_t *)rec)[0] = (uint32_t)n;
                    ((uint32_t *)rec)[1] = (uint32_t)n;
                    ((int *)rec)[2] = Ecur;
                    ((uint32_t *)rec)[3] = 0;
                    a.pool.off[b0 + pos] = o + 8;
                }
                if (has) {
                    ((uint32_t *)(rec + 16))[lane] = id_cur;
                    double *om = (double *)(rec + 16 + idb);
                    om[lane] = m;
                    om[n + lane] = ii;
                    om[2 * n + lane] = d;
                }
            }
        }
        HPROF(5)
        // ---- the column becomes the previous one; weights of the next position (its record has arrived)
        pm = m;
        pi = ii;
        pd = d;
        Eprev = Ecur;
        n_prev = n;
#pragma unroll
        for (int q = 0; q < ADJ_DEG; q++) w_cur[q] = q < (int)rc_nx.npar ? trans[rc_nx.pedge[q]] : 0.0;
        o_cur = o_nx;
        o_nx = o_n2;
        o_n2 = o_n3;
        n_cur = n_nx;
        n_nx = n_n2;
        id_cur = id_nx;
        id_nx = id_n2;
        rc_cur = rc_nx;
        in_cur = in_nx;
        x_cur = x_nx;
        x_nx = x_n2;
        HPROF(6)
    }
#ifdef PHMM_LEAN_PROF
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && len > 0)
        printf("hinted_lean prof: len %d | requests %lld hash %lld exp+lookups %lld fm %lld del %lld rescale+store %lld rotate+weights %lld (cycles/step)\n",
               len, pt[0] / len, pt[1] / len, pt[2] / len, pt[3] / len, pt[4] / len, pt[5] / len, pt[6] / len);
#endif
    for (int off = 32; off >= 1; off >>= 1) err |= (uint32_t)__shfl_xor((int)err, off);
    // fe (forward.rs:554-558) of the last column
    const double stot = wave_sum(lane < n_prev ? __dadd_rn(__dadd_rn(pm, pi), pd) : 0.0);
    const double lpv = err ? NAN : (collapse ? -INFINITY : HStep::log_end(lp, stot, Eprev));
    if (lane == 0) {
        a.out_logp[(size_t)cand * a.R + rd] = lpv;
        a.err[(size_t)cand * a.R + rd] = err;
    }
}

// ---------------------------------------------------------------- candidate batches: several candidates per wave
// The loop of sample_posterior_once (multi_dbg/posterior.rs:483-515) runs the SAME reads over the SAME lists for
// every candidate copy-number vector; only init / trans differ.  A mapping list holds ~5 nodes (cfg3: 98.9 % of
// the positions <= 8, all but 3e-5 <= 16), so one lane per list entry leaves 9 lanes in 10 idle, and a wave per
// (read, candidate) repeats the whole topology walk -- list, hash, parent lookups -- per candidate.  Here a wave
// owns one read and G = 64 / WG candidates: lane = (candidate group, list slot), slots < WG.  The list, its hash
// and the parent / own-previous lookups are done once per position for all groups; per candidate only init[node],
// trans[edge] and the arithmetic differ; reductions (column maximum, end sum) are segmented over the WG lanes of a
// group.  Reads whose longest list exceeds WG take the next class (16, 32, then the one-candidate kernels).
// Same sums in the same order as hinted_lean_kernel: bit-equal results.
// maximum of NON-NEGATIVE values over the WG lanes of a group, in every lane, on the DPP path (VALU moves, no LDS
// round trips -- the kernel is VALU-issue bound, profiles/r2_candidates64_sq_counters.txt, and a __shfl_xor of a
// double is two ds_bpermute plus their address arithmetic): quad permutes and the half-row mirror for 8 lanes, row
// rotations for a row of 16, one cross-row shuffle on top for 32.
template <int WG> __device__ __forceinline__ double group_max(double v) {
    static_assert(WG == 8 || WG == 16 || WG == 32, "group width");
    if (WG == 8) {
        v = fmax(v, dpp_d<0xB1, 0xf>(0.0, v));   // quad_perm [1,0,3,2]
        v = fmax(v, dpp_d<0x4E, 0xf>(0.0, v));   // quad_perm [2,3,0,1]
        v = fmax(v, dpp_d<0x141, 0xf>(0.0, v));  // row_half_mirror: the other quad of the 8
        return v;
    }
    v = fmax(v, dpp_d<0x128, 0xf>(0.0, v));  // row_ror 8, 4, 2, 1
    v = fmax(v, dpp_d<0x124, 0xf>(0.0, v));
    v = fmax(v, dpp_d<0x122, 0xf>(0.0, v));
    v = fmax(v, dpp_d<0x121, 0xf>(0.0, v));
    if (WG == 32) v = fmax(v, __shfl_xor(v, 16));
    return v;
}
// Sum over the WG lanes of a group in the association of wave_sum (sparse_dev.h: an inclusive Hillis-Steele scan read
// at the last lane) restricted to the group -- lanes beyond a list are zeros there and here, so a candidate's end
// sum has the same bits whether its list sat alone on a wave or in a group.
template <int WG> __device__ __forceinline__ double group_sum(double v, int slot) {
#pragma unroll
    for (int off = 1; off < WG; off <<= 1) {
        const double t = __shfl_up(v, off, WG);
        v += slot >= off ? t : 0.0;
    }
    return __shfl(v, WG - 1, WG);
}

template <int WG, int CPL>
__global__ void __launch_bounds__(64) hinted_packed_kernel(const HintedArgs a, const uint32_t n_cand) {
    // CPL candidates per lane on top of the G = 64 / WG candidate groups of a wave: the topology work of a position
    // (list, records, hash, lookups: ~100 VALU wave-instructions) is shared by G x CPL candidates, whose own work
    // (~70 each: weights, G / H, shuffles, Del levels, rescale) runs CPL times per lane.
    constexpr int G = 64 / WG;
    __shared__ HintedLeanShared sh;
    const int lane = threadIdx.x;
    const int slot = lane % WG, grp = lane / WG, gbase = grp * WG;
    const uint32_t rd = a.read_ids[blockIdx.x];
    uint32_t cand[CPL];
    bool cand_ok[CPL];
    const double *init[CPL], *trans[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        cand[c] = (blockIdx.y * G + (uint32_t)grp) * CPL + c;
        cand_ok[c] = cand[c] < n_cand;
        if (!cand_ok[c]) cand[c] = n_cand - 1;  // (idle slots of the last wave compute a copy; nothing is written)
        init[c] = a.init_c + (size_t)cand[c] * a.M.N;
        trans[c] = a.trans_c + (size_t)cand[c] * a.E;
    }
    const ParRec *prec = a.M.prec;
    const LinParams &lp = a.M.lp;
    const uint64_t b0 = a.read_off[rd];
    const int len = (int)(a.read_off[rd + 1] - b0);
    const uint64_t *po = a.map_pos_off + b0;
    uint32_t err = 0;
    uint64_t o_cur = po[0], o_nx = po[1], o_n2 = po[len >= 2 ? 2 : 1];
    int n_cur = (int)(o_nx - o_cur), n_nx = len >= 2 ? (int)(o_n2 - o_nx) : 0;
    uint32_t id_cur = (slot < n_cur && n_cur <= WG) ? a.map_nodes[o_cur + slot] : 0u;
    uint32_t id_nx = (slot < n_nx && n_nx <= WG) ? a.map_nodes[o_nx + slot] : 0u;
    ParRec rc_cur = prec[id_cur];
    double in_cur[CPL], w_cur[CPL][ADJ_DEG];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        in_cur[c] = init[c][id_cur];
#pragma unroll
        for (int q = 0; q < ADJ_DEG; q++) w_cur[c][q] = q < (int)rc_cur.npar ? trans[c][rc_cur.pedge[q]] : 0.0;
    }
    uint8_t x_cur = a.bases[b0], x_nx = len >= 2 ? a.bases[b0 + 1] : (uint8_t)0;
    double pm[CPL], pi[CPL], pd[CPL], ibs[CPL];
    int Eprev[CPL];
    bool collapse[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        pm[c] = pi[c] = pd[c] = ibs[c] = 0.0;
        Eprev[c] = 0;
        collapse[c] = false;
    }
    int n_prev = 0;
    for (int pos = 0; pos < len; pos++) {
        if (n_cur > WG) {
            err |= SP_ERR_CAPACITY;
            break;
        }
        // ---- requests for the positions ahead
        const ParRec rc_nx = prec[id_nx];
        double in_nx[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++) in_nx[c] = init[c][id_nx];
        const uint64_t o_n3 = po[pos + 3 <= len ? pos + 3 : len];
        const int n_n2 = pos + 2 < len ? (int)(o_n3 - o_n2) : 0;
        const uint32_t id_n2 = (slot < n_n2 && n_n2 <= WG) ? a.map_nodes[o_n2 + slot] : 0u;
        const uint8_t x_n2 = pos + 2 < len ? a.bases[b0 + pos + 2] : (uint8_t)0;
        // ---- hash of this position's list: node -> slot (group 0 inserts; every group reads)
        const bool first = pos == 0;
        const int n = n_cur;
        const bool has = slot < n;
        uint2 *hc = sh.ent[pos & 1];
        const uint2 *hp = sh.ent[(pos + 1) & 1];
        for (int h = lane; h < HL_HASH; h += 64) hc[h].x = 0xffffffffu;
        wave_sync();
        if (has && grp == 0) {
            uint32_t h = hl_hash(id_cur);
            for (;;) {
                const uint32_t old = atomicCAS(&hc[h].x, 0xffffffffu, id_cur);
                if (old == 0xffffffffu) break;
                if (old == id_cur) {
                    err |= SP_ERR_DUPLICATE;
                    break;
                }
                h = (h + 1) & (HL_HASH - 1);
            }
            hc[h].y = (uint32_t)slot;
        }
        wave_sync();
        // ---- the topology of the position, once for every candidate of the wave: lanes of the in-list parents in the
        // previous (ps) and the current (cs) list, own lane in the previous list (os)
        const bool hadp = slot < n_prev;
        int ps[ADJ_DEG], cs[ADJ_DEG];
        uint32_t anyq = 0;  // wave-uniform: some lane has a q-th parent (on a DBG mostly q = 0 only)
#pragma unroll
        for (int q = 0; q < ADJ_DEG; q++) {
            // (the topology test only: a candidate whose weight is 0 multiplies by it -- the one-candidate kernel
            // skips such a parent, which adds the same +0.0)
            const bool use = has && q < (int)rc_cur.npar;
            ps[q] = cs[q] = -1;
            if (__ballot(use) != 0ull) {
                anyq |= 1u << q;
                ps[q] = (use && !first) ? hl_find(hp, rc_cur.par[q]) : -1;
                cs[q] = use ? hl_find(hc, rc_cur.par[q]) : -1;
            }
        }
        const int os = (has && !first) ? hl_find(hp, id_cur) : -1;
        const double pe = rc_cur.emis == x_cur ? lp.p_match : lp.p_mismatch;
        // byte addresses of the shuffles (ds_bpermute), once for all candidates of the lane
        int aps[ADJ_DEG], acs[ADJ_DEG];
#pragma unroll
        for (int q = 0; q < ADJ_DEG; q++) {
            aps[q] = (gbase + (ps[q] < 0 ? 0 : ps[q])) << 2;
            acs[q] = (gbase + (cs[q] < 0 ? 0 : cs[q])) << 2;
        }
        const int aos = (gbase + (os < 0 ? 0 : os)) << 2;
        auto pull = [](int addr, double v) -> double {
            const long long b = __double_as_longlong(v);
            const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)(b & 0xffffffffll));
            const int hi = __builtin_amdgcn_ds_bpermute(addr, (int)(b >> 32));
            return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
        };
        // ---- per candidate: fm, fi (forward.rs:337-388), fib (541-545), fd0 + n_max_gaps x fdt restricted to the list
        // (forward.rs:423-524), rescale.  Branch-free: a parent that is not in the list (or whose edge has weight 0 for
        // this candidate) enters with weight 0 -- fma(0, v, s) = s exactly, what skipping it gives.
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const double c_begin = HStep::c_begin(lp, first, ibs[c]);
            const double ib_cur = HStep::ib_cur(lp, first, ibs[c]);
            const double c_del = __dmul_rn(lp.p_ID, ib_cur);
            const double Gv = hadp ? HStep::lin3(lp.p_MM, pm[c], lp.p_IM, pi[c], lp.p_DM, pd[c]) : 0.0;
            const double Hv = hadp ? HStep::lin3(lp.p_MI, pm[c], lp.p_II, pi[c], lp.p_DI, pd[c]) : 0.0;
            const double hv = pull(aos, Hv);
            double m, ii, d = 0.0;
            if (anyq <= 1u) {
                // the common case on a DBG: at most one in-list parent per entry
                const double wp = ps[0] >= 0 ? w_cur[c][0] : 0.0, wc = cs[0] >= 0 ? w_cur[c][0] : 0.0;
                const double acc = HStep::acc(wp, pull(aps[0], Gv), 0.0);
                m = has ? HStep::match(pe, acc, in_cur[c], c_begin) : 0.0;
                ii = (has && os >= 0) ? __dmul_rn(lp.p_random, hv) : 0.0;
                double lv = HStep::lv(lp, m, ii);
                for (int t = 0; t <= lp.n_max_gaps; t++) {
                    double sacc = HStep::acc(wc, pull(acs[0], lv), 0.0);
                    sacc = t == 0 ? __fma_rn(in_cur[c], c_del, sacc) : __dmul_rn(sacc, lp.p_DD);
                    sacc = has ? sacc : 0.0;
                    d = __dadd_rn(d, sacc);
                    lv = sacc;
                }
            } else {
                double wp[ADJ_DEG], wc[ADJ_DEG];
#pragma unroll
                for (int q = 0; q < ADJ_DEG; q++) {
                    wp[q] = ps[q] >= 0 ? w_cur[c][q] : 0.0;
                    wc[q] = cs[q] >= 0 ? w_cur[c][q] : 0.0;
                }
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < ADJ_DEG; q++)
                    if (anyq & (1u << q)) acc = HStep::acc(wp[q], pull(aps[q], Gv), acc);
                m = has ? HStep::match(pe, acc, in_cur[c], c_begin) : 0.0;
                ii = (has && os >= 0) ? __dmul_rn(lp.p_random, hv) : 0.0;
                double lv = HStep::lv(lp, m, ii);
                for (int t = 0; t <= lp.n_max_gaps; t++) {
                    double sacc = 0.0;
#pragma unroll
                    for (int q = 0; q < ADJ_DEG; q++)
                        if (anyq & (1u << q)) sacc = HStep::acc(wc[q], pull(acs[q], lv), sacc);
                    sacc = t == 0 ? __fma_rn(in_cur[c], c_del, sacc) : __dmul_rn(sacc, lp.p_DD);
                    sacc = has ? sacc : 0.0;
                    d = __dadd_rn(d, sacc);
                    lv = sacc;
                }
            }
            // rescale so that the column maximum of THIS candidate is in [0.5, 1)
            const double mx = group_max<WG>(fmax(fmax(fmax(m, ii), d), ib_cur));
            const int e = sp_exp_of(mx);
            collapse[c] |= !first && e < HINT_COLLAPSE_EXP;
            const double sc = sp_pow2(-e);
            pm[c] = __dmul_rn(m, sc);
            pi[c] = __dmul_rn(ii, sc);
            pd[c] = __dmul_rn(d, sc);
            Eprev[c] = (first ? 0 : Eprev[c]) + e;
            ibs[c] = __dmul_rn(ib_cur, sc);
        }
        // ---- the column becomes the previous one; weights of the next position (its record has arrived)
        n_prev = n;
        {
            // (edge weights only for parent slots some lane of the wave uses: mostly the first)
            int npmax = (int)rc_nx.npar;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) npmax = max(npmax, __shfl_xor(npmax, off));
            npmax = __builtin_amdgcn_readfirstlane(npmax);
#pragma unroll
            for (int c = 0; c < CPL; c++) {
#pragma unroll
                for (int q = 0; q < ADJ_DEG; q++) {
                    if (q < npmax) w_cur[c][q] = q < (int)rc_nx.npar ? trans[c][rc_nx.pedge[q]] : 0.0;
                    else w_cur[c][q] = 0.0;
                }
                in_cur[c] = in_nx[c];
            }
        }
        o_cur = o_nx;
        o_nx = o_n2;
        o_n2 = o_n3;
        n_cur = n_nx;
        n_nx = n_n2;
        id_cur = id_nx;
        id_nx = id_n2;
        rc_cur = rc_nx;
        x_cur = x_nx;
        x_nx = x_n2;
    }
    for (int off = 32; off >= 1; off >>= 1) err |= (uint32_t)__shfl_xor((int)err, off);
    // fe (forward.rs:554-558) of the last column, per candidate
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const double stot = group_sum<WG>(slot < n_prev ? __dadd_rn(__dadd_rn(pm[c], pi[c]), pd[c]) : 0.0, slot);
        const double lpv = err ? NAN : (collapse[c] ? -INFINITY : HStep::log_end(lp, stot, Eprev[c]));
        if (slot == 0 && cand_ok[c]) {
            a.out_logp[(size_t)cand[c] * a.R + rd] = lpv;
            a.err[(size_t)cand[c] * a.R + rd] = err;
        }
    }
}

// ---- wide-range fallback of the hinted forward -----------------------------------------------------------
// forward_with_mappings (forward.rs:79-89, 276-306, 337-388, 423-524, 541-558: fm, fi, fib, fd0 + n_max_gaps x fdt
// over the position's list, fe at the end) with EVERY value carrying its own binary exponent -- a double mantissa
// in [0.5, 1) and an int exponent -- so that nothing underflows whatever its distance from the column's best value:
// the range of the reference's natural-log f64 with plain multiply-adds instead of a log-sum-exp per term.  One
// wave per (candidate, read).  It runs for the pairs the scaled kernels returned -inf for.  That happens when every
// node of a read's lists dies -- a candidate that sets a k-mer on the read's path to copy number 0 -- hundreds of
// positions into the read: the reference then still holds the InsBegin chain (p_random * p_II per base, never
// zero), which re-enters the graph behind the cut, and returns a finite ln P of about -7 per base of the cut-off
// prefix.  In the scaled linear domain (ONE exponent per column) that chain is 2^-1075 below the read's own path
// after ~105 bases and is gone, so the restart is lost; here it is not.
struct XF {
    double m;  // 0, or in [0.5, 1)
    int e;
};
__device__ __forceinline__ XF xf_norm(double v, int e) {
    if (v == 0.0) return XF{0.0, 0};
    int k;
    const double m = frexp(v, &k);
    return XF{m, e + k};
}
__device__ __forceinline__ XF xf_from(double p) { return xf_norm(p, 0); }
__device__ __forceinline__ XF xf_mul(XF a, double p) { return xf_norm(a.m * p, a.e); }
__device__ __forceinline__ XF xf_mulx(XF a, XF b) { return xf_norm(a.m * b.m, a.e + b.e); }
__device__ __forceinline__ XF xf_add(XF a, XF b) {
    if (a.m == 0.0) return b;
    if (b.m == 0.0) return a;
    if (a.e < b.e) {
        const XF t = a;
        a = b;
        b = t;
    }
    const int d = b.e - a.e;
    if (d < -1000) return a;
    return xf_norm(a.m + ldexp(b.m, d), a.e);
}
__device__ __forceinline__ double xf_log(XF a) { return a.m == 0.0 ? -INFINITY : log(a.m) + (double)a.e * SP_LN2; }

// (two sizes: lists of up to 64 nodes -- nearly every read -- keep 9 KB of LDS per wave, so that thousands of pairs
// run at once; the full 400-slot version holds 46 KB)
template <int XCAP, int XHASH> struct ExactCol {
    uint32_t id[XCAP];
    double m[XCAP], i[XCAP], d[XCAP];
    int me[XCAP], ie[XCAP], de[XCAP];
    uint32_t hkey[XHASH];
    uint16_t hval[XHASH];
};
template <int XCAP, int XHASH> __device__ __forceinline__ int xl_find(const ExactCol<XCAP, XHASH> &c, uint32_t key) {
    uint32_t h = ((key * 2654435761u) >> 16) & (XHASH - 1);
    for (;;) {
        const uint32_t k = c.hkey[h];
        if (k == key) return (int)c.hval[h];
        if (k == 0xffffffffu) return -1;
        h = (h + 1) & (XHASH - 1);
    }
}
template <int XCAP, int XHASH>
__global__ void __launch_bounds__(64) hinted_exact_kernel(const HintedArgs a, const uint2 *pairs, double *res) {
    typedef ExactCol<XCAP, XHASH> XCol;
    __shared__ XCol cols[2];
    __shared__ double ta[XCAP], tb[XCAP];
    __shared__ int tae[XCAP], tbe[XCAP];
    const int lane = threadIdx.x;
    const uint32_t cand = pairs[blockIdx.x].x, rd = pairs[blockIdx.x].y;
    const double *init = a.init_c + (size_t)cand * a.M.N;
    const double *trans = a.trans_c + (size_t)cand * a.E;
    const LinParams &lp = a.M.lp;
    const uint64_t b0 = a.read_off[rd], len = a.read_off[rd + 1] - b0;
    XF mb = xf_from(1.0), ib = XF{0.0, 0};  // f_init (forward.rs:255-266)
    int np = 0, cur = 0;
    bool bad = false;
    for (uint64_t pos = 0; pos < len; pos++, cur ^= 1) {
        XCol &C = cols[cur];
        const XCol &P = cols[cur ^ 1];
        const uint64_t e0 = a.map_pos_off[b0 + pos];
        const int n = (int)(a.map_pos_off[b0 + pos + 1] - e0);
        if (n > XCAP) {
            bad = true;
            break;
        }
        const uint8_t x = a.bases[b0 + pos];
        for (int h = lane; h < XHASH; h += 64) C.hkey[h] = 0xffffffffu;
        wave_sync();
        for (int j = lane; j < n; j += 64) {
            const uint32_t k = a.map_nodes[e0 + j];
            C.id[j] = k;
            uint32_t h = ((k * 2654435761u) >> 16) & (XHASH - 1);
            for (;;) {
                const uint32_t old = atomicCAS(&C.hkey[h], 0xffffffffu, k);
                if (old == 0xffffffffu) {
                    C.hval[h] = (uint16_t)j;
                    break;
                }
                if (old == k) break;  // (a node listed twice: the first entry stands)
                h = (h + 1) & (XHASH - 1);
            }
        }
        wave_sync();
        // fm, fi from the previous column (empty at pos 0)
        const XF beg_m = xf_add(xf_mul(mb, lp.p_MM), xf_mul(ib, lp.p_IM));
        for (int j = lane; j < n; j += 64) {
            const uint32_t k = C.id[j];
            XF from_normal{0.0, 0};
            for (uint32_t e = a.M.par_off[k]; e < a.M.par_off[k + 1]; e++) {
                const int q = np > 0 ? xl_find(P, a.M.par_node[e]) : -1;
                if (q < 0) continue;
                const XF inner = xf_add(xf_add(xf_mul(XF{P.m[q], P.me[q]}, lp.p_MM), xf_mul(XF{P.i[q], P.ie[q]}, lp.p_IM)),
                                        xf_mul(XF{P.d[q], P.de[q]}, lp.p_DM));
                from_normal = xf_add(from_normal, xf_mul(inner, trans[a.M.par_edge[e]]));
            }
            const XF mv = xf_mul(xf_add(from_normal, xf_mul(beg_m, init[k])), a.M.emis[k] == x ? lp.p_match : lp.p_mismatch);
            C.m[j] = mv.m;
            C.me[j] = mv.e;
            const int me = np > 0 ? xl_find(P, k) : -1;
            XF iv{0.0, 0};
            if (me >= 0)
                iv = xf_mul(xf_add(xf_add(xf_mul(XF{P.m[me], P.me[me]}, lp.p_MI), xf_mul(XF{P.i[me], P.ie[me]}, lp.p_II)),
                                   xf_mul(XF{P.d[me], P.de[me]}, lp.p_DI)),
                            lp.p_random);
            C.i[j] = iv.m;
            C.ie[j] = iv.e;
        }
        ib = xf_mul(xf_add(xf_mul(mb, lp.p_MI), xf_mul(ib, lp.p_II)), lp.p_random);  // fib; fmb = 0
        mb = XF{0.0, 0};
        wave_sync();
        // fd0 from this column's m, i; then the Del levels over the same list
        const XF beg_d = xf_mul(ib, lp.p_ID);  // (mb = 0)
        for (int j = lane; j < n; j += 64) {
            const uint32_t k = C.id[j];
            XF from_normal{0.0, 0};
            for (uint32_t e = a.M.par_off[k]; e < a.M.par_off[k + 1]; e++) {
                const int q = xl_find(C, a.M.par_node[e]);
                if (q < 0) continue;
                const XF g = xf_add(xf_mul(XF{C.m[q], C.me[q]}, lp.p_MD), xf_mul(XF{C.i[q], C.ie[q]}, lp.p_ID));
                from_normal = xf_add(from_normal, xf_mul(g, trans[a.M.par_edge[e]]));
            }
            const XF v = xf_add(from_normal, xf_mul(beg_d, init[k]));
            ta[j] = v.m;
            tae[j] = v.e;
            C.d[j] = v.m;
            C.de[j] = v.e;
        }
        double *src = ta, *dst = tb;
        int *srce = tae, *dste = tbe;
        for (int t = 0; t < lp.n_max_gaps; t++) {
            wave_sync();
            for (int j = lane; j < n; j += 64) {
                const uint32_t k = C.id[j];
                XF sacc{0.0, 0};
                for (uint32_t e = a.M.par_off[k]; e < a.M.par_off[k + 1]; e++) {
                    const int q = xl_find(C, a.M.par_node[e]);
                    if (q < 0) continue;
                    sacc = xf_add(sacc, xf_mul(XF{src[q], srce[q]}, trans[a.M.par_edge[e]] * lp.p_DD));
                }
                dst[j] = sacc.m;
                dste[j] = sacc.e;
                const XF dv = xf_add(XF{C.d[j], C.de[j]}, sacc);
                C.d[j] = dv.m;
                C.de[j] = dv.e;
            }
            double *tmp = src;
            src = dst;
            dst = tmp;
            int *tmpe = srce;
            srce = dste;
            dste = tmpe;
        }
        wave_sync();
        np = n;
    }
    // fe over the last column's list
    double lpv = -INFINITY;
    if (!bad && len > 0) {
        const XCol &L = cols[cur ^ 1];
        XF e{0.0, 0};
        for (int j = lane; j < np; j += 64)
            e = xf_add(e, xf_add(xf_add(XF{L.m[j], L.me[j]}, XF{L.i[j], L.ie[j]}), XF{L.d[j], L.de[j]}));
        for (int off = 32; off >= 1; off >>= 1) {
            XF o;
            o.m = __shfl_xor(e.m, off);
            o.e = __shfl_xor(e.e, off);
            e = xf_add(e, o);
        }
        lpv = xf_log(xf_mul(e, lp.p_end));
    }
    if (lane == 0) res[blockIdx.x] = bad ? NAN : lpv;
}

__global__ void __launch_bounds__(256) exp_kernel(const double *in, double *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const double v = in[i];
        out[i] = v == -INFINITY ? 0.0 : exp(v);
    }
}

void upload_reads(const phmm_reads *r) {
    if (r->on_device) return;
    r->d_bases.upload(r->bases.data(), r->bases.size());
    r->d_off.upload(r->off.data(), r->off.size() * sizeof(uint64_t));
    HIP_CHECK(hipStreamSynchronize(current_stream()));
    r->on_device = true;
}
void upload_mappings(const phmm_mappings *mp) {
    if (mp->on_device) return;
    mp->d_pos_off.upload(mp->pos_off.data(), mp->pos_off.size() * sizeof(uint64_t));
    mp->d_nodes.upload(mp->nodes.data(), std::max<size_t>(mp->nodes.size(), 1) * sizeof(uint32_t));
    HIP_CHECK(hipStreamSynchronize(current_stream()));
    mp->on_device = true;
}

namespace {

SparseModel sparse_model(const phmm_model *m) {
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
    s.prec = d.prec.as<ParRec>();
    s.lp = m->lin;
    s.logib = d.logib.as<double>();
    return s;
}

struct EvTimer {
    hipEvent_t a = nullptr, b = nullptr;
    bool on;
    explicit EvTimer(bool on_) : on(on_) {
        if (on) {
            HIP_CHECK(hipEventCreate(&a));
            HIP_CHECK(hipEventCreate(&b));
            HIP_CHECK(hipEventRecord(a, current_stream()));
        }
    }
    ~EvTimer() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
    double stop() {
        if (!on) return 0.0;
        HIP_CHECK(hipEventRecord(b, current_stream()));
        HIP_CHECK(hipEventSynchronize(b));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        return ms;
    }
};

template <int CAP, int LPN> void launch_hinted(const HintedArgs &a, uint32_t n_reads, uint32_t n_cand) {
    if (!n_reads) return;
    hipLaunchKernelGGL((hinted_score_kernel<CAP, LPN>), dim3(n_reads, n_cand), dim3(64), 0, current_stream(), a);
}

}  // namespace

// Candidate (init, trans) vectors straight from copy-number vectors, in the linear domain
// (SeqGraph::to_phmm_node / to_phmm_edge without edge copy numbers, seq_graph.rs:160-209, with
// total_emittable_copy_num / total_emittable_child_copy_nums, seq_graph.rs:110-135):
//   init[v]        = emittable(v) ? max(cn[v], min) / sum_{emittable u} max(cn[u], min) : 0
//   trans[v -> w]  = emittable(w) and T_v > 0 ? max(cn[w], min) / T_v : 0,
//   T_v            = sum over the emittable children u of v of max(cn[u], min)
// Copy numbers are integers: the sums are exact and order-independent.
__global__ void __launch_bounds__(256) cn_totals(const uint32_t *cn, const uint8_t *emis, uint32_t N, uint32_t min_cn,
                                                 unsigned long long *tot) {
    const uint32_t c = blockIdx.y;
    unsigned long long s = 0;
    for (uint32_t v = blockIdx.x * 256 + threadIdx.x; v < N; v += gridDim.x * 256)
        if (emis[v] != (uint8_t)'n') s += max(cn[(size_t)c * N + v], min_cn);
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&tot[c], s);
}
__global__ void __launch_bounds__(256) cn_probs(const uint32_t *cn, const uint8_t *emis, const uint32_t *chi_off,
                                                const uint32_t *chi_node, const uint32_t *chi_edge, uint32_t N, uint32_t E,
                                                uint32_t min_cn, const unsigned long long *tot, double *init, double *trans) {
    const uint32_t c = blockIdx.y;
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= N) return;
    const uint32_t *cc = cn + (size_t)c * N;
    const unsigned long long total = tot[c];
    init[(size_t)c * N + v] = (emis[v] != (uint8_t)'n' && total > 0) ? (double)max(cc[v], min_cn) / (double)total : 0.0;
    unsigned long long tv = 0;
    for (uint32_t a = chi_off[v]; a < chi_off[v + 1]; a++) {
        const uint32_t u = chi_node[a];
        if (emis[u] != (uint8_t)'n') tv += max(cc[u], min_cn);
    }
    for (uint32_t a = chi_off[v]; a < chi_off[v + 1]; a++) {
        const uint32_t u = chi_node[a];
        const uint32_t k = max(cc[u], min_cn);
        trans[(size_t)c * E + chi_edge[a]] = (emis[u] != (uint8_t)'n' && tv > 0 && k > 0) ? (double)k / (double)tv : 0.0;
    }
}

void full_prob_reads_hinted(phmm_model *m, const phmm_reads *reads, const phmm_mappings *mp, uint32_t n_cand,
                            const double *init_logp, const double *trans_logp, double *out_logp,
                            double *out_total, const RecPool *pool, const uint32_t *copy_nums, uint32_t min_copy_num) {
    hipStream_t s = current_stream();
    CallStats &st = stats();
    st = CallStats();
    const uint64_t R = reads->R;
    const uint32_t N = m->N, E = m->E;
    if (m->dev.max_degree > 8)
        PHMM_THROW(PHMM_EINVAL, "sparse path supports node degree <= 8 (MultiDbg MAX_DEGREE is 5)");
    upload_reads(reads);
    upload_mappings(mp);
    ensure_logib(m, reads->max_len + 1);

    // candidate probabilities in the linear domain: [C][N], [C][E]
    DevBuf &cand_init = m->wset().aux[7], &cand_trans = m->wset().aux[8], &staging = m->wset().aux[9];
    const double *d_init = m->dev.init.as<double>();
    const double *d_trans = m->dev.trans_lin.as<double>();
    if (init_logp) {
        const size_t ni = (size_t)n_cand * N, ne = (size_t)n_cand * E;
        staging.reserve(std::max(ni, ne) * sizeof(double));
        cand_init.reserve(ni * sizeof(double));
        cand_trans.reserve(std::max<size_t>(ne, 1) * sizeof(double));
        HIP_CHECK(hipMemcpyAsync(staging.p, init_logp, ni * sizeof(double), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(exp_kernel, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, s, staging.as<double>(),
                           cand_init.as<double>(), ni);
        if (ne) {
            HIP_CHECK(hipStreamSynchronize(s));
            HIP_CHECK(hipMemcpyAsync(staging.p, trans_logp, ne * sizeof(double), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(exp_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, staging.as<double>(),
                               cand_trans.as<double>(), ne);
        }
        HIP_CHECK(hipStreamSynchronize(s));
        d_init = cand_init.as<double>();
        d_trans = cand_trans.as<double>();
    } else if (copy_nums) {
        const size_t ni = (size_t)n_cand * N, ne = (size_t)n_cand * E;
        staging.reserve(ni * sizeof(uint32_t) + 256 + n_cand * sizeof(unsigned long long));
        cand_init.reserve(ni * sizeof(double));
        cand_trans.reserve(std::max<size_t>(ne, 1) * sizeof(double));
        uint32_t *d_cn = staging.as<uint32_t>();
        unsigned long long *d_tot = (unsigned long long *)(staging.as<char>() + (ni * sizeof(uint32_t) + 255) / 256 * 256);
        HIP_CHECK(hipMemcpyAsync(d_cn, copy_nums, ni * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemsetAsync(d_tot, 0, n_cand * sizeof(unsigned long long), s));
        const unsigned nb = (unsigned)((N + 255) / 256);
        hipLaunchKernelGGL(cn_totals, dim3(std::min(nb, 256u), n_cand), dim3(256), 0, s, d_cn, m->dev.emis.as<uint8_t>(), N,
                           min_copy_num, d_tot);
        hipLaunchKernelGGL(cn_probs, dim3(nb, n_cand), dim3(256), 0, s, d_cn, m->dev.emis.as<uint8_t>(),
                           m->dev.chi_off.as<uint32_t>(), m->dev.chi_node.as<uint32_t>(), m->dev.chi_edge.as<uint32_t>(), N, E,
                           min_copy_num, d_tot, cand_init.as<double>(), cand_trans.as<double>());
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(s));  // the caller's copy_nums may go away
        d_init = cand_init.as<double>();
        d_trans = cand_trans.as<double>();
    }

    // capacity classes by the longest node list of each read.  Candidate batches: reads with short lists go to the
    // packed kernels (several candidates per wave) first.
    std::vector<uint32_t> cls[3], pcls[3];
    uint64_t cells = 0;
    const bool lean_ok = m->dev.max_degree <= (uint32_t)ADJ_DEG && !knobs().no_lean;
    const bool packed_ok = n_cand >= 2 && !pool && lean_ok && !knobs().no_packed;
    for (uint64_t r = 0; r < R; r++) {
        const uint32_t mx = mp->read_max_list[r];
        if (packed_ok && mx <= 32) pcls[mx <= 8 ? 0 : (mx <= 16 ? 1 : 2)].push_back((uint32_t)r);
        else cls[mx <= 64 ? 0 : (mx <= 128 ? 1 : 2)].push_back((uint32_t)r);
    }
    DevBuf &d_ids = m->wset().aux[10], &d_out = m->wset().aux[11], &d_err = m->wset().aux[12];
    d_ids.reserve(R * sizeof(uint32_t));
    d_out.reserve((size_t)n_cand * R * sizeof(double));
    d_err.reserve((size_t)n_cand * R * sizeof(uint32_t));
    std::vector<double> h_out((size_t)n_cand * R);
    std::vector<uint32_t> h_err((size_t)n_cand * R);

    HintedArgs a{};
    a.M = sparse_model(m);
    a.init_c = d_init;
    a.trans_c = d_trans;
    a.E = E;
    a.bases = reads->d_bases.as<uint8_t>();
    a.read_off = reads->d_off.as<uint64_t>();
    a.map_pos_off = mp->d_pos_off.as<uint64_t>();
    a.map_nodes = mp->d_nodes.as<uint32_t>();
    a.read_ids = d_ids.as<uint32_t>();
    a.R = R;
    a.out_logp = d_out.as<double>();
    a.err = d_err.as<uint32_t>();
    if (pool) a.pool = *pool;

    EvTimer tm(timing_enabled());
    for (int c = 0; c < 3; c++) {
        if (pcls[c].empty()) continue;
        HIP_CHECK(hipMemcpyAsync(d_ids.p, pcls[c].data(), pcls[c].size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        const unsigned nr = (unsigned)pcls[c].size();
        // candidates per wave = (64 / WG) x CPL; two per lane once a batch fills such waves
        const int cpl_env = knobs().packed_cpl;
        const int G = c == 0 ? 8 : (c == 1 ? 4 : 2);
        const int cpl = cpl_env > 0 ? cpl_env : (n_cand >= (uint32_t)(2 * G) ? 2 : 1);
        const unsigned per_wave = (unsigned)(G * (cpl >= 2 ? 2 : 1));
        const dim3 grid(nr, (n_cand + per_wave - 1) / per_wave);
        if (cpl >= 2) {
            if (c == 0) hipLaunchKernelGGL((hinted_packed_kernel<8, 2>), grid, dim3(64), 0, s, a, n_cand);
            else if (c == 1) hipLaunchKernelGGL((hinted_packed_kernel<16, 2>), grid, dim3(64), 0, s, a, n_cand);
            else hipLaunchKernelGGL((hinted_packed_kernel<32, 2>), grid, dim3(64), 0, s, a, n_cand);
        } else {
            if (c == 0) hipLaunchKernelGGL((hinted_packed_kernel<8, 1>), grid, dim3(64), 0, s, a, n_cand);
            else if (c == 1) hipLaunchKernelGGL((hinted_packed_kernel<16, 1>), grid, dim3(64), 0, s, a, n_cand);
            else hipLaunchKernelGGL((hinted_packed_kernel<32, 1>), grid, dim3(64), 0, s, a, n_cand);
        }
        HIP_CHECK(hipGetLastError());
        st.launches[2]++;
        HIP_CHECK(hipMemcpyAsync(h_err.data(), d_err.p, h_err.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));  // (the read-id list is reused by the next class)
        for (uint32_t rd : pcls[c]) {
            uint32_t e = 0;
            for (uint32_t k = 0; k < n_cand; k++) e |= h_err[(size_t)k * R + rd];
            if (!e) continue;
            if (e & SP_ERR_DUPLICATE) PHMM_THROW(PHMM_EINVAL, "duplicate node in a mapping list");
            cls[0].push_back(rd);  // (cannot happen with read_max_list right: the one-candidate kernels take it)
        }
    }
    for (int c = 0; c < 3; c++) {
        if (cls[c].empty()) continue;
        HIP_CHECK(hipMemcpyAsync(d_ids.p, cls[c].data(), cls[c].size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        if (c == 0 && lean_ok) {
            hipLaunchKernelGGL(hinted_lean_kernel, dim3((unsigned)cls[c].size(), n_cand), dim3(64), 0, s, a);
        } else if (c == 0) launch_hinted<64, 2>(a, (uint32_t)cls[c].size(), n_cand);
        else if (c == 1) launch_hinted<128, 4>(a, (uint32_t)cls[c].size(), n_cand);
        else launch_hinted<400, 8>(a, (uint32_t)cls[c].size(), n_cand);
        HIP_CHECK(hipGetLastError());
        st.launches[2]++;
        HIP_CHECK(hipMemcpyAsync(h_out.data(), d_out.p, h_out.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(h_err.data(), d_err.p, h_err.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        // reads whose in-list fan-in exceeded this class's link budget are promoted
        for (uint32_t rd : cls[c]) {
            uint32_t e = 0;
            for (uint32_t k = 0; k < n_cand; k++) e |= h_err[(size_t)k * R + rd];
            if (!e) continue;
            if (e & SP_ERR_POOL) PHMM_THROW(PHMM_EINTERNAL, "forward record pool exhausted");
            if (e & SP_ERR_DUPLICATE) PHMM_THROW(PHMM_EINVAL, "duplicate node in a mapping list");
            if ((e & (SP_ERR_LINKS | SP_ERR_CAPACITY)) && c < 2) cls[c + 1].push_back(rd);
            else PHMM_THROW(PHMM_ECAPACITY, "mapping list needs more than 400 slots / 8 in-list parents");
        }
    }
    HIP_CHECK(hipMemcpyAsync(h_out.data(), d_out.p, h_out.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    // (candidate, read) pairs that came back -inf: the reference's InsBegin chain may still carry them (see
    // hinted_exact_kernel) -- recomputed in its own arithmetic
    if (!pool && !knobs().no_exact_hinted) {
        std::vector<uint2> pairs;
        for (uint32_t k = 0; k < n_cand; k++)
            for (uint64_t r = 0; r < R; r++)
                if (h_out[(size_t)k * R + r] == -INFINITY && reads->off[r + 1] > reads->off[r]) pairs.push_back(make_uint2(k, (uint32_t)r));
        if (!pairs.empty()) {
            // short lists first (the small kernel), the rest behind them
            const size_t n_small = (size_t)(std::stable_partition(pairs.begin(), pairs.end(),
                                                                  [&](const uint2 &q) { return mp->read_max_list[q.y] <= 64; }) -
                                            pairs.begin());
            DevBuf &d_pairs = m->wset().aux[10];  // (the read ids of the classes are spent)
            d_pairs.reserve(pairs.size() * (sizeof(uint2) + sizeof(double)) + 256);
            uint2 *dp = d_pairs.as<uint2>();
            double *dres = (double *)(d_pairs.as<char>() + (pairs.size() * sizeof(uint2) + 255) / 256 * 256);
            HIP_CHECK(hipMemcpyAsync(dp, pairs.data(), pairs.size() * sizeof(uint2), hipMemcpyHostToDevice, s));
            if (n_small)
                hipLaunchKernelGGL((hinted_exact_kernel<64, 256>), dim3((unsigned)n_small), dim3(64), 0, s, a, (const uint2 *)dp, dres);
            if (n_small < pairs.size())
                hipLaunchKernelGGL((hinted_exact_kernel<PHMM_MAX_ACTIVE_NODES, 1024>), dim3((unsigned)(pairs.size() - n_small)),
                                   dim3(64), 0, s, a, (const uint2 *)(dp + n_small), dres + n_small);
            HIP_CHECK(hipGetLastError());
            std::vector<double> hres(pairs.size());
            HIP_CHECK(hipMemcpyAsync(hres.data(), dres, hres.size() * sizeof(double), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            for (size_t q = 0; q < pairs.size(); q++)
                if (hres[q] == hres[q]) h_out[(size_t)pairs[q].x * R + pairs[q].y] = hres[q];
            st.launches[2]++;
        }
    }
    if (pool) {
        // generate_mappings WITH lists (forward + backward over the lists): the backward pass needs the forward columns
        // of the scaled kernels, which a read that every list node cuts (a k-mer at probability 0 on its path) does not
        // have.  The reference calls this on to_non_zero_phmm (multi_dbg/posterior.rs:609-618), where no transition
        // is 0; a model that cuts a read is refused here instead of returning -inf / NaN posteriors.
        for (uint64_t r = 0; r < R; r++)
            if (h_out[r] == -INFINITY && reads->off[r + 1] > reads->off[r])
                PHMM_THROW(PHMM_EINVAL, "generate_mappings with mappings: read " + std::to_string(r) +
                                            " has probability 0 on its lists under this model (a zero-copy k-mer cuts it); "
                                            "use the non-zero PHMM (to_non_zero_phmm) for mapping, as the reference does");
    }
    st.ms[2] += tm.stop();
    cells = mp->total_entries;
    st.cells[2] = cells * n_cand;

    std::vector<double> tot(n_cand, 0.0);
    for (uint32_t k = 0; k < n_cand; k++)
        for (uint64_t r = 0; r < R; r++) tot[k] += h_out[(size_t)k * R + r];  // rayon .product(): sum of logs
    put_doubles(out_logp, h_out.data(), h_out.size());
    put_doubles(out_total, tot.data(), n_cand);
}

}  // namespace phmm

