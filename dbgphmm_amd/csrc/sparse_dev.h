// Device-side building blocks of the sparse (active-node frontier) kernels: one wave64
// per (read[, candidate]); the frontier of the previous and the current read position
// live in LDS as insertion-ordered (node, m, i, d) lists plus an open-addressing hash
// node -> slot, mirroring the reference's `SparseVec<Prob, NodeIndex, 400>` tables
// (src/hmmv2/table.rs:28, 42-73).  Values are linear probabilities scaled by 2^-E per
// column (exact power-of-two rescale after every column), so ln P = ln(sum) + E ln 2.
#pragma once

#include "phmm_internal.h"

namespace phmm {

static constexpr uint32_t H_EMPTY = 0xffffffffu;
static constexpr double SP_LN2 = 0.693147180559945309417232121458;

// error bits reported per read
enum : uint32_t {
    SP_ERR_LINKS = 1u,      // more in-list parents than LPN: rerun in a bigger class
    SP_ERR_CAPACITY = 2u,   // frontier needs more than CAP slots: rerun in a bigger class / ECAPACITY
    SP_ERR_DUPLICATE = 4u,  // duplicate node in a mapping list
    SP_ERR_DEGREE = 8u      // node degree above the supported maximum
};

template <int CAP> struct HashSize {
    static constexpr int LOG2 = CAP <= 64 ? 8 : (CAP <= 128 ? 9 : 10);
    static constexpr int N = 1 << LOG2;
};

// A Col never holds more than CAP keys (no speculative inserts): half load is enough, and the
// smaller table keeps the <64> kernels at 15-16 waves per CU.
template <int CAP> struct ColHashSize {
    static constexpr int LOG2 = CAP <= 64 ? 7 : (CAP <= 128 ? 8 : 10);
    static constexpr int N = 1 << LOG2;
};

template <int CAP> struct Col {
    double m[CAP], i[CAP], d[CAP];
    uint32_t id[CAP];
    uint32_t hkey[ColHashSize<CAP>::N];
    uint16_t hslot[ColHashSize<CAP>::N];
    int n;   // stored entries (the reference's nodevec elements)
    int na;  // entries that carry m/i (the active list handed to f_step / b_step)
    int E;   // column exponent: true value = stored * 2^E
};

template <int CAP> __device__ __forceinline__ uint32_t hash_of(uint32_t id) {
    return (id * 2654435761u) >> (32 - HashSize<CAP>::LOG2);
}
template <int CAP> __device__ __forceinline__ uint32_t col_hash_of(uint32_t id) {
    return (id * 2654435761u) >> (32 - ColHashSize<CAP>::LOG2);
}
template <int CAP> __device__ __forceinline__ void hash_clear(Col<CAP> &c) {
    for (int h = threadIdx.x; h < ColHashSize<CAP>::N; h += 64) c.hkey[h] = H_EMPTY;
}
// returns false when the id is already present (slot is left unchanged)
template <int CAP> __device__ __forceinline__ bool hash_insert(Col<CAP> &c, uint32_t id, int slot) {
    uint32_t h = col_hash_of<CAP>(id);
    for (;;) {
        const uint32_t old = atomicCAS(&c.hkey[h], H_EMPTY, id);
        if (old == H_EMPTY) {
            c.hslot[h] = (uint16_t)slot;
            return true;
        }
        if (old == id) return false;
        h = (h + 1) & (ColHashSize<CAP>::N - 1);
    }
}
template <int CAP> __device__ __forceinline__ int hash_find(const Col<CAP> &c, uint32_t id) {
    uint32_t h = col_hash_of<CAP>(id);
    for (;;) {
        const uint32_t k = c.hkey[h];
        if (k == id) return (int)c.hslot[h];
        if (k == H_EMPTY) return -1;
        h = (h + 1) & (ColHashSize<CAP>::N - 1);
    }
}

// Cross-lane reductions of a FULL wave64 (every caller is a 64-thread block with all lanes active) on the DPP
// path: row_shr 1, 2, 4, 8 gives the inclusive scan inside each row of 16 lanes, row_bcast15 / row_bcast31
// carry it across the rows, lane 63 then holds the total.  Six VALU steps instead of six dependent
// ds_bpermute round trips (~100+ cycles each); the frontier kernels do several of these per read position.
// Lanes a DPP step cannot read (bound_ctrl = 0) take `ident`.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_i(int ident, int v) {
    return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_d(double ident, double v) {
    const long long iv = __double_as_longlong(v), ii = __double_as_longlong(ident);
    const int lo = dpp_i<CTRL, ROW_MASK>((int)(ii & 0xffffffffll), (int)(iv & 0xffffffffll));
    const int hi = dpp_i<CTRL, ROW_MASK>((int)(ii >> 32), (int)(iv >> 32));
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}
static constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118,
                     DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
__device__ __forceinline__ double wave_bcast63(double v) {
    const long long iv = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(iv & 0xffffffffll), 63);
    const int hi = __builtin_amdgcn_readlane((int)(iv >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}
// inclusive scans (sum); the reductions are their last lane
__device__ __forceinline__ double wave_scan_sum(double v) {
    v += dpp_d<DPP_ROW_SHR1, 0xf>(0.0, v);
    v += dpp_d<DPP_ROW_SHR2, 0xf>(0.0, v);
    v += dpp_d<DPP_ROW_SHR4, 0xf>(0.0, v);
    v += dpp_d<DPP_ROW_SHR8, 0xf>(0.0, v);
    v += dpp_d<DPP_ROW_BCAST15, 0xa>(0.0, v);
    v += dpp_d<DPP_ROW_BCAST31, 0xc>(0.0, v);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) { return wave_bcast63(wave_scan_sum(v)); }
// maximum of NON-NEGATIVE values (identity 0): what every caller has (probabilities, totals)
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_d<DPP_ROW_SHR1, 0xf>(0.0, v));
    v = fmax(v, dpp_d<DPP_ROW_SHR2, 0xf>(0.0, v));
    v = fmax(v, dpp_d<DPP_ROW_SHR4, 0xf>(0.0, v));
    v = fmax(v, dpp_d<DPP_ROW_SHR8, 0xf>(0.0, v));
    v = fmax(v, dpp_d<DPP_ROW_BCAST15, 0xa>(0.0, v));
    v = fmax(v, dpp_d<DPP_ROW_BCAST31, 0xc>(0.0, v));
    return wave_bcast63(v);
}
// maximum of unsigned integers over the 64 lanes (one VALU op per step: the DPP move folds into v_max_u32)
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
    v = max(v, (uint32_t)dpp_i<DPP_ROW_SHR1, 0xf>(0, (int)v));
    v = max(v, (uint32_t)dpp_i<DPP_ROW_SHR2, 0xf>(0, (int)v));
    v = max(v, (uint32_t)dpp_i<DPP_ROW_SHR4, 0xf>(0, (int)v));
    v = max(v, (uint32_t)dpp_i<DPP_ROW_SHR8, 0xf>(0, (int)v));
    v = max(v, (uint32_t)dpp_i<DPP_ROW_BCAST15, 0xa>(0, (int)v));
    v = max(v, (uint32_t)dpp_i<DPP_ROW_BCAST31, 0xc>(0, (int)v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// the same maximum as wave_max (NON-NEGATIVE doubles order like their bit patterns) in two integer passes: the high
// words, then the low words of the lanes that hold the largest high word -- half the instructions of the f64 form
__device__ __forceinline__ double wave_max_pos(double v) {
    const uint32_t hi = (uint32_t)__double2hiint(v), lo = (uint32_t)__double2loint(v);
    const uint32_t mh = wave_umax(hi);
    const uint32_t ml = wave_umax(hi == mh ? lo : 0u);
    return __hiloint2double((int)mh, (int)ml);
}
// binary exponent e with max * 2^-e in [0.5, 1) of the largest of NON-NEGATIVE doubles given by their high words
// (sp_exp_of of the maximum); false when the maximum is zero or denormal (the caller takes the exact route)
__device__ __forceinline__ bool wave_exp_of_max_hi(uint32_t hi, int &e) {
    const uint32_t mh = wave_umax(hi);
    const int be = (int)((mh >> 20) & 0x7ffu);
    e = be - 1022;
    return be != 0;
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ int wave_iscan(int v) {
    v += dpp_i<DPP_ROW_SHR1, 0xf>(0, v);
    v += dpp_i<DPP_ROW_SHR2, 0xf>(0, v);
    v += dpp_i<DPP_ROW_SHR4, 0xf>(0, v);
    v += dpp_i<DPP_ROW_SHR8, 0xf>(0, v);
    v += dpp_i<DPP_ROW_BCAST15, 0xa>(0, v);
    v += dpp_i<DPP_ROW_BCAST31, 0xc>(0, v);
    return v;
}
__device__ __forceinline__ int wave_isum(int v) { return __builtin_amdgcn_readlane(wave_iscan(v), 63); }

// Ordering point between the LDS accesses of the ONE wave of a 64-thread block (all frontier kernels).  The LDS
// executes a wave's instructions in issue order, so lanes see each other's earlier writes without a barrier;
// the compiler must keep the order, that is all.  __syncthreads() adds s_waitcnt vmcnt(0) on top: every
// outstanding global load / store of the wave would be waited for at each sync point of a read position.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int sp_exp_of(double v) {  // v*2^-e in [0.5,1); 0 for v == 0
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const int be = (int)((bits >> 52) & 0x7ff);
    if (bits == 0ull) return 0;
    if (be == 0) return -1022;
    return be - 1022;
}
__device__ __forceinline__ double sp_pow2(int e) {
    return __longlong_as_double((long long)(e + 1023) << 52);
}

// What a sparse kernel needs to know about the model (all linear domain).
struct SparseModel {
    int N;
    const uint8_t *emis;
    const double *init;  // [N] (per candidate: offset applied by the caller)
    const uint32_t *par_off, *par_node, *par_edge;
    const uint32_t *chi_off, *chi_node, *chi_edge;
    const double *trans;  // [E] by edge id (per candidate: offset applied by the caller)
    const double *par_w, *chi_w;  // [E] linear trans prob aligned with par_node / chi_node (the model's own)
    const FwdAdj *fadj;           // [N] packed per-node records (the model's own init / trans)
    const BwdAdj *badj;
    const ParRec *prec;           // [N] parents + edge ids (hinted forward)
    LinParams lp;
    const double *logib;  // forward InsBegin chain (log), [>= max read length]
    int packed;           // every node has at most ADJ_DEG parents and children: the packed records are complete, and the
                          // generic vector kernels read ONE record per node instead of the CSR's offsets -> ids -> weights
                          // chains (two or three dependent global loads per node and Del level: their whole time on a
                          // wide frontier)
};

// Rescale a freshly computed column so that its maximum lies in [0.5, 1).
template <int CAP> __device__ __forceinline__ void col_rescale(Col<CAP> &c, int E_in, double extra_max) {
    double mx = extra_max;
    for (int j = threadIdx.x; j < c.n; j += 64) mx = fmax(mx, fmax(fmax(c.m[j], c.i[j]), c.d[j]));
    mx = wave_max(mx);
    const int e = sp_exp_of(mx);
    const double s = sp_pow2(-e);
    for (int j = threadIdx.x; j < c.n; j += 64) {
        c.m[j] *= s;
        c.i[j] *= s;
        c.d[j] *= s;
    }
    if (threadIdx.x == 0) c.E = E_in + e;
}

// ---------------------------------------------------------------------------------
// One forward column over a GIVEN node list (non-adaptive f_step: forward.rs:276-306 with
// is_adaptive = false, i.e. forward_with_mapping*, forward.rs:51-89):
//   fm (337-359), fi (378-388), fib (541-545), fd = fd0 + G x fdt restricted to the list
//   (423-524), values of nodes outside the list read as 0 (SparseVec default).
// prev: the previous column (cur.n == 0 and first == true for f_init).
// lnk_slot/lnk_w: LDS scratch [CAP*LPN]; dA/dB: LDS scratch [CAP].
// Returns error bits (wave-uniform).
template <int CAP, int LPN>
__device__ uint32_t fwd_list_step(const SparseModel &M, const Col<CAP> &prev, Col<CAP> &cur, const uint32_t *list,
                                  int n, uint8_t x, bool first, int pos, int16_t *lnk_slot, double *lnk_w,
                                  double *dA, double *dB) {
    const LinParams &lp = M.lp;
    uint32_t err = 0;
    hash_clear(cur);
    if (threadIdx.x == 0) {
        cur.n = n;
        cur.na = n;
    }
    wave_sync();
    for (int j = threadIdx.x; j < n; j += 64) {
        const uint32_t k = list[j];
        cur.id[j] = k;
        if (!hash_insert(cur, k, j)) err |= SP_ERR_DUPLICATE;
    }
    wave_sync();
    // InsBegin of the previous column in the previous column's scale (fib, forward.rs:541-545)
    const double ibs = first ? 0.0 : exp(M.logib[pos - 1] - (double)prev.E * SP_LN2);
    const double c_begin = first ? lp.p_MM : lp.p_IM * ibs;  // p_MM*mb' + p_IM*ib'
    // fib of THIS column, same scale:  ib = p_r (p_MI mb' + p_II ib')
    const double ib_cur = first ? lp.p_random * lp.p_MI : lp.p_random * lp.p_II * ibs;
    const double c_del = lp.p_ID * ib_cur;  // fd0 from_begin: p_MD*mb + p_ID*ib with mb = 0
    for (int j = threadIdx.x; j < n; j += 64) {
        const uint32_t k = cur.id[j];
        const double pe = M.emis[k] == x ? lp.p_match : lp.p_mismatch;
        double acc = 0.0;
        int nl = 0;
        const uint32_t a0 = M.par_off[k], a1 = M.par_off[k + 1];
        for (uint32_t a = a0; a < a1; a++) {
            const uint32_t l = M.par_node[a];
            const double w = M.trans[M.par_edge[a]];
            if (w == 0.0) continue;
            if (!first) {
                const int ps = hash_find(prev, l);
                if (ps >= 0) acc += w * (lp.p_MM * prev.m[ps] + lp.p_IM * prev.i[ps] + lp.p_DM * prev.d[ps]);
            }
            const int cs = hash_find(cur, l);
            if (cs >= 0) {
                if (nl < LPN) {
                    lnk_slot[j * LPN + nl] = (int16_t)cs;
                    lnk_w[j * LPN + nl] = w;
                    nl++;
                } else err |= SP_ERR_LINKS;
            }
        }
        for (int q = nl; q < LPN; q++) lnk_slot[j * LPN + q] = -1;
        const double in = M.init[k];
        double inew = 0.0;
        if (!first) {
            const int os = hash_find(prev, k);
            if (os >= 0) inew = lp.p_random * (lp.p_MI * prev.m[os] + lp.p_II * prev.i[os] + lp.p_DI * prev.d[os]);
        }
        const double mnew = pe * (acc + in * c_begin);
        cur.m[j] = mnew;
        cur.i[j] = inew;
        dA[j] = lp.p_MD * mnew + lp.p_ID * inew;  // g
    }
    wave_sync();
    // fd0 (forward.rs:480-501) then n_max_gaps x fdt (510-524), restricted to the list
    double *src = dA, *dst = dB;
    for (int t = 0; t <= lp.n_max_gaps; t++) {
        for (int j = threadIdx.x; j < n; j += 64) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < LPN; q++) {
                const int cs = lnk_slot[j * LPN + q];
                if (cs >= 0) s += lnk_w[j * LPN + q] * src[cs];
            }
            if (t == 0) {
                s += M.init[cur.id[j]] * c_del;
                cur.d[j] = s;
            } else {
                s *= lp.p_DD;
                cur.d[j] += s;
            }
            dst[j] = s;
        }
        wave_sync();
        double *tmp = src;
        src = dst;
        dst = tmp;
    }
    col_rescale(cur, first ? 0 : prev.E, ib_cur);
    wave_sync();
    return err;
}

// ---------------------------------------------------------------------------------
// One backward column over a GIVEN node list (non-adaptive b_step, backward.rs:216-261 with
// is_adaptive = false: backward_with_mapping / backward_by_forward, backward.rs:59-142):
//   bd = bd0 + G x bdt restricted to the list (299-404), bm (423-444), bi (462-483);
//   the begin states bmb/bib are not needed by the mapping flow and are left out.
// prev: the column of position pos+1; prev_is_init: it is b_init (m = i = d = p_end for every
// node, backward.rs:197-211).  list may live in LDS or global memory.
template <int CAP>
__device__ void bwd_list_step(const SparseModel &M, const Col<CAP> &prev, bool prev_is_init, Col<CAP> &cur,
                              const uint32_t *list, int n, uint8_t x, double *dA, double *dB) {
    // in-list child links cached in LDS by the bd0 pass (small classes only): the n_max_gaps Del
    // sweeps and the bm/bi pass then need no global (CSR) access at all
    constexpr bool LINKS = CAP <= 128;
    constexpr int LPN = 2;
    __shared__ int16_t lk_slot[LINKS ? CAP * LPN : 1];
    __shared__ double lk_w[LINKS ? CAP * LPN : 1];
    __shared__ int lk_overflow;
    if (LINKS && threadIdx.x == 0) lk_overflow = 0;
    const LinParams &lp = M.lp;
    hash_clear(cur);
    if (threadIdx.x == 0) {
        cur.n = n;
        cur.na = n;
    }
    wave_sync();
    for (int j = threadIdx.x; j < n; j += 64) {
        const uint32_t k = list[j];
        cur.id[j] = k;
        hash_insert(cur, k, j);
    }
    wave_sync();
    const double pend = lp.p_end;
    // bd0 (backward.rs:354-377); keep A1 = sum_w t e_w m'[w] and q0 = p_r i'[v] for bm/bi
    for (int j = threadIdx.x; j < n; j += 64) {
        const uint32_t v = cur.id[j];
        double a1 = 0.0;
        int nl = 0;
        for (uint32_t a = M.chi_off[v]; a < M.chi_off[v + 1]; a++) {
            const double w = M.chi_w[a];
            if (w == 0.0) continue;
            const uint32_t u = M.chi_node[a];
            double mu = 0.0;
            if (prev_is_init) mu = pend;
            else {
                const int ps = hash_find(prev, u);
                if (ps >= 0) mu = prev.m[ps];
            }
            a1 += w * (M.emis[u] == x ? lp.p_match : lp.p_mismatch) * mu;
            if (LINKS) {
                const int cs = hash_find(cur, u);
                if (cs >= 0) {
                    if (nl < LPN) {
                        lk_slot[j * LPN + nl] = (int16_t)cs;
                        lk_w[j * LPN + nl] = w;
                        nl++;
                    } else lk_overflow = 1;
                }
            }
        }
        if (LINKS)
            for (int q = nl; q < LPN; q++) lk_slot[j * LPN + q] = -1;
        double iv = 0.0;
        if (prev_is_init) iv = pend;
        else {
            const int os = hash_find(prev, v);
            if (os >= 0) iv = prev.i[os];
        }
        const double q0 = lp.p_random * iv;
        const double d0 = lp.p_DM * a1 + lp.p_DI * q0;
        cur.m[j] = a1;  // stash
        cur.i[j] = q0;  // stash
        cur.d[j] = d0;
        dA[j] = d0;
    }
    wave_sync();
    // bdt (backward.rs:387-404), restricted to the list
    double *src = dA, *dst = dB;
    for (int t = 1; t <= lp.n_max_gaps; t++) {
        for (int j = threadIdx.x; j < n; j += 64) {
            const uint32_t v = cur.id[j];
            double s = 0.0;
            if (LINKS && !lk_overflow) {
#pragma unroll
                for (int q = 0; q < LPN; q++) {
                    const int cs = lk_slot[j * LPN + q];
                    if (cs >= 0) s += lk_w[j * LPN + q] * src[cs];
                }
            } else {
                for (uint32_t a = M.chi_off[v]; a < M.chi_off[v + 1]; a++) {
                    const double w = M.chi_w[a];
                    if (w == 0.0) continue;
                    const int cs = hash_find(cur, M.chi_node[a]);
                    if (cs >= 0) s += w * src[cs];
                }
            }
            s *= lp.p_DD;
            dst[j] = s;
            cur.d[j] += s;
        }
        wave_sync();
        double *tmp = src;
        src = dst;
        dst = tmp;
    }
    // bm, bi: sum_w t (p_XM e_w m'[w] + p_XD d[w]) + p_XI p_r i'[v]
    for (int j = threadIdx.x; j < n; j += 64) {
        const uint32_t v = cur.id[j];
        double td = 0.0;
        if (LINKS && !lk_overflow) {
#pragma unroll
            for (int q = 0; q < LPN; q++) {
                const int cs = lk_slot[j * LPN + q];
                if (cs >= 0) td += lk_w[j * LPN + q] * cur.d[cs];
            }
        } else {
            for (uint32_t a = M.chi_off[v]; a < M.chi_off[v + 1]; a++) {
                const double w = M.chi_w[a];
                if (w == 0.0) continue;
                const int cs = hash_find(cur, M.chi_node[a]);
                if (cs >= 0) td += w * cur.d[cs];
            }
        }
        dA[j] = td;
    }
    wave_sync();
    for (int j = threadIdx.x; j < n; j += 64) {
        const double a1 = cur.m[j], q0 = cur.i[j], td = dA[j];
        cur.m[j] = lp.p_MM * a1 + lp.p_MD * td + lp.p_MI * q0;
        cur.i[j] = lp.p_IM * a1 + lp.p_ID * td + lp.p_II * q0;
    }
    wave_sync();
    col_rescale(cur, prev_is_init ? 0 : prev.E, 0.0);
    wave_sync();
}

// fe (forward.rs:554-558): ln(p_end * sum over the active list of m+i+d) + E ln2
template <int CAP> __device__ __forceinline__ double col_log_end(const SparseModel &M, const Col<CAP> &c) {
    double s = 0.0;
    for (int j = threadIdx.x; j < c.na; j += 64) s += c.m[j] + c.i[j] + c.d[j];
    s = wave_sum(s);
    return log(M.lp.p_end * s) + (double)c.E * SP_LN2;
}

}  // namespace phmm
