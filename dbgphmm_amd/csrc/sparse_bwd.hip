// PHMMModel::backward_sparse (src/hmmv2/backward.rs:146-185) and to_full_prob_sparse_backward
// (freq.rs:153-163): the backward recursion with its OWN frontier -- dense b_step for the last
// n_warmup positions of the read, then per position
//     active = B.tables[i+1].top_nodes(n_active_nodes)                     table.rs:127-131
//     b_step(i, x_i, B.tables[i+1], active, is_dense = false, is_adaptive = true)  backward.rs:216-261
// where the adaptive b_step computes Del on growing sets S_0 = to_parents_and_us(active),
// S_t = to_parents_and_us(S_{t-1}) (backward.rs:299-343; active_nodes.rs:38-56) and Match / Ins and the
// Begin states on S_0.
//
// GPU shape: the dense tail is the ordinary dense backward kernel (dense.hip) run on the reads' suffixes
// with no forward tables; the rest is one wave64 per read on the 400-slot LDS frontier of frontier_dev.h.
// Every S_t is a prefix of the column's insertion-ordered vector (the "and_us" part keeps the earlier
// elements in front), which is also the element order of the reference's `to_nodevec()`.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "sparse_dyn.h"
#include "sparse_fwd_kernel.h"

namespace phmm {

struct SparseBwdAdArgs {
    SparseModel M;
    int W, N, Lc, Lb, bcols;
    const double *Bm, *Bi, *Bd;       // [ng][bcols][N][W]: column 0 of the suffix run sits in slot 0
    const int *BE;                    // [ng][Lc+1][W]
    const unsigned long long *cmaxB;  // [ng][Lc][W]
    const double *logibB;             // [ng][Lc+1][W]
    const int *len;                   // [lanes] full read lengths
    const int *wr;                    // [lanes] dense tail columns of each read
    const uint8_t *bases;             // [ng][Lb][W] full reads
    const uint32_t *cand_node;        // top_nodes of the dense column, unsorted
    const double *cand_tot;
    const int *cand_n;
    const uint32_t *lanes;
    int topk;
    double *out_logp;  // [lanes]
    uint32_t *err;     // [lanes]
    RecPool pool;      // base == nullptr: columns are not kept
    const uint64_t *lane_pos0;
    double *scal;      // [positions][2] ln mb, ln ib of every kept column
};

__device__ __forceinline__ double sb_logadd(double x, double y) {
    const double hi = x >= y ? x : y, lo = x >= y ? y : x;
    if (lo == -INFINITY) return hi;
    return hi + log1p(exp(lo - hi));
}

// One adaptive backward column.  On entry cur holds the top nodes of the previous column in sorted order
// (cur.n == ntop, hash filled, m/i/d zero).  s1 / s2: the node sums of bmb / bib (backward.rs:499-555) in the
// previous column's scale.
template <int CAP>
__device__ void bwd_adaptive_step(const SparseModel &M, const PrevRef<CAP> &prev, FVec<CAP> &cur, FScratch<CAP> &sc,
                                  uint8_t x, double &s1, double &s2) {
    const LinParams &lp = M.lp;
    const int lane = threadIdx.x;
    const int ntop = cur.n;
    // S_0 = to_parents_and_us(active)  (backward.rs:246; active_nodes.rs:48-56)
    for (int j = lane; j < ntop; j += 64) sc.order[j] = (uint16_t)j;
    wave_sync();
    append_neighbours<CAP>(M, false, cur, sc, sc.order, ntop, nullptr, nullptr, nullptr, 0);
    const int na = cur.n;
    if (lane == 0) cur.na = na;
    wave_sync();
    // bd0 on S_0 (backward.rs:354-377)
    for (int j = lane; j < na; j += 64) {
        const uint32_t k = cur.id[j];
        double acc = 0.0;
        for (uint32_t a = M.chi_off[k]; a < M.chi_off[k + 1]; a++) {
            const double w = M.chi_w[a];
            if (w == 0.0) continue;
            const uint32_t u = M.chi_node[a];
            double pm, pi, pd;
            prev_get(prev, u, pm, pi, pd);
            acc += w * (M.emis[u] == x ? lp.p_match : lp.p_mismatch) * pm;
        }
        double om, oi, od;
        prev_get(prev, k, om, oi, od);
        const double v = lp.p_DM * acc + lp.p_DI * lp.p_random * oi;
        sc.tot[j] = v;
        cur.d[j] = v;
    }
    wave_sync();
    // bdt on S_t = to_parents_and_us(S_{t-1}) (backward.rs:322-340, 387-404): level values of t-1 exist on the
    // first nprev slots only
    int nprev = na;
    for (int t = 1; t <= lp.n_max_gaps; t++) {
        for (int j = lane; j < nprev; j += 64) sc.order[j] = (uint16_t)j;
        wave_sync();
        append_neighbours<CAP>(M, false, cur, sc, sc.order, nprev, nullptr, nullptr, nullptr, 0);
        const int nt = cur.n;
        const double *lv_prev = (t & 1) ? sc.tot : sc.lvb;
        double *lv_cur = (t & 1) ? sc.lvb : sc.tot;
        for (int j = lane; j < nt; j += 64) {
            const uint32_t k = cur.id[j];
            double acc = 0.0;
            for (uint32_t a = M.chi_off[k]; a < M.chi_off[k + 1]; a++) {
                const double w = M.chi_w[a];
                if (w == 0.0) continue;
                const int s = fv_find(cur, M.chi_node[a]);
                if (s >= 0 && s < nprev) acc += w * lv_prev[s];
            }
            const double v = lp.p_DD * acc;
            lv_cur[j] = v;
            cur.d[j] += v;
        }
        wave_sync();
        nprev = nt;
    }
    // bm, bi on S_0 with the finished Del column (backward.rs:423-483); bmb, bib (499-555)
    double a1 = 0.0, a2 = 0.0;
    for (int j = lane; j < na; j += 64) {
        const uint32_t k = cur.id[j];
        double accm = 0.0, acci = 0.0;
        for (uint32_t a = M.chi_off[k]; a < M.chi_off[k + 1]; a++) {
            const double w = M.chi_w[a];
            if (w == 0.0) continue;
            const uint32_t u = M.chi_node[a];
            double pm, pi, pd;
            prev_get(prev, u, pm, pi, pd);
            const double em = (M.emis[u] == x ? lp.p_match : lp.p_mismatch) * pm;
            const int s = fv_find(cur, u);
            const double du = s >= 0 ? cur.d[s] : 0.0;
            accm += w * (lp.p_MM * em + lp.p_MD * du);
            acci += w * (lp.p_IM * em + lp.p_ID * du);
        }
        double om, oi, od;
        prev_get(prev, k, om, oi, od);
        cur.m[j] = accm + lp.p_MI * lp.p_random * oi;
        cur.i[j] = acci + lp.p_II * lp.p_random * oi;
        const double ek = (M.emis[k] == x ? lp.p_match : lp.p_mismatch) * om;
        const double in = M.init[k], dk = cur.d[j];
        a1 += in * (lp.p_MM * ek + lp.p_MD * dk);
        a2 += in * (lp.p_IM * ek + lp.p_ID * dk);
    }
    s1 = wave_sum(a1);
    s2 = wave_sum(a2);
    wave_sync();
    // rescale so that the column maximum is in [0.5, 1)
    double mx = 0.0;
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

template <int CAP>
__global__ void __launch_bounds__(64) sparse_backward_adaptive_kernel(const SparseBwdAdArgs a) {
    __shared__ FVec<CAP> cols[2];
    __shared__ FScratch<CAP> sc;
    const int lane = threadIdx.x;
    const uint32_t gi = a.lanes[blockIdx.x];
    const int g = (int)(gi / a.W), r = (int)(gi % a.W);
    const int len = a.len[gi], wr = a.wr[gi];
    const size_t NW = (size_t)a.N * a.W;
    const LinParams &lp = a.M.lp;
    const uint64_t p0 = a.lane_pos0 ? a.lane_pos0[gi] : 0;
    if (lane == 0) sc.dropped = 0;
    uint32_t err = 0;
    int pos = len - wr - 1;  // first sparse position (the host only sends reads with len > wr)
    // top list of the dense column: candidates sorted by (total desc, node asc) = the reference's stable
    // sort over the dense nodevec
    {
        const int nc = a.cand_n[gi];
        FVec<CAP> &c0 = cols[pos & 1];
        fv_clear(c0);
        wave_sync();
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
    }
    PrevRef<CAP> pr{};
    {
        const double cm = __longlong_as_double((long long)a.cmaxB[((size_t)g * a.Lc + 0) * a.W + r]);
        const int e = sp_exp_of(cm);
        pr.vec = nullptr;
        pr.gm = a.Bm + (size_t)g * a.bcols * NW;
        pr.gi = a.Bi + (size_t)g * a.bcols * NW;
        pr.gd = a.Bd + (size_t)g * a.bcols * NW;
        pr.W = a.W;
        pr.lane = r;
        pr.sc = sp_pow2(-e);
        pr.E = a.BE[((size_t)g * (a.Lc + 1) + 0) * a.W + r] + e;
        pr.is_init = false;
    }
    double ibl = a.logibB[((size_t)g * (a.Lc + 1) + 0) * a.W + r];
    double mbl = -INFINITY;
    const double l_mi = log(lp.p_MI * lp.p_random), l_ii = log(lp.p_II * lp.p_random);
    bool first = true;
    for (; pos >= 0; pos--) {
        FVec<CAP> &cur = cols[pos & 1];
        if (!first) {
            const FVec<CAP> &prev = cols[(pos + 1) & 1];
            select_top<CAP>(prev, cur, sc, false, 0.0, a.topk);
            pr.vec = &prev;
            pr.E = prev.E;
        }
        first = false;
        double s1, s2;
        bwd_adaptive_step<CAP>(a.M, pr, cur, sc, a.bases[((size_t)g * a.Lb + pos) * a.W + r], s1, s2);
        const double El = (double)pr.E * SP_LN2;
        mbl = sb_logadd(log(s1) + El, l_mi + ibl);
        ibl = sb_logadd(log(s2) + El, l_ii + ibl);
        if (a.pool.base) {
            if (!store_record<CAP>(a.pool, p0 + (uint64_t)pos, cur)) {
                err |= SP_ERR_POOL;
                break;
            }
            if (lane == 0) {
                a.scal[(p0 + (uint64_t)pos) * 2 + 0] = mbl;
                a.scal[(p0 + (uint64_t)pos) * 2 + 1] = ibl;
            }
        }
    }
    if (lane == 0) {
        a.out_logp[gi] = err ? NAN : mbl;
        a.err[gi] = err;
    }
}

// ------------------------------------------------------------------ run_sparse posteriors
// PHMMOutput::to_node_freqs (freq.rs:245-255) of run_sparse (freq.rs:51-55): state_probs = sum over merged
// indices j of F.merged(j) (.) B.merged(j) / P (table.rs:414-434, 500-517), where the product of two tables keeps
// the stored elements of the sparse operand (`self` first) and is dense only when both are (table.rs:320-331).
// F.merged(j) = F.tables[j-1] is dense for j-1 < n_warmup, B.merged(j) = B.tables[j] for j >= len - n_warmup
// (and b_init at j = len).
struct CombineArgs {
    int W, N, Lc, bcols, nw;
    const double *Fm, *Fi, *Fd;  // [ng][Lc][N][W] dense forward columns of the prefix run
    const int *FE;               // [ng][Lc+1][W]
    const double *Bm, *Bi, *Bd;  // [ng][bcols][N][W] dense backward columns of the suffix run
    const int *BE;
    RecPool fpool, bpool;        // sparse columns, by lane_pos0[lane] + position
    const uint64_t *lane_pos0;
    const int *len, *wr;
    const double *logP;          // [lanes] forward ln P(read)
    double p_end;
    double *freq;                // [N]
    const uint32_t *lanes;
};
struct RecView {
    int n, na, E;
    const uint32_t *ids;
    const double *m, *i, *d;
};
__device__ __forceinline__ bool rec_view(const RecPool &p, uint64_t idx, RecView &v) {
    const uint64_t o1 = p.off[idx];
    if (o1 == 0) return false;
    const uint8_t *rec = p.base + (o1 - 8);
    const int *hw = (const int *)rec;
    v.n = hw[0];
    v.na = hw[1];
    v.E = hw[2];
    const uint64_t idb = (uint64_t)((v.n + 1) & ~1) * 4;
    v.ids = (const uint32_t *)(rec + 16);
    v.m = (const double *)(rec + 16 + idb);
    v.i = v.m + v.na;
    v.d = v.i + v.na;
    return true;
}

__global__ void __launch_bounds__(64) run_sparse_combine(const CombineArgs a) {
    __shared__ uint32_t bid[PHMM_MAX_ACTIVE_NODES];
    const int lane = threadIdx.x;
    const uint32_t gi = a.lanes[blockIdx.x];
    const int g = (int)(gi / a.W), r = (int)(gi % a.W);
    const int len = a.len[gi], wr = a.wr[gi];
    const double logP = a.logP[gi];
    if (!(logP > -INFINITY)) return;
    const uint64_t p0 = a.lane_pos0[gi];
    for (int j = 1; j <= len; j++) {
        const bool fdense = (j - 1) < a.nw;
        const bool bdense = j >= len - wr;
        if (fdense && bdense) continue;  // run_sparse_dense_pair
        if (!fdense && !bdense) {
            RecView F, B;
            if (!rec_view(a.fpool, p0 + (uint64_t)(j - 1), F) || !rec_view(a.bpool, p0 + (uint64_t)j, B)) continue;
            wave_sync();
            for (int q = lane; q < B.n; q += 64) bid[q] = B.ids[q];
            wave_sync();
            const double w = exp((double)(F.E + B.E) * SP_LN2 - logP);
            for (int q = lane; q < F.n; q += 64) {
                const uint32_t id = F.ids[q];
                int sidx = -1;
                for (int t = 0; t < B.n; t++)
                    if (bid[t] == id) {
                        sidx = t;
                        break;
                    }
                if (sidx < 0) continue;
                double v = F.d[q] * B.d[sidx];
                if (q < F.na && sidx < B.na) v += F.m[q] * B.m[sidx] + F.i[q] * B.i[sidx];
                if (v != 0.0) atomicAdd(&a.freq[id], w * v);
            }
        } else if (fdense) {
            RecView B;
            if (!rec_view(a.bpool, p0 + (uint64_t)j, B)) continue;
            const int col = j - 1;
            const double w = exp((double)(a.FE[((size_t)g * (a.Lc + 1) + col) * a.W + r] + B.E) * SP_LN2 - logP);
            for (int q = lane; q < B.n; q += 64) {
                const uint32_t id = B.ids[q];
                const size_t ix = (((size_t)g * a.Lc + col) * a.N + id) * a.W + r;
                double v = a.Fd[ix] * B.d[q];
                if (q < B.na) v += a.Fm[ix] * B.m[q] + a.Fi[ix] * B.i[q];
                if (v != 0.0) atomicAdd(&a.freq[id], w * v);
            }
        } else {
            RecView F;
            if (!rec_view(a.fpool, p0 + (uint64_t)(j - 1), F)) continue;
            const bool init = j == len;  // b_init: m = i = d = p_end (backward.rs:197-211)
            const int jb = j - (len - wr);
            const int EB = init ? 0 : a.BE[((size_t)g * (a.Lc + 1) + jb) * a.W + r];
            const double w = exp((double)(F.E + EB) * SP_LN2 - logP);
            for (int q = lane; q < F.n; q += 64) {
                const uint32_t id = F.ids[q];
                double bm = a.p_end, bi = a.p_end, bd = a.p_end;
                if (!init) {
                    const size_t ix = (((size_t)g * a.bcols + jb) * a.N + id) * a.W + r;
                    bm = a.Bm[ix];
                    bi = a.Bi[ix];
                    bd = a.Bd[ix];
                }
                double v = F.d[q] * bd;
                if (q < F.na) v += F.m[q] * bm + F.i[q] * bi;
                if (v != 0.0) atomicAdd(&a.freq[id], w * v);
            }
        }
    }
}

// merged indices where both tables are dense (reads shorter than 2 n_warmup): tasks = (lane, j)
__global__ void __launch_bounds__(BLOCK) run_sparse_dense_pair(const CombineArgs a, const int2 *tasks) {
    const int2 t = tasks[blockIdx.y];
    const int gi = t.x, j = t.y;
    const int g = gi / a.W, r = gi % a.W;
    const int len = a.len[gi], wr = a.wr[gi];
    const double logP = a.logP[gi];
    if (!(logP > -INFINITY)) return;
    const int col = j - 1;
    const bool init = j == len;
    const int jb = j - (len - wr);
    const int EB = init ? 0 : a.BE[((size_t)g * (a.Lc + 1) + jb) * a.W + r];
    const double w = exp((double)(a.FE[((size_t)g * (a.Lc + 1) + col) * a.W + r] + EB) * SP_LN2 - logP);
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k < a.N; k += gridDim.x * BLOCK) {
        const size_t fx = (((size_t)g * a.Lc + col) * a.N + k) * a.W + r;
        double bm = a.p_end, bi = a.p_end, bd = a.p_end;
        if (!init) {
            const size_t bx = (((size_t)g * a.bcols + jb) * a.N + k) * a.W + r;
            bm = a.Bm[bx];
            bi = a.Bi[bx];
            bd = a.Bd[bx];
        }
        const double v = a.Fm[fx] * bm + a.Fi[fx] * bi + a.Fd[fx] * bd;
        if (v != 0.0) atomicAdd(&a.freq[k], w * v);
    }
}

// ------------------------------------------------------------------ host side
namespace {

struct Scratch {
    size_t sb = 0;
    size_t carve(size_t bytes) {
        sb = (sb + 255) / 256 * 256;
        const size_t o = sb;
        sb += bytes;
        return o;
    }
};

// the first / last min(len, n) bases of every read as a read set of its own
void clip_reads(const uint8_t *bases, const uint64_t *off, uint64_t R, int n, bool tail, phmm_reads &out) {
    out.R = R;
    out.off.assign(R + 1, 0);
    for (uint64_t r = 0; r < R; r++) out.off[r + 1] = out.off[r] + std::min<uint64_t>(off[r + 1] - off[r], (uint64_t)n);
    out.total = out.off[R];
    out.bases.resize(out.total);
    out.max_len = 0;
    for (uint64_t r = 0; r < R; r++) {
        const uint64_t w = out.off[r + 1] - out.off[r];
        std::memcpy(out.bases.data() + out.off[r], tail ? bases + off[r + 1] - w : bases + off[r], w);
        out.max_len = std::max(out.max_len, w);
    }
}

// One chunk of read groups of backward_sparse.
struct BwdChunk {
    // in
    int g0 = 0, ngc = 0, Lc = 0;
    bool full_b = false, keep = false;  // keep every dense column / every sparse column (records)
    // out
    DenseArgs a{};
    SparseBwdAdArgs ba{};
    int Lb = 1;
    std::vector<int> hfull, hwr;
    std::vector<uint32_t> sparse_lanes;
    std::vector<double> logp;  // [lanes] ln P(read) = B.tables[0].mb
    std::vector<uint64_t> lane_pos0;
    uint64_t n_pos = 0;
};
struct BwdBufs {
    DevBuf *tables, *misc;
    DevBuf sel, pool, meta;
    // run_sparse: the backward tables sit behind the forward tables in the SAME grow-only buffer (a second buffer
    // beside a table buffer that an earlier call grew to the whole budget would not fit)
    char *ext_tables = nullptr;
    size_t ext_bytes = 0;
};

void backward_chunk(phmm_model *m, const Plan &plan, const DenseArgs &base, const phmm_reads &suf, const uint8_t *bases,
                    const uint64_t *off, uint64_t R, int K, BwdChunk &c, BwdBufs &b) {
    hipStream_t s = current_stream();
    const int W = plan.W, ngc = c.ngc, Lc = c.Lc, g0 = c.g0;
    const int lanes = ngc * W;
    DenseArgs &a = c.a;
    a = base;
    a.ng = ngc;
    a.Lc = Lc;
    size_t tb = 0, mb = 0;
    layout(a, W, c.full_b, nullptr, nullptr, tb, mb, true);
    if (b.ext_tables) {
        if (tb > b.ext_bytes) PHMM_THROW(PHMM_EINTERNAL, "backward_sparse: table region too small");
    } else {
        b.tables->reserve(tb);
    }
    b.misc->reserve(mb);
    layout(a, W, c.full_b, b.ext_tables ? (void *)b.ext_tables : b.tables->p, b.misc->p, tb, mb, true);
    a.tmaxF = nullptr;
    HIP_CHECK(hipMemsetAsync(b.misc->p, 0, mb, s));
    // staging: suffix bases + lengths for the dense kernel; full reads for the frontier kernel
    std::vector<uint8_t> hb((size_t)ngc * Lc * W, 0xff);
    std::vector<int> hl((size_t)lanes, 0);
    c.hfull.assign((size_t)lanes, 0);
    c.hwr.assign((size_t)lanes, 0);
    c.sparse_lanes.clear();
    c.Lb = 1;
    for (int gi = 0; gi < lanes; gi++) {
        const size_t slot = (size_t)g0 * W + gi;
        if (slot >= R) continue;
        const uint32_t rd = plan.order[slot];
        const int wr = (int)(suf.off[rd + 1] - suf.off[rd]);
        const int full = (int)(off[rd + 1] - off[rd]);
        hl[gi] = wr;
        c.hwr[gi] = wr;
        c.hfull[gi] = full;
        const int g = gi / W, r = gi % W;
        for (int i = 0; i < wr; i++) hb[((size_t)g * Lc + i) * W + r] = suf.bases[suf.off[rd] + i];
        c.Lb = std::max(c.Lb, full);
        if (full > wr) c.sparse_lanes.push_back((uint32_t)gi);
    }
    HIP_CHECK(hipMemcpyAsync((void *)a.bases, hb.data(), hb.size(), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync((void *)a.len, hl.data(), hl.size() * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (int pos = Lc - 1; pos >= 0; pos--) launch_bwd_step(W, a, pos);
    launch_bwd_finish(W, a);
    c.logp.assign((size_t)lanes, 0.0);
    // (logmbB is [ng][Lc+1][W]: row 0 of every group)
    for (int g = 0; g < ngc; g++)
        HIP_CHECK(hipMemcpyAsync(c.logp.data() + (size_t)g * W, a.logmbB + (size_t)g * (Lc + 1) * W, sizeof(double) * W,
                                 hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    trace("backward_sparse: dense tail");
    c.lane_pos0.assign((size_t)lanes + 1, 0);
    for (int gi = 0; gi < lanes; gi++) c.lane_pos0[gi + 1] = c.lane_pos0[gi] + (uint64_t)c.hfull[gi];
    c.n_pos = c.lane_pos0[lanes];
    c.ba = SparseBwdAdArgs{};
    if (c.sparse_lanes.empty()) return;
    const std::vector<uint32_t> &sparse_lanes = c.sparse_lanes;
    const int Lb = c.Lb;
    // ---- device scratch of the frontier pass
    Scratch sc;
    const size_t nsel = std::min<size_t>(sparse_lanes.size(), std::max<size_t>(1, ((size_t)256 << 20) / (12 * (size_t)m->N)));
    const size_t o_bases = sc.carve((size_t)ngc * Lb * W), o_len = sc.carve(sizeof(int) * lanes), o_wr = sc.carve(sizeof(int) * lanes),
                 o_sw = sc.carve(sizeof(int) * lanes), o_tmax = sc.carve(sizeof(unsigned long long) * (size_t)ngc * a.bcols * W),
                 o_lanes = sc.carve(sizeof(uint32_t) * lanes), o_cn = sc.carve(sizeof(uint32_t) * (size_t)lanes * PHMM_MAX_ACTIVE_NODES),
                 o_ct = sc.carve(sizeof(double) * (size_t)lanes * PHMM_MAX_ACTIVE_NODES), o_cc = sc.carve(sizeof(int) * lanes),
                 o_out = sc.carve(sizeof(double) * lanes), o_err = sc.carve(sizeof(uint32_t) * lanes),
                 o_need = sc.carve(sizeof(uint32_t) * nsel), o_sn = sc.carve(sizeof(int) * nsel),
                 o_snode = sc.carve(sizeof(uint32_t) * nsel * m->N), o_stot = sc.carve(sizeof(double) * nsel * m->N);
    b.sel.reserve(sc.sb);
    char *sp = (char *)b.sel.p;
    std::vector<uint8_t> hfb((size_t)ngc * Lb * W, 0xff);
    for (uint32_t gi : sparse_lanes) {
        const uint32_t rd = plan.order[(size_t)g0 * W + gi];
        const int g = (int)gi / W, r = (int)gi % W;
        for (int i = 0; i < c.hfull[gi]; i++) hfb[((size_t)g * Lb + i) * W + r] = bases[off[rd] + i];
    }
    std::vector<int> ones((size_t)lanes, 1);
    std::vector<unsigned long long> big((size_t)ngc * a.bcols * W, 0x7fefffffffffffffull);  // DBL_MAX: bounds any total
    HIP_CHECK(hipMemcpyAsync(sp + o_bases, hfb.data(), hfb.size(), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(sp + o_len, c.hfull.data(), sizeof(int) * lanes, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(sp + o_wr, c.hwr.data(), sizeof(int) * lanes, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(sp + o_sw, ones.data(), sizeof(int) * lanes, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(sp + o_tmax, big.data(), sizeof(unsigned long long) * big.size(), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(sp + o_lanes, sparse_lanes.data(), sizeof(uint32_t) * sparse_lanes.size(), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemsetAsync(sp + o_cc, 0, sizeof(int) * lanes, s));
    // ---- top_nodes(n_active_nodes) of the dense column (table.rs:127-131), a batch of reads at a time
    for (size_t nb0 = 0; nb0 < sparse_lanes.size(); nb0 += nsel) {
        const size_t nn = std::min(nsel, sparse_lanes.size() - nb0);
        HIP_CHECK(hipMemcpyAsync(sp + o_need, sparse_lanes.data() + nb0, sizeof(uint32_t) * nn, hipMemcpyHostToDevice, s));
        Top400Args ta{};
        ta.d = a;
        ta.d.Fm = a.Bm;
        ta.d.Fi = a.Bi;
        ta.d.Fd = a.Bd;
        ta.d.Lc = a.bcols;
        ta.d.tmaxF = (unsigned long long *)(sp + o_tmax);
        ta.W = W;
        ta.sw = (const int *)(sp + o_sw);
        ta.need = (const uint32_t *)(sp + o_need);
        ta.sc_node = (uint32_t *)(sp + o_snode);
        ta.sc_tot = (double *)(sp + o_stot);
        ta.sc_n = (int *)(sp + o_sn);
        ta.cand_node = (uint32_t *)(sp + o_cn);
        ta.cand_tot = (double *)(sp + o_ct);
        ta.cand_n = (int *)(sp + o_cc);
        ta.ratio_lin = 0.0;
        ta.K = K;
        launch_select_top(ta, (unsigned)nn, s);
        HIP_CHECK(hipStreamSynchronize(s));
    }
    SparseBwdAdArgs &ba = c.ba;
    ba.M = sparse_model_of(m);
    ba.W = W;
    ba.N = (int)m->N;
    ba.Lc = Lc;
    ba.Lb = Lb;
    ba.bcols = a.bcols;
    ba.Bm = a.Bm;
    ba.Bi = a.Bi;
    ba.Bd = a.Bd;
    ba.BE = a.BE;
    ba.cmaxB = a.cmaxB;
    ba.logibB = a.logibB;
    ba.len = (const int *)(sp + o_len);
    ba.wr = (const int *)(sp + o_wr);
    ba.bases = (const uint8_t *)(sp + o_bases);
    ba.cand_node = (const uint32_t *)(sp + o_cn);
    ba.cand_tot = (const double *)(sp + o_ct);
    ba.cand_n = (const int *)(sp + o_cc);
    ba.lanes = (const uint32_t *)(sp + o_lanes);
    ba.topk = K;
    ba.out_logp = (double *)(sp + o_out);
    ba.err = (uint32_t *)(sp + o_err);
    const uint64_t n_pos = c.n_pos;
    uint64_t pool_cap = c.keep ? n_pos * 2048 + (1u << 20) : 0;
    std::vector<double> hout((size_t)lanes);
    std::vector<uint32_t> herr((size_t)lanes);
    for (int attempt = 0;; attempt++) {
        if (c.keep) {
            b.pool.reserve(pool_cap);
            const size_t meta = 8 + sizeof(uint64_t) * (n_pos + 1) + sizeof(uint64_t) * ((size_t)lanes + 1) + sizeof(double) * 2 * (n_pos + 1);
            b.meta.reserve(meta);
            HIP_CHECK(hipMemsetAsync(b.meta.p, 0, meta, s));
            ba.pool.base = b.pool.as<uint8_t>();
            ba.pool.cap = pool_cap;
            ba.pool.top = b.meta.as<unsigned long long>();
            ba.pool.off = (uint64_t *)(b.meta.as<char>() + 8);
            uint64_t *d_lp0 = ba.pool.off + (n_pos + 1);
            HIP_CHECK(hipMemcpyAsync(d_lp0, c.lane_pos0.data(), sizeof(uint64_t) * lanes, hipMemcpyHostToDevice, s));
            ba.lane_pos0 = d_lp0;
            ba.scal = (double *)(d_lp0 + lanes + 1);
        }
        hipLaunchKernelGGL((sparse_backward_adaptive_kernel<PHMM_MAX_ACTIVE_NODES>), dim3((unsigned)sparse_lanes.size()), dim3(64),
                           0, s, ba);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(hout.data(), ba.out_logp, sizeof(double) * lanes, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(herr.data(), ba.err, sizeof(uint32_t) * lanes, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        bool pool_full = false;
        for (uint32_t gi : sparse_lanes) pool_full |= (herr[gi] & SP_ERR_POOL) != 0;
        if (!pool_full) break;
        if (attempt >= 4) PHMM_THROW(PHMM_ENOMEM, "backward_sparse: record pool");
        pool_cap *= 4;
    }
    for (uint32_t gi : sparse_lanes) {
        if (herr[gi]) PHMM_THROW(PHMM_EINTERNAL, "backward_sparse: frontier kernel error");
        c.logp[gi] = hout[gi];
    }
    trace("backward_sparse: frontier");
}

void check_sparse_params(const phmm_params &prm) {
    if (prm.n_warmup < 1)
        PHMM_THROW(PHMM_EINVAL, "backward_sparse with n_warmup = 0: the reference panics in last_table() (table.rs:388)");
    if (prm.n_active_nodes < 1) PHMM_THROW(PHMM_EINVAL, "n_active_nodes must be positive");
}

// tabs != nullptr (one read): keep every sparse column and return it as [L][N] natural-log arrays (-inf where
// the reference's SparseVec has no element)
struct BwdSparseTables {
    double *b_m, *b_i, *b_d, *b_scal;
};

void backward_sparse_impl(phmm_model *m, const uint8_t *bases, const uint64_t *off, uint64_t R, double *out_logp,
                          const BwdSparseTables *tabs) {
    const phmm_params &prm = m->params;
    check_sparse_params(prm);
    const int K = (int)std::min<int64_t>(prm.n_active_nodes, PHMM_MAX_ACTIVE_NODES);
    const int nw = (int)std::min<int64_t>(prm.n_warmup, INT32_MAX);
    phmm_reads suf;
    clip_reads(bases, off, R, nw, true, suf);
    Plan plan = make_plan(m, &suf, tabs ? 1 : 0);
    const int W = plan.W;
    const size_t NW = (size_t)m->N * W;
    const uint64_t limit = table_budget(*m->pool);
    DenseArgs base{};
    fill_model_args(base, m);
    base.nblk = plan.nblk;
    base.nblk8 = plan.nblk8;
    base.npt = plan.npt;
    std::vector<double> res(R, 0.0);
    BwdBufs bufs;
    bufs.tables = &m->wset().tables;
    bufs.misc = &m->wset().misc;
    int g0 = 0;
    while (g0 < plan.ng_total) {
        const uint32_t r0 = plan.order[(size_t)g0 * W];
        BwdChunk c;
        c.g0 = g0;
        c.Lc = (int)(suf.off[r0 + 1] - suf.off[r0]);
        const size_t per_group = (size_t)6 * NW * 8;
        c.ngc = (int)std::min<uint64_t>(plan.ng_total - g0, std::max<uint64_t>(1, limit / std::max<size_t>(per_group, 1)));
        c.keep = tabs != nullptr;
        backward_chunk(m, plan, base, suf, bases, off, R, K, c, bufs);
        for (int gi = 0; gi < c.ngc * W; gi++) {
            const size_t slot = (size_t)g0 * W + gi;
            if (slot < R) res[plan.order[slot]] = c.logp[gi];
        }
        if (tabs && !c.sparse_lanes.empty()) {
            // one read: decode the kept columns
            const SparseBwdAdArgs &ba = c.ba;
            const int L = c.hfull[0], N = (int)m->N, wr = c.hwr[0];
            std::vector<uint64_t> hoff((size_t)L);
            unsigned long long used = 0;
            HIP_CHECK(hipMemcpy(&used, ba.pool.top, 8, hipMemcpyDeviceToHost));
            std::vector<uint8_t> hp((size_t)used);
            HIP_CHECK(hipMemcpy(hoff.data(), ba.pool.off, sizeof(uint64_t) * L, hipMemcpyDeviceToHost));
            if (used) HIP_CHECK(hipMemcpy(hp.data(), ba.pool.base, used, hipMemcpyDeviceToHost));
            std::vector<double> hs((size_t)L * 2);
            HIP_CHECK(hipMemcpy(hs.data(), ba.scal, sizeof(double) * 2 * L, hipMemcpyDeviceToHost));
            for (int i = 0; i < L - wr; i++) {
                if (hoff[i] == 0) PHMM_THROW(PHMM_EINTERNAL, "backward_sparse: missing column");
                const uint8_t *rec = hp.data() + (hoff[i] - 8);
                const int n = ((const int *)rec)[0], na = ((const int *)rec)[1], E = ((const int *)rec)[2];
                const uint64_t idb = (uint64_t)((n + 1) & ~1) * 4;
                const uint32_t *ids = (const uint32_t *)(rec + 16);
                const double *vm = (const double *)(rec + 16 + idb), *vi = vm + na, *vd = vi + na;
                const double El = (double)E * SP_LN2;
                for (int j = 0; j < n; j++) {
                    const size_t ix = (size_t)i * N + ids[j];
                    if (tabs->b_d) tabs->b_d[ix] = std::log(vd[j]) + El;
                    if (j < na) {
                        if (tabs->b_m) tabs->b_m[ix] = std::log(vm[j]) + El;
                        if (tabs->b_i) tabs->b_i[ix] = std::log(vi[j]) + El;
                    }
                }
                if (tabs->b_scal) {
                    tabs->b_scal[3 * i + 0] = hs[2 * i + 0];
                    tabs->b_scal[3 * i + 1] = hs[2 * i + 1];
                    tabs->b_scal[3 * i + 2] = -INFINITY;  // be (backward.rs:563-565)
                }
            }
        }
        g0 += c.ngc;
    }
    if (out_logp) std::memcpy(out_logp, res.data(), sizeof(double) * R);
}

void put_any(double *dst, const double *src, size_t n) {
    if (!dst) return;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeDevice) {
        HIP_CHECK(hipMemcpy(dst, src, n * sizeof(double), hipMemcpyHostToDevice));
    } else {
        (void)hipGetLastError();
        std::memcpy(dst, src, n * sizeof(double));
    }
}

}  // namespace

// to_full_prob_sparse_backward (freq.rs:153-163): per-read ln P from backward_sparse, and their sum
void full_prob_sparse_backward(phmm_model *m, const phmm_reads *reads, double *out_logp, double *out_total) {
    std::vector<double> lp(reads->R);
    backward_sparse_impl(m, reads->bases.data(), reads->off.data(), reads->R, lp.data(), nullptr);
    double tot = 0.0;
    for (double v : lp) tot += v;
    put_any(out_logp, lp.data(), lp.size());
    put_any(out_total, &tot, 1);
}

// backward_sparse tables of ONE read (parity tests / `inspect`-style tools): [L][N] natural-log arrays
void backward_sparse_tables(phmm_model *m, const uint8_t *read, uint64_t len, double *b_m, double *b_i, double *b_d,
                            double *b_scal, uint8_t *is_dense) {
    const size_t L = (size_t)len, N = m->N;
    const int64_t nw = m->params.n_warmup;
    const size_t wr = (size_t)std::min<uint64_t>(len, (uint64_t)std::max<int64_t>(nw, 0));
    for (double *t : {b_m, b_i, b_d})
        if (t) std::fill(t, t + L * N, -INFINITY);
    const uint64_t off[2] = {0, len};
    BwdSparseTables tabs{b_m, b_i, b_d, b_scal};
    double lp = 0.0;
    backward_sparse_impl(m, read, off, 1, &lp, &tabs);
    // the dense tail: B.tables[len-wr ..] only depend on the suffix (b_init at its end)
    const size_t o = (L - wr);
    dense_tables(m, read + o, wr, nullptr, nullptr, nullptr, nullptr, b_m ? b_m + o * N : nullptr, b_i ? b_i + o * N : nullptr,
                 b_d ? b_d + o * N : nullptr, b_scal ? b_scal + 3 * o : nullptr);
    if (is_dense)
        for (size_t i = 0; i < L; i++) is_dense[i] = i >= o ? 1 : 0;
}

// PHMMModel::run_sparse (freq.rs:51-55) over a read set: forward_sparse(use_max_ratio = false) and backward_sparse,
// their totals and the summed to_node_freqs of the pair.
void run_sparse(phmm_model *m, const phmm_reads *reads, double *out_lf, double *out_lb, double *out_nf) {
    hipStream_t s = current_stream();
    const phmm_params &prm = m->params;
    check_sparse_params(prm);
    const uint64_t R = reads->R;
    const uint8_t *bases = reads->bases.data();
    const uint64_t *off = reads->off.data();
    const int K = (int)std::min<int64_t>(prm.n_active_nodes, PHMM_MAX_ACTIVE_NODES);
    const int nw = (int)std::min<int64_t>(prm.n_warmup, INT32_MAX);
    ensure_logib(m, reads->max_len + 1);
    phmm_reads pre, suf;
    clip_reads(bases, off, R, nw, false, pre);
    clip_reads(bases, off, R, nw, true, suf);
    Plan plan = make_plan(m, &suf, 0);  // (prefix and suffix lengths are the same: one plan for both)
    const int W = plan.W;
    const size_t NW = (size_t)m->N * W;
    // (half of the table budget: the record pools of both directions and the second table buffer live beside it)
    const uint64_t limit = table_budget(*m->pool) / 2;
    DenseArgs base{};
    fill_model_args(base, m);
    base.nblk = plan.nblk;
    base.nblk8 = plan.nblk8;
    base.npt = plan.npt;
    std::vector<double> lf(R, 0.0), lb(R, 0.0);
    DevBuf freq, bmisc, fsel, fpool, fmeta, cbuf;
    freq.reserve(sizeof(double) * m->N);
    HIP_CHECK(hipMemsetAsync(freq.p, 0, sizeof(double) * m->N, s));
    BwdBufs bufs;
    bufs.tables = nullptr;
    bufs.misc = &bmisc;
    int g0 = 0;
    while (g0 < plan.ng_total) {
        const uint32_t r0 = plan.order[(size_t)g0 * W];
        const int Lc = (int)(suf.off[r0 + 1] - suf.off[r0]);
        const size_t per_group = (size_t)(6 * Lc + 4) * NW * 8;
        const int ngc = (int)std::min<uint64_t>(plan.ng_total - g0, std::max<uint64_t>(1, limit / std::max<size_t>(per_group, 1)));
        const int lanes = ngc * W;
        // ---- forward: dense over the prefixes, every column kept
        DenseArgs a = base;
        a.ng = ngc;
        a.Lc = Lc;
        size_t tb = 0, mb = 0, tbb = 0, mbb = 0;
        layout(a, W, false, nullptr, nullptr, tb, mb);
        {
            DenseArgs ab = a;
            layout(ab, W, true, nullptr, nullptr, tbb, mbb, true);
        }
        tb = (tb + 255) / 256 * 256;
        m->wset().tables.reserve(tb + tbb);
        m->wset().misc.reserve(mb);
        layout(a, W, false, m->wset().tables.p, m->wset().misc.p, tb, mb);
        tb = (tb + 255) / 256 * 256;
        bufs.ext_tables = m->wset().tables.as<char>() + tb;
        bufs.ext_bytes = tbb;
        a.tmaxF = nullptr;
        HIP_CHECK(hipMemsetAsync(m->wset().misc.p, 0, mb, s));
        std::vector<uint8_t> hb((size_t)ngc * Lc * W, 0xff);
        std::vector<int> hl((size_t)lanes, 0), hfull((size_t)lanes, 0);
        std::vector<uint32_t> sparse_lanes, all_lanes;
        int Lb = 1;
        for (int gi = 0; gi < lanes; gi++) {
            const size_t slot = (size_t)g0 * W + gi;
            if (slot >= R) continue;
            const uint32_t rd = plan.order[slot];
            const int wr = (int)(pre.off[rd + 1] - pre.off[rd]);
            hl[gi] = wr;
            hfull[gi] = (int)(off[rd + 1] - off[rd]);
            const int g = gi / W, r = gi % W;
            for (int i = 0; i < wr; i++) hb[((size_t)g * Lc + i) * W + r] = pre.bases[pre.off[rd] + i];
            Lb = std::max(Lb, hfull[gi]);
            all_lanes.push_back((uint32_t)gi);
            if (hfull[gi] > wr) sparse_lanes.push_back((uint32_t)gi);
        }
        std::vector<double> hib;
        host_logib(m, (size_t)Lc, hib);
        HIP_CHECK(hipMemcpyAsync((void *)a.bases, hb.data(), hb.size(), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync((void *)a.len, hl.data(), hl.size() * sizeof(int), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync((void *)a.logib, hib.data(), hib.size() * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
        for (int pos = 0; pos <= Lc; pos++) launch_fwd_step(W, a, pos);
        launch_fwd_finish(W, a);
        std::vector<double> flogp((size_t)lanes, 0.0);
        HIP_CHECK(hipMemcpyAsync(flogp.data(), a.logPf, sizeof(double) * lanes, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        trace("run_sparse: dense head");
        std::vector<uint64_t> lane_pos0((size_t)lanes + 1, 0);
        for (int gi = 0; gi < lanes; gi++) lane_pos0[gi + 1] = lane_pos0[gi] + (uint64_t)hfull[gi];
        const uint64_t n_pos = lane_pos0[lanes];
        SparseFwdArgs fa{};
        // ---- forward: fixed top-k frontier from column n_warmup - 1 (forward.rs:134-150)
        Scratch sc;
        const size_t nsel = std::max<size_t>(1, std::min<size_t>(std::max<size_t>(sparse_lanes.size(), 1),
                                                                 ((size_t)256 << 20) / (12 * (size_t)m->N)));
        const size_t o_bases = sc.carve((size_t)ngc * Lb * W), o_sw = sc.carve(sizeof(int) * lanes),
                     o_tmax = sc.carve(sizeof(unsigned long long) * (size_t)ngc * Lc * W), o_lanes = sc.carve(sizeof(uint32_t) * lanes),
                     o_cn = sc.carve(sizeof(uint32_t) * (size_t)lanes * PHMM_MAX_ACTIVE_NODES),
                     o_ct = sc.carve(sizeof(double) * (size_t)lanes * PHMM_MAX_ACTIVE_NODES), o_cc = sc.carve(sizeof(int) * lanes),
                     o_out = sc.carve(sizeof(double) * lanes), o_err = sc.carve(sizeof(uint32_t) * lanes),
                     o_stop = sc.carve(sizeof(int) * lanes), o_need = sc.carve(sizeof(uint32_t) * nsel),
                     o_sn = sc.carve(sizeof(int) * nsel), o_snode = sc.carve(sizeof(uint32_t) * nsel * m->N),
                     o_stot = sc.carve(sizeof(double) * nsel * m->N);
        fsel.reserve(sc.sb);
        char *sp = (char *)fsel.p;
        if (!sparse_lanes.empty()) {
            std::vector<uint8_t> hfb((size_t)ngc * Lb * W, 0xff);
            for (uint32_t gi : sparse_lanes) {
                const uint32_t rd = plan.order[(size_t)g0 * W + gi];
                const int g = (int)gi / W, r = (int)gi % W;
                for (int i = 0; i < hfull[gi]; i++) hfb[((size_t)g * Lb + i) * W + r] = bases[off[rd] + i];
            }
            std::vector<unsigned long long> big((size_t)ngc * Lc * W, 0x7fefffffffffffffull);
            HIP_CHECK(hipMemcpyAsync(sp + o_bases, hfb.data(), hfb.size(), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(sp + o_sw, hl.data(), sizeof(int) * lanes, hipMemcpyHostToDevice, s));  // switch = prefix length
            HIP_CHECK(hipMemcpyAsync(sp + o_tmax, big.data(), sizeof(unsigned long long) * big.size(), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(sp + o_lanes, sparse_lanes.data(), sizeof(uint32_t) * sparse_lanes.size(), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemsetAsync(sp + o_cc, 0, sizeof(int) * lanes, s));
            // the frontier kernel indexes the full reads
            HIP_CHECK(hipMemcpyAsync((void *)a.len, hfull.data(), sizeof(int) * lanes, hipMemcpyHostToDevice, s));
            for (size_t nb0 = 0; nb0 < sparse_lanes.size(); nb0 += nsel) {
                const size_t nn = std::min(nsel, sparse_lanes.size() - nb0);
                HIP_CHECK(hipMemcpyAsync(sp + o_need, sparse_lanes.data() + nb0, sizeof(uint32_t) * nn, hipMemcpyHostToDevice, s));
                Top400Args ta{};
                ta.d = a;
                ta.d.tmaxF = (unsigned long long *)(sp + o_tmax);
                ta.W = W;
                ta.sw = (const int *)(sp + o_sw);
                ta.need = (const uint32_t *)(sp + o_need);
                ta.sc_node = (uint32_t *)(sp + o_snode);
                ta.sc_tot = (double *)(sp + o_stot);
                ta.sc_n = (int *)(sp + o_sn);
                ta.cand_node = (uint32_t *)(sp + o_cn);
                ta.cand_tot = (double *)(sp + o_ct);
                ta.cand_n = (int *)(sp + o_cc);
                ta.ratio_lin = 0.0;
                ta.K = K;
                launch_select_top(ta, (unsigned)nn, s);
                HIP_CHECK(hipStreamSynchronize(s));
            }
            fa.M = sparse_model_of(m);
            fa.d = a;
            fa.W = W;
            fa.sw = (const int *)(sp + o_sw);
            fa.cand_node = (const uint32_t *)(sp + o_cn);
            fa.cand_tot = (const double *)(sp + o_ct);
            fa.cand_n = (const int *)(sp + o_cc);
            fa.lanes = (const uint32_t *)(sp + o_lanes);
            fa.bases = (const uint8_t *)(sp + o_bases);
            fa.Lb = Lb;
            fa.ratio_lin = 0.0;
            fa.topk = K;
            fa.out_logp = (double *)(sp + o_out);
            fa.err = (uint32_t *)(sp + o_err);
            fa.stop = (int *)(sp + o_stop);
            fa.mode = 0;
            fa.max_steps = 0;
            uint64_t pool_cap = n_pos * 2048 + (1u << 20);
            std::vector<double> hout((size_t)lanes);
            std::vector<uint32_t> herr((size_t)lanes);
            for (int attempt = 0;; attempt++) {
                fpool.reserve(pool_cap);
                const size_t meta = 8 + sizeof(uint64_t) * (n_pos + 1) + sizeof(uint64_t) * ((size_t)lanes + 1);
                fmeta.reserve(meta);
                HIP_CHECK(hipMemsetAsync(fmeta.p, 0, meta, s));
                fa.pool.base = fpool.as<uint8_t>();
                fa.pool.cap = pool_cap;
                fa.pool.top = fmeta.as<unsigned long long>();
                fa.pool.off = (uint64_t *)(fmeta.as<char>() + 8);
                uint64_t *d_lp0 = fa.pool.off + (n_pos + 1);
                HIP_CHECK(hipMemcpyAsync(d_lp0, lane_pos0.data(), sizeof(uint64_t) * lanes, hipMemcpyHostToDevice, s));
                fa.lane_pos0 = d_lp0;
                hipLaunchKernelGGL((sparse_forward_kernel<PHMM_MAX_ACTIVE_NODES>), dim3((unsigned)sparse_lanes.size()), dim3(64), 0,
                                   s, fa);
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipMemcpyAsync(hout.data(), fa.out_logp, sizeof(double) * lanes, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipMemcpyAsync(herr.data(), fa.err, sizeof(uint32_t) * lanes, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                bool pool_full = false;
                for (uint32_t gi : sparse_lanes) pool_full |= (herr[gi] & SP_ERR_POOL) != 0;
                if (!pool_full) break;
                if (attempt >= 4) PHMM_THROW(PHMM_ENOMEM, "run_sparse: record pool");
                pool_cap *= 4;
            }
            for (uint32_t gi : sparse_lanes) {
                if (herr[gi]) PHMM_THROW(PHMM_EINTERNAL, "run_sparse: forward frontier kernel error");
                flogp[gi] = hout[gi];
            }
            trace("run_sparse: forward frontier");
        }
        // ---- backward_sparse with every column kept
        BwdChunk c;
        c.g0 = g0;
        c.ngc = ngc;
        c.Lc = Lc;
        c.full_b = true;
        c.keep = true;
        backward_chunk(m, plan, base, suf, bases, off, R, K, c, bufs);
        for (int gi = 0; gi < lanes; gi++) {
            const size_t slot = (size_t)g0 * W + gi;
            if (slot >= R) continue;
            lf[plan.order[slot]] = flogp[gi];
            lb[plan.order[slot]] = c.logp[gi];
        }
        // ---- F (.) B / P over the merged indices
        if (out_nf) {
            std::vector<int2> tasks;
            for (uint32_t gi : all_lanes) {
                const int len = hfull[gi], wr = hl[gi];
                for (int j = std::max(1, len - wr); j <= std::min(nw, len); j++) tasks.push_back(make_int2((int)gi, j));
            }
            Scratch cs;
            const size_t c_len = cs.carve(sizeof(int) * lanes), c_wr = cs.carve(sizeof(int) * lanes),
                         c_lp = cs.carve(sizeof(double) * lanes), c_lanes = cs.carve(sizeof(uint32_t) * lanes),
                         c_lp0 = cs.carve(sizeof(uint64_t) * ((size_t)lanes + 1)), c_tasks = cs.carve(sizeof(int2) * std::max<size_t>(tasks.size(), 1));
            cbuf.reserve(cs.sb);
            char *cp = (char *)cbuf.p;
            HIP_CHECK(hipMemcpyAsync(cp + c_len, hfull.data(), sizeof(int) * lanes, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(cp + c_wr, hl.data(), sizeof(int) * lanes, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(cp + c_lp, flogp.data(), sizeof(double) * lanes, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(cp + c_lanes, all_lanes.data(), sizeof(uint32_t) * all_lanes.size(), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(cp + c_lp0, lane_pos0.data(), sizeof(uint64_t) * ((size_t)lanes + 1), hipMemcpyHostToDevice, s));
            if (!tasks.empty())
                HIP_CHECK(hipMemcpyAsync(cp + c_tasks, tasks.data(), sizeof(int2) * tasks.size(), hipMemcpyHostToDevice, s));
            CombineArgs ca{};
            ca.W = W;
            ca.N = (int)m->N;
            ca.Lc = Lc;
            ca.bcols = c.a.bcols;
            ca.nw = nw;
            ca.Fm = a.Fm;
            ca.Fi = a.Fi;
            ca.Fd = a.Fd;
            ca.FE = a.FE;
            ca.Bm = c.a.Bm;
            ca.Bi = c.a.Bi;
            ca.Bd = c.a.Bd;
            ca.BE = c.a.BE;
            ca.fpool = fa.pool;
            ca.bpool = c.ba.pool;
            ca.lane_pos0 = (const uint64_t *)(cp + c_lp0);
            ca.len = (const int *)(cp + c_len);
            ca.wr = (const int *)(cp + c_wr);
            ca.logP = (const double *)(cp + c_lp);
            ca.p_end = m->lin.p_end;
            ca.freq = freq.as<double>();
            ca.lanes = (const uint32_t *)(cp + c_lanes);
            if (!sparse_lanes.empty()) {
                // (reads that are all warm-up have no sparse column on either side)
                HIP_CHECK(hipMemcpyAsync(cp + c_lanes, sparse_lanes.data(), sizeof(uint32_t) * sparse_lanes.size(), hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(run_sparse_combine, dim3((unsigned)sparse_lanes.size()), dim3(64), 0, s, ca);
                HIP_CHECK(hipGetLastError());
            }
            if (!tasks.empty()) {
                const unsigned nbx = (unsigned)std::min<size_t>(64, (m->N + BLOCK - 1) / BLOCK);
                for (size_t t0 = 0; t0 < tasks.size(); t0 += 32768) {
                    const unsigned nt = (unsigned)std::min<size_t>(32768, tasks.size() - t0);
                    hipLaunchKernelGGL(run_sparse_dense_pair, dim3(nbx, nt), dim3(BLOCK), 0, s, ca, (const int2 *)(cp + c_tasks) + t0);
                    HIP_CHECK(hipGetLastError());
                }
            }
            HIP_CHECK(hipStreamSynchronize(s));
            trace("run_sparse: posteriors");
        }
        g0 += ngc;
    }
    put_any(out_lf, lf.data(), lf.size());
    put_any(out_lb, lb.data(), lb.size());
    if (out_nf) copy_out(out_nf, freq.p, sizeof(double) * m->N);
}

}  // namespace phmm
