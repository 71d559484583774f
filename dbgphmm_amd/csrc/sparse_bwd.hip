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

namespace phmm {

struct SparseBwdAdArgs {
    SparseModel M;
    int W, N, Lc, Lb;
    const double *Bm, *Bi, *Bd;       // [ng][2][N][W]: column 0 of the suffix run sits in slot 0
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
    __syncthreads();
    append_neighbours<CAP>(M, false, cur, sc, sc.order, ntop, nullptr, nullptr, nullptr, 0);
    const int na = cur.n;
    if (lane == 0) cur.na = na;
    __syncthreads();
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
    __syncthreads();
    // bdt on S_t = to_parents_and_us(S_{t-1}) (backward.rs:322-340, 387-404): level values of t-1 exist on the
    // first nprev slots only
    int nprev = na;
    for (int t = 1; t <= lp.n_max_gaps; t++) {
        for (int j = lane; j < nprev; j += 64) sc.order[j] = (uint16_t)j;
        __syncthreads();
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
        __syncthreads();
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
    __syncthreads();
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
    __syncthreads();
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
        __syncthreads();
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
        __syncthreads();
        for (int j = lane; j < nc; j += 64) {
            const uint32_t cell = fv_cell(c0, c0.id[j]);
            c0.hslot[cell] = (uint16_t)j;
        }
        if (lane == 0) c0.n = nc;
        __syncthreads();
    }
    PrevRef<CAP> pr{};
    {
        const double cm = __longlong_as_double((long long)a.cmaxB[((size_t)g * a.Lc + 0) * a.W + r]);
        const int e = sp_exp_of(cm);
        pr.vec = nullptr;
        pr.gm = a.Bm + (size_t)g * 2 * NW;
        pr.gi = a.Bi + (size_t)g * 2 * NW;
        pr.gd = a.Bd + (size_t)g * 2 * NW;
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

// Host side.  tabs != nullptr (one read): keep every column and return the tables as [L][N] natural-log
// arrays (-inf where the reference's SparseVec has no element).
struct BwdSparseTables {
    double *b_m, *b_i, *b_d, *b_scal;
    uint8_t *is_dense;
};

static void backward_sparse_impl(phmm_model *m, const uint8_t *bases, const uint64_t *off, uint64_t R, double *out_logp,
                                 const BwdSparseTables *tabs) {
    hipStream_t s = current_stream();
    const phmm_params &prm = m->params;
    if (prm.n_warmup < 1)
        PHMM_THROW(PHMM_EINVAL, "backward_sparse with n_warmup = 0: the reference panics in last_table() (table.rs:388)");
    if (prm.n_active_nodes < 1) PHMM_THROW(PHMM_EINVAL, "n_active_nodes must be positive");
    const int K = (int)std::min<int64_t>(prm.n_active_nodes, PHMM_MAX_ACTIVE_NODES);
    const int nw = (int)std::min<int64_t>(prm.n_warmup, INT32_MAX);
    // the dense tail of every read as a read of its own
    phmm_reads suf;
    suf.R = R;
    suf.off.assign(R + 1, 0);
    for (uint64_t r = 0; r < R; r++) {
        const uint64_t len = off[r + 1] - off[r];
        suf.off[r + 1] = suf.off[r] + std::min<uint64_t>(len, (uint64_t)nw);
    }
    suf.total = suf.off[R];
    suf.bases.resize(suf.total);
    for (uint64_t r = 0; r < R; r++) {
        const uint64_t wr = suf.off[r + 1] - suf.off[r];
        std::memcpy(suf.bases.data() + suf.off[r], bases + off[r + 1] - wr, wr);
        suf.max_len = std::max(suf.max_len, wr);
    }
    Plan plan = make_plan(m, &suf, tabs ? 1 : 0);
    const int W = plan.W;
    const size_t NW = (size_t)m->N * W;
    const uint64_t limit = table_budget(m->wset().tables.bytes);
    DenseArgs base{};
    fill_model_args(base, m);
    base.nblk = plan.nblk;
    base.nblk8 = plan.nblk8;
    base.npt = plan.npt;
    std::vector<double> res(R, 0.0);
    DevBuf sel, fpool, fmeta;
    int g0 = 0;
    while (g0 < plan.ng_total) {
        const uint32_t r0 = plan.order[(size_t)g0 * W];
        const int Lc = (int)(suf.off[r0 + 1] - suf.off[r0]);
        const size_t per_group = (size_t)6 * NW * 8;
        const int ngc = (int)std::min<uint64_t>(plan.ng_total - g0, std::max<uint64_t>(1, limit / std::max<size_t>(per_group, 1)));
        const int lanes = ngc * W;
        DenseArgs a = base;
        a.ng = ngc;
        a.Lc = Lc;
        size_t tb = 0, mb = 0;
        layout(a, W, false, nullptr, nullptr, tb, mb, true);
        m->wset().tables.reserve(tb);
        m->wset().misc.reserve(mb);
        layout(a, W, false, m->wset().tables.p, m->wset().misc.p, tb, mb, true);
        a.tmaxF = nullptr;
        HIP_CHECK(hipMemsetAsync(m->wset().misc.p, 0, mb, s));
        // staging: suffix bases + lengths for the dense kernel; full reads for the frontier kernel
        std::vector<uint8_t> hb((size_t)ngc * Lc * W, 0xff);
        std::vector<int> hl((size_t)lanes, 0), hfull((size_t)lanes, 0), hwr((size_t)lanes, 0);
        int Lb = 1;
        std::vector<uint32_t> sparse_lanes;
        for (int gi = 0; gi < lanes; gi++) {
            const size_t slot = (size_t)g0 * W + gi;
            if (slot >= R) continue;
            const uint32_t rd = plan.order[slot];
            const int wr = (int)(suf.off[rd + 1] - suf.off[rd]);
            const int full = (int)(off[rd + 1] - off[rd]);
            hl[gi] = wr;
            hwr[gi] = wr;
            hfull[gi] = full;
            const int g = gi / W, r = gi % W;
            for (int i = 0; i < wr; i++) hb[((size_t)g * Lc + i) * W + r] = suf.bases[suf.off[rd] + i];
            if (full > wr) {
                sparse_lanes.push_back((uint32_t)gi);
                Lb = std::max(Lb, full);
            }
        }
        HIP_CHECK(hipMemcpyAsync((void *)a.bases, hb.data(), hb.size(), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync((void *)a.len, hl.data(), hl.size() * sizeof(int), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
        for (int pos = Lc - 1; pos >= 0; pos--) launch_bwd_step(W, a, pos);
        launch_bwd_finish(W, a);
        std::vector<double> tmb((size_t)lanes);
        // (logmbB is [ng][Lc+1][W]: row 0 of every group)
        for (int g = 0; g < ngc; g++)
            HIP_CHECK(hipMemcpyAsync(tmb.data() + (size_t)g * W, a.logmbB + (size_t)g * (Lc + 1) * W, sizeof(double) * W,
                                     hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        for (int gi = 0; gi < lanes; gi++) {
            const size_t slot = (size_t)g0 * W + gi;
            if (slot < R) res[plan.order[slot]] = tmb[gi];
        }
        trace("backward_sparse: dense tail");
        if (!sparse_lanes.empty()) {
            // ---- device scratch of the frontier pass
            size_t sb = 0;
            auto carve = [&](size_t bytes) {
                sb = (sb + 255) / 256 * 256;
                const size_t o = sb;
                sb += bytes;
                return o;
            };
            const size_t nsel = std::min<size_t>(sparse_lanes.size(), std::max<size_t>(1, ((size_t)256 << 20) / (12 * (size_t)m->N)));
            const size_t o_bases = carve((size_t)ngc * Lb * W), o_len = carve(sizeof(int) * lanes), o_wr = carve(sizeof(int) * lanes),
                         o_sw = carve(sizeof(int) * lanes), o_tmax = carve(sizeof(unsigned long long) * (size_t)ngc * 2 * W),
                         o_lanes = carve(sizeof(uint32_t) * lanes), o_cn = carve(sizeof(uint32_t) * (size_t)lanes * PHMM_MAX_ACTIVE_NODES),
                         o_ct = carve(sizeof(double) * (size_t)lanes * PHMM_MAX_ACTIVE_NODES), o_cc = carve(sizeof(int) * lanes),
                         o_out = carve(sizeof(double) * lanes), o_err = carve(sizeof(uint32_t) * lanes),
                         o_need = carve(sizeof(uint32_t) * nsel), o_sn = carve(sizeof(int) * nsel),
                         o_snode = carve(sizeof(uint32_t) * nsel * m->N), o_stot = carve(sizeof(double) * nsel * m->N);
            sel.reserve(sb);
            char *sp = (char *)sel.p;
            std::vector<uint8_t> hfb((size_t)ngc * Lb * W, 0xff);
            for (uint32_t gi : sparse_lanes) {
                const uint32_t rd = plan.order[(size_t)g0 * W + gi];
                const int g = (int)gi / W, r = (int)gi % W;
                for (int i = 0; i < hfull[gi]; i++) hfb[((size_t)g * Lb + i) * W + r] = bases[off[rd] + i];
            }
            std::vector<int> ones((size_t)lanes, 1);
            std::vector<unsigned long long> big((size_t)ngc * 2 * W, 0x7fefffffffffffffull);  // DBL_MAX: bounds any total
            HIP_CHECK(hipMemcpyAsync(sp + o_bases, hfb.data(), hfb.size(), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(sp + o_len, hfull.data(), sizeof(int) * lanes, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(sp + o_wr, hwr.data(), sizeof(int) * lanes, hipMemcpyHostToDevice, s));
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
                ta.d.Lc = 2;
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
            SparseBwdAdArgs ba{};
            ba.M = sparse_model_of(m);
            ba.W = W;
            ba.N = (int)m->N;
            ba.Lc = Lc;
            ba.Lb = Lb;
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
            std::vector<uint64_t> lane_pos0((size_t)lanes + 1, 0);
            for (int gi = 0; gi < lanes; gi++) lane_pos0[gi + 1] = lane_pos0[gi] + (uint64_t)hfull[gi];
            const uint64_t n_pos = lane_pos0[lanes];
            uint64_t pool_cap = tabs ? n_pos * 2048 + (1u << 20) : 0;
            std::vector<double> hout((size_t)lanes);
            std::vector<uint32_t> herr((size_t)lanes);
            for (int attempt = 0;; attempt++) {
                if (tabs) {
                    fpool.reserve(pool_cap);
                    const size_t meta = 8 + sizeof(uint64_t) * (n_pos + 1) + sizeof(uint64_t) * ((size_t)lanes + 1) + sizeof(double) * 2 * (n_pos + 1);
                    fmeta.reserve(meta);
                    HIP_CHECK(hipMemsetAsync(fmeta.p, 0, meta, s));
                    ba.pool.base = fpool.as<uint8_t>();
                    ba.pool.cap = pool_cap;
                    ba.pool.top = fmeta.as<unsigned long long>();
                    ba.pool.off = (uint64_t *)(fmeta.as<char>() + 8);
                    uint64_t *d_lp0 = ba.pool.off + (n_pos + 1);
                    HIP_CHECK(hipMemcpyAsync(d_lp0, lane_pos0.data(), sizeof(uint64_t) * lanes, hipMemcpyHostToDevice, s));
                    ba.lane_pos0 = d_lp0;
                    ba.scal = (double *)(d_lp0 + lanes + 1);
                }
                hipLaunchKernelGGL((sparse_backward_adaptive_kernel<PHMM_MAX_ACTIVE_NODES>), dim3((unsigned)sparse_lanes.size()),
                                   dim3(64), 0, s, ba);
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
                res[plan.order[(size_t)g0 * W + gi]] = hout[gi];
            }
            trace("backward_sparse: frontier");
            if (tabs) {
                // one read: decode the kept columns
                const int L = hfull[0], N = (int)m->N, wr = hwr[0];
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
        }
        g0 += ngc;
    }
    if (out_logp) std::memcpy(out_logp, res.data(), sizeof(double) * R);
}

// to_full_prob_sparse_backward (freq.rs:153-163): per-read ln P from backward_sparse, and their sum
void full_prob_sparse_backward(phmm_model *m, const phmm_reads *reads, double *out_logp, double *out_total) {
    std::vector<double> lp(reads->R);
    backward_sparse_impl(m, reads->bases.data(), reads->off.data(), reads->R, lp.data(), nullptr);
    double tot = 0.0;
    for (double v : lp) tot += v;
    auto put = [&](double *dst, const double *src, size_t n) {
        if (!dst) return;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeDevice) {
            HIP_CHECK(hipMemcpy(dst, src, n * sizeof(double), hipMemcpyHostToDevice));
        } else {
            (void)hipGetLastError();
            std::memcpy(dst, src, n * sizeof(double));
        }
    };
    put(out_logp, lp.data(), lp.size());
    put(out_total, &tot, 1);
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
    BwdSparseTables tabs{b_m, b_i, b_d, b_scal, is_dense};
    double lp = 0.0;
    backward_sparse_impl(m, read, off, 1, &lp, &tabs);
    // the dense tail: B.tables[len-wr ..] only depend on the suffix (b_init at its end)
    const size_t o = (L - wr);
    dense_tables(m, read + o, wr, nullptr, nullptr, nullptr, nullptr, b_m ? b_m + o * N : nullptr, b_i ? b_i + o * N : nullptr,
                 b_d ? b_d + o * N : nullptr, b_scal ? b_scal + 3 * o : nullptr);
    if (is_dense)
        for (size_t i = 0; i < L; i++) is_dense[i] = i >= o ? 1 : 0;
}

}  // namespace phmm
