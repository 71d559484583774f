// Log-domain dense forward / backward for the reads the scaled linear kernels cannot certify.
//
// dense.hip keeps a column as linear values under ONE power-of-two exponent per (read, column): a cell more
// than ~700 nats below the column maximum is flushed to zero.  That is harmless unless such a cell later
// grows back to significance -- a chimeric read, whose second half aligns somewhere the first half made
// astronomically unlikely.  The dense driver bounds the mass that flushed cells could have carried
// (certify_dense below); reads over the bound are recomputed here with the reference's own arithmetic:
// every value a natural-log f64, sums by log-sum-exp (prob.rs:181-221), the recursions as written in
// forward.rs:255-306, 337-558 and backward.rs:197-261, 299-565, posteriors as table.rs:500-517 /
// freq.rs:236-255.  One block per read walks the columns; correctness over speed (these reads are rare).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "dense_internal.h"

namespace phmm {

namespace {

constexpr int XB = 512;

struct ExactArgs {
    int N;
    const uint8_t *emis;
    const double *init_l;   // [N] ln init_prob
    const double *trans_l;  // [E] ln trans_prob by edge id
    const uint32_t *par_off, *par_node, *par_edge, *chi_off, *chi_node, *chi_edge;
    phmm_params p;
    const uint8_t *bases;   // the reads, concatenated
    const uint64_t *roff;   // [n+1] position offsets
    double *Fm, *Fi, *Fd;   // [positions][N]
    double *Bm, *Bi, *Bd;   // [positions][N] (keep_b) or [n][2][N]
    int keep_b;
    double *lev;            // [n][2][N] Del levels
    double *fscal, *bscal;  // [positions][3] mb, ib, e
    double *out_lf, *out_lb;
    double *freq;           // [N] linear node posteriors (atomic) or null
};

__device__ __forceinline__ double lse2(double a, double b) {
    const double hi = a >= b ? a : b, lo = a >= b ? b : a;
    if (lo == -INFINITY) return hi;
    return hi + log1p(exp(lo - hi));
}
__device__ __forceinline__ double lse3(double a, double b, double c) { return lse2(lse2(a, b), c); }

__device__ double block_lse(double v, double *lds) {
    __syncthreads();
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int s = XB / 2; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) lds[threadIdx.x] = lse2(lds[threadIdx.x], lds[threadIdx.x + s]);
        __syncthreads();
    }
    const double r = lds[0];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(XB) exact_dense_kernel(const ExactArgs a) {
    __shared__ double lds[XB];
    const int rid = blockIdx.x, tid = threadIdx.x, N = a.N;
    const uint64_t o = a.roff[rid];
    const int L = (int)(a.roff[rid + 1] - o);
    const uint8_t *x = a.bases + o;
    const phmm_params &p = a.p;
    double *Fm = a.Fm + o * N, *Fi = a.Fi + o * N, *Fd = a.Fd + o * N;
    double *lvA = a.lev + (size_t)rid * 2 * N, *lvB = lvA + N;
    double *fs = a.fscal + o * 3, *bs = a.bscal + o * 3;
    const double NI = -INFINITY;
    // ---------------------------------------------------------------- forward (forward.rs:24-45, 276-306)
    double mbp = 0.0, ibp = NI;  // f_init: mb = 1
    for (int i = 0; i < L; i++) {
        const size_t po = (size_t)(i > 0 ? i - 1 : 0) * N;
        const double *pm = Fm + po, *pi = Fi + po, *pd = Fd + po;
        double *cm = Fm + (size_t)i * N, *ci = Fi + (size_t)i * N, *cd = Fd + (size_t)i * N;
        const uint8_t xi = x[i];
        const double beg_m = lse2(p.p_MM + mbp, p.p_IM + ibp);
        for (int k = tid; k < N; k += XB) {  // fm, fi (forward.rs:337-388)
            double acc = NI, own = NI;
            if (i > 0) {
                for (uint32_t e = a.par_off[k]; e < a.par_off[k + 1]; e++) {
                    const uint32_t l = a.par_node[e];
                    acc = lse2(acc, a.trans_l[a.par_edge[e]] + lse3(p.p_MM + pm[l], p.p_IM + pi[l], p.p_DM + pd[l]));
                }
                own = lse3(p.p_MI + pm[k], p.p_II + pi[k], p.p_DI + pd[k]);
            }
            cm[k] = (a.emis[k] == xi ? p.p_match : p.p_mismatch) + lse2(acc, a.init_l[k] + beg_m);
            ci[k] = p.p_random + own;
        }
        const double mbc = NI;                                              // fmb (forward.rs:531-533)
        const double ibc = p.p_random + lse2(p.p_MI + mbp, p.p_II + ibp);   // fib (541-545)
        __syncthreads();
        const double beg_d = lse2(p.p_MD + mbc, p.p_ID + ibc);
        for (int k = tid; k < N; k += XB) {  // fd0 (forward.rs:480-501)
            double acc = NI;
            for (uint32_t e = a.par_off[k]; e < a.par_off[k + 1]; e++) {
                const uint32_t l = a.par_node[e];
                acc = lse2(acc, a.trans_l[a.par_edge[e]] + lse2(p.p_MD + cm[l], p.p_ID + ci[l]));
            }
            const double v = lse2(acc, a.init_l[k] + beg_d);
            lvA[k] = v;
            cd[k] = v;
        }
        __syncthreads();
        double *lp = lvA, *lc = lvB;
        for (int t = 0; t < (int)p.n_max_gaps; t++) {  // fdt (forward.rs:510-524)
            for (int k = tid; k < N; k += XB) {
                double acc = NI;
                for (uint32_t e = a.par_off[k]; e < a.par_off[k + 1]; e++)
                    acc = lse2(acc, a.trans_l[a.par_edge[e]] + (p.p_DD + lp[a.par_node[e]]));
                lc[k] = acc;
                cd[k] = lse2(cd[k], acc);
            }
            __syncthreads();
            double *tmp = lp;
            lp = lc;
            lc = tmp;
        }
        double part = NI;  // fe (forward.rs:554-558)
        for (int k = tid; k < N; k += XB) part = lse2(part, lse3(cm[k], ci[k], cd[k]));
        const double e = p.p_end + block_lse(part, lds);
        if (tid == 0) {
            fs[3 * i + 0] = mbc;
            fs[3 * i + 1] = ibc;
            fs[3 * i + 2] = e;
        }
        mbp = mbc;
        ibp = ibc;
        if (i == L - 1 && tid == 0) a.out_lf[rid] = e;
        __syncthreads();
    }
    const double P = fs[3 * (L - 1) + 2];
    // ---------------------------------------------------------------- backward (backward.rs:24-53, 216-261)
    double ibn = NI;  // b_init: mb = ib = e = 0
    for (int i = L - 1; i >= 0; i--) {
        const bool init = i == L - 1;  // previous table is b_init: m = i = d = p_end (backward.rs:197-211)
        const size_t cslot = a.keep_b ? (size_t)(o + i) : (size_t)rid * 2 + (size_t)(i & 1);
        const size_t nslot = a.keep_b ? (size_t)(o + i + 1) : (size_t)rid * 2 + (size_t)((i + 1) & 1);
        double *cm = a.Bm + cslot * N, *ci = a.Bi + cslot * N, *cd = a.Bd + cslot * N;
        const double *nm = a.Bm + nslot * N, *ni = a.Bi + nslot * N;
        const uint8_t xi = x[i];
        for (int v = tid; v < N; v += XB) {  // bd0 (backward.rs:354-377)
            double acc = NI;
            for (uint32_t e = a.chi_off[v]; e < a.chi_off[v + 1]; e++) {
                const uint32_t w = a.chi_node[e];
                acc = lse2(acc, a.trans_l[a.chi_edge[e]] + p.p_DM + (a.emis[w] == xi ? p.p_match : p.p_mismatch) +
                                    (init ? p.p_end : nm[w]));
            }
            const double d0 = lse2(acc, p.p_DI + p.p_random + (init ? p.p_end : ni[v]));
            lvA[v] = d0;
            cd[v] = d0;
        }
        __syncthreads();
        double *lp = lvA, *lc = lvB;
        for (int t = 0; t < (int)p.n_max_gaps; t++) {  // bdt (backward.rs:387-404)
            for (int v = tid; v < N; v += XB) {
                double acc = NI;
                for (uint32_t e = a.chi_off[v]; e < a.chi_off[v + 1]; e++)
                    acc = lse2(acc, a.trans_l[a.chi_edge[e]] + (p.p_DD + lp[a.chi_node[e]]));
                lc[v] = acc;
                cd[v] = lse2(cd[v], acc);
            }
            __syncthreads();
            double *tmp = lp;
            lp = lc;
            lc = tmp;
        }
        double s1 = NI, s2 = NI;
        for (int v = tid; v < N; v += XB) {  // bm, bi (backward.rs:423-483); bmb, bib (499-555)
            double am = NI, ai = NI;
            for (uint32_t e = a.chi_off[v]; e < a.chi_off[v + 1]; e++) {
                const uint32_t w = a.chi_node[e];
                const double em = (a.emis[w] == xi ? p.p_match : p.p_mismatch) + (init ? p.p_end : nm[w]);
                const double tw = a.trans_l[a.chi_edge[e]];
                am = lse2(am, tw + lse2(p.p_MM + em, p.p_MD + cd[w]));
                ai = lse2(ai, tw + lse2(p.p_IM + em, p.p_ID + cd[w]));
            }
            const double in = init ? p.p_end : ni[v];
            cm[v] = lse2(am, p.p_MI + p.p_random + in);
            ci[v] = lse2(ai, p.p_II + p.p_random + in);
            const double ev = (a.emis[v] == xi ? p.p_match : p.p_mismatch) + (init ? p.p_end : nm[v]);
            s1 = lse2(s1, a.init_l[v] + lse2(p.p_MM + ev, p.p_MD + cd[v]));
            s2 = lse2(s2, a.init_l[v] + lse2(p.p_IM + ev, p.p_ID + cd[v]));
        }
        const double t1 = block_lse(s1, lds), t2 = block_lse(s2, lds);
        const double mbc = lse2(t1, p.p_MI + p.p_random + ibn);
        const double ibc = lse2(t2, p.p_II + p.p_random + ibn);
        if (tid == 0) {
            bs[3 * i + 0] = mbc;
            bs[3 * i + 1] = ibc;
            bs[3 * i + 2] = NI;  // be (backward.rs:563-565)
            if (i == 0) a.out_lb[rid] = mbc;
        }
        ibn = ibc;
        // posteriors: merged index i pairs F.tables[i-1] with B.tables[i]; merged index L pairs F.tables[L-1] with
        // b_init (table.rs:414-434, 500-505; freq.rs:236-255)
        if (a.freq && P > NI) {
            for (int v = tid; v < N; v += XB) {
                double c = 0.0;
                if (i >= 1) {
                    const size_t q = (size_t)(i - 1) * N + v;
                    c += exp(Fm[q] + cm[v] - P) + exp(Fi[q] + ci[v] - P) + exp(Fd[q] + cd[v] - P);
                }
                if (init) {
                    const size_t q = (size_t)(L - 1) * N + v;
                    c += exp(Fm[q] + p.p_end - P) + exp(Fi[q] + p.p_end - P) + exp(Fd[q] + p.p_end - P);
                }
                if (c != 0.0) atomicAdd(&a.freq[v], c);
            }
        }
        __syncthreads();
    }
}

// Transition posteriors of the same reads from their kept tables (freq.rs:276-298, 332-389): item < E is the PHMM
// edge k -> l (kinds mm im dm md id dd over the merged indices 1..len), item >= E the Begin -> l transitions
// (mm im md id over 0..len).  One thread per (item, read).
__global__ void __launch_bounds__(256) exact_edge_kernel(const ExactArgs a, const uint32_t *esrc, const uint32_t *edst, int E,
                                                         double *ef, double *inf) {
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= E + a.N) return;
    const int rid = blockIdx.y, N = a.N;
    const bool is_edge = item < E;
    const int k = is_edge ? (int)esrc[item] : 0;
    const int l = is_edge ? (int)edst[item] : item - E;
    const double t = is_edge ? a.trans_l[item] : a.init_l[l];
    if (!(t > -INFINITY)) return;
    const uint64_t o = a.roff[rid];
    const int L = (int)(a.roff[rid + 1] - o);
    const uint8_t *x = a.bases + o;
    const phmm_params &p = a.p;
    const double *Fm = a.Fm + o * N, *Fi = a.Fi + o * N, *Fd = a.Fd + o * N;
    const double *Bm = a.Bm + o * N, *Bd = a.Bd + o * N;
    const double *fs = a.fscal + o * 3;
    const double P = fs[3 * (L - 1) + 2];
    if (!(P > -INFINITY)) return;
    const uint8_t el = a.emis[l];
    double acc = 0.0;
    for (int i = is_edge ? 1 : 0; i <= L; i++) {
        double sm, sd;  // source side: F.table_merged(i), to Match / to Del
        if (is_edge) {
            const size_t q = (size_t)(i - 1) * N + k;
            sm = lse3(p.p_MM + Fm[q], p.p_IM + Fi[q], p.p_DM + Fd[q]);
            sd = lse3(p.p_MD + Fm[q], p.p_ID + Fi[q], p.p_DD + Fd[q]);
        } else if (i == 0) {
            sm = p.p_MM;
            sd = p.p_MD;
        } else {
            const double ib = fs[3 * (i - 1) + 1];
            sm = p.p_IM + ib;
            sd = p.p_ID + ib;
        }
        if (i < L) {
            const double pe = el == x[i] ? p.p_match : p.p_mismatch;
            const double bm = i + 1 < L ? Bm[(size_t)(i + 1) * N + l] : p.p_end;
            acc += exp(t + sm + pe + bm - P) + exp(t + sd + Bd[(size_t)i * N + l] - P);
        } else {
            acc += exp(t + sd + p.p_end - P);
        }
    }
    if (acc != 0.0) atomicAdd(is_edge ? &ef[item] : &inf[l], acc);
}

}  // namespace

// Reads `ids` of the read set, exactly.  lf / lb: per listed read; freq_dev: device [N], accumulated into;
// tabs (one read): [L][N] tables and [L][3] scalars to the host.
void exact_dense_reads(phmm_model *m, const uint8_t *bases, const uint64_t *off, const std::vector<uint32_t> &ids, double *lf,
                       double *lb, double *freq_dev, const ExactTables *tabs, double *edge_freq_dev, double *init_freq_dev) {
    hipStream_t s = current_stream();
    const size_t n = ids.size();
    if (n == 0) return;
    const size_t N = m->N;
    std::vector<uint64_t> roff(n + 1, 0);
    for (size_t j = 0; j < n; j++) roff[j + 1] = roff[j] + (off[ids[j] + 1] - off[ids[j]]);
    const uint64_t npos = roff[n];
    std::vector<uint8_t> hb(npos);
    for (size_t j = 0; j < n; j++) std::memcpy(hb.data() + roff[j], bases + off[ids[j]], roff[j + 1] - roff[j]);
    const bool want_edges = edge_freq_dev || init_freq_dev;
    const bool keep_b = tabs != nullptr || want_edges;
    DevBuf d_init, d_trans, d_bases, d_roff, tF, tB, lev, scal, outs;
    d_init.upload(m->init_logp.data(), sizeof(double) * N);
    d_trans.upload(m->trans_logp.data(), sizeof(double) * m->E);
    d_bases.upload(hb.data(), npos);
    d_roff.upload(roff.data(), sizeof(uint64_t) * (n + 1));
    tF.reserve(sizeof(double) * 3 * npos * N);
    tB.reserve(sizeof(double) * 3 * (keep_b ? npos : 2 * n) * N);
    lev.reserve(sizeof(double) * 2 * n * N);
    scal.reserve(sizeof(double) * 6 * npos);
    outs.reserve(sizeof(double) * 2 * n);
    ExactArgs a{};
    a.N = (int)N;
    a.emis = m->dev.emis.as<uint8_t>();
    a.init_l = d_init.as<double>();
    a.trans_l = d_trans.as<double>();
    a.par_off = m->dev.par_off.as<uint32_t>();
    a.par_node = m->dev.par_node.as<uint32_t>();
    a.par_edge = m->dev.par_edge.as<uint32_t>();
    a.chi_off = m->dev.chi_off.as<uint32_t>();
    a.chi_node = m->dev.chi_node.as<uint32_t>();
    a.chi_edge = m->dev.chi_edge.as<uint32_t>();
    a.p = m->params;
    a.bases = d_bases.as<uint8_t>();
    a.roff = d_roff.as<uint64_t>();
    a.Fm = tF.as<double>();
    a.Fi = a.Fm + npos * N;
    a.Fd = a.Fi + npos * N;
    const size_t bcols = keep_b ? npos : 2 * n;
    a.Bm = tB.as<double>();
    a.Bi = a.Bm + bcols * N;
    a.Bd = a.Bi + bcols * N;
    a.keep_b = keep_b ? 1 : 0;
    a.lev = lev.as<double>();
    a.fscal = scal.as<double>();
    a.bscal = a.fscal + 3 * npos;
    a.out_lf = outs.as<double>();
    a.out_lb = a.out_lf + n;
    a.freq = freq_dev;
    hipLaunchKernelGGL(exact_dense_kernel, dim3((unsigned)n), dim3(XB), 0, s, a);
    HIP_CHECK(hipGetLastError());
    DevBuf d_esrc, d_edst, dump;
    if (want_edges) {
        d_esrc.upload(m->esrc.data(), sizeof(uint32_t) * m->E);
        d_edst.upload(m->edst.data(), sizeof(uint32_t) * m->E);
        dump.reserve(sizeof(double) * std::max<size_t>(std::max<size_t>(m->E, N), 1));  // sink of an output not asked for
        const unsigned items = (unsigned)(m->E + N);
        hipLaunchKernelGGL(exact_edge_kernel, dim3((items + 255) / 256, (unsigned)n), dim3(256), 0, s, a, d_esrc.as<uint32_t>(),
                           d_edst.as<uint32_t>(), (int)m->E, edge_freq_dev ? edge_freq_dev : dump.as<double>(),
                           init_freq_dev ? init_freq_dev : dump.as<double>());
        HIP_CHECK(hipGetLastError());
    }
    std::vector<double> ho(2 * n);
    HIP_CHECK(hipMemcpyAsync(ho.data(), outs.p, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (size_t j = 0; j < n; j++) {
        if (lf) lf[j] = ho[j];
        if (lb) lb[j] = ho[n + j];
    }
    if (tabs) {
        const size_t L = (size_t)npos;
        auto get = [&](double *dst, const double *src, size_t cnt) {
            if (dst) HIP_CHECK(hipMemcpy(dst, src, sizeof(double) * cnt, hipMemcpyDeviceToHost));
        };
        get(tabs->f_m, a.Fm, L * N);
        get(tabs->f_i, a.Fi, L * N);
        get(tabs->f_d, a.Fd, L * N);
        get(tabs->f_scal, a.fscal, 3 * L);
        get(tabs->b_m, a.Bm, L * N);
        get(tabs->b_i, a.Bi, L * N);
        get(tabs->b_d, a.Bd, L * N);
        get(tabs->b_scal, a.bscal, 3 * L);
    }
    trace("exact dense reads");
}

// Certificate of the scaled linear dense run of one read.  A stored value of column i is lost (flushed) below
// 2^(E_i - 1022); the forward mass such cells of column i could have carried to the end is at most
// 3 N * 2^(FE_i - 1022 + slack) * max B_{i+1}, the backward mass 3 N * 2^(BE_i - 1022 + slack) * max F_{i-1}.
// Certified when the sum over the columns stays 2^-50 below P.  Without backward maxima (forward-only call)
// max B <= 1 is used.  All arguments are log2 of true values.
bool certify_dense(int N, int len, double log2P, const int *FE, const double *log2maxF, const int *BE, const double *log2maxB,
                   double log2_p_end) {
    if (!(log2P > -INFINITY)) return true;  // every column is exactly zero on both sides
    const double slack = 64.0;              // products inside a step sit below their operands
    const double cells = std::log2(3.0 * (double)N * (double)std::max(len, 1)) + 1.0;
    const double need = log2P - 50.0;
    for (int i = 0; i < len; i++) {
        const double floorF = (double)FE[i] - 1022.0 + slack + cells;
        double maxB = 0.0;  // B <= 1
        if (log2maxB) maxB = i + 1 < len ? std::max(log2maxB[i + 1], log2maxB[i]) + 2.0 : log2_p_end;
        if (floorF + std::min(maxB, 0.0) > need) return false;
        if (BE) {
            const double floorB = (double)BE[i] - 1022.0 + slack + cells;
            const double maxF = i >= 1 ? log2maxF[i - 1] + 2.0 : 0.0;
            if (floorB + std::min(maxF, 0.0) > need) return false;
        }
    }
    return true;
}

}  // namespace phmm
