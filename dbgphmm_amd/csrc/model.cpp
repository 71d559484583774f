// Host side of the model: parameter derivation, CSR adjacency, and the "Del closure"
// tables that let the dense kernels evaluate the silent Del chain of a column without
// intra-column dependencies.
//
// Reference semantics followed: src/hmmv2/params.rs:73-125 (parameter derivation),
// src/graph/iterators.rs:104-155 + petgraph 0.6.3 adjacency order (newest edge first),
// src/hmmv2/forward.rs:423-524 and src/hmmv2/backward.rs:299-404 (the 1 + n_max_gaps
// Del sweeps that the closures unroll).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "phmm_internal.h"

namespace phmm {

static inline double lin(double logp) { return logp == -INFINITY ? 0.0 : std::exp(logp); }

static double ipow(double x, int n) {
    double r = 1.0;
    for (int i = 0; i < n; i++) r *= x;
    return r;
}

void model_build_host(phmm_model *m) {
    const uint32_t N = m->N, E = m->E;
    const phmm_params &p = m->params;
    LinParams &l = m->lin;
    l.p_mismatch = lin(p.p_mismatch);
    l.p_match = lin(p.p_match);
    l.p_random = lin(p.p_random);
    l.p_end = lin(p.p_end);
    l.p_MM = lin(p.p_MM);
    l.p_IM = lin(p.p_IM);
    l.p_DM = lin(p.p_DM);
    l.p_MI = lin(p.p_MI);
    l.p_II = lin(p.p_II);
    l.p_DI = lin(p.p_DI);
    l.p_MD = lin(p.p_MD);
    l.p_ID = lin(p.p_ID);
    l.p_DD = lin(p.p_DD);
    l.n_max_gaps = (int)p.n_max_gaps;

    m->par_off.assign(N + 1, 0);
    m->chi_off.assign(N + 1, 0);
    m->par_node.resize(E);
    m->par_edge.resize(E);
    m->chi_node.resize(E);
    m->chi_edge.resize(E);
    for (uint32_t e = 0; e < E; e++) {
        m->par_off[m->edst[e] + 1]++;
        m->chi_off[m->esrc[e] + 1]++;
    }
    for (uint32_t v = 0; v < N; v++) {
        m->par_off[v + 1] += m->par_off[v];
        m->chi_off[v + 1] += m->chi_off[v];
    }
    std::vector<uint32_t> pc(N, 0), cc(N, 0);
    for (uint32_t r = 0; r < E; r++) {  // newest edge first
        uint32_t e = E - 1 - r, s = m->esrc[e], d = m->edst[e];
        uint32_t a = m->par_off[d] + pc[d]++;
        m->par_node[a] = s;
        m->par_edge[a] = e;
        uint32_t b = m->chi_off[s] + cc[s]++;
        m->chi_node[b] = d;
        m->chi_edge[b] = e;
    }
}

// Enumerate all walks of 1..H edges that end (forward=true: ancestors) or start
// (forward=false: descendants) at `root`, merging per reached node.
template <class Visit>
static void walk_closure(const std::vector<uint32_t> &off, const std::vector<uint32_t> &nb,
                         const std::vector<uint32_t> &eid, const std::vector<double> &tlin,
                         uint32_t root, int H, Visit &&visit) {
    struct Frame {
        uint32_t node;
        int hop;
        double w;
    };
    Frame stack[64 * 8];
    int sp = 0;
    stack[sp++] = {root, 0, 1.0};
    std::vector<Frame> overflow;  // only used for pathological fan-in
    while (sp > 0 || !overflow.empty()) {
        Frame f;
        if (sp > 0) f = stack[--sp];
        else {
            f = overflow.back();
            overflow.pop_back();
        }
        if (f.hop > 0) visit(f.node, f.hop, f.w);
        if (f.hop == H) continue;
        for (uint32_t a = off[f.node]; a < off[f.node + 1]; a++) {
            double w = f.w * tlin[eid[a]];
            if (w == 0.0) continue;
            Frame g{nb[a], f.hop + 1, w};
            if (sp < (int)(sizeof stack / sizeof stack[0])) stack[sp++] = g;
            else overflow.push_back(g);
        }
    }
}

void model_upload(phmm_model *m) {
    const uint32_t N = m->N, E = m->E;
    const LinParams &l = m->lin;
    const int G = l.n_max_gaps;
    const int H = G + 2;
    std::vector<double> tlin(E), ilin(N);
    for (uint32_t e = 0; e < E; e++) tlin[e] = lin(m->trans_logp[e]);
    for (uint32_t v = 0; v < N; v++) ilin[v] = lin(m->init_logp[v]);
    std::vector<double> pdd(H + 1);
    for (int t = 0; t <= H; t++) pdd[t] = ipow(l.p_DD, t);

    // ---- forward closure over ancestors
    std::vector<uint32_t> fc_off(N + 1, 0), bc_off(N + 1, 0);
    std::vector<FwdEntry> fc;
    std::vector<BwdEntry> bc;
    std::vector<double> dinit(N), tdinit(N);
    double max_sd = 0.0, max_dinit = 0.0;
    fc.reserve((size_t)N * 7);
    bc.reserve((size_t)N * 7);
    std::vector<FwdEntry> tmpf;
    std::vector<BwdEntry> tmpb;
    std::vector<uint32_t> fh_off(N + 1, 0), bh_off(N + 1, 0);
    std::vector<HopEntry> fh, bh, tmph;
    fh.reserve((size_t)N * 7);
    bh.reserve((size_t)N * 7);
    auto add_hop = [&](uint32_t a, int hop, double w) {
        for (auto &e : tmph)
            if (e.node == a && (int)(e.hop_emis & 0xff) == hop) {
                e.w += w;
                return;
            }
        tmph.push_back(HopEntry{a, (uint32_t)hop | ((uint32_t)m->emission[a] << 8), w});
    };
    std::vector<NodeRec> nodes(N);
    const bool chain_ok = H <= CHAIN_HOPS;  // n_max_gaps <= 4: hop entries + window (dense.hip)
    for (uint32_t k = 0; k < N; k++) {
        tmpf.clear();
        tmph.clear();
        double di = ilin[k], tdi = 0.0;
        walk_closure(m->par_off, m->par_node, m->par_edge, tlin, k, H,
                     [&](uint32_t a, int hop, double w) {
                         add_hop(a, hop, w);
                         double w1 = hop == 1 ? w : 0.0;
                         double wD = hop <= G + 1 ? pdd[hop - 1] * w : 0.0;
                         double wT = hop >= 2 ? pdd[hop - 2] * w : 0.0;
                         if (hop <= G) di += pdd[hop] * w * ilin[a];
                         if (hop <= G + 1) tdi += pdd[hop - 1] * w * ilin[a];
                         if (w1 == 0.0 && wD == 0.0 && wT == 0.0) return;
                         for (auto &e : tmpf)
                             if (e.node == a) {
                                 e.w1 += w1;
                                 e.wD += wD;
                                 e.wT += wT;
                                 return;
                             }
                         tmpf.push_back(FwdEntry{a, 0, w1, wD, wT});
                     });
        dinit[k] = di;
        tdinit[k] = tdi;
        {   // bound of a column total by the column's max(m,i) and InsBegin (warm-up count, dense.hip)
            double sd = 0.0;
            for (const auto &e : tmpf) sd += e.wD;
            max_sd = std::max(max_sd, sd);
            max_dinit = std::max(max_dinit, di);
        }
        uint32_t flags = 0;
        fh.insert(fh.end(), tmph.begin(), tmph.end());
        fh_off[k + 1] = (uint32_t)fh.size();
        // the only parent is k-1 over an edge of weight 1: the ancestor window just slides
        if (chain_ok && k >= 1 && m->par_off[k + 1] - m->par_off[k] == 1 && m->par_node[m->par_off[k]] == k - 1 &&
            tlin[m->par_edge[m->par_off[k]]] == 1.0)
            flags |= CHAIN_F;
        fc.insert(fc.end(), tmpf.begin(), tmpf.end());
        fc_off[k + 1] = (uint32_t)fc.size();

        tmpb.clear();
        tmph.clear();
        walk_closure(m->chi_off, m->chi_node, m->chi_edge, tlin, k, H,
                     [&](uint32_t u, int hop, double w) {
                         add_hop(u, hop, w);
                         double c1 = hop == 1 ? w : 0.0;
                         double cAd = hop <= G + 1 ? pdd[hop - 1] * w : 0.0;
                         double cAt = hop >= 2 ? pdd[hop - 2] * w : 0.0;
                         double cQd = hop <= G ? pdd[hop] * w : 0.0;
                         if (c1 == 0.0 && cAd == 0.0 && cAt == 0.0 && cQd == 0.0) return;
                         for (auto &e : tmpb)
                             if (e.node == u) {
                                 e.c1 += c1;
                                 e.cAd += cAd;
                                 e.cAt += cAt;
                                 e.cQd += cQd;
                                 return;
                             }
                         tmpb.push_back(BwdEntry{u, m->emission[u], c1, cAd, cAt, cQd});
                     });
        bh.insert(bh.end(), tmph.begin(), tmph.end());
        bh_off[k + 1] = (uint32_t)bh.size();
        if (chain_ok && (uint64_t)k + 1 < N && m->chi_off[k + 1] - m->chi_off[k] == 1 &&
            m->chi_node[m->chi_off[k]] == k + 1 && tlin[m->chi_edge[m->chi_off[k]]] == 1.0)
            flags |= CHAIN_B;
        nodes[k] = NodeRec{ilin[k], dinit[k], tdinit[k], m->emission[k], flags};
        bc.insert(bc.end(), tmpb.begin(), tmpb.end());
        bc_off[k + 1] = (uint32_t)bc.size();
    }

    ModelDev &d = m->dev;
    d.N = N;
    d.E = E;
    d.nodes.upload(nodes.data(), sizeof(NodeRec) * N);
    d.emis.upload(m->emission.data(), N);
    d.init.upload(ilin.data(), sizeof(double) * N);
    m->wf_ub_a = 2.0 + (l.p_MD + l.p_ID) * max_sd;
    m->wf_ub_b = max_dinit;
    d.dinit.upload(dinit.data(), sizeof(double) * N);
    d.tdinit.upload(tdinit.data(), sizeof(double) * N);
    d.fc_off.upload(fc_off.data(), sizeof(uint32_t) * (N + 1));
    d.fc_ent.upload(fc.data(), sizeof(FwdEntry) * fc.size());
    d.bc_off.upload(bc_off.data(), sizeof(uint32_t) * (N + 1));
    d.bc_ent.upload(bc.data(), sizeof(BwdEntry) * bc.size());
    d.fh_off.upload(fh_off.data(), sizeof(uint32_t) * (N + 1));
    d.fh_ent.upload(fh.data(), sizeof(HopEntry) * std::max<size_t>(fh.size(), 1));
    d.bh_off.upload(bh_off.data(), sizeof(uint32_t) * (N + 1));
    d.bh_ent.upload(bh.data(), sizeof(HopEntry) * std::max<size_t>(bh.size(), 1));
    // linear-domain CSR for the sparse kernels
    std::vector<double> pw(E), cw(E);
    for (uint32_t a = 0; a < E; a++) {
        pw[a] = tlin[m->par_edge[a]];
        cw[a] = tlin[m->chi_edge[a]];
    }
    d.par_off.upload(m->par_off.data(), sizeof(uint32_t) * (N + 1));
    d.par_node.upload(m->par_node.data(), sizeof(uint32_t) * E);
    d.par_edge.upload(m->par_edge.data(), sizeof(uint32_t) * E);
    d.par_w.upload(pw.data(), sizeof(double) * E);
    d.chi_off.upload(m->chi_off.data(), sizeof(uint32_t) * (N + 1));
    d.chi_node.upload(m->chi_node.data(), sizeof(uint32_t) * E);
    d.chi_edge.upload(m->chi_edge.data(), sizeof(uint32_t) * E);
    d.chi_w.upload(cw.data(), sizeof(double) * E);
    d.trans_lin.upload(tlin.data(), sizeof(double) * E);
    // packed per-node records of the frontier kernels (CSR order kept)
    std::vector<FwdAdj> fadj(N);
    std::vector<BwdAdj> badj(N);
    for (uint32_t v = 0; v < N; v++) {
        FwdAdj f{};
        BwdAdj b{};
        const uint32_t np = m->par_off[v + 1] - m->par_off[v], nc = m->chi_off[v + 1] - m->chi_off[v];
        f.over = b.over = (np > (uint32_t)ADJ_DEG || nc > (uint32_t)ADJ_DEG) ? 1 : 0;
        f.npar = (uint8_t)std::min<uint32_t>(np, ADJ_DEG);
        f.nchi = b.nchi = (uint8_t)std::min<uint32_t>(nc, ADJ_DEG);
        f.emis = b.emis = m->emission[v];
        f.init = ilin[v];
        b.par0 = np > 0 ? m->par_node[m->par_off[v]] : 0xffffffffu;
        for (uint32_t q = 0; q < (uint32_t)ADJ_DEG; q++) {
            f.par[q] = f.chi[q] = b.chi[q] = 0xffffffffu;
            if (q < f.npar) {
                f.par[q] = m->par_node[m->par_off[v] + q];
                f.par_w[q] = pw[m->par_off[v] + q];
            }
            if (q < f.nchi) {
                const uint32_t u = m->chi_node[m->chi_off[v] + q];
                f.chi[q] = b.chi[q] = u;
                b.chi_w[q] = cw[m->chi_off[v] + q];
                b.chi_emis[q] = m->emission[u];
            }
        }
        fadj[v] = f;
        badj[v] = b;
    }
    d.fadj.upload(fadj.data(), sizeof(FwdAdj) * N);
    d.badj.upload(badj.data(), sizeof(BwdAdj) * N);
    std::vector<ParRec> prec(N);
    for (uint32_t v = 0; v < N; v++) {
        ParRec r{};
        const uint32_t np = m->par_off[v + 1] - m->par_off[v];
        r.over = np > (uint32_t)ADJ_DEG ? 1 : 0;
        r.npar = (uint8_t)std::min<uint32_t>(np, ADJ_DEG);
        r.emis = m->emission[v];
        for (uint32_t q = 0; q < (uint32_t)ADJ_DEG; q++) {
            r.par[q] = q < r.npar ? m->par_node[m->par_off[v] + q] : 0xffffffffu;
            r.pedge[q] = q < r.npar ? m->par_edge[m->par_off[v] + q] : 0u;
        }
        prec[v] = r;
    }
    d.prec.upload(prec.data(), sizeof(ParRec) * N);
    d.max_degree = 0;
    for (uint32_t v = 0; v < N; v++)
        d.max_degree = std::max(d.max_degree, std::max(m->par_off[v + 1] - m->par_off[v], m->chi_off[v + 1] - m->chi_off[v]));
    d.logib_len = 0;  // parameters may have changed
    // the host vectors above die at scope exit: make sure the async copies are done
    HIP_CHECK(hipStreamSynchronize(current_stream()));
}

// forward InsBegin chain in the log domain: ib_0 = p_r*(p_MI*1 + p_II*0); ib_i = p_r*p_II*ib_{i-1}
// (fib, forward.rs:541-545 with f_init / fmb, forward.rs:255-266, 531-533)
void ensure_logib(phmm_model *m, size_t len) {
    if (m->dev.logib_len >= len) return;
    std::vector<double> v(len);
    const phmm_params &p = m->params;
    double ib = p.p_random + p.p_MI;
    for (size_t i = 0; i < len; i++) {
        v[i] = ib;
        ib = p.p_random + p.p_II + ib;
    }
    m->dev.logib.upload(v.data(), len * sizeof(double));
    HIP_CHECK(hipStreamSynchronize(current_stream()));
    m->dev.logib_len = len;
}

}  // namespace phmm
