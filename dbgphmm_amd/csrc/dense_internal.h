// Shared between dense.hip (kernels + dense driver) and sparse_dyn.hip (dense warm-up of the
// adaptive sparse forward).
#pragma once

#include <vector>

#include "phmm_internal.h"

namespace phmm {

static constexpr double LN2 = 0.693147180559945309417232121458;
static constexpr int BLOCK = 256;

struct DenseArgs {
    int N, ng, Lc, nblk, npt;
    int g_off;  // first read group of this launch (fwd_step / bwd_step: grid.y counts from here; 0 = all groups)
    // model
    const NodeRec *nodes;
    const uint8_t *emis;
    const double *init, *dinit, *tdinit;
    const uint32_t *fc_off;
    const FwdEntry *fc;
    const uint32_t *bc_off;
    const BwdEntry *bc;
    // n_max_gaps <= 4: closure entries by hop (hop_mode = 1).  A thread keeps, for the node it is at, the partial sums
    //   D_j = sum_{h < j} p_DD^h A[h]   over the per-hop sums A[h] (h = hop-1) of its ancestors (descendants):
    // the d-closure sum is D_{G+1}, its one-hop-shifted twin the D_{G+1} of the node before, the backward Ins terms
    // p_DD Q_G and Q_{G+1}; along a chain D_j' = own + p_DD D_{j-1}
    const uint32_t *fh_off, *bh_off;
    const HopEntry *fh, *bh;
    int hop_mode;
    LinParams lp;
    const double *logib;  // [Lc] forward InsBegin chain (log)
    // read batch
    const uint8_t *bases;  // [ng][Lc][W]
    const int *len;        // [ng][W]
    // forward tables
    double *Fm, *Fi, *Fd;          // [ng][Lc][N][W]
    int *FE;                       // [ng][Lc+1][W]
    unsigned long long *cmaxF;     // [ng][Lc][W]
    double *epart;                 // [ng][ecols][nblk8][W]
    int eall;                      // 1: end sum for every column (debug tables)
    double *logPf;                 // [ng][W]
    double *logE;                  // [ng][Lc][W] per-column e (debug)
    // backward tables
    double *Bm, *Bi, *Bd;          // [ng][bcols][N][W]  (Bd may be null)
    int bcols;                     // 2 (ping-pong) or Lc
    int *BE;                       // [ng][Lc+1][W]
    unsigned long long *cmaxB;     // [ng][Lc][W]
    double *bpart;                 // [2][ng][nblk8][W][2]
    double *logmbB, *logibB;       // [ng][Lc+1][W]
    double *accg;                  // [ng][N]
    int want_freq;
    int nblk8;                     // nblk rounded up to a multiple of 8 (grid.x)
    unsigned long long *tmaxF;     // [ng][Lc][W] max over nodes of m+i+d per column (null: off)
    // backward_by_forward / mapping extraction (sparse_dyn.hip)
    const int *bstart;             // [ng][W] last dense backward column of each read (null: len-1)
    int want_map;                  // keep the per-node emit probs of the column in Pa/Pb
    double *Pa, *Pb;               // [ng][N][W] emit probs of merged index pos / len
    double *Prun;                  // [ng][nblk8][BLOCK] per thread: maximum of Pa over its run of npt nodes (null: off)
    unsigned long long *pmax;      // [ng][Lc+1][W] their maxima, by merged index
    // dense warm-up of the adaptive sparse forward (sparse_dyn.hip): launch pos also counts the nodes of
    // column pos-1 inside the score ratio (top_nodes_by_score_ratio, table.rs:134-149) -- see WarmFuse
    const int *wf_sw;              // [ng][W] switch position (-1 = still dense); null: off.  Decided lanes are skipped
    const uint8_t *wf_mode;        // [ng][W] 1: collect the candidates of this lane in this launch
    int *wf_sub;                   // [ng][W] nodes certainly inside the ratio (total > U * ratio)
    int *wf_cnt;                   // [ng][W] nodes possibly inside (total > L * ratio) = slots used in wf_node/wf_tot
    uint32_t *wf_node;             // [ng][W][WF_CAP]
    double *wf_tot;                // [ng][W][WF_CAP] totals in the column's stored exponent
    double wf_ratio;               // exp(-active_node_max_ratio)
    double wf_ub_a, wf_ub_b;       // column total <= wf_ub_a * max(m,i) + wf_ub_b * p_ID * ib   (model.cpp)
};
static constexpr int WF_CAP = 1024;


struct Plan {
    int W, ng_total, npt, nblk, nblk8;
    std::vector<uint32_t> order;  // reads sorted by length, descending
};

Plan make_plan(const phmm_model *m, const phmm_reads *reads, int forced_w);
Plan make_plan_ids(const phmm_model *m, const phmm_reads *reads, const std::vector<uint32_t> &ids);
void layout(DenseArgs &a, int W, bool full_b, void *tables, void *misc, size_t &tb, size_t &mb,
            bool backward_only = false);
void fill_model_args(DenseArgs &a, const phmm_model *m);
void host_logib(const phmm_model *m, size_t n, std::vector<double> &out);
void launch_fwd_step(int W, const DenseArgs &a, int pos);
void launch_fwd_finish(int W, const DenseArgs &a);
void launch_bwd_step(int W, const DenseArgs &a, int pos);
void launch_bwd_finish(int W, const DenseArgs &a);

// exact_dense.hip: the log-domain dense recursion for reads the scaled linear kernels cannot certify
struct ExactTables {
    double *f_m, *f_i, *f_d, *f_scal, *b_m, *b_i, *b_d, *b_scal;  // [L][N] / [L][3] host arrays, any may be null
};
void exact_dense_reads(phmm_model *m, const uint8_t *bases, const uint64_t *off, const std::vector<uint32_t> &ids, double *lf,
                       double *lb, double *freq_dev, const ExactTables *tabs, double *edge_freq_dev = nullptr,
                       double *init_freq_dev = nullptr);
bool certify_dense(int N, int len, double log2P, const int *FE, const double *log2maxF, const int *BE, const double *log2maxB,
                   double log2_p_end);

}  // namespace phmm
