/*
 * phmm_oracle.h -- CPU ORACLE for the dbgphmm profile-HMM read-likelihood path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (dbgphmm_amd/, include/phmm_amd.h) never links, imports or calls it.
 *
 * It is a plain, scalar C restatement of the reference algorithm (f64 log-space
 * `Prob` arithmetic, one table per read position, the same node sets per position),
 * written from the reference's Rust sources which cannot be compiled in this image
 * (no cargo/rustc; three git dependencies are not vendored).  Every function cites
 * the reference file:line it follows, paths relative to /root/reference/.
 *
 * Parity pin: the dense forward / backward / posterior / transition functions are
 * pinned by the reference's own known-answer tests (tests/golden/kat_hmmv2.json,
 * transcribed from src/hmmv2/{forward,backward,freq}.rs, src/graph/seq_graph.rs and
 * src/multi_dbg/posterior/test.rs).  The sparse container `sparsevec@3634d27` is an
 * un-vendored git dependency: its tie-breaking / capacity-overflow behaviour is
 * restated from its call sites and is pinned only to the reference's property
 * tolerances ("exact selection semantics unpinned").
 */
#ifndef PHMM_ORACLE_H
#define PHMM_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_ACTIVE_NODES 400 /* src/hmmv2/table.rs:22 */

/* src/hmmv2/params.rs:16-66; all p_* are LOG probabilities (Prob). */
typedef struct orc_params {
    double p_mismatch, p_match, p_random, p_gap_open, p_gap_ext, p_end;
    double p_MM, p_IM, p_DM, p_MI, p_II, p_DI, p_MD, p_ID, p_DD;
    int64_t n_active_nodes;
    double active_node_max_ratio;
    int64_t n_warmup;
    int64_t warmup_threshold;
    int64_t n_max_gaps;
} orc_params;

typedef struct orc_model orc_model;
typedef struct orc_tables orc_tables;
typedef struct orc_mappings orc_mappings;

/* One read's Mapping (src/hmmv2/hint.rs:27-30) as a CSR view. */
typedef struct orc_mapping_view {
    const uint64_t *pos_off; /* [L+1], offsets into nodes/logp */
    const uint32_t *nodes;
    const double *logp;
} orc_mapping_view;

const char *orc_last_error(void);
/* SparseVec capacity overflow: 0 (default) = drop the insert, 1 = fail the call. */
void orc_set_overflow_is_error(int on);

/* params.rs:73-113 (`new`, arguments are LINEAR probabilities) and 116-125 (`uniform`). */
void orc_params_new(double p_mismatch, double p_gap_open, double p_gap_ext, double p_end,
                    int64_t n_active_nodes, int64_t n_warmup, orc_params *out);
void orc_params_uniform(double p, orc_params *out);

/* Prob ops (src/prob.rs:181-221), exported for unit tests. */
double orc_logadd(double x, double y);

/* PHMMModel (src/hmmv2/common.rs:61-64) flattened. Edges are given in petgraph
 * insertion order; parents()/childs() iterate most-recently-added edge first
 * (petgraph 0.6.3 adjacency lists; src/graph/iterators.rs:104-155). */
orc_model *orc_model_create(uint32_t n_nodes, uint32_t n_edges, const uint8_t *emission,
                            const double *init_logp, const uint32_t *edge_src,
                            const uint32_t *edge_dst, const double *trans_logp);
void orc_model_destroy(orc_model *m);

/* forward drivers, src/hmmv2/forward.rs:24-251 */
enum {
    ORC_FWD_DENSE = 0,          /* forward                 forward.rs:24-45  */
    ORC_FWD_MAPPING = 1,        /* forward_with_mapping    forward.rs:51-75  */
    ORC_FWD_SPARSE_TOPK = 2,    /* forward_sparse(false)   forward.rs:93-154 */
    ORC_FWD_SPARSE_RATIO = 3,   /* forward_sparse(true)    forward.rs:93-154 */
    ORC_FWD_SPARSE_V0_TOPK = 4, /* forward_sparse_v0(false) forward.rs:210-251 */
    ORC_FWD_SPARSE_V0_RATIO = 5
};
/* backward drivers, src/hmmv2/backward.rs:24-185 */
enum {
    ORC_BWD_DENSE = 0,      /* backward              backward.rs:24-53   */
    ORC_BWD_MAPPING = 1,    /* backward_with_mapping backward.rs:59-93   */
    ORC_BWD_SPARSE = 2,     /* backward_sparse       backward.rs:146-185 */
    ORC_BWD_BY_FORWARD = 3  /* backward_by_forward   backward.rs:101-142 */
};

orc_tables *orc_forward(const orc_model *m, const orc_params *p, const uint8_t *read,
                        uint64_t len, int mode, const orc_mapping_view *mapping);
orc_tables *orc_backward(const orc_model *m, const orc_params *p, const uint8_t *read,
                         uint64_t len, int mode, const orc_mapping_view *mapping,
                         const orc_tables *forward);
void orc_tables_destroy(orc_tables *t);

/* score-only drivers: forward.rs:79-89 (mapping != NULL) / forward.rs:158-206 */
int orc_forward_score_only(const orc_model *m, const orc_params *p, const uint8_t *read,
                           uint64_t len, const orc_mapping_view *mapping, int use_max_ratio,
                           double *out_logp);

/* table accessors. index -1 = init_table, 0..n-1 = tables[i] (table.rs:368-373). */
int64_t orc_tables_len(const orc_tables *t);
int orc_tables_is_dense(const orc_tables *t, int64_t i);
/* expand table i to dense arrays (absent sparse entries = default);
 * scal = {mb, ib, e}. */
int orc_tables_get(const orc_tables *t, int64_t i, double *m, double *ins, double *d,
                   double *scal);
/* stored indices of the m (which=0), i (1) or d (2) vector in insertion order.
 * Dense vectors report n = N and 0..N-1. Returns n; idx may be NULL to count. */
int64_t orc_tables_nodes(const orc_tables *t, int64_t i, int which, uint32_t *idx);
/* table.rs:395-401 */
double orc_tables_full_prob(const orc_tables *t);

/* PHMMOutput (table.rs:450-517, freq.rs:230-255, hint.rs:124-142). */
int orc_emit_probs(const orc_tables *f, const orc_tables *b, int64_t merged_index,
                   double *m, double *ins, double *d, double *scal);
int orc_node_freqs(const orc_tables *f, const orc_tables *b, double *out_freq);
/* mapping of one read; by_ratio=0: to_mapping(n_active)  by_ratio=1:
 * to_mapping_by_score_ratio(max_ratio). Output CSR: pos_off[L+1]; nodes/logp must
 * hold L*400 entries. */
int orc_output_mapping(const orc_tables *f, const orc_tables *b, int by_ratio,
                       int64_t n_active, double max_ratio, uint64_t *pos_off,
                       uint32_t *nodes, double *logp);
/* freq.rs:332-389: trans probs at merged index i -> tp[E][6] {mm,im,dm,md,id,dd},
 * ip[N][6] (only mm,im,md,id used). Log values. */
int orc_trans_and_init_probs(const orc_model *m, const orc_params *p, const orc_tables *f,
                             const orc_tables *b, const uint8_t *read, uint64_t len,
                             uint64_t i, double *tp, double *ip);
/* freq.rs:276-298 */
int orc_edge_and_init_freqs(const orc_model *m, const orc_params *p, const orc_tables *f,
                            const orc_tables *b, const uint8_t *read, uint64_t len,
                            double *edge_freq, double *init_freq);

/* ---- read-set drivers (rayon par_iter stand-in: OpenMP over reads) ---- */

/* freq.rs:175-192 to_full_prob_reads. map_pos_off: [total_bases+1] global offsets
 * (positions of all reads concatenated in read order), or NULL for no mappings.
 * out_logp[R] per read; returns 0 on success. */
int orc_full_prob_reads(const orc_model *m, const orc_params *p, const uint8_t *bases,
                        const uint64_t *read_off, uint64_t n_reads,
                        const uint64_t *map_pos_off, const uint32_t *map_nodes,
                        const double *map_logp, int use_max_ratio, int n_threads,
                        double *out_logp);

/* hint.rs:193-220 generate_mappings. If input mappings given: run_with_mapping,
 * else run_sparse_adaptive(use_max_ratio). */
orc_mappings *orc_generate_mappings(const orc_model *m, const orc_params *p,
                                    const uint8_t *bases, const uint64_t *read_off,
                                    uint64_t n_reads, const uint64_t *map_pos_off,
                                    const uint32_t *map_nodes, const double *map_logp,
                                    int use_max_ratio, int n_threads);
uint64_t orc_mappings_total_positions(const orc_mappings *mp);
uint64_t orc_mappings_total_entries(const orc_mappings *mp);
/* copy out: pos_off[total_positions+1], nodes[total_entries], logp[total_entries] */
void orc_mappings_export(const orc_mappings *mp, uint64_t *pos_off, uint32_t *nodes,
                         double *logp);
/* hint.rs:161-171 Mappings::to_node_freqs */
void orc_mappings_node_freqs(const orc_mappings *mp, uint32_t n_nodes, double *out_freq);
void orc_mappings_destroy(orc_mappings *mp);

/* dense `run` over a read set (freq.rs:42-46, 93-103, 105-119): per-read forward and
 * backward totals and the summed node freqs (PHMMOutput::to_node_freqs). */
int orc_run_dense_reads(const orc_model *m, const orc_params *p, const uint8_t *bases,
                        const uint64_t *read_off, uint64_t n_reads, int n_threads,
                        double *out_logp_forward, double *out_logp_backward,
                        double *out_node_freq);

/* test hook: see phmm_oracle.c (tie order of the value sorts) */
void orc_set_tie_rule(double eps, int reverse);

#ifdef __cplusplus
}
#endif
#endif
