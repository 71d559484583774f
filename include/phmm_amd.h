/*
 * phmm_amd.h -- C ABI of the MI355X-native profile-HMM read-likelihood path.
 *
 * Drop-in boundary for dbgphmm's `src/hmmv2`: the reference has no FFI layer; its
 * boundary is the set of `impl PHMMModel` methods.  Each entry point below names the
 * reference method it replaces (file:line relative to the dbgphmm source tree).  A Rust
 * shim flattens its petgraph `PHMMModel` into these arrays and calls through
 * `extern "C"` (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - plain pointers and sizes only; all probabilities cross the boundary as f64
 *    natural-log values (the reference's `Prob`), node ids as u32 (`NodeIndex`),
 *    bases as ASCII u8.
 *  - input pointers are HOST pointers.  Output pointers may be host OR device
 *    pointers (detected with hipPointerGetAttributes): device outputs let the caller
 *    all-reduce `node_freq` / totals over RCCL without a host round trip.
 *  - every function returns 0 on success or a negative PHMM_E* code;
 *    phmm_last_error() returns the message (thread-local).  Nothing unwinds across
 *    the ABI.  Where the reference panics (empty read, capacity overflow) the call
 *    fails with PHMM_EINVAL / PHMM_ECAPACITY.
 *  - handles own their model / read / mapping arrays; destroy them explicitly.  The DP
 *    workspaces belong to the device (phmm_release_workspace).  Compute calls on one
 *    device are serialised internally; the natural cut is one call per read set (all
 *    reads x all candidates), replacing the rayon `par_iter` loops of the reference.
 */
#ifndef PHMM_AMD_H
#define PHMM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHMM_OK 0
#define PHMM_EINVAL (-1)    /* bad argument (reference: assert!/panic!) */
#define PHMM_ENODEVICE (-2) /* no usable MI355X / HIP runtime error */
#define PHMM_ENOMEM (-3)    /* device or host allocation failed */
#define PHMM_ECAPACITY (-4) /* >400 active nodes (reference: ArrayVec panic, table.rs:22) */
#define PHMM_EINTERNAL (-5)

#define PHMM_MAX_ACTIVE_NODES 400 /* src/hmmv2/table.rs:22 */
#define PHMM_MAX_GAPS 6           /* supported upper bound of n_max_gaps (reference default 4) */

/* PHMMParams, src/hmmv2/params.rs:16-66.  p_* are natural-log probabilities. */
typedef struct phmm_params {
    double p_mismatch, p_match, p_random, p_gap_open, p_gap_ext, p_end;
    double p_MM, p_IM, p_DM, p_MI, p_II, p_DI, p_MD, p_ID, p_DD;
    int64_t n_active_nodes;
    double active_node_max_ratio;
    int64_t n_warmup;
    int64_t warmup_threshold;
    int64_t n_max_gaps;
} phmm_params;

typedef struct phmm_model phmm_model;       /* PHMMModel<N,E>, src/hmmv2/common.rs:61-64 */
typedef struct phmm_reads phmm_reads;       /* ReadCollection<S>, src/common/collection.rs:38-83 */
typedef struct phmm_mappings phmm_mappings; /* Mappings, src/hmmv2/hint.rs:150-152 */

/* thread-local message of the last failing call */
const char *phmm_last_error(void);
/* "dbgphmm_amd <version> gfx950" */
const char *phmm_version(void);
/* number of visible HIP devices (0 without a GPU); does not initialise a context */
int phmm_device_count(void);
/* select the device for this thread's subsequent calls (one process per GPU: LOCAL_RANK) */
int phmm_set_device(int device);
/* run subsequent work of this thread on the given hipStream_t (NULL = default stream) */
int phmm_set_stream(void *hip_stream);
/* upper bound (bytes) for the workspace of one call; 0 = automatic: the adaptive flow plans 95 % of (free + already
 * held) HBM for ALL its buffers, the dense drivers take 90 % for their tables */
int phmm_set_workspace_limit(uint64_t bytes);
/* Workspaces (DP tables, record pools, scratch) belong to the DEVICE and are shared by every handle on it:
 * a mapping model and a scoring model can be alive together, as in multi_dbg/posterior.rs:247-255, 609-630
 * (the reference builds a fresh PModel per call and drops its tables on return).  They grow on demand and are
 * kept between calls; so are the device arrays of destroyed phmm_mappings (up to 6 buffers / 8 GB: the next
 * phmm_generate_mappings call takes them instead of allocating).  phmm_release_workspace() returns all of it to the
 * device (the calling thread's current device must be idle), phmm_workspace_bytes() says how much is held.  Calls that compute on one device are
 * serialised inside the library whatever handle or thread they come from. */
int phmm_release_workspace(void);
uint64_t phmm_workspace_bytes(void);

/* PHMMParams::new / uniform, params.rs:73-125 (arguments are LINEAR probabilities) */
int phmm_params_new(double p_mismatch, double p_gap_open, double p_gap_ext, double p_end,
                    int64_t n_active_nodes, int64_t n_warmup, phmm_params *out);
int phmm_params_uniform(double p, phmm_params *out);

/* ---- model ------------------------------------------------------------------
 * Replaces SeqGraph::to_phmm's output (src/graph/seq_graph.rs:213-223): a PModel
 * flattened to arrays.  Edges in petgraph insertion order.  The topology stays on the
 * device; phmm_model_set_probs swaps init/trans for the next candidate copy-number
 * vector (what `dbg.clone(); set_copy_nums; to_phmm` does per candidate,
 * src/multi_dbg/posterior.rs:483-501). */
int phmm_model_create(uint32_t n_nodes, uint32_t n_edges, const uint8_t *emission,
                      const double *init_logp, const uint32_t *edge_src,
                      const uint32_t *edge_dst, const double *trans_logp,
                      const phmm_params *params, phmm_model **out);
int phmm_model_set_probs(phmm_model *m, const double *init_logp, const double *trans_logp);
int phmm_model_set_params(phmm_model *m, const phmm_params *params);
uint32_t phmm_model_n_nodes(const phmm_model *m);
uint32_t phmm_model_n_edges(const phmm_model *m);
void phmm_model_destroy(phmm_model *m);

/* ---- reads ------------------------------------------------------------------
 * bases: concatenated reads; offsets[R+1].  Empty reads are rejected (the reference
 * panics in last_table(), table.rs:388). */
int phmm_reads_create(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                      phmm_reads **out);
uint64_t phmm_reads_count(const phmm_reads *r);
uint64_t phmm_reads_total_bases(const phmm_reads *r);
void phmm_reads_destroy(phmm_reads *r);
/* What the most recent adaptive-sparse call on these reads (phmm_generate_mappings / phmm_full_prob_reads with
 * mappings == NULL and use_max_ratio != 0) did with each read -- for tools and parity tests that want to sample
 * the rare code paths.  out_dense_columns[R]: the read's dense/sparse switch position = number of dense warm-up
 * tables (forward.rs:107-137; the read length when it never switched).  out_flags[R]: PHMM_READ_* bits.
 * Either may be NULL.  PHMM_EINVAL before the first such call. */
#define PHMM_READ_DEFERRED 1u      /* still dense after the main plan's kept columns: redone in the all-columns plan */
#define PHMM_READ_WIDE_FRONTIER 2u /* its frontier outgrew the one-lane-per-node class at least once (400-slot bursts) */
#define PHMM_READ_FORCED_SWITCH 4u /* switched at n_warmup with more than 400 nodes inside the ratio (top-400 selection) */
int phmm_reads_last_call_info(const phmm_reads *r, uint16_t *out_dense_columns, uint32_t *out_flags);

/* ---- dense forward + backward + posteriors ----------------------------------
 * PHMMModel::run over a read set (src/hmmv2/freq.rs:42-46) followed by
 * PHMMOutput::{to_full_prob_forward, to_full_prob_backward, to_node_freqs}
 * (table.rs:482-494; freq.rs:245-255), i.e. PHMMModel::to_node_freqs /
 * to_full_prob_parallel (freq.rs:89-119) in one call.
 *   out_logp_forward[R]  = ln P(read) from the last forward table's `e`
 *   out_logp_backward[R] = ln P(read) from the first backward table's `mb`   (may be NULL)
 *   out_node_freq[N]     = sum over reads of node usage posteriors          (may be NULL)
 * Any output may be a device pointer. */
int phmm_run_dense(phmm_model *m, const phmm_reads *reads, double *out_logp_forward,
                   double *out_logp_backward, double *out_node_freq);

/* Transition posteriors: PHMMOutput::to_edge_and_init_freqs (freq.rs:276-298, 332-389) of a dense
 * `run`, summed over the reads (EM-style uses / tests; not called by `infer`).
 *   out_edge_freq[E] = sum over reads, positions and the six transition kinds (mm im dm md id dd) of
 *                      P(edge used | read);  out_init_freq[N] = the same for Begin -> node (mm im md id).
 * Any output may be NULL or a device pointer.  Keeps the full backward tables of a chunk in HBM. */
int phmm_run_dense_edges(phmm_model *m, const phmm_reads *reads, double *out_logp_forward,
                         double *out_edge_freq, double *out_init_freq);

/* Q function of an EM step from those posteriors: q_score_exact (src/hmmv2/q.rs:66-96).
 *   out_q[0] = init  = sum over emittable v of init_freq[v] * ln init_prob(v)
 *   out_q[1] = trans = sum over edges v -> w between emittable nodes of edge_freq[e] * ln trans_prob(e)
 *   out_q[2] = prior = 0 (q.rs:94-95);  QScore::total() is the sum of the three (q.rs:30-36).
 * Folded in the reference's order (nodes by id, the children of a node newest edge first).  A -inf
 * init / trans probability on that walk is PHMM_EINVAL (the reference asserts is_finite, q.rs:79, 88).
 * Host pointers; O(N + E) on the host from the probabilities the model was created / last set with. */
int phmm_q_score_exact(const phmm_model *m, const double *edge_freq, const double *init_freq,
                       double *out_q);

/* Dense tables of ONE read for parity tests / `inspect`-style tools:
 * PHMMModel::forward / backward (forward.rs:24-45; backward.rs:24-53).
 * f_m/f_i/f_d: [L][N] natural-log values of F.tables[i]; f_scal: [L][3] = mb, ib, e.
 * b_*: the same for B.tables[i].  Any pointer may be NULL.  Host pointers. */
int phmm_dense_tables(phmm_model *m, const uint8_t *read, uint64_t len, double *f_m,
                      double *f_i, double *f_d, double *f_scal, double *b_m, double *b_i,
                      double *b_d, double *b_scal);

/* ---- mappings (hints) ---------------------------------------------------------
 * Mappings (hint.rs:27-30,150-152) as a 3-level CSR over the reads of `reads`:
 * pos_off[total_bases+1] indexes nodes[]/logp[] for every read position in read order. */
int phmm_mappings_create(const phmm_reads *reads, const uint64_t *pos_off,
                         const uint32_t *nodes, const double *logp, phmm_mappings **out);
uint64_t phmm_mappings_total_positions(const phmm_mappings *mp);
uint64_t phmm_mappings_total_entries(const phmm_mappings *mp);
int phmm_mappings_export(const phmm_mappings *mp, uint64_t *pos_off, uint32_t *nodes,
                         double *logp);
/* Mappings::to_node_freqs, hint.rs:161-171 */
int phmm_mappings_node_freqs(const phmm_mappings *mp, uint32_t n_nodes, double *out_freq);
/* ln P(read) (forward `e` of the last table, table.rs:395-401) of the run_sparse_adaptive pass
 * that produced `mp` in phmm_generate_mappings: PHMMOutput::to_full_prob_forward per read.
 * out_logp[R] / out_total (their sum) may be NULL or device pointers. */
int phmm_mappings_read_logp(const phmm_mappings *mp, double *out_logp, double *out_total);
void phmm_mappings_destroy(phmm_mappings *mp);

/* ---- read-set likelihood ------------------------------------------------------
 * PHMMModel::to_full_prob_reads (freq.rs:175-192): per read
 * forward_with_mapping_score_only (forward.rs:79-89) when mappings != NULL, else
 * forward_sparse_score_only(use_max_ratio) (forward.rs:158-206).
 *   out_logp[R] per-read ln P; out_total = their sum (rayon `.product()`).
 * Either may be NULL / a device pointer.
 * With mappings, a read whose listed nodes all die under the model (a k-mer on its path at copy
 * number 0) still gets the reference's finite ln P: the InsBegin chain re-enters the graph behind the
 * cut (forward.rs:337-359, 541-545); -inf only where the reference's own value is -inf. */
int phmm_full_prob_reads(phmm_model *m, const phmm_reads *reads,
                         const phmm_mappings *mappings, int use_max_ratio,
                         double *out_logp, double *out_total);

/* Candidate-batched form of the loop in sample_posterior_once
 * (src/multi_dbg/posterior.rs:483-515): C candidate (init, trans) vectors on one
 * topology, all reads, hinted forward.  init_logp [C][N], trans_logp [C][E] host
 * pointers; out_logp [C][R] (may be NULL), out_total [C]. */
int phmm_full_prob_reads_candidates(phmm_model *m, const phmm_reads *reads,
                                    const phmm_mappings *mappings, uint32_t n_candidates,
                                    const double *init_logp, const double *trans_logp,
                                    double *out_logp, double *out_total);

/* The same loop with the candidates given as COPY-NUMBER vectors (what the sampler actually
 * varies: `dbg.set_copy_nums(copy_nums); dbg.to_phmm(param)`, posterior.rs:485-487, 253) instead of
 * probability vectors: copy_nums [C][N] (host pointer; PHMM node v = DBG k-mer v), and init / trans
 * are derived on the device as SeqGraph::to_phmm does (seq_graph.rs:110-135, 160-209, no edge copy
 * numbers): init[v] = max(cn[v], min_copy_num) / sum over emittable nodes, trans[v->w] =
 * max(cn[w], min_copy_num) / sum over the emittable children of v; 0 into / from a non-emittable
 * node (emission 'n').  min_copy_num: 0 = to_phmm, 1 = to_non_zero_phmm (seq_graph.rs:263-273).
 * Removes the per-candidate host rebuild and the C x (N + E) x 8-byte upload of the form above. */
int phmm_full_prob_reads_copy_nums(phmm_model *m, const phmm_reads *reads,
                                   const phmm_mappings *mappings, uint32_t n_candidates,
                                   const uint32_t *copy_nums, uint32_t min_copy_num,
                                   double *out_logp, double *out_total);

/* PHMMModel::to_full_prob_sparse_backward (freq.rs:153-163): ln P(read) from PHMMModel::backward_sparse
 * (backward.rs:146-185) -- dense b_step over the last n_warmup positions, then the backward recursion on
 * its own frontier: top_nodes(n_active_nodes) of the previous column, adaptive b_step (backward.rs:216-261).
 *   out_logp[R] = B.tables[0].mb per read;  out_total = their sum.  Either may be NULL or a device pointer.
 * n_warmup = 0 is PHMM_EINVAL (the reference panics in last_table(), table.rs:388). */
int phmm_full_prob_sparse_backward(phmm_model *m, const phmm_reads *reads, double *out_logp,
                                   double *out_total);

/* PHMMModel::run_sparse over a read set (freq.rs:51-55): forward_sparse(use_max_ratio = false) paired with
 * backward_sparse, followed by PHMMOutput::{to_full_prob_forward, to_full_prob_backward, to_node_freqs}
 * (table.rs:482-494; freq.rs:245-255): the product of a dense and a sparse table keeps the sparse operand's
 * elements, of two sparse tables the forward one's (table.rs:320-331).
 *   out_logp_forward[R], out_logp_backward[R], out_node_freq[N] (summed over the reads); any may be NULL or a
 * device pointer.  Keeps the dense head / tail columns and every sparse column of a chunk in HBM. */
int phmm_run_sparse(phmm_model *m, const phmm_reads *reads, double *out_logp_forward,
                    double *out_logp_backward, double *out_node_freq);

/* The backward_sparse tables of ONE read for parity tests / `inspect`-style tools (bin/table.rs:41):
 * b_m/b_i/b_d: [L][N] natural-log values of B.tables[i], -inf where the reference's SparseVec holds no
 * element; b_scal: [L][3] = mb, ib, e; is_dense: [L] 1 for the dense tail.  Any pointer may be NULL.
 * Host pointers. */
int phmm_backward_sparse_tables(phmm_model *m, const uint8_t *read, uint64_t len, double *b_m,
                                double *b_i, double *b_d, double *b_scal, uint8_t *is_dense);

/* Mapping carry-over between graphs: Mapping::map_nodes (hint.rs:60-88) for every read, the carrier of
 * MultiDbg::hint_kp1_from_hint_k (multi_dbg.rs:1325-1335: node of the k-HMM -> the k+1-HMM nodes of its
 * parent edges) and PurgeEdgeMap::update_mapping (multi_dbg.rs:1783-1793: node -> its id after purging, or
 * nothing).  The node map is a CSR over the nodes of the graph the mappings were made on:
 * images of node v = map_nodes[map_off[v] .. map_off[v+1]) (ids of `model_after`'s nodes; may be empty).
 * Per position: prob(image) += prob(node) / |images(node)|, then the 400 most probable, descending.
 * read_logp of the result is copied from `mappings`. */
int phmm_mappings_map_nodes(phmm_model *model_after, const phmm_reads *reads,
                            const phmm_mappings *mappings, const uint32_t *map_off,
                            const uint32_t *map_nodes, uint32_t n_nodes_before,
                            phmm_mappings **out);

/* PHMMModel::generate_mappings (hint.rs:193-220): run_with_mapping when `mappings`
 * is given else run_sparse_adaptive(use_max_ratio); then to_mapping_by_score_ratio /
 * to_mapping.  out_node_freq[N] (may be NULL) = Mappings::to_node_freqs.
 * With `mappings` the model must give every read a positive probability on its lists (the reference maps on
 * to_non_zero_phmm, multi_dbg/posterior.rs:609-618): a model that cuts a read -- a k-mer at copy number 0 on its
 * path -- is refused with PHMM_EINVAL; only phmm_full_prob_reads* score such reads (wide-range pass). */
int phmm_generate_mappings(phmm_model *m, const phmm_reads *reads,
                           const phmm_mappings *mappings, int use_max_ratio,
                           phmm_mappings **out, double *out_node_freq);

/* ---- instrumentation ------------------------------------------------------------
 * Device time (ms, HIP events on the call's stream) and launch count of the dominant
 * kernel class in the most recent call of this thread, plus the algorithmic cell
 * count it processed (bench.py's roofline block).  which: 0 = dense forward step,
 * 1 = dense backward+posterior step, 2 = hinted/sparse forward, 3 = sparse backward. */
int phmm_last_call_stats(int which, double *out_ms, uint64_t *out_launches,
                         uint64_t *out_cells);
int phmm_enable_timing(int on);

#ifdef __cplusplus
}
#endif
#endif /* PHMM_AMD_H */
