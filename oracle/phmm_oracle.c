/*
 * phmm_oracle.c -- CPU ORACLE (test infrastructure, see phmm_oracle.h).
 *
 * Scalar C restatement of dbgphmm's hmmv2 path.  Citations are file:line relative to
 * /root/reference/.  The arithmetic is the reference's: every probability is an f64
 * log value, `+` is the pairwise logaddexp of src/prob.rs:181-197, sums are left
 * folds from -inf (prob.rs:235-244), and adjacency lists are walked in petgraph
 * order so even the summation order matches.
 */
#include "phmm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NEG_INF (-INFINITY)
#define CAP ORC_MAX_ACTIVE_NODES

static __thread char g_err[256];
static char g_err_shared[256];
const char *orc_last_error(void) { return g_err[0] ? g_err : g_err_shared; }
static void set_err(const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    snprintf(g_err_shared, sizeof g_err_shared, "%s", msg);
}

/* ------------------------------------------------------------------ Prob */

/* prob.rs:181-197 */
double orc_logadd(double a, double b) {
    double x = a >= b ? a : b;
    double y = a >= b ? b : a;
    if (y == NEG_INF) return x;
    if (x == y) return x + log(2.0);
    return x + log1p(exp(y - x));
}
#define LADD(a, b) orc_logadd((a), (b))

/* ------------------------------------------------------------------ params */

/* params.rs:73-113 */
void orc_params_new(double p_mismatch, double p_gap_open, double p_gap_ext, double p_end,
                    int64_t n_active_nodes, int64_t n_warmup, orc_params *o) {
    o->p_mismatch = log(p_mismatch);
    o->p_gap_open = log(p_gap_open);
    o->p_gap_ext = log(p_gap_ext);
    o->p_end = log(p_end);
    o->p_DD = o->p_gap_ext;
    o->p_II = o->p_gap_ext;
    o->p_MI = o->p_gap_open;
    o->p_MD = o->p_gap_open;
    o->p_ID = o->p_gap_open;
    o->p_DI = o->p_gap_open;
    /* the reference goes through Prob::to_value() = exp(ln p) */
    double go = exp(o->p_gap_open), ge = exp(o->p_gap_ext), pe = exp(o->p_end);
    o->p_MM = log(1.0 - 2.0 * go - pe);
    o->p_DM = log(1.0 - go - ge - pe);
    o->p_IM = log(1.0 - go - ge - pe);
    o->p_match = log(1.0 - exp(o->p_mismatch));
    o->p_random = log(0.25);
    o->n_active_nodes = n_active_nodes;
    o->active_node_max_ratio = 30.0;
    o->n_warmup = n_warmup;
    o->n_max_gaps = 4;
    o->warmup_threshold = CAP / 2;
}
/* params.rs:116-125 */
void orc_params_uniform(double p, orc_params *o) { orc_params_new(p, p, p, 0.00001, 40, 50, o); }

/* ------------------------------------------------------------------ model */

struct orc_model {
    uint32_t N, E;
    uint8_t *emission;
    double *init;
    uint32_t *esrc, *edst;
    double *trans;
    /* adjacency in petgraph iteration order (newest edge first) */
    uint32_t *par_off, *par_node, *par_edge;
    uint32_t *chi_off, *chi_node, *chi_edge;
};

orc_model *orc_model_create(uint32_t N, uint32_t E, const uint8_t *emission,
                            const double *init_logp, const uint32_t *esrc,
                            const uint32_t *edst, const double *trans_logp) {
    for (uint32_t e = 0; e < E; e++)
        if (esrc[e] >= N || edst[e] >= N) {
            set_err("orc_model_create: edge endpoint out of range");
            return NULL;
        }
    orc_model *m = calloc(1, sizeof *m);
    m->N = N;
    m->E = E;
    m->emission = malloc(N ? N : 1);
    memcpy(m->emission, emission, N);
    m->init = malloc(sizeof(double) * (N ? N : 1));
    memcpy(m->init, init_logp, sizeof(double) * N);
    m->esrc = malloc(sizeof(uint32_t) * (E ? E : 1));
    m->edst = malloc(sizeof(uint32_t) * (E ? E : 1));
    m->trans = malloc(sizeof(double) * (E ? E : 1));
    memcpy(m->esrc, esrc, sizeof(uint32_t) * E);
    memcpy(m->edst, edst, sizeof(uint32_t) * E);
    memcpy(m->trans, trans_logp, sizeof(double) * E);
    m->par_off = calloc(N + 1, sizeof(uint32_t));
    m->chi_off = calloc(N + 1, sizeof(uint32_t));
    m->par_node = malloc(sizeof(uint32_t) * (E ? E : 1));
    m->par_edge = malloc(sizeof(uint32_t) * (E ? E : 1));
    m->chi_node = malloc(sizeof(uint32_t) * (E ? E : 1));
    m->chi_edge = malloc(sizeof(uint32_t) * (E ? E : 1));
    for (uint32_t e = 0; e < E; e++) {
        m->par_off[edst[e] + 1]++;
        m->chi_off[esrc[e] + 1]++;
    }
    for (uint32_t v = 0; v < N; v++) {
        m->par_off[v + 1] += m->par_off[v];
        m->chi_off[v + 1] += m->chi_off[v];
    }
    uint32_t *pc = calloc(N + 1, sizeof(uint32_t)), *cc = calloc(N + 1, sizeof(uint32_t));
    /* newest edge first: walk edges in reverse insertion order */
    for (uint32_t r = 0; r < E; r++) {
        uint32_t e = E - 1 - r;
        uint32_t s = esrc[e], d = edst[e];
        uint32_t a = m->par_off[d] + pc[d]++;
        m->par_node[a] = s;
        m->par_edge[a] = e;
        uint32_t b = m->chi_off[s] + cc[s]++;
        m->chi_node[b] = d;
        m->chi_edge[b] = e;
    }
    free(pc);
    free(cc);
    return m;
}
void orc_model_destroy(orc_model *m) {
    if (!m) return;
    free(m->emission); free(m->init); free(m->esrc); free(m->edst); free(m->trans);
    free(m->par_off); free(m->par_node); free(m->par_edge);
    free(m->chi_off); free(m->chi_node); free(m->chi_edge);
    free(m);
}

/* ------------------------------------------------------------------ SparseVec
 * Restatement of `sparsevec::SparseVec<Prob, NodeIndex, 400>` from its call sites
 * (table.rs:87-89,106-112,121,128,139,199-211,309-338).  Dense = Vec of len values;
 * sparse = insertion-ordered (index,value) pairs, capacity 400, reads of absent
 * indices give the default, writes insert.
 *
 * Capacity overflow (UNPINNED: sparsevec@3634d27 is not vendored).  In the reference's
 * production flow the adaptive Del sweeps right after the dense->sparse switch insert
 * up to 6 x |top| <= 1200 distinct nodes into one vector (forward.rs:436-465 with
 * |top| <= warmup_threshold = 200), so an overflowing insert cannot be fatal there.
 * Default here: the insert is DROPPED (the write goes to a scratch slot, later reads
 * give the default).  orc_set_overflow_is_error(1) turns it into an error instead. */
typedef struct {
    int dense;
    uint32_t len;
    double dflt;
    double *dv;
    int n;
    uint32_t *idx;
    double *val;
} nvec;

static int g_overflow_is_error = 0;
static __thread int t_overflow = 0;
static __thread double t_sink;
void orc_set_overflow_is_error(int on) { g_overflow_is_error = on; }

static void nv_init(nvec *v, uint32_t len, double dflt, int dense) {
    v->dense = dense;
    v->len = len;
    v->dflt = dflt;
    v->n = 0;
    v->dv = NULL;
    v->idx = NULL;
    v->val = NULL;
    if (dense) {
        v->dv = malloc(sizeof(double) * (len ? len : 1));
        for (uint32_t i = 0; i < len; i++) v->dv[i] = dflt;
    }
}
static void nv_free(nvec *v) {
    free(v->dv); free(v->idx); free(v->val);
    v->dv = NULL; v->idx = NULL; v->val = NULL;
}
static inline double nv_get(const nvec *v, uint32_t i) {
    if (v->dense) return v->dv[i];
    for (int j = 0; j < v->n; j++)
        if (v->idx[j] == i) return v->val[j];
    return v->dflt;
}
static inline double *nv_ref(nvec *v, uint32_t i) {
    if (v->dense) return &v->dv[i];
    for (int j = 0; j < v->n; j++)
        if (v->idx[j] == i) return &v->val[j];
    if (v->n >= CAP) {
        if (g_overflow_is_error) t_overflow = 1;
        t_sink = v->dflt;
        return &t_sink;
    }
    if (!v->idx) {
        v->idx = malloc(sizeof(uint32_t) * CAP);
        v->val = malloc(sizeof(double) * CAP);
    }
    v->idx[v->n] = i;
    v->val[v->n] = v->dflt;
    return &v->val[v->n++];
}
static inline int nv_count(const nvec *v) { return v->dense ? (int)v->len : v->n; }
static inline uint32_t nv_idx_at(const nvec *v, int j) { return v->dense ? (uint32_t)j : v->idx[j]; }
static inline double nv_val_at(const nvec *v, int j) { return v->dense ? v->dv[j] : v->val[j]; }
/* self += other  (log-space add per stored element of other) */
static void nv_add_assign(nvec *a, const nvec *b) {
    int n = nv_count(b);
    for (int j = 0; j < n; j++) {
        double *r = nv_ref(a, nv_idx_at(b, j));
        *r = LADD(*r, nv_val_at(b, j));
    }
}
/* sorted (index,value) pairs, descending by value; ties keep iteration order
 * (assumed stable; unpinned).  Returns up to `k` (<= CAP) entries. */
typedef struct { uint32_t idx; double val; int ord; } pair_t;
/* Test hook (tools/fuzz_parity.py): orc_set_tie_rule(e, mode) treats values in the same bucket of width e
 * (floor(val / e), or floor(val / e + 1/2) with mode & 2) as tied and, with mode & 1, turns the order of tied
 * entries round -- to show that a difference between two restatements is the (unpinned) tie order and nothing
 * else.  Buckets, not "within e of each other": the comparator stays a strict weak order, as qsort requires.
 * Default (0, 0): the rule above. */
static double g_tie_eps = 0.0;
static int g_tie_rev = 0;
void orc_set_tie_rule(double eps, int mode) { g_tie_eps = eps; g_tie_rev = mode; }
static inline double tie_bucket(double v) {
    if (!(g_tie_eps > 0.0) || !isfinite(v)) return v;
    return floor(v / g_tie_eps + ((g_tie_rev & 2) ? 0.5 : 0.0));
}
static int pair_cmp(const void *a, const void *b) {
    const pair_t *x = a, *y = b;
    const double bx = tie_bucket(x->val), by = tie_bucket(y->val);
    if (bx > by) return -1;
    if (bx < by) return 1;
    int o = x->ord < y->ord ? -1 : (x->ord > y->ord ? 1 : 0);
    return (g_tie_rev & 1) ? -o : o;
}
static int nv_sorted_top(const nvec *v, int k, uint32_t *out_idx, double *out_val) {
    int n = nv_count(v);
    if (k > CAP) k = CAP;
    if (n == 0 || k <= 0) return 0;
    pair_t *ps = malloc(sizeof(pair_t) * n);
    for (int j = 0; j < n; j++) {
        ps[j].idx = nv_idx_at(v, j);
        ps[j].val = nv_val_at(v, j);
        ps[j].ord = j;
    }
    if (n > 4 * CAP) {
        /* prefilter: keep everything >= the k-th largest value (ties included) */
        double *tmp = malloc(sizeof(double) * n);
        for (int j = 0; j < n; j++) tmp[j] = ps[j].val;
        /* quickselect for k-th largest */
        int lo = 0, hi = n - 1, target = k - 1;
        while (lo < hi) {
            double piv = tmp[(lo + hi) / 2];
            int i = lo, j2 = hi;
            while (i <= j2) {
                while (tmp[i] > piv) i++;
                while (tmp[j2] < piv) j2--;
                if (i <= j2) { double t = tmp[i]; tmp[i] = tmp[j2]; tmp[j2] = t; i++; j2--; }
            }
            if (target <= j2) hi = j2;
            else if (target >= i) lo = i;
            else break;
        }
        double kth = tmp[target];
        free(tmp);
        int w = 0;
        for (int j = 0; j < n; j++)
            if (ps[j].val >= kth) ps[w++] = ps[j];
        n = w;
    }
    qsort(ps, n, sizeof(pair_t), pair_cmp);
    int r = n < k ? n : k;
    for (int j = 0; j < r; j++) {
        out_idx[j] = ps[j].idx;
        if (out_val) out_val[j] = ps[j].val;
    }
    free(ps);
    return r;
}

/* ------------------------------------------------------------------ PHMMTable (table.rs:42-73) */
typedef struct {
    nvec m, i, d;
    double mb, ib, e;
} table_t;

static void tb_init(table_t *t, int dense, uint32_t N, double m, double i, double d, double mb,
                    double ib, double e) {
    nv_init(&t->m, N, m, dense);
    nv_init(&t->i, N, i, dense);
    nv_init(&t->d, N, d, dense);
    t->mb = mb; t->ib = ib; t->e = e;
}
static void tb_zero(table_t *t, int dense, uint32_t N) {
    tb_init(t, dense, N, NEG_INF, NEG_INF, NEG_INF, NEG_INF, NEG_INF, NEG_INF);
}
static void tb_free(table_t *t) { nv_free(&t->m); nv_free(&t->i); nv_free(&t->d); }

/* table.rs:199-211 to_nodevec */
static void tb_to_nodevec(const table_t *t, nvec *v) {
    nv_init(v, t->m.len, NEG_INF, t->m.dense);
    nv_add_assign(v, &t->m);
    nv_add_assign(v, &t->i);
    nv_add_assign(v, &t->d);
}
/* table.rs:127-129 */
static int tb_top_nodes(const table_t *t, int k, uint32_t *out, double *outv) {
    nvec v;
    tb_to_nodevec(t, &v);
    int r = nv_sorted_top(&v, k, out, outv);
    nv_free(&v);
    return r;
}
/* table.rs:134-149 */
static int tb_top_nodes_by_score_ratio(const table_t *t, double max_ratio, uint32_t *out,
                                       double *outv) {
    uint32_t idx[CAP];
    double val[CAP];
    nvec v;
    tb_to_nodevec(t, &v);
    int n = nv_sorted_top(&v, CAP, idx, val);
    nv_free(&v);
    int r = 0;
    if (n > 0) {
        double p0 = val[0];
        for (int j = 0; j < n; j++)
            if (p0 - val[j] < max_ratio) {
                out[r] = idx[j];
                if (outv) outv[r] = val[j];
                r++;
            }
    }
    return r;
}
/* table.rs:117-123 filled_nodes (sparse only) */
static int tb_filled_nodes(const table_t *t, uint32_t *out) {
    return tb_top_nodes(t, t->m.n, out, NULL);
}

struct orc_tables {
    uint32_t N;
    int kind; /* 0 forward, 1 backward */
    table_t init;
    table_t *t;
    int64_t n;
};

void orc_tables_destroy(orc_tables *ts) {
    if (!ts) return;
    tb_free(&ts->init);
    for (int64_t i = 0; i < ts->n; i++) tb_free(&ts->t[i]);
    free(ts->t);
    free(ts);
}

/* ------------------------------------------------------------------ context */
typedef struct {
    const orc_model *m;
    const orc_params *p;
    uint32_t *stamp; /* for .unique() */
    uint32_t epoch;
} ctx_t;
static void ctx_init(ctx_t *c, const orc_model *m, const orc_params *p) {
    c->m = m; c->p = p;
    c->stamp = calloc(m->N ? m->N : 1, sizeof(uint32_t));
    c->epoch = 0;
}
static void ctx_free(ctx_t *c) { free(c->stamp); }

/* common.rs:168-174 */
static inline double p_match_emit(const ctx_t *c, uint32_t k, uint8_t x) {
    return c->m->emission[k] == x ? c->p->p_match : c->p->p_mismatch;
}

/* active_nodes.rs:15-56: chain(us?, flat_map(neighbors)).unique().take(400) */
static int expand_nodes(ctx_t *c, const uint32_t *nodes, int n, int children, int and_us,
                        uint32_t *out) {
    const orc_model *m = c->m;
    if (++c->epoch == 0) { memset(c->stamp, 0, sizeof(uint32_t) * m->N); c->epoch = 1; }
    int r = 0;
    if (and_us)
        for (int j = 0; j < n && r < CAP; j++)
            if (c->stamp[nodes[j]] != c->epoch) { c->stamp[nodes[j]] = c->epoch; out[r++] = nodes[j]; }
    const uint32_t *off = children ? m->chi_off : m->par_off;
    const uint32_t *nb = children ? m->chi_node : m->par_node;
    for (int j = 0; j < n && r < CAP; j++)
        for (uint32_t a = off[nodes[j]]; a < off[nodes[j] + 1] && r < CAP; a++)
            if (c->stamp[nb[a]] != c->epoch) { c->stamp[nb[a]] = c->epoch; out[r++] = nb[a]; }
    return r;
}

/* ------------------------------------------------------------------ forward kernels */

/* forward.rs:337-359 fm */
static void fm(ctx_t *c, table_t *t0, const table_t *t1, uint8_t x, const uint32_t *nodes, int64_t n) {
    const orc_model *m = c->m; const orc_params *p = c->p;
    for (int64_t j = 0; j < n; j++) {
        uint32_t k = nodes[j];
        double p_emit = p_match_emit(c, k, x);
        double from_normal = NEG_INF;
        for (uint32_t a = m->par_off[k]; a < m->par_off[k + 1]; a++) {
            uint32_t l = m->par_node[a];
            double pt = m->trans[m->par_edge[a]];
            double inner = LADD(LADD(p->p_MM + nv_get(&t1->m, l), p->p_IM + nv_get(&t1->i, l)),
                                p->p_DM + nv_get(&t1->d, l));
            from_normal = LADD(from_normal, pt + inner);
        }
        double from_begin = m->init[k] + LADD(p->p_MM + t1->mb, p->p_IM + t1->ib);
        *nv_ref(&t0->m, k) = p_emit + LADD(from_normal, from_begin);
    }
}
/* forward.rs:378-388 fi */
static void fi(ctx_t *c, table_t *t0, const table_t *t1, const uint32_t *nodes, int64_t n) {
    const orc_params *p = c->p;
    for (int64_t j = 0; j < n; j++) {
        uint32_t k = nodes[j];
        double from_me = LADD(LADD(p->p_MI + nv_get(&t1->m, k), p->p_II + nv_get(&t1->i, k)),
                              p->p_DI + nv_get(&t1->d, k));
        *nv_ref(&t0->i, k) = p->p_random + from_me;
    }
}
/* forward.rs:480-501 fd0 */
static void fd0(ctx_t *c, const table_t *t0, const uint32_t *nodes, int64_t n, int dense, table_t *out) {
    const orc_model *m = c->m; const orc_params *p = c->p;
    tb_zero(out, dense, m->N);
    for (int64_t j = 0; j < n; j++) {
        uint32_t k = nodes[j];
        double from_normal = NEG_INF;
        for (uint32_t a = m->par_off[k]; a < m->par_off[k + 1]; a++) {
            uint32_t l = m->par_node[a];
            double pt = m->trans[m->par_edge[a]];
            from_normal = LADD(from_normal,
                               pt + LADD(p->p_MD + nv_get(&t0->m, l), p->p_ID + nv_get(&t0->i, l)));
        }
        double from_begin = m->init[k] + LADD(p->p_MD + t0->mb, p->p_ID + t0->ib);
        *nv_ref(&out->d, k) = LADD(from_normal, from_begin);
    }
}
/* forward.rs:510-524 fdt */
static void fdt(ctx_t *c, const table_t *fdt1, const uint32_t *nodes, int64_t n, int dense, table_t *out) {
    const orc_model *m = c->m; const orc_params *p = c->p;
    tb_zero(out, dense, m->N);
    for (int64_t j = 0; j < n; j++) {
        uint32_t k = nodes[j];
        double s = NEG_INF;
        for (uint32_t a = m->par_off[k]; a < m->par_off[k + 1]; a++) {
            uint32_t l = m->par_node[a];
            double pt = m->trans[m->par_edge[a]];
            s = LADD(s, pt + (p->p_DD + nv_get(&fdt1->d, l)));
        }
        *nv_ref(&out->d, k) = s;
    }
}
/* forward.rs:423-466 fd */
static void fd(ctx_t *c, table_t *t0, const uint32_t *nodes, int64_t n, int adaptive) {
    int dense = t0->m.dense;
    uint32_t act[CAP], act2[CAP];
    const uint32_t *cur = nodes;
    int64_t ncur = n;
    if (adaptive) {
        ncur = expand_nodes(c, nodes, (int)n, 1, 0, act);
        cur = act;
    }
    table_t a, b;
    fd0(c, t0, cur, ncur, dense, &a);
    nv_add_assign(&t0->d, &a.d);
    for (int64_t t = 0; t < c->p->n_max_gaps; t++) {
        if (adaptive) {
            int n2 = expand_nodes(c, cur, (int)ncur, 1, 0, act2);
            memcpy(act, act2, sizeof(uint32_t) * n2);
            cur = act;
            ncur = n2;
        }
        fdt(c, &a, cur, ncur, dense, &b);
        nv_add_assign(&t0->d, &b.d);
        tb_free(&a);
        a = b;
    }
    tb_free(&a);
}
/* forward.rs:554-558 fe */
static void fe(ctx_t *c, table_t *t0, const uint32_t *nodes, int64_t n) {
    double s = NEG_INF;
    for (int64_t j = 0; j < n; j++) {
        uint32_t k = nodes[j];
        s = LADD(s, LADD(LADD(nv_get(&t0->m, k), nv_get(&t0->i, k)), nv_get(&t0->d, k)));
    }
    t0->e = c->p->p_end + s;
}
/* forward.rs:255-266 */
static void f_init(const ctx_t *c, table_t *t) {
    tb_init(t, 1, c->m->N, NEG_INF, NEG_INF, NEG_INF, 0.0, NEG_INF, NEG_INF);
}
/* forward.rs:276-306 f_step */
static void f_step(ctx_t *c, uint8_t x, const table_t *prev, const uint32_t *nodes, int64_t n,
                   int dense, int adaptive, table_t *out) {
    const orc_params *p = c->p;
    tb_zero(out, dense, c->m->N);
    fm(c, out, prev, x, nodes, n);
    fi(c, out, prev, nodes, n);
    out->mb = NEG_INF;                                                    /* fmb forward.rs:531-533 */
    out->ib = p->p_random + LADD(p->p_MI + prev->mb, p->p_II + prev->ib); /* fib forward.rs:541-545 */
    fd(c, out, nodes, n, adaptive);
    fe(c, out, nodes, n);
}

static uint32_t *all_nodes(uint32_t N) {
    uint32_t *a = malloc(sizeof(uint32_t) * (N ? N : 1));
    for (uint32_t i = 0; i < N; i++) a[i] = i;
    return a;
}

/* decides dense vs sparse for column i of forward_sparse (forward.rs:107-137) and
 * returns the active node list for a sparse step. */
static int sparse_plan(ctx_t *c, const table_t *prev, int64_t i, int use_max_ratio,
                       uint32_t *active, int *n_active) {
    const orc_params *p = c->p;
    uint32_t top[CAP];
    int ntop = use_max_ratio ? tb_top_nodes_by_score_ratio(prev, p->active_node_max_ratio, top, NULL)
                             : tb_top_nodes(prev, (int)p->n_active_nodes, top, NULL);
    int use_dense;
    if (use_max_ratio) {
        if (prev->m.dense) {
            if (i == 0) use_dense = 1;
            else if (i < p->n_warmup) use_dense = ntop > p->warmup_threshold;
            else use_dense = 0;
        } else use_dense = 0;
    } else use_dense = i < p->n_warmup;
    if (!use_dense) *n_active = expand_nodes(c, top, ntop, 1, 1, active); /* to_childs_and_us */
    return use_dense;
}

static orc_tables *tables_new(uint32_t N, int kind, int64_t len) {
    orc_tables *ts = calloc(1, sizeof *ts);
    ts->N = N; ts->kind = kind; ts->n = len;
    ts->t = calloc(len ? len : 1, sizeof(table_t));
    return ts;
}

static int mapping_check(const orc_model *m, const orc_mapping_view *mp, uint64_t len) {
    if (!mp) { set_err("mapping required"); return -1; }
    for (uint64_t i = 0; i < len; i++) {
        if (mp->pos_off[i + 1] < mp->pos_off[i] || mp->pos_off[i + 1] - mp->pos_off[i] > CAP) {
            set_err("mapping position list longer than 400 or offsets not monotone");
            return -1;
        }
        for (uint64_t a = mp->pos_off[i]; a < mp->pos_off[i + 1]; a++)
            if (mp->nodes[a] >= m->N) { set_err("mapping node out of range"); return -1; }
    }
    return 0;
}

orc_tables *orc_forward(const orc_model *m, const orc_params *p, const uint8_t *read,
                        uint64_t len, int mode, const orc_mapping_view *mp) {
    g_err[0] = 0;
    t_overflow = 0;
    if (mode == ORC_FWD_MAPPING && mapping_check(m, mp, len)) return NULL;
    ctx_t c;
    ctx_init(&c, m, p);
    orc_tables *ts = tables_new(m->N, 0, (int64_t)len);
    f_init(&c, &ts->init);
    uint32_t *alln = all_nodes(m->N);
    for (uint64_t i = 0; i < len; i++) {
        const table_t *prev = i == 0 ? &ts->init : &ts->t[i - 1];
        switch (mode) {
        case ORC_FWD_DENSE:
            f_step(&c, read[i], prev, alln, m->N, 1, 0, &ts->t[i]);
            break;
        case ORC_FWD_MAPPING:
            f_step(&c, read[i], prev, mp->nodes + mp->pos_off[i],
                   (int64_t)(mp->pos_off[i + 1] - mp->pos_off[i]), 0, 0, &ts->t[i]);
            break;
        case ORC_FWD_SPARSE_TOPK:
        case ORC_FWD_SPARSE_RATIO: {
            uint32_t act[CAP];
            int nact = 0;
            if (sparse_plan(&c, prev, (int64_t)i, mode == ORC_FWD_SPARSE_RATIO, act, &nact))
                f_step(&c, read[i], prev, alln, m->N, 1, 0, &ts->t[i]);
            else
                f_step(&c, read[i], prev, act, nact, 0, 1, &ts->t[i]);
            break;
        }
        case ORC_FWD_SPARSE_V0_TOPK:
        case ORC_FWD_SPARSE_V0_RATIO: { /* forward.rs:210-251 */
            if ((int64_t)i < p->n_warmup)
                f_step(&c, read[i], prev, alln, m->N, 1, 0, &ts->t[i]);
            else {
                uint32_t top[CAP], act[CAP];
                int ntop = mode == ORC_FWD_SPARSE_V0_RATIO
                               ? tb_top_nodes_by_score_ratio(prev, p->active_node_max_ratio, top, NULL)
                               : tb_top_nodes(prev, (int)p->n_active_nodes, top, NULL);
                int nact = expand_nodes(&c, top, ntop, 1, 1, act);
                f_step(&c, read[i], prev, act, nact, 0, 1, &ts->t[i]);
            }
            break;
        }
        default:
            set_err("orc_forward: bad mode");
            t_overflow = 1;
        }
    }
    free(alln);
    ctx_free(&c);
    if (t_overflow) {
        if (!g_err[0]) set_err("SparseVec capacity (400) exceeded: the reference would panic");
        orc_tables_destroy(ts);
        return NULL;
    }
    return ts;
}

int orc_forward_score_only(const orc_model *m, const orc_params *p, const uint8_t *read,
                           uint64_t len, const orc_mapping_view *mp, int use_max_ratio,
                           double *out_logp) {
    g_err[0] = 0;
    t_overflow = 0;
    if (len == 0) { set_err("empty read: the reference panics in last_table()"); return -1; }
    if (mp && mapping_check(m, mp, len)) return -1;
    ctx_t c;
    ctx_init(&c, m, p);
    table_t cur, nxt;
    f_init(&c, &cur);
    uint32_t *alln = mp ? NULL : all_nodes(m->N);
    for (uint64_t i = 0; i < len; i++) {
        if (mp) { /* forward.rs:79-89 */
            f_step(&c, read[i], &cur, mp->nodes + mp->pos_off[i],
                   (int64_t)(mp->pos_off[i + 1] - mp->pos_off[i]), 0, 0, &nxt);
        } else { /* forward.rs:158-206 */
            uint32_t act[CAP];
            int nact = 0;
            if (sparse_plan(&c, &cur, (int64_t)i, use_max_ratio, act, &nact))
                f_step(&c, read[i], &cur, alln, m->N, 1, 0, &nxt);
            else
                f_step(&c, read[i], &cur, act, nact, 0, 1, &nxt);
        }
        tb_free(&cur);
        cur = nxt;
    }
    *out_logp = cur.e;
    tb_free(&cur);
    free(alln);
    ctx_free(&c);
    if (t_overflow) { set_err("SparseVec capacity (400) exceeded: the reference would panic"); return -1; }
    return 0;
}

/* ------------------------------------------------------------------ backward kernels */

/* backward.rs:354-377 bd0 */
static void bd0(ctx_t *c, const table_t *t1, uint8_t x, const uint32_t *nodes, int64_t n, int dense, table_t *out) {
    const orc_model *m = c->m; const orc_params *p = c->p;
    tb_zero(out, dense, m->N);
    for (int64_t j = 0; j < n; j++) {
        uint32_t k = nodes[j];
        double to_match = NEG_INF;
        for (uint32_t a = m->chi_off[k]; a < m->chi_off[k + 1]; a++) {
            uint32_t l = m->chi_node[a];
            double pt = m->trans[m->chi_edge[a]];
            to_match = LADD(to_match, pt + p->p_DM + p_match_emit(c, l, x) + nv_get(&t1->m, l));
        }
        double to_ins = p->p_DI + p->p_random + nv_get(&t1->i, k);
        *nv_ref(&out->d, k) = LADD(to_match, to_ins);
    }
}
/* backward.rs:387-404 bdt */
static void bdt(ctx_t *c, const table_t *bdt1, const uint32_t *nodes, int64_t n, int dense, table_t *out) {
    const orc_model *m = c->m; const orc_params *p = c->p;
    tb_zero(out, dense, m->N);
    for (int64_t j = 0; j < n; j++) {
        uint32_t k = nodes[j];
        double s = NEG_INF;
        for (uint32_t a = m->chi_off[k]; a < m->chi_off[k + 1]; a++) {
            uint32_t l = m->chi_node[a];
            double pt = m->trans[m->chi_edge[a]];
            s = LADD(s, pt + p->p_DD + nv_get(&bdt1->d, l));
        }
        *nv_ref(&out->d, k) = s;
    }
}
/* backward.rs:299-343 bd */
static void bd(ctx_t *c, table_t *t0, const table_t *t1, uint8_t x, const uint32_t *nodes, int64_t n, int adaptive) {
    int dense = t0->m.dense;
    uint32_t act[CAP], act2[CAP];
    const uint32_t *cur = nodes;
    int64_t ncur = n;
    if (adaptive) {
        ncur = expand_nodes(c, nodes, (int)n, 0, 1, act); /* to_parents_and_us */
        cur = act;
    }
    table_t a, b;
    bd0(c, t1, x, cur, ncur, dense, &a);
    nv_add_assign(&t0->d, &a.d);
    for (int64_t t = 0; t < c->p->n_max_gaps; t++) {
        if (adaptive) {
            int n2 = expand_nodes(c, cur, (int)ncur, 0, 1, act2);
            memcpy(act, act2, sizeof(uint32_t) * n2);
            cur = act;
            ncur = n2;
        }
        bdt(c, &a, cur, ncur, dense, &b);
        nv_add_assign(&t0->d, &b.d);
        tb_free(&a);
        a = b;
    }
    tb_free(&a);
}
/* backward.rs:423-444 bm and 462-483 bi */
static void bm_bi(ctx_t *c, table_t *t0, const table_t *t1, uint8_t x, const uint32_t *nodes, int64_t n) {
    const orc_model *m = c->m; const orc_params *p = c->p;
    for (int64_t j = 0; j < n; j++) {
        uint32_t k = nodes[j];
        double s = NEG_INF;
        for (uint32_t a = m->chi_off[k]; a < m->chi_off[k + 1]; a++) {
            uint32_t l = m->chi_node[a];
            double pt = m->trans[m->chi_edge[a]];
            double pe = p_match_emit(c, l, x);
            s = LADD(s, pt + LADD(p->p_MM + pe + nv_get(&t1->m, l), p->p_MD + nv_get(&t0->d, l)));
        }
        double to_ins = p->p_MI + p->p_random + nv_get(&t1->i, k);
        *nv_ref(&t0->m, k) = LADD(s, to_ins);
    }
    for (int64_t j = 0; j < n; j++) {
        uint32_t k = nodes[j];
        double s = NEG_INF;
        for (uint32_t a = m->chi_off[k]; a < m->chi_off[k + 1]; a++) {
            uint32_t l = m->chi_node[a];
            double pt = m->trans[m->chi_edge[a]];
            double pe = p_match_emit(c, l, x);
            s = LADD(s, pt + LADD(p->p_IM + pe + nv_get(&t1->m, l), p->p_ID + nv_get(&t0->d, l)));
        }
        double to_ins = p->p_II + p->p_random + nv_get(&t1->i, k);
        *nv_ref(&t0->i, k) = LADD(s, to_ins);
    }
}
/* backward.rs:535-555 bib and 499-519 bmb */
static void bib_bmb(ctx_t *c, table_t *t0, const table_t *t1, uint8_t x, const uint32_t *nodes, int64_t n) {
    const orc_model *m = c->m; const orc_params *p = c->p;
    double s = NEG_INF;
    for (int64_t j = 0; j < n; j++) {
        uint32_t l = nodes[j];
        double pe = p_match_emit(c, l, x);
        s = LADD(s, m->init[l] + LADD(p->p_IM + pe + nv_get(&t1->m, l), p->p_ID + nv_get(&t0->d, l)));
    }
    t0->ib = LADD(s, p->p_II + p->p_random + t1->ib);
    s = NEG_INF;
    for (int64_t j = 0; j < n; j++) {
        uint32_t l = nodes[j];
        double pe = p_match_emit(c, l, x);
        s = LADD(s, m->init[l] + LADD(p->p_MM + pe + nv_get(&t1->m, l), p->p_MD + nv_get(&t0->d, l)));
    }
    t0->mb = LADD(s, p->p_MI + p->p_random + t1->ib);
}
/* backward.rs:197-211 */
static void b_init(const ctx_t *c, table_t *t) {
    double pe = c->p->p_end;
    tb_init(t, 1, c->m->N, pe, pe, pe, NEG_INF, NEG_INF, NEG_INF);
}
/* backward.rs:216-261 b_step */
static void b_step(ctx_t *c, uint8_t x, const table_t *prev, const uint32_t *nodes, int64_t n,
                   int dense, int adaptive, table_t *out) {
    tb_zero(out, dense, c->m->N);
    bd(c, out, prev, x, nodes, n, adaptive);
    out->e = NEG_INF; /* be backward.rs:563-565 */
    uint32_t act[CAP];
    const uint32_t *use = nodes;
    int64_t nuse = n;
    if (adaptive) {
        nuse = expand_nodes(c, nodes, (int)n, 0, 1, act);
        use = act;
    }
    bm_bi(c, out, prev, x, use, nuse);
    bib_bmb(c, out, prev, x, use, nuse);
}

orc_tables *orc_backward(const orc_model *m, const orc_params *p, const uint8_t *read,
                         uint64_t len, int mode, const orc_mapping_view *mp,
                         const orc_tables *fwd) {
    g_err[0] = 0;
    t_overflow = 0;
    if (mode == ORC_BWD_MAPPING && mapping_check(m, mp, len)) return NULL;
    if (mode == ORC_BWD_BY_FORWARD && (!fwd || fwd->n != (int64_t)len)) {
        set_err("backward_by_forward: forward tables required");
        return NULL;
    }
    ctx_t c;
    ctx_init(&c, m, p);
    orc_tables *ts = tables_new(m->N, 1, (int64_t)len);
    b_init(&c, &ts->init);
    uint32_t *alln = all_nodes(m->N);
    for (uint64_t r = 0; r < len; r++) {
        uint64_t i = len - 1 - r;
        const table_t *prev = (i == len - 1) ? &ts->init : &ts->t[i + 1];
        switch (mode) {
        case ORC_BWD_DENSE:
            b_step(&c, read[i], prev, alln, m->N, 1, 0, &ts->t[i]);
            break;
        case ORC_BWD_MAPPING:
            b_step(&c, read[i], prev, mp->nodes + mp->pos_off[i],
                   (int64_t)(mp->pos_off[i + 1] - mp->pos_off[i]), 0, 0, &ts->t[i]);
            break;
        case ORC_BWD_BY_FORWARD:
            if (i == 0 || fwd->t[i - 1].m.dense)
                b_step(&c, read[i], prev, alln, m->N, 1, 0, &ts->t[i]);
            else {
                uint32_t act[CAP];
                int nact = tb_filled_nodes(&fwd->t[i - 1], act);
                b_step(&c, read[i], prev, act, nact, 0, 0, &ts->t[i]);
            }
            break;
        case ORC_BWD_SPARSE:
            if ((int64_t)(len - i - 1) < p->n_warmup)
                b_step(&c, read[i], prev, alln, m->N, 1, 0, &ts->t[i]);
            else {
                uint32_t act[CAP];
                int nact = tb_top_nodes(prev, (int)p->n_active_nodes, act, NULL);
                b_step(&c, read[i], prev, act, nact, 0, 1, &ts->t[i]);
            }
            break;
        default:
            set_err("orc_backward: bad mode");
            t_overflow = 1;
        }
    }
    free(alln);
    ctx_free(&c);
    if (t_overflow) {
        if (!g_err[0]) set_err("SparseVec capacity (400) exceeded: the reference would panic");
        orc_tables_destroy(ts);
        return NULL;
    }
    return ts;
}

/* ------------------------------------------------------------------ accessors */
int64_t orc_tables_len(const orc_tables *t) { return t->n; }
static const table_t *tb_at(const orc_tables *t, int64_t i) {
    if (i == -1) return &t->init;
    if (i < 0 || i >= t->n) return NULL;
    return &t->t[i];
}
int orc_tables_is_dense(const orc_tables *t, int64_t i) {
    const table_t *tb = tb_at(t, i);
    return tb ? tb->m.dense : -1;
}
static void nv_expand(const nvec *v, double *out) {
    if (v->dense) memcpy(out, v->dv, sizeof(double) * v->len);
    else {
        for (uint32_t k = 0; k < v->len; k++) out[k] = v->dflt;
        for (int j = 0; j < v->n; j++) out[v->idx[j]] = v->val[j];
    }
}
int orc_tables_get(const orc_tables *t, int64_t i, double *m, double *ins, double *d, double *scal) {
    const table_t *tb = tb_at(t, i);
    if (!tb) { set_err("table index out of range"); return -1; }
    if (m) nv_expand(&tb->m, m);
    if (ins) nv_expand(&tb->i, ins);
    if (d) nv_expand(&tb->d, d);
    if (scal) { scal[0] = tb->mb; scal[1] = tb->ib; scal[2] = tb->e; }
    return 0;
}
int64_t orc_tables_nodes(const orc_tables *t, int64_t i, int which, uint32_t *idx) {
    const table_t *tb = tb_at(t, i);
    if (!tb) return -1;
    const nvec *v = which == 0 ? &tb->m : (which == 1 ? &tb->i : &tb->d);
    int n = nv_count(v);
    if (idx) for (int j = 0; j < n; j++) idx[j] = nv_idx_at(v, j);
    return n;
}
/* table.rs:395-401 */
double orc_tables_full_prob(const orc_tables *t) {
    if (t->n == 0) return NAN; /* the reference panics */
    return t->kind == 0 ? t->t[t->n - 1].e : t->t[0].mb;
}

/* ------------------------------------------------------------------ PHMMOutput */

/* table.rs:414-434 table_merged */
static const table_t *merged(const orc_tables *ts, int64_t j) {
    if (ts->kind == 0) return j == 0 ? &ts->init : &ts->t[j - 1];
    return j >= ts->n ? &ts->init : &ts->t[j];
}
/* `&a * &b` on SparseVec (table.rs:320-331): dense iff both dense; otherwise the
 * stored elements of the sparse operand (self first).  Then `/ p` (table.rs:333-345). */
static void nv_mul_div(const nvec *a, const nvec *b, double p, nvec *out) {
    if (a->dense && b->dense) {
        nv_init(out, a->len, a->dflt + b->dflt - p, 1);
        for (uint32_t k = 0; k < a->len; k++) out->dv[k] = a->dv[k] + b->dv[k] - p;
        return;
    }
    const nvec *s = !a->dense ? a : b;
    nv_init(out, a->len, a->dflt + b->dflt - p, 0);
    for (int j = 0; j < s->n; j++) {
        uint32_t k = s->idx[j];
        *nv_ref(out, k) = nv_get(a, k) + nv_get(b, k) - p;
    }
}
/* table.rs:500-505 to_emit_probs */
static void emit_probs(const orc_tables *f, const orc_tables *b, int64_t j, table_t *out) {
    double p = f->t[f->n - 1].e;
    const table_t *tf = merged(f, j), *tbk = merged(b, j);
    nv_mul_div(&tf->m, &tbk->m, p, &out->m);
    nv_mul_div(&tf->i, &tbk->i, p, &out->i);
    nv_mul_div(&tf->d, &tbk->d, p, &out->d);
    out->mb = tf->mb + tbk->mb - p;
    out->ib = tf->ib + tbk->ib - p;
    out->e = tf->e + tbk->e - p;
}
int orc_emit_probs(const orc_tables *f, const orc_tables *b, int64_t j, double *m, double *ins,
                   double *d, double *scal) {
    if (f->n == 0 || f->n != b->n || j < 0 || j > f->n) { set_err("emit_probs: bad index"); return -1; }
    t_overflow = 0;
    table_t t;
    emit_probs(f, b, j, &t);
    if (m) nv_expand(&t.m, m);
    if (ins) nv_expand(&t.i, ins);
    if (d) nv_expand(&t.d, d);
    if (scal) { scal[0] = t.mb; scal[1] = t.ib; scal[2] = t.e; }
    tb_free(&t);
    return t_overflow ? -1 : 0;
}
/* freq.rs:236-255: state_probs = sum_j emit_probs(j) (table.rs:347-357, 307-318);
 * node_freq[v] = exp(m+i+d). */
int orc_node_freqs(const orc_tables *f, const orc_tables *b, double *out) {
    if (f->n == 0 || f->n != b->n) { set_err("node_freqs: empty or mismatched tables"); return -1; }
    t_overflow = 0;
    table_t acc, t;
    emit_probs(f, b, 0, &acc);
    for (int64_t j = 1; j <= f->n; j++) {
        emit_probs(f, b, j, &t);
        nv_add_assign(&acc.m, &t.m);
        nv_add_assign(&acc.i, &t.i);
        nv_add_assign(&acc.d, &t.d);
        tb_free(&t);
    }
    nvec v;
    tb_to_nodevec(&acc, &v);
    for (uint32_t k = 0; k < f->N; k++) out[k] = 0.0;
    int n = nv_count(&v);
    for (int j = 0; j < n; j++) out[nv_idx_at(&v, j)] = exp(nv_val_at(&v, j));
    nv_free(&v);
    tb_free(&acc);
    if (t_overflow) { set_err("SparseVec capacity (400) exceeded in to_state_probs"); return -1; }
    return 0;
}
/* hint.rs:124-142 */
int orc_output_mapping(const orc_tables *f, const orc_tables *b, int by_ratio, int64_t n_active,
                       double max_ratio, uint64_t *pos_off, uint32_t *nodes, double *logp) {
    if (f->n == 0 || f->n != b->n) { set_err("mapping: empty or mismatched tables"); return -1; }
    t_overflow = 0;
    uint64_t w = 0;
    pos_off[0] = 0;
    for (int64_t j = 1; j <= f->n; j++) {
        table_t t;
        emit_probs(f, b, j, &t);
        int r = by_ratio ? tb_top_nodes_by_score_ratio(&t, max_ratio, nodes + w, logp + w)
                         : tb_top_nodes(&t, (int)n_active, nodes + w, logp + w);
        tb_free(&t);
        w += (uint64_t)r;
        pos_off[j] = w;
    }
    return t_overflow ? -1 : 0;
}

/* freq.rs:332-389 */
int orc_trans_and_init_probs(const orc_model *m, const orc_params *p, const orc_tables *f,
                             const orc_tables *b, const uint8_t *read, uint64_t len,
                             uint64_t i, double *tp, double *ip) {
    if ((int64_t)len != f->n || (int64_t)len != b->n || i > len || len == 0) {
        set_err("trans_probs: bad arguments");
        return -1;
    }
    double P = f->t[f->n - 1].e;
    const table_t *fi0 = merged(f, (int64_t)i), *bi2 = merged(b, (int64_t)i + 1), *bi1 = merged(b, (int64_t)i);
    for (uint32_t e = 0; e < m->E; e++) {
        double *t = tp + 6 * (size_t)e;
        for (int q = 0; q < 6; q++) t[q] = NEG_INF;
        uint32_t k = m->esrc[e], l = m->edst[e];
        double pt = m->trans[e];
        if (i < len) {
            double pe = m->emission[l] == read[i] ? p->p_match : p->p_mismatch;
            t[0] = nv_get(&fi0->m, k) + pt + p->p_MM + pe + nv_get(&bi2->m, l) - P;
            t[1] = nv_get(&fi0->i, k) + pt + p->p_IM + pe + nv_get(&bi2->m, l) - P;
            t[2] = nv_get(&fi0->d, k) + pt + p->p_DM + pe + nv_get(&bi2->m, l) - P;
        }
        t[3] = nv_get(&fi0->m, k) + pt + p->p_MD + nv_get(&bi1->d, l) - P;
        t[4] = nv_get(&fi0->i, k) + pt + p->p_ID + nv_get(&bi1->d, l) - P;
        t[5] = nv_get(&fi0->d, k) + pt + p->p_DD + nv_get(&bi1->d, l) - P;
    }
    for (uint32_t v = 0; v < m->N; v++) {
        double *t = ip + 6 * (size_t)v;
        for (int q = 0; q < 6; q++) t[q] = NEG_INF;
        if (i < len) {
            double pe = m->emission[v] == read[i] ? p->p_match : p->p_mismatch;
            t[0] = fi0->mb + m->init[v] + p->p_MM + pe + nv_get(&bi2->m, v) - P;
            t[1] = fi0->ib + m->init[v] + p->p_IM + pe + nv_get(&bi2->m, v) - P;
        }
        t[3] = fi0->mb + m->init[v] + p->p_MD + nv_get(&bi1->d, v) - P;
        t[4] = fi0->ib + m->init[v] + p->p_ID + nv_get(&bi1->d, v) - P;
    }
    return 0;
}
/* trans_table.rs TransProb::sum: mm+im+dm+md+id+dd */
static double tp_sum(const double *t) {
    return LADD(LADD(LADD(LADD(LADD(t[0], t[1]), t[2]), t[3]), t[4]), t[5]);
}
/* freq.rs:276-298 */
int orc_edge_and_init_freqs(const orc_model *m, const orc_params *p, const orc_tables *f,
                            const orc_tables *b, const uint8_t *read, uint64_t len,
                            double *ef, double *inf_) {
    double *tp = malloc(sizeof(double) * 6 * (m->E ? m->E : 1));
    double *ip = malloc(sizeof(double) * 6 * (m->N ? m->N : 1));
    for (uint32_t e = 0; e < m->E; e++) ef[e] = 0.0;
    for (uint32_t v = 0; v < m->N; v++) inf_[v] = 0.0;
    int rc = 0;
    for (uint64_t i = 0; i <= len && !rc; i++) {
        rc = orc_trans_and_init_probs(m, p, f, b, read, len, i, tp, ip);
        if (rc) break;
        for (uint32_t e = 0; e < m->E; e++) ef[e] += exp(tp_sum(tp + 6 * (size_t)e));
        for (uint32_t v = 0; v < m->N; v++) inf_[v] += exp(tp_sum(ip + 6 * (size_t)v));
    }
    free(tp);
    free(ip);
    return rc;
}

/* ------------------------------------------------------------------ read-set drivers */

static int resolve_threads(int n_threads) {
#ifdef _OPENMP
    return n_threads > 0 ? n_threads : omp_get_max_threads();
#else
    (void)n_threads;
    return 1;
#endif
}

int orc_full_prob_reads(const orc_model *m, const orc_params *p, const uint8_t *bases,
                        const uint64_t *roff, uint64_t R, const uint64_t *mpo,
                        const uint32_t *mnodes, const double *mlogp, int use_max_ratio,
                        int n_threads, double *out) {
    int fail = 0;
    int nt = resolve_threads(n_threads);
    (void)nt;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (int64_t r = 0; r < (int64_t)R; r++) {
        uint64_t b0 = roff[r], len = roff[r + 1] - roff[r];
        orc_mapping_view mv, *pmv = NULL;
        if (mpo) {
            mv.pos_off = mpo + b0;
            mv.nodes = mnodes;
            mv.logp = mlogp;
            pmv = &mv;
        }
        if (orc_forward_score_only(m, p, bases + b0, len, pmv, use_max_ratio, &out[r])) {
#pragma omp atomic write
            fail = 1;
        }
    }
    return fail ? -1 : 0;
}

struct orc_mappings {
    uint64_t R, total_pos;
    uint64_t *read_pos0; /* [R+1] */
    uint64_t **pos_off;  /* per read [L+1] local */
    uint32_t **nodes;
    double **logp;
};

orc_mappings *orc_generate_mappings(const orc_model *m, const orc_params *p, const uint8_t *bases,
                                    const uint64_t *roff, uint64_t R, const uint64_t *mpo,
                                    const uint32_t *mnodes, const double *mlogp,
                                    int use_max_ratio, int n_threads) {
    orc_mappings *mp = calloc(1, sizeof *mp);
    mp->R = R;
    mp->read_pos0 = calloc(R + 1, sizeof(uint64_t));
    mp->pos_off = calloc(R ? R : 1, sizeof(uint64_t *));
    mp->nodes = calloc(R ? R : 1, sizeof(uint32_t *));
    mp->logp = calloc(R ? R : 1, sizeof(double *));
    for (uint64_t r = 0; r < R; r++) mp->read_pos0[r + 1] = mp->read_pos0[r] + (roff[r + 1] - roff[r]);
    mp->total_pos = mp->read_pos0[R];
    int fail = 0;
    int nt = resolve_threads(n_threads);
    (void)nt;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (int64_t r = 0; r < (int64_t)R; r++) {
        uint64_t b0 = roff[r], len = roff[r + 1] - roff[r];
        orc_tables *f = NULL, *b = NULL;
        if (len == 0) {
#pragma omp atomic write
            fail = 1;
            continue;
        }
        if (mpo) { /* run_with_mapping freq.rs:72-76 */
            orc_mapping_view mv = {mpo + b0, mnodes, mlogp};
            f = orc_forward(m, p, bases + b0, len, ORC_FWD_MAPPING, &mv);
            if (f) b = orc_backward(m, p, bases + b0, len, ORC_BWD_MAPPING, &mv, NULL);
        } else { /* run_sparse_adaptive freq.rs:60-68 */
            f = orc_forward(m, p, bases + b0, len, use_max_ratio ? ORC_FWD_SPARSE_RATIO : ORC_FWD_SPARSE_TOPK, NULL);
            if (f) b = orc_backward(m, p, bases + b0, len, ORC_BWD_BY_FORWARD, NULL, f);
        }
        if (!f || !b) {
#pragma omp atomic write
            fail = 1;
            orc_tables_destroy(f);
            orc_tables_destroy(b);
            continue;
        }
        uint64_t *po = malloc(sizeof(uint64_t) * (len + 1));
        uint32_t *nd = malloc(sizeof(uint32_t) * len * CAP);
        double *lp = malloc(sizeof(double) * len * CAP);
        if (orc_output_mapping(f, b, use_max_ratio, p->n_active_nodes, p->active_node_max_ratio, po, nd, lp)) {
#pragma omp atomic write
            fail = 1;
        }
        uint64_t tot = po[len];
        mp->pos_off[r] = po;
        mp->nodes[r] = realloc(nd, sizeof(uint32_t) * (tot ? tot : 1));
        mp->logp[r] = realloc(lp, sizeof(double) * (tot ? tot : 1));
        orc_tables_destroy(f);
        orc_tables_destroy(b);
    }
    if (fail) {
        if (!g_err_shared[0]) set_err("generate_mappings failed");
        orc_mappings_destroy(mp);
        return NULL;
    }
    return mp;
}
uint64_t orc_mappings_total_positions(const orc_mappings *mp) { return mp->total_pos; }
uint64_t orc_mappings_total_entries(const orc_mappings *mp) {
    uint64_t t = 0;
    for (uint64_t r = 0; r < mp->R; r++) {
        uint64_t len = mp->read_pos0[r + 1] - mp->read_pos0[r];
        t += mp->pos_off[r][len];
    }
    return t;
}
void orc_mappings_export(const orc_mappings *mp, uint64_t *pos_off, uint32_t *nodes, double *logp) {
    uint64_t w = 0;
    pos_off[0] = 0;
    for (uint64_t r = 0; r < mp->R; r++) {
        uint64_t len = mp->read_pos0[r + 1] - mp->read_pos0[r], p0 = mp->read_pos0[r];
        for (uint64_t i = 0; i < len; i++) pos_off[p0 + i + 1] = w + mp->pos_off[r][i + 1];
        uint64_t tot = mp->pos_off[r][len];
        memcpy(nodes + w, mp->nodes[r], sizeof(uint32_t) * tot);
        memcpy(logp + w, mp->logp[r], sizeof(double) * tot);
        w += tot;
    }
}
/* hint.rs:161-171 */
void orc_mappings_node_freqs(const orc_mappings *mp, uint32_t N, double *out) {
    for (uint32_t v = 0; v < N; v++) out[v] = 0.0;
    for (uint64_t r = 0; r < mp->R; r++) {
        uint64_t len = mp->read_pos0[r + 1] - mp->read_pos0[r];
        uint64_t tot = mp->pos_off[r][len];
        for (uint64_t a = 0; a < tot; a++) out[mp->nodes[r][a]] += exp(mp->logp[r][a]);
    }
}
void orc_mappings_destroy(orc_mappings *mp) {
    if (!mp) return;
    for (uint64_t r = 0; r < mp->R; r++) {
        free(mp->pos_off[r]); free(mp->nodes[r]); free(mp->logp[r]);
    }
    free(mp->pos_off); free(mp->nodes); free(mp->logp); free(mp->read_pos0);
    free(mp);
}

int orc_run_dense_reads(const orc_model *m, const orc_params *p, const uint8_t *bases,
                        const uint64_t *roff, uint64_t R, int n_threads, double *lf, double *lb,
                        double *nf) {
    int fail = 0;
    int nt = resolve_threads(n_threads);
    double *acc = calloc((size_t)nt * (m->N ? m->N : 1), sizeof(double));
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        double *mine = acc + (size_t)tid * m->N;
        double *tmp = malloc(sizeof(double) * (m->N ? m->N : 1));
#pragma omp for schedule(dynamic, 1)
        for (int64_t r = 0; r < (int64_t)R; r++) {
            uint64_t b0 = roff[r], len = roff[r + 1] - roff[r];
            orc_tables *f = len ? orc_forward(m, p, bases + b0, len, ORC_FWD_DENSE, NULL) : NULL;
            orc_tables *b = f ? orc_backward(m, p, bases + b0, len, ORC_BWD_DENSE, NULL, NULL) : NULL;
            if (!f || !b || orc_node_freqs(f, b, tmp)) {
#pragma omp atomic write
                fail = 1;
            } else {
                if (lf) lf[r] = orc_tables_full_prob(f);
                if (lb) lb[r] = orc_tables_full_prob(b);
                for (uint32_t v = 0; v < m->N; v++) mine[v] += tmp[v];
            }
            orc_tables_destroy(f);
            orc_tables_destroy(b);
        }
        free(tmp);
    }
    if (nf) {
        for (uint32_t v = 0; v < m->N; v++) {
            double s = 0.0;
            for (int t = 0; t < nt; t++) s += acc[(size_t)t * m->N + v];
            nf[v] = s;
        }
    }
    free(acc);
    return fail ? -1 : 0;
}
