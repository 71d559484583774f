// extern "C" surface of libphmm_amd.so (include/phmm_amd.h).  Argument validation mirrors
// the reference's asserts/panics; nothing throws across the ABI.
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstring>

#include "phmm_internal.h"

namespace phmm {

static thread_local std::string g_error;
static thread_local hipStream_t g_stream = nullptr;
static thread_local uint64_t g_ws_limit = 0;
static thread_local CallStats g_stats;
static thread_local bool g_timing = false;

void set_error(const std::string &msg) { g_error = msg; }
int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
hipStream_t current_stream() { return g_stream; }
int &workset_index() {
    static thread_local int idx = 0;
    return idx;
}
ThreadContext capture_thread_context() {
    ThreadContext c{g_ws_limit, g_timing, 0};
    (void)hipGetDevice(&c.device);
    return c;
}
void adopt_thread_context(const ThreadContext &c, hipStream_t stream, int workset) {
    (void)hipSetDevice(c.device);
    g_ws_limit = c.ws_limit;
    g_timing = c.timing;
    g_stream = stream;
    g_stats = CallStats();
    workset_index() = workset;
}
CallStats &stats() { return g_stats; }
bool timing_enabled() { return g_timing; }

uint64_t workspace_limit() {
    if (g_ws_limit) return g_ws_limit;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return (uint64_t)16 << 30;
    return (uint64_t)(0.8 * (double)fr);
}

static Knobs g_knobs;
const Knobs &knobs() { return g_knobs; }
void refresh_knobs() {
    auto flag = [](const char *n) {  // set, and not "" / "0"
        const char *e = std::getenv(n);
        return e != nullptr && e[0] != '\0' && !(e[0] == '0' && e[1] == '\0');
    };
    auto num = [](const char *n, int dflt) {
        const char *e = std::getenv(n);
        return e ? std::atoi(e) : dflt;
    };
    Knobs k;
    k.trace = flag("PHMM_TRACE");
    k.no_lean = flag("PHMM_NO_LEAN");
    k.no_wide_class = flag("PHMM_NO_WIDE_CLASS");
    k.no_packed = flag("PHMM_NO_PACKED");
    k.packed_cpl = num("PHMM_PACKED_CPL", 0);
    k.no_exact_hinted = flag("PHMM_NO_EXACT_HINTED");
    k.no_side_worker = flag("PHMM_NO_SIDE_WORKER");
    k.no_wide_handover = flag("PHMM_NO_WIDE_HANDOVER");
    k.workers = std::max(1, std::min(MAX_WORKERS, num("PHMM_WORKERS", 1)));
    k.warm_cols = num("PHMM_WARM_COLS", 0);
    k.chunk_groups = std::max(0, num("PHMM_CHUNK_GROUPS", 0));
    k.no_keep_all = flag("PHMM_NO_KEEP_ALL");
    k.pipeline_min_groups = std::max(1, num("PHMM_PIPELINE_MIN_GROUPS", 8));
    k.no_runmax = flag("PHMM_NO_RUNMAX");
    k.force_radix = flag("PHMM_FORCE_RADIX");
    k.serial_emit = flag("PHMM_SERIAL_EMIT");
    k.emit_high_priority = flag("PHMM_EMIT_HIGH_PRIORITY");
    k.no_dma = flag("PHMM_NO_DMA");
    k.bwd_dma = flag("PHMM_BWD_DMA");
    k.dense_streams = num("PHMM_DENSE_STREAMS", 0);
    k.dense_w = num("PHMM_DENSE_W", 0);
    k.dense_npt = num("PHMM_DENSE_NPT", 0);
    if (const char *e = std::getenv("PHMM_MEM_FRACTION")) k.mem_fraction = std::min(0.97, std::max(0.1, std::atof(e)));
    g_knobs = k;
}

void trace(const char *tag) {
    if (!g_knobs.trace) return;
    static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    (void)hipStreamSynchronize(g_stream);
    const auto now = std::chrono::steady_clock::now();
    static const auto t0 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[phmm trace] w%d @%9.2f %-28s %8.2f ms\n", workset_index(),
                 std::chrono::duration<double, std::milli>(now - t0).count(), tag,
                 std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
}

uint64_t table_budget(const DevicePool &pool, uint64_t reserve) {
    if (g_ws_limit) return g_ws_limit;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return (uint64_t)16 << 30;
    const double frac = g_knobs.mem_fraction > 0.0 ? g_knobs.mem_fraction : 0.9;
    const double b = frac * (double)(fr + pool.owned_table_bytes()) - (double)reserve;
    return (uint64_t)std::max(b, 64.0 * 1024 * 1024);
}
// The adaptive flow plans ALL of its memory (tables, emit-prob planes, record pools, control arrays): everything the
// pool already holds is reusable by it, so the plan starts from PHMM_MEM_FRACTION (0.95 here: nothing else is left to
// grow beside the plan) of free + pool-owned bytes.
uint64_t planned_budget(const DevicePool &pool) {
    if (g_ws_limit) return g_ws_limit;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return (uint64_t)16 << 30;
    const double frac = g_knobs.mem_fraction > 0.0 ? g_knobs.mem_fraction : 0.95;
    return (uint64_t)(frac * (double)(fr + pool.owned_bytes()));
}

// ---------------------------------------------------------------- per-device workspace pool
static std::mutex g_pool_mu;
static DevicePool *g_pools[64] = {};
DevicePool &device_pool() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) PHMM_THROW(PHMM_ENODEVICE, "device index out of range");
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!g_pools[dev]) {
        g_pools[dev] = new DevicePool();  // never destroyed: the HIP runtime may be gone by static-destructor time
        g_pools[dev]->device = dev;
    }
    return *g_pools[dev];
}
void DevicePool::recycle(DevBuf &b) {
    if (!b.p) return;
    std::lock_guard<std::mutex> lk(spare_mu);
    if (b.bytes < ((size_t)1 << 20) || b.bytes > SPARE_BYTES) {
        b.release();
        return;
    }
    // the newest buffers stay (the next call most likely wants what the last one produced): the oldest go first
    while (!spare.empty() && (spare.size() >= SPARE_MAX || spare_total + b.bytes > SPARE_BYTES)) {
        (void)hipFree(spare.front().first);
        spare_total -= spare.front().second;
        spare.erase(spare.begin());
    }
    spare_total += b.bytes;
    const size_t n = b.bytes;
    spare.emplace_back(b.detach(), n);
}
void DevicePool::take(DevBuf &b, size_t n) {
    if (n <= b.bytes) return;
    {
        std::lock_guard<std::mutex> lk(spare_mu);
        int best = -1;
        for (int i = 0; i < (int)spare.size(); i++)
            if (spare[i].second >= n && spare[i].second <= 2 * n &&
                (best < 0 || spare[i].second < spare[best].second))
                best = i;
        if (best >= 0) {
            b.adopt(spare[best].first, spare[best].second);
            spare_total -= spare[best].second;
            spare.erase(spare.begin() + best);
            return;
        }
    }
    b.reserve(n);
}
void DevicePool::flush_spares() {
    std::lock_guard<std::mutex> lk(spare_mu);
    for (auto &e : spare) (void)hipFree(e.first);
    spare.clear();
    spare_total = 0;
}
void flush_spare_buffers() { device_pool().flush_spares(); }

void DevicePool::release() {
    flush_spares();
    for (auto &w : wsets) w.release();
    ws_out.release();
    for (auto &ws : wstream)
        if (ws) {
            (void)hipStreamDestroy(ws);
            ws = nullptr;
        }
    for (auto &ws : cstream)
        if (ws) {
            (void)hipStreamDestroy(ws);
            ws = nullptr;
        }
    for (auto &we : cevent)
        for (auto &e : we)
            if (e) {
                (void)hipEventDestroy(e);
                e = nullptr;
            }
}

void copy_out(void *dst, const void *src_dev, size_t bytes) {
    if (!dst || !bytes) return;
    hipPointerAttribute_t at;
    const bool dev = hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeDevice;
    if (!dev) (void)hipGetLastError();
    HIP_CHECK(hipMemcpyAsync(dst, src_dev, bytes, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                             current_stream()));
    HIP_CHECK(hipStreamSynchronize(current_stream()));
}

void put_doubles(double *dst, const double *src, size_t n) {
    if (!dst || !n) return;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeDevice) {
        HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, current_stream()));
        HIP_CHECK(hipStreamSynchronize(current_stream()));
    } else {
        (void)hipGetLastError();
        std::memcpy(dst, src, n * sizeof(double));
    }
}

// device-resident mappings -> host vectors (first host access only)
void mappings_materialize_host(const phmm_mappings *cmp) {
    if (cmp->host_valid) return;
    auto *mp = const_cast<phmm_mappings *>(cmp);
    mp->pos_off.resize(mp->total_pos + 1);
    mp->nodes.resize(mp->total_entries);
    mp->logp.resize(mp->total_entries);
    HIP_CHECK(hipMemcpy(mp->pos_off.data(), mp->d_pos_off.p, sizeof(uint64_t) * (mp->total_pos + 1), hipMemcpyDeviceToHost));
    if (mp->total_entries) {
        HIP_CHECK(hipMemcpy(mp->nodes.data(), mp->d_nodes.p, sizeof(uint32_t) * mp->total_entries, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(mp->logp.data(), mp->d_logp.p, sizeof(double) * mp->total_entries, hipMemcpyDeviceToHost));
    }
    mp->host_valid = true;
}

template <class F> static int guarded(F &&f) {
    try {
        f();
        return PHMM_OK;
    } catch (const Error &e) {
        return fail(e.code, e.msg);
    } catch (const std::bad_alloc &) {
        return fail(PHMM_ENOMEM, "host allocation failed");
    } catch (const std::exception &e) {
        return fail(PHMM_EINTERNAL, e.what());
    } catch (...) {
        return fail(PHMM_EINTERNAL, "unknown error");
    }
}

// compute entry points: one call at a time per device (the workspaces are the device's), on the model's device
template <class F> static int guarded_on(phmm_model *m, F &&f) {
    if (!m || !m->pool) return fail(PHMM_EINVAL, "NULL model");
    std::lock_guard<std::recursive_mutex> lk(m->pool->call_mu);
    refresh_knobs();
    return guarded([&] {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        if (dev != m->pool->device)
            PHMM_THROW(PHMM_EINVAL, "the model lives on device " + std::to_string(m->pool->device) +
                                        " but this thread's device is " + std::to_string(dev) + " (phmm_set_device)");
        f();
    });
}

static void check_params(const phmm_params *p) {
    if (!p) PHMM_THROW(PHMM_EINVAL, "params is NULL");
    // params.rs:81-83
    if (p->n_active_nodes <= 0) PHMM_THROW(PHMM_EINVAL, "n_active_nodes must be > 0");
    if (p->n_warmup <= 0) PHMM_THROW(PHMM_EINVAL, "n_warmup must be > 0");
    if (p->n_active_nodes >= PHMM_MAX_ACTIVE_NODES)
        PHMM_THROW(PHMM_EINVAL, "n_active_nodes must be < MAX_ACTIVE_NODES (400)");
    if (p->n_max_gaps < 0 || p->n_max_gaps > PHMM_MAX_GAPS)
        PHMM_THROW(PHMM_EINVAL, "n_max_gaps out of the supported range [0, 6]");
    const double *lp = &p->p_mismatch;
    for (int i = 0; i < 15; i++)
        if (std::isnan(lp[i]) || lp[i] > 1e-12) PHMM_THROW(PHMM_EINVAL, "params: log-probability > 0 or NaN");
}

static void require_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        PHMM_THROW(PHMM_ENODEVICE, "no HIP device: the MI355X path has no CPU fallback");
    }
}

}  // namespace phmm

using namespace phmm;

phmm::WorkSet &phmm_model::wset() { return pool->wsets[phmm::workset_index()]; }

extern "C" {

const char *phmm_last_error(void) { return g_error.c_str(); }
const char *phmm_version(void) { return "dbgphmm_amd 0.1.0 gfx950"; }

int phmm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
int phmm_set_device(int device) {
    return guarded([&] {
        require_device();
        HIP_CHECK(hipSetDevice(device));
    });
}
int phmm_set_stream(void *s) {
    g_stream = (hipStream_t)s;
    return PHMM_OK;
}
int phmm_set_workspace_limit(uint64_t bytes) {
    g_ws_limit = bytes;
    return PHMM_OK;
}
int phmm_enable_timing(int on) {
    g_timing = on != 0;
    return PHMM_OK;
}
int phmm_last_call_stats(int which, double *ms, uint64_t *launches, uint64_t *cells) {
    if (which < 0 || which > 3) return fail(PHMM_EINVAL, "which out of range");
    if (ms) *ms = g_stats.ms[which];
    if (launches) *launches = g_stats.launches[which];
    if (cells) *cells = g_stats.cells[which];
    return PHMM_OK;
}

static double lnz(double x) { return x > 0.0 ? std::log(x) : -INFINITY; }

int phmm_params_new(double p_mismatch, double p_gap_open, double p_gap_ext, double p_end,
                    int64_t n_active_nodes, int64_t n_warmup, phmm_params *o) {
    return guarded([&] {
        if (!o) PHMM_THROW(PHMM_EINVAL, "out is NULL");
        // params.rs:73-113
        o->p_mismatch = lnz(p_mismatch);
        o->p_gap_open = lnz(p_gap_open);
        o->p_gap_ext = lnz(p_gap_ext);
        o->p_end = lnz(p_end);
        o->p_DD = o->p_II = o->p_gap_ext;
        o->p_MI = o->p_MD = o->p_ID = o->p_DI = o->p_gap_open;
        const double go = std::exp(o->p_gap_open), ge = std::exp(o->p_gap_ext), pe = std::exp(o->p_end);
        o->p_MM = lnz(1.0 - 2.0 * go - pe);
        o->p_DM = o->p_IM = lnz(1.0 - go - ge - pe);
        o->p_match = lnz(1.0 - std::exp(o->p_mismatch));
        o->p_random = std::log(0.25);
        o->n_active_nodes = n_active_nodes;
        o->active_node_max_ratio = 30.0;
        o->n_warmup = n_warmup;
        o->n_max_gaps = 4;
        o->warmup_threshold = PHMM_MAX_ACTIVE_NODES / 2;
        check_params(o);
    });
}
int phmm_params_uniform(double p, phmm_params *o) { return phmm_params_new(p, p, p, 0.00001, 40, 50, o); }

int phmm_model_create(uint32_t N, uint32_t E, const uint8_t *emission, const double *init_logp,
                      const uint32_t *esrc, const uint32_t *edst, const double *trans_logp,
                      const phmm_params *params, phmm_model **out) {
    phmm_model *m = nullptr;
    int rc = guarded([&] {
        if (!out) PHMM_THROW(PHMM_EINVAL, "out is NULL");
        *out = nullptr;
        if (N == 0) PHMM_THROW(PHMM_EINVAL, "model has no nodes");
        if (!emission || !init_logp || (E && (!esrc || !edst || !trans_logp)))
            PHMM_THROW(PHMM_EINVAL, "NULL model array");
        check_params(params);
        for (uint32_t e = 0; e < E; e++)
            if (esrc[e] >= N || edst[e] >= N) PHMM_THROW(PHMM_EINVAL, "edge endpoint out of range");
        for (uint32_t v = 0; v < N; v++)
            if (std::isnan(init_logp[v]) || init_logp[v] > 1e-9) PHMM_THROW(PHMM_EINVAL, "init_logp > 0 or NaN");
        for (uint32_t e = 0; e < E; e++)
            if (std::isnan(trans_logp[e]) || trans_logp[e] > 1e-9) PHMM_THROW(PHMM_EINVAL, "trans_logp > 0 or NaN");
        require_device();
        m = new phmm_model();
        m->pool = &device_pool();
        m->N = N;
        m->E = E;
        m->params = *params;
        m->emission.assign(emission, emission + N);
        m->init_logp.assign(init_logp, init_logp + N);
        m->esrc.assign(esrc, esrc + E);
        m->edst.assign(edst, edst + E);
        m->trans_logp.assign(trans_logp, trans_logp + E);
        model_build_host(m);
        model_upload(m);
        *out = m;
    });
    if (rc != PHMM_OK) delete m;
    return rc;
}

int phmm_model_set_probs(phmm_model *m, const double *init_logp, const double *trans_logp) {
    return guarded_on(m, [&] {
        if (!m || !init_logp || (m->E && !trans_logp)) PHMM_THROW(PHMM_EINVAL, "NULL argument");
        m->init_logp.assign(init_logp, init_logp + m->N);
        m->trans_logp.assign(trans_logp, trans_logp + m->E);
        model_upload(m);
    });
}
int phmm_model_set_params(phmm_model *m, const phmm_params *params) {
    return guarded_on(m, [&] {
        if (!m) PHMM_THROW(PHMM_EINVAL, "NULL model");
        check_params(params);
        m->params = *params;
        model_build_host(m);
        model_upload(m);
    });
}
uint32_t phmm_model_n_nodes(const phmm_model *m) { return m ? m->N : 0; }
uint32_t phmm_model_n_edges(const phmm_model *m) { return m ? m->E : 0; }
void phmm_model_destroy(phmm_model *m) { delete m; }  // (the workspaces belong to the device: phmm_release_workspace)

int phmm_release_workspace(void) {
    return guarded([&] {
        require_device();
        DevicePool &pool = device_pool();
        std::lock_guard<std::recursive_mutex> lk(pool.call_mu);
        HIP_CHECK(hipDeviceSynchronize());
        pool.release();
    });
}
uint64_t phmm_workspace_bytes(void) {
    if (phmm_device_count() <= 0) return 0;
    uint64_t b = 0;
    (void)guarded([&] { b = device_pool().owned_bytes(); });
    return b;
}

int phmm_reads_create(const uint8_t *bases, const uint64_t *offsets, uint64_t R, phmm_reads **out) {
    phmm_reads *r = nullptr;
    int rc = guarded([&] {
        if (!out) PHMM_THROW(PHMM_EINVAL, "out is NULL");
        *out = nullptr;
        if (!offsets || (R && !bases)) PHMM_THROW(PHMM_EINVAL, "NULL argument");
        if (offsets[0] != 0) PHMM_THROW(PHMM_EINVAL, "offsets[0] must be 0");
        r = new phmm_reads();
        r->R = R;
        r->off.assign(offsets, offsets + R + 1);
        for (uint64_t i = 0; i < R; i++) {
            if (offsets[i + 1] <= offsets[i])
                PHMM_THROW(PHMM_EINVAL, "empty read (the reference panics in last_table()) or offsets not increasing");
            r->max_len = std::max<uint64_t>(r->max_len, offsets[i + 1] - offsets[i]);
        }
        if (r->max_len > (uint64_t)1 << 30) PHMM_THROW(PHMM_EINVAL, "read longer than 2^30 bases");
        r->total = offsets[R];
        r->bases.assign(bases, bases + r->total);
        *out = r;
    });
    if (rc != PHMM_OK) delete r;
    return rc;
}
uint64_t phmm_reads_count(const phmm_reads *r) { return r ? r->R : 0; }
uint64_t phmm_reads_total_bases(const phmm_reads *r) { return r ? r->total : 0; }
int phmm_reads_last_call_info(const phmm_reads *r, uint16_t *out_dense_columns, uint32_t *out_flags) {
    return guarded([&] {
        if (!r) PHMM_THROW(PHMM_EINVAL, "NULL reads");
        if (r->warm_hint.size() != r->R || r->last_flags.size() != r->R)
            PHMM_THROW(PHMM_EINVAL, "no adaptive-sparse call (use_max_ratio = 1, no mappings) has run on these reads yet");
        if (out_dense_columns) std::memcpy(out_dense_columns, r->warm_hint.data(), sizeof(uint16_t) * r->R);
        if (out_flags) std::memcpy(out_flags, r->last_flags.data(), sizeof(uint32_t) * r->R);
    });
}
void phmm_reads_destroy(phmm_reads *r) { delete r; }

int phmm_run_dense(phmm_model *m, const phmm_reads *reads, double *lf, double *lb, double *nf) {
    return guarded_on(m, [&] {
        if (!m || !reads) PHMM_THROW(PHMM_EINVAL, "NULL model or reads");
        if (reads->R == 0) {
            if (nf) {
                std::vector<double> z(m->N, 0.0);
                DevBuf b;
                b.upload(z.data(), z.size() * sizeof(double));
                copy_out(nf, b.p, z.size() * sizeof(double));
            }
            return;
        }
        run_dense(m, reads, lf, lb, nf);
    });
}

int phmm_run_dense_edges(phmm_model *m, const phmm_reads *reads, double *out_lf, double *out_ef, double *out_if) {
    return guarded_on(m, [&] {
        if (!m || !reads) PHMM_THROW(PHMM_EINVAL, "NULL model or reads");
        if (reads->R == 0) return;
        for (uint64_t r = 0; r < reads->R; r++)
            if (reads->off[r + 1] == reads->off[r]) PHMM_THROW(PHMM_EINVAL, "empty read (reference panics: table.rs:388)");
        run_dense_edges(m, reads, out_lf, out_ef, out_if);
    });
}

int phmm_full_prob_sparse_backward(phmm_model *m, const phmm_reads *reads, double *out_logp, double *out_total) {
    return guarded_on(m, [&] {
        if (!m || !reads) PHMM_THROW(PHMM_EINVAL, "NULL model or reads");
        if (reads->R == 0) {
            const double zero = 0.0;  // empty product = Prob::one()
            put_doubles(out_total, &zero, 1);
            return;
        }
        for (uint64_t r = 0; r < reads->R; r++)
            if (reads->off[r + 1] == reads->off[r]) PHMM_THROW(PHMM_EINVAL, "empty read (reference panics: table.rs:388)");
        full_prob_sparse_backward(m, reads, out_logp, out_total);
    });
}

int phmm_run_sparse(phmm_model *m, const phmm_reads *reads, double *out_lf, double *out_lb, double *out_nf) {
    return guarded_on(m, [&] {
        if (!m || !reads) PHMM_THROW(PHMM_EINVAL, "NULL model or reads");
        if (reads->R == 0) {
            if (out_nf) {
                std::vector<double> z(m->N, 0.0);
                DevBuf b;
                b.upload(z.data(), z.size() * sizeof(double));
                copy_out(out_nf, b.p, z.size() * sizeof(double));
            }
            return;
        }
        for (uint64_t r = 0; r < reads->R; r++)
            if (reads->off[r + 1] == reads->off[r]) PHMM_THROW(PHMM_EINVAL, "empty read (reference panics: table.rs:388)");
        run_sparse(m, reads, out_lf, out_lb, out_nf);
    });
}

int phmm_backward_sparse_tables(phmm_model *m, const uint8_t *read, uint64_t len, double *b_m, double *b_i, double *b_d,
                                double *b_scal, uint8_t *is_dense) {
    return guarded_on(m, [&] {
        if (!m || !read) PHMM_THROW(PHMM_EINVAL, "NULL model or read");
        if (len == 0) PHMM_THROW(PHMM_EINVAL, "empty read");
        backward_sparse_tables(m, read, len, b_m, b_i, b_d, b_scal, is_dense);
    });
}

int phmm_q_score_exact(const phmm_model *m, const double *edge_freq, const double *init_freq, double *out_q) {
    return guarded([&] {
        if (!m || !init_freq || !out_q || (m->E && !edge_freq)) PHMM_THROW(PHMM_EINVAL, "NULL argument");
        double init = 0.0, trans = 0.0;
        for (uint32_t v = 0; v < m->N; v++) {
            if (m->emission[v] == (uint8_t)'n') continue;
            if (!std::isfinite(m->init_logp[v])) PHMM_THROW(PHMM_EINVAL, "init_prob of an emittable node is not finite (q.rs:79)");
            init += init_freq[v] * m->init_logp[v];
            for (uint32_t a = m->chi_off[v]; a < m->chi_off[v + 1]; a++) {
                if (m->emission[m->chi_node[a]] == (uint8_t)'n') continue;
                const uint32_t e = m->chi_edge[a];
                if (!std::isfinite(m->trans_logp[e])) PHMM_THROW(PHMM_EINVAL, "trans_prob between emittable nodes is not finite (q.rs:88)");
                trans += edge_freq[e] * m->trans_logp[e];
            }
        }
        out_q[0] = init;
        out_q[1] = trans;
        out_q[2] = 0.0;
    });
}

int phmm_dense_tables(phmm_model *m, const uint8_t *read, uint64_t len, double *f_m, double *f_i, double *f_d,
                      double *f_scal, double *b_m, double *b_i, double *b_d, double *b_scal) {
    return guarded_on(m, [&] {
        if (!m || !read) PHMM_THROW(PHMM_EINVAL, "NULL model or read");
        if (len == 0) PHMM_THROW(PHMM_EINVAL, "empty read");
        dense_tables(m, read, len, f_m, f_i, f_d, f_scal, b_m, b_i, b_d, b_scal);
    });
}

int phmm_mappings_create(const phmm_reads *reads, const uint64_t *pos_off, const uint32_t *nodes,
                         const double *logp, phmm_mappings **out) {
    phmm_mappings *mp = nullptr;
    int rc = guarded([&] {
        if (!out) PHMM_THROW(PHMM_EINVAL, "out is NULL");
        *out = nullptr;
        if (!reads || !pos_off) PHMM_THROW(PHMM_EINVAL, "NULL argument");
        mp = new phmm_mappings();
        mp->R = reads->R;
        mp->total_pos = reads->total;
        mp->read_off = reads->off;
        mp->pos_off.assign(pos_off, pos_off + reads->total + 1);
        if (pos_off[0] != 0) PHMM_THROW(PHMM_EINVAL, "pos_off[0] must be 0");
        for (uint64_t i = 0; i < reads->total; i++) {
            if (pos_off[i + 1] < pos_off[i]) PHMM_THROW(PHMM_EINVAL, "pos_off not monotone");
            if (pos_off[i + 1] - pos_off[i] > PHMM_MAX_ACTIVE_NODES)
                PHMM_THROW(PHMM_ECAPACITY, "a mapping position lists more than 400 nodes");
        }
        const uint64_t te = pos_off[reads->total];
        if (te && !nodes) PHMM_THROW(PHMM_EINVAL, "NULL nodes");
        mp->nodes.assign(nodes, nodes + te);
        if (logp) mp->logp.assign(logp, logp + te);
        else mp->logp.assign(te, 0.0);
        mp->total_entries = te;
        mp->read_max_list.assign(reads->R, 0);
        for (uint64_t r = 0; r < reads->R; r++)
            for (uint64_t i = reads->off[r]; i < reads->off[r + 1]; i++)
                mp->read_max_list[r] = std::max<uint32_t>(mp->read_max_list[r], (uint32_t)(pos_off[i + 1] - pos_off[i]));
        *out = mp;
    });
    if (rc != PHMM_OK) delete mp;
    return rc;
}
uint64_t phmm_mappings_total_positions(const phmm_mappings *mp) { return mp ? mp->total_pos : 0; }
uint64_t phmm_mappings_total_entries(const phmm_mappings *mp) { return mp ? mp->total_entries : 0; }
int phmm_mappings_export(const phmm_mappings *mp, uint64_t *pos_off, uint32_t *nodes, double *logp) {
    return guarded([&] {
        if (!mp) PHMM_THROW(PHMM_EINVAL, "NULL mappings");
        mappings_materialize_host(mp);
        if (pos_off) std::memcpy(pos_off, mp->pos_off.data(), mp->pos_off.size() * sizeof(uint64_t));
        if (nodes) std::memcpy(nodes, mp->nodes.data(), mp->nodes.size() * sizeof(uint32_t));
        if (logp) std::memcpy(logp, mp->logp.data(), mp->logp.size() * sizeof(double));
    });
}
void phmm_mappings_destroy(phmm_mappings *mp) {
    if (!mp) return;
    // the device CSR goes back to its device's pool (a spare for the next generate_mappings call)
    int dev = -1;
    if (mp->device >= 0 && hipGetDevice(&dev) == hipSuccess && dev == mp->device) {
        try {
            DevicePool &pool = device_pool();
            pool.recycle(mp->d_nodes);
            pool.recycle(mp->d_logp);
            pool.recycle(mp->d_pos_off);
        } catch (...) {
        }
    }
    delete mp;
}

int phmm_mappings_read_logp(const phmm_mappings *mp, double *out_logp, double *out_total) {
    return guarded([&] {
        if (!mp) PHMM_THROW(PHMM_EINVAL, "NULL mappings");
        if (mp->read_logp.size() != mp->R)
            PHMM_THROW(PHMM_EINVAL, "these mappings were not produced by phmm_generate_mappings");
        double tot = 0.0;
        for (double v : mp->read_logp) tot += v;
        put_doubles(out_logp, mp->read_logp.data(), mp->R);
        put_doubles(out_total, &tot, 1);
    });
}

// Mappings::to_node_freqs (hint.rs:161-171): freq[v] = sum of linear probs over all lists
int phmm_mappings_node_freqs(const phmm_mappings *mp, uint32_t n_nodes, double *out) {
    return guarded([&] {
        if (!mp || !out) PHMM_THROW(PHMM_EINVAL, "NULL argument");
        mappings_materialize_host(mp);
        std::vector<double> f(n_nodes, 0.0);
        for (size_t a = 0; a < mp->nodes.size(); a++) {
            if (mp->nodes[a] >= n_nodes) PHMM_THROW(PHMM_EINVAL, "mapping node out of range");
            f[mp->nodes[a]] += std::exp(mp->logp[a]);
        }
        put_doubles(out, f.data(), n_nodes);
    });
}

static void check_mapping_nodes(const phmm_model *m, const phmm_mappings *mp, const phmm_reads *reads) {
    if (mp->R != reads->R || mp->total_pos != reads->total || mp->read_off != reads->off)
        PHMM_THROW(PHMM_EINVAL, "mappings were built for a different read set");
    if (mp->trusted) return;
    for (uint32_t v : mp->nodes)
        if (v >= m->N) PHMM_THROW(PHMM_EINVAL, "mapping node out of range");
}

int phmm_full_prob_reads(phmm_model *m, const phmm_reads *reads, const phmm_mappings *mp, int use_max_ratio,
                         double *out_logp, double *out_total) {
    return guarded_on(m, [&] {
        if (!m || !reads) PHMM_THROW(PHMM_EINVAL, "NULL model or reads");
        if (reads->R == 0) {
            const double zero = 0.0;  // empty product = Prob::one()
            put_doubles(out_total, &zero, 1);
            return;
        }
        if (mp) {
            check_mapping_nodes(m, mp, reads);
            full_prob_reads_hinted(m, reads, mp, 1, nullptr, nullptr, out_logp, out_total);
        } else {
            full_prob_reads_sparse(m, reads, out_logp, out_total, nullptr, use_max_ratio != 0);
        }
    });
}

int phmm_full_prob_reads_candidates(phmm_model *m, const phmm_reads *reads, const phmm_mappings *mp, uint32_t n_cand,
                                    const double *init_logp, const double *trans_logp, double *out_logp,
                                    double *out_total) {
    return guarded_on(m, [&] {
        if (!m || !reads || !mp) PHMM_THROW(PHMM_EINVAL, "NULL model, reads or mappings");
        if (n_cand == 0) return;
        if (!init_logp || (m->E && !trans_logp)) PHMM_THROW(PHMM_EINVAL, "NULL candidate arrays");
        check_mapping_nodes(m, mp, reads);
        full_prob_reads_hinted(m, reads, mp, n_cand, init_logp, trans_logp, out_logp, out_total);
    });
}

int phmm_full_prob_reads_copy_nums(phmm_model *m, const phmm_reads *reads, const phmm_mappings *mp, uint32_t n_cand,
                                   const uint32_t *copy_nums, uint32_t min_copy_num, double *out_logp,
                                   double *out_total) {
    return guarded_on(m, [&] {
        if (!m || !reads || !mp) PHMM_THROW(PHMM_EINVAL, "NULL model, reads or mappings");
        if (n_cand == 0) return;
        if (!copy_nums) PHMM_THROW(PHMM_EINVAL, "NULL copy numbers");
        check_mapping_nodes(m, mp, reads);
        full_prob_reads_hinted(m, reads, mp, n_cand, nullptr, nullptr, out_logp, out_total, nullptr, copy_nums,
                               min_copy_num);
    });
}

int phmm_mappings_map_nodes(phmm_model *model_after, const phmm_reads *reads, const phmm_mappings *mp,
                            const uint32_t *map_off, const uint32_t *map_nodes, uint32_t n_nodes_before,
                            phmm_mappings **out) {
    return guarded_on(model_after, [&] {
        if (!model_after || !reads || !mp || !map_off || !out) PHMM_THROW(PHMM_EINVAL, "NULL argument");
        *out = nullptr;
        if (mp->R != reads->R || mp->total_pos != reads->total) PHMM_THROW(PHMM_EINVAL, "mappings do not belong to these reads");
        if (map_off[0] != 0) PHMM_THROW(PHMM_EINVAL, "map_off[0] must be 0");
        for (uint32_t v = 0; v < n_nodes_before; v++)
            if (map_off[v + 1] < map_off[v]) PHMM_THROW(PHMM_EINVAL, "map_off not monotone");
        if (map_off[n_nodes_before] && !map_nodes) PHMM_THROW(PHMM_EINVAL, "NULL map_nodes");
        mappings_map_nodes(model_after, reads, mp, map_off, map_nodes, n_nodes_before, out);
    });
}

int phmm_generate_mappings(phmm_model *m, const phmm_reads *reads, const phmm_mappings *mp, int use_max_ratio,
                           phmm_mappings **out, double *out_node_freq) {
    return guarded_on(m, [&] {
        if (!m || !reads || !out) PHMM_THROW(PHMM_EINVAL, "NULL argument");
        *out = nullptr;
        if (reads->R == 0) PHMM_THROW(PHMM_EINVAL, "no reads");
        if (mp) {
            check_mapping_nodes(m, mp, reads);
            generate_mappings_hinted(m, reads, mp, use_max_ratio, out, out_node_freq);
            return;
        }
        generate_mappings_sparse(m, reads, out, out_node_freq, use_max_ratio != 0);
    });
}

}  // extern "C"
