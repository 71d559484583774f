// Internal declarations shared by the host side and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/phmm_amd.h"

namespace phmm {

// ---------------------------------------------------------------- errors
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

struct Error {
    int code;
    std::string msg;
};
#define PHMM_THROW(code, msg) throw ::phmm::Error{(code), (msg)}
#define HIP_CHECK(expr)                                                                      \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            PHMM_THROW(_e == hipErrorOutOfMemory ? PHMM_ENOMEM : PHMM_ENODEVICE,             \
                       std::string(#expr) + ": " + hipGetErrorString(_e));                   \
    } while (0)

hipStream_t current_stream();
// workspace set of the calling thread (0 unless the thread is a pipeline worker)
int &workset_index();
// a pipeline worker adopts the API thread's per-call settings with its own stream and workspace set
struct ThreadContext {
    uint64_t ws_limit;
    bool timing;
    int device;
};
ThreadContext capture_thread_context();
void adopt_thread_context(const ThreadContext &c, hipStream_t stream, int workset);
uint64_t workspace_limit();
struct DevicePool;
// DP-table budget of a call: PHMM_MEM_FRACTION (0.9) of what is free plus the pool's own tables (they are
// re-allocated), less `reserve` bytes the call's other buffers still have to grow by
uint64_t table_budget(const DevicePool &pool, uint64_t reserve = 0);
// budget of a call that plans every buffer it uses (sparse_dyn.hip): a fraction of free + everything the pool holds
uint64_t planned_budget(const DevicePool &pool);

// ---------------------------------------------------------------- device buffers
void flush_spare_buffers();  // (api.cpp) frees the spare buffers of the current device's pool
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    // grow-only
    void reserve(size_t n) {
        if (n <= bytes) return;
        release();
        if (hipMalloc(&p, n ? n : 1) != hipSuccess) {
            // the spare result buffers (DevicePool::recycle) are the one thing this library holds that nobody needs
            (void)hipGetLastError();
            p = nullptr;
            flush_spare_buffers();
            HIP_CHECK(hipMalloc(&p, n ? n : 1));
        }
        bytes = n;
    }
    void adopt(void *q, size_t n) {
        release();
        p = q;
        bytes = n;
    }
    void *detach() {
        void *q = p;
        p = nullptr;
        bytes = 0;
        return q;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
    void upload(const void *src, size_t n) {
        reserve(n);
        if (n) HIP_CHECK(hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, current_stream()));
    }
};
// copy device -> (host or device) destination
void copy_out(void *dst, const void *src_dev, size_t bytes);

// ---------------------------------------------------------------- per-device workspace pool
// One set of grow-only buffers per pipeline worker (sparse_dyn.hip runs chunks of read groups concurrently on
// their own streams); the API thread and single-stream paths use set 0.  aux[] = per-call scratch kept across
// calls (no hipMalloc in the steady state).  phmm_release_workspace() gives everything back.
struct WorkSet {
    DevBuf tables, misc, aux[24];
    size_t bytes() const {
        size_t b = tables.bytes + misc.bytes;
        for (const auto &a : aux) b += a.bytes;
        return b;
    }
    void release() {
        tables.release();
        misc.release();
        for (auto &a : aux) a.release();
    }
};
static constexpr int MAX_WORKERS = 4;
struct DevicePool {
    int device = 0;
    std::recursive_mutex call_mu;  // compute calls on one device run one at a time (any handle, any thread)
    WorkSet wsets[MAX_WORKERS];
    hipStream_t wstream[MAX_WORKERS] = {};  // worker streams, created on first use
    hipStream_t cstream[MAX_WORKERS] = {};  // per worker: side stream of the mapping-list kernels (mapping_flow.hip)
    hipEvent_t cevent[MAX_WORKERS][4] = {};
    DevBuf ws_out;
    // The device buffers of destroyed phmm_mappings (their CSR: gigabytes on a tandem repeat), kept for the next
    // generate_mappings call: a hipFree + hipMalloc pair per buffer and call otherwise (0.1-2 s per call on `rep20`,
    // and it varies from box to box).  At most SPARE_MAX buffers / SPARE_BYTES; given back by release(), and by any
    // allocation that fails (DevBuf::reserve).
    static constexpr size_t SPARE_MAX = 6, SPARE_BYTES = (size_t)8 << 30;
    std::mutex spare_mu;
    std::vector<std::pair<void *, size_t>> spare;
    size_t spare_total = 0;
    void recycle(DevBuf &b);         // takes b's memory (frees it when the cache is full)
    void take(DevBuf &b, size_t n);  // b.reserve(n), out of the cache when a spare of n .. 2n bytes is there
    void flush_spares();
    size_t owned_table_bytes() const {
        size_t b = 0;
        for (const auto &w : wsets) b += w.tables.bytes;
        return b;
    }
    size_t owned_bytes() const {
        size_t b = ws_out.bytes + spare_total;
        for (const auto &w : wsets) b += w.bytes();
        return b;
    }
    void release();  // frees every buffer, stream and event (the device must be idle)
};
// the pool of the calling thread's current device (created on first use, never destroyed: HIP may be gone at exit)
DevicePool &device_pool();

// ---------------------------------------------------------------- linear-domain parameters
// The kernels run in the (scaled) linear probability domain; see DESIGN.md.
struct LinParams {
    double p_mismatch, p_match, p_random, p_end;
    double p_MM, p_IM, p_DM, p_MI, p_II, p_DI, p_MD, p_ID, p_DD;
    int n_max_gaps;
};

// ancestor closure entry (forward): node a reaches k in `hop` edges with path weight w.
//   w1 = w if hop==1 (direct parent: p_MM/p_IM terms)
//   wD = pDD^(hop-1) w  for hop <= G+1   -> d[k]      = sum wD * g[a] + c*dinit[k]
//   wT = pDD^(hop-2) w  for 2<=hop<=G+2  -> (T d)[k]  = sum wT * g[a] + c*tdinit[k]
// with g[a] = p_MD m[a] + p_ID i[a]   (forward.rs:423-524 unrolled; DESIGN.md "Del closure")
struct FwdEntry {
    uint32_t node;
    uint32_t pad;
    double w1, wD, wT;
};
// descendant closure entry (backward): v reaches u in `hop` edges with path weight w.
//   c1  = w if hop==1
//   cAd = pDD^(hop-1) w for hop <= G+1 ; cAt = pDD^(hop-2) w for 2<=hop<=G+2
//   cQd = pDD^hop w     for hop <= G      (cQt == cAd)
// (backward.rs:299-483 unrolled)
struct BwdEntry {
    uint32_t node;
    uint32_t emis;
    double c1, cAd, cAt, cQd;
};

// Closure entry by hop (n_max_gaps <= 4): node `node` reaches (is reached from) k in exactly `hop`
// edges with total path weight w.  The kernels keep, per thread, the per-hop sums
//   A[h] = sum over entries with hop h+1 of w * value(node)        h = 0 .. CHAIN_HOPS-1
// ("window") from which every Del-closure term follows with fixed coefficients (DenseArgs::cD ..).
struct HopEntry {
    uint32_t node;
    uint32_t hop_emis;  // hop | emission(node) << 8
    double w;
};

// Per-node record of the dense kernels (one 32-byte scalar load per node).
//   CHAIN_F: k's only parent is k-1 and that edge has weight 1 -> the window of k is the window of k-1
//            shifted by one hop with k-1's own values in front (no gather at all)
//   CHAIN_B: k's only child is k+1, weight 1 (the same for the descendant window)
// Node ids follow the k-mer order of the haplotypes, so this holds along unitigs AND behind a merge;
// the window is rebuilt from the hop entries at the first node of a thread's run, behind a branch and
// at a merge node only.  (n_max_gaps > 4: no flags, merged closure entries, no window.)
struct NodeRec {
    double init, dinit, tdinit;
    uint32_t emis;
    uint32_t flags;
};
static constexpr uint32_t CHAIN_F = 1u, CHAIN_B = 2u;
static constexpr int CHAIN_HOPS = 6;

// Everything the sparse (frontier) kernels need to know about one node, in ONE aligned record: a
// wave that meets a new node pays one memory round trip for it instead of the CSR's three dependent
// ones (offsets -> neighbour ids / weights -> their emissions).  Neighbours keep the CSR (petgraph)
// order.  MultiDbg bounds the degree by 5 (multi_dbg.rs:82); `over` marks nodes beyond that, which
// only the dense kernels support.
static constexpr int ADJ_DEG = 5;
struct alignas(16) FwdAdj {   // forward: expansion to children, sums over parents
    double par_w[ADJ_DEG];    // linear transition probability parent -> this node (the model's own)
    double init;
    uint32_t par[ADJ_DEG];
    uint32_t chi[ADJ_DEG];
    uint8_t emis, npar, nchi, over;
    uint32_t pad;
};
struct alignas(16) BwdAdj {   // backward: sums over children
    double chi_w[ADJ_DEG];
    uint32_t chi[ADJ_DEG];
    uint8_t chi_emis[ADJ_DEG];
    uint8_t emis, nchi, over;
    uint32_t par0;            // first parent (0xffffffff: none): the node a backward column takes in next on a unitig
};
// Topology-only record of the hinted forward (candidate batches bring their own init / trans): parents with the
// ids of their edges, so that trans[candidate][edge] is one dependent load behind the record.
struct alignas(16) ParRec {
    uint32_t par[ADJ_DEG];
    uint32_t pedge[ADJ_DEG];
    uint8_t npar, emis, over, pad;
    uint32_t pad2;
};
static_assert(sizeof(FwdAdj) == 96 && sizeof(BwdAdj) == 80 && sizeof(ParRec) == 48, "adjacency record layout");

struct ModelDev {
    uint32_t N = 0, E = 0;
    DevBuf nodes;            // NodeRec[N]
    DevBuf emis;             // u8[N]
    DevBuf init;             // f64[N] linear
    DevBuf dinit, tdinit;    // f64[N]
    DevBuf fc_off, fc_ent;   // u32[N+1], FwdEntry[]
    DevBuf bc_off, bc_ent;   // u32[N+1], BwdEntry[]
    DevBuf fh_off, fh_ent;   // u32[N+1], HopEntry[]  (ancestors by hop)
    DevBuf bh_off, bh_ent;   // u32[N+1], HopEntry[]  (descendants by hop)
    // parent / child CSR in linear domain for the sparse kernels
    DevBuf par_off, par_node, par_w;  // u32[N+1], u32[E], f64[E]
    DevBuf chi_off, chi_node, chi_w;
    DevBuf par_edge, chi_edge;        // u32[E] edge ids (candidate batches index trans by edge)
    DevBuf trans_lin;                 // f64[E] by edge id
    DevBuf fadj, badj;                // FwdAdj[N], BwdAdj[N]
    DevBuf prec;                      // ParRec[N]
    DevBuf logib;                     // f64[logib_len] forward InsBegin chain (log)
    size_t logib_len = 0;
    uint32_t max_degree = 0;
};

}  // namespace phmm

struct phmm_model {
    uint32_t N = 0, E = 0;
    phmm_params params{};
    phmm::LinParams lin{};
    std::vector<uint8_t> emission;
    std::vector<double> init_logp, trans_logp;
    std::vector<uint32_t> esrc, edst;
    // host CSR (petgraph order: newest edge first)
    std::vector<uint32_t> par_off, par_node, par_edge, chi_off, chi_node, chi_edge;
    std::vector<double> logib;  // forward InsBegin chain, log domain (forward.rs:541-545)
    phmm::ModelDev dev;
    double wf_ub_a = 0.0, wf_ub_b = 0.0;  // column total <= ub_a * max(m,i) + ub_b * p_ID * ib (model.cpp)
    // Workspaces (DP tables, record pools, per-call scratch) belong to the DEVICE, not to the model: every
    // handle on a device shares one phmm::DevicePool, so a mapping model and a scoring model coexist
    // (multi_dbg/posterior.rs:247-255, 609-630) without each sizing its own tables from what is free.
    phmm::DevicePool *pool = nullptr;
    phmm::WorkSet &wset();
};

struct phmm_reads {
    uint64_t R = 0, total = 0;
    std::vector<uint8_t> bases;
    std::vector<uint64_t> off;
    uint64_t max_len = 0;
    // lazily uploaded device copies (per device of first use)
    mutable phmm::DevBuf d_bases, d_off;
    mutable bool on_device = false;
    // dense warm-up columns each read needed in the last adaptive-sparse call on this handle (empty: none
    // yet).  Only used to GROUP reads with similar warm-up lengths (results do not depend on the grouping):
    // a read group runs dense columns until its slowest read switches, and rows of 64 reads in which only a
    // few are still dense cost whole 64-byte sectors per live read.
    mutable std::vector<uint16_t> warm_hint;
    // what the last adaptive-sparse call did with each read (phmm_reads_last_call_info; diagnostics / parity tests):
    // PHMM_READ_* bits of include/phmm_amd.h
    mutable std::vector<uint32_t> last_flags;
};

struct phmm_mappings {
    uint64_t R = 0, total_pos = 0;
    std::vector<uint64_t> read_off;  // [R+1] position offsets per read
    std::vector<uint64_t> pos_off;   // [total_pos+1]
    std::vector<uint32_t> nodes;
    std::vector<double> logp;
    std::vector<uint32_t> read_max_list;  // [R] longest node list of each read
    std::vector<double> read_logp;        // [R] ln P(read) of the forward pass that produced the mappings (may be empty)
    mutable phmm::DevBuf d_pos_off, d_nodes, d_logp;
    int device = -1;  // >= 0: the device whose pool takes the buffers back (phmm_mappings_destroy)
    mutable bool on_device = false;
    // Mappings produced by phmm_generate_mappings stay on the device (their next consumer is the
    // hinted forward kernel); the host vectors above are filled on first host access.
    mutable bool host_valid = true;
    uint64_t total_entries = 0;
    bool trusted = false;  // node ids come from our own kernels: no range check needed
};
namespace phmm {
void mappings_materialize_host(const phmm_mappings *mp);
}

namespace phmm {

struct CallStats {
    double ms[4] = {0, 0, 0, 0};
    uint64_t launches[4] = {0, 0, 0, 0};
    uint64_t cells[4] = {0, 0, 0, 0};
};
CallStats &stats();
bool timing_enabled();

// Per-launch HIP-event timing of one kernel class on the call's stream (bench.py's roofline):
// an event pair brackets every launch, the elapsed times are summed after one final sync.
struct LaunchTimer {
    bool on;
    std::vector<hipEvent_t> ev;
    explicit LaunchTimer(bool on_) : on(on_) {}
    ~LaunchTimer() {
        for (auto e : ev) (void)hipEventDestroy(e);
    }
    void begin() {
        if (!on) return;
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        HIP_CHECK(hipEventRecord(e, current_stream()));
        ev.push_back(e);
    }
    void end() { begin(); }
    double total_ms() {
        if (!on || ev.empty()) return 0.0;
        HIP_CHECK(hipEventSynchronize(ev.back()));
        double t = 0.0;
        for (size_t k = 0; k + 1 < ev.size(); k += 2) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
            t += ms;
        }
        return t;
    }
};

// PHMM_TRACE=1: print host-side phase times (syncs the stream; diagnostics only)
void trace(const char *tag);

// Developer knobs (environment variables; none is needed for normal use).  They are read in ONE place, when a compute
// entry point takes the device lock (api.cpp: guarded_on), never on the path of a call.
struct Knobs {
    bool trace = false;              // PHMM_TRACE: phase timings on stderr
    bool no_lean = false;            // PHMM_NO_LEAN: generic vector kernels instead of the one-lane-per-node ones
    bool no_wide_class = false;      // PHMM_NO_WIDE_CLASS: generic 400-slot kernels instead of the one-thread-per-node ones
    bool no_packed = false;          // PHMM_NO_PACKED: one candidate per wave
    int packed_cpl = 0;              // PHMM_PACKED_CPL: candidates per lane (0: automatic)
    bool no_exact_hinted = false;    // PHMM_NO_EXACT_HINTED: no wide-range pass over reads that a candidate cuts
    bool no_side_worker = false;     // PHMM_NO_SIDE_WORKER
    bool no_wide_handover = false;   // PHMM_NO_WIDE_HANDOVER
    int workers = 1;                 // PHMM_WORKERS: chunk pipeline
    int warm_cols = 0;               // PHMM_WARM_COLS: dense columns kept by the main plan (0: from N)
    int chunk_groups = 0;            // PHMM_CHUNK_GROUPS
    bool no_keep_all = false;        // PHMM_NO_KEEP_ALL: a small read set defers its slow reads too (see sparse_dyn.hip)
    int pipeline_min_groups = 8;     // PHMM_PIPELINE_MIN_GROUPS
    bool no_runmax = false;          // PHMM_NO_RUNMAX
    bool force_radix = false;        // PHMM_FORCE_RADIX
    bool serial_emit = false;        // PHMM_SERIAL_EMIT
    bool emit_high_priority = false; // PHMM_EMIT_HIGH_PRIORITY: list kernels of the dense head at the highest stream priority
    bool no_dma = false;             // PHMM_NO_DMA: forward rows through registers
    bool bwd_dma = false;            // PHMM_BWD_DMA
    int dense_streams = 0;           // PHMM_DENSE_STREAMS
    int dense_w = 0;                 // PHMM_DENSE_W
    int dense_npt = 0;               // PHMM_DENSE_NPT
    double mem_fraction = 0.0;       // PHMM_MEM_FRACTION (0: the defaults, 0.9 tables / 0.95 planned)
};
const Knobs &knobs();
void refresh_knobs();

void model_build_host(phmm_model *m);    // CSR + logib
void model_upload(phmm_model *m);        // closures + device arrays

// dense driver (dense.hip)
void run_dense(phmm_model *m, const phmm_reads *reads, double *out_lf, double *out_lb,
               double *out_nf);
// backward_sparse (backward.rs:146-185): sparse_bwd.hip
void full_prob_sparse_backward(phmm_model *m, const phmm_reads *reads, double *out_logp, double *out_total);
void backward_sparse_tables(phmm_model *m, const uint8_t *read, uint64_t len, double *b_m, double *b_i, double *b_d,
                            double *b_scal, uint8_t *is_dense);
void run_sparse(phmm_model *m, const phmm_reads *reads, double *out_lf, double *out_lb, double *out_nf);
void run_dense_edges(phmm_model *m, const phmm_reads *reads, double *out_lf, double *out_ef, double *out_if);
struct RecPool;
// sparse / hinted drivers (sparse.hip)
void full_prob_reads_hinted(phmm_model *m, const phmm_reads *reads, const phmm_mappings *mp, uint32_t n_cand,
                            const double *init_logp, const double *trans_logp, double *out_logp,
                            double *out_total, const RecPool *pool = nullptr, const uint32_t *copy_nums = nullptr,
                            uint32_t min_copy_num = 0);
void upload_reads(const phmm_reads *r);
void upload_mappings(const phmm_mappings *mp);
void mappings_map_nodes(phmm_model *m, const phmm_reads *reads, const phmm_mappings *mp_in, const uint32_t *map_off,
                        const uint32_t *map_nodes, uint32_t n_old, phmm_mappings **out);
void generate_mappings_hinted(phmm_model *m, const phmm_reads *reads, const phmm_mappings *mp_in, int use_max_ratio,
                              phmm_mappings **out, double *out_node_freq);
struct MappingSink;
// by_ratio: use_max_ratio of forward_sparse (forward.rs:93-154); false = fixed warm-up + top n_active_nodes
void full_prob_reads_sparse(phmm_model *m, const phmm_reads *reads, double *out_logp, double *out_total,
                            MappingSink *sink, bool by_ratio = true);
void generate_mappings_sparse(phmm_model *m, const phmm_reads *reads, phmm_mappings **out, double *out_node_freq,
                              bool by_ratio = true);
void ensure_logib(phmm_model *m, size_t len);
void put_doubles(double *dst, const double *src_host, size_t n);
void dense_tables(phmm_model *m, const uint8_t *read, uint64_t len, double *f_m, double *f_i,
                  double *f_d, double *f_scal, double *b_m, double *b_i, double *b_d,
                  double *b_scal);

}  // namespace phmm
