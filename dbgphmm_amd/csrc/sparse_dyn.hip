// Adaptive sparse forward = PHMMModel::forward_sparse / forward_sparse_score_only with
// use_max_ratio = true (src/hmmv2/forward.rs:93-206):
//   column 0 dense; column i < n_warmup dense while the previous (dense) column has more
//   than warmup_threshold nodes within `active_node_max_ratio` of its best node
//   (table.rs:134-149); afterwards sparse for good: active = to_childs_and_us(top nodes),
//   f_step(.., is_dense = false, is_adaptive = true).
//
// GPU formulation
//   * the dense warm-up columns of ALL reads of a read group run through the batched dense
//     kernel (dense.hip) -- that is where the HBM traffic of this mode is;
//   * col_count counts, per read, the nodes of a dense column within the ratio of the
//     column's best node and collects them (<= 400) as the first sparse step's top list;
//   * each read then continues on its own wave64 with the frontier in LDS (frontier_dev.h).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "dense_internal.h"
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "frontier_dev.h"
#include "sparse_dyn.h"
#include "sparse_fwd_kernel.h"
#include "wide_fwd_kernel.h"
#include "lean_fwd_kernel.h"

namespace phmm {

struct WarmArgs {
    DenseArgs d;
    int *sw;                   // [ng][W] switch position s_r (-1 = undecided)
    int *cnt;                  // [ng][W] nodes within the ratio in the column being counted
    uint32_t *cand_node;       // [ng][W][400]
    double *cand_tot;          // [ng][W][400]
    int *cand_n;               // [ng][W] candidates kept for the switch column
    int *undecided;            // [1]
    int *amb;                  // [ng][W] 1: the fused count of fwd_step could not settle this lane's column
    int *n_amb;                // [1]
    double ratio_lin;          // exp(-active_node_max_ratio)
    int n_warmup, threshold;
};

// reads (resident, concatenated) -> bases transposed to [group][pos][W]; 0xff past the end / padding lanes
struct StageArgs {
    const uint8_t *bases;
    const uint64_t *off;
    const uint32_t *order;  // slot -> read id for the chunk's first slot onwards
    uint32_t n_slots;
    int W, Lc, Lfull;
    uint8_t *out_dense, *out_full;
};
__global__ void __launch_bounds__(BLOCK) stage_bases(const StageArgs sa) {
    const int g = blockIdx.y;
    const unsigned t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= (unsigned)sa.Lfull * sa.W) return;
    const int i = (int)(t / sa.W), r = (int)(t % sa.W);
    const uint32_t slot = (uint32_t)g * sa.W + r;
    uint8_t b = 0xff;
    if (slot < sa.n_slots) {
        const uint32_t rd = sa.order[slot];
        const uint64_t o0 = sa.off[rd], o1 = sa.off[rd + 1];
        if ((uint64_t)i < o1 - o0) b = sa.bases[o0 + i];
    }
    sa.out_full[((size_t)g * sa.Lfull + i) * sa.W + r] = b;
    if (i < sa.Lc) sa.out_dense[((size_t)g * sa.Lc + i) * sa.W + r] = b;
}

// count (and collect) the nodes of dense column `col` with total > tmax * exp(-ratio)
template <int W>
__global__ void __launch_bounds__(BLOCK) col_count(const WarmArgs wa, const int col) {
    const DenseArgs &a = wa.d;
    const int g = blockIdx.y;
    const int lb = blockIdx.x;
    constexpr int ROWS = BLOCK / W;
    const int r = threadIdx.x % W, row = threadIdx.x / W;
    const int gi = g * W + r;
    const int len = a.len[gi];
    const bool live = wa.sw[gi] < 0 && col < len && wa.amb[gi] != 0;
    if (lb >= a.nblk) return;
    if (!__syncthreads_or(live)) return;
    const size_t NW = (size_t)a.N * W;
    const double *fm = a.Fm + ((size_t)g * a.Lc + col) * NW;
    const double *fi = a.Fi + ((size_t)g * a.Lc + col) * NW;
    const double *fd = a.Fd + ((size_t)g * a.Lc + col) * NW;
    double thr = 0.0;
    if (live) {
        const unsigned long long tb = a.tmaxF[((size_t)g * a.Lc + col) * W + r];
        thr = __longlong_as_double((long long)tb) * wa.ratio_lin;
    }
    const int kbase = lb * (a.npt * ROWS) + row;
    int local = 0;
    if (live)
        for (int j = 0; j < a.npt; j++) {
            const int k = kbase + j * ROWS;
            if (k >= a.N) break;
            const size_t ix = (size_t)k * W + r;
            const double t = fm[ix] + fi[ix] + fd[ix];
            local += (t > 0.0 && t > thr) ? 1 : 0;
        }
    if (local > 0) {
        int slot = atomicAdd(&wa.cnt[gi], local);
        if (slot < PHMM_MAX_ACTIVE_NODES)
            for (int j = 0; j < a.npt; j++) {
                const int k = kbase + j * ROWS;
                if (k >= a.N) break;
                const size_t ix = (size_t)k * W + r;
                const double t = fm[ix] + fi[ix] + fd[ix];
                if (t > 0.0 && t > thr) {
                    if (slot < PHMM_MAX_ACTIVE_NODES) {
                        wa.cand_node[(size_t)gi * PHMM_MAX_ACTIVE_NODES + slot] = (uint32_t)k;
                        wa.cand_tot[(size_t)gi * PHMM_MAX_ACTIVE_NODES + slot] = t;
                    }
                    slot++;
                }
            }
    }
}

// decision for column `pos` from the count of column pos-1 (forward.rs:107-137):
//   sparse at pos  iff  pos >= n_warmup  or  count(pos-1) <= warmup_threshold
// a read that ends at pos while still dense is finished (sw = len).
__global__ void __launch_bounds__(BLOCK) warm_decide(const WarmArgs wa, const int pos, const int total_lanes) {
    const int gi = blockIdx.x * BLOCK + threadIdx.x;
    if (gi >= total_lanes) return;
    const int len = wa.d.len[gi];
    if (len == 0) {
        wa.sw[gi] = 0;
        return;
    }
    if (wa.sw[gi] >= 0 || !wa.amb[gi]) return;
    bool decided = false;
    if (pos >= len) {
        wa.sw[gi] = len;  // all columns were dense
        decided = true;
    } else if (pos >= 1) {
        const int c = wa.cnt[gi];
        const int ntop = c < PHMM_MAX_ACTIVE_NODES ? c : PHMM_MAX_ACTIVE_NODES;  // ArrayVec capacity
        if (pos >= wa.n_warmup || ntop <= wa.threshold) {
            wa.sw[gi] = pos;
            wa.cand_n[gi] = c;
            decided = true;
        }
    }
    if (!decided) atomicAdd(wa.undecided, 1);
    wa.cnt[gi] = 0;
}

// The same decision from the counts that fwd_step gathered while it wrote column pos-1's Del values
// (dense.hip): one wave per lane.  fwd_step could only bracket the column maximum T of the totals,
// L <= T <= U, so it counted `sub` = #{t > U*ratio} (certainly inside the ratio) and, for lanes in
// collect mode, stored every (node, t) with t > L*ratio (a superset).  Now T is known:
//   * superset complete (<= WF_CAP entries): filter it with T*ratio -> the exact list and count;
//   * else sub > threshold: the column stays dense whatever the exact count is;
//   * else the lane is flagged ambiguous and the exact col_count + warm_decide pair settles it.
__global__ void __launch_bounds__(BLOCK) warm_decide_fused(const WarmArgs wa, const int pos, const int total_lanes, const int W) {
    const int gi = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (gi >= total_lanes) return;
    const int len = wa.d.len[gi];
    if (len == 0) {
        if (lane == 0) wa.sw[gi] = 0;
        return;
    }
    if (wa.sw[gi] >= 0) return;
    const int nsub = wa.d.wf_sub[gi], cn = wa.d.wf_cnt[gi];
    const bool collected = wa.d.wf_mode[gi] != 0 && cn <= WF_CAP;
    bool decided = false, amb = false;
    int new_sw = -1, new_cn = 0;
    if (pos >= len) {
        new_sw = len;  // all columns were dense
        decided = true;
    } else if (pos >= 1) {
        const bool forced = pos >= wa.n_warmup;
        if (collected) {
            const int g = gi / W, r = gi % W;
            const double tmax = __longlong_as_double((long long)wa.d.tmaxF[((size_t)g * wa.d.Lc + (pos - 1)) * W + r]);
            const double thr = tmax * wa.ratio_lin;
            int c = 0;
            for (int base = 0; base < cn; base += 64) {
                const int j = base + lane;
                const double t = j < cn ? wa.d.wf_tot[(size_t)gi * WF_CAP + j] : 0.0;
                const bool ok = j < cn && t > 0.0 && t > thr;
                const unsigned long long mask = __ballot(ok);
                const int slot = c + __popcll(mask & ((1ull << lane) - 1ull));
                if (ok && slot < PHMM_MAX_ACTIVE_NODES) {
                    wa.cand_node[(size_t)gi * PHMM_MAX_ACTIVE_NODES + slot] = wa.d.wf_node[(size_t)gi * WF_CAP + j];
                    wa.cand_tot[(size_t)gi * PHMM_MAX_ACTIVE_NODES + slot] = t;
                }
                c += __popcll(mask);
            }
            const int ntop = c < PHMM_MAX_ACTIVE_NODES ? c : PHMM_MAX_ACTIVE_NODES;  // ArrayVec capacity
            if (forced || ntop <= wa.threshold) {
                new_sw = pos;
                new_cn = c;
                decided = true;
            }
        } else if (!forced && (nsub < PHMM_MAX_ACTIVE_NODES ? nsub : PHMM_MAX_ACTIVE_NODES) > wa.threshold) {
            // certainly more than `threshold` nodes inside the ratio: dense
        } else {
            amb = true;
        }
    }
    if (lane == 0) {
        if (decided) {
            wa.sw[gi] = new_sw;
            if (new_sw < len) wa.cand_n[gi] = new_cn;
        }
        wa.amb[gi] = amb ? 1 : 0;
        if (amb) atomicAdd(wa.n_amb, 1);
        else if (!decided) atomicAdd(wa.undecided, 1);
        // collect the next column's candidates once the count has come down far enough
        ((uint8_t *)wa.d.wf_mode)[gi] = (!decided && pos >= 1 && nsub <= 4096) ? 1 : 0;
        wa.d.wf_sub[gi] = 0;
        wa.d.wf_cnt[gi] = 0;
    }
}

// Forced switch (pos >= n_warmup) with more than 400 nodes inside the ratio: the reference's
// `to_sorted_arrayvec()` keeps the 400 best of the dense nodevec (table.rs:139; ties resolved
// by node index here).  One block per such read: compact the qualifying (node, total) pairs,
// bisect the 400th-largest total on its bit pattern, then bisect the node id among ties.
__device__ __forceinline__ int block_count(int local, int *lds_i) {
    __syncthreads();
    if (threadIdx.x == 0) *lds_i = 0;
    __syncthreads();
    local = wave_isum(local);
    if ((threadIdx.x & 63) == 0) atomicAdd(lds_i, local);
    __syncthreads();
    return *lds_i;
}

__global__ void __launch_bounds__(BLOCK) select_top400(const Top400Args a) {
    __shared__ int cnt;
    const uint32_t gi = a.need[blockIdx.x];
    const int g = (int)(gi / a.W), r = (int)(gi % a.W);
    const int col = a.sw[gi] - 1;
    const int N = a.d.N;
    const size_t NW = (size_t)N * a.W;
    const double *fm = a.d.Fm + ((size_t)g * a.d.Lc + col) * NW;
    const double *fi = a.d.Fi + ((size_t)g * a.d.Lc + col) * NW;
    const double *fd = a.d.Fd + ((size_t)g * a.d.Lc + col) * NW;
    const double tmax = __longlong_as_double((long long)a.d.tmaxF[((size_t)g * a.d.Lc + col) * a.W + r]);
    const double thr = tmax * a.ratio_lin;
    uint32_t *sn = a.sc_node + (size_t)blockIdx.x * N;
    double *stt = a.sc_tot + (size_t)blockIdx.x * N;
    if (threadIdx.x == 0) a.sc_n[blockIdx.x] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < N; k += BLOCK) {
        const size_t ix = (size_t)k * a.W + r;
        const double t = fm[ix] + fi[ix] + fd[ix];
        if (t > 0.0 && t > thr) {
            const int s = atomicAdd(&a.sc_n[blockIdx.x], 1);
            sn[s] = (uint32_t)k;
            stt[s] = t;
        }
    }
    __threadfence_block();
    __syncthreads();
    const int n = a.sc_n[blockIdx.x];
    const int K = a.K;
    if (n <= K) {
        // fewer qualify than are asked for: all of them
        for (int j = threadIdx.x; j < n; j += BLOCK) {
            a.cand_node[(size_t)gi * PHMM_MAX_ACTIVE_NODES + j] = sn[j];
            a.cand_tot[(size_t)gi * PHMM_MAX_ACTIVE_NODES + j] = stt[j];
        }
        if (threadIdx.x == 0) a.cand_n[gi] = n;
        return;
    }
    // largest T with count(v >= T) >= K   (bit patterns of positive doubles are ordered)
    unsigned long long lo = 0ull, hi = (unsigned long long)__double_as_longlong(tmax);
    while (lo < hi) {
        const unsigned long long mid = lo + (hi - lo + 1ull) / 2ull;
        int c = 0;
        for (int j = threadIdx.x; j < n; j += BLOCK) c += (unsigned long long)__double_as_longlong(stt[j]) >= mid;
        c = block_count(c, &cnt);
        if (c >= K) lo = mid;
        else hi = mid - 1ull;
    }
    const unsigned long long T = lo;
    int above = 0;
    for (int j = threadIdx.x; j < n; j += BLOCK) above += (unsigned long long)__double_as_longlong(stt[j]) > T;
    above = block_count(above, &cnt);
    const int need_ties = K - above;
    // smallest node id bound B with count(v == T && node <= B) >= need_ties
    uint32_t blo = 0u, bhi = (uint32_t)N - 1u;
    while (blo < bhi) {
        const uint32_t mid = blo + (bhi - blo) / 2u;
        int c = 0;
        for (int j = threadIdx.x; j < n; j += BLOCK)
            c += ((unsigned long long)__double_as_longlong(stt[j]) == T && sn[j] <= mid) ? 1 : 0;
        c = block_count(c, &cnt);
        if (c >= need_ties) bhi = mid;
        else blo = mid + 1u;
    }
    __syncthreads();
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += BLOCK) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(stt[j]);
        if (b > T || (b == T && sn[j] <= blo)) {
            const int s = atomicAdd(&cnt, 1);
            if (s < K) {
                a.cand_node[(size_t)gi * PHMM_MAX_ACTIVE_NODES + s] = sn[j];
                a.cand_tot[(size_t)gi * PHMM_MAX_ACTIVE_NODES + s] = stt[j];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) a.cand_n[gi] = cnt < K ? cnt : K;
}

void launch_select_top(const Top400Args &ta, unsigned n, hipStream_t s) {
    hipLaunchKernelGGL(select_top400, dim3(n), dim3(BLOCK), 0, s, ta);
    HIP_CHECK(hipGetLastError());
}

namespace {

template <int W> void launch_col_count(const WarmArgs &wa, int col) {
    hipLaunchKernelGGL(col_count<W>, dim3(wa.d.nblk, wa.d.ng), dim3(BLOCK), 0, current_stream(), wa, col);
}
void launch_col_count_w(int W, const WarmArgs &wa, int col) {
    switch (W) {
    case 1: launch_col_count<1>(wa, col); break;
    case 2: launch_col_count<2>(wa, col); break;
    case 4: launch_col_count<4>(wa, col); break;
    case 8: launch_col_count<8>(wa, col); break;
    case 16: launch_col_count<16>(wa, col); break;
    case 32: launch_col_count<32>(wa, col); break;
    case 64: launch_col_count<64>(wa, col); break;
    default: PHMM_THROW(PHMM_EINTERNAL, "bad read-group width");
    }
}

}  // namespace

// PHMMModel::to_full_prob_reads without mappings: forward_sparse_score_only(use_max_ratio = true)
void full_prob_reads_sparse(phmm_model *m, const phmm_reads *reads, double *out_logp, double *out_total,
                            MappingSink *sink, bool by_ratio) {
    stats() = CallStats();
    const uint64_t R = reads->R;
    if (m->dev.max_degree > 8)
        PHMM_THROW(PHMM_EINVAL, "sparse path supports node degree <= 8 (MultiDbg MAX_DEGREE is 5)");
    ensure_logib(m, reads->max_len + 1);
    const phmm_params &prm = m->params;
    std::vector<double> lf(R, 0.0);
    std::vector<uint16_t> new_hint(R, 0);
    std::vector<uint32_t> new_flags(R, 0);  // PHMM_READ_* (written by the chunk that owns the read; chunks are disjoint)
    upload_reads(reads);

    // ---- work items.  A plan is a grouping of reads (W per group, longest first); it is cut into
    // chunks of read groups whose dense warm-up tables fit the worker's share of HBM.  Dense columns
    // kept per read group: the switch normally happens after ~log4(N/200) + a few columns, far before
    // n_warmup.  The first plan keeps 20; the few reads that are still dense there (e.g. a read that
    // starts with errors) are deferred to a small plan of their own that keeps all n_warmup+2 columns.
    //
    // Chunks are independent, and each one alternates between HBM-bound phases (dense warm-up, dense
    // backward) and latency-bound ones (one wave per read on the sparse frontier): up to MAX_WORKERS
    // host threads, each with its own stream and workspace set, run different chunks concurrently so
    // that the dense phase of one chunk fills the machine while another walks its frontiers.
    struct PlanCtx {
        Plan plan;
        int64_t lc_cap;
        bool may_defer;
        DevBuf d_order;
        uint64_t max_len = 0;
    };
    struct Item {
        PlanCtx *pc;
        int g0, ngc, Lc, Lfull;
    };
    std::mutex mu;
    // HBM-bound phases (dense warm-up, dense backward) run one at a time: a single dense kernel already
    // saturates the memory system, so overlapping two only interleaves their streams; the latency-bound
    // sparse phases of the other workers run underneath whichever chunk holds the token.
    std::mutex dense_token;
    std::condition_variable cv;
    std::deque<Item> queue;
    // Plans of deferred reads (a handful of reads, latency-bound from end to end) run on a side worker
    // with its own stream while the calling thread goes on with the next chunk of the main plan.
    std::deque<Item> side_queue;
    std::thread side_thread;
    bool side_started = false, main_done = false, single_mode = false;
    CallStats side_stats;
    const bool side_on = !knobs().no_side_worker;
    std::vector<std::unique_ptr<PlanCtx>> plans;
    int active = 0;
    std::exception_ptr first_error;
    // Default: one worker.  Measured on cfg3 (MI355X): 3 workers cut the step from 514 to 465 ms, but the
    // sparse waves that share the SIMDs with the dense kernels take bwd_step<64> from 4.0 to 2.9 TB/s;
    // until the frontier kernels are cheaper the pipeline stays opt-in (PHMM_WORKERS=2..4).
    int n_workers = knobs().workers;
    // Dense columns kept per read group by the first plan.  A read leaves the warm-up when at most warmup_threshold
    // (200) nodes are inside the ratio: with every base the count falls by ~4, so the switch comes ~log4(N / 200)
    // columns after the first ~10 (measured means: 15.1 at N = 1.3e5, 17.2 at N = 1.3e6); three more columns leave
    // ~1 % of the reads to the deferred plan (18 at cfg3; with a fixed 18, cfg5 deferred 10 % of its reads).
    int64_t warm_cols = 13 + (int64_t)std::ceil(std::log((double)std::max<uint64_t>(m->N, 201) / 200.0) / std::log(4.0));
    warm_cols = std::max<int64_t>(12, warm_cols);
    if (knobs().warm_cols > 0) warm_cols = std::max(4, knobs().warm_cols);
    const int chunk_groups = knobs().chunk_groups;  // 0: automatic
    // Memory plan of the call, fixed HERE (nothing below reads the free-memory counter again).  The budget is a share
    // of free + pool-owned bytes (the pool's buffers are grow-only and reused: all of it is this call's to use).
    // Set aside first: what the buffers that do not scale with the read groups need for this read set -- forward
    // record pool (~1 KB per sparse position, once per workspace set in use), mapping sink, control arrays, top-400
    // scratch -- or what they already hold if that is more.  The emit-prob planes of the mapping flow scale with the
    // groups and are part of the per-group cost below.  Then the deferred reads' plan (it runs BESIDE the main plan
    // on its own stream and workspace set): what the reads that needed more than the kept columns last time would
    // take with all n_warmup + 2 columns (no hints yet: 2 % of the reads), padded by a quarter, at most 1/8.  Too
    // small a share only cuts that plan into more chunks (or sends it behind the main plan), never fails.
    uint64_t min_len = UINT64_MAX;
    for (uint64_t r = 0; r < R; r++) min_len = std::min<uint64_t>(min_len, reads->off[r + 1] - reads->off[r]);
    // the plane of merged index `len` (Pb) is only written for reads that end inside the dense columns
    const int map_planes = sink ? (min_len <= (uint64_t)prm.n_warmup + 2 ? 3 : 2) : 0;
    uint64_t limit_total = 0, limit_side = 0;
    {
        const uint64_t est_fpool = reads->total * 1024 + R * 65536 + (1u << 20), est_meta = reads->total * 40,
                       est_ctl = R * (uint64_t)(PHMM_MAX_ACTIVE_NODES * 12 * 5 + WF_CAP * 12 + 4096) + reads->total,
                       est_sink = sink ? reads->total * 176 + R * 131072 + (1u << 21) : 0,
                       est_sel = (uint64_t)256 * m->N * 12;
        const uint64_t est = (est_fpool + est_meta + est_ctl) * 5 / 4 + est_sink + est_sel;
        uint64_t have = m->pool->ws_out.bytes;
        for (const auto &w : m->pool->wsets) {
            have += w.misc.bytes;
            for (int k = 0; k < 16; k++)
                if (k != 4) have += w.aux[k].bytes;  // (aux[4]: the emit-prob planes, per-group cost)
        }
        const uint64_t total = planned_budget(*m->pool), fixed = std::max(est, have);
        limit_total = total > fixed + ((uint64_t)64 << 20) ? total - fixed : (uint64_t)64 << 20;
        uint64_t n_def = std::max<uint64_t>(8, R / 50);
        if (by_ratio && reads->warm_hint.size() == R) {
            n_def = 8;
            for (uint16_t h : reads->warm_hint) n_def += h >= warm_cols ? 1 : 0;
        }
        const uint64_t per_read = ((uint64_t)(prm.n_warmup + 2) * 24 + 4 * 8 + (uint64_t)map_planes * 8) * m->N;
        // (lanes of that plan: its own read-group width pads a short list of reads up to a power of two)
        uint64_t lanes_def = 1;
        while (lanes_def < std::min<uint64_t>(n_def, 64)) lanes_def <<= 1;
        if (n_def > 64) lanes_def = (n_def + 63) / 64 * 64;
        limit_side = std::min<uint64_t>(limit_total / 8, lanes_def * per_read * 9 / 8);
    }
    const uint64_t limit_main = limit_total - limit_side;

    // cut a plan into items (caller holds `mu` or is the only thread)
    auto enqueue_plan = [&](std::unique_ptr<PlanCtx> pcu, std::deque<Item> &dst) {
        PlanCtx *pc = pcu.get();
        Plan &plan = pc->plan;
        for (uint32_t rd : plan.order) pc->max_len = std::max<uint64_t>(pc->max_len, reads->off[rd + 1] - reads->off[rd]);
        if (by_ratio && reads->warm_hint.size() == reads->R)
            // reads that stayed dense equally long last time share groups (see phmm_reads::warm_hint)
            std::stable_sort(plan.order.begin(), plan.order.end(),
                             [&](uint32_t x, uint32_t y) { return reads->warm_hint[x] > reads->warm_hint[y]; });
        const int W = plan.W;
        const size_t NW = (size_t)m->N * W;
        pc->d_order.upload(plan.order.data(), sizeof(uint32_t) * plan.order.size());
        HIP_CHECK(hipStreamSynchronize(current_stream()));
        int target = chunk_groups;
        if (target == 0) target = n_workers > 1 ? std::max(4, (plan.ng_total + 3 * n_workers - 1) / (3 * n_workers)) : plan.ng_total;
        // a plan of deferred reads runs BESIDE the main plan, whose tables hold most of the memory: its share was
        // set aside when the call started
        const uint64_t limit = (&dst == &side_queue) ? limit_side : limit_main / (uint64_t)n_workers;
        int g0 = 0;
        while (g0 < plan.ng_total) {
            // dense columns kept: at most n_warmup (+1 so that the launch that writes d of the last
            // dense column has somewhere to put its speculative next column)
            int Lc = (int)std::min<int64_t>((int64_t)pc->max_len, std::min<int64_t>(prm.n_warmup + 2, pc->lc_cap));
            if (pc->may_defer && g0 == 0 && n_workers == 1) {
                // The kept columns follow the memory: as many as ONE chunk of the whole plan affords, up to all
                // n_warmup + 2 -- a small read set (a shard of a multi-GPU run) then defers nothing, and the side plan,
                // a latency-bound chain of its own, cannot become the critical path.  A plan that misses one chunk
                // at the default keeps a column or two less instead (a few more reads are deferred): a second chunk
                // would add its whole frontier phase to the critical path.
                auto cost = [&](int lc) { return (size_t)lc * NW * 24 + 4 * NW * 8 + (sink ? (size_t)map_planes * NW * 8 + (size_t)plan.nblk8 * BLOCK * 8 : 0); };
                // (two settings only, all columns or the default: anything in between would follow the small changes of
                // the budget from call to call and re-allocate a quarter of a terabyte for one column more)
                const int lc_full = (int)std::min<int64_t>((int64_t)pc->max_len, prm.n_warmup + 2);
                if ((uint64_t)plan.ng_total * cost(lc_full) <= limit && !knobs().no_keep_all) Lc = lc_full;
                else
                    for (int cut = 0; cut <= 2 && Lc - cut >= 8; cut++)
                        if ((uint64_t)plan.ng_total * cost(Lc - cut) <= limit) {
                            Lc -= cut;
                            break;
                        }
            }
            // tables + (mapping flow) three emit-prob planes and the per-run maxima (mapping_flow.hip: pbuf)
            const size_t per_group = (size_t)Lc * NW * 24 + 4 * NW * 8 + (sink ? (size_t)map_planes * NW * 8 + (size_t)plan.nblk8 * BLOCK * 8 : 0);
            int ngc = (int)std::min<uint64_t>(plan.ng_total - g0, std::max<uint64_t>(1, limit / std::max<size_t>(per_group, 1)));
            ngc = std::min(ngc, std::max(1, target));
            // longest read of the chunk (the order need not be by length)
            int Lfull = 1;
            for (size_t slot = (size_t)g0 * W; slot < std::min<size_t>(plan.order.size(), (size_t)(g0 + ngc) * W); slot++) {
                const uint32_t rd = plan.order[slot];
                Lfull = std::max(Lfull, (int)(reads->off[rd + 1] - reads->off[rd]));
            }
            dst.push_back(Item{pc, g0, ngc, std::min(Lc, Lfull), Lfull});
            g0 += ngc;
        }
        plans.push_back(std::move(pcu));
    };

    std::function<void()> start_side_worker;
    auto run_chunk = [&](const Item &it) {
        hipStream_t s = current_stream();
        CallStats &st = stats();
        DevBuf &warm = m->wset().aux[0];  // per-chunk warm-up control arrays
        DevBuf &fpool = m->wset().aux[1], &fpool_meta = m->wset().aux[2];  // forward table records (generate_mappings)
        const Plan &plan = it.pc->plan;
        const DevBuf &d_order = it.pc->d_order;
        const int W = plan.W;
        const uint64_t R = plan.order.size();  // reads of THIS plan (slots beyond it are padding)
        const int g0 = it.g0, ngc = it.ngc, Lc = it.Lc, Lfull = it.Lfull;
        std::vector<uint32_t> deferred_ids;
        std::vector<uint32_t> *deferred = it.pc->may_defer ? &deferred_ids : nullptr;
        DenseArgs base{};
        fill_model_args(base, m);
        base.nblk = plan.nblk;
        base.nblk8 = plan.nblk8;
        base.npt = plan.npt;
        base.eall = 0;
        base.want_freq = 0;
        {
        DenseArgs a = base;
        a.ng = ngc;
        a.Lc = Lc;
        size_t tb = 0, mb = 0;
        layout(a, W, false, nullptr, nullptr, tb, mb);
        m->wset().tables.reserve(tb);
        m->wset().misc.reserve(mb);
        layout(a, W, false, m->wset().tables.p, m->wset().misc.p, tb, mb);
        HIP_CHECK(hipMemsetAsync(m->wset().misc.p, 0, mb, s));
        const int lanes = ngc * W;
        // warm-up control
        size_t wb = 0;
        auto carve = [&](size_t bytes) {
            wb = (wb + 255) / 256 * 256;
            size_t o = wb;
            wb += bytes;
            return o;
        };
        const size_t o_sw = carve(sizeof(int) * lanes), o_cnt = carve(sizeof(int) * lanes),
                     o_cn = carve(sizeof(int) * lanes), o_und = carve(sizeof(int) * 2),
                     o_amb = carve(sizeof(int) * lanes), o_wsub = carve(sizeof(int) * lanes),
                     o_wcnt = carve(sizeof(int) * lanes), o_mode = carve((size_t)lanes),
                     o_cnode = carve(sizeof(uint32_t) * (size_t)lanes * PHMM_MAX_ACTIVE_NODES),
                     o_ctot = carve(sizeof(double) * (size_t)lanes * PHMM_MAX_ACTIVE_NODES),
                     o_lanes = carve(sizeof(uint32_t) * lanes), o_out = carve(sizeof(double) * lanes),
                     o_err = carve(sizeof(uint32_t) * lanes),
                     o_wnode = carve(sizeof(uint32_t) * (size_t)lanes * WF_CAP),
                     o_wtot = carve(sizeof(double) * (size_t)lanes * WF_CAP),
                     o_bases = carve((size_t)ngc * Lfull * W);
        warm.reserve(wb);
        char *wp = (char *)warm.p;
        HIP_CHECK(hipMemsetAsync(wp, 0, o_cnode, s));
        HIP_CHECK(hipMemsetAsync(wp + o_sw, 0xff, sizeof(int) * lanes, s));  // sw = -1

        // staging on the device: the reads are resident (phmm_reads), the kernels want the bases
        // transposed to [group][pos][W] (dense: the Lc kept columns; sparse: the full length)
        trace("chunk setup");
        if (knobs().trace)
            std::fprintf(stderr, "      chunk: groups %d..%d of %d, W %d, kept columns %d, longest read %d\n", g0, g0 + ngc, plan.ng_total, W, Lc, Lfull);
        std::vector<int> hl((size_t)lanes, 0);
        uint64_t dense_cells = 0;
        for (int gi = 0; gi < lanes; gi++) {
            const size_t slot = (size_t)g0 * W + gi;
            if (slot >= R) continue;
            const uint32_t rd = plan.order[slot];
            hl[gi] = (int)(reads->off[rd + 1] - reads->off[rd]);
        }
        std::vector<double> hib;
        host_logib(m, (size_t)Lc, hib);
        // the dense kernels see lengths clamped to the kept columns; the true lengths are
        // restored for the sparse kernel below
        std::vector<int> hlc(hl);
        for (auto &v : hlc) v = std::min(v, Lc);
        {
            StageArgs sa{};
            sa.bases = reads->d_bases.as<uint8_t>();
            sa.off = reads->d_off.as<uint64_t>();
            sa.order = d_order.as<uint32_t>() + (size_t)g0 * W;
            sa.n_slots = (uint32_t)(R - std::min<uint64_t>(R, (uint64_t)g0 * W));
            sa.W = W;
            sa.Lc = Lc;
            sa.Lfull = Lfull;
            sa.out_dense = (uint8_t *)a.bases;
            sa.out_full = (uint8_t *)(wp + o_bases);
            const unsigned per_group = (unsigned)Lfull * W;
            hipLaunchKernelGGL(stage_bases, dim3((per_group + BLOCK - 1) / BLOCK, ngc), dim3(BLOCK), 0, s, sa);
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipMemcpyAsync((void *)a.len, hlc.data(), hlc.size() * sizeof(int), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync((void *)a.logib, hib.data(), hib.size() * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));

        WarmArgs wa{};
        wa.d = a;
        wa.sw = (int *)(wp + o_sw);
        wa.cnt = (int *)(wp + o_cnt);
        wa.cand_n = (int *)(wp + o_cn);
        wa.undecided = (int *)(wp + o_und);
        wa.n_amb = wa.undecided + 1;
        wa.amb = (int *)(wp + o_amb);
        a.wf_sw = by_ratio ? wa.sw : nullptr;
        a.wf_mode = (const uint8_t *)(wp + o_mode);
        a.wf_sub = (int *)(wp + o_wsub);
        a.wf_cnt = (int *)(wp + o_wcnt);
        a.wf_node = (uint32_t *)(wp + o_wnode);
        a.wf_tot = (double *)(wp + o_wtot);
        a.wf_ratio = std::exp(-prm.active_node_max_ratio);
        a.wf_ub_a = m->wf_ub_a;
        a.wf_ub_b = m->wf_ub_b;
        wa.d = a;
        wa.cand_node = (uint32_t *)(wp + o_cnode);
        wa.cand_tot = (double *)(wp + o_ctot);
        wa.ratio_lin = std::exp(-prm.active_node_max_ratio);
        wa.n_warmup = (int)prm.n_warmup;
        wa.threshold = (int)prm.warmup_threshold;

        trace("staging+upload");
        // ---- dense warm-up with per-read switch decisions
        int pos = 0;
        // per-launch statistics (bench.py's roofline) cover the full-width instantiation fwd_step<64> only
        const bool st_on = W == 64 && (it.pc->may_defer || !by_ratio);  // (not the few-read plans beside the main one)
        LaunchTimer lt(timing_enabled() && st_on);
        std::unique_lock<std::mutex> dense_lock(dense_token, std::defer_lock);
        // (plans of deferred reads are a handful of lanes: they neither take nor wait for the token)
        const bool use_token = it.pc->may_defer || !by_ratio;
        if (use_token) dense_lock.lock();
        if (!by_ratio) {
            // not adaptive (forward.rs:134-137): the first n_warmup tables are dense for every read; launch
            // min(n_warmup, Lc) completes column n_warmup-1 (its Del values) or ends the short reads
            const int last = std::min<int>((int)prm.n_warmup, Lc);
            for (pos = 0; pos <= last; pos++) {
                lt.begin();
                launch_fwd_step(W, a, pos);
                lt.end();
                if (st_on) st.launches[0]++;
            }
            std::vector<int> fsw(lanes);
            for (int gi = 0; gi < lanes; gi++) fsw[gi] = std::min<int>((int)prm.n_warmup, hl[gi]);
            HIP_CHECK(hipMemcpyAsync(wa.sw, fsw.data(), sizeof(int) * lanes, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipStreamSynchronize(s));
        }
        for (; by_ratio; pos++) {
            lt.begin();
            launch_fwd_step(W, a, pos);  // column pos (if pos < Lc), d + totals maximum of column pos-1
            lt.end();
            if (st_on) st.launches[0]++;
            HIP_CHECK(hipMemsetAsync(wa.undecided, 0, sizeof(int) * 2, s));
            hipLaunchKernelGGL(warm_decide_fused, dim3((lanes + BLOCK / 64 - 1) / (BLOCK / 64)), dim3(BLOCK), 0, s, wa, pos,
                               lanes, W);
            int und[2] = {0, 0};
            HIP_CHECK(hipMemcpyAsync(und, wa.undecided, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            if (und[1] > 0) {
                // lanes the fused count could not settle: exact count of column pos-1, then the plain decision
                launch_col_count_w(W, wa, pos - 1);
                hipLaunchKernelGGL(warm_decide, dim3((lanes + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, wa, pos, lanes);
                HIP_CHECK(hipMemcpyAsync(und, wa.undecided, sizeof(int), hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                st.launches[3]++;
            }
            if (und[0] == 0) break;
            if (pos >= Lc) PHMM_THROW(PHMM_EINTERNAL, "warm-up did not terminate");
        }
        st.ms[0] += lt.total_ms();
        if (dense_lock.owns_lock()) dense_lock.unlock();
        trace("dense warm-up");
        // reads that ended inside the warm-up: fe of their last (dense) column
        launch_fwd_finish(W, a);
        std::vector<int> hsw(lanes);
        std::vector<int> hcn(lanes);
        std::vector<double> tlf(lanes);
        HIP_CHECK(hipMemcpyAsync(hsw.data(), wa.sw, sizeof(int) * lanes, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(hcn.data(), wa.cand_n, sizeof(int) * lanes, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(tlf.data(), a.logPf, sizeof(double) * lanes, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        {
            // a read that ran out of kept dense columns before its switch: redo the chunk with all of them
            for (int gi = 0; gi < lanes; gi++)
                if (hl[gi] > Lc && hsw[gi] >= Lc) {
                    if (!deferred) PHMM_THROW(PHMM_EINTERNAL, "warm-up ran past n_warmup");
                    const size_t slot = (size_t)g0 * W + gi;
                    deferred->push_back(plan.order[slot]);
                    new_flags[plan.order[slot]] |= PHMM_READ_DEFERRED;
                    if (knobs().trace)
                        std::fprintf(stderr, "      read %u (len %d) still dense at column %d: deferred\n", plan.order[slot],
                                     hl[gi], Lc);
                    hl[gi] = 0;  // not part of this chunk any more
                }
        }
        // A small plan of reads this chunk gives up (all n_warmup + 2 columns, nothing deferred again) runs on the side
        // stream under the rest of this chunk; where that is not possible it queues behind the main plan.
        auto hand_over = [&](const std::vector<uint32_t> &ids) -> bool {
            std::unique_ptr<PlanCtx> pc(new PlanCtx{make_plan_ids(m, reads, ids), (int64_t)prm.n_warmup + 2, false, {}});
            // (a group of that plan that does not fit the side budget waits for the main plan's tables instead)
            const size_t side_group = ((size_t)(prm.n_warmup + 2) * 24 + 4 * 8 + (size_t)map_planes * 8) * m->N * pc->plan.W;
            std::lock_guard<std::mutex> lk(mu);
            const bool beside = single_mode && side_on && side_group <= limit_side;
            if (beside) {
                enqueue_plan(std::move(pc), side_queue);
                if (!side_started) {
                    side_started = true;
                    start_side_worker();
                }
            } else {
                enqueue_plan(std::move(pc), queue);
            }
            cv.notify_all();
            return beside;
        };
        // hand the deferred reads over NOW
        if (!deferred_ids.empty()) hand_over(deferred_ids);
        std::vector<uint32_t> sparse_lanes, need400;
        for (int gi = 0; gi < lanes; gi++) {
            if (hl[gi] == 0) continue;
            dense_cells += (uint64_t)std::min(hsw[gi] + 1, hl[gi]) * m->N;
            if (hsw[gi] < hl[gi]) {
                // ratio mode: a forced switch with more than 400 nodes inside the ratio; fixed mode: always
                // (top_nodes(n_active_nodes) of the last dense column)
                if (!by_ratio || hcn[gi] > PHMM_MAX_ACTIVE_NODES) {
                    need400.push_back((uint32_t)gi);
                    if (by_ratio) new_flags[plan.order[(size_t)g0 * W + gi]] |= PHMM_READ_FORCED_SWITCH;
                }
                sparse_lanes.push_back((uint32_t)gi);
            }
        }
        if (st_on) st.cells[0] += dense_cells;
        DevBuf sel;
        for (size_t nb0 = 0; nb0 < need400.size(); nb0 += 256) {
            // keep the K best of the switch column (scratch: N candidates per read, 256 reads at a time)
            const size_t nn = std::min<size_t>(256, need400.size() - nb0);
            size_t sb = 0;
            auto c2 = [&](size_t bytes) {
                sb = (sb + 255) / 256 * 256;
                size_t o = sb;
                sb += bytes;
                return o;
            };
            const size_t p_need = c2(sizeof(uint32_t) * nn), p_n = c2(sizeof(int) * nn),
                         p_node = c2(sizeof(uint32_t) * nn * m->N), p_tot = c2(sizeof(double) * nn * m->N);
            sel.reserve(sb);
            char *sp = (char *)sel.p;
            HIP_CHECK(hipMemcpyAsync(sp + p_need, need400.data() + nb0, sizeof(uint32_t) * nn, hipMemcpyHostToDevice, s));
            Top400Args ta{};
            ta.d = a;
            ta.W = W;
            ta.sw = wa.sw;
            ta.need = (const uint32_t *)(sp + p_need);
            ta.sc_node = (uint32_t *)(sp + p_node);
            ta.sc_tot = (double *)(sp + p_tot);
            ta.sc_n = (int *)(sp + p_n);
            ta.cand_node = wa.cand_node;
            ta.cand_tot = wa.cand_tot;
            ta.cand_n = wa.cand_n;
            ta.ratio_lin = by_ratio ? wa.ratio_lin : 0.0;
            ta.K = by_ratio ? PHMM_MAX_ACTIVE_NODES : (int)std::min<int64_t>(prm.n_active_nodes, PHMM_MAX_ACTIVE_NODES);
            hipLaunchKernelGGL(select_top400, dim3((unsigned)nn), dim3(BLOCK), 0, s, ta);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(s));
        }

        trace("switch bookkeeping");
        // ---- sparse continuation, one wave per read
        std::vector<double> slp(lanes, 0.0);
        if (!sparse_lanes.empty()) {
            // the sparse kernel indexes lengths and bases of the full reads
            HIP_CHECK(hipMemcpyAsync((void *)a.len, hl.data(), hl.size() * sizeof(int), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(wp + o_lanes, sparse_lanes.data(), sparse_lanes.size() * sizeof(uint32_t),
                                     hipMemcpyHostToDevice, s));
            SparseFwdArgs fa{};
            fa.M = sparse_model_of(m);
            fa.d = a;
            fa.d.Lc = Lc;
            fa.W = W;
            fa.sw = wa.sw;
            fa.cand_node = wa.cand_node;
            fa.cand_tot = wa.cand_tot;
            fa.cand_n = wa.cand_n;
            fa.lanes = (const uint32_t *)(wp + o_lanes);
            fa.bases = (const uint8_t *)(wp + o_bases);
            fa.Lb = Lfull;
            fa.ratio_lin = wa.ratio_lin;
            fa.topk = by_ratio ? 0 : (int)std::min<int64_t>(prm.n_active_nodes, PHMM_MAX_ACTIVE_NODES);
            fa.out_logp = (double *)(wp + o_out);
            fa.err = (uint32_t *)(wp + o_err);
            // table storage for generate_mappings: one record per sparse position
            std::vector<uint64_t> lane_pos0(lanes + 1, 0);
            for (int gi = 0; gi < lanes; gi++) lane_pos0[gi + 1] = lane_pos0[gi] + (uint64_t)hl[gi];
            const uint64_t n_pos = lane_pos0[lanes];
            uint64_t sparse_pos = 0;
            for (uint32_t gi : sparse_lanes) sparse_pos += (uint64_t)(hl[gi] - hsw[gi]);
            uint64_t pool_cap = std::max<uint64_t>(fpool.bytes, sparse_pos * 1024 + (uint64_t)sparse_lanes.size() * 65536 + (1u << 20));
            const size_t o_stop = o_cnt;  // the per-lane count array of the warm-up is free again
            fa.stop = (int *)(wp + o_stop);
            for (int attempt = 0;; attempt++) {
                fpool.reserve(pool_cap + 4096);  // (the backward kernel fetches 1 KB from a record's start whatever its size)
                fpool_meta.reserve(sizeof(unsigned long long) + sizeof(uint64_t) * (n_pos + 1) * 2);
                HIP_CHECK(hipMemsetAsync(fpool_meta.p, 0, sizeof(unsigned long long) + sizeof(uint64_t) * (n_pos + 1) * 2, s));
                fa.pool.base = fpool.as<uint8_t>();
                fa.pool.cap = pool_cap;
                fa.pool.top = fpool_meta.as<unsigned long long>();
                fa.pool.off = (uint64_t *)(fpool_meta.as<char>() + 8);
                uint64_t *d_lp0 = fa.pool.off + n_pos;
                HIP_CHECK(hipMemcpyAsync(d_lp0, lane_pos0.data(), sizeof(uint64_t) * lanes, hipMemcpyHostToDevice, s));
                fa.lane_pos0 = d_lp0;
                // phase A <400>: the first positions after the switch (frontier up to 400 nodes);
                // phase B <64>: the rest; phase C <400>: whatever phase B could not hold
                std::vector<uint32_t> todo = sparse_lanes;
                std::vector<uint32_t> herr(lanes);
                std::vector<int> hstop(lanes);
                bool pool_full = false;
                const bool lean_ok =
                    by_ratio && m->dev.max_degree <= (uint32_t)ADJ_DEG && !knobs().no_lean;
                // one launch of a phase over `who`: 0 = A <400> from the dense column (6 positions), 1 = B (one lane per
                // node; `steps` positions at most, 0 = to the end), 2 = C <400> burst of `steps` positions.
                // -> the lanes that still have positions left; `wide`: those of them that stopped because their frontier
                // did not fit the class (phase B only)
                const bool wide_ok = lean_ok && !knobs().no_wide_class;
                auto run_phase = [&](int phase, int steps, const std::vector<uint32_t> &who, std::vector<uint32_t> &rest,
                                     std::vector<uint32_t> &wide) {
                    rest.clear();
                    wide.clear();
                    // kind 0: one lane per node (lean_fwd_kernel.h), 1: the 400-slot vectors on a block of 448 threads
                    // (wide_fwd_kernel.h), 2: the same on one wave (generic), 3: the generic 128-slot one
                    auto launch = [&](int kind, int mode, const std::vector<uint32_t> &l) {
                        HIP_CHECK(hipMemcpyAsync(wp + o_lanes, l.data(), l.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
                        fa.mode = mode;
                        fa.max_steps = steps;
                        const dim3 grid((unsigned)l.size());
                        if (kind == 0) hipLaunchKernelGGL(lean_forward_kernel, grid, dim3(64), 0, s, fa);
                        else if (kind == 1) hipLaunchKernelGGL(wide_forward_kernel, grid, dim3(WFK_T), 0, s, fa);
                        else if (kind == 3) hipLaunchKernelGGL((sparse_forward_kernel<128>), grid, dim3(64), 0, s, fa);
                        else hipLaunchKernelGGL((sparse_forward_kernel<PHMM_MAX_ACTIVE_NODES>), grid, dim3(64), 0, s, fa);
                        HIP_CHECK(hipGetLastError());
                        st.launches[2]++;
                        HIP_CHECK(hipMemcpyAsync(herr.data(), fa.err, sizeof(uint32_t) * lanes, hipMemcpyDeviceToHost, s));
                        HIP_CHECK(hipMemcpyAsync(hstop.data(), fa.stop, sizeof(int) * lanes, hipMemcpyDeviceToHost, s));
                        HIP_CHECK(hipStreamSynchronize(s));
                    };
                    // B class: graphs beyond the lean kernel's degree bound use the generic 128-slot vector kernel
                    if (phase == 1) launch(lean_ok ? 0 : 3, 1, who);
                    else launch(wide_ok ? 1 : 2, phase == 0 ? 0 : 1, who);  // A / C: the 400-slot class
                    for (uint32_t gi : who) {
                        if (herr[gi] & SP_ERR_POOL) pool_full = true;
                        else if ((herr[gi] & SP_ERR_CAPACITY) && phase != 1)
                            PHMM_THROW(PHMM_ECAPACITY, "sparse forward: frontier does not fit 400 slots");
                        else if (herr[gi] & ~SP_ERR_CAPACITY)
                            PHMM_THROW(PHMM_EINTERNAL, "sparse forward error " + std::to_string(herr[gi]));
                        if (hstop[gi] < hl[gi]) ((phase == 1 && (herr[gi] & SP_ERR_CAPACITY)) ? wide : rest).push_back(gi);
                    }
                    trace(phase == 0 ? "   phase A <400>" : (phase == 1 ? "   phase B <64>" : "   phase C <400>"));
                    if (knobs().trace) std::fprintf(stderr, "      remaining lanes %zu (+ %zu wide)\n", rest.size(), wide.size());
                };
                // A <400> from the dense column; then B to the end of the read.  A read whose frontier outgrows B's
                // class takes a short C <400> burst and returns to B; C bursts grow (8, 16, ... 512 positions) so that
                // a frontier that stays wide still ends.  Long reads walk B in slices: a read that goes wide is seen
                // -- and, on the main plan, handed to a plan of its own on the side stream -- at the end of its slice
                // and not after every other read of the chunk has walked all of its 10 000 positions.
                const bool can_hand_over = deferred && by_ratio && single_mode && side_on && !knobs().no_wide_handover;
                const int b_slice = (can_hand_over && Lfull > 3072) ? 2048 : 0;
                size_t handed = 0;
                int bursts = 0;
                std::vector<uint32_t> rest, wide, tmp1, tmp2;
                run_phase(0, 6, todo, rest, wide);
                todo.swap(rest);
                for (int turn = 0; turn < 100000 && !todo.empty() && !pool_full; turn++) {
                    run_phase(1, b_slice, todo, rest, wide);
                    if (pool_full) break;
                    todo.swap(rest);
                    if (wide.empty()) continue;
                    for (uint32_t gi : wide) new_flags[plan.order[(size_t)g0 * W + gi]] |= PHMM_READ_WIDE_FRONTIER;
                    if (can_hand_over && handed + wide.size() <= std::max<size_t>(8, (size_t)lanes / 64)) {
                        // The few reads whose frontier outgrew the one-lane-per-node class (5 of 4 026 on cfg3) would
                        // now take a 400-slot burst and then walk the rest of the read ALONE -- pure latency on this
                        // chunk's critical path, and as much again in the backward pass.  They leave the chunk instead
                        // and are done from their first base in a plan of their own on the side stream, beside this
                        // chunk's remaining phases (the same route the deferred reads take).
                        std::vector<uint32_t> ids;
                        for (uint32_t gi : wide) ids.push_back(plan.order[(size_t)g0 * W + gi]);
                        for (uint32_t gi : wide) hl[gi] = 0;  // not part of this chunk any more
                        sparse_lanes.erase(std::remove_if(sparse_lanes.begin(), sparse_lanes.end(), [&](uint32_t gi) { return hl[gi] == 0; }),
                                           sparse_lanes.end());
                        HIP_CHECK(hipMemcpyAsync((void *)a.len, hl.data(), hl.size() * sizeof(int), hipMemcpyHostToDevice, s));
                        hand_over(ids);
                        handed += ids.size();
                        if (knobs().trace) std::fprintf(stderr, "      %zu wide reads handed to a plan of their own\n", ids.size());
                        continue;
                    }
                    run_phase(2, 8 << std::min(bursts, 6), wide, tmp1, tmp2);
                    bursts++;
                    todo.insert(todo.end(), tmp1.begin(), tmp1.end());
                }
                if (!pool_full && !todo.empty()) PHMM_THROW(PHMM_EINTERNAL, "sparse forward did not finish");
                HIP_CHECK(hipMemcpyAsync(slp.data(), fa.out_logp, sizeof(double) * lanes, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                if (!pool_full) break;
                if (attempt >= 4) PHMM_THROW(PHMM_ENOMEM, "forward table pool keeps overflowing");
                pool_cap *= 2;
            }
            if (sink) {
                MapChunk mc{};
                mc.m = m;
                mc.W = W;
                mc.Lc = Lc;
                mc.Lfull = Lfull;
                mc.ngc = ngc;
                mc.lanes = lanes;
                mc.a = a;
                mc.fa_M = fa.M;
                mc.d_sw = wa.sw;
                mc.d_bases_full = fa.bases;
                mc.fpool = fa.pool;
                mc.d_lane_pos0 = fa.lane_pos0;
                mc.hl = &hl;
                mc.hsw = &hsw;
                mc.lane_pos0 = &lane_pos0;
                mc.ratio_lin = by_ratio ? wa.ratio_lin : 0.0;
                mc.topk = by_ratio ? 0 : (int)std::min<int64_t>(prm.n_active_nodes, PHMM_MAX_ACTIVE_NODES);
                mc.d_logp_sparse = fa.out_logp;
                mc.cand_node = wa.cand_node;
                mc.cand_tot = wa.cand_tot;
                mc.dense_token = use_token ? &dense_token : nullptr;
                mc.main_plan = it.pc->may_defer || !by_ratio;
                trace("sparse forward");
                mapping_backward_chunk(mc, sparse_lanes, sink, plan, g0, R);
                trace("mapping backward total");
            }
        } else if (sink) {
            // every read of the chunk ended inside the dense warm-up
            std::vector<uint64_t> lane_pos0(lanes + 1, 0);
            for (int gi = 0; gi < lanes; gi++) lane_pos0[gi + 1] = lane_pos0[gi] + (uint64_t)hl[gi];
            HIP_CHECK(hipMemcpyAsync((void *)a.len, hl.data(), hl.size() * sizeof(int), hipMemcpyHostToDevice, s));
            MapChunk mc{};
            mc.m = m;
            mc.W = W;
            mc.Lc = Lc;
            mc.Lfull = Lfull;
            mc.ngc = ngc;
            mc.lanes = lanes;
            mc.a = a;
            mc.fa_M = sparse_model_of(m);
            mc.d_sw = wa.sw;
            mc.d_bases_full = (const uint8_t *)(wp + o_bases);
            mc.hl = &hl;
            mc.hsw = &hsw;
            mc.lane_pos0 = &lane_pos0;
            mc.ratio_lin = by_ratio ? wa.ratio_lin : 0.0;
                mc.topk = by_ratio ? 0 : (int)std::min<int64_t>(prm.n_active_nodes, PHMM_MAX_ACTIVE_NODES);
            mc.d_logp_sparse = (double *)(wp + o_out);
            mc.cand_node = wa.cand_node;
            mc.cand_tot = wa.cand_tot;
                mc.dense_token = use_token ? &dense_token : nullptr;
                mc.main_plan = it.pc->may_defer || !by_ratio;
            mapping_backward_chunk(mc, sparse_lanes, sink, plan, g0, R);
        }
        for (int gi = 0; gi < lanes; gi++) {
            const size_t slot = (size_t)g0 * W + gi;
            if (slot >= R || hl[gi] == 0) continue;
            const uint32_t rd = plan.order[slot];
            lf[rd] = hsw[gi] < hl[gi] ? slp[gi] : tlf[gi];
            new_hint[rd] = (uint16_t)std::min(hsw[gi], 65535);
        }
        }
    };  // run_chunk
    start_side_worker = [&]() {
        const ThreadContext ctx = capture_thread_context();
        if (!m->pool->wstream[1]) HIP_CHECK(hipStreamCreateWithFlags(&m->pool->wstream[1], hipStreamNonBlocking));
        side_thread = std::thread([&, ctx]() {
            adopt_thread_context(ctx, m->pool->wstream[1], 1);
            for (;;) {
                Item it{};
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !side_queue.empty() || main_done || first_error; });
                    if (first_error) break;
                    if (side_queue.empty()) break;  // main_done
                    it = side_queue.front();
                    side_queue.pop_front();
                }
                try {
                    run_chunk(it);
                } catch (...) {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!first_error) first_error = std::current_exception();
                }
            }
            (void)hipStreamSynchronize(m->pool->wstream[1]);
            std::lock_guard<std::mutex> lk(mu);
            side_stats = stats();
        });
    };

    auto worker_loop = [&]() {
        for (;;) {
            Item it{};
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !queue.empty() || active == 0 || first_error; });
                if (first_error || queue.empty()) break;  // queue empty and nobody can add to it
                it = queue.front();
                queue.pop_front();
                active++;
            }
            try {
                run_chunk(it);
            } catch (...) {
                std::lock_guard<std::mutex> lk(mu);
                if (!first_error) first_error = std::current_exception();
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                active--;
            }
            cv.notify_all();
        }
    };

    {
        // fixed mode: every read keeps all n_warmup dense columns, nothing is deferred
        std::unique_ptr<PlanCtx> pc(
            new PlanCtx{make_plan(m, reads, 0), by_ratio ? warm_cols : (int64_t)prm.n_warmup + 2, by_ratio, {}});
        // few groups: one worker (the calling thread, on the caller's stream)
        const int min_groups = knobs().pipeline_min_groups;
        if (pc->plan.ng_total < min_groups) n_workers = 1;
        enqueue_plan(std::move(pc), queue);
    }
    const int n_threads = (int)std::min<size_t>((size_t)n_workers, queue.size());
    if (n_threads <= 1) {
        single_mode = true;
        worker_loop();
        {
            std::lock_guard<std::mutex> lk(mu);
            main_done = true;
        }
        cv.notify_all();
        if (side_started && side_thread.joinable()) {
            side_thread.join();
            CallStats &st = stats();
            for (int k = 0; k < 4; k++) {
                st.ms[k] += side_stats.ms[k];
                st.launches[k] += side_stats.launches[k];
                st.cells[k] += side_stats.cells[k];
            }
        }
    } else {
        HIP_CHECK(hipStreamSynchronize(current_stream()));  // uploads / memsets the chunks depend on
        const ThreadContext ctx = capture_thread_context();
        CallStats merged;
        // every stream exists before the first thread does: a failing create must not unwind past joinable threads
        for (int t = 0; t < n_threads; t++)
            if (!m->pool->wstream[t]) HIP_CHECK(hipStreamCreateWithFlags(&m->pool->wstream[t], hipStreamNonBlocking));
        std::vector<std::thread> threads;
        struct JoinAll {
            std::vector<std::thread> &t;
            ~JoinAll() {
                for (auto &th : t)
                    if (th.joinable()) th.join();
            }
        } join_all{threads};
        threads.reserve(n_threads);
        for (int t = 0; t < n_threads; t++) {
            threads.emplace_back([&, t]() {
                adopt_thread_context(ctx, m->pool->wstream[t], t);
                worker_loop();
                (void)hipStreamSynchronize(m->pool->wstream[t]);
                std::lock_guard<std::mutex> lk(mu);
                const CallStats &ws = stats();
                for (int k = 0; k < 4; k++) {
                    merged.ms[k] += ws.ms[k];
                    merged.launches[k] += ws.launches[k];
                    merged.cells[k] += ws.cells[k];
                }
            });
        }
        for (auto &th : threads) th.join();
        stats() = merged;
    }
    if (first_error) std::rethrow_exception(first_error);
    if (by_ratio) {
        reads->warm_hint = new_hint;
        reads->last_flags = new_flags;
    }
    double tot = 0.0;
    for (uint64_t r = 0; r < R; r++) tot += lf[r];
    put_doubles(out_logp, lf.data(), R);
    put_doubles(out_total, &tot, 1);
}

}  // namespace phmm
