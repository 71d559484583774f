// generate_mappings without input mappings (src/hmmv2/hint.rs:193-220):
//   run_sparse_adaptive (freq.rs:60-68) = forward_sparse (sparse_dyn.hip) +
//   backward_by_forward (backward.rs:101-142), then
//   to_mapping_by_score_ratio (hint.rs:135-142): for every read position the nodes of
//   S = F (.) B / P (table.rs:500-505) within `active_node_max_ratio` of the best, sorted.
//
// Split of the work
//   * sparse tail of a read (positions >= its dense/sparse switch s0): one wave64 per read,
//     B columns over filled_nodes(F.tables[i-1]) (table.rs:117-123), frontier in LDS;
//   * dense head (positions < s0, and whole reads that never left the warm-up): the batched
//     dense backward kernel (dense.hip) with the emit probs of a column kept in a side
//     buffer, post_collect gathers the nodes inside the ratio, emit_dense_map sorts them.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

#include <hipcub/hipcub.hpp>

#include "sparse_dyn.h"
#include "lean_bwd_kernel.h"
#include "wide_bwd_kernel.h"

namespace phmm {

static constexpr int KMAX = PHMM_MAX_ACTIVE_NODES;

// lighter than FVec: a loaded forward-table record
template <int CAP> struct FRec {
    double m[CAP], i[CAP], d[CAP];
    uint32_t id[CAP];
    int n, na, E;
};

template <int CAP> __device__ bool load_record(const RecPool &p, uint64_t pos_index, FRec<CAP> &f) {
    const uint64_t o1 = p.off[pos_index];
    if (o1 == 0) return false;
    const uint8_t *rec = p.base + (o1 - 8);
    const int *hw = (const int *)rec;
    const int n = hw[0], na = hw[1], E = hw[2];
    if (n > CAP) return false;
    if (threadIdx.x == 0) {
        f.n = n;
        f.na = na;
        f.E = E;
    }
    const uint64_t idb = (uint64_t)((n + 1) & ~1) * 4;
    const uint32_t *ids = (const uint32_t *)(rec + 16);
    const double *m = (const double *)(rec + 16 + idb), *i = m + na, *d = i + na;
    for (int j = threadIdx.x; j < n; j += 64) {
        f.id[j] = ids[j];
        f.d[j] = d[j];
        f.m[j] = j < na ? m[j] : 0.0;
        f.i[j] = j < na ? i[j] : 0.0;
    }
    wave_sync();
    return true;
}

// to_mapping_by_score_ratio for one position (table.rs:134-149, 163-169): sort val desc
// (equal values by node id -- every caller passes by_node: a list must not depend on the slot order its read's
// nodes happened to have; stable in slot order otherwise), keep while ln p0 - ln p < ratio,
// write [n][pad] ids logp.  Returns false if the pool is full.
template <int CAP>
__device__ bool emit_mapping(const RecPool &mp, uint64_t pos_index, const uint32_t *ids, double *val, int n,
                             double ratio_lin, bool by_node, uint16_t *order, int topk = 0, long long prealloc = -1) {
    // Only the entries that stay need a rank: with the ratio rule they are the largest ones, so their rank among
    // all entries is their rank among themselves (a column next to the dense/sparse switch has up to 400
    // entries of which a handful stay).
    // The list is ordered by the values it holds -- the LOGS -- and equal logs by node id: two nodes whose linear
    // values differ in the last bit (the two haplotype copies of a k-mer) usually share one log value, and which of
    // the two is a bit larger depends on the order of additions, i.e. on how the reads were grouped.  Ordered this
    // way a list is the same whatever the grouping.  val[] is overwritten: log value of an entry that stays, NaN
    // (never ranked, compares false) otherwise.
    double thr = -1.0;  // topk: every entry is ranked
    if (topk <= 0) {
        double mx = 0.0;
        for (int j = threadIdx.x; j < n; j += 64) mx = fmax(mx, val[j]);
        thr = wave_max(mx) * ratio_lin;
    }
    __syncthreads();
    int c = 0;
    for (int j = threadIdx.x; j < n; j += 64) {
        const double v = val[j];
        const bool stay = topk > 0 || (v > 0.0 && v > thr);
        val[j] = stay ? (v > 0.0 ? log(v) : -INFINITY) : __longlong_as_double(0x7ff8000000000000ll);
        c += stay ? 1 : 0;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += 64) {
        const double v = val[j];
        if (v != v) continue;
        const uint32_t id = ids[j];
        int rank = 0;
        for (int q0 = 0; q0 < n; q0 += 8) {  // (eight LDS reads in flight)
            double u[8];
            uint32_t ui[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int q = q0 + k < n ? q0 + k : n - 1;
                u[k] = val[q];
                ui[k] = by_node ? ids[q] : (uint32_t)q;
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
                rank += (q0 + k < n) && ((u[k] > v) || (u[k] == v && (by_node ? ui[k] < id : (int)ui[k] < j)));
        }
        order[rank] = (uint16_t)j;
    }
    __syncthreads();
    int keep = 0;
    if (topk > 0) keep = topk < n ? topk : n;  // to_mapping(n_active_nodes): the k best whatever their value (hint.rs:124-131)
    else keep = wave_isum(c);
    const uint64_t idb = (uint64_t)((keep + 1) & ~1) * 4;
    const uint64_t bytes = 8 + idb + (uint64_t)keep * 8;
    // (prealloc: the caller reserved room for this record -- emit_offsets -- instead of one atomic per record)
    const uint64_t o = prealloc >= 0 ? (uint64_t)prealloc : pool_alloc(mp, bytes);
    if (o + bytes > mp.cap) return false;
    uint8_t *rec = mp.base + o;
    if (threadIdx.x == 0) {
        ((uint32_t *)rec)[0] = (uint32_t)keep;
        ((uint32_t *)rec)[1] = 0;
        mp.off[pos_index] = o + 8;
    }
    uint32_t *oid = (uint32_t *)(rec + 8);
    double *olp = (double *)(rec + 8 + idb);
    for (int j = threadIdx.x; j < keep; j += 64) {
        const int s = order[j];
        oid[j] = ids[s];
        olp[j] = val[s];
    }
    __syncthreads();
    return true;
}

template <int CAP>
__global__ void __launch_bounds__(64) sparse_backward_kernel(const SparseBwdArgs a) {
    __shared__ FRec<CAP> fr;
    __shared__ Col<CAP> cols[2];
    __shared__ double dA[CAP], dB[CAP];
    __shared__ uint32_t list[CAP];
    // `val` (node totals before a step, posteriors after it) and `order` (sort scratch, before and after) are never
    // live while bwd_list_step uses its level buffers dA / dB: they share their memory.  4 KB less: the 400-slot
    // class at 53.9 KB instead of 57.9, three waves per CU instead of two.
    double *const val = dA;
    uint16_t *const order = (uint16_t *)dB;
    const int lane = threadIdx.x;
    const uint32_t gi = a.lanes[blockIdx.x];
    const int g = (int)(gi / a.W), r = (int)(gi % a.W);
    const int len = a.d.len[gi];
    const int s0 = a.sw[gi];
    const uint64_t p0 = a.lane_pos0[gi];
    const uint64_t q0 = a.map_pos0[gi];
    const double logP = a.d.logPf[gi];
    const LinParams &lp = a.M.lp;
    uint32_t err = 0;
    const bool ok = logP > -INFINITY;
    int pos;            // next position to compute
    int have_cols = 0;  // cols[(pos+1)&1] holds B.tables[pos+1]
    bool stopped = false;
    int stop_at = 0;
    if (a.mode == 0) {
        pos = len - 1;
        // merged index len: F.tables[len-1] (.) b_init / P   (table.rs:414-434, backward.rs:197-211)
        if (!load_record<CAP>(a.fpool, p0 + (uint64_t)(len - 1), fr)) {
            stopped = true;  // does not fit this class (or missing): nothing done
            stop_at = len;
        } else {
            const double w = ok ? exp((double)fr.E * SP_LN2 - logP) * lp.p_end : 0.0;
            for (int j = lane; j < fr.n; j += 64) val[j] = w * (fr.m[j] + fr.i[j] + fr.d[j]);
            wave_sync();
            if (!emit_mapping<CAP>(a.mpool, q0 + (uint64_t)(len - 1), fr.id, val, fr.n, a.ratio_lin, true, order, a.topk))
                err |= SP_ERR_POOL;
        }
    } else {
        pos = a.stop[gi];
        if (pos < len - 1) {
            // B.tables[pos+1] from the hand-off slot
            const BHandoff &h = a.hand[gi];
            Col<CAP> &c = cols[(pos + 1) & 1];
            const int n = h.n;
            hash_clear(c);
            if (lane == 0) {
                c.n = c.na = n;
                c.E = h.E;
            }
            wave_sync();
            for (int j = lane; j < n; j += 64) {
                c.id[j] = h.id[j];
                c.m[j] = h.m[j];
                c.i[j] = h.i[j];
                c.d[j] = h.d[j];
                hash_insert(c, h.id[j], j);
            }
            wave_sync();
            have_cols = 1;
        }
    }
    int steps_done = 0;
    for (; !stopped && pos >= s0 + 1 && !err; pos--) {
        // B.tables[pos] over filled_nodes(F.tables[pos-1]) (backward.rs:122-129)
        if (!load_record<CAP>(a.fpool, p0 + (uint64_t)(pos - 1), fr)) {
            stopped = true;  // the forward record is larger than this class
            stop_at = pos;
            break;
        }
        int nl;
        if (a.list_off) {
            const uint64_t l0 = a.list_off[q0 + (uint64_t)pos];
            nl = (int)(a.list_off[q0 + (uint64_t)pos + 1] - l0);
            if (nl > CAP) {
                err |= SP_ERR_CAPACITY;
                break;
            }
            for (int j = lane; j < nl; j += 64) list[j] = a.list_nodes[l0 + j];
        } else {
            for (int j = lane; j < fr.n; j += 64) val[j] = fr.m[j] + fr.i[j] + fr.d[j];
            wave_sync();
            sort_desc<CAP>(val, fr.n, order);
            wave_sync();
            nl = fr.na < fr.n ? fr.na : fr.n;
            for (int j = lane; j < nl; j += 64) list[j] = fr.id[order[j]];
        }
        wave_sync();
        Col<CAP> &prev = cols[(pos + 1) & 1];
        Col<CAP> &cur = cols[pos & 1];
        bwd_list_step<CAP>(a.M, prev, pos == len - 1, cur, list, nl, a.bases[((size_t)g * a.Lb + pos) * a.W + r], dA, dB);
        have_cols = 1;
        // S = F.tables[pos-1] (.) B.tables[pos] / P over F's elements (table.rs:320-345, 500-505)
        const double w = ok ? exp((double)(fr.E + cur.E) * SP_LN2 - logP) : 0.0;
        for (int j = lane; j < fr.n; j += 64) {
            const int bs = hash_find(cur, fr.id[j]);
            val[j] = bs >= 0 ? w * (fr.m[j] * cur.m[bs] + fr.i[j] * cur.i[bs] + fr.d[j] * cur.d[bs]) : 0.0;
        }
        wave_sync();
        if (!emit_mapping<CAP>(a.mpool, q0 + (uint64_t)(pos - 1), fr.id, val, fr.n, a.ratio_lin, true, order, a.topk))
            err |= SP_ERR_POOL;
        // a burst (max_steps) ends where the column fits the one-lane-per-node class again
        if (!a.list_off && a.max_steps > 0 && !err && ++steps_done >= a.max_steps && cur.n <= 64 && pos - 1 >= s0 + 1) {
            stopped = true;
            stop_at = pos - 1;
            break;
        }
    }
    if (a.list_off) {
        if (stopped) err |= SP_ERR_CAPACITY;  // a forward record is missing or larger than the list class
        if (lane == 0) a.stop[gi] = stopped ? stop_at : s0;
    } else if (stopped && !err) {
        // park B.tables[stop_at + 1] for the next phase
        if (have_cols && stop_at < len) {
            const Col<CAP> &c = cols[(stop_at + 1) & 1];
            BHandoff &h = a.hand[gi];
            if (c.n > HANDOFF_CAP) err |= SP_ERR_CAPACITY;
            else {
                if (lane == 0) {
                    h.n = c.n;
                    h.E = c.E;
                }
                for (int j = lane; j < c.n; j += 64) {
                    h.id[j] = c.id[j];
                    h.m[j] = c.m[j];
                    h.i[j] = c.i[j];
                    h.d[j] = c.d[j];
                }
            }
        }
        if (lane == 0) a.stop[gi] = stop_at;
    } else if (!err) {
        // hand B.tables[s0+1] to the dense backward kernel: dense column (zeros elsewhere = the
        // SparseVec default), its exponent and maximum
        if (have_cols) {
            const Col<CAP> &c = cols[(s0 + 1) & 1];
            const size_t NW = (size_t)a.d.N * a.W;
            const int pc = (s0 + 1) & 1;
            // (the host zeroed the chunk's B buffers before the sparse backward: a lane-strided clear
            // from here would cost a 64-byte sector per 8-byte store)
            double *bm = a.d.Bm + ((size_t)g * a.d.bcols + pc) * NW;
            double *bi = a.d.Bi + ((size_t)g * a.d.bcols + pc) * NW;
            double mx = 0.0;
            for (int j = lane; j < c.n; j += 64) {
                bm[(size_t)c.id[j] * a.W + r] = c.m[j];
                bi[(size_t)c.id[j] * a.W + r] = c.i[j];
                mx = fmax(mx, fmax(c.m[j], c.i[j]));
            }
            mx = wave_max(mx);
            if (lane == 0) {
                a.d.cmaxB[((size_t)g * a.d.Lc + (s0 + 1)) * a.W + r] = (unsigned long long)__double_as_longlong(mx);
                a.d.BE[((size_t)g * (a.d.Lc + 1) + (s0 + 1)) * a.W + r] = c.E;
            }
        }
        if (lane == 0) a.stop[gi] = s0;
    }
    for (int off = 32; off >= 1; off >>= 1) err |= (uint32_t)__shfl_xor((int)err, off);
    if (lane == 0) a.err[gi] = err;
}

// ---------------------------------------------------------------- dense head
struct DenseMapArgs {
    DenseArgs d;
    int W;
    RecPool mpool;
    const uint64_t *lane_pos0;
    int *cntA, *cntB;           // [lanes]
    uint32_t *candA_node, *candB_node;  // [lanes][400]
    double *candA_val, *candB_val;
    double ratio_lin;
    int topk;
    uint32_t *err;
    unsigned long long *eoff;  // [2][lanes] record offsets of the column being emitted (emit_offsets)
    int force_radix;           // tests: take the radix fallback of block_top_from_column
    // work lists of the column being emitted (emit_offsets): items = which * lanes + lane
    uint32_t *wl_small;        // lists of 1..64 candidates under the ratio rule: one wave each (emit_dense_small)
    uint32_t *wl_big;          // the rest: one block each (emit_dense_map)
    int *wl_count;             // [2]
};

template <int W>
__global__ void __launch_bounds__(BLOCK) post_collect(const DenseMapArgs ma, const int pos) {
    const DenseArgs &a = ma.d;
    const int g = blockIdx.y, lb = blockIdx.x;
    constexpr int ROWS = BLOCK / W;
    const int r = threadIdx.x % W, row = threadIdx.x / W;
    const int gi = g * W + r;
    const int len = a.len[gi];
    const int braw = a.bstart[gi];
    const bool sparse_tail = (braw & (1 << 30)) != 0;
    const int bstart = braw & ~(1 << 30);
    const bool live = pos < len && pos <= bstart;
    const bool doA = live && pos >= 1;
    const bool doB = live && pos == len - 1 && !sparse_tail;
    if (lb >= a.nblk) return;
    const size_t NW = (size_t)a.N * W;
    if (a.Prun && doA) {
        // bwd_step left the maximum of each thread's run (npt consecutive nodes): one read per run, and only
        // the runs above the threshold -- a handful per read -- go back to the emit-prob plane
        const double thr = __longlong_as_double((long long)a.pmax[((size_t)g * (a.Lc + 1) + pos) * W + r]) * ma.ratio_lin;
        const double rm = a.Prun[((size_t)g * a.nblk8 + lb) * BLOCK + threadIdx.x];
        if (rm > 0.0 && rm > thr) {
            const double *P = a.Pa + (size_t)g * NW;
            const int k0 = lb * (a.npt * ROWS) + row * a.npt;
            uint32_t *cn = ma.candA_node + (size_t)gi * KMAX;
            double *cv = ma.candA_val + (size_t)gi * KMAX;
            int local = 0;
            for (int j = 0; j < a.npt && k0 + j < a.N; j++) {
                const double t = P[(size_t)(k0 + j) * W + r];
                local += (t > 0.0 && t > thr) ? 1 : 0;
            }
            int slot = atomicAdd(&ma.cntA[gi], local);
            for (int j = 0; j < a.npt && k0 + j < a.N && slot < KMAX; j++) {
                const double t = P[(size_t)(k0 + j) * W + r];
                if (t > 0.0 && t > thr) {
                    cn[slot] = (uint32_t)(k0 + j);
                    cv[slot] = t;
                    slot++;
                }
            }
        }
    }
    const int kbase = lb * (a.npt * ROWS) + row;
    for (int which = (a.Prun ? 1 : 0); which < 2; which++) {
        if (!(which == 0 ? doA : doB)) continue;
        const double *P = (which == 0 ? a.Pa : a.Pb) + (size_t)g * NW;
        const int mi = which == 0 ? pos : len;
        const double thr =
            __longlong_as_double((long long)a.pmax[((size_t)g * (a.Lc + 1) + mi) * W + r]) * ma.ratio_lin;
        int *cnt = which == 0 ? ma.cntA : ma.cntB;
        uint32_t *cn = (which == 0 ? ma.candA_node : ma.candB_node) + (size_t)gi * KMAX;
        double *cv = (which == 0 ? ma.candA_val : ma.candB_val) + (size_t)gi * KMAX;
        int local = 0;
        for (int j = 0; j < a.npt; j++) {
            const int k = kbase + j * ROWS;
            if (k >= a.N) break;
            const double t = P[(size_t)k * W + r];
            local += (t > 0.0 && t > thr) ? 1 : 0;
        }
        if (local == 0) continue;
        int slot = atomicAdd(&cnt[gi], local);
        for (int j = 0; j < a.npt && slot < KMAX; j++) {
            const int k = kbase + j * ROWS;
            if (k >= a.N) break;
            const double t = P[(size_t)k * W + r];
            if (t > 0.0 && t > thr) {
                cn[slot] = (uint32_t)k;
                cv[slot] = t;
                slot++;
            }
        }
    }
}

// More than 400 nodes inside the ratio: keep the 400 best of the column (ties: lowest node id).
// One 256-thread block: 8-bit radix select of the 400th-largest bit pattern (positive doubles
// order like integers), then an ordered collect (everything above it, then ties in node order).
__device__ int block_top_radix(const double *col, int stride, int N, double thr, uint32_t *ids, double *val) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_k, s_wcnt[BLOCK / 64], s_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        s_prefix = 0ull;
        s_k = KMAX;
    }
    __syncthreads();
    for (int shift = 56; shift >= 0; shift -= 8) {
        hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        for (int k = tid; k < N; k += BLOCK) {
            const double v = col[(size_t)k * stride];
            if (!(v > thr)) continue;
            const unsigned long long b = (unsigned long long)__double_as_longlong(v);
            if (shift < 56 && (b >> (shift + 8)) != (prefix >> (shift + 8))) continue;
            atomicAdd(&hist[(b >> shift) & 255ull], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int k = s_k, d = 255;
            for (; d > 0; d--) {
                if ((int)hist[d] >= k) break;
                k -= (int)hist[d];
            }
            s_k = k;
            s_prefix = prefix | ((unsigned long long)d << shift);
        }
        __syncthreads();
    }
    const unsigned long long T = s_prefix;  // bit pattern of the 400th-largest value (or the smallest one)
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int pass = 0; pass < 2; pass++)
        for (int base = 0; base < N; base += BLOCK) {
            const int k = base + tid;
            double v = 0.0;
            bool take = false;
            if (k < N) {
                v = col[(size_t)k * stride];
                const unsigned long long b = (unsigned long long)__double_as_longlong(v);
                take = v > thr && (pass == 0 ? b > T : b == T);
            }
            const unsigned long long mk = __ballot(take);
            if (lane == 0) s_wcnt[wave] = __popcll(mk);
            __syncthreads();
            int p = s_n + __popcll(mk & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; w++) p += s_wcnt[w];
            if (take && p < KMAX) {
                ids[p] = (uint32_t)k;
                val[p] = v;
            }
            __syncthreads();
            if (tid == 0) {
                int t = s_n;
                for (int w = 0; w < BLOCK / 64; w++) t += s_wcnt[w];
                s_n = t;
            }
            __syncthreads();
            if (s_n >= KMAX) break;
        }
    __syncthreads();
    return s_n < KMAX ? s_n : KMAX;
}

// Room for the records of one column in ONE allocation: every (lane, which) that emit_dense_map will write gets
// an offset from an upper bound of its record size (the candidate count; the ratio rule can only shorten the
// list).  One atomic per column instead of one per record: 4 000 blocks adding to the same word cost 2.8 ms a
// column, and the columns in which every read is still dense are the last ones, with nothing left to hide behind.
__global__ void __launch_bounds__(1024) emit_offsets(const DenseMapArgs ma, const int pos, const int lanes) {
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long s_base;
    const DenseArgs &a = ma.d;
    const int tid = threadIdx.x;
    const int items = 2 * lanes;
    const int per = (items + 1023) / 1024;
    auto bytes_of = [&](int it) -> unsigned long long {
        if (it >= items) return 0ull;
        const int which = it / lanes, gi = it % lanes;
        const int len = a.len[gi];
        if (len == 0) return 0ull;
        const int braw = a.bstart[gi];
        const bool sparse_tail = (braw & (1 << 30)) != 0;
        const int bstart = braw & ~(1 << 30);
        const bool live = pos < len && pos <= bstart;
        const bool mine = which == 0 ? (live && pos >= 1) : (live && pos == len - 1 && !sparse_tail);
        if (!mine) return 0ull;
        const int c = (which == 0 ? ma.cntA : ma.cntB)[gi];
        int k = c < KMAX ? c : KMAX;
        if (ma.topk > 0) {
            const int want = ma.topk < a.N ? ma.topk : a.N;
            k = k > want ? k : want;
        }
        return 8ull + (unsigned long long)((k + 1) & ~1) * 4ull + (unsigned long long)k * 8ull;
    };
    unsigned long long loc = 0ull;
    for (int q = 0; q < per; q++) loc += bytes_of(tid * per + q);
    part[tid] = loc;
    __syncthreads();
    // exclusive scan of the 1024 partial sums (Hillis-Steele)
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned long long v = tid >= off ? part[tid - off] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    if (tid == 1023) s_base = atomicAdd(ma.mpool.top, part[1023]);
    __syncthreads();
    unsigned long long o = s_base + part[tid] - loc;
    for (int q = 0; q < per; q++) {
        const int it = tid * per + q;
        if (it < items) ma.eoff[it] = o;
        o += bytes_of(it);
    }
    // The work lists of the column: which (lane, which) pairs have a list to write, and by whom.  A list of up to 64
    // candidates under the ratio rule -- all but the first few columns of a read, 5 entries on average -- is one
    // wave's work (emit_dense_small); longer ones, fixed-size lists and over-full columns take a block.
    auto class_of = [&](int it) -> int {  // 0: nothing, 1: small, 2: big
        if (bytes_of(it) == 0ull) return 0;
        const int c = (it / lanes == 0 ? ma.cntA : ma.cntB)[it % lanes];
        return (ma.topk <= 0 && c >= 1 && c <= 64) ? 1 : 2;
    };
    __syncthreads();
    unsigned long long cl = 0ull;  // low word: small items of this thread, high word: big ones
    for (int q = 0; q < per; q++) {
        const int c = class_of(tid * per + q);
        cl += c == 1 ? 1ull : (c == 2 ? (1ull << 32) : 0ull);
    }
    part[tid] = cl;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned long long v = tid >= off ? part[tid - off] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long at = part[tid] - cl;
    for (int q = 0; q < per; q++) {
        const int it = tid * per + q;
        const int c = class_of(it);
        if (c == 1) ma.wl_small[(uint32_t)at] = (uint32_t)it, at += 1ull;
        else if (c == 2) ma.wl_big[(uint32_t)(at >> 32)] = (uint32_t)it, at += 1ull << 32;
    }
    if (tid == 1023) {
        ma.wl_count[0] = (int)(uint32_t)part[1023];
        ma.wl_count[1] = (int)(uint32_t)(part[1023] >> 32);
    }
}

// A list of up to 64 candidates (node, posterior) of one (lane, which): one wave, one candidate per lane --
// to_mapping_by_score_ratio (hint.rs:135-142, table.rs:134-149): kept = inside the ratio of the best, ordered by the
// value the list holds (the log), equal logs by node id (see emit_mapping).  Four independent waves per block.
__global__ void __launch_bounds__(BLOCK) emit_dense_small(const DenseMapArgs ma, const int pos, const int lanes) {
    const DenseArgs &a = ma.d;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_items = ma.wl_count[0];
    for (int w = blockIdx.x * (BLOCK / 64) + wave; w < n_items; w += gridDim.x * (BLOCK / 64)) {
        const uint32_t it = ma.wl_small[w];
        const int which = (int)(it / (uint32_t)lanes), gi = (int)(it % (uint32_t)lanes);
        int *cnt = which == 0 ? ma.cntA : ma.cntB;
        const int c = cnt[gi];
        const uint32_t *cn = (which == 0 ? ma.candA_node : ma.candB_node) + (size_t)gi * KMAX;
        const double *cv = (which == 0 ? ma.candA_val : ma.candB_val) + (size_t)gi * KMAX;
        const bool has = lane < c;
        const uint32_t id = has ? cn[lane] : 0u;
        const double v = has ? cv[lane] : 0.0;
        const double thr = wave_max(v) * ma.ratio_lin;
        const bool stay = has && v > 0.0 && v > thr;
        const double lv = stay ? log(v) : 0.0;
        const unsigned long long sm = __ballot(stay);
        const int k = __popcll(sm);
        int rank = 0;
        const long long vb = __double_as_longlong(lv);
        for (unsigned long long mm = sm; mm != 0ull; mm &= mm - 1ull) {
            const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)mm) - 1);
            const int ulo = __builtin_amdgcn_readlane((int)(vb & 0xffffffffll), l);
            const int uhi = __builtin_amdgcn_readlane((int)(vb >> 32), l);
            const double u = __longlong_as_double(((long long)uhi << 32) | (long long)(unsigned int)ulo);
            const uint32_t un = (uint32_t)__builtin_amdgcn_readlane((int)id, l);
            rank += (u > lv) || (u == lv && un < id);
        }
        const int mi = which == 0 ? pos : a.len[gi];
        const uint64_t pidx = ma.lane_pos0[gi] + (uint64_t)(mi - 1);
        const uint64_t idb = (uint64_t)((k + 1) & ~1) * 4;
        const uint64_t bytes = 8 + idb + (uint64_t)k * 8;
        const uint64_t o = ma.eoff[it];
        if (o + bytes > ma.mpool.cap) {
            if (lane == 0) atomicOr(&ma.err[gi], SP_ERR_POOL);
        } else {
            uint8_t *rec = ma.mpool.base + o;
            if (lane == 0) {
                ((uint32_t *)rec)[0] = (uint32_t)k;
                ((uint32_t *)rec)[1] = 0;
                ma.mpool.off[pidx] = o + 8;
            }
            if (stay) {
                ((uint32_t *)(rec + 8))[rank] = id;
                ((double *)(rec + 8 + idb))[rank] = lv;
            }
        }
        if (lane == 0) cnt[gi] = 0;
    }
}

// The same selection in three passes over the column instead of ten.  A read whose first bases fit nowhere has
// tens of thousands of nodes inside the ratio at its first positions, the column is strided by the read-group
// width (a 64-byte sector per value), and the bwd_step two columns on waits for this column's plane: at 2.8 ms
// a column two such reads cost cfg3 more than 10 ms per step.  Values inside the ratio span < 2^11 bins of
// bits(v) >> shift: one histogram pass finds the bin of the 400th largest, one pass collects everything above
// it plus the boundary bin (staged in LDS, normally a few dozen values), a rank-by-counting over the boundary
// bin settles the rest (ties: lowest node id, as before).  A boundary bin beyond the staging room falls back to
// the radix select.
static constexpr int TOPH_BINS = 2048, TOPH_STAGE = 1024;
__device__ int block_top_from_column(const double *col, int stride, int N, double thr, double vmax, uint32_t *ids, double *val,
                                     bool force_radix) {
    if (force_radix) return block_top_radix(col, stride, N, thr, ids, val);
    __shared__ unsigned int hist[TOPH_BINS];
    __shared__ uint32_t bid[TOPH_STAGE];
    __shared__ double bval[TOPH_STAGE];
    __shared__ int s_bin, s_need, s_n, s_nb;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long lo = (unsigned long long)__double_as_longlong(thr > 0.0 ? thr : 0.0);
    const unsigned long long hi = (unsigned long long)__double_as_longlong(vmax);
    const unsigned long long range = hi > lo ? hi - lo : 0ull;
    int shift = 0;
    while ((range >> shift) >= (unsigned long long)TOPH_BINS) shift++;
    for (int h = tid; h < TOPH_BINS; h += BLOCK) hist[h] = 0u;
    __syncthreads();
    // (eight strided loads in flight per thread: a thread that waits for each value of its read's column in turn --
    // a 64-byte sector per value -- spends the pass waiting)
    for (int k0 = tid; k0 < N; k0 += 8 * BLOCK) {
        double v8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = k0 + u * BLOCK;
            v8[u] = k < N ? col[(size_t)k * stride] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const double v = v8[u];
            if (!(v > thr)) continue;
            const unsigned long long b = ((unsigned long long)__double_as_longlong(v) - lo) >> shift;
            atomicAdd(&hist[b < (unsigned long long)TOPH_BINS ? (int)b : TOPH_BINS - 1], 1u);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int k = KMAX, d = TOPH_BINS - 1;
        for (; d > 0; d--) {
            if ((int)hist[d] >= k) break;
            k -= (int)hist[d];
        }
        s_bin = d;    // bin of the KMAX-th largest value (or 0: fewer than KMAX qualify above it)
        s_need = k;   // how many of that bin are wanted
        s_n = 0;
        s_nb = 0;
    }
    __syncthreads();
    const int bstar = s_bin;
    // (slots are handed out per wave, in no particular order: the list is sorted by (value, node id) afterwards and
    // everything collected here stays -- no barrier inside the pass over the column)
    for (int base0 = 0; base0 < N; base0 += 8 * BLOCK) {
        double v8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = base0 + u * BLOCK + tid;
            v8[u] = k < N ? col[(size_t)k * stride] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = base0 + u * BLOCK + tid;
            const double v = v8[u];
            bool take = false, edge = false;
            if (k < N && v > thr) {
                unsigned long long b = ((unsigned long long)__double_as_longlong(v) - lo) >> shift;
                if (b >= (unsigned long long)TOPH_BINS) b = TOPH_BINS - 1;
                take = (int)b > bstar;
                edge = (int)b == bstar;
            }
            const unsigned long long mk = __ballot(take), me = __ballot(edge);
            if (mk == 0ull && me == 0ull) continue;
            int wb = 0, wbb = 0;
            if (lane == 0) {
                if (mk) wb = atomicAdd(&s_n, __popcll(mk));
                if (me) wbb = atomicAdd(&s_nb, __popcll(me));
            }
            wb = __shfl(wb, 0);
            wbb = __shfl(wbb, 0);
            const int p = wb + __popcll(mk & ((1ull << lane) - 1ull)), pb = wbb + __popcll(me & ((1ull << lane) - 1ull));
            if (take && p < KMAX) {
                ids[p] = (uint32_t)k;
                val[p] = v;
            }
            if (edge && pb < TOPH_STAGE) {
                bid[pb] = (uint32_t)k;
                bval[pb] = v;
            }
        }
    }
    __syncthreads();
    const int nabove = s_n, nb = s_nb, need = s_need;
    if (nb > TOPH_STAGE) return block_top_radix(col, stride, N, thr, ids, val);  // (uniform: every thread sees s_nb)
    // the `need` largest of the boundary bin, ties by node id
    __syncthreads();
    for (int j = tid; j < nb; j += BLOCK) {
        const double v = bval[j];
        const uint32_t id = bid[j];
        int rank = 0;
        for (int q = 0; q < nb; q++) {
            const double u = bval[q];
            rank += (u > v) || (u == v && bid[q] < id);
        }
        if (rank < need && nabove + rank < KMAX) {
            ids[nabove + rank] = id;
            val[nabove + rank] = v;
        }
    }
    __syncthreads();
    const int tot = nabove + (nb < need ? nb : need);
    return tot < KMAX ? tot : KMAX;
}

// one wave per (lane, which): sort the collected nodes and write the mapping record
__global__ void __launch_bounds__(BLOCK) emit_dense_map(const DenseMapArgs ma, const int pos, const int lanes) {
    __shared__ uint32_t ids[KMAX];
    __shared__ double val[KMAX];
    __shared__ uint16_t order[KMAX];
    const DenseArgs &a = ma.d;
    if ((int)blockIdx.x >= ma.wl_count[1]) return;  // (the grid is the host's upper bound of the list)
    const uint32_t item = ma.wl_big[blockIdx.x];
    const int gi = (int)(item % (uint32_t)lanes), which = (int)(item / (uint32_t)lanes);
    const int len = a.len[gi];
    if (len == 0) return;
    const int braw = a.bstart[gi];
    const bool sparse_tail = (braw & (1 << 30)) != 0;
    const int bstart = braw & ~(1 << 30);
    const bool live = pos < len && pos <= bstart;
    const bool mine = which == 0 ? (live && pos >= 1) : (live && pos == len - 1 && !sparse_tail);
    if (!mine) return;
    int *cnt = which == 0 ? ma.cntA : ma.cntB;
    const int c = cnt[gi];
    const int g = gi / ma.W, r = gi % ma.W;
    const int mi = which == 0 ? pos : len;
    int n = c < KMAX ? c : KMAX;
    if (c > KMAX) {
        const double *P = (which == 0 ? a.Pa : a.Pb) + (size_t)g * a.N * ma.W + r;
        const double vmax = __longlong_as_double((long long)a.pmax[((size_t)g * (a.Lc + 1) + mi) * ma.W + r]);
        n = block_top_from_column(P, ma.W, a.N, vmax * ma.ratio_lin, vmax, ids, val, ma.force_radix != 0);
    } else {
        const uint32_t *cn = (which == 0 ? ma.candA_node : ma.candB_node) + (size_t)gi * KMAX;
        const double *cv = (which == 0 ? ma.candA_val : ma.candB_val) + (size_t)gi * KMAX;
        for (int j = threadIdx.x; j < n; j += BLOCK) {
            ids[j] = cn[j];
            val[j] = cv[j];
        }
    }
    __syncthreads();
    if (ma.topk > 0 && n < ma.topk && n < a.N && threadIdx.x == 0) {
        // to_mapping(k) of a DENSE column always returns k nodes (hint.rs:124-131): fewer than k have a
        // non-zero probability here, the rest are zero-probability nodes (which ones is unpinned: lowest ids)
        const int want = ma.topk < a.N ? ma.topk : a.N;
        const int n0 = n;
        for (uint32_t k = 0; n < want && k < (uint32_t)a.N; k++) {
            bool present = false;
            for (int j = 0; j < n0; j++) present |= ids[j] == k;
            if (!present) {
                ids[n] = k;
                val[n] = 0.0;
                n++;
            }
        }
        cnt[gi] = -n;  // hand the new length to the other threads
    }
    __syncthreads();
    if (cnt[gi] < 0) n = -cnt[gi];
    __syncthreads();
    if (threadIdx.x >= 64) return;  // the sort + record write is a single-wave job
    const uint64_t pidx = ma.lane_pos0[gi] + (uint64_t)(mi - 1);
    if (!emit_mapping<KMAX>(ma.mpool, pidx, ids, val, n, ma.ratio_lin, true, order, ma.topk,
                            (long long)ma.eoff[item])) {
        if (threadIdx.x == 0) atomicOr(&ma.err[gi], SP_ERR_POOL);
    }
    if (threadIdx.x == 0) cnt[gi] = 0;
}

__global__ void __launch_bounds__(BLOCK) merge_logp(const uint32_t *lanes, int n, const double *src, double *dst) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j < n) dst[lanes[j]] = src[lanes[j]];
}

namespace {
template <int W> void launch_post_collect(const DenseMapArgs &ma, int pos) {
    hipLaunchKernelGGL(post_collect<W>, dim3(ma.d.nblk, ma.d.ng), dim3(BLOCK), 0, current_stream(), ma, pos);
}
void launch_post_collect_w(int W, const DenseMapArgs &ma, int pos) {
    switch (W) {
    case 1: launch_post_collect<1>(ma, pos); break;
    case 2: launch_post_collect<2>(ma, pos); break;
    case 4: launch_post_collect<4>(ma, pos); break;
    case 8: launch_post_collect<8>(ma, pos); break;
    case 16: launch_post_collect<16>(ma, pos); break;
    case 32: launch_post_collect<32>(ma, pos); break;
    case 64: launch_post_collect<64>(ma, pos); break;
    default: PHMM_THROW(PHMM_EINTERNAL, "bad read-group width");
    }
}
}  // namespace

void mapping_backward_chunk(MapChunk &mc, const std::vector<uint32_t> &sparse_lanes, MappingSink *sink,
                            const Plan &plan, int g0, uint64_t R) {
    hipStream_t s = current_stream();
    CallStats &st = stats();
    phmm_model *m = mc.m;
    const int W = mc.W, lanes = mc.lanes;
    const std::vector<int> &hl = *mc.hl, &hsw = *mc.hsw;
    const std::vector<uint64_t> &lp0 = *mc.lane_pos0;
    const size_t NW = (size_t)m->N * W;

    // per-lane control: last dense backward column, global position base of the lane's read
    std::vector<int> hb(lanes, 0);
    std::vector<uint64_t> gp0(lanes + 1, 0);
    int pos_max = -1;
    for (int gi = 0; gi < lanes; gi++) {
        const size_t slot = (size_t)g0 * W + gi;
        if (hl[gi] == 0 || slot >= R) {
            hb[gi] = -1;
            continue;
        }
        gp0[gi] = sink->reads->off[plan.order[slot]];
        if (hsw[gi] < hl[gi]) hb[gi] = hsw[gi] | (1 << 30);
        else hb[gi] = hl[gi] - 1;
        pos_max = std::max(pos_max, hb[gi] & ~(1 << 30));
    }
    DevBuf &ctl = m->wset().aux[3], &pbuf = m->wset().aux[4];
    size_t cb = 0;
    auto carve = [&](size_t bytes) {
        cb = (cb + 255) / 256 * 256;
        size_t o = cb;
        cb += bytes;
        return o;
    };
    const size_t o_bs = carve(sizeof(int) * lanes), o_ca = carve(sizeof(int) * lanes * 2), o_cb = carve(sizeof(int) * lanes * 2),
                 o_err = carve(sizeof(uint32_t) * lanes), o_lanes = carve(sizeof(uint32_t) * std::max<size_t>(sparse_lanes.size(), 1)),
                 o_lp0 = carve(sizeof(uint64_t) * (lanes + 1)), o_gp0 = carve(sizeof(uint64_t) * (lanes + 1)),
                 o_stop = carve(sizeof(int) * lanes),
                 o_bn = carve(sizeof(uint32_t) * (size_t)lanes * KMAX * 2), o_bv = carve(sizeof(double) * (size_t)lanes * KMAX * 2),
                 o_an = carve(sizeof(uint32_t) * (size_t)lanes * KMAX * 2), o_av = carve(sizeof(double) * (size_t)lanes * KMAX * 2),
                 o_hand = carve(sizeof(BHandoff) * (size_t)lanes), o_eoff = carve(sizeof(unsigned long long) * 2 * (size_t)lanes),
                 o_wls = carve(sizeof(uint32_t) * 2 * (size_t)lanes), o_wlb = carve(sizeof(uint32_t) * 2 * (size_t)lanes),
                 o_wlc = carve(sizeof(int) * 2);
    ctl.reserve(cb);
    char *cp = (char *)ctl.p;
    // Pa (two buffers, by position parity), the per-run maxima of Pa, and Pb -- the plane of merged index `len`, which
    // only a read that ends inside the dense columns writes (none on a HiFi read set: no third plane then)
    const size_t prun_n = (size_t)mc.ngc * mc.a.nblk8 * BLOCK;
    bool need_pb = false;
    for (int gi = 0; gi < lanes; gi++) need_pb |= hb[gi] >= 0 && !(hb[gi] & (1 << 30));
    pbuf.reserve(((need_pb ? 3 : 2) * (size_t)mc.ngc * NW + prun_n) * sizeof(double));
    unsigned long long top_before = 0;
    HIP_CHECK(hipMemcpyAsync(&top_before, sink->mp.top, sizeof(top_before), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));

    for (int attempt = 0;; attempt++) {
        HIP_CHECK(hipMemsetAsync(cp, 0, o_bn, s));
        HIP_CHECK(hipMemcpyAsync(cp + o_bs, hb.data(), sizeof(int) * lanes, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(cp + o_lp0, lp0.data(), sizeof(uint64_t) * (lanes + 1), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(cp + o_gp0, gp0.data(), sizeof(uint64_t) * (lanes + 1), hipMemcpyHostToDevice, s));
        if (!sparse_lanes.empty())
            HIP_CHECK(hipMemcpyAsync(cp + o_lanes, sparse_lanes.data(), sizeof(uint32_t) * sparse_lanes.size(),
                                     hipMemcpyHostToDevice, s));
        const RecPool mp = sink->mp;

        DenseArgs a = mc.a;
        a.want_freq = 0;
        a.want_map = 1;
        a.bstart = (const int *)(cp + o_bs);
        a.Pa = pbuf.as<double>();
        a.Pb = need_pb ? pbuf.as<double>() + 2 * (size_t)mc.ngc * NW : pbuf.as<double>();  // (never touched without such a read)
        a.Prun = knobs().no_runmax ? nullptr : pbuf.as<double>() + (need_pb ? 3 : 2) * (size_t)mc.ngc * NW;
        // backward scratch of the chunk must start clean (a previous attempt may have used it)
        HIP_CHECK(hipMemsetAsync(a.cmaxB, 0, sizeof(unsigned long long) * (size_t)a.ng * a.Lc * W, s));
        HIP_CHECK(hipMemsetAsync(a.pmax, 0, sizeof(unsigned long long) * (size_t)a.ng * (a.Lc + 1) * W, s));
        HIP_CHECK(hipMemsetAsync(a.BE, 0, sizeof(int) * (size_t)a.ng * (a.Lc + 1) * W, s));
        // the sparse backward hands B.tables[s0+1] over as a dense column: zeros except its own nodes
        HIP_CHECK(hipMemsetAsync(a.Bm, 0, sizeof(double) * (size_t)a.ng * a.bcols * NW, s));
        HIP_CHECK(hipMemsetAsync(a.Bi, 0, sizeof(double) * (size_t)a.ng * a.bcols * NW, s));

        if (!sparse_lanes.empty()) {
            // ln P of the sparse reads joins the dense ones (posterior weights)
            hipLaunchKernelGGL(merge_logp, dim3((unsigned)((sparse_lanes.size() + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s,
                               (const uint32_t *)(cp + o_lanes), (int)sparse_lanes.size(), mc.d_logp_sparse, a.logPf);
            SparseBwdArgs ba{};
            ba.M = mc.fa_M;
            ba.d = a;
            ba.W = W;
            ba.Lb = mc.Lfull;
            ba.sw = mc.d_sw;
            ba.bases = mc.d_bases_full;
            ba.fpool = mc.fpool;
            ba.mpool = mp;
            ba.lane_pos0 = (const uint64_t *)(cp + o_lp0);
            ba.map_pos0 = (const uint64_t *)(cp + o_gp0);
            ba.lanes = (const uint32_t *)(cp + o_lanes);
            ba.ratio_lin = mc.ratio_lin;
            ba.topk = mc.topk;
            ba.err = (uint32_t *)(cp + o_err);
            ba.stop = (int *)(cp + o_stop);
            ba.hand = (BHandoff *)(cp + o_hand);
            // <64> kernels (one lane per node): the tail of every read; the 400-slot kernel: the positions next to the
            // switch (and the hand-over to the dense kernel) and wherever a forward record outgrew the <64> class.
            // Long reads walk in slices, as in the forward pass (sparse_dyn.hip): the other plans' 400-slot kernels
            // wait for a running one-wave-per-read kernel of this plan, and a read that met one wide spot returns to
            // the <64> kernel after a burst instead of crawling to its first base in the 400-slot one.
            std::vector<uint32_t> todo;
            std::vector<int> hstop(lanes);
            std::vector<uint32_t> herr2(lanes);
            const bool lean_ok =
                mc.topk == 0 && m->dev.max_degree <= (uint32_t)ADJ_DEG && !knobs().no_lean;
            const bool wide_ok = lean_ok && !knobs().no_wide_class;
            const int slice = mc.Lfull > 6144 ? 4096 : 0;  // (a relaunch costs ~0.8 ms: few, long slices)
            std::vector<uint32_t> cont = sparse_lanes, big, fresh;
            bool first = true, any_err = false;
            int bursts = 0;
            auto launch = [&](int kind, int mode, int steps, const std::vector<uint32_t> &who) {
                HIP_CHECK(hipMemcpyAsync(cp + o_lanes, who.data(), sizeof(uint32_t) * who.size(), hipMemcpyHostToDevice, s));
                ba.mode = mode;
                ba.max_steps = steps;
                if (kind == 0 && lean_ok)
                    hipLaunchKernelGGL(lean_backward_kernel, dim3((unsigned)who.size()), dim3(64), 0, s, ba);
                else if (kind == 0)
                    hipLaunchKernelGGL((sparse_backward_kernel<64>), dim3((unsigned)who.size()), dim3(64), 0, s, ba);
                else if (wide_ok)  // the 400-slot class on a block of 448 threads (wide_bwd_kernel.h)
                    hipLaunchKernelGGL(wide_backward_kernel, dim3((unsigned)who.size()), dim3(WBK_T), 0, s, ba);
                else
                    hipLaunchKernelGGL((sparse_backward_kernel<KMAX>), dim3((unsigned)who.size()), dim3(64), 0, s, ba);
                HIP_CHECK(hipGetLastError());
                st.launches[3]++;
                HIP_CHECK(hipMemcpyAsync(hstop.data(), ba.stop, sizeof(int) * lanes, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipMemcpyAsync(herr2.data(), ba.err, sizeof(uint32_t) * lanes, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
            };
            for (int turn = 0; turn < 100000 && !any_err && (!cont.empty() || !big.empty() || !fresh.empty()); turn++) {
                std::vector<uint32_t> ncont;
                if (!cont.empty()) {
                    launch(0, first ? 0 : 1, lean_ok ? slice : 0, cont);
                    for (uint32_t gi : cont) {
                        if (herr2[gi] & ~SP_STOP_SLICE) any_err = true;
                        else if (hstop[gi] == hsw[gi]) continue;
                        else if (herr2[gi] & SP_STOP_SLICE) ncont.push_back(gi);
                        // (a read whose LAST record did not fit <64> has done nothing: it starts in the 400-slot kernel)
                        else if (first && hstop[gi] == hl[gi]) fresh.push_back(gi);
                        else big.push_back(gi);
                    }
                    first = false;
                }
                if (any_err) break;  // reported below (pool growth / internal error)
                const int burst = slice > 0 ? (64 << std::min(bursts, 5)) : 0;
                for (int pass = 0; pass < 2 && !any_err; pass++) {
                    std::vector<uint32_t> &who = pass == 0 ? fresh : big;
                    if (who.empty()) continue;
                    launch(1, pass == 0 ? 0 : 1, burst, who);
                    for (uint32_t gi : who) {
                        if (herr2[gi] & ~SP_STOP_SLICE) any_err = true;
                        else if (hstop[gi] != hsw[gi]) ncont.push_back(gi);
                    }
                    who.clear();
                    bursts++;
                }
                cont.swap(ncont);
            }
            todo = cont;
            todo.insert(todo.end(), big.begin(), big.end());
            todo.insert(todo.end(), fresh.begin(), fresh.end());
            if (!todo.empty()) {
                bool perr = false;
                HIP_CHECK(hipMemcpy(herr2.data(), ba.err, sizeof(uint32_t) * lanes, hipMemcpyDeviceToHost));
                for (uint32_t gi : sparse_lanes) perr |= (herr2[gi] & ~SP_STOP_SLICE) != 0;
                if (!perr) PHMM_THROW(PHMM_EINTERNAL, "sparse backward did not finish");
            }
        }
        trace("  sparse backward");
        DenseMapArgs ma{};
        ma.d = a;
        ma.W = W;
        ma.mpool = mp;
        ma.lane_pos0 = (const uint64_t *)(cp + o_gp0);
        // (counters and candidate lists exist twice, by position parity: emit_dense_map of a column runs while
        // post_collect fills the next one)
        ma.ratio_lin = mc.ratio_lin;
        ma.topk = mc.topk;
        ma.err = (uint32_t *)(cp + o_err);
        ma.eoff = (unsigned long long *)(cp + o_eoff);
        ma.wl_small = (uint32_t *)(cp + o_wls);
        ma.wl_big = (uint32_t *)(cp + o_wlb);
        ma.wl_count = (int *)(cp + o_wlc);
        // lanes still in their dense columns at a position (an upper bound of the lists a column has to write)
        std::vector<int> live_at((size_t)pos_max + 2, 0);
        for (int gi = 0; gi < lanes; gi++)
            if (hb[gi] >= 0) live_at[(size_t)(hb[gi] & ~(1 << 30))]++;
        for (int p = pos_max - 1; p >= 0; p--) live_at[(size_t)p] += live_at[(size_t)p + 1];
        ma.force_radix = knobs().force_radix ? 1 : 0;
        const bool st_on = W == 64 && mc.main_plan;  // statistics of the main plan's bwd_step<64> only (bench.py's roofline)
        LaunchTimer lt(timing_enabled() && st_on);
        std::unique_lock<std::mutex> dense_lock;
        if (mc.dense_token) dense_lock = std::unique_lock<std::mutex>(*mc.dense_token);
        // The sort + record write of a column's lists (emit_dense_map: latency-bound, and a radix select over
        // the column where more than 400 nodes are inside the ratio) runs on a side stream under the next
        // columns' bwd_step; the emit-prob plane is double-buffered by position parity for that.
        const int wi = workset_index();
        if (!m->pool->cstream[wi]) {
            // Lowest priority: the list kernels fill the gaps the HBM-bound bwd_step leaves.  Their launches then stay
            // resident for as long as the bwd_step beside them (1.1 ms average: 60 ms of kernel DURATION per cfg3 step
            // for 0.1 s of single-wave work, SQ_BUSY_CYCLES in profiles/) -- at the highest priority
            // (PHMM_EMIT_HIGH_PRIORITY=1) they are gone in 0.1-0.5 ms each, 30 ms of duration per step, and bwd_step
            // pays for it: 5.85 -> 6.3 ms per launch, 221 -> 229 ms per step (measured back to back on one box).
            int least = 0, greatest = 0;
            HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIP_CHECK(hipStreamCreateWithPriority(&m->pool->cstream[wi], hipStreamNonBlocking, knobs().emit_high_priority ? greatest : least));
        }
        for (auto &e : m->pool->cevent[wi])
            if (!e) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        // (PHMM_SERIAL_EMIT: the list kernels on the main stream, for profiling them alone)
        hipStream_t s2 = knobs().serial_emit ? s : m->pool->cstream[wi];
        hipEvent_t *ev_col = &m->pool->cevent[wi][0], *ev_emit = &m->pool->cevent[wi][2];
        bool emitted[2] = {false, false};
        for (int pos = pos_max; pos >= 0; pos--) {
            const int par = pos & 1;
            a.Pa = pbuf.as<double>() + (size_t)par * mc.ngc * NW;
            ma.d = a;
            ma.cntA = (int *)(cp + o_ca) + (size_t)par * lanes;
            ma.cntB = (int *)(cp + o_cb) + (size_t)par * lanes;
            ma.candA_node = (uint32_t *)(cp + o_an) + (size_t)par * lanes * KMAX;
            ma.candA_val = (double *)(cp + o_av) + (size_t)par * lanes * KMAX;
            ma.candB_node = (uint32_t *)(cp + o_bn) + (size_t)par * lanes * KMAX;
            ma.candB_val = (double *)(cp + o_bv) + (size_t)par * lanes * KMAX;
            if (emitted[par]) HIP_CHECK(hipStreamWaitEvent(s, ev_emit[par], 0));  // plane `par` is free again
            lt.begin();
            launch_bwd_step(W, a, pos);
            lt.end();
            launch_post_collect_w(W, ma, pos);
            HIP_CHECK(hipEventRecord(ev_col[par], s));
            HIP_CHECK(hipStreamWaitEvent(s2, ev_col[par], 0));
            hipLaunchKernelGGL(emit_offsets, dim3(1), dim3(1024), 0, s2, ma, pos, lanes);
            {
                const int ub = std::max(1, live_at[(size_t)pos] * (need_pb ? 2 : 1));
                hipLaunchKernelGGL(emit_dense_small, dim3((unsigned)std::min(1024, (ub + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0,
                                   s2, ma, pos, lanes);
                hipLaunchKernelGGL(emit_dense_map, dim3((unsigned)ub), dim3(BLOCK), 0, s2, ma, pos, lanes);
            }
            HIP_CHECK(hipEventRecord(ev_emit[par], s2));
            emitted[par] = true;
            if (st_on) st.launches[1]++;
        }
        for (int par = 0; par < 2; par++)
            if (emitted[par]) HIP_CHECK(hipStreamWaitEvent(s, ev_emit[par], 0));
        HIP_CHECK(hipGetLastError());
        st.ms[1] += lt.total_ms();
        for (int gi = 0; gi < lanes; gi++)
            if (st_on && hb[gi] >= 0) st.cells[1] += (uint64_t)((hb[gi] & ~(1 << 30)) + 1) * m->N;
        std::vector<uint32_t> herr(lanes);
        HIP_CHECK(hipMemcpyAsync(herr.data(), cp + o_err, sizeof(uint32_t) * lanes, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (dense_lock.owns_lock()) dense_lock.unlock();
        trace("  dense backward+collect");
        bool pool_full = false;
        for (int gi = 0; gi < lanes; gi++) {
            if (herr[gi] & SP_ERR_POOL) pool_full = true;
            else if (herr[gi] & ~SP_STOP_SLICE) PHMM_THROW(PHMM_EINTERNAL, "mapping backward error " + std::to_string(herr[gi]));
        }
        if (!pool_full) break;
        // the pool is shared by every chunk in flight: the whole call restarts with a bigger one
        (void)attempt;
        throw SinkOverflow{};
    }
}

// ---------------------------------------------------------------- final CSR on the device
__global__ void __launch_bounds__(BLOCK) map_counts(RecPool mp, uint64_t n_pos, uint64_t *cnt) {
    const uint64_t p = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p > n_pos) return;
    uint64_t c = 0;
    if (p < n_pos) {
        const uint64_t o1 = mp.off[p];
        if (o1) c = ((const uint32_t *)(mp.base + (o1 - 8)))[0];
    }
    cnt[p] = c;
}
__global__ void __launch_bounds__(BLOCK) map_compact(RecPool mp, uint64_t n_pos, const uint64_t *pos_off, uint32_t *nodes,
                                                     double *logp) {
    const uint64_t p = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= n_pos) return;
    const uint64_t o1 = mp.off[p];
    if (!o1) return;
    const uint8_t *rec = mp.base + (o1 - 8);
    const uint32_t n = ((const uint32_t *)rec)[0];
    const uint64_t idb = (uint64_t)((n + 1) & ~1u) * 4;
    const uint32_t *ids = (const uint32_t *)(rec + 8);
    const double *lps = (const double *)(rec + 8 + idb);
    const uint64_t w = pos_off[p];
    if (((const uint32_t *)rec)[1] == 0u) {
        for (uint32_t j = 0; j < n; j++) {
            nodes[w + j] = ids[j];
            logp[w + j] = lps[j];
        }
        return;
    }
    // flag 1: a list of the one-lane-per-node backward kernel, still in lane order (lean_bwd_kernel.h).  Its order:
    // descending by the value it holds, equal values by node id (hint.rs:135-142; ties: DESIGN.md section 2)
    for (uint32_t j = 0; j < n; j++) {
        const double v = lps[j];
        const uint32_t id = ids[j];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < n; q++) {
            const double u = lps[q];
            rank += (u > v) || (u == v && ids[q] < id);
        }
        nodes[w + rank] = id;
        logp[w + rank] = v;
    }
}
// longest list of every read: one wave per read (a thread per read walked 10 000 positions alone: 4.8 ms on 10 kb reads)
__global__ void __launch_bounds__(BLOCK) map_read_max(const uint64_t *read_off, uint64_t R, const uint64_t *pos_off,
                                                      uint32_t *out) {
    const uint64_t r = (uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    uint32_t mx = 0;
    for (uint64_t p = read_off[r] + (uint64_t)lane; p < read_off[r + 1]; p += 64) mx = max(mx, (uint32_t)(pos_off[p + 1] - pos_off[p]));
    mx = wave_umax(mx);
    if (lane == 0) out[r] = mx;
}
__global__ void __launch_bounds__(BLOCK) map_probs(const double *logp, uint64_t n, double *prob) {
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j < n) prob[j] = exp(logp[j]);
}
// Mappings::to_node_freqs (hint.rs:161-171) in a fixed order: entries are sorted by node (stable radix sort); ONE WAVE
// per node sums its segment -- lane l the entries l, l + 64, ... in order, then the lanes in a fixed tree.  The order
// depends on the segment alone, so the sums are the same whatever the read grouping.  (One thread per node walked
// its segment alone: 57 000 entries per node on a short-unit tandem repeat, 0.28 s per call on `rep20`.)
static constexpr int NODE_FREQ_WAVES = 4;
__global__ void __launch_bounds__(64 * NODE_FREQ_WAVES) map_node_freq(const uint32_t *sorted_nodes, const double *sorted_prob,
                                                                        uint64_t n, uint32_t N, double *freq) {
    const uint32_t v = blockIdx.x * NODE_FREQ_WAVES + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (v >= N) return;  // (a whole wave)
    uint64_t lo = 0, hi = n;
    while (lo < hi) {  // first index with node >= v
        const uint64_t mid = (lo + hi) >> 1;
        if (sorted_nodes[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    uint64_t e0 = lo, e1 = n;
    while (e0 < e1) {  // first index with node > v
        const uint64_t mid = (e0 + e1) >> 1;
        if (sorted_nodes[mid] <= v) e0 = mid + 1;
        else e1 = mid;
    }
    double s = 0.0;
    for (uint64_t j = lo + (uint64_t)lane; j < e0; j += 64) s += sorted_prob[j];
    s = wave_sum(s);
    if (lane == 0) freq[v] = s;
}

static void init_sink(phmm_model *m, const phmm_reads *reads, MappingSink &sink, int topk = 0) {
    hipStream_t s = current_stream();
    const uint64_t n_pos = reads->total;
    sink.reads = reads;
    sink.total_pos = n_pos;
    // (+ 2 x 64 KB per read: the frontier kernels claim the pool in 32 KB slabs per wave)
    // ratio lists hold ~5 entries per position; fixed lists exactly topk
    const uint64_t per_pos = topk > 0 ? 16 + 12 * (uint64_t)topk : 160;
    sink.cap = std::max<uint64_t>(m->wset().aux[5].bytes, n_pos * per_pos + reads->R * 131072 + (1u << 20));
    m->wset().aux[5].reserve(sink.cap);
    m->wset().aux[6].reserve(sizeof(unsigned long long) + sizeof(uint64_t) * (n_pos + 1));
    HIP_CHECK(hipMemsetAsync(m->wset().aux[6].p, 0, sizeof(unsigned long long) + sizeof(uint64_t) * (n_pos + 1), s));
    sink.mp.base = m->wset().aux[5].as<uint8_t>();
    sink.mp.cap = sink.cap;
    sink.mp.top = m->wset().aux[6].as<unsigned long long>();
    sink.mp.off = (uint64_t *)(m->wset().aux[6].as<char>() + 8);

}

static void finish_mappings(phmm_model *m, const phmm_reads *reads, MappingSink &sink, const std::vector<double> &lf,
                            phmm_mappings **out, double *out_node_freq) {
    hipStream_t s = current_stream();
    const uint64_t n_pos = reads->total;
    std::unique_ptr<phmm_mappings> mp(new phmm_mappings());
    mp->R = reads->R;
    mp->total_pos = n_pos;
    mp->read_off = reads->off;
    mp->read_logp = lf;
    mp->host_valid = false;
    mp->trusted = true;
    (void)hipGetDevice(&mp->device);
    DevicePool &dpool = device_pool();
    // counts -> exclusive scan -> compaction, all on the device
    DevBuf &cnt = m->wset().aux[13], &tmp = m->wset().aux[14];
    cnt.reserve(sizeof(uint64_t) * (n_pos + 1));
    dpool.take(mp->d_pos_off, sizeof(uint64_t) * (n_pos + 1));
    const unsigned nb = (unsigned)((n_pos + 1 + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(map_counts, dim3(nb), dim3(BLOCK), 0, s, sink.mp, n_pos, cnt.as<uint64_t>());
    size_t tb = 0;
    HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cnt.as<uint64_t>(), mp->d_pos_off.as<uint64_t>(), (int)(n_pos + 1), s));
    tmp.reserve(tb);
    HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, cnt.as<uint64_t>(), mp->d_pos_off.as<uint64_t>(), (int)(n_pos + 1), s));
    uint64_t total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, mp->d_pos_off.as<uint64_t>() + n_pos, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    mp->total_entries = total;
    dpool.take(mp->d_nodes, sizeof(uint32_t) * std::max<uint64_t>(total, 1));
    dpool.take(mp->d_logp, sizeof(double) * std::max<uint64_t>(total, 1));
    hipLaunchKernelGGL(map_compact, dim3(nb), dim3(BLOCK), 0, s, sink.mp, n_pos, mp->d_pos_off.as<uint64_t>(),
                       mp->d_nodes.as<uint32_t>(), mp->d_logp.as<double>());
    // longest list per read (capacity class of the hinted kernel)
    // (scratch of this function: workspace buffers, not a hipMalloc / hipFree pair per call)
    DevBuf &d_roff = m->wset().aux[15], &d_rmax = m->wset().aux[16];
    d_roff.upload(reads->off.data(), sizeof(uint64_t) * (reads->R + 1));
    d_rmax.reserve(sizeof(uint32_t) * reads->R);
    hipLaunchKernelGGL(map_read_max, dim3((unsigned)((reads->R + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0, s,
                       d_roff.as<uint64_t>(), reads->R, mp->d_pos_off.as<uint64_t>(), d_rmax.as<uint32_t>());
    mp->read_max_list.resize(reads->R);
    HIP_CHECK(hipMemcpyAsync(mp->read_max_list.data(), d_rmax.p, sizeof(uint32_t) * reads->R, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    mp->on_device = true;
    trace("device CSR");
    if (out_node_freq) {
        DevBuf &d_prob = m->wset().aux[17], &d_sn = m->wset().aux[18], &d_sp = m->wset().aux[19], &d_freq = m->wset().aux[20];
        const uint64_t n = std::max<uint64_t>(total, 1);
        d_prob.reserve(sizeof(double) * n);
        d_sn.reserve(sizeof(uint32_t) * n);
        d_sp.reserve(sizeof(double) * n);
        d_freq.reserve(sizeof(double) * m->N);
        if (total) {
            hipLaunchKernelGGL(map_probs, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s,
                               mp->d_logp.as<double>(), total, d_prob.as<double>());
            int end_bit = 1;
            while (end_bit < 32 && (1ull << end_bit) < m->N) end_bit++;
            size_t sb = 0;
            HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, sb, mp->d_nodes.as<uint32_t>(), d_sn.as<uint32_t>(),
                                                         d_prob.as<double>(), d_sp.as<double>(), (int)total, 0, end_bit, s));
            tmp.reserve(sb);
            HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, sb, mp->d_nodes.as<uint32_t>(), d_sn.as<uint32_t>(),
                                                         d_prob.as<double>(), d_sp.as<double>(), (int)total, 0, end_bit, s));
        }
        hipLaunchKernelGGL(map_node_freq, dim3((m->N + NODE_FREQ_WAVES - 1) / NODE_FREQ_WAVES), dim3(64 * NODE_FREQ_WAVES), 0, s, d_sn.as<uint32_t>(),
                           d_sp.as<double>(), total, m->N, d_freq.as<double>());
        HIP_CHECK(hipGetLastError());
        copy_out(out_node_freq, d_freq.p, sizeof(double) * m->N);
        trace("node freqs");
    }
    *out = mp.release();
}

// PHMMModel::generate_mappings(reads, None, use_max_ratio = true)
void generate_mappings_sparse(phmm_model *m, const phmm_reads *reads, phmm_mappings **out, double *out_node_freq,
                              bool by_ratio) {
    MappingSink sink{};
    std::vector<double> lf(reads->R);
    double tot = 0.0;
    for (int attempt = 0;; attempt++) {
        init_sink(m, reads, sink, by_ratio ? 0 : (int)m->params.n_active_nodes);
        try {
            full_prob_reads_sparse(m, reads, lf.data(), &tot, &sink, by_ratio);
            break;
        } catch (const SinkOverflow &) {
            if (attempt >= 4) PHMM_THROW(PHMM_ENOMEM, "mapping pool keeps overflowing");
            HIP_CHECK(hipDeviceSynchronize());
            m->wset().aux[5].reserve(sink.cap * 2);  // init_sink sizes the pool from the buffer
        }
    }
    trace("forward+backward chunks");
    finish_mappings(m, reads, sink, lf, out, out_node_freq);
}

// ---------------------------------------------------------------- Mapping::map_nodes
// Mapping::map_nodes (hint.rs:60-88), the carrier of MultiDbg::hint_kp1_from_hint_k (multi_dbg.rs:1325-1335)
// and PurgeEdgeMap::update_mapping (multi_dbg.rs:1783-1793): for every read position
//   m[node_after] += prob / |node_map(node)|  over the position's (node, prob) and node_after in node_map(node),
// then the MAX_ACTIVE_NODES most probable in descending order.  node_map arrives as a CSR over the OLD nodes.
// One wave per position; the entries are taken in list order (lanes = the images of one node), so every sum
// has a fixed order.
static constexpr int MN_HASH = 4096, MN_CELLS = MN_HASH / 2;
struct MapNodesArgs {
    const uint64_t *pos_off;
    const uint32_t *nodes;
    const double *logp;
    uint64_t n_pos;
    const uint32_t *map_off, *map_nodes;
    uint32_t n_old, n_new;
    RecPool out;
    uint32_t *err;  // [1]
};
// The images of one position are merged in an LDS hash (MN_CELLS distinct images at most).  A position with more
// than that -- 400 entries x a fan-out above 5, far from anything MultiDbg produces -- is done in P passes over
// the key classes `image % P`, each pass folding its cells into a running top-400 list; a class that still
// overflows doubles P and the position starts again (n_new / P <= MN_CELLS ends that for good).  Nothing here
// can spin: probes are bounded by the table never being more than half full + one round of 64 inserts.
__global__ void __launch_bounds__(64) map_nodes_kernel(const MapNodesArgs a) {
    __shared__ uint32_t keys[MN_HASH];
    __shared__ double acc[MN_HASH];
    __shared__ uint16_t cells[MN_CELLS + 64];  // claimed cells in insertion order
    __shared__ uint32_t run_id[2][PHMM_MAX_ACTIVE_NODES];
    __shared__ double run_v[2][PHMM_MAX_ACTIVE_NODES];
    __shared__ int ncell;
    const uint64_t p = blockIdx.x;
    const int lane = threadIdx.x;
    const uint64_t o0 = a.pos_off[p], o1 = a.pos_off[p + 1];
    uint32_t bad = 0;
    // how many images this position has at most (repeats included)
    unsigned long long total = 0;
    for (uint64_t j = o0 + lane; j < o1; j += 64) {
        const uint32_t node = a.nodes[j];
        if (node >= a.n_old) bad = 1;
        else total += a.map_off[node + 1] - a.map_off[node];
    }
    for (int off = 32; off >= 1; off >>= 1) {
        total += __shfl_xor(total, off);
        bad |= (uint32_t)__shfl_xor((int)bad, off);
    }
    uint32_t P = total <= (unsigned long long)MN_CELLS ? 1u : (uint32_t)((total + MN_CELLS / 2 - 1) / (MN_CELLS / 2));
    int run_n = 0, cur = 0;
    for (bool done = bad != 0; !done;) {
        run_n = 0;
        cur = 0;
        bool overflow = false;
        for (uint32_t cls = 0; cls < P && !overflow; cls++) {
            for (int h = lane; h < MN_HASH; h += 64) keys[h] = H_EMPTY;
            if (lane == 0) ncell = 0;
            __syncthreads();
            for (uint64_t j = o0; j < o1 && !overflow; j++) {
                const uint32_t node = a.nodes[j];
                const uint32_t m0 = a.map_off[node], m1 = a.map_off[node + 1];
                const uint32_t len = m1 - m0;
                const double v = exp(a.logp[j]) / (double)(len ? len : 1);
                for (uint32_t q0 = 0; q0 < len && !overflow; q0 += 64) {
                    const uint32_t q = q0 + lane;
                    int cell = -1;
                    bool fresh = false;
                    if (q < len) {
                        const uint32_t key = a.map_nodes[m0 + q];
                        if (key >= a.n_new) bad = 1;
                        else if (P == 1 || key % P == cls) {
                            uint32_t h = (key * 2654435761u) >> 20;
                            for (int probe = 0; probe < MN_HASH; probe++) {
                                const uint32_t old = atomicCAS(&keys[h], H_EMPTY, key);
                                if (old == H_EMPTY) {
                                    fresh = true;
                                    acc[h] = 0.0;
                                }
                                if (old == H_EMPTY || old == key) {
                                    cell = (int)h;
                                    break;
                                }
                                h = (h + 1) & (MN_HASH - 1);
                            }
                        }
                    }
                    const unsigned long long fm = __ballot(fresh);
                    const int base = ncell;
                    if (fresh) cells[base + __popcll(fm & ((1ull << lane) - 1ull))] = (uint16_t)cell;
                    __syncthreads();
                    if (lane == 0) ncell = base + __popcll(fm);
                    // (images of one node are distinct in a well-formed map; a repeated image still adds up, just
                    // through an atomic)
                    if (cell >= 0) atomicAdd(&acc[cell], v);
                    __syncthreads();
                    overflow = ncell > MN_CELLS;  // (uniform: every lane reads the same word)
                }
            }
            if (overflow) break;
            // fold this class into the running list: rank among (cells of the pass) + (running list), keep 400
            const int n = ncell, rn = run_n;
            const int nxt = cur ^ 1;
            for (int j = lane; j < n + rn; j += 64) {
                const double v = j < n ? acc[cells[j]] : run_v[cur][j - n];
                const uint32_t id = j < n ? keys[cells[j]] : run_id[cur][j - n];
                int rank = 0;
                for (int q = 0; q < n; q++) {
                    const double u = acc[cells[q]];
                    rank += (u > v) || (u == v && keys[cells[q]] < id);
                }
                for (int q = 0; q < rn; q++) {
                    const double u = run_v[cur][q];
                    rank += (u > v) || (u == v && run_id[cur][q] < id);
                }
                if (rank < PHMM_MAX_ACTIVE_NODES) {
                    run_id[nxt][rank] = id;
                    run_v[nxt][rank] = v;
                }
            }
            __syncthreads();
            run_n = n + rn < PHMM_MAX_ACTIVE_NODES ? n + rn : PHMM_MAX_ACTIVE_NODES;
            cur = nxt;
        }
        for (int off = 32; off >= 1; off >>= 1) bad |= (uint32_t)__shfl_xor((int)bad, off);
        if (overflow && !bad && P < a.n_new) P *= 2;  // (P >= n_new: one key per class, cannot overflow)
        else done = true;
    }
    const int keep = run_n;
    const uint64_t idb = (uint64_t)((keep + 1) & ~1) * 4;
    const uint64_t bytes = 8 + idb + (uint64_t)keep * 8;
    const uint64_t o = pool_alloc(a.out, bytes);
    if (o + bytes > a.out.cap) bad |= 4;
    if (bad) {
        if (lane == 0) atomicOr(a.err, bad);
        return;
    }
    uint8_t *rec = a.out.base + o;
    if (lane == 0) {
        ((uint32_t *)rec)[0] = (uint32_t)keep;
        ((uint32_t *)rec)[1] = 0;
        a.out.off[p] = o + 8;
    }
    uint32_t *oid = (uint32_t *)(rec + 8);
    double *olp = (double *)(rec + 8 + idb);
    for (int j = lane; j < keep; j += 64) {
        const double v = run_v[cur][j];
        oid[j] = run_id[cur][j];
        olp[j] = v > 0.0 ? log(v) : -INFINITY;
    }
}

void mappings_map_nodes(phmm_model *m, const phmm_reads *reads, const phmm_mappings *mp_in, const uint32_t *map_off,
                        const uint32_t *map_nodes, uint32_t n_old, phmm_mappings **out) {
    hipStream_t s = current_stream();
    const uint64_t n_pos = reads->total;
    upload_mappings(mp_in);
    if (mp_in->d_logp.p == nullptr && mp_in->host_valid)  // host-made mappings upload their probabilities on demand
        mp_in->d_logp.upload(mp_in->logp.data(), std::max<size_t>(mp_in->logp.size(), 1) * sizeof(double));
    const uint64_t n_img = map_off[n_old];
    DevBuf d_off, d_img, d_err;
    d_off.upload(map_off, sizeof(uint32_t) * ((size_t)n_old + 1));
    d_img.upload(map_nodes, sizeof(uint32_t) * std::max<uint64_t>(n_img, 1));
    d_err.reserve(sizeof(uint32_t));
    HIP_CHECK(hipMemsetAsync(d_err.p, 0, sizeof(uint32_t), s));
    uint32_t max_fan = 1;
    for (uint32_t v = 0; v < n_old; v++) max_fan = std::max(max_fan, map_off[v + 1] - map_off[v]);
    MappingSink sink{};
    init_sink(m, reads, sink);
    {
        const uint64_t need = n_pos * 16 + std::min<uint64_t>(mp_in->total_entries * max_fan, n_pos * PHMM_MAX_ACTIVE_NODES) * 12 +
                              (1u << 20);
        if (need > sink.cap) {
            sink.cap = need;
            m->wset().aux[5].reserve(sink.cap);
            sink.mp.base = m->wset().aux[5].as<uint8_t>();
            sink.mp.cap = sink.cap;
        }
    }
    MapNodesArgs a{};
    a.pos_off = mp_in->d_pos_off.as<uint64_t>();
    a.nodes = mp_in->d_nodes.as<uint32_t>();
    a.logp = mp_in->d_logp.as<double>();
    a.n_pos = n_pos;
    a.map_off = d_off.as<uint32_t>();
    a.map_nodes = d_img.as<uint32_t>();
    a.n_old = n_old;
    a.n_new = m->N;
    a.out = sink.mp;
    a.err = d_err.as<uint32_t>();
    if (n_pos) hipLaunchKernelGGL(map_nodes_kernel, dim3((unsigned)n_pos), dim3(64), 0, s, a);
    HIP_CHECK(hipGetLastError());
    uint32_t herr = 0;
    HIP_CHECK(hipMemcpyAsync(&herr, d_err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (herr & 1) PHMM_THROW(PHMM_EINVAL, "map_nodes: a node id is out of range");
    if (herr & 4) PHMM_THROW(PHMM_EINTERNAL, "map_nodes: output pool exhausted");
    std::vector<double> lf = mp_in->read_logp;
    if (lf.size() != reads->R) lf.assign(reads->R, 0.0);
    finish_mappings(m, reads, sink, lf, out, nullptr);
    if (mp_in->read_logp.size() != reads->R) (*out)->read_logp.clear();
}

// PHMMModel::generate_mappings(reads, Some(mappings), use_max_ratio): run_with_mapping
// (freq.rs:72-76) = forward_with_mapping (forward.rs:51-75) + backward_with_mapping
// (backward.rs:59-93), then to_mapping_by_score_ratio / to_mapping(n_active) (hint.rs:124-142).
void generate_mappings_hinted(phmm_model *m, const phmm_reads *reads, const phmm_mappings *mp_in, int use_max_ratio,
                              phmm_mappings **out, double *out_node_freq) {
    hipStream_t s = current_stream();
    const uint64_t R = reads->R, n_pos = reads->total;
    MappingSink sink{};
    init_sink(m, reads, sink);
    {   // every output list is a subset of its input list: the pool bound is exact
        const uint64_t need = n_pos * 16 + mp_in->total_entries * 12 + (1u << 20);
        if (need > sink.cap) {
            sink.cap = need;
            m->wset().aux[5].reserve(sink.cap);
            sink.mp.base = m->wset().aux[5].as<uint8_t>();
            sink.mp.cap = sink.cap;
        }
    }
    upload_reads(reads);
    upload_mappings(mp_in);
    // forward_with_mapping with one record per position
    DevBuf &fpool = m->wset().aux[1], &fmeta = m->wset().aux[2];
    RecPool fp{};
    fp.cap = (n_pos * 24 + mp_in->total_entries * 28) * 3 + (1u << 20);  // a read may be redone in up to 3 capacity classes
    fpool.reserve(fp.cap);
    fmeta.reserve(sizeof(unsigned long long) + sizeof(uint64_t) * (n_pos + 1));
    HIP_CHECK(hipMemsetAsync(fmeta.p, 0, sizeof(unsigned long long) + sizeof(uint64_t) * (n_pos + 1), s));
    fp.base = fpool.as<uint8_t>();
    fp.top = fmeta.as<unsigned long long>();
    fp.off = (uint64_t *)(fmeta.as<char>() + 8);
    std::vector<double> lf(R);
    double tot = 0.0;
    full_prob_reads_hinted(m, reads, mp_in, 1, nullptr, nullptr, lf.data(), &tot, &fp);

    // backward_with_mapping + emit probs, one wave per read (W = 1 "lanes" = reads)
    const int Lb = (int)reads->max_len;
    DevBuf &ctl = m->wset().aux[3];
    size_t cb = 0;
    auto carve = [&](size_t bytes) {
        cb = (cb + 255) / 256 * 256;
        size_t o = cb;
        cb += bytes;
        return o;
    };
    const size_t o_len = carve(sizeof(int) * R), o_sw = carve(sizeof(int) * R), o_stop = carve(sizeof(int) * R),
                 o_err = carve(sizeof(uint32_t) * R), o_lanes = carve(sizeof(uint32_t) * R), o_logp = carve(sizeof(double) * R),
                 o_bases = carve((size_t)R * Lb), o_hand = carve(sizeof(BHandoff));
    ctl.reserve(cb);
    char *cp = (char *)ctl.p;
    HIP_CHECK(hipMemsetAsync(cp, 0, o_bases, s));
    std::vector<int> hlen(R);
    std::vector<uint8_t> hb((size_t)R * Lb, 0xff);
    for (uint64_t r = 0; r < R; r++) {
        const uint64_t len = reads->off[r + 1] - reads->off[r];
        hlen[r] = (int)len;
        std::memcpy(hb.data() + (size_t)r * Lb, reads->bases.data() + reads->off[r], len);
    }
    HIP_CHECK(hipMemcpyAsync(cp + o_len, hlen.data(), sizeof(int) * R, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(cp + o_logp, lf.data(), sizeof(double) * R, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(cp + o_bases, hb.data(), hb.size(), hipMemcpyHostToDevice, s));
    SparseBwdArgs ba{};
    ba.M = sparse_model_of(m);
    ba.d.N = (int)m->N;
    ba.d.len = (const int *)(cp + o_len);
    ba.d.logPf = (double *)(cp + o_logp);
    ba.W = 1;
    ba.Lb = Lb;
    ba.sw = (const int *)(cp + o_sw);
    ba.bases = (const uint8_t *)(cp + o_bases);
    ba.fpool = fp;
    ba.lane_pos0 = reads->d_off.as<uint64_t>();
    ba.map_pos0 = reads->d_off.as<uint64_t>();
    ba.lanes = (const uint32_t *)(cp + o_lanes);
    ba.ratio_lin = std::exp(-m->params.active_node_max_ratio);
    ba.err = (uint32_t *)(cp + o_err);
    ba.list_off = mp_in->d_pos_off.as<uint64_t>();
    ba.list_nodes = mp_in->d_nodes.as<uint32_t>();
    ba.topk = use_max_ratio ? 0 : (int)m->params.n_active_nodes;
    ba.mode = 0;
    ba.stop = (int *)(cp + o_stop);
    ba.hand = (BHandoff *)(cp + o_hand);
    std::vector<uint32_t> cls[2];
    for (uint64_t r = 0; r < R; r++) cls[mp_in->read_max_list[r] <= 64 ? 0 : 1].push_back((uint32_t)r);
    for (int attempt = 0;; attempt++) {
        ba.mpool = sink.mp;
        for (int c = 0; c < 2; c++) {
            if (cls[c].empty()) continue;
            HIP_CHECK(hipMemcpyAsync(cp + o_lanes, cls[c].data(), sizeof(uint32_t) * cls[c].size(), hipMemcpyHostToDevice, s));
            if (c == 0) hipLaunchKernelGGL((sparse_backward_kernel<64>), dim3((unsigned)cls[c].size()), dim3(64), 0, s, ba);
            else hipLaunchKernelGGL((sparse_backward_kernel<KMAX>), dim3((unsigned)cls[c].size()), dim3(64), 0, s, ba);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(s));  // the lane list is reused by the next class
        }
        std::vector<uint32_t> herr(R);
        HIP_CHECK(hipMemcpy(herr.data(), cp + o_err, sizeof(uint32_t) * R, hipMemcpyDeviceToHost));
        bool pool_full = false;
        for (uint64_t r = 0; r < R; r++) {
            if (herr[r] & SP_ERR_POOL) pool_full = true;
            else if (herr[r]) PHMM_THROW(PHMM_EINTERNAL, "backward_with_mapping error " + std::to_string(herr[r]));
        }
        if (!pool_full) break;
        if (attempt >= 4) PHMM_THROW(PHMM_ENOMEM, "mapping pool keeps overflowing");
        sink.cap *= 2;
        m->wset().aux[5].reserve(sink.cap);  // nothing to keep: every record is rewritten
        const unsigned long long zero = 0;
        HIP_CHECK(hipMemcpy(sink.mp.top, &zero, sizeof(zero), hipMemcpyHostToDevice));
        sink.mp.base = m->wset().aux[5].as<uint8_t>();
        sink.mp.cap = sink.cap;
    }
    finish_mappings(m, reads, sink, lf, out, out_node_freq);
}

}  // namespace phmm
