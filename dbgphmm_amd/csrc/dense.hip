// Dense forward / backward / posterior kernels for gfx950 (MI355X).
//
// What they compute (reference file:line, relative to the dbgphmm tree):
//   forward  column: f_step = fm; fi; fmb; fib; fd(fd0 + n_max_gaps x fdt); fe
//                    src/hmmv2/forward.rs:276-306, 337-558
//   backward column: b_step = bd(bd0 + n_max_gaps x bdt); be; bm; bi; bib; bmb
//                    src/hmmv2/backward.rs:216-261, 299-565
//   posteriors:      to_emit_probs / to_state_probs / to_node_freqs
//                    src/hmmv2/table.rs:500-505, src/hmmv2/freq.rs:230-255
//
// How (DESIGN.md sections 3-5):
//   * HBM layout  T[group][pos][node][W]  (f64): the W reads of a read group are the
//     fastest dimension, so a wave touches W consecutive doubles per node and every
//     parent/child gather is a contiguous 8*W-byte row whatever the graph looks like.
//   * scaled linear domain: a column is stored as value * 2^-E[pos] with a power-of-two
//     (hence exact) rescale chosen from the previous column's maximum; log P is
//     recovered as log(sum) + E*ln2.  No exp/log in the inner loop.
//   * the silent Del chain (1 + n_max_gaps dependent sweeps per column in the
//     reference) is folded into per-node "closure" lists built on the host
//     (model.cpp), so one launch per read position has no intra-column dependency:
//     launch `pos` writes m,i of column pos and d of column pos-1.
//   * all reductions (column maxima, end sums, begin sums, node posteriors) are done
//     in a fixed order: results are bit-reproducible run to run.
//   * blockIdx -> node-range mapping is XCD-aware (blocks b and b+8 share an XCD/L2):
//     each XCD owns one contiguous eighth of the node range, so closure gathers of
//     neighbouring node blocks hit the same L2.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <type_traits>

#include "dense_internal.h"
#include "lds_dma.h"

#ifndef PHMM_FWD_PF
#define PHMM_FWD_PF 2
#endif
#ifndef PHMM_BWD_PF
#define PHMM_BWD_PF 1
#endif
// W < 64 instantiations (small read sets, plans of a few deferred reads): their launches do not fill the chip anyway, and
// at 4 waves per SIMD fwd_step<16> spills 19 VGPRs into its row loop (31 -> 27 us per cfg2 launch without them)
#ifndef PHMM_SMALLW_WAVES
#define PHMM_SMALLW_WAVES 3
#endif

namespace phmm {


__device__ __forceinline__ int xcd_block(int b, int nblk8) {
    // physical block b runs on XCD (b % 8); give each XCD a contiguous node range.
    int per = nblk8 >> 3;
    return (b & 7) * per + (b >> 3);
}

// Node record through the constant address space: with a wave-uniform index (W = 64) the compiler
// emits one scalar s_load_dwordx8 instead of per-lane vector loads, which keeps the record out of the
// vector-memory queue (vmcnt is in order: a record load there would make every use wait for the
// row prefetches issued after it).
__device__ __forceinline__ NodeRec load_node(const NodeRec *nodes, int k) {
    typedef const NodeRec __attribute__((address_space(4))) *KPtr;
    const KPtr q = (KPtr)(uintptr_t)nodes;
    NodeRec r;
    r.init = q[k].init;
    r.dinit = q[k].dinit;
    r.tdinit = q[k].tdinit;
    r.emis = q[k].emis;
    r.flags = q[k].flags;
    return r;
}

// ---- LDS-DMA row prefetch (fwd_step<64, true>): glds16, lds_dma.h ----------------------------------------
// wait until at most n vector-memory operations of this wave are outstanding (n wave-uniform, 0..31)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define PHMM_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        PHMM_VMW(1) PHMM_VMW(2) PHMM_VMW(3) PHMM_VMW(4) PHMM_VMW(5) PHMM_VMW(6) PHMM_VMW(7) PHMM_VMW(8)
        PHMM_VMW(9) PHMM_VMW(10) PHMM_VMW(11) PHMM_VMW(12) PHMM_VMW(13) PHMM_VMW(14) PHMM_VMW(15) PHMM_VMW(16)
        PHMM_VMW(17) PHMM_VMW(18) PHMM_VMW(19) PHMM_VMW(20) PHMM_VMW(21) PHMM_VMW(22) PHMM_VMW(23) PHMM_VMW(24)
        PHMM_VMW(25) PHMM_VMW(26) PHMM_VMW(27) PHMM_VMW(28) PHMM_VMW(29) PHMM_VMW(30) PHMM_VMW(31)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef PHMM_VMW
}
#ifndef PHMM_DMA_DEPTH
#define PHMM_DMA_DEPTH 4
#endif
static constexpr int DMA_DEPTH = PHMM_DMA_DEPTH;  // rows in flight per wave
static_assert(DMA_DEPTH - 1 + DMA_DEPTH * 3 <= 31, "vmcnt immediates");

// exponent e with v * 2^-e in [0.5, 1) for v > 0 (normal); 0 for v == 0
__device__ __forceinline__ int exp_of_bits(unsigned long long bits) {
    int be = (int)((bits >> 52) & 0x7ff);
    if (bits == 0ull) return 0;
    if (be == 0) return -1022;  // subnormal maximum: scale up as far as is safe
    return be - 1022;
}
__device__ __forceinline__ double pow2(int e) {  // e in [-1022, 1023]
    return __longlong_as_double((long long)(e + 1023) << 52);
}

// Reduce over all threads of the block that share r = tid % W.  Result valid for tid < W.
template <int W, class Op>
__device__ __forceinline__ double block_reduce_rows(double v, Op op, double *lds) {
#pragma unroll
    for (int off = W; off < 64; off <<= 1) v = op(v, __shfl_xor(v, off));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane < W) lds[wave * 64 + lane] = v;
    __syncthreads();
    double out = v;
    if (threadIdx.x < W) {
        out = lds[threadIdx.x];
#pragma unroll
        for (int w = 1; w < BLOCK / 64; w++) out = op(out, lds[w * 64 + threadIdx.x]);
    }
    return out;
}
// Sum over the W lanes that share one node (contiguous lane group); valid in lane r == 0 of the group (W = 16: in
// every lane).  W = 16 is one DPP row: four row rotations (VALU moves) instead of four ds_bpermute round trips per
// row of the run -- this sum sits in the row loop of bwd_step when node usage is wanted.
template <int W> __device__ __forceinline__ double lanes_sum(double v) {
    if (W == 16) {
        auto ror = [](double x, auto ctrl) {
            const long long b = __double_as_longlong(x);
            const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), decltype(ctrl)::value, 0xf, 0xf, false);
            const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), decltype(ctrl)::value, 0xf, 0xf, false);
            return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
        };
        v += ror(v, std::integral_constant<int, 0x128>());  // row_ror 8, 4, 2, 1
        v += ror(v, std::integral_constant<int, 0x124>());
        v += ror(v, std::integral_constant<int, 0x122>());
        v += ror(v, std::integral_constant<int, 0x121>());
        return v;
    }
#pragma unroll
    for (int off = 1; off < W; off <<= 1) v += __shfl_xor(v, off);
    return v;
}

struct OpMax {
    __device__ double operator()(double a, double b) const { return a > b ? a : b; }
};
struct OpAdd {
    __device__ double operator()(double a, double b) const { return a + b; }
};

// ------------------------------------------------------------------ forward step
// Launch `pos` (0..Lc): column pos of m,i for lanes with pos < len; d of column pos-1
// for lanes with 1 <= pos <= len; the end sum of the last column for lanes with pos == len.
template <int W, bool DMA>
__global__ void __launch_bounds__(BLOCK, (W == 64 ? 4 : PHMM_SMALLW_WAVES)) fwd_step(const DenseArgs a, const int pos) {
    __shared__ double lds[(BLOCK / 64) * 64];
    constexpr bool DMA_ = DMA && W == 64;
    // per wave: DMA_DEPTH slots of [m row 512 B][i row 512 B]
    __shared__ double ring[DMA_ ? (BLOCK / 64) * DMA_DEPTH * 128 : 1];
    const int g = blockIdx.y + a.g_off;
    const int lb = xcd_block(blockIdx.x, a.nblk8);
    constexpr int ROWS = BLOCK / W;
    const int r = threadIdx.x % W;
    const int row = threadIdx.x / W;
    const int len = a.len[g * W + r];
    // warm-up: a lane whose switch to the sparse frontier is decided takes no further dense column
    const bool active = a.wf_sw == nullptr || a.wf_sw[g * W + r] < 0;
    const bool newcol = active && pos < len;
    const bool have_prev = active && pos >= 1 && pos <= len;
    const bool fin = have_prev && pos == len;
    if (!__syncthreads_or(newcol || have_prev)) return;  // the whole read group is done
    const LinParams &lp = a.lp;
    const size_t NW = (size_t)a.N * W;

    int Epos = 0;
    double sc = 1.0, isc = 1.0, ibs = 0.0;
    double thrL = INFINITY, thrU = INFINITY;  // warm-up count thresholds (scaled domain)
    bool collect = false;
    if (have_prev) {
        // column maximum over the nodes (cmaxF) and the InsBegin value of column pos-1, both in that
        // column's stored exponent: the rescale keeps either below 1
        const unsigned long long cm = a.cmaxF[((size_t)g * a.Lc + (pos - 1)) * W + r];
        const int Eprev = a.FE[((size_t)g * (a.Lc + 1) + (pos - 1)) * W + r];
        const double ib_st = exp(a.logib[pos - 1] - (double)Eprev * LN2);
        const unsigned long long ibb = (unsigned long long)__double_as_longlong(ib_st);
        const int e = exp_of_bits(cm > ibb ? cm : ibb);
        sc = pow2(-e);
        isc = pow2(e);
        Epos = Eprev + e;
        // fib, forward.rs:541-545 (read-independent chain, log domain on the host)
        ibs = exp(a.logib[pos - 1] - (double)Epos * LN2);
        if (a.wf_sw) {
            // max over nodes of the column total t = m+i+d lies in [L, U]:
            //   L = max(m,i) (t >= its m and i);  U = ub_a * max(m,i) + ub_b * p_ID * ib  (closure weights)
            const double cmF = __longlong_as_double((long long)cm) * sc;
            thrL = cmF * a.wf_ratio;
            thrU = (a.wf_ub_a * cmF + a.wf_ub_b * lp.p_ID * ibs) * (1.0 + 1e-9) * a.wf_ratio;
            collect = a.wf_mode[g * W + r] != 0;
        }
    }
    const uint8_t x = newcol ? a.bases[((size_t)g * a.Lc + pos) * W + r] : (uint8_t)0;

    const double *pm = a.Fm + ((size_t)g * a.Lc + (pos > 0 ? pos - 1 : 0)) * NW;
    const double *pi = a.Fi + ((size_t)g * a.Lc + (pos > 0 ? pos - 1 : 0)) * NW;
    double *pd = a.Fd + ((size_t)g * a.Lc + (pos > 0 ? pos - 1 : 0)) * NW;
    double *cm_ = a.Fm + ((size_t)g * a.Lc + (pos < a.Lc ? pos : 0)) * NW;
    double *ci_ = a.Fi + ((size_t)g * a.Lc + (pos < a.Lc ? pos : 0)) * NW;

    double vmax = 0.0, esum = 0.0, tmax = 0.0;
    int nsub = 0;
    const bool want_e = fin || (a.eall && have_prev);
    if (lb < a.nblk) {
        // Each row of W lanes walks a RUN of npt consecutive node ids.  On a unitig run
        // (NodeRec flag CHAIN_F: the closure of k is exactly k-1 .. k-6 with unit weights) the
        // ancestors' values come from a register window that is refreshed by the node's own
        // m, i loads -- 16 B read + 24 B written per cell, no closure traffic at all.
        constexpr int H = CHAIN_HOPS;
        const int kbase = lb * (a.npt * ROWS) + row * a.npt;
        // partial sums D_j = sum_{h < j} p_DD^h A[h] over the per-hop sums A[h] of g = p_MD m + p_ID i at the ancestors
        // h+1 hops up (prev column, rescaled), and T = the D_{G+1} of the node before
        double D1 = 0.0, D2 = 0.0, D3 = 0.0, D4 = 0.0, D5 = 0.0, Tk = 0.0;
        double wm0 = 0.0, wi0 = 0.0;
        int nvalid = 0;
        const double c = lp.p_ID * ibs;  // p_MD*mb + p_ID*ib with mb = 0 (fmb)
        const double cb = lp.p_IM * ibs;
        // software pipeline: a ring of PF own-value loads stays in flight -- node k+PF is requested
        // before node k is computed (npt is a multiple of PF; the ring is indexed statically)
        // The prefetches are UNCONDITIONAL loads of a clamped row (every lane, also past the end of the run):
        // a load under a branch makes hipcc wait for it on the spot (s_waitcnt vmcnt(0) at the join), which
        // is what defeated the ring before.  Lanes without work compute on whatever the loads return; every
        // store and every accumulator below is predicated instead.
        constexpr int PF = DMA_ ? DMA_DEPTH : PHMM_FWD_PF;
        double rm[PF], ri[PF];
        const int klast = a.N - 1;
        const bool lane_on = newcol || have_prev;
        // DMA: lane l < 32 fetches reads 2l, 2l+1 of the m row, lane l >= 32 those of the i row
        const double *dma_src = ((threadIdx.x & 63) < 32 ? pm : pi) + 2 * (threadIdx.x & 31);
        const uint32_t ring_base =
            DMA_ ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)&ring[(threadIdx.x >> 6) * DMA_DEPTH * 128]) : 0u;
        const double *ring_w = &ring[DMA_ ? (threadIdx.x >> 6) * DMA_DEPTH * 128 : 0];
        // vector-memory operations a row issues besides its one DMA request: the d store of column pos-1 and the
        // m, i stores of column pos (each skipped by the compiler's branch only when NO lane of the wave has it)
        const int ns = (__any(have_prev && pos >= 1) ? 1 : 0) + (__any(newcol) ? 2 : 0);
#pragma unroll
        for (int u = 0; u < PF; u++) {
            int k0 = kbase + u < klast ? kbase + u : klast;
            if (W == 64) k0 = __builtin_amdgcn_readfirstlane(k0);
            if (DMA_) {
                glds16(dma_src + (size_t)k0 * W, ring_base + (uint32_t)u * 1024u);
            } else {
                // (lanes without work read row 0 over and over: a cache hit instead of HBM traffic)
                rm[u] = pm[(size_t)(lane_on ? k0 : 0) * W + r];
                ri[u] = pi[(size_t)(lane_on ? k0 : 0) * W + r];
            }
        }
        NodeRec nr_next = load_node(a.nodes, W == 64 ? __builtin_amdgcn_readfirstlane(kbase < klast ? kbase : klast)
                                                     : (kbase < klast ? kbase : klast));
        for (int j0 = 0; j0 < a.npt; j0 += PF) {
#pragma unroll
          for (int u = 0; u < PF; u++) {
            const int j = j0 + u;
            if (j >= a.npt) continue;  // (PF need not divide the run length)
            int k = kbase + j;
            if (W == 64) k = __builtin_amdgcn_readfirstlane(k);
            if (k >= a.N) continue;
            const NodeRec nr = nr_next;
            const size_t ik = (size_t)k * W + r;
            // Take delivery of what was requested earlier -- this node's record (a row ago; scalar loads
            // return out of order, so a wait placed after the next request would wait for that too) and its
            // own m, i (PF rows ago) -- BEFORE the next requests go out: the ring slot is then dead when the
            // new load is issued, the load lands in the same registers, and no copy (= no wait) is needed at
            // the loop's back edge.
            double om, oi;
            if (DMA_) {
                // the request of this row went out PF rows ago (j0 == 0: in the prologue); everything issued
                // since is younger: PF-1 requests and ns stores per row
                wait_vmcnt(PF - 1 + (j0 == 0 ? u : PF) * ns);
                om = ring_w[u * 128 + r] * sc;
                oi = ring_w[u * 128 + 64 + r] * sc;
            } else {
                om = rm[u] * sc;
                oi = ri[u] * sc;
            }
            if (W == 64) {
                asm volatile("" : : "s"(nr.flags), "s"(nr.emis), "v"(om), "v"(oi));
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                int kn = k + 1 < klast ? k + 1 : klast;
                if (W == 64) kn = __builtin_amdgcn_readfirstlane(kn);
                nr_next = load_node(a.nodes, kn);
                int kp = k + PF < klast ? k + PF : klast;
                if (W == 64) kp = __builtin_amdgcn_readfirstlane(kp);
                if (DMA_) {
                    glds16(dma_src + (size_t)kp * W, ring_base + (uint32_t)u * 1024u);
                } else {
                    rm[u] = pm[(size_t)(lane_on ? kp : 0) * W + r];
                    ri[u] = pi[(size_t)(lane_on ? kp : 0) * W + r];
                }
            }
            double mnew, inew = 0.0;
            const double pe = (uint8_t)nr.emis == x ? lp.p_match : lp.p_mismatch;
            {   // (column 0 is fwd_init's: no branch on pos here -- the loop-carried state would be copied at its join)
                double m1, i1, dacc, tacc;
                if (a.hop_mode) {
                    if (!((nr.flags & CHAIN_F) && nvalid)) {
                        // first node of the run, behind a branch, or a merge: the per-hop sums A[h] of the ancestors
                        // from the hop entries, folded into the partial sums D_j = sum_{h < j} p_DD^h A[h]
                        double wg[H];
#pragma unroll
                        for (int h = 0; h < H; h++) wg[h] = 0.0;
                        wm0 = wi0 = 0.0;
                        const uint32_t o0 = a.fh_off[k], o1 = a.fh_off[k + 1];
                        for (uint32_t q = o0; q < o1; q++) {
                            const HopEntry en = a.fh[q];
                            const size_t ix = (size_t)en.node * W + r;
                            const double vm = pm[ix] * sc, vi = pi[ix] * sc;
                            const double wgv = en.w * (lp.p_MD * vm + lp.p_ID * vi);
                            const int hh = (int)(en.hop_emis & 0xff) - 1;
#pragma unroll
                            for (int h = 0; h < H; h++)
                                if (hh == h) wg[h] += wgv;
                            if (hh == 0) {
                                wm0 += en.w * vm;
                                wi0 += en.w * vi;
                            }
                        }
                        const double q = lp.p_DD;
                        D1 = wg[0];
                        D2 = wg[0] + q * wg[1];
                        D3 = wg[0] + q * (wg[1] + q * wg[2]);
                        D4 = wg[0] + q * (wg[1] + q * (wg[2] + q * wg[3]));
                        D5 = wg[0] + q * (wg[1] + q * (wg[2] + q * (wg[3] + q * wg[4])));
                        // T = sum_{1 <= h <= G+1} p_DD^(h-1) A[h]: what the node before this one would have had as
                        // its D_{G+1} if the window had just slid
                        Tk = 0.0;
#pragma unroll
                        for (int h = H - 1; h >= 1; h--)
                            if (h <= lp.n_max_gaps + 1) Tk = wg[h] + q * Tk;
                        nvalid = 1;
                    }
                    m1 = wm0;
                    i1 = wi0;
                    // d-closure sum = D_{G+1}, its one-hop-shifted twin = the D_{G+1} of the node before (forward.rs:423-524
                    // unrolled: sum_{h <= G} p_DD^h A[h] and sum_{1 <= h <= G+1} p_DD^(h-1) A[h])
                    const int G = lp.n_max_gaps;
                    dacc = G == 4 ? D5 : (G == 3 ? D4 : (G == 2 ? D3 : (G == 1 ? D2 : D1)));
                    tacc = Tk;
                } else {
                    m1 = i1 = dacc = tacc = 0.0;
                    const uint32_t o0 = a.fc_off[k], o1 = a.fc_off[k + 1];
                    for (uint32_t q = o0; q < o1; q++) {
                        const FwdEntry en = a.fc[q];
                        const size_t ix = (size_t)en.node * W + r;
                        const double vm = pm[ix] * sc, vi = pi[ix] * sc;
                        const double gg = lp.p_MD * vm + lp.p_ID * vi;
                        m1 += en.w1 * vm;
                        i1 += en.w1 * vi;
                        dacc += en.wD * gg;
                        tacc += en.wT * gg;
                    }
                }
                const double dprev = dacc + c * nr.dinit;
                const double td = tacc + c * nr.tdinit;
                if (have_prev) pd[ik] = dprev * isc;  // stored in column pos-1's own exponent
                mnew = pe * (lp.p_MM * m1 + lp.p_IM * i1 + lp.p_DM * td + nr.init * cb);
                inew = lp.p_random * (lp.p_MI * om + lp.p_II * oi + lp.p_DI * dprev);
                const double tk = om + oi + dprev;
                if (want_e) esum += tk;
                if (have_prev) tmax = fmax(tmax, tk);
                if (tk > thrL) {
                    // (rare) inside the score ratio of the column maximum, as far as this launch can tell
                    nsub += tk > thrU ? 1 : 0;
                    if (collect) {
                        const int gi = g * W + r;
                        const int slot = atomicAdd(&a.wf_cnt[gi], 1);
                        if (slot < WF_CAP) {
                            a.wf_node[(size_t)gi * WF_CAP + slot] = (uint32_t)k;
                            a.wf_tot[(size_t)gi * WF_CAP + slot] = tk * isc;
                        }
                    }
                }
                // Slide: this node becomes "k-1" of the next one.  The window of per-hop sums shifts by one hop with
                // this node's own g in front, so the partial sums obey D_j' = g + p_DD D_{j-1}: four multiply-adds in
                // place, no window registers to move, no polynomial to re-evaluate.
                {
                    const double gk = lp.p_MD * om + lp.p_ID * oi, q = lp.p_DD;
                    Tk = dacc;
                    D5 = gk + q * D4;
                    D4 = gk + q * D3;
                    D3 = gk + q * D2;
                    D2 = gk + q * D1;
                    D1 = gk;
                }
                wm0 = om;
                wi0 = oi;
            }
            if (newcol) {
                cm_[ik] = mnew;
                ci_[ik] = inew;
                vmax = fmax(vmax, fmax(mnew, inew));
            }
          }
        }
    }
    if (DMA_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may outlive the wave's use of LDS
    // column maximum over the nodes -> next launch's rescale (which also looks at the InsBegin value)
    const double bm = block_reduce_rows<W>(vmax, OpMax(), lds);
    if (threadIdx.x < W && newcol && lb < a.nblk)
        atomicMax(&a.cmaxF[((size_t)g * a.Lc + pos) * W + r], (unsigned long long)__double_as_longlong(bm));
    if (a.tmaxF && pos >= 1) {
        // column maximum of the per-node totals m+i+d (top_nodes_by_score_ratio, table.rs:134-149),
        // in the stored column's own exponent
        const double bt = block_reduce_rows<W>(tmax * isc, OpMax(), lds);
        if (threadIdx.x < W && have_prev && lb < a.nblk)
            atomicMax(&a.tmaxF[((size_t)g * a.Lc + (pos - 1)) * W + r], (unsigned long long)__double_as_longlong(bt));
    }
    if (a.wf_sw) {
        const double bn = block_reduce_rows<W>((double)nsub, OpAdd(), lds);
        if (threadIdx.x < W && bn > 0.0) atomicAdd(&a.wf_sub[g * W + r], (int)bn);
    }
    if (a.eall || __syncthreads_or(fin)) {
        const double bs = block_reduce_rows<W>(esum, OpAdd(), lds);
        if (threadIdx.x < W && want_e && lb < a.nblk) {
            const size_t col = a.eall ? (size_t)(pos - 1) : 0;
            const size_t ecols = a.eall ? (size_t)a.Lc : 1;
            a.epart[(((size_t)g * ecols + col) * a.nblk8 + lb) * W + r] = bs;
        }
    }
    if (lb == 0 && threadIdx.x < W && (newcol || have_prev))
        a.FE[((size_t)g * (a.Lc + 1) + pos) * W + r] = Epos;
}

// Column 0 (f_init + the first f_step, forward.rs:255-306 with mb = 1 and everything else 0): m = e_k(x) init_k p_MM,
// i = 0; the column maximum for launch 1's rescale; exponent 0.  Same thread -> node mapping as fwd_step.
template <int W>
__global__ void __launch_bounds__(BLOCK) fwd_init(const DenseArgs a) {
    __shared__ double lds[(BLOCK / 64) * 64];
    const int g = blockIdx.y + a.g_off;
    const int lb = xcd_block(blockIdx.x, a.nblk8);
    constexpr int ROWS = BLOCK / W;
    const int r = threadIdx.x % W, row = threadIdx.x / W;
    const int len = a.len[g * W + r];
    const bool active = a.wf_sw == nullptr || a.wf_sw[g * W + r] < 0;
    const bool newcol = active && 0 < len;
    if (!__syncthreads_or(newcol)) return;
    const size_t NW = (size_t)a.N * W;
    double *cm_ = a.Fm + (size_t)g * a.Lc * NW, *ci_ = a.Fi + (size_t)g * a.Lc * NW;
    const uint8_t x = newcol ? a.bases[(size_t)g * a.Lc * W + r] : (uint8_t)0;
    double vmax = 0.0;
    if (lb < a.nblk && newcol) {
        const int kbase = lb * (a.npt * ROWS) + row * a.npt;
        for (int j = 0; j < a.npt; j++) {
            const int k = kbase + j;
            if (k >= a.N) break;
            const NodeRec nr = a.nodes[k];
            const double mnew = ((uint8_t)nr.emis == x ? a.lp.p_match : a.lp.p_mismatch) * nr.init * a.lp.p_MM;
            cm_[(size_t)k * W + r] = mnew;
            ci_[(size_t)k * W + r] = 0.0;
            vmax = fmax(vmax, mnew);
        }
    }
    const double bm = block_reduce_rows<W>(vmax, OpMax(), lds);
    if (threadIdx.x < W && newcol && lb < a.nblk)
        atomicMax(&a.cmaxF[(size_t)g * a.Lc * W + r], (unsigned long long)__double_as_longlong(bm));
    if (lb == 0 && threadIdx.x < W && newcol) a.FE[(size_t)g * (a.Lc + 1) * W + r] = 0;
}

// log P(read) = ln(p_end * sum_k (m+i+d)) of the last column (fe, forward.rs:554-558)
template <int W>
__global__ void __launch_bounds__(BLOCK) fwd_finish(const DenseArgs a) {
    __shared__ double lds[(BLOCK / 64) * 64];
    const int g = blockIdx.x;
    const int col = blockIdx.y;  // 0 unless eall
    constexpr int ROWS = BLOCK / W;
    const int r = threadIdx.x % W, row = threadIdx.x / W;
    const int len = a.len[g * W + r];
    const size_t ecols = a.eall ? (size_t)a.Lc : 1;
    double s = 0.0;
    const bool live = a.eall ? col < len : len > 0;
    if (live)
        for (int b = row; b < a.nblk; b += ROWS)
            s += a.epart[(((size_t)g * ecols + col) * a.nblk8 + b) * W + r];
    const double tot = block_reduce_rows<W>(s, OpAdd(), lds);
    if (threadIdx.x < W && live) {
        // the sum of column c was taken in the exponent of launch c+1
        const int E = a.FE[((size_t)g * (a.Lc + 1) + (a.eall ? col + 1 : len)) * W + r];
        const double lp = log(a.lp.p_end * tot) + (double)E * LN2;
        if (a.eall) a.logE[((size_t)g * a.Lc + col) * W + r] = lp;
        if (!a.eall || col == len - 1) a.logPf[g * W + r] = lp;
    }
}

__device__ __forceinline__ double dev_logadd(double x, double y) {
    const double hi = x >= y ? x : y, lo = x >= y ? y : x;
    if (lo == -INFINITY) return hi;
    return hi + log1p(exp(lo - hi));
}

// Begin-state chain of backward column p from the per-block partial sums of launch p
// (bmb / bib, backward.rs:499-555), log domain.  Runs in block lb==0 of launch p-1 and
// in bwd_finish for p == 0.
template <int W>
__device__ __forceinline__ void bwd_chain(const DenseArgs &a, int g, int p, double *lds) {
    constexpr int ROWS = BLOCK / W;
    const int r = threadIdx.x % W, row = threadIdx.x / W;
    const int len = a.len[g * W + r];
    const bool live = p < len;
    double s1 = 0.0, s2 = 0.0;
    if (live) {
        const double *bp = a.bpart + ((size_t)(p & 1) * a.ng + g) * a.nblk8 * W * 2;
        for (int b = row; b < a.nblk; b += ROWS) {
            s1 += bp[((size_t)b * W + r) * 2 + 0];
            s2 += bp[((size_t)b * W + r) * 2 + 1];
        }
    }
    const double t1 = block_reduce_rows<W>(s1, OpAdd(), lds);
    const double t2 = block_reduce_rows<W>(s2, OpAdd(), lds);
    if (threadIdx.x < W && live) {
        const size_t ix = ((size_t)g * (a.Lc + 1) + p) * W + r;
        const double E = (double)a.BE[ix] * LN2;
        const double ibn = (p + 1 < len) ? a.logibB[ix + W] : -INFINITY;  // b_init: ib = 0
        const double lr = log(a.lp.p_random);
        a.logmbB[ix] = dev_logadd(log(t1) + E, log(a.lp.p_MI) + lr + ibn);
        a.logibB[ix] = dev_logadd(log(t2) + E, log(a.lp.p_II) + lr + ibn);
    }
}

// ------------------------------------------------------------------ backward step
// Launch `pos` (Lc-1 .. 0): column pos of m,i,d for lanes with pos < len, fused with the
// posterior accumulation  S[pos] = F.tables[pos-1] (.) B.tables[pos] / P  (and the
// j = L term F.tables[L-1] (.) B.init / P when pos == len-1).
static constexpr int BDMA_DEPTH = 3;      // rows in flight per wave (5 planes = 2.5 KB per row)
static constexpr int BDMA_SLOT = 5 * 64;  // doubles per slot: [B' m][B' i][F m][F i][F d]
template <int W, bool DMA>
__global__ void __launch_bounds__(BLOCK, (W == 64 ? 4 : PHMM_SMALLW_WAVES)) bwd_step(const DenseArgs a, const int pos) {
    __shared__ double lds[(BLOCK / 64) * 64];
    constexpr bool DMA_ = DMA && W == 64;
    __shared__ double ring[DMA_ ? (BLOCK / 64) * BDMA_DEPTH * BDMA_SLOT : 1];
    const int g = blockIdx.y + a.g_off;
    const int lb = xcd_block(blockIdx.x, a.nblk8);
    constexpr int ROWS = BLOCK / W;
    const int r = threadIdx.x % W;
    const int row = threadIdx.x / W;
    const int len = a.len[g * W + r];
    // dense backward columns of this read: 0 .. bstart (the whole read for dense runs; up to the
    // dense/sparse switch for backward_by_forward, backward.rs:101-142)
    // bit 30 of bstart: the read has a sparse tail whose F.tables[len-1] (.) b_init term is
    // produced by the sparse backward kernel
    const int braw = a.bstart ? a.bstart[g * W + r] : len - 1;
    const bool sparse_tail = (braw & (1 << 30)) != 0;
    const int bstart = braw & ~(1 << 30);
    const bool live = pos < len && pos <= bstart;
    const bool first = live && pos == len - 1;  // previous table is b_init (backward.rs:197-211)
    const LinParams &lp = a.lp;
    const size_t NW = (size_t)a.N * W;

    if (lb == 0 && pos + 1 < a.Lc) bwd_chain<W>(a, g, pos + 1, lds);

    // A block none of whose lanes has this column (a read group past its last dense backward column: groups are
    // sorted by their reads' switch positions, so whole groups go quiet at the high columns) leaves its zero partial
    // sums and is done; without this it walked its run of rows for nothing.
    if (!__syncthreads_or(live)) {
        if (threadIdx.x < W && lb < a.nblk) {
            double *bp = a.bpart + ((size_t)(pos & 1) * a.ng + g) * a.nblk8 * W * 2;
            bp[((size_t)lb * W + r) * 2 + 0] = 0.0;
            bp[((size_t)lb * W + r) * 2 + 1] = 0.0;
        }
        if (a.want_map && a.Prun) a.Prun[((size_t)g * a.nblk8 + lb) * BLOCK + threadIdx.x] = 0.0;
        return;
    }

    int Epos = 0;
    double sc = 1.0;
    if (live && !first) {
        const unsigned long long cm = a.cmaxB[((size_t)g * a.Lc + (pos + 1)) * W + r];
        const int e = exp_of_bits(cm);
        sc = pow2(-e);
        Epos = a.BE[((size_t)g * (a.Lc + 1) + (pos + 1)) * W + r] + e;
    }
    const uint8_t x = live ? a.bases[((size_t)g * a.Lc + pos) * W + r] : (uint8_t)0;
    const int cn = a.bcols == 2 ? ((pos + 1) & 1) : (pos + 1 < a.Lc ? pos + 1 : 0);
    const int cc = a.bcols == 2 ? (pos & 1) : pos;
    const double *nm = a.Bm + ((size_t)g * a.bcols + cn) * NW;
    const double *ni = a.Bi + ((size_t)g * a.bcols + cn) * NW;
    double *om = a.Bm + ((size_t)g * a.bcols + cc) * NW;
    double *oi = a.Bi + ((size_t)g * a.bcols + cc) * NW;
    double *od = a.Bd ? a.Bd + ((size_t)g * a.bcols + cc) * NW : nullptr;

    // posterior weights  2^(FE+BE) / P
    double wgt = 0.0, wgt2 = 0.0;
    const double *fm = nullptr, *fi = nullptr, *fd = nullptr, *gm = nullptr, *gi = nullptr, *gd = nullptr;
    const bool want_post = a.want_freq || a.want_map;
    if (want_post && live) {
        const double lpf = a.logPf[g * W + r];
        if (lpf > -INFINITY) {
            if (pos >= 1) {
                const int fe = a.FE[((size_t)g * (a.Lc + 1) + (pos - 1)) * W + r];
                wgt = exp((double)(fe + Epos) * LN2 - lpf);
            }
            if (first && !sparse_tail) {
                const int fe = a.FE[((size_t)g * (a.Lc + 1) + pos) * W + r];
                wgt2 = exp((double)fe * LN2 - lpf) * lp.p_end;
            }
        }
    }
    if (want_post) {
        const size_t cprev = ((size_t)g * a.Lc + (pos > 0 ? pos - 1 : 0)) * NW;
        const size_t ccur = ((size_t)g * a.Lc + pos) * NW;
        fm = a.Fm + cprev; fi = a.Fi + cprev; fd = a.Fd + cprev;
        gm = a.Fm + ccur; gi = a.Fi + ccur; gd = a.Fd + ccur;
    }

    double vmax = 0.0, s1 = 0.0, s2 = 0.0, pmx = 0.0, pmx2 = 0.0;
    double *Pa = a.want_map ? a.Pa + (size_t)g * NW : nullptr;
    double *Pb = a.want_map ? a.Pb + (size_t)g * NW : nullptr;
    if (lb < a.nblk) {
        // Rows walk their run of npt consecutive nodes in DECREASING order; on a unitig run
        // (CHAIN_B: the descendants within 6 hops are v+1 .. v+6 with unit weights) the values
        // h[u] = e_u(x) m'[u], q[u] = p_r i'[u] come from a register window fed by own loads.
        constexpr int H = CHAIN_HOPS;
        const int kbase = lb * (a.npt * ROWS) + row * a.npt;
        // partial sums over the per-hop sums of h, q at the descendants (column pos+1, rescaled): E_j = sum_{h < j}
        // p_DD^h Hs[h], Q_j likewise, TE = the E_{G+1} of the node above
        double E1 = 0.0, E2 = 0.0, E3 = 0.0, E4 = 0.0, E5 = 0.0, Q1 = 0.0, Q2 = 0.0, Q3 = 0.0, Q4 = 0.0, Q5 = 0.0, TE = 0.0;
        int nvalid = 0;
        // software pipeline: a ring of PFB rows (own B values of column pos+1 and the F column of the
        // posterior) stays in flight.  As in fwd_step the loads are unconditional (clamped rows, every
        // lane; pointers that a launch does not use alias a valid plane) and a ring slot is consumed before
        // it is requested again, so that hipcc neither waits at a join nor copies at the back edge:
        // B values are consumed at the top of a row and re-requested there, F values at its end.
        constexpr int PFB = DMA_ ? BDMA_DEPTH : PHMM_BWD_PF;
        double nx_m[PFB], nx_i[PFB], nx_fm[PFB], nx_fi[PFB], nx_fd[PFB];
        const double *lfm = want_post ? fm : nm, *lfi = want_post ? fi : nm, *lfd = want_post ? fd : nm;
        // rows kbase+jtop .. kbase (jtop clamps the last block of the column)
        const int jtop = a.npt - 1 < a.N - 1 - kbase ? a.npt - 1 : a.N - 1 - kbase;
        // LDS-DMA variant (see fwd_step): three requests per row -- A: B' m | B' i, B: F m | F i, C: F d (lower
        // half of the wave) -- into slot u of this wave's ring; A is consumed at the top of a row and requested
        // again there, B and C at its end.  Vector-memory operations of a row besides the 3 requests: n1 stores
        // of B (before the F values are read) and n2 of the emit-prob plane (after).
        const int l64 = threadIdx.x & 63;
        const double *srcA = (l64 < 32 ? nm : ni) + 2 * (l64 & 31);
        const double *srcB = (l64 < 32 ? lfm : lfi) + 2 * (l64 & 31);
        const double *srcC = lfd + 2 * (l64 & 31);
        const uint32_t ring_base =
            DMA_ ? (uint32_t)__builtin_amdgcn_readfirstlane(
                       (int)(uint32_t)(uintptr_t)&ring[(threadIdx.x >> 6) * BDMA_DEPTH * BDMA_SLOT])
                 : 0u;
        const double *ring_w = &ring[DMA_ ? (threadIdx.x >> 6) * BDMA_DEPTH * BDMA_SLOT : 0];
        const int n1 = __any(live) ? 2 + (od ? 1 : 0) : 0;
        const int ns = n1 + ((a.want_map && __any(live)) ? 1 : 0) + (a.want_freq ? 2 : 0);
#pragma unroll
        for (int u = 0; u < PFB; u++) {
            int v0 = kbase + (jtop - u > 0 ? jtop - u : 0);
            if (v0 > a.N - 1) v0 = a.N - 1;  // (a run past the end of the column: nothing is consumed)
            if (W == 64) v0 = __builtin_amdgcn_readfirstlane(v0);
            // (lanes that are not live read row 0 over and over: a cache hit instead of HBM traffic)
            const size_t i0 = (size_t)(live ? v0 : 0) * W + r;
            nx_m[u] = nx_i[u] = nx_fm[u] = nx_fi[u] = nx_fd[u] = 0.0;
            if (DMA_) {
                const uint32_t slot = ring_base + (uint32_t)u * (BDMA_SLOT * 8);
                glds16(srcA + (size_t)v0 * W, slot);
                glds16(srcB + (size_t)v0 * W, slot + 1024u);
                if (l64 < 32) glds16(srcC + (size_t)v0 * W, slot + 2048u);
            } else if (jtop >= 0) {
                nx_m[u] = nm[i0];
                nx_i[u] = ni[i0];
                nx_fm[u] = lfm[i0];
                nx_fi[u] = lfi[i0];
                nx_fd[u] = lfd[i0];
            }
        }
        NodeRec nr_next{};
        if (jtop >= 0) {
            int v0 = kbase + jtop;
            if (W == 64) v0 = __builtin_amdgcn_readfirstlane(v0);
            nr_next = load_node(a.nodes, v0);
        }
        for (int j0 = jtop; j0 >= 0; j0 -= PFB) {
#pragma unroll
          for (int u = 0; u < PFB; u++) {
            const int j = j0 - u;
            if (j < 0) continue;
            int v = kbase + j;
            if (W == 64) v = __builtin_amdgcn_readfirstlane(v);
            double contrib = 0.0;
            {
                const NodeRec nr = nr_next;
                const size_t iv = (size_t)v * W + r;
                const bool first_group = j0 == jtop;
                double cur_m, cur_i;
                if (DMA_) {
                    // younger than request A of this row: its B and C, then per row since 3 requests + ns others
                    wait_vmcnt(2 + 3 * (PFB - 1) + (first_group ? u : PFB) * ns);
                    cur_m = ring_w[u * BDMA_SLOT + r] * sc;
                    cur_i = ring_w[u * BDMA_SLOT + 64 + r] * sc;
                } else {
                    cur_m = nx_m[u] * sc;
                    cur_i = nx_i[u] * sc;
                }
                if (W == 64) {
                    asm volatile("" : : "s"(nr.flags), "s"(nr.emis), "v"(cur_m), "v"(cur_i));
                    __builtin_amdgcn_sched_barrier(0);
                }
                int vp = v - PFB > kbase ? v - PFB : kbase;
                if (W == 64) vp = __builtin_amdgcn_readfirstlane(vp);
                {
                    int vn = v - 1 > kbase ? v - 1 : kbase;
                    if (W == 64) vn = __builtin_amdgcn_readfirstlane(vn);
                    nr_next = load_node(a.nodes, vn);
                    if (DMA_) {
                        glds16(srcA + (size_t)vp * W, ring_base + (uint32_t)u * (BDMA_SLOT * 8));
                    } else {
                        nx_m[u] = nm[(size_t)(live ? vp : 0) * W + r];
                        nx_i[u] = ni[(size_t)(live ? vp : 0) * W + r];
                    }
                }
                double cur_fm = nx_fm[u], cur_fi = nx_fi[u], cur_fd = nx_fd[u];
                const double m0 = first ? lp.p_end : cur_m;
                const double q0 = lp.p_random * (first ? lp.p_end : cur_i);
                const double ev = (uint8_t)nr.emis == x ? lp.p_match : lp.p_mismatch;
                double a1, ad, at, qd, qt;
                if (a.hop_mode) {
                    if (!((nr.flags & CHAIN_B) && nvalid)) {
                        // first node of the run, in front of a merge, or a branch node: the per-hop sums of the
                        // descendants from the hop entries, folded into E_j = sum_{h < j} p_DD^h Hs[h], Q_j likewise
                        double wh[H], wq[H];
#pragma unroll
                        for (int h = 0; h < H; h++) wh[h] = wq[h] = 0.0;
                        const uint32_t o0 = a.bh_off[v], o1 = a.bh_off[v + 1];
                        for (uint32_t q = o0; q < o1; q++) {
                            const HopEntry en = a.bh[q];
                            const size_t ix = (size_t)en.node * W + r;
                            const double mu = first ? lp.p_end : nm[ix] * sc;
                            const double iu = first ? lp.p_end : ni[ix] * sc;
                            const double hv = en.w * ((uint8_t)(en.hop_emis >> 8) == x ? lp.p_match : lp.p_mismatch) * mu;
                            const double qv = en.w * lp.p_random * iu;
                            const int hh = (int)(en.hop_emis & 0xff) - 1;
#pragma unroll
                            for (int h = 0; h < H; h++)
                                if (hh == h) {
                                    wh[h] += hv;
                                    wq[h] += qv;
                                }
                        }
                        const double q = lp.p_DD;
                        E1 = wh[0];
                        E2 = wh[0] + q * wh[1];
                        E3 = wh[0] + q * (wh[1] + q * wh[2]);
                        E4 = wh[0] + q * (wh[1] + q * (wh[2] + q * wh[3]));
                        E5 = wh[0] + q * (wh[1] + q * (wh[2] + q * (wh[3] + q * wh[4])));
                        Q1 = wq[0];
                        Q2 = wq[0] + q * wq[1];
                        Q3 = wq[0] + q * (wq[1] + q * wq[2]);
                        Q4 = wq[0] + q * (wq[1] + q * (wq[2] + q * wq[3]));
                        Q5 = wq[0] + q * (wq[1] + q * (wq[2] + q * (wq[3] + q * wq[4])));
                        TE = 0.0;
#pragma unroll
                        for (int h = H - 1; h >= 1; h--)
                            if (h <= lp.n_max_gaps + 1) TE = wh[h] + q * TE;
                        nvalid = 1;
                    }
                    // a1 = the one-hop sum; ad = sum_{h <= G} p_DD^h Hs[h] = E_{G+1}; at = its one-hop-shifted twin =
                    // the E_{G+1} of the node above; qt = Q_{G+1}; qd = p_DD Q_G   (backward.rs:299-483 unrolled)
                    const int G = lp.n_max_gaps;
                    a1 = E1;
                    ad = G == 4 ? E5 : (G == 3 ? E4 : (G == 2 ? E3 : (G == 1 ? E2 : E1)));
                    at = TE;
                    qt = G == 4 ? Q5 : (G == 3 ? Q4 : (G == 2 ? Q3 : (G == 1 ? Q2 : Q1)));
                    qd = lp.p_DD * (G == 4 ? Q4 : (G == 3 ? Q3 : (G == 2 ? Q2 : (G == 1 ? Q1 : 0.0))));
                } else {
                    a1 = ad = at = qd = qt = 0.0;
                    const uint32_t o0 = a.bc_off[v], o1 = a.bc_off[v + 1];
                    for (uint32_t q = o0; q < o1; q++) {
                        const BwdEntry en = a.bc[q];
                        const size_t ix = (size_t)en.node * W + r;
                        const double mu = first ? lp.p_end : nm[ix] * sc;
                        const double iu = first ? lp.p_end : ni[ix] * sc;
                        const double h = ((uint8_t)en.emis == x ? lp.p_match : lp.p_mismatch) * mu;
                        const double qq = lp.p_random * iu;
                        a1 += en.c1 * h;
                        ad += en.cAd * h;
                        at += en.cAt * h;
                        qd += en.cQd * qq;
                        qt += en.cAd * qq;
                    }
                }
                // slide: this node becomes "v+1" of the next (lower) one -- E_j' = h + p_DD E_{j-1}, Q_j' = q + p_DD Q_{j-1}
                // in place (see fwd_step)
                if (a.hop_mode) {
                    const double hk = ev * m0, q = lp.p_DD;
                    TE = ad;
                    E5 = hk + q * E4;
                    E4 = hk + q * E3;
                    E3 = hk + q * E2;
                    E2 = hk + q * E1;
                    E1 = hk;
                    Q5 = q0 + q * Q4;
                    Q4 = q0 + q * Q3;
                    Q3 = q0 + q * Q2;
                    Q2 = q0 + q * Q1;
                    Q1 = q0;
                }
                const double d = lp.p_DM * ad + lp.p_DI * (q0 + qd);
                const double td = lp.p_DM * at + lp.p_DI * qt;
                const double m = lp.p_MM * a1 + lp.p_MD * td + lp.p_MI * q0;
                const double i = lp.p_IM * a1 + lp.p_ID * td + lp.p_II * q0;
                if (live) {
                    om[iv] = m;
                    oi[iv] = i;
                    if (od) od[iv] = d;
                    vmax = fmax(vmax, fmax(m, i));
                    const double in = nr.init;
                    s1 += in * (lp.p_MM * ev * m0 + lp.p_MD * d);
                    s2 += in * (lp.p_IM * ev * m0 + lp.p_ID * d);
                }
                if (DMA_) {
                    // younger than request C of this row: per row since 3 requests + ns others, plus request A
                    // and the n1 stores of this row
                    wait_vmcnt(1 + n1 + 3 * (PFB - 1) + (first_group ? (u < PFB - 1 ? u : PFB - 1) : PFB - 1) * ns);
                    cur_fm = ring_w[u * BDMA_SLOT + 128 + r];
                    cur_fi = ring_w[u * BDMA_SLOT + 192 + r];
                    cur_fd = ring_w[u * BDMA_SLOT + 256 + r];
                }
                double c1 = 0.0, c2 = 0.0;
                if (wgt != 0.0) c1 = wgt * (cur_fm * m + cur_fi * i + cur_fd * d);
                if (wgt2 != 0.0) c2 = wgt2 * (gm[iv] + gi[iv] + gd[iv]);
                contrib = c1 + c2;
                if (a.want_map && live) {
                    // emit probs of merged index pos (and of merged index len when `first`):
                    // kept for post_collect (to_mapping_by_score_ratio, hint.rs:135-142)
                    Pa[iv] = c1;
                    pmx = fmax(pmx, c1);
                    if (first && !sparse_tail) {
                        Pb[iv] = c2;
                        pmx2 = fmax(pmx2, c2);
                    }
                }
                {
                    // this slot's F values are spent: request the row PFB below
                    if (DMA_) {
                        if (W == 64) {
                            asm volatile("" : : "v"(contrib));
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        const uint32_t slot = ring_base + (uint32_t)u * (BDMA_SLOT * 8);
                        glds16(srcB + (size_t)vp * W, slot + 1024u);
                        if (l64 < 32) glds16(srcC + (size_t)vp * W, slot + 2048u);
                    } else {
                        nx_fm[u] = lfm[(size_t)(live ? vp : 0) * W + r];
                        nx_fi[u] = lfi[(size_t)(live ? vp : 0) * W + r];
                        nx_fd[u] = lfd[(size_t)(live ? vp : 0) * W + r];
                    }
                }
            }
            if (a.want_freq) {
                // Fire-and-forget hardware f64 add (no value returned, nothing to wait for): a plain `+=` is a
                // load -> add -> store chain that put one memory round trip into EVERY row of the run (vmcnt is in
                // order), which is what bounded this kernel on small graphs.  One row of one launch owns the
                // address and launches are ordered, so the sum keeps its fixed order.
                const double tot = lanes_sum<W>(contrib);
                if (r == 0) (void)unsafeAtomicAdd(&a.accg[(size_t)g * a.N + v], tot);
            }
          }
        }
    }
    if (DMA_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may outlive the wave's use of LDS
    const double bmx = block_reduce_rows<W>(vmax, OpMax(), lds);
    if (threadIdx.x < W && live && lb < a.nblk)
        atomicMax(&a.cmaxB[((size_t)g * a.Lc + pos) * W + r], (unsigned long long)__double_as_longlong(bmx));
    if (a.want_map) {
        // the run's own maximum: post_collect reads one value per run and only revisits the runs above the threshold
        if (a.Prun) a.Prun[((size_t)g * a.nblk8 + lb) * BLOCK + threadIdx.x] = pmx;
        const double p1 = block_reduce_rows<W>(pmx, OpMax(), lds);
        if (threadIdx.x < W && live && lb < a.nblk)
            atomicMax(&a.pmax[((size_t)g * (a.Lc + 1) + pos) * W + r], (unsigned long long)__double_as_longlong(p1));
        const double p2 = block_reduce_rows<W>(pmx2, OpMax(), lds);
        if (threadIdx.x < W && first && !sparse_tail && lb < a.nblk)
            atomicMax(&a.pmax[((size_t)g * (a.Lc + 1) + len) * W + r], (unsigned long long)__double_as_longlong(p2));
    }
    const double t1 = block_reduce_rows<W>(s1, OpAdd(), lds);
    const double t2 = block_reduce_rows<W>(s2, OpAdd(), lds);
    if (threadIdx.x < W && lb < a.nblk) {
        double *bp = a.bpart + ((size_t)(pos & 1) * a.ng + g) * a.nblk8 * W * 2;
        bp[((size_t)lb * W + r) * 2 + 0] = t1;
        bp[((size_t)lb * W + r) * 2 + 1] = t2;
    }
    if (lb == 0 && threadIdx.x < W && live) a.BE[((size_t)g * (a.Lc + 1) + pos) * W + r] = Epos;
}

template <int W>
__global__ void __launch_bounds__(BLOCK) bwd_finish(const DenseArgs a) {
    __shared__ double lds[(BLOCK / 64) * 64];
    bwd_chain<W>(a, blockIdx.x, 0, lds);
}

// node_freq[k] = sum over groups (fixed order) of accg[g][k]
__global__ void __launch_bounds__(BLOCK) freq_reduce(const double *accg, int ng, int N, double *out, int accumulate) {
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= N) return;
    double s = accumulate ? out[k] : 0.0;
    for (int g = 0; g < ng; g++) s += accg[(size_t)g * N + k];
    out[k] = s;
}

// debug: scaled linear table -> natural-log table in the caller's [L][N] layout (W == 1)
// Transition posteriors summed over the positions of every read (PHMMOutput::to_edge_and_init_freqs,
// freq.rs:276-298 over to_trans_and_init_probs, freq.rs:332-389), from the FULL forward and backward
// tables of the chunk.  For edge e = (k -> l) and merged index i = 1..len (F.table_merged(i) = column i-1):
//   to Match (i < len):  t_e e_l(x[i]) B[i+1].m[l] (p_MM F.m[k] + p_IM F.i[k] + p_DM F.d[k]) / P
//   to Del:              t_e           B[i].d[l]   (p_MD F.m[k] + p_ID F.i[k] + p_DD F.d[k]) / P
// (B[len] = b_init: m = d = p_end); the Begin state gives the per-node init frequencies
//   i = 0: init_v (p_MM e_v(x[0]) B[1].m[v] + p_MD B[0].d[v]) / P        (mb = 1, forward.rs:255-266)
//   i >= 1: ib_{i-1} init_v (p_IM e_v(x[i]) B[i+1].m[v] + p_ID B[i].d[v]) / P
// One row of W lanes per edge (blockIdx.x: edges, then nodes for the init part), lanes = reads.
template <int W>
__global__ void __launch_bounds__(BLOCK) edge_freq_kernel(const DenseArgs a, const uint32_t *esrc, const uint32_t *edst,
                                                          const double *trans, const double *init, int E, double *accE,
                                                          double *accI) {
    constexpr int ROWS = BLOCK / W;
    const int g = blockIdx.y;
    const int r = threadIdx.x % W, row = threadIdx.x / W;
    const int item = blockIdx.x * ROWS + row;  // < E: edge; else node item - E
    if (item >= E + a.N) return;
    const bool is_edge = item < E;
    const int k = is_edge ? (int)esrc[item] : 0;
    const int l = is_edge ? (int)edst[item] : item - E;
    const double t = is_edge ? trans[item] : init[l];
    const int len = a.len[g * W + r];
    const LinParams &lp = a.lp;
    const size_t NW = (size_t)a.N * W;
    const double lpf = a.logPf[g * W + r];
    const uint8_t el = a.emis[l];
    double acc = 0.0;
    if (len > 0 && lpf > -INFINITY && t != 0.0) {
        const int *FE = a.FE + (size_t)g * (a.Lc + 1) * W + r;
        const int *BE = a.BE + (size_t)g * (a.Lc + 1) * W + r;
        for (int i = is_edge ? 1 : 0; i <= len; i++) {
            // source side: F.table_merged(i)
            double sm, sd;  // coefficients of the "to Match" and "to Del" terms, in units of 2^fe * exp(lib)
            int fe = 0;
            double lib = 0.0;
            if (is_edge) {
                const size_t ix = ((size_t)g * a.Lc + (i - 1)) * NW + (size_t)k * W + r;
                const double fm = a.Fm[ix], fi = a.Fi[ix], fd = a.Fd[ix];
                sm = lp.p_MM * fm + lp.p_IM * fi + lp.p_DM * fd;
                sd = lp.p_MD * fm + lp.p_ID * fi + lp.p_DD * fd;
                fe = FE[(size_t)(i - 1) * W];
            } else if (i == 0) {
                sm = lp.p_MM;
                sd = lp.p_MD;
            } else {
                // (the InsBegin chain stays in the exponent: exp(logib) alone is 0 from base ~105 on, and 0 times the
                // exp(-ln P) = inf of a read with ln P < -709 would be NaN)
                sm = lp.p_IM;
                sd = lp.p_ID;
                lib = a.logib[i - 1];
            }
            // target side
            double term = 0.0;
            if (i < len) {
                const uint8_t x = a.bases[((size_t)g * a.Lc + i) * W + r];
                const double pe = el == x ? lp.p_match : lp.p_mismatch;
                double bm;
                int be;
                if (i + 1 < len) {
                    bm = a.Bm[((size_t)g * a.bcols + (i + 1)) * NW + (size_t)l * W + r];
                    be = BE[(size_t)(i + 1) * W];
                } else {
                    bm = lp.p_end;
                    be = 0;
                }
                // (a zero factor stays zero whatever the weight: no 0 * inf)
                const double v1 = sm * pe * bm;
                if (v1 != 0.0) term += v1 * exp(lib + (double)(fe + be) * LN2 - lpf);
                const double bd = a.Bd[((size_t)g * a.bcols + i) * NW + (size_t)l * W + r];
                const double v2 = sd * bd;
                if (v2 != 0.0) term += v2 * exp(lib + (double)(fe + BE[(size_t)i * W]) * LN2 - lpf);
            } else {
                const double v3 = sd * lp.p_end;
                if (v3 != 0.0) term += v3 * exp(lib + (double)fe * LN2 - lpf);
            }
            acc += t * term;
        }
    }
    const double tot = lanes_sum<W>(acc);
    if (r == 0) {
        if (is_edge) accE[(size_t)g * E + item] = tot;
        else accI[(size_t)g * a.N + l] = tot;
    }
}

__global__ void __launch_bounds__(BLOCK) to_log_tables(const double *T, const int *E, int Lc, int N, int L,
                                                       double *out) {
    const size_t idx = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= (size_t)L * N) return;
    const int pos = (int)(idx / N);
    const double v = T[idx];
    out[idx] = v > 0.0 ? log(v) + (double)E[pos] * LN2 : -INFINITY;
}

// ------------------------------------------------------------------ host driver

struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
    bool on;
    explicit Timer(bool on_) : on(on_) {
        if (on) {
            HIP_CHECK(hipEventCreate(&a));
            HIP_CHECK(hipEventCreate(&b));
        }
    }
    ~Timer() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
    void start() {
        if (on) HIP_CHECK(hipEventRecord(a, current_stream()));
    }
    double stop() {
        if (!on) return 0.0;
        HIP_CHECK(hipEventRecord(b, current_stream()));
        HIP_CHECK(hipEventSynchronize(b));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        return ms;
    }
};

// one forward / backward column for the read groups [g_off, g_off + g_cnt) on stream s (g_cnt < 0: all, current stream)
template <int W> static void launch_fwd_one(const DenseArgs &a0, int pos, int g_off = 0, int g_cnt = -1, hipStream_t s = nullptr) {
    const bool dma = !knobs().no_dma;
    DenseArgs a = a0;
    a.g_off = g_off;
    if (g_cnt < 0) {
        g_cnt = a.ng;
        s = current_stream();
    }
    if (pos == 0) hipLaunchKernelGGL(fwd_init<W>, dim3(a.nblk8, g_cnt), dim3(BLOCK), 0, s, a);
    else if (W == 64 && dma && a.npt % DMA_DEPTH == 0)
        hipLaunchKernelGGL((fwd_step<W, true>), dim3(a.nblk8, g_cnt), dim3(BLOCK), 0, s, a, pos);
    else
        hipLaunchKernelGGL((fwd_step<W, false>), dim3(a.nblk8, g_cnt), dim3(BLOCK), 0, s, a, pos);
}
template <int W> static void launch_bwd_one(const DenseArgs &a0, int pos, int g_off = 0, int g_cnt = -1, hipStream_t s = nullptr) {
    const bool dma = knobs().bwd_dma;
    DenseArgs a = a0;
    a.g_off = g_off;
    if (g_cnt < 0) {
        g_cnt = a.ng;
        s = current_stream();
    }
    if (W == 64 && dma)
        hipLaunchKernelGGL((bwd_step<W, true>), dim3(a.nblk8, g_cnt), dim3(BLOCK), 0, s, a, pos);
    else
        hipLaunchKernelGGL((bwd_step<W, false>), dim3(a.nblk8, g_cnt), dim3(BLOCK), 0, s, a, pos);
}
// Small graphs (cfg2: N = 1e4): one column of one chunk is a launch of a few hundred blocks that lives ~30 us, most of
// it the dependent-load chain of a wave (column header -> window rebuild -> rows -> reductions), with the machine
// neither full nor streaming.  Read groups are independent from the first forward column to the last backward one, so
// the chunk's groups are dealt to up to 4 streams whose launch sequences overlap: while one is in its latency
// phases another streams.  Large launches (cfg3) fill the chip on their own and stay on one stream.
static int dense_stream_count(const DenseArgs &a) {
    const int forced = knobs().dense_streams;
    int k = forced;
    if (k <= 0) {
        const long blocks = (long)a.nblk8 * a.ng;
        k = blocks >= 4096 ? 1 : (blocks >= 1536 ? 2 : 3);
    }
    return std::max(1, std::min(std::min(k, MAX_WORKERS), a.ng));
}
template <int W>
void launch_chunk(phmm_model *m, const DenseArgs &a, bool do_bwd, CallStats &st, bool timing) {
    hipStream_t s = current_stream();
    dim3 blk(BLOCK);
    Timer tf(timing), tb(timing);
    const int K = dense_stream_count(a);
    hipStream_t ks[MAX_WORKERS] = {s, nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[MAX_WORKERS] = {};
    if (K > 1) {
        HIP_CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        for (int k = 1; k < K; k++) {
            if (!m->pool->wstream[k]) HIP_CHECK(hipStreamCreateWithFlags(&m->pool->wstream[k], hipStreamNonBlocking));
            ks[k] = m->pool->wstream[k];
            HIP_CHECK(hipEventCreateWithFlags(&join[k], hipEventDisableTiming));
        }
    }
    auto g_lo = [&](int k) { return (int)((long)a.ng * k / K); };
    auto fan_out = [&] {
        if (K == 1) return;
        HIP_CHECK(hipEventRecord(fork, s));
        for (int k = 1; k < K; k++) HIP_CHECK(hipStreamWaitEvent(ks[k], fork, 0));
    };
    auto fan_in = [&] {
        for (int k = 1; k < K; k++) {
            HIP_CHECK(hipEventRecord(join[k], ks[k]));
            HIP_CHECK(hipStreamWaitEvent(s, join[k], 0));
        }
    };
    tf.start();
    fan_out();
    for (int pos = 0; pos <= a.Lc; pos++)
        for (int k = 0; k < K; k++) launch_fwd_one<W>(a, pos, g_lo(k), g_lo(k + 1) - g_lo(k), ks[k]);  // (W = 64: the LDS-DMA variant)
    fan_in();
    st.ms[0] += tf.stop();
    st.launches[0] += (uint64_t)a.Lc + 1;
    hipLaunchKernelGGL(fwd_finish<W>, dim3(a.ng, a.eall ? a.Lc : 1), blk, 0, s, a);
    if (do_bwd) {
        tb.start();
        fan_out();
        for (int pos = a.Lc - 1; pos >= 0; pos--)
            for (int k = 0; k < K; k++) launch_bwd_one<W>(a, pos, g_lo(k), g_lo(k + 1) - g_lo(k), ks[k]);
        fan_in();
        st.ms[1] += tb.stop();
        st.launches[1] += (uint64_t)a.Lc;
        hipLaunchKernelGGL(bwd_finish<W>, dim3(a.ng), blk, 0, s, a);
    }
    const hipError_t le = hipGetLastError();
    if (fork) (void)hipEventDestroy(fork);
    for (int k = 1; k < K; k++)
        if (join[k]) (void)hipEventDestroy(join[k]);
    HIP_CHECK(le);
}

template <int W> static void launch_fwd_fin(const DenseArgs &a) {
    hipLaunchKernelGGL(fwd_finish<W>, dim3(a.ng, a.eall ? a.Lc : 1), dim3(BLOCK), 0, current_stream(), a);
}
#define PHMM_W_SWITCH(W, CALL)                                  \
    switch (W) {                                                \
    case 1: CALL(1); break;                                     \
    case 2: CALL(2); break;                                     \
    case 4: CALL(4); break;                                     \
    case 8: CALL(8); break;                                     \
    case 16: CALL(16); break;                                   \
    case 32: CALL(32); break;                                   \
    case 64: CALL(64); break;                                   \
    default: PHMM_THROW(PHMM_EINTERNAL, "bad read-group width"); \
    }
void launch_fwd_step(int W, const DenseArgs &a, int pos) {
#define CALL_(w) launch_fwd_one<w>(a, pos)
    PHMM_W_SWITCH(W, CALL_)
#undef CALL_
}
void launch_bwd_step(int W, const DenseArgs &a, int pos) {
#define CALL_(w) launch_bwd_one<w>(a, pos)
    PHMM_W_SWITCH(W, CALL_)
#undef CALL_
}
template <int W> static void launch_bwd_fin(const DenseArgs &a) {
    hipLaunchKernelGGL(bwd_finish<W>, dim3(a.ng), dim3(BLOCK), 0, current_stream(), a);
}
void launch_bwd_finish(int W, const DenseArgs &a) {
#define CALL_(w) launch_bwd_fin<w>(a)
    PHMM_W_SWITCH(W, CALL_)
#undef CALL_
    HIP_CHECK(hipGetLastError());
}
void launch_fwd_finish(int W, const DenseArgs &a) {
#define CALL_(w) launch_fwd_fin<w>(a)
    PHMM_W_SWITCH(W, CALL_)
#undef CALL_
}

void launch_chunk_w(phmm_model *m, int W, const DenseArgs &a, bool do_bwd, CallStats &st, bool timing) {
    switch (W) {
    case 1: launch_chunk<1>(m, a, do_bwd, st, timing); break;
    case 2: launch_chunk<2>(m, a, do_bwd, st, timing); break;
    case 4: launch_chunk<4>(m, a, do_bwd, st, timing); break;
    case 8: launch_chunk<8>(m, a, do_bwd, st, timing); break;
    case 16: launch_chunk<16>(m, a, do_bwd, st, timing); break;
    case 32: launch_chunk<32>(m, a, do_bwd, st, timing); break;
    case 64: launch_chunk<64>(m, a, do_bwd, st, timing); break;
    default: PHMM_THROW(PHMM_EINTERNAL, "bad read-group width");
    }
}

// largest W in {64,..,1} whose padding waste is <= 1/16 of the lanes; a read set that fits one group takes the
// smallest width that holds it (a plan of 35 deferred reads as 9 groups of 4 ran its dense columns 2.5x slower than as
// one group of 64: padding lanes cost next to nothing, per-lane node work does)
int choose_width(uint64_t R) {
    // (at most 32: the full-width instantiations then only ever run full-size plans, and their rocprof averages
    // are not mixed with the one-group launches of the small plans beside them)
    if (R <= 64) {
        int W = 1;
        while ((uint64_t)W < R && W < 32) W <<= 1;
        return W;
    }
    for (int W = 64; W >= 2; W >>= 1) {
        uint64_t padded = (R + W - 1) / W * W;
        if ((padded - R) * 16 <= padded) return W;
    }
    return 1;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// carve sub-buffers out of one allocation
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *b) : base((char *)b) {}
    template <class T> T *take(size_t n) {
        off = align_up(off, 256);
        T *p = base ? (T *)(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

Plan make_plan(const phmm_model *m, const phmm_reads *reads, int forced_w) {
    std::vector<uint32_t> all(reads->R);
    std::iota(all.begin(), all.end(), 0u);
    Plan p = make_plan_ids(m, reads, all);
    if (forced_w > 0 && forced_w != p.W) {
        // only the single-read debug path forces a width
        p.W = forced_w;
        p.ng_total = (int)((p.order.size() + p.W - 1) / p.W);
        const int rows = BLOCK / p.W;
        p.nblk = (int)((m->N + (int64_t)p.npt * rows - 1) / ((int64_t)p.npt * rows));
        p.nblk8 = (p.nblk + 7) / 8 * 8;
    }
    return p;
}

// plan over a subset of the reads (ids index reads->off)
Plan make_plan_ids(const phmm_model *m, const phmm_reads *reads, const std::vector<uint32_t> &ids) {
    Plan p;
    const uint64_t R = ids.size();
    const int forced_w = 0;
    p.order = ids;
    std::stable_sort(p.order.begin(), p.order.end(), [&](uint32_t x, uint32_t y) {
        return reads->off[x + 1] - reads->off[x] > reads->off[y + 1] - reads->off[y];
    });
    p.W = forced_w > 0 ? forced_w : choose_width(R);
    // (tuning knobs: PHMM_DENSE_W forces the read-group width, PHMM_DENSE_NPT the run length)
    {
        const int w = knobs().dense_w;
        if (w == 1 || w == 2 || w == 4 || w == 8 || w == 16 || w == 32 || w == 64) p.W = w;
    }
    p.ng_total = (int)((R + p.W - 1) / p.W);
    const int rows = BLOCK / p.W;
    // run length (consecutive nodes walked by one row of W lanes): as long as the launch still
    // has >= ~8192 waves (the window fill at a run start costs 12 loads, a node on the run 2),
    // and long enough that a column has <= ~4096 per-block partials
    // (the floor of 8: measured on cfg2 -- N = 1e4, 200 reads -- runs of 4 spend their time rebuilding windows)
    int npt = 8;
    const double lanes_total = (double)m->N * (double)(p.ng_total * p.W);
    while (npt < 64 && lanes_total / (npt * 2) >= 524288.0) npt *= 2;
    while ((int64_t)((m->N + (int64_t)npt * rows - 1) / ((int64_t)npt * rows)) > 4096 && npt < 64) npt *= 2;
    if (knobs().dense_npt > 0) npt = std::max(2, std::min(256, knobs().dense_npt & ~1));
    p.npt = npt;
    p.nblk = (int)((m->N + (int64_t)npt * rows - 1) / ((int64_t)npt * rows));
    p.nblk8 = (p.nblk + 7) / 8 * 8;
    return p;
}

// lay out one chunk (ngc groups, Lc columns); pass null bases to measure
void layout(DenseArgs &a, int W, bool full_b, void *tables, void *misc, size_t &tb, size_t &mb, bool backward_only) {
    const size_t NW = (size_t)a.N * W;
    Carver t(tables), s(misc);
    // backward_only (backward_sparse's dense tail, sparse_bwd.hip): no forward tables, Del kept in the ping-pong
    const size_t fcols = backward_only ? 0 : (size_t)a.Lc;
    a.Fm = t.take<double>((size_t)a.ng * fcols * NW);
    a.Fi = t.take<double>((size_t)a.ng * fcols * NW);
    a.Fd = t.take<double>((size_t)a.ng * fcols * NW);
    a.bcols = full_b ? a.Lc : 2;
    a.Bm = t.take<double>((size_t)a.ng * a.bcols * NW);
    a.Bi = t.take<double>((size_t)a.ng * a.bcols * NW);
    a.Bd = (full_b || backward_only) ? t.take<double>((size_t)a.ng * a.bcols * NW) : nullptr;
    tb = t.off;
    a.bases = s.take<uint8_t>((size_t)a.ng * a.Lc * W);
    a.len = s.take<int>((size_t)a.ng * W);
    a.FE = s.take<int>((size_t)a.ng * (a.Lc + 1) * W);
    a.BE = s.take<int>((size_t)a.ng * (a.Lc + 1) * W);
    a.cmaxF = s.take<unsigned long long>((size_t)a.ng * a.Lc * W);
    a.cmaxB = s.take<unsigned long long>((size_t)a.ng * a.Lc * W);
    a.epart = s.take<double>((size_t)a.ng * (a.eall ? a.Lc : 1) * a.nblk8 * W);
    a.logPf = s.take<double>((size_t)a.ng * W);
    a.logE = s.take<double>(a.eall ? (size_t)a.ng * a.Lc * W : 1);
    a.bpart = s.take<double>((size_t)2 * a.ng * a.nblk8 * W * 2);
    a.logmbB = s.take<double>((size_t)a.ng * (a.Lc + 1) * W);
    a.logibB = s.take<double>((size_t)a.ng * (a.Lc + 1) * W);
    a.accg = s.take<double>((size_t)a.ng * a.N);
    a.logib = s.take<double>((size_t)a.Lc + 1);
    a.tmaxF = s.take<unsigned long long>((size_t)a.ng * a.Lc * W);
    a.pmax = s.take<unsigned long long>((size_t)a.ng * (a.Lc + 1) * W);
    mb = s.off;
}

void fill_model_args(DenseArgs &a, const phmm_model *m) {
    const ModelDev &d = m->dev;
    a.N = (int)m->N;
    a.nodes = d.nodes.as<NodeRec>();
    a.emis = d.emis.as<uint8_t>();
    a.init = d.init.as<double>();
    a.dinit = d.dinit.as<double>();
    a.tdinit = d.tdinit.as<double>();
    a.fc_off = d.fc_off.as<uint32_t>();
    a.fc = d.fc_ent.as<FwdEntry>();
    a.bc_off = d.bc_off.as<uint32_t>();
    a.bc = d.bc_ent.as<BwdEntry>();
    a.lp = m->lin;
    a.fh_off = d.fh_off.as<uint32_t>();
    a.fh = d.fh_ent.as<HopEntry>();
    a.bh_off = d.bh_off.as<uint32_t>();
    a.bh = d.bh_ent.as<HopEntry>();
    // n_max_gaps <= 4: closure entries by hop + partial sums in p_DD (fwd_step / bwd_step); else merged closure entries
    a.hop_mode = m->lin.n_max_gaps + 2 <= CHAIN_HOPS ? 1 : 0;
}

// forward InsBegin chain in the log domain: ib_0 = p_r*(p_MI*1 + p_II*0); ib_i = p_r*p_II*ib_{i-1}
// (fib, forward.rs:541-545 with f_init / fmb, forward.rs:255-266, 531-533)
void host_logib(const phmm_model *m, size_t n, std::vector<double> &out) {
    out.resize(n + 1);
    const phmm_params &p = m->params;
    double ib = p.p_random + p.p_MI;
    for (size_t i = 0; i <= n; i++) {
        out[i] = ib;
        ib = p.p_random + p.p_II + ib;
    }
}

template <int W>
static void launch_edge_freq_w(const DenseArgs &a, const uint32_t *esrc, const uint32_t *edst, const double *trans,
                               const double *init, int E, double *accE, double *accI) {
    constexpr int ROWS = BLOCK / W;
    const unsigned nb = (unsigned)((E + a.N + ROWS - 1) / ROWS);
    hipLaunchKernelGGL(edge_freq_kernel<W>, dim3(nb, a.ng), dim3(BLOCK), 0, current_stream(), a, esrc, edst, trans, init, E,
                       accE, accI);
}
static void launch_edge_freq(int W, const DenseArgs &a, const uint32_t *esrc, const uint32_t *edst, const double *trans,
                             const double *init, int E, double *accE, double *accI) {
    switch (W) {
    case 1: launch_edge_freq_w<1>(a, esrc, edst, trans, init, E, accE, accI); break;
    case 2: launch_edge_freq_w<2>(a, esrc, edst, trans, init, E, accE, accI); break;
    case 4: launch_edge_freq_w<4>(a, esrc, edst, trans, init, E, accE, accI); break;
    case 8: launch_edge_freq_w<8>(a, esrc, edst, trans, init, E, accE, accI); break;
    case 16: launch_edge_freq_w<16>(a, esrc, edst, trans, init, E, accE, accI); break;
    case 32: launch_edge_freq_w<32>(a, esrc, edst, trans, init, E, accE, accI); break;
    case 64: launch_edge_freq_w<64>(a, esrc, edst, trans, init, E, accE, accI); break;
    default: PHMM_THROW(PHMM_EINTERNAL, "bad read-group width");
    }
}

void run_dense_impl(phmm_model *m, const uint8_t *bases, const uint64_t *off, uint64_t R,
                    const Plan &plan, bool full_b, bool eall, bool want_b, bool want_freq,
                    double *out_lf, double *out_lb, double *out_nf, DenseArgs *dbg_args, double *out_ef,
                    double *out_if, std::vector<uint32_t> *flagged_out) {
    hipStream_t s = current_stream();
    CallStats &st = stats();
    st = CallStats();
    const int W = plan.W;
    const size_t NW = (size_t)m->N * W;
    const uint64_t limit = table_budget(*m->pool);

    DenseArgs base{};
    fill_model_args(base, m);
    base.nblk = plan.nblk;
    base.nblk8 = plan.nblk8;
    base.npt = plan.npt;
    base.eall = eall ? 1 : 0;
    base.want_freq = want_freq ? 1 : 0;

    // per-read results gathered on the host in the caller's order
    std::vector<double> lf(R), lb(want_b ? R : 0);
    DevBuf &nf_dev = m->pool->ws_out;
    if (want_freq) {
        nf_dev.reserve(sizeof(double) * m->N);
        HIP_CHECK(hipMemsetAsync(nf_dev.p, 0, sizeof(double) * m->N, s));
    }

    const bool want_edge = out_ef || out_if;
    DevBuf ef_dev, if_dev, accE, accI;
    if (want_edge) {
        ef_dev.reserve(sizeof(double) * std::max<uint32_t>(m->E, 1));
        if_dev.reserve(sizeof(double) * m->N);
        HIP_CHECK(hipMemsetAsync(ef_dev.p, 0, ef_dev.bytes, s));
        HIP_CHECK(hipMemsetAsync(if_dev.p, 0, if_dev.bytes, s));
    }
    DevBuf d_esrc, d_edst;
    if (want_edge && m->E) {
        d_esrc.upload(m->esrc.data(), sizeof(uint32_t) * m->E);
        d_edst.upload(m->edst.data(), sizeof(uint32_t) * m->E);
    }
    int g0 = 0;
    bool first_chunk = true;
    while (g0 < plan.ng_total) {
        // chunk = as many consecutive groups as fit; Lc = longest read of the chunk
        const uint32_t r0 = plan.order[(size_t)g0 * W];
        const int Lc = (int)(off[r0 + 1] - off[r0]);
        const size_t per_group = (size_t)Lc * NW * 24 + (size_t)(full_b ? Lc * 3 : 4) * NW * 8;
        int ngc = (int)std::min<uint64_t>(plan.ng_total - g0, std::max<uint64_t>(1, limit / std::max<size_t>(per_group, 1)));
        DenseArgs a = base;
        a.ng = ngc;
        a.Lc = Lc;
        size_t tb = 0, mb = 0;
        layout(a, W, full_b, nullptr, nullptr, tb, mb);
        m->wset().tables.reserve(tb);
        m->wset().misc.reserve(mb);
        layout(a, W, full_b, m->wset().tables.p, m->wset().misc.p, tb, mb);
        a.tmaxF = nullptr;  // only the adaptive sparse warm-up needs the per-column totals maximum
        HIP_CHECK(hipMemsetAsync(m->wset().misc.p, 0, mb, s));

        // host staging: transposed bases, lengths, logib
        std::vector<uint8_t> hb((size_t)ngc * Lc * W, 0xff);
        std::vector<int> hl((size_t)ngc * W, 0);
        uint64_t cells = 0;
        for (int g = 0; g < ngc; g++)
            for (int r = 0; r < W; r++) {
                const size_t slot = (size_t)(g0 + g) * W + r;
                if (slot >= R) continue;
                const uint32_t rd = plan.order[slot];
                const uint64_t len = off[rd + 1] - off[rd];
                hl[(size_t)g * W + r] = (int)len;
                cells += len * m->N;
                for (uint64_t i = 0; i < len; i++) hb[((size_t)g * Lc + i) * W + r] = bases[off[rd] + i];
            }
        std::vector<double> hib;
        host_logib(m, (size_t)Lc, hib);
        HIP_CHECK(hipMemcpyAsync((void *)a.bases, hb.data(), hb.size(), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync((void *)a.len, hl.data(), hl.size() * sizeof(int), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync((void *)a.logib, hib.data(), hib.size() * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));  // staging vectors may now die; timed region excludes upload

        launch_chunk_w(m, W, a, want_b || want_freq, st, timing_enabled());
        st.cells[0] += cells;
        if (want_b || want_freq) st.cells[1] += cells;

        // ---- certificate of the scaled linear domain (exact_dense.hip): reads whose flushed cells could matter
        std::vector<uint32_t> flagged;  // lanes of this chunk
        {
            const size_t nE = (size_t)ngc * (Lc + 1) * W, nC = (size_t)ngc * Lc * W;
            std::vector<int> hFE(nE), hBE;
            std::vector<unsigned long long> hcF(nC), hcB;
            std::vector<double> hP((size_t)ngc * W);
            HIP_CHECK(hipMemcpyAsync(hFE.data(), a.FE, sizeof(int) * nE, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(hcF.data(), a.cmaxF, sizeof(unsigned long long) * nC, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(hP.data(), a.logPf, sizeof(double) * hP.size(), hipMemcpyDeviceToHost, s));
            bool have_b = want_b || want_freq;
            auto fetch_b = [&] {
                hBE.resize(nE);
                hcB.resize(nC);
                HIP_CHECK(hipMemcpyAsync(hBE.data(), a.BE, sizeof(int) * nE, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipMemcpyAsync(hcB.data(), a.cmaxB, sizeof(unsigned long long) * nC, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
            };
            if (have_b) fetch_b();
            else HIP_CHECK(hipStreamSynchronize(s));
            const double l2pe = std::log2(m->lin.p_end);
            std::vector<int> fe(Lc), be(Lc);
            std::vector<double> mf(Lc), mbx(Lc);
            auto l2 = [](unsigned long long bits) {
                double v;
                std::memcpy(&v, &bits, 8);
                return v > 0.0 ? std::log2(v) : -INFINITY;
            };
            auto check = [&](int gi, bool with_b) {
                const int g = gi / W, r = gi % W, len = hl[gi];
                for (int i = 0; i < len; i++) {
                    fe[i] = hFE[((size_t)g * (Lc + 1) + i) * W + r];
                    mf[i] = l2(hcF[((size_t)g * Lc + i) * W + r]) + fe[i];
                    if (with_b) {
                        be[i] = hBE[((size_t)g * (Lc + 1) + i) * W + r];
                        mbx[i] = l2(hcB[((size_t)g * Lc + i) * W + r]) + be[i];
                    }
                }
                return certify_dense((int)m->N, len, hP[gi] / LN2, fe.data(), mf.data(), with_b ? be.data() : nullptr,
                                     with_b ? mbx.data() : nullptr, l2pe);
            };
            std::vector<uint32_t> doubt;
            for (int gi = 0; gi < ngc * W; gi++)
                if (hl[gi] > 0 && !check(gi, have_b)) doubt.push_back((uint32_t)gi);
            if (!doubt.empty() && !have_b) {
                // forward-only call: the bound with B <= 1 is loose; get the backward maxima and look again
                for (int pos = Lc - 1; pos >= 0; pos--) launch_bwd_step(W, a, pos);
                launch_bwd_finish(W, a);
                fetch_b();
                for (uint32_t gi : doubt)
                    if (!check((int)gi, true)) flagged.push_back(gi);
            } else {
                flagged.swap(doubt);
            }
        }
        if (!flagged.empty()) {
            if (want_freq || want_edge) {
                // their (unreliable) posteriors are in accg: redo the chunk without them
                HIP_CHECK(hipMemsetAsync(m->wset().misc.p, 0, mb, s));
                std::vector<int> hl2 = hl;
                for (uint32_t gi : flagged) hl2[gi] = 0;
                HIP_CHECK(hipMemcpyAsync((void *)a.bases, hb.data(), hb.size(), hipMemcpyHostToDevice, s));
                HIP_CHECK(hipMemcpyAsync((void *)a.len, hl2.data(), hl2.size() * sizeof(int), hipMemcpyHostToDevice, s));
                HIP_CHECK(hipMemcpyAsync((void *)a.logib, hib.data(), hib.size() * sizeof(double), hipMemcpyHostToDevice, s));
                HIP_CHECK(hipStreamSynchronize(s));
                launch_chunk_w(m, W, a, true, st, false);
            }
            trace("dense: reads outside the scaled range");
        }

        if (want_freq)
            hipLaunchKernelGGL(freq_reduce, dim3((m->N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, a.accg, ngc,
                               (int)m->N, nf_dev.as<double>(), first_chunk ? 0 : 1);
        if (want_edge) {
            const int E = (int)m->E;
            accE.reserve(sizeof(double) * (size_t)ngc * std::max(E, 1));
            accI.reserve(sizeof(double) * (size_t)ngc * m->N);
            launch_edge_freq(W, a, d_esrc.as<uint32_t>(), d_edst.as<uint32_t>(), m->dev.trans_lin.as<double>(),
                             m->dev.init.as<double>(), E, accE.as<double>(), accI.as<double>());
            if (E)
                hipLaunchKernelGGL(freq_reduce, dim3((E + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, accE.as<double>(), ngc, E,
                                   ef_dev.as<double>(), first_chunk ? 0 : 1);
            hipLaunchKernelGGL(freq_reduce, dim3((m->N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, accI.as<double>(), ngc,
                               (int)m->N, if_dev.as<double>(), first_chunk ? 0 : 1);
            HIP_CHECK(hipGetLastError());
        }
        // per-read totals back to caller order
        std::vector<double> tlf((size_t)ngc * W), tlb;
        HIP_CHECK(hipMemcpyAsync(tlf.data(), a.logPf, tlf.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        std::vector<double> tmb;
        if (want_b) {
            tmb.resize((size_t)ngc * (Lc + 1) * W);
            HIP_CHECK(hipMemcpyAsync(tmb.data(), a.logmbB, tmb.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        }
        HIP_CHECK(hipStreamSynchronize(s));
        for (int g = 0; g < ngc; g++)
            for (int r = 0; r < W; r++) {
                const size_t slot = (size_t)(g0 + g) * W + r;
                if (slot >= R) continue;
                const uint32_t rd = plan.order[slot];
                lf[rd] = tlf[(size_t)g * W + r];
                if (want_b) lb[rd] = tmb[((size_t)g * (Lc + 1) + 0) * W + r];
            }
        if (!flagged.empty()) {
            // exact log-domain recursion for the flagged reads, a few at a time (their full tables stay in HBM)
            std::vector<uint32_t> ids;
            for (uint32_t gi : flagged) ids.push_back(plan.order[(size_t)g0 * W + gi]);
            if (flagged_out) flagged_out->insert(flagged_out->end(), ids.begin(), ids.end());
            size_t j0 = 0;
            while (j0 < ids.size()) {
                size_t j1 = j0, bytes = 0;
                while (j1 < ids.size()) {
                    const size_t need = (size_t)(off[ids[j1] + 1] - off[ids[j1]]) * m->N * (want_edge ? 48 : 24);
                    if (j1 > j0 && bytes + need > ((size_t)8 << 30)) break;
                    bytes += need;
                    j1++;
                }
                std::vector<uint32_t> part(ids.begin() + j0, ids.begin() + j1);
                std::vector<double> xlf(part.size()), xlb(part.size());
                exact_dense_reads(m, bases, off, part, xlf.data(), xlb.data(), want_freq ? nf_dev.as<double>() : nullptr, nullptr,
                                  want_edge && m->E ? ef_dev.as<double>() : nullptr, want_edge ? if_dev.as<double>() : nullptr);
                for (size_t j = 0; j < part.size(); j++) {
                    lf[part[j]] = xlf[j];
                    if (want_b) lb[part[j]] = xlb[j];
                }
                j0 = j1;
            }
        }
        if (dbg_args) *dbg_args = a;
        g0 += ngc;
        first_chunk = false;
    }
    // outputs: host vectors -> caller (host or device)
    auto put = [&](double *dst, const std::vector<double> &src) {
        if (!dst) return;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeDevice) {
            HIP_CHECK(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(double), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipStreamSynchronize(s));
        } else {
            (void)hipGetLastError();
            std::memcpy(dst, src.data(), src.size() * sizeof(double));
        }
    };
    put(out_lf, lf);
    if (want_b) put(out_lb, lb);
    if (want_freq && out_nf) copy_out(out_nf, nf_dev.p, sizeof(double) * m->N);
    if (out_ef && m->E) copy_out(out_ef, ef_dev.p, sizeof(double) * m->E);
    if (out_if) copy_out(out_if, if_dev.p, sizeof(double) * m->N);
}

// PHMMOutput::to_edge_and_init_freqs summed over the reads (freq.rs:276-298): needs the full backward tables
void run_dense_edges(phmm_model *m, const phmm_reads *reads, double *out_lf, double *out_ef, double *out_if) {
    Plan plan = make_plan(m, reads, 0);
    run_dense_impl(m, reads->bases.data(), reads->off.data(), reads->R, plan, true, false, true, false, out_lf, nullptr,
                   nullptr, nullptr, out_ef, out_if, nullptr);
}

void run_dense(phmm_model *m, const phmm_reads *reads, double *out_lf, double *out_lb, double *out_nf) {
    Plan plan = make_plan(m, reads, 0);
    run_dense_impl(m, reads->bases.data(), reads->off.data(), reads->R, plan, false, false, out_lb != nullptr,
                   out_nf != nullptr, out_lf, out_lb, out_nf, nullptr, nullptr, nullptr, nullptr);
}

void dense_tables(phmm_model *m, const uint8_t *read, uint64_t len, double *f_m, double *f_i, double *f_d,
                  double *f_scal, double *b_m, double *b_i, double *b_d, double *b_scal) {
    hipStream_t s = current_stream();
    phmm_reads one;
    one.R = 1;
    one.total = len;
    one.bases.assign(read, read + len);
    one.off = {0, len};
    one.max_len = len;
    Plan plan = make_plan(m, &one, 1);
    const bool want_b = b_m || b_i || b_d || b_scal;
    DenseArgs a{};
    double lf = 0, lb = 0;
    std::vector<uint32_t> flagged;
    run_dense_impl(m, one.bases.data(), one.off.data(), 1, plan, true, true, want_b, false, &lf, want_b ? &lb : nullptr,
                   nullptr, &a, nullptr, nullptr, &flagged);
    if (!flagged.empty()) {
        // outside the scaled range: the tables of the exact recursion instead (exact_dense.hip)
        ExactTables et{f_m, f_i, f_d, f_scal, b_m, b_i, b_d, b_scal};
        exact_dense_reads(m, one.bases.data(), one.off.data(), flagged, nullptr, nullptr, nullptr, &et);
        return;
    }
    const int L = (int)len, N = (int)m->N;
    const size_t n = (size_t)L * N;
    DevBuf tmp;
    tmp.reserve(n * sizeof(double));
    auto conv = [&](const double *T, const int *E, double *dst) {
        if (!dst) return;
        hipLaunchKernelGGL(to_log_tables, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, T, E, a.Lc, N, L,
                           tmp.as<double>());
        HIP_CHECK(hipMemcpyAsync(dst, tmp.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    };
    conv(a.Fm, a.FE, f_m);
    conv(a.Fi, a.FE, f_i);
    conv(a.Fd, a.FE, f_d);
    if (f_scal) {
        std::vector<double> e((size_t)L), hib;
        HIP_CHECK(hipMemcpy(e.data(), a.logE, sizeof(double) * L, hipMemcpyDeviceToHost));
        host_logib(m, (size_t)L, hib);
        for (int i = 0; i < L; i++) {
            f_scal[3 * i + 0] = -INFINITY;  // fmb: mb = 0 (forward.rs:531-533)
            f_scal[3 * i + 1] = hib[i];
            f_scal[3 * i + 2] = e[i];
        }
    }
    if (want_b) {
        conv(a.Bm, a.BE, b_m);
        conv(a.Bi, a.BE, b_i);
        conv(a.Bd, a.BE, b_d);
        if (b_scal) {
            std::vector<double> mbv((size_t)L + 1), ibv((size_t)L + 1);
            HIP_CHECK(hipMemcpy(mbv.data(), a.logmbB, sizeof(double) * (L + 1), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(ibv.data(), a.logibB, sizeof(double) * (L + 1), hipMemcpyDeviceToHost));
            for (int i = 0; i < L; i++) {
                b_scal[3 * i + 0] = mbv[i];
                b_scal[3 * i + 1] = ibv[i];
                b_scal[3 * i + 2] = -INFINITY;  // be (backward.rs:563-565)
            }
        }
    }
}

}  // namespace phmm
