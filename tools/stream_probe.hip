// dev probe: what does the dense kernels' ACCESS PATTERN reach with (almost) no arithmetic?
// Layout as in dense.hip: planes [pos][node][64] f64; a wave walks `npt` consecutive node rows; per row it
// reads RD planes and writes WR planes of 512 B each.  Variants: prefetch depth, 16-byte lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static constexpr int BLOCK = 256, W = 64, ROWS = BLOCK / W;

template <int RD, int WR, int PF>
__global__ void __launch_bounds__(BLOCK, 4) probe8(const double *__restrict__ src, double *__restrict__ dst, int N, int npt,
                                                   int nblk8, size_t plane) {
    const int g = blockIdx.y;
    const int per = nblk8 >> 3;
    const int lb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int r = threadIdx.x % W, row = threadIdx.x / W;
    const int kbase = lb * (npt * ROWS) + row * npt;
    const double *s = src + (size_t)g * RD * plane;
    double *d = dst + (size_t)g * WR * plane;
    double ring[PF][RD];
#pragma unroll
    for (int u = 0; u < PF; u++)
#pragma unroll
        for (int p = 0; p < RD; p++) {
            const int k = kbase + u;
            ring[u][p] = k < N ? s[(size_t)p * plane + (size_t)k * W + r] : 0.0;
        }
    double carry = 0.0;
    for (int j0 = 0; j0 < npt; j0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int k = kbase + j0 + u;
            if (k >= N) continue;
            double v[RD];
#pragma unroll
            for (int p = 0; p < RD; p++) v[p] = ring[u][p];
            if (j0 + u + PF < npt && k + PF < N)
#pragma unroll
                for (int p = 0; p < RD; p++) ring[u][p] = s[(size_t)p * plane + (size_t)(k + PF) * W + r];
            double acc = carry;
#pragma unroll
            for (int p = 0; p < RD; p++) acc = acc * 0.5 + v[p];
            carry = acc * 0.25;
#pragma unroll
            for (int p = 0; p < WR; p++) d[(size_t)p * plane + (size_t)k * W + r] = acc + p;
        }
    }
}

// 16 B per lane: one instruction covers two consecutive node rows of a plane
template <int RD, int WR>
__global__ void __launch_bounds__(BLOCK, 4) probe16(const double2 *__restrict__ src, double2 *__restrict__ dst, int N, int npt,
                                                    int nblk8, size_t plane2) {
    const int g = blockIdx.y;
    const int per = nblk8 >> 3;
    const int lb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int r = threadIdx.x % W, row = threadIdx.x / W;
    const int kbase = lb * (npt * ROWS) + row * npt;
    const double2 *s = src + (size_t)g * RD * plane2;
    double2 *d = dst + (size_t)g * WR * plane2;
    double carry = 0.0;
    for (int j = 0; j < npt; j += 2) {
        const int k = kbase + j;
        if (k + 1 >= N) continue;
        double2 v[RD];
#pragma unroll
        for (int p = 0; p < RD; p++) v[p] = s[(size_t)p * plane2 + (size_t)k * (W / 2) + r];
        double acc = carry;
#pragma unroll
        for (int p = 0; p < RD; p++) acc = acc * 0.5 + v[p].x + v[p].y;
        carry = acc * 0.25;
#pragma unroll
        for (int p = 0; p < WR; p++) d[(size_t)p * plane2 + (size_t)k * (W / 2) + r] = make_double2(acc + p, acc - p);
    }
}

template <class F> static double time_ms(F f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const int N = 131694, NG = 54, npt = 64;
    const int nblk = (N + npt * ROWS - 1) / (npt * ROWS), nblk8 = (nblk + 7) / 8 * 8;
    const size_t plane = (size_t)N * W;  // doubles per plane per group
    constexpr int RD = 5, WR = 3;
    double *src, *dst;
    CK(hipMalloc(&src, sizeof(double) * plane * RD * NG));
    CK(hipMalloc(&dst, sizeof(double) * plane * WR * NG));
    CK(hipMemset(src, 0, sizeof(double) * plane * RD * NG));
    const double cells = (double)N * W * NG;
    dim3 grid(nblk8, NG), blk(BLOCK);
    auto report = [&](const char *name, double ms, int rd, int wr) {
        printf("%-28s %7.3f ms  %6.0f GB/s (rd %d + wr %d planes, %.1f GB)\n", name, ms, cells * 8 * (rd + wr) / ms / 1e6, rd, wr,
               cells * 8 * (rd + wr) / 1e9);
    };
    report("8B/lane rd5 wr3 PF1", time_ms([&] { hipLaunchKernelGGL((probe8<5, 3, 1>), grid, blk, 0, 0, src, dst, N, npt, nblk8, plane); }, 5), 5, 3);
    report("8B/lane rd5 wr3 PF2", time_ms([&] { hipLaunchKernelGGL((probe8<5, 3, 2>), grid, blk, 0, 0, src, dst, N, npt, nblk8, plane); }, 5), 5, 3);
    report("8B/lane rd5 wr3 PF4", time_ms([&] { hipLaunchKernelGGL((probe8<5, 3, 4>), grid, blk, 0, 0, src, dst, N, npt, nblk8, plane); }, 5), 5, 3);
    report("8B/lane rd2 wr3 PF2 (fwd)", time_ms([&] { hipLaunchKernelGGL((probe8<2, 3, 2>), grid, blk, 0, 0, src, dst, N, npt, nblk8, plane); }, 5), 2, 3);
    report("8B/lane rd2 wr3 PF4 (fwd)", time_ms([&] { hipLaunchKernelGGL((probe8<2, 3, 4>), grid, blk, 0, 0, src, dst, N, npt, nblk8, plane); }, 5), 2, 3);
    report("8B/lane rd5 wr0 PF2", time_ms([&] { hipLaunchKernelGGL((probe8<5, 0, 2>), grid, blk, 0, 0, src, dst, N, npt, nblk8, plane); }, 5), 5, 0);
    report("8B/lane rd1 wr3 PF2", time_ms([&] { hipLaunchKernelGGL((probe8<1, 3, 2>), grid, blk, 0, 0, src, dst, N, npt, nblk8, plane); }, 5), 1, 3);
    report("16B/lane rd5 wr3", time_ms([&] { hipLaunchKernelGGL((probe16<5, 3>), grid, blk, 0, 0, (const double2 *)src, (double2 *)dst, N, npt, nblk8, plane / 2); }, 5), 5, 3);
    report("16B/lane rd2 wr3", time_ms([&] { hipLaunchKernelGGL((probe16<2, 3>), grid, blk, 0, 0, (const double2 *)src, (double2 *)dst, N, npt, nblk8, plane / 2); }, 5), 2, 3);
    return 0;
}
