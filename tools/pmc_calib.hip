// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access pattern of the
// dense PHMM kernels: every lane of a wave moves ONE f64 (8 B), 64 lanes = 512 contiguous bytes.
// MI355X_MICROARCH.md ("HBM") calibrates the counters for 16 B/lane streams only and asks for a
// known-byte-count run in the kernel's own pattern before trusting an absolute value.  Each kernel
// below moves exactly BYTES bytes through a buffer far larger than the 256 MiB Infinity Cache.
//
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- tools/pmc_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out -- tools/pmc_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e = (x);                                                     \
        if (e != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));              \
            exit(1);                                                            \
        }                                                                       \
    } while (0)

static constexpr size_t BYTES = 2ull << 30;  // per kernel

__global__ void __launch_bounds__(256) calib_read8(const double *__restrict__ src, size_t n, double *sink) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc == 12345.678) sink[0] = acc;  // never true for the zero-filled buffer; keeps the loads alive
}
__global__ void __launch_bounds__(256) calib_write8(double *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = 1.0;
}
__global__ void __launch_bounds__(256) calib_read16(const double2 *__restrict__ src, size_t n, double *sink) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double2 v = src[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678) sink[0] = acc;
}
__global__ void __launch_bounds__(256) calib_write16(double2 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = make_double2(1.0, 2.0);
}
// the dense kernels' actual shape: three 8 B/lane streams read, three written, per "node" row of 64 lanes
__global__ void __launch_bounds__(256) calib_copy3x8(const double *__restrict__ a, const double *__restrict__ b,
                                                     const double *__restrict__ c, double *__restrict__ x,
                                                     double *__restrict__ y, double *__restrict__ z, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double u = a[i], v = b[i], w = c[i];
        x[i] = u + v;
        y[i] = v + w;
        z[i] = w + u;
    }
}

int main() {
    double *buf = nullptr, *buf2 = nullptr, *sink = nullptr;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMalloc(&buf2, BYTES));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(buf, 0, BYTES));
    CK(hipMemset(buf2, 0, BYTES));
    const unsigned grid = 256 * 16;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_read8, dim3(grid), dim3(256), 0, 0, buf, BYTES / 8, sink);
        hipLaunchKernelGGL(calib_write8, dim3(grid), dim3(256), 0, 0, buf, BYTES / 8);
        hipLaunchKernelGGL(calib_read16, dim3(grid), dim3(256), 0, 0, (const double2 *)buf, BYTES / 16, sink);
        hipLaunchKernelGGL(calib_write16, dim3(grid), dim3(256), 0, 0, (double2 *)buf, BYTES / 16);
        const size_t n3 = BYTES / 8 / 3;
        hipLaunchKernelGGL(calib_copy3x8, dim3(grid), dim3(256), 0, 0, buf, buf + n3, buf + 2 * n3, buf2, buf2 + n3,
                           buf2 + 2 * n3, n3);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
    }
    printf("{\"bytes_per_kernel\": %zu, \"copy3x8_bytes_read\": %zu, \"copy3x8_bytes_written\": %zu}\n", BYTES,
           (BYTES / 8 / 3) * 24, (BYTES / 8 / 3) * 24);
    CK(hipFree(buf));
    CK(hipFree(buf2));
    CK(hipFree(sink));
    return 0;
}
