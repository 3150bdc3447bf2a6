// stream_probe.hip — what can ANY fused PT sweep reach on this chip?  Times halo-free streaming kernels with the
// sweep's traffic mix (3 reads + 2 writes of 1 GiB-class fp64 arrays) at different access widths / alignments.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o gpurun_out/stream_probe && gpurun_out/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(8))) d2u {   // 16-byte pair that is only 8-byte aligned (packed odd-sized rows)
    double x, y;
    __device__ d2u operator*(double s) const { return {x * s, y * s}; }
    __device__ d2u operator+(d2u o) const { return {x + o.x, y + o.y}; }
    __device__ d2u operator-(d2u o) const { return {x - o.x, y - o.y}; }
};
__device__ inline d2u operator*(double s, d2u v) { return {v.x * s, v.y * s}; }

template <bool NT> __device__ inline double ld(const double* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ inline void st(double* p, double v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// 1 element per lane
template <bool NT>
__global__ __launch_bounds__(256) void k32w8(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ c,
                                              double* __restrict__ d, size_t n, size_t off) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * 256) {
        double x = a[i], y = ld<NT>(b + i), z = ld<NT>(c + i + off);
        double dn = z * 0.99 + 0.1 * (x - y);
        st<NT>(c + i + off, dn);
        st<NT>(d + i, x + 0.1 * dn);
    }
}
// 2 elements per lane (16 B); U = pointers only 8-B aligned (off odd)
template <bool NT, class V>
__global__ __launch_bounds__(256) void k32w16(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ c,
                                               double* __restrict__ d, size_t n, size_t off) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    for (; i + 1 < n; i += (size_t)gridDim.x * 512) {
        V x = *(const V*)(a + i);
        V y = *(const V*)(b + i);
        V z = *(const V*)(c + i + off);
        V dn = z * 0.99 + 0.1 * (x - y);
        V pn = x + 0.1 * dn;
        *(V*)(c + i + off) = dn; *(V*)(d + i) = pn;
    }
}
__global__ __launch_bounds__(256) void kcopy16(const d2* __restrict__ a, d2* __restrict__ b, size_t n2) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i < n2; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void kread16(const d2* __restrict__ a, double* __restrict__ out, size_t n2) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    d2 acc = {0, 0};
    for (; i < n2; i += (size_t)gridDim.x * 256) acc += a[i];
    if (acc.x + acc.y == 12345.678) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void kwrite16(d2* __restrict__ b, size_t n2) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    d2 v = {1.0, 2.0};
    for (; i < n2; i += (size_t)gridDim.x * 256) b[i] = v;
}

template <class F> float timeit(F f, int reps = 10) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    return ts[0];
}

int main() {
    const size_t n = (size_t)512 * 512 * 512;
    double *a, *b, *c, *d;
    CK(hipMalloc(&a, (n + 16) * 8)); CK(hipMalloc(&b, (n + 16) * 8)); CK(hipMalloc(&c, (n + 16) * 8)); CK(hipMalloc(&d, (n + 16) * 8));
    CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8)); CK(hipMemset(c, 0, n * 8)); CK(hipMemset(d, 0, n * 8));
    const double GB5 = 5.0 * n * 8 / 1e9, GB2 = 2.0 * n * 8 / 1e9, GB1 = n * 8 / 1e9;
    for (int grid : {2048, 4096, 8192, 65536}) {
        float t;
        t = timeit([&] { hipLaunchKernelGGL(kcopy16, dim3(grid), dim3(256), 0, 0, (const d2*)a, (d2*)b, n / 2); });
        printf("grid %6d copy16            %.3f ms  %.0f GB/s\n", grid, t, GB2 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL(kread16, dim3(grid), dim3(256), 0, 0, (const d2*)a, d, n / 2); });
        printf("grid %6d read16            %.3f ms  %.0f GB/s\n", grid, t, GB1 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL(kwrite16, dim3(grid), dim3(256), 0, 0, (d2*)b, n / 2); });
        printf("grid %6d write16           %.3f ms  %.0f GB/s\n", grid, t, GB1 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL(k32w8<false>, dim3(grid), dim3(256), 0, 0, a, b, c, d, n, (size_t)0); });
        printf("grid %6d 3r2w  8B          %.3f ms  %.0f GB/s\n", grid, t, GB5 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL(k32w8<true>, dim3(grid), dim3(256), 0, 0, a, b, c, d, n, (size_t)0); });
        printf("grid %6d 3r2w  8B nt       %.3f ms  %.0f GB/s\n", grid, t, GB5 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL(k32w8<false>, dim3(grid), dim3(256), 0, 0, a, b, c, d, n, (size_t)1); });
        printf("grid %6d 3r2w  8B c+1      %.3f ms  %.0f GB/s\n", grid, t, GB5 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k32w16<false, d2>), dim3(grid), dim3(256), 0, 0, a, b, c, d, n, (size_t)0); });
        printf("grid %6d 3r2w 16B          %.3f ms  %.0f GB/s\n", grid, t, GB5 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k32w16<false, d2u>), dim3(grid), dim3(256), 0, 0, a, b, c, d, n, (size_t)0); });
        printf("grid %6d 3r2w 16B (u8 type)%.3f ms  %.0f GB/s\n", grid, t, GB5 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k32w16<false, d2u>), dim3(grid), dim3(256), 0, 0, a, b, c, d, n, (size_t)1); });
        printf("grid %6d 3r2w 16B c+1 (u8) %.3f ms  %.0f GB/s\n", grid, t, GB5 / t * 1e3);
    }
    return 0;
}
