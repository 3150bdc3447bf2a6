// What one workgroup-to-workgroup signal costs on MI355X (round 3, the k_pt_coop experiment): workgroup 0 and workgroup p play
// ping-pong on two flags with relaxed agent-scope atomics (global_load/store sc1 — what the LLVM memory model prescribes for
// monotonic agent-scope accesses on gfx942/gfx950), 2000 round trips, timed with the 100 MHz real-time counter; once on an idle
// chip and once while every other workgroup streams through HBM.  Block b lands on XCD b mod 8.
//   hipcc --offload-arch=gfx950 -O3 tools/ab/signal_latency.hip -o /tmp/signal_latency && /tmp/signal_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k(unsigned long long *flags, int partner, int rounds, const double *src, double *sink, long n, int load,
                                         unsigned long long *ticks)
{
    const int b = blockIdx.x;
    if (b == 0 || b == partner) {
        if (threadIdx.x == 0) {
            unsigned long long *mine = flags + (b == 0 ? 0 : 64), *other = flags + (b == 0 ? 64 : 0);
            const unsigned long long t0 = wall_clock64();
            for (int r = 1; r <= rounds; ++r) {
                if (b == 0) {
                    __hip_atomic_store(other, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned spins = 0;
                    while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)r && ++spins < (1u << 24)) {}
                } else {
                    unsigned spins = 0;
                    while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)r && ++spins < (1u << 24)) {}
                    __hip_atomic_store(other, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (b == 0) *ticks = wall_clock64() - t0;
            __hip_atomic_store(flags + 128, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // done: the streamers may stop
        }
        return;
    }
    if (!load) return;
    // everybody else: stream until the ping-pong is over (bounded)
    double acc = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (int pass = 0; pass < 4000; ++pass) {
        for (long i = (long)b * blockDim.x + threadIdx.x; i < n; i += stride * 16) acc += src[i];
        if (__hip_atomic_load(flags + 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (acc == 12345.678) sink[0] = acc;
}

int main()
{
    unsigned long long *flags, *ticks;
    double *src, *sink;
    const long n = 1l << 28;                                   // 2 GB of doubles
    CHK(hipMalloc(&flags, 4096)); CHK(hipMalloc(&ticks, 8)); CHK(hipMalloc(&src, n * 8)); CHK(hipMalloc(&sink, 8));
    CHK(hipMemset(src, 0, n * 8));
    const int rounds = 2000;
    for (int load = 0; load < 2; ++load)
        for (int partner : {8, 16, 1, 2, 3, 9, 255}) {
            CHK(hipMemset(flags, 0, 4096));
            hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, flags, partner, rounds, src, sink, n, load, ticks);
            CHK(hipDeviceSynchronize());
            unsigned long long t = 0;
            CHK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
            printf("%s partner block %3d (XCD %d vs 0): %.2f us per round trip (two one-way signals)\n", load ? "under HBM load," : "idle chip,     ", partner,
                   partner % 8, t * 10.0 / 1000.0 / rounds);
        }
    return 0;
}
