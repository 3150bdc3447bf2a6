/* A plain-C host of libns3d.so (no Python, no C++): what a compiled caller of include/ns3d.h looks like.  TEST INFRASTRUCTURE.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c_host/ns3d_c_host.c \
 *       -L navierstokes3d_amd -lns3d -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,… -o ns3d_c_host
 *
 * Runs the inner loop multi.jl:459-463 twice on the same inputs — kernel by kernel through the reference-signature entry
 * points (update_dPrdτ!, update_Pr!, set_bc_Pr!) and as ONE ns3d_pt_iterate call (temporally blocked passes inside the
 * library) — and compares the two results bit for bit; checks the error path (a call with a bad argument returns
 * NS3D_ERR_ARG with a message and leaves the context usable); and runs the multi-GPU half of the header (two virtual z-slab
 * ranks, ns3d_pt_solve_slab) against ns3d_pt_solve on the global grid.  Exit code 0 = "C HOST OK"; 3 = no usable GPU (message from
 * ns3d_last_error on stderr: the library has no CPU path); anything else = failure. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "ns3d.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != NS3D_OK) {                                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ns3d_last_error());                \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)
#define HIPCHECK(call)                                                                       \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));                     \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

static double lcg(unsigned long long *s)
{
    *s = *s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)((*s >> 11) & 0xFFFFF) / 1048576.0 - 0.5;
}

int main(int argc, char **argv)
{
    const int nx = 70, ny = 21, nz = 18, iters = argc > 1 ? atoi(argv[1]) : 11;
    const size_t n = (size_t)nx * ny * nz, ni = (size_t)(nx - 2) * (ny - 2) * (nz - 2);
    ns3d_ctx *ctx = ns3d_create(0, NS3D_STRICT);
    if (!ctx) {
        fprintf(stderr, "ns3d_create failed: %s\n", ns3d_last_error());
        return 3;
    }
    double *hP = malloc(n * sizeof(double)), *hD = malloc(ni * sizeof(double)), *hR = malloc(n * sizeof(double));
    double *outA = malloc(n * sizeof(double)), *outB = malloc(n * sizeof(double)), *dA = malloc(ni * sizeof(double)),
           *dB = malloc(ni * sizeof(double));
    unsigned long long seed = 2024;
    for (size_t q = 0; q < n; ++q) { hP[q] = lcg(&seed); hR[q] = 1e-3 * lcg(&seed); }
    for (size_t q = 0; q < ni; ++q) hD[q] = 1e-2 * lcg(&seed);
    double *P1, *D1, *P2, *D2, *R;
    HIPCHECK(hipMalloc((void **)&P1, n * sizeof(double)));  HIPCHECK(hipMalloc((void **)&P2, n * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&D1, ni * sizeof(double))); HIPCHECK(hipMalloc((void **)&D2, ni * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&R, n * sizeof(double)));
    HIPCHECK(hipMemcpy(P1, hP, n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(P2, hP, n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(D1, hD, ni * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(D2, hD, ni * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(R, hR, n * sizeof(double), hipMemcpyHostToDevice));
    const double rho = 1000.0, dt = 0.013, dtau = 0.009, damp = 2.0 / nx, dx = 1.0 / nx, dy = 0.6 / ny, dz = 0.7 / nz;
    /* (1) kernel by kernel, the reference's call sequence */
    for (int it = 0; it < iters; ++it) {
        CHECK(ns3d_update_dPrdtau_f64(ctx, P1, D1, R, rho, dt, dtau, damp, dx, dy, dz, nx, ny, nz));   /* multi.jl:459 */
        CHECK(ns3d_update_Pr_f64(ctx, P1, D1, dtau, nx, ny, nz));                                      /* :461 */
        CHECK(ns3d_set_bc_Pr_f64(ctx, P1, NS3D_BC_MULTI, 1, 0.25, dz, nz, 0.0, rho, nx, ny, nz));      /* :463 */
    }
    /* (2) the fused path */
    ns3d_pt_params p;
    memset(&p, 0, sizeof p);
    p.rho = rho; p.dt = dt; p.dtau = dtau; p.damp = damp; p.dx = dx; p.dy = dy; p.dz = dz;
    p.nx = nx; p.ny = ny; p.nz = nz; p.bc_kind = NS3D_BC_MULTI; p.owns_outlet = 1; p.outlet_val = 0.25;
    CHECK(ns3d_set_pt_depth(ctx, 3));                       /* three iterations per pass where the count allows */
    CHECK(ns3d_pt_iterate_f64(ctx, P2, D2, R, &p, iters));
    CHECK(ns3d_sync(ctx));
    HIPCHECK(hipMemcpy(outA, P1, n * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(outB, P2, n * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(dA, D1, ni * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(dB, D2, ni * sizeof(double), hipMemcpyDeviceToHost));
    if (memcmp(outA, outB, n * sizeof(double)) || memcmp(dA, dB, ni * sizeof(double))) {
        fprintf(stderr, "fused and kernel-by-kernel results differ\n");
        return 2;
    }
    double sum = 0.0;
    int moved = 0;
    for (size_t q = 0; q < n; ++q) { sum += outA[q]; moved |= outA[q] != hP[q]; }
    if (!moved || sum != sum) { fprintf(stderr, "the loop did not run (sum %g)\n", sum); return 2; }
    /* (3) error path: bad grid → status + message, context still usable */
    int rc = ns3d_update_Pr_f64(ctx, P1, D1, dtau, 1, ny, nz);
    if (rc != NS3D_ERR_ARG || !strlen(ns3d_last_error())) { fprintf(stderr, "bad-argument call returned %d\n", rc); return 2; }
    CHECK(ns3d_update_Pr_f64(ctx, P1, D1, dtau, nx, ny, nz));
    CHECK(ns3d_sync(ctx));
    /* (4) the multi-GPU half of the header from C: two z-slab ranks of nz_l planes (virtual ranks on device 0) run the inner
     * loop with a residual check every 5 iterations through ns3d_pt_solve_slab; the single-device ns3d_pt_solve of the global
     * grid (nz planes) must give the same iteration count, error history and fields */
    {
        const int nz_l = (nz - 2) / 2 + 2;                 /* nz = 2·(nz_l−2)+2 needs an odd number of inner planes… */
        if (2 * (nz_l - 2) + 2 != nz) { fprintf(stderr, "nz = %d does not split into two slabs\n", nz); return 2; }
        const int devs[2] = {0, 0};
        ns3d_mgpu *m = ns3d_mgpu_create(2, devs, nx, ny, nz_l, NS3D_STRICT);
        if (!m) { fprintf(stderr, "ns3d_mgpu_create failed: %s\n", ns3d_last_error()); return 1; }
        const size_t pl = (size_t)nx * ny, ipl = (size_t)(nx - 2) * (ny - 2);
        double *Ps[2], *Ds[2], *Rs[2];
        for (int r = 0; r < 2; ++r) {
            const size_t lo = (size_t)r * (nz_l - 2);
            HIPCHECK(hipMalloc((void **)&Ps[r], pl * nz_l * sizeof(double)));
            HIPCHECK(hipMalloc((void **)&Rs[r], pl * nz_l * sizeof(double)));
            HIPCHECK(hipMalloc((void **)&Ds[r], ipl * (nz_l - 2) * sizeof(double)));
            HIPCHECK(hipMemcpy(Ps[r], hP + pl * lo, pl * nz_l * sizeof(double), hipMemcpyHostToDevice));
            HIPCHECK(hipMemcpy(Rs[r], hR + pl * lo, pl * nz_l * sizeof(double), hipMemcpyHostToDevice));
            HIPCHECK(hipMemcpy(Ds[r], hD + ipl * lo, ipl * (nz_l - 2) * sizeof(double), hipMemcpyHostToDevice));
        }
        HIPCHECK(hipMemcpy(P2, hP, n * sizeof(double), hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(D2, hD, ni * sizeof(double), hipMemcpyHostToDevice));
        double errs_g[16], errs_s[16];
        int it_g = 0, it_s = 0, nc_g = 0, nc_s = 0;
        CHECK(ns3d_pt_solve_f64(ctx, P2, D2, R, &p, -1.0, iters, 5, 0.36, 1000.0, &it_g, errs_g, 16, &nc_g));
        ns3d_pt_params pl_ = p;
        pl_.nz = nz_l;
        CHECK(ns3d_pt_solve_slab_f64(m, Ps, Ds, (const double *const *)Rs, &pl_, -1.0, iters, 5, 0.36, 1000.0, &it_s, errs_s, 16, &nc_s));
        CHECK(ns3d_mgpu_sync(m));
        CHECK(ns3d_sync(ctx));
        if (it_g != it_s || nc_g != nc_s || memcmp(errs_g, errs_s, (size_t)(nc_g < 16 ? nc_g : 16) * sizeof(double))) {
            fprintf(stderr, "slab solve: %d iterations / %d checks against %d / %d\n", it_s, nc_s, it_g, nc_g);
            return 2;
        }
        HIPCHECK(hipMemcpy(outA, P2, n * sizeof(double), hipMemcpyDeviceToHost));
        for (int r = 0; r < 2; ++r) {
            const size_t lo = (size_t)r * (nz_l - 2);
            HIPCHECK(hipMemcpy(outB, Ps[r], pl * nz_l * sizeof(double), hipMemcpyDeviceToHost));
            if (memcmp(outB, outA + pl * lo, pl * nz_l * sizeof(double))) { fprintf(stderr, "slab rank %d differs from the global solve\n", r); return 2; }
            hipFree(Ps[r]); hipFree(Ds[r]); hipFree(Rs[r]);
        }
        ns3d_mgpu_destroy(m);
    }
    printf("C HOST OK: %d iterations, passes of %d, checksum %.17g, library version %d\n", iters, ns3d_last_pt_depth(ctx), sum,
           ns3d_version());
    hipFree(P1); hipFree(P2); hipFree(D1); hipFree(D2); hipFree(R);
    free(hP); free(hD); free(hR); free(outA); free(outB); free(dA); free(dB);
    ns3d_destroy(ctx);
    return 0;
}
