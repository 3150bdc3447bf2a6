/*
 * ns3d_oracle.c — CPU ORACLE for the NavierStokes3D hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the kernel bodies and the two time loops of
 * mattbuergler/NavierStokes3D (reference paths below are relative to /root/reference):
 *     multi.jl = scripts/NavierStokes3D_multi_gpu.jl      gpu.jl = scripts/NavierStokes3D_gpu.jl
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check /
 * time the HIP path against something.  Nothing in navierstokes3d_amd/ may import, link or call it.
 *
 * PARITY UNPINNED: the reference is Julia + ParallelStencil/ImplicitGlobalGrid (un-vendored,
 * unpinned); no Julia exists in the build container, and the reference's single known-answer test
 * (test/test3D.jl:8-32) is stale and structurally unrunnable.  The oracle is
 * therefore pinned by (i) an independent NumPy transcription (oracle/numpy_ref.py) that must agree
 * bit-for-bit, (ii) analytic properties (tests/test_oracle.py) and (iii) the reference's own SOURCE
 * TEXT evaluated mechanically (oracle/jl_eval.py: every kernel and both drivers of the two scripts,
 * executed token by token under ParallelStencil's published macro table and ImplicitGlobalGrid's
 * one-rank formulas; outputs in tests/golden/jl_eval_*.npz), which this file reproduces bit for bit.
 * "Unpinned" remains true by the letter: no vector held by the reference and no run of the reference
 * itself stands behind it — the semantics of the two un-vendored packages are stated, not executed.
 *
 * Arithmetic contract: IEEE-754, one rounding per operation, NO fused multiply-add
 * (compile with -ffp-contract=off), operation order exactly that of the Julia expressions
 * (Julia folds a+b+c left to right, x/dx/dx = (x/dx)/dx, dt/ρ*X = (dt/ρ)*X, 2μ*X = (2*μ)*X and never
 * contracts to FMA) — SURVEY.md Appendix A.
 *
 * Layout: packed column-major, x fastest, exactly the reference's array shapes
 * (multi.jl:343-360):  Pr,C,τxx,τyy,τzz,∇V (nx,ny,nz) · Vx (nx+1,ny,nz) · Vy (nx,ny+1,nz) ·
 * Vz (nx,ny,nz+1) · τxy,τxz,τyz (nx-1,ny-1,nz-1) · dPrdτ,Rp (nx-2,ny-2,nz-2).
 * FiniteDifferences3D macro meanings ([upstream]) are restated inline.
 *
 * Built twice (REAL=double, SUF=f64 and REAL=float, SUF=f32) by oracle/Makefile.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifndef REAL
#define REAL double
#define SUF f64
#endif
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)
#define R(x) ((REAL)(x))

typedef ptrdiff_t idx;
/* 0-based column-major index into an array of extents (sx,sy,*) */
#define IX(i, j, k, sx, sy) ((idx)(i) + (idx)(sx) * ((idx)(j) + (idx)(sy) * (idx)(k)))

/* ---------------------------------------------------------------------------------------------
 * @∇V() = @d_xa(Vx)/dx + @d_ya(Vy)/dy + @d_za(Vz)/dz          (multi.jl:15, gpu.jl:175)
 * ------------------------------------------------------------------------------------------- */
static inline REAL divV_at(const REAL *Vx, const REAL *Vy, const REAL *Vz, int i, int j, int k,
                           int nx, int ny, REAL dx, REAL dy, REAL dz)
{
    REAL dVx = Vx[IX(i + 1, j, k, nx + 1, ny)] - Vx[IX(i, j, k, nx + 1, ny)];
    REAL dVy = Vy[IX(i, j + 1, k, nx, ny + 1)] - Vy[IX(i, j, k, nx, ny + 1)];
    REAL dVz = Vz[IX(i, j, k + 1, nx, ny)] - Vz[IX(i, j, k, nx, ny)];
    return (dVx / dx + dVy / dy) + dVz / dz;
}

/* update_τ!(τxx,τyy,τzz,τxy,τxz,τyz,Vx,Vy,Vz,μ,dx,dy,dz)      multi.jl:36-44, gpu.jl:177-185 */
void FN(ns3d_ref_update_tau)(REAL *txx, REAL *tyy, REAL *tzz, REAL *txy, REAL *txz, REAL *tyz,
                             const REAL *Vx, const REAL *Vy, const REAL *Vz, double mu_, double dx_,
                             double dy_, double dz_, int nx, int ny, int nz)
{
    const REAL mu = R(mu_), dx = R(dx_), dy = R(dy_), dz = R(dz_);
    const REAL two_mu = R(2) * mu;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                /* @all(τxx) = 2μ*(@d_xa(Vx)/dx - @∇V()/3.0)  … multi.jl:37-39 */
                REAL dVx = Vx[IX(i + 1, j, k, nx + 1, ny)] - Vx[IX(i, j, k, nx + 1, ny)];
                REAL dVy = Vy[IX(i, j + 1, k, nx, ny + 1)] - Vy[IX(i, j, k, nx, ny + 1)];
                REAL dVz = Vz[IX(i, j, k + 1, nx, ny)] - Vz[IX(i, j, k, nx, ny)];
                REAL div = (dVx / dx + dVy / dy) + dVz / dz;
                idx c = IX(i, j, k, nx, ny);
                txx[c] = two_mu * (dVx / dx - div / R(3.0));
                tyy[c] = two_mu * (dVy / dy - div / R(3.0));
                tzz[c] = two_mu * (dVz / dz - div / R(3.0));
                /* @all(τxy) = μ*(@d_yi(Vx)/dy + @d_xi(Vy)/dx) … multi.jl:40-42; guarded by the
                 * (nx-1,ny-1,nz-1) extents of the shear arrays                              */
                if (i < nx - 1 && j < ny - 1 && k < nz - 1) {
                    idx s = IX(i, j, k, nx - 1, ny - 1);
                    REAL vx111 = Vx[IX(i + 1, j + 1, k + 1, nx + 1, ny)];
                    REAL vy111 = Vy[IX(i + 1, j + 1, k + 1, nx, ny + 1)];
                    REAL vz111 = Vz[IX(i + 1, j + 1, k + 1, nx, ny)];
                    txy[s] = mu * ((vx111 - Vx[IX(i + 1, j, k + 1, nx + 1, ny)]) / dy +
                                   (vy111 - Vy[IX(i, j + 1, k + 1, nx, ny + 1)]) / dx);
                    txz[s] = mu * ((vx111 - Vx[IX(i + 1, j + 1, k, nx + 1, ny)]) / dz +
                                   (vz111 - Vz[IX(i, j + 1, k + 1, nx, ny)]) / dx);
                    tyz[s] = mu * ((vy111 - Vy[IX(i + 1, j + 1, k, nx, ny + 1)]) / dz +
                                   (vz111 - Vz[IX(i + 1, j, k + 1, nx, ny)]) / dy);
                }
            }
}

/* predict_V!(Vx,Vy,Vz,τxx,τyy,τzz,τxy,τxz,τyz,ρ,g,dt,dx,dy,dz)  multi.jl:50-55, gpu.jl:187-192 */
void FN(ns3d_ref_predict_V)(REAL *Vx, REAL *Vy, REAL *Vz, const REAL *txx, const REAL *tyy,
                            const REAL *tzz, const REAL *txy, const REAL *txz, const REAL *tyz,
                            double rho_, double g_, double dt_, double dx_, double dy_, double dz_,
                            int nx, int ny, int nz)
{
    const REAL rho = R(rho_), g = R(g_), dt = R(dt_), dx = R(dx_), dy = R(dy_), dz = R(dz_);
    const REAL dt_rho = dt / rho;   /* dt/ρ*(…) = (dt/ρ)*(…) */
    const REAL rho_g = rho * g;
    const int sx = nx - 1, sy = ny - 1; /* shear-array extents */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz - 1; ++k)
        for (int j = 0; j < ny - 1; ++j)
            for (int i = 0; i < nx - 1; ++i) {
                /* @inn(Vx): i<nx-1, j<ny-2, k<nz-2 (multi.jl:51) */
                if (j < ny - 2 && k < nz - 2) {
                    idx v = IX(i + 1, j + 1, k + 1, nx + 1, ny);
                    REAL a = (txx[IX(i + 1, j + 1, k + 1, nx, ny)] - txx[IX(i, j + 1, k + 1, nx, ny)]) / dx;
                    REAL b = (txy[IX(i, j + 1, k, sx, sy)] - txy[IX(i, j, k, sx, sy)]) / dy;
                    REAL c = (txz[IX(i, j, k + 1, sx, sy)] - txz[IX(i, j, k, sx, sy)]) / dz;
                    Vx[v] = Vx[v] + dt_rho * ((a + b) + c);
                }
                /* @inn(Vy): i<nx-2, j<ny-1, k<nz-2 (multi.jl:52) */
                if (i < nx - 2 && k < nz - 2) {
                    idx v = IX(i + 1, j + 1, k + 1, nx, ny + 1);
                    REAL a = (tyy[IX(i + 1, j + 1, k + 1, nx, ny)] - tyy[IX(i + 1, j, k + 1, nx, ny)]) / dy;
                    REAL b = (txy[IX(i + 1, j, k, sx, sy)] - txy[IX(i, j, k, sx, sy)]) / dx;
                    REAL c = (tyz[IX(i, j, k + 1, sx, sy)] - tyz[IX(i, j, k, sx, sy)]) / dz;
                    Vy[v] = Vy[v] + dt_rho * ((a + b) + c);
                }
                /* @inn(Vz): i<nx-2, j<ny-2, k<nz-1 (multi.jl:53), with the body force −ρ*g */
                if (i < nx - 2 && j < ny - 2) {
                    idx v = IX(i + 1, j + 1, k + 1, nx, ny);
                    REAL a = (tzz[IX(i + 1, j + 1, k + 1, nx, ny)] - tzz[IX(i + 1, j + 1, k, nx, ny)]) / dz;
                    REAL b = (txz[IX(i + 1, j, k, sx, sy)] - txz[IX(i, j, k, sx, sy)]) / dx;
                    REAL c = (tyz[IX(i, j + 1, k, sx, sy)] - tyz[IX(i, j, k, sx, sy)]) / dy;
                    Vz[v] = Vz[v] + dt_rho * (((a + b) + c) - rho_g);
                }
            }
}

/* update_∇V!(∇V,Vx,Vy,Vz,dx,dy,dz)                             multi.jl:61-64, gpu.jl:194-197 */
void FN(ns3d_ref_update_divV)(REAL *divV, const REAL *Vx, const REAL *Vy, const REAL *Vz,
                              double dx_, double dy_, double dz_, int nx, int ny, int nz)
{
    const REAL dx = R(dx_), dy = R(dy_), dz = R(dz_);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i)
                divV[IX(i, j, k, nx, ny)] = divV_at(Vx, Vy, Vz, i, j, k, nx, ny, dx, dy, dz);
}

/* @d2_xi(Pr)/dx/dx + @d2_yi(Pr)/dy/dy + @d2_zi(Pr)/dz/dz − ρ/dt*@inn(∇V)   (multi.jl:71,89) */
static inline REAL poisson_rhs(const REAL *Pr, const REAL *divV, int i, int j, int k, int nx, int ny,
                               REAL dx, REAL dy, REAL dz, REAL rho_dt)
{
    REAL c = Pr[IX(i + 1, j + 1, k + 1, nx, ny)];
    REAL d2x = (Pr[IX(i + 2, j + 1, k + 1, nx, ny)] - c) - (c - Pr[IX(i, j + 1, k + 1, nx, ny)]);
    REAL d2y = (Pr[IX(i + 1, j + 2, k + 1, nx, ny)] - c) - (c - Pr[IX(i + 1, j, k + 1, nx, ny)]);
    REAL d2z = (Pr[IX(i + 1, j + 1, k + 2, nx, ny)] - c) - (c - Pr[IX(i + 1, j + 1, k, nx, ny)]);
    REAL lap = (d2x / dx / dx + d2y / dy / dy) + d2z / dz / dz;
    return lap - rho_dt * divV[IX(i + 1, j + 1, k + 1, nx, ny)];
}

/* update_dPrdτ!(Pr,dPrdτ,∇V,ρ,dt,dτ,damp,dx,dy,dz)            multi.jl:70-73, gpu.jl:199-202 */
void FN(ns3d_ref_update_dPrdtau)(const REAL *Pr, REAL *dPrdtau, const REAL *divV, double rho_,
                                 double dt_, double dtau_, double damp_, double dx_, double dy_,
                                 double dz_, int nx, int ny, int nz)
{
    const REAL dx = R(dx_), dy = R(dy_), dz = R(dz_), dtau = R(dtau_);
    const REAL rho_dt = R(rho_) / R(dt_);
    const REAL one_m_damp = R(1.0) - R(damp_);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz - 2; ++k)
        for (int j = 0; j < ny - 2; ++j)
            for (int i = 0; i < nx - 2; ++i) {
                idx d = IX(i, j, k, nx - 2, ny - 2);
                dPrdtau[d] = dPrdtau[d] * one_m_damp +
                             dtau * poisson_rhs(Pr, divV, i, j, k, nx, ny, dx, dy, dz, rho_dt);
            }
}

/* update_Pr!(Pr,dPrdτ,dτ)                                     multi.jl:79-82, gpu.jl:204-207 */
void FN(ns3d_ref_update_Pr)(REAL *Pr, const REAL *dPrdtau, double dtau_, int nx, int ny, int nz)
{
    const REAL dtau = R(dtau_);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz - 2; ++k)
        for (int j = 0; j < ny - 2; ++j)
            for (int i = 0; i < nx - 2; ++i) {
                idx p = IX(i + 1, j + 1, k + 1, nx, ny);
                Pr[p] = Pr[p] + dtau * dPrdtau[IX(i, j, k, nx - 2, ny - 2)];
            }
}

/* compute_res!(Rp,Pr,∇V,ρ,dt,dx,dy,dz)                        multi.jl:88-91, gpu.jl:209-212 */
void FN(ns3d_ref_compute_res)(REAL *Rp, const REAL *Pr, const REAL *divV, double rho_, double dt_,
                              double dx_, double dy_, double dz_, int nx, int ny, int nz)
{
    const REAL dx = R(dx_), dy = R(dy_), dz = R(dz_);
    const REAL rho_dt = R(rho_) / R(dt_);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz - 2; ++k)
        for (int j = 0; j < ny - 2; ++j)
            for (int i = 0; i < nx - 2; ++i)
                Rp[IX(i, j, k, nx - 2, ny - 2)] =
                    poisson_rhs(Pr, divV, i, j, k, nx, ny, dx, dy, dz, rho_dt);
}

/* maximum(abs.(A)) — Julia's maximum propagates NaN           multi.jl:466, gpu.jl:132 */
double FN(ns3d_ref_max_abs)(const REAL *A, long n)
{
    double m = 0.0;
    int has_nan = 0;
#pragma omp parallel for schedule(static) reduction(max : m) reduction(| : has_nan)
    for (long q = 0; q < n; ++q) {
        double a = fabs((double)A[q]);
        if (a != a) has_nan = 1;
        else if (a > m) m = a;
    }
    return has_nan ? (double)NAN : m;
}

/* correct_V!(Vx,Vy,Vz,Pr,dt,ρ,dx,dy,dz)                        multi.jl:97-102, gpu.jl:214-219 */
void FN(ns3d_ref_correct_V)(REAL *Vx, REAL *Vy, REAL *Vz, const REAL *Pr, double dt_, double rho_,
                            double dx_, double dy_, double dz_, int nx, int ny, int nz)
{
    const REAL dx = R(dx_), dy = R(dy_), dz = R(dz_);
    const REAL dt_rho = R(dt_) / R(rho_);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz - 1; ++k)
        for (int j = 0; j < ny - 1; ++j)
            for (int i = 0; i < nx - 1; ++i) {
                REAL c = Pr[IX(i + 1, j + 1, k + 1, nx, ny)];
                if (j < ny - 2 && k < nz - 2) { /* @inn(Vx) − dt/ρ*@d_xi(Pr)/dx */
                    idx v = IX(i + 1, j + 1, k + 1, nx + 1, ny);
                    Vx[v] = Vx[v] - (dt_rho * (c - Pr[IX(i, j + 1, k + 1, nx, ny)])) / dx;
                }
                if (i < nx - 2 && k < nz - 2) {
                    idx v = IX(i + 1, j + 1, k + 1, nx, ny + 1);
                    Vy[v] = Vy[v] - (dt_rho * (c - Pr[IX(i + 1, j, k + 1, nx, ny)])) / dy;
                }
                if (i < nx - 2 && j < ny - 2) {
                    idx v = IX(i + 1, j + 1, k + 1, nx, ny);
                    Vz[v] = Vz[v] - (dt_rho * (c - Pr[IX(i + 1, j + 1, k, nx, ny)])) / dz;
                }
            }
}

/* ---- boundary-plane kernels; A has arbitrary extents (sx,sy,sz) --------------------------- */
/* bc_x!(A): A[1]=A[2]; A[end]=A[end-1] along x                multi.jl:108-112, gpu.jl:221-225 */
void FN(ns3d_ref_bc_x)(REAL *A, int sx, int sy, int sz)
{
#pragma omp parallel for schedule(static)
    for (int k = 0; k < sz; ++k)
        for (int j = 0; j < sy; ++j) {
            A[IX(0, j, k, sx, sy)] = A[IX(1, j, k, sx, sy)];
            A[IX(sx - 1, j, k, sx, sy)] = A[IX(sx - 2, j, k, sx, sy)];
        }
}
/* bc_y!(A)                                                     multi.jl:118-122, gpu.jl:227-231 */
void FN(ns3d_ref_bc_y)(REAL *A, int sx, int sy, int sz)
{
#pragma omp parallel for schedule(static)
    for (int k = 0; k < sz; ++k)
        for (int i = 0; i < sx; ++i) {
            A[IX(i, 0, k, sx, sy)] = A[IX(i, 1, k, sx, sy)];
            A[IX(i, sy - 1, k, sx, sy)] = A[IX(i, sy - 2, k, sx, sy)];
        }
}
/* bc_z!(A)                                                     multi.jl:128-132, gpu.jl:233-237 */
void FN(ns3d_ref_bc_z)(REAL *A, int sx, int sy, int sz)
{
#pragma omp parallel for schedule(static)
    for (int j = 0; j < sy; ++j)
        for (int i = 0; i < sx; ++i) {
            A[IX(i, j, 0, sx, sy)] = A[IX(i, j, 1, sx, sy)];
            A[IX(i, j, sz - 1, sx, sy)] = A[IX(i, j, sz - 2, sx, sy)];
        }
}
/* bc_zV!(A): no-slip bed (=0), free-slip lid                   gpu.jl:239-243 */
void FN(ns3d_ref_bc_zV)(REAL *A, int sx, int sy, int sz)
{
#pragma omp parallel for schedule(static)
    for (int j = 0; j < sy; ++j)
        for (int i = 0; i < sx; ++i) {
            A[IX(i, j, 0, sx, sy)] = R(0.0);
            A[IX(i, j, sz - 1, sx, sy)] = A[IX(i, j, sz - 2, sx, sy)];
        }
}
/* bc_xhydstatic!(A,dz,nz,g,ρ): A[1,iy,iz]=ρ*g*(nz-iz+0.5)*dz+100 ; A[end,iy,iz]=ρ*g*(nz-iz+0.5)*dz
 *                                                              gpu.jl:257-261 (iz is 1-based) */
void FN(ns3d_ref_bc_xhydstatic)(REAL *A, double dz_, int nz_arg, double g_, double rho_, int sx,
                                int sy, int sz)
{
    const REAL dz = R(dz_), rho_g = R(rho_) * R(g_);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < sz; ++k)
        for (int j = 0; j < sy; ++j) {
            REAL h = (rho_g * (R(nz_arg - (k + 1)) + R(0.5))) * dz;
            A[IX(0, j, k, sx, sy)] = h + R(100);
            A[IX(sx - 1, j, k, sx, sy)] = h;
        }
}
/* bc_x_Vx!(A,V): A[1,iy,iz]=V                                  multi.jl:138-141 */
void FN(ns3d_ref_bc_x_Vx)(REAL *A, double v, int sx, int sy, int sz)
{
#pragma omp parallel for schedule(static)
    for (int k = 0; k < sz; ++k)
        for (int j = 0; j < sy; ++j) A[IX(0, j, k, sx, sy)] = R(v);
}
/* bc_x_Pr!(A,val): A[end,iy,iz]=val                            multi.jl:147-150 */
void FN(ns3d_ref_bc_x_Pr)(REAL *A, double v, int sx, int sy, int sz)
{
#pragma omp parallel for schedule(static)
    for (int k = 0; k < sz; ++k)
        for (int j = 0; j < sy; ++j) A[IX(sx - 1, j, k, sx, sy)] = R(v);
}
/* X_o .= X                                                     multi.jl:475, gpu.jl:141 */
void FN(ns3d_ref_copy)(REAL *dst, const REAL *src, long n) { memcpy(dst, src, (size_t)n * sizeof(REAL)); }

/* ---- set_cylinder! ------------------------------------------------------------------------ */
static inline int in_ellipse(REAL xq, REAL yq, REAL ox, REAL oy, REAL sinb, REAL cosb, REAL a2,
                             REAL b2, REAL thr)
{
    REAL xr = (xq - ox) * cosb - (yq - oy) * sinb;
    REAL yr = (xq - ox) * sinb + (yq - oy) * cosb;
    return (xr * xr / a2 + yr * yr / b2) < thr;
}
static void set_cylinder_impl(REAL *C, REAL *Vx, REAL *Vy, REAL *Vz, REAL a2, REAL b2, REAL ox,
                              REAL oy, REAL sinb, REAL cosb, int local_form, REAL xco, REAL yco,
                              REAL lx, REAL ly, REAL dx, REAL dy, int nx, int ny, int nz)
{
    /* thread range = element-wise max of the argument sizes = (nx+1,ny+1,nz+1) [upstream] */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz + 1; ++k)
        for (int j = 0; j < ny + 1; ++j)
            for (int i = 0; i < nx + 1; ++i) {
                REAL xc, yc, xv, yv;
                if (!local_form) { /* multi.jl:250-251 (ix-1 == i) */
                    xc = xco + R(i) * dx;
                    yc = yco + R(j) * dy;
                    xv = xc - dx / R(2);
                    yv = yc - dy / R(2);
                } else { /* gpu.jl:337-338 — note yc = yv + dx/2 (sic, App. B2) */
                    xv = R(i) * dx - lx / R(2);
                    yv = R(j) * dy - ly / R(2);
                    xc = xv + dx / R(2);
                    yc = yv + dx / R(2);
                }
                if (i < nx && j < ny && k < nz && /* multi.jl:252-258 */
                    in_ellipse(xc, yc, ox, oy, sinb, cosb, a2, b2, R(1.05)))
                    C[IX(i, j, k, nx, ny)] = R(1.0);
                if (j < ny && k < nz && /* Vx (nx+1,ny,nz), multi.jl:259-265 */
                    in_ellipse(xv, yc, ox, oy, sinb, cosb, a2, b2, R(1.0)))
                    Vx[IX(i, j, k, nx + 1, ny)] = R(0.0);
                if (i < nx && k < nz && /* Vy (nx,ny+1,nz), multi.jl:266-272 */
                    in_ellipse(xc, yv, ox, oy, sinb, cosb, a2, b2, R(1.0)))
                    Vy[IX(i, j, k, nx, ny + 1)] = R(0.0);
                if (i < nx && j < ny && /* Vz (nx,ny,nz+1), multi.jl:273-279 */
                    in_ellipse(xc, yc, ox, oy, sinb, cosb, a2, b2, R(1.0)))
                    Vz[IX(i, j, k, nx, ny)] = R(0.0);
            }
}
/* set_cylinder!(C,Vx,Vy,Vz,a2,b2,ox,oy,sinβ,cosβ,xco_g,yco_g,zco_g,lx,ly,lz,dx,dy,dz)  multi.jl:249-281 */
void FN(ns3d_ref_set_cylinder)(REAL *C, REAL *Vx, REAL *Vy, REAL *Vz, double a2, double b2, double ox,
                               double oy, double sinb, double cosb, double xco_g, double yco_g,
                               double zco_g, double lx, double ly, double lz, double dx, double dy,
                               double dz, int nx, int ny, int nz)
{
    (void)zco_g; (void)lz; (void)dz; /* z never enters the mask: vertical cylinder */
    set_cylinder_impl(C, Vx, Vy, Vz, R(a2), R(b2), R(ox), R(oy), R(sinb), R(cosb), 0, R(xco_g),
                      R(yco_g), R(lx), R(ly), R(dx), R(dy), nx, ny, nz);
}
/* set_cylinder!(C,Vx,Vy,Vz,a2,b2,ox,oy,sinβ,cosβ,lx,ly,lz,dx,dy,dz)                    gpu.jl:336-368 */
void FN(ns3d_ref_set_cylinder_local)(REAL *C, REAL *Vx, REAL *Vy, REAL *Vz, double a2, double b2,
                                     double ox, double oy, double sinb, double cosb, double lx,
                                     double ly, double lz, double dx, double dy, double dz, int nx,
                                     int ny, int nz)
{
    (void)lz; (void)dz;
    set_cylinder_impl(C, Vx, Vy, Vz, R(a2), R(b2), R(ox), R(oy), R(sinb), R(cosb), 1, R(0), R(0),
                      R(lx), R(ly), R(dx), R(dy), nx, ny, nz);
}

/* ---- advect! / backtrack! / lerp ---------------------------------------------------------- */
/* lerp(a,b,t) = b*t + a*(1-t)                                  multi.jl:211, gpu.jl:306 */
static inline REAL lerp_(REAL a, REAL b, REAL t) { return b * t + a * (R(1) - t); }
static inline int clampi(long v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); }
static inline REAL fmod_(REAL a, REAL b) { return sizeof(REAL) == 4 ? (REAL)fmodf((float)a, (float)b) : (REAL)fmod((double)a, (double)b); }
static inline REAL floor_(REAL a) { return sizeof(REAL) == 4 ? (REAL)floorf((float)a) : (REAL)floor((double)a); }

/* backtrack!(A,A_o,vxc,vyc,vzc,dt,dx,dy,dz,ix,iy,iz)  — ix,iy,iz are 1-based as in the reference;
 * A has extents (sx,sy,sz)                                     multi.jl:190-205, gpu.jl:288-304 */
/* koff / szg (NOT in the reference; 0 / sz reproduce it): the array is a WINDOW of a global array of szg planes that starts koff
 * planes into it (ns3d_advect_wide).  The departure index is then computed from the GLOBAL plane number — `Float(iz) − δ` rounds
 * differently for iz = 2 and iz = 12 when δ is below an ulp of either (multi.jl:194 on a rank's local indices is therefore
 * decomposition-dependent even where no clamp bites) — clamped to the global array and translated back. */
static int g_koff = 0, g_nzg = 0, g_nz_local = 0; /* set by ns3d_ref_advect_window around its call of ns3d_ref_advect */
static inline void backtrack(REAL *A, const REAL *A_o, REAL vxc, REAL vyc, REAL vzc, REAL dt, REAL dx,
                             REAL dy, REAL dz, int ix, int iy, int iz, int sx, int sy, int sz)
{
    REAL ddx = dt * vxc / dx, ddy = dt * vyc / dy, ddz = dt * vzc / dz;
    const int koff = g_koff, szg = g_nzg > 0 ? g_nzg + (sz - g_nz_local) : sz;
    int ix1 = clampi((long)floor_(R(ix) - ddx), 1, sx);
    int iy1 = clampi((long)floor_(R(iy) - ddy), 1, sy);
    int iz1 = clampi((long)floor_(R(iz + koff) - ddz), 1, szg);
    int ix2 = clampi(ix1 + 1, 1, sx), iy2 = clampi(iy1 + 1, 1, sy), iz2 = clampi(iz1 + 1, 1, szg);
    iz1 = clampi(iz1 - koff, 1, sz); iz2 = clampi(iz2 - koff, 1, sz);
    /* δ = (δ>0) − (δ % 1): Julia `%` on floats is rem = C fmod (sign of the dividend) */
    REAL wx = (ddx > R(0) ? R(1) : R(0)) - fmod_(ddx, R(1));
    REAL wy = (ddy > R(0) ? R(1) : R(0)) - fmod_(ddy, R(1));
    REAL wz = (ddz > R(0) ? R(1) : R(0)) - fmod_(ddz, R(1));
#define AO(i, j, k) A_o[IX((i)-1, (j)-1, (k)-1, sx, sy)]
    REAL fy1z1 = lerp_(AO(ix1, iy1, iz1), AO(ix2, iy1, iz1), wx);
    REAL fy1z2 = lerp_(AO(ix1, iy1, iz2), AO(ix2, iy1, iz2), wx);
    REAL fy2z1 = lerp_(AO(ix1, iy2, iz1), AO(ix2, iy2, iz1), wx);
    REAL fy2z2 = lerp_(AO(ix1, iy2, iz2), AO(ix2, iy2, iz2), wx);
#undef AO
    REAL fz1 = lerp_(fy1z1, fy2z1, wy);
    REAL fz2 = lerp_(fy1z2, fy2z2, wy);
    A[IX(ix - 1, iy - 1, iz - 1, sx, sy)] = lerp_(fz1, fz2, wz);
}

/* advect!(Vx,Vx_o,Vy,Vy_o,Vz,Vz_o,C,C_o,dt,dx,dy,dz)           multi.jl:217-243, gpu.jl:308-334
 * faithful != 0 reproduces the reference: the third branch back-tracks **Vy** with Vz-located
 * velocities and Vz is never written.  faithful == 0 is the documented "fixed"
 * variant (third branch advects Vz); it is not the reference's behaviour.                      */
void FN(ns3d_ref_advect)(REAL *Vx, const REAL *Vx_o, REAL *Vy, const REAL *Vy_o, REAL *Vz,
                         const REAL *Vz_o, REAL *C, const REAL *C_o, double dt_, double dx_,
                         double dy_, double dz_, int nx, int ny, int nz, int faithful)
{
    const REAL dt = R(dt_), dx = R(dx_), dy = R(dy_), dz = R(dz_);
    g_nz_local = nz;
#define VXO(i, j, k) Vx_o[IX((i)-1, (j)-1, (k)-1, nx + 1, ny)]
#define VYO(i, j, k) Vy_o[IX((i)-1, (j)-1, (k)-1, nx, ny + 1)]
#define VZO(i, j, k) Vz_o[IX((i)-1, (j)-1, (k)-1, nx, ny)]
#pragma omp parallel for schedule(static)
    for (int iz = 1; iz <= nz + 1; ++iz)
        for (int iy = 1; iy <= ny + 1; ++iy)
            for (int ix = 1; ix <= nx + 1; ++ix) {
                REAL vxc, vyc, vzc;
                if (ix > 1 && ix < nx + 1 && iy <= ny && iz <= nz) { /* multi.jl:218-223 */
                    vxc = VXO(ix, iy, iz);
                    vyc = R(0.25) * (((VYO(ix - 1, iy, iz) + VYO(ix - 1, iy + 1, iz)) + VYO(ix, iy, iz)) + VYO(ix, iy + 1, iz));
                    vzc = R(0.25) * (((VZO(ix - 1, iy, iz) + VZO(ix - 1, iy, iz + 1)) + VZO(ix, iy, iz)) + VZO(ix, iy, iz + 1));
                    backtrack(Vx, Vx_o, vxc, vyc, vzc, dt, dx, dy, dz, ix, iy, iz, nx + 1, ny, nz);
                }
                if (iy > 1 && iy < ny + 1 && ix <= nx && iz <= nz) { /* multi.jl:224-229 */
                    vxc = R(0.25) * (((VXO(ix, iy - 1, iz) + VXO(ix + 1, iy - 1, iz)) + VXO(ix, iy, iz)) + VXO(ix + 1, iy, iz));
                    vyc = VYO(ix, iy, iz);
                    vzc = R(0.25) * (((VZO(ix, iy - 1, iz) + VZO(ix, iy - 1, iz + 1)) + VZO(ix, iy, iz)) + VZO(ix, iy, iz + 1));
                    backtrack(Vy, Vy_o, vxc, vyc, vzc, dt, dx, dy, dz, ix, iy, iz, nx, ny + 1, nz);
                }
                if (iz > 1 && iz < nz + 1 && ix <= nx && iy <= ny) { /* multi.jl:230-235 */
                    vxc = R(0.25) * (((VXO(ix, iy, iz - 1) + VXO(ix + 1, iy, iz - 1)) + VXO(ix, iy, iz)) + VXO(ix + 1, iy, iz));
                    vyc = R(0.25) * (((VYO(ix, iy, iz - 1) + VYO(ix, iy + 1, iz - 1)) + VYO(ix, iy, iz)) + VYO(ix, iy + 1, iz));
                    vzc = VZO(ix, iy, iz);
                    if (faithful) /* multi.jl:234 / gpu.jl:325: backtrack!(Vy,Vy_o,…) — sic */
                        backtrack(Vy, Vy_o, vxc, vyc, vzc, dt, dx, dy, dz, ix, iy, iz, nx, ny + 1, nz);
                    else
                        backtrack(Vz, Vz_o, vxc, vyc, vzc, dt, dx, dy, dz, ix, iy, iz, nx, ny, nz + 1);
                }
                if (ix <= nx && iy <= ny && iz <= nz) { /* multi.jl:236-241 */
                    vxc = R(0.5) * (VXO(ix, iy, iz) + VXO(ix + 1, iy, iz));
                    vyc = R(0.5) * (VYO(ix, iy, iz) + VYO(ix, iy + 1, iz));
                    vzc = R(0.5) * (VZO(ix, iy, iz) + VZO(ix, iy, iz + 1));
                    backtrack(C, C_o, vxc, vyc, vzc, dt, dx, dy, dz, ix, iy, iz, nx, ny, nz);
                }
            }
#undef VXO
#undef VYO
#undef VZO
}

/* advect! on a window of a global grid (see backtrack): koff planes into a global array of nz_glob cell planes */
void FN(ns3d_ref_advect_window)(REAL *Vx, const REAL *Vx_o, REAL *Vy, const REAL *Vy_o, REAL *Vz, const REAL *Vz_o, REAL *C,
                                const REAL *C_o, double dt_, double dx_, double dy_, double dz_, int nx, int ny, int nz,
                                int faithful, int koff, int nz_glob)
{
    g_koff = koff; g_nzg = nz_glob;
    FN(ns3d_ref_advect)(Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, C, C_o, dt_, dx_, dy_, dz_, nx, ny, nz, faithful);
    g_koff = 0; g_nzg = 0;
}

/* ---- host sequences ------------------------------------------------------------------------ */
/* set_bc_Pr!: multi.jl:175-184 (bc_kind 0: x,y,z Neumann then outlet Dirichlet if owns_outlet)
 *             gpu.jl:281-286   (bc_kind 1: y,z Neumann then hydrostatic x faces)               */
void FN(ns3d_ref_set_bc_Pr)(REAL *Pr, int bc_kind, int owns_outlet, double outlet_val, double dz,
                            int nz_arg, double g, double rho, int nx, int ny, int nz)
{
    if (bc_kind == 0) {
        FN(ns3d_ref_bc_x)(Pr, nx, ny, nz);
        FN(ns3d_ref_bc_y)(Pr, nx, ny, nz);
        FN(ns3d_ref_bc_z)(Pr, nx, ny, nz);
        if (owns_outlet) FN(ns3d_ref_bc_x_Pr)(Pr, outlet_val, nx, ny, nz);
    } else {
        FN(ns3d_ref_bc_y)(Pr, nx, ny, nz);
        FN(ns3d_ref_bc_z)(Pr, nx, ny, nz);
        FN(ns3d_ref_bc_xhydstatic)(Pr, dz, nz_arg, g, rho, nx, ny, nz);
    }
}

/* set_bc_Vel!: multi.jl:156-169 (kind 0) / gpu.jl:264-279 (kind 1) — halo update excluded */
void FN(ns3d_ref_set_bc_Vel)(REAL *Vx, REAL *Vy, REAL *Vz, int bc_kind, int owns_inlet, double vin,
                             int nx, int ny, int nz)
{
    if (bc_kind == 0) {
        FN(ns3d_ref_bc_x)(Vx, nx + 1, ny, nz);
        FN(ns3d_ref_bc_y)(Vx, nx + 1, ny, nz);
        FN(ns3d_ref_bc_z)(Vx, nx + 1, ny, nz);
        FN(ns3d_ref_bc_x)(Vy, nx, ny + 1, nz);
        FN(ns3d_ref_bc_z)(Vy, nx, ny + 1, nz);
        FN(ns3d_ref_bc_x)(Vz, nx, ny, nz + 1);
        FN(ns3d_ref_bc_y)(Vz, nx, ny, nz + 1);
        if (owns_inlet) FN(ns3d_ref_bc_x_Vx)(Vx, vin, nx + 1, ny, nz);
    } else {
        FN(ns3d_ref_bc_x)(Vx, nx + 1, ny, nz);
        FN(ns3d_ref_bc_y)(Vx, nx + 1, ny, nz);
        FN(ns3d_ref_bc_zV)(Vx, nx + 1, ny, nz);
        FN(ns3d_ref_bc_x)(Vy, nx, ny + 1, nz);
        FN(ns3d_ref_bc_y)(Vy, nx, ny + 1, nz);
        FN(ns3d_ref_bc_zV)(Vy, nx, ny + 1, nz);
        FN(ns3d_ref_bc_x)(Vz, nx, ny, nz + 1);
        FN(ns3d_ref_bc_y)(Vz, nx, ny, nz + 1);
        FN(ns3d_ref_bc_zV)(Vz, nx, ny, nz + 1);
    }
}

/* The pseudo-transient inner loop, multi.jl:458-471 / gpu.jl:126-137, single rank.
 * Runs at most niter iterations; every nchk iterations computes err = max|Rp|*err_mul/err_div
 * (= maximum(abs.(Rp))*ly^2/psc, multi.jl:466: err_mul = ly^2, err_div = psc) and stops if err < eps or !isfinite(err).  eps < 0 disables the
 * early exit (fixed-iteration benchmark mode) but residuals are still recorded.
 * Returns the number of iterations done; err_hist[] receives one value per check
 * (capacity max_checks), *n_checks their count.                                                 */
int FN(ns3d_ref_pt_solve)(REAL *Pr, REAL *dPrdtau, const REAL *divV, REAL *Rp, double rho, double dt,
                          double dtau, double damp, double dx, double dy, double dz, int nx, int ny,
                          int nz, int bc_kind, int owns_outlet, double outlet_val, double g,
                          double eps, int niter, int nchk, double err_mul, double err_div, double *err_hist,
                          int max_checks, int *n_checks)
{
    int checks = 0, iter;
    for (iter = 1; iter <= niter; ++iter) {
        FN(ns3d_ref_update_dPrdtau)(Pr, dPrdtau, divV, rho, dt, dtau, damp, dx, dy, dz, nx, ny, nz);
        FN(ns3d_ref_update_Pr)(Pr, dPrdtau, dtau, nx, ny, nz);
        FN(ns3d_ref_set_bc_Pr)(Pr, bc_kind, owns_outlet, outlet_val, dz, nz, g, rho, nx, ny, nz);
        if (nchk > 0 && iter % nchk == 0) {
            FN(ns3d_ref_compute_res)(Rp, Pr, divV, rho, dt, dx, dy, dz, nx, ny, nz);
            double err = FN(ns3d_ref_max_abs)(Rp, (long)(nx - 2) * (ny - 2) * (nz - 2)) * err_mul / err_div;
            if (err_hist && checks < max_checks) err_hist[checks] = err;
            ++checks;
            if (eps >= 0 && (err < eps || !isfinite(err))) break;
        }
    }
    if (iter > niter) iter = niter; /* Julia's loop variable after a full loop */
    if (n_checks) *n_checks = checks;
    return iter;
}
