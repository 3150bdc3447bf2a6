// ns3d_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the NavierStokes3D hot path.
//
// Compiled four times by navierstokes3d_amd/build.py:
//   -DNS3D_MODE_STRICT -ffp-contract=off   → namespace ns3d_strict : the reference's operation order, IEEE
//                                            divisions, no FMA  (bit-identical to oracle/ns3d_oracle.c)
//   … -DNS3D_EXACT_RECIP                   → namespace ns3d_strictx: same bits, x/d by the correctly rounded
//                                            divisor-known-in-advance sequence (guarded; plain divisions otherwise)
//   … -DNS3D_POW2_RECIP                    → namespace ns3d_strictp: same bits when dx, dy, dz are powers of two
//                                            (x/d ≡ x·(1/d) exactly), chosen by the host for such grids (512³ with lx = 1)
//   -DNS3D_MODE_FAST   -ffp-contract=fast  → namespace ns3d_fast   : reciprocal constants + FMA
//
// Reference kernel bodies restated here: scripts/NavierStokes3D_multi_gpu.jl:15-281 ("multi.jl"),
// scripts/NavierStokes3D_gpu.jl:175-368 ("gpu.jl"); macro meanings per SURVEY.md Appendix A.
// Layout: packed column-major, x fastest — a wave64 spans 64 consecutive x (512 B of fp64 per row).
// All of these are HBM-bound 7-point-class stencils: no MFMA anywhere.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "ns3d_launch.h"

#if defined(NS3D_MODE_FAST)
#define NS3D_NS ns3d_fast
#define NS3D_FASTMATH 1
#elif defined(NS3D_MODE_STRICT) && defined(NS3D_EXACT_RECIP)
#define NS3D_NS ns3d_strictx
#define NS3D_FASTMATH 0
#elif defined(NS3D_MODE_STRICT) && defined(NS3D_POW2_RECIP)
#define NS3D_NS ns3d_strictp
#define NS3D_FASTMATH 0
#elif defined(NS3D_MODE_STRICT)
#define NS3D_NS ns3d_strict
#define NS3D_FASTMATH 0
#else
#error "compile with -DNS3D_MODE_STRICT or -DNS3D_MODE_FAST"
#endif

namespace NS3D_NS {

typedef long long idx_t;
#ifndef NS3D_STEP_UNROLL
#define NS3D_STEP_UNROLL 0      // 0: chosen per tile shape (k_pt_sweepN); 1, 2, 4: forced, for A/B builds
#endif
#ifndef NS3D_SHAPE24_F64
#define NS3D_SHAPE24_F64 0      // A/B: the 1024-thread 64×32 shape for fp64 as well (it spills: 128 registers per lane)
#endif
#define IX3(i, j, k, sx, sy) ((idx_t)(i) + (idx_t)(sx) * ((idx_t)(j) + (idx_t)(sy) * (idx_t)(k)))

// Grid spacings.  STRICT divides (x/dx, x/dx/dx) exactly like the Julia expressions; FAST multiplies by
// reciprocals computed once on the host in double precision.
template <class T>
struct Geo {
    T dx, dy, dz;
    T rdx, rdy, rdz;    // 1/dx …
    T rdx2, rdy2, rdz2; // 1/dx² …
};
template <class T>
static Geo<T> make_geo(double dx, double dy, double dz)
{
    Geo<T> g;
    g.dx = (T)dx; g.dy = (T)dy; g.dz = (T)dz;
    g.rdx = (T)1 / g.dx; g.rdy = (T)1 / g.dy; g.rdz = (T)1 / g.dz;   // RN(1/d) in the element type
    g.rdx2 = (T)(1.0 / (dx * dx)); g.rdy2 = (T)(1.0 / (dy * dy)); g.rdz2 = (T)(1.0 / (dz * dz));
    return g;
}
// ---- correctly rounded division by a divisor known in advance -------------------------------------------
// STRICT mode must return RN(x/d) bit for bit, but the generic IEEE fp64 division costs ≈14 dependent VALU
// instructions and the PT stencil does six of them per cell.  With r = RN(1/d) precomputed (Markstein 1990; Brisebarre,
// Muller & Raina, IEEE TC 2004: "the advanced computation of 1/y allows performing correctly rounded division in one
// multiplication plus two FMACs"):
//        q = RN(x·r);   e = x − q·d  (exact, one FMA);   q' = RN(q + e·r)  =  RN(x/d)
// valid while no intermediate over/underflows, i.e. for 2^-900 < |q| < 2^900 with 2^-100 < d < 2^100 (checked on the
// host, which also refuses divisors whose significand is all ones); everything else — zeros (sign preserved), huge,
// tiny, Inf, NaN — takes the plain division, behind two branches so that a wave whose only outliers are ZEROS (fields at rest:
// the cylinder case starts with Vy = Vz = 0 and a uniform Vx) does not run it: the predictor of the 255×153×153 case was three
// times slower per plane than the power-of-two build before (round 4).  `ns3d_selftest_exact_div` compares the two on the
// GPU bit for bit.
template <class T> struct DivLim;
template <> struct DivLim<double> { static constexpr double lo = 0x1p-900, hi = 0x1p900; };
template <> struct DivLim<float> { static constexpr float lo = 0x1p-100f, hi = 0x1p100f; };
template <class T>
__device__ __forceinline__ T div_by_known(T x, T d, T r)
{
    const T q = x * r;
    const T e = __builtin_fma(-q, d, x);
    T q1 = __builtin_fma(e, r, q);
    const T aq = __builtin_fabs(q);
    if (__builtin_expect(!(aq > DivLim<T>::lo && aq < DivLim<T>::hi), 0)) {
        q1 = x;                                 // ±0 / d = ±0: a wave whose only outliers are zeros skips the division
        if (x != (T)0) q1 = x / d;
    }
    return q1;
}
__device__ __forceinline__ float div_by_known(float x, float d, float r)
{
    const float q = x * r;
    const float e = __builtin_fmaf(-q, d, x);
    float q1 = __builtin_fmaf(e, r, q);
    const float aq = __builtin_fabsf(q);
    if (__builtin_expect(!(aq > DivLim<float>::lo && aq < DivLim<float>::hi), 0)) {
        q1 = x;
        if (x != 0.0f) q1 = x / d;
    }
    return q1;
}

#if NS3D_FASTMATH
#define DIV_X(v) ((v)*g.rdx)
#define DIV_Y(v) ((v)*g.rdy)
#define DIV_Z(v) ((v)*g.rdz)
#define DIV_XX(v) ((v)*g.rdx2)
#define DIV_YY(v) ((v)*g.rdy2)
#define DIV_ZZ(v) ((v)*g.rdz2)
#define DIV_3(v) ((v) * (T)(1.0 / 3.0))
#elif defined(NS3D_POW2_RECIP)
// Every spacing is a power of two 2^-30 … 1 (checked on the host): r = 1/d = 2^k, k ≥ 0, is exact, so x·r and x/d are the SAME
// real number and round identically — always, overflow included, and scaling UP never rounds (subnormal x too).  Hence
// x/d/d = (x·r)·r = x·r² with ONE multiplication: if x·r is finite it is exact and x·r² is the one rounding the reference's
// second division makes; if x·r overflows, so does x·r².  No guard, no fall-back.  (Fusing the products into the sums with
// FMAs would be exact too EXCEPT where a product overflows and the sum does not — not taken.)
#define DIV_X(v) ((v)*g.rdx)
#define DIV_Y(v) ((v)*g.rdy)
#define DIV_Z(v) ((v)*g.rdz)
#define DIV_XX(v) ((v)*g.rdx2)
#define DIV_YY(v) ((v)*g.rdy2)
#define DIV_ZZ(v) ((v)*g.rdz2)
#define DIV_3(v) div_by_known((v), (T)3, (T)1 / (T)3)     /* RN(v/3): 3 is an admissible divisor (test_divide_by_three_is_exact) */
#elif defined(NS3D_EXACT_RECIP)
#define DIV_X(v) div_by_known((v), g.dx, g.rdx)
#define DIV_Y(v) div_by_known((v), g.dy, g.rdy)
#define DIV_Z(v) div_by_known((v), g.dz, g.rdz)
#define DIV_XX(v) div_by_known(div_by_known((v), g.dx, g.rdx), g.dx, g.rdx)
#define DIV_YY(v) div_by_known(div_by_known((v), g.dy, g.rdy), g.dy, g.rdy)
#define DIV_ZZ(v) div_by_known(div_by_known((v), g.dz, g.rdz), g.dz, g.rdz)
#define DIV_3(v) div_by_known((v), (T)3, (T)1 / (T)3)
#else
#define DIV_X(v) ((v) / g.dx)
#define DIV_Y(v) ((v) / g.dy)
#define DIV_Z(v) ((v) / g.dz)
#define DIV_XX(v) ((v) / g.dx / g.dx)
#define DIV_YY(v) ((v) / g.dy / g.dy)
#define DIV_ZZ(v) ((v) / g.dz / g.dz)
#define DIV_3(v) ((v) / (T)3.0)
#endif

static inline dim3 grid3(int X, int Y, int Z, dim3 b)
{
    return dim3((unsigned)((X + (int)b.x - 1) / (int)b.x), (unsigned)((Y + (int)b.y - 1) / (int)b.y),
                (unsigned)((Z + (int)b.z - 1) / (int)b.z));
}
#define BLK3 dim3(64, 4, 1)
#define TID3                                                                                                \
    const int i = blockIdx.x * blockDim.x + threadIdx.x;                                                    \
    const int j = blockIdx.y * blockDim.y + threadIdx.y;                                                    \
    const int k = blockIdx.z * blockDim.z + threadIdx.z;

// ---------------------------------------------------------------------------------------------------------
// update_τ!   multi.jl:36-44 / gpu.jl:177-185.   One thread per cell (i<nx, j<ny, k<nz); the shear
// statements are guarded by the (nx-1,ny-1,nz-1) extents of τxy,τxz,τyz.
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_update_tau(T *__restrict__ txx, T *__restrict__ tyy, T *__restrict__ tzz,
                                                    T *__restrict__ txy, T *__restrict__ txz, T *__restrict__ tyz,
                                                    const T *__restrict__ Vx, const T *__restrict__ Vy,
                                                    const T *__restrict__ Vz, T mu, Geo<T> g, int nx, int ny, int nz)
{
    TID3
    if (i >= nx || j >= ny || k >= nz) return;
    const T two_mu = (T)2 * mu;
    const T dVx = Vx[IX3(i + 1, j, k, nx + 1, ny)] - Vx[IX3(i, j, k, nx + 1, ny)];
    const T dVy = Vy[IX3(i, j + 1, k, nx, ny + 1)] - Vy[IX3(i, j, k, nx, ny + 1)];
    const T dVz = Vz[IX3(i, j, k + 1, nx, ny)] - Vz[IX3(i, j, k, nx, ny)];
    const T div = (DIV_X(dVx) + DIV_Y(dVy)) + DIV_Z(dVz); // @∇V()  multi.jl:15
    const idx_t c = IX3(i, j, k, nx, ny);
    txx[c] = two_mu * (DIV_X(dVx) - DIV_3(div));
    tyy[c] = two_mu * (DIV_Y(dVy) - DIV_3(div));
    tzz[c] = two_mu * (DIV_Z(dVz) - DIV_3(div));
    if (i < nx - 1 && j < ny - 1 && k < nz - 1) {
        const idx_t s = IX3(i, j, k, nx - 1, ny - 1);
        const T vx111 = Vx[IX3(i + 1, j + 1, k + 1, nx + 1, ny)];
        const T vy111 = Vy[IX3(i + 1, j + 1, k + 1, nx, ny + 1)];
        const T vz111 = Vz[IX3(i + 1, j + 1, k + 1, nx, ny)];
        txy[s] = mu * (DIV_Y(vx111 - Vx[IX3(i + 1, j, k + 1, nx + 1, ny)]) +
                       DIV_X(vy111 - Vy[IX3(i, j + 1, k + 1, nx, ny + 1)]));
        txz[s] = mu * (DIV_Z(vx111 - Vx[IX3(i + 1, j + 1, k, nx + 1, ny)]) +
                       DIV_X(vz111 - Vz[IX3(i, j + 1, k + 1, nx, ny)]));
        tyz[s] = mu * (DIV_Z(vy111 - Vy[IX3(i + 1, j + 1, k, nx, ny + 1)]) +
                       DIV_Y(vz111 - Vz[IX3(i + 1, j, k + 1, nx, ny)]));
    }
}
template <class T>
hipError_t update_tau(hipStream_t s, T *txx, T *tyy, T *tzz, T *txy, T *txz, T *tyz, const T *Vx, const T *Vy,
                      const T *Vz, double mu, double dx, double dy, double dz, int nx, int ny, int nz)
{
    hipLaunchKernelGGL(k_update_tau<T>, grid3(nx, ny, nz, BLK3), BLK3, 0, s, txx, tyy, tzz, txy, txz, tyz, Vx, Vy,
                       Vz, (T)mu, make_geo<T>(dx, dy, dz), nx, ny, nz);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// predict_V!  multi.jl:50-55 / gpu.jl:187-192.  Thread (i,j,k) updates Vx/Vy/Vz[i+1,j+1,k+1] where inner.
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_predict_V(T *__restrict__ Vx, T *__restrict__ Vy, T *__restrict__ Vz,
                                                   const T *__restrict__ txx, const T *__restrict__ tyy,
                                                   const T *__restrict__ tzz, const T *__restrict__ txy,
                                                   const T *__restrict__ txz, const T *__restrict__ tyz, T dt_rho,
                                                   T rho_g, Geo<T> g, int nx, int ny, int nz)
{
    TID3
    if (i >= nx - 1 || j >= ny - 1 || k >= nz - 1) return;
    const int sx = nx - 1, sy = ny - 1;
    if (j < ny - 2 && k < nz - 2) {
        const idx_t v = IX3(i + 1, j + 1, k + 1, nx + 1, ny);
        const T a = DIV_X(txx[IX3(i + 1, j + 1, k + 1, nx, ny)] - txx[IX3(i, j + 1, k + 1, nx, ny)]);
        const T b = DIV_Y(txy[IX3(i, j + 1, k, sx, sy)] - txy[IX3(i, j, k, sx, sy)]);
        const T c = DIV_Z(txz[IX3(i, j, k + 1, sx, sy)] - txz[IX3(i, j, k, sx, sy)]);
        Vx[v] = Vx[v] + dt_rho * ((a + b) + c);
    }
    if (i < nx - 2 && k < nz - 2) {
        const idx_t v = IX3(i + 1, j + 1, k + 1, nx, ny + 1);
        const T a = DIV_Y(tyy[IX3(i + 1, j + 1, k + 1, nx, ny)] - tyy[IX3(i + 1, j, k + 1, nx, ny)]);
        const T b = DIV_X(txy[IX3(i + 1, j, k, sx, sy)] - txy[IX3(i, j, k, sx, sy)]);
        const T c = DIV_Z(tyz[IX3(i, j, k + 1, sx, sy)] - tyz[IX3(i, j, k, sx, sy)]);
        Vy[v] = Vy[v] + dt_rho * ((a + b) + c);
    }
    if (i < nx - 2 && j < ny - 2) {
        const idx_t v = IX3(i + 1, j + 1, k + 1, nx, ny);
        const T a = DIV_Z(tzz[IX3(i + 1, j + 1, k + 1, nx, ny)] - tzz[IX3(i + 1, j + 1, k, nx, ny)]);
        const T b = DIV_X(txz[IX3(i + 1, j, k, sx, sy)] - txz[IX3(i, j, k, sx, sy)]);
        const T c = DIV_Y(tyz[IX3(i, j + 1, k, sx, sy)] - tyz[IX3(i, j, k, sx, sy)]);
        Vz[v] = Vz[v] + dt_rho * (((a + b) + c) - rho_g);
    }
}
template <class T>
hipError_t predict_V(hipStream_t s, T *Vx, T *Vy, T *Vz, const T *txx, const T *tyy, const T *tzz, const T *txy,
                     const T *txz, const T *tyz, double rho, double gg, double dt, double dx, double dy, double dz,
                     int nx, int ny, int nz)
{
    const T dt_rho = (T)dt / (T)rho, rho_g = (T)rho * (T)gg;
    hipLaunchKernelGGL(k_predict_V<T>, grid3(nx - 1, ny - 1, nz - 1, BLK3), BLK3, 0, s, Vx, Vy, Vz, txx, tyy, tzz,
                       txy, txz, tyz, dt_rho, rho_g, make_geo<T>(dx, dy, dz), nx, ny, nz);
    return hipGetLastError();
}

// Planes per workgroup for the z-marching window kernels (k_predict_fused, k_advect_win2: one workgroup per CU at a time, three
// planes of lead-in per chunk).  Large grids keep the fixed chunk; where that leaves fewer than four rounds of workgroups (255×153×153:
// 40 columns × 3 chunks on 256 CUs) the chunk count is the one with the least rounds × (planes + lead-in).
static int device_cus();
static int window_kz(int ncols, int nz, int dflt)
{
    const long cus = device_cus();
    if ((long)ncols * ((nz + dflt - 1) / dflt) >= 4 * cus) return dflt;
    long best = -1;
    int best_kz = dflt;
    for (int c = 1; c <= max(1, nz / 4); ++c) {
        const int kz = (nz + c - 1) / c;
        const long chunks = (nz + kz - 1) / kz, rounds = ((long)ncols * chunks + cus - 1) / cus, cost = rounds * (kz + 3);
        if (best < 0 || cost < best) { best = cost; best_kz = kz; }
    }
    return best_kz;
}

// ---------------------------------------------------------------------------------------------------------
// update_τ! + predict_V! in one pass (round 3): k_predict_fused.  The six stress arrays are temporaries of the
// predictor (multi.jl:449-451: written by update_τ!, read by predict_V!, nothing else), 72 + 96 B per cell in two kernels
// (3.0 + 3.1 ms at 512³).  Here a 1024-thread workgroup marches 64×16 columns in z with a ring of three xy-planes (+ the one
// being fetched) of Vx, Vy, Vz in LDS, evaluates the stresses a cell's three faces need from the window — the same expressions
// on the same operands as k_update_tau — and writes the predicted velocities into buffers of their own (the stencil on V now
// has radius two: not in place; entries predict_V! leaves alone are written through, so the outputs are complete and the
// caller swaps names).  What sits at the same (x, y) one plane lower — τzz, two τxz, two τyz — is carried in registers from the
// previous step.  Per cell and step: 3 divergences, 5 normal and 7 shear stresses instead of 1 + 3 + 3 (the stresses are
// recomputed where faces share them), 24 + 24 B of HBM traffic instead of 168.  ix, iy, iz are the reference's 1-based indices.
// τ is not stored: a caller that wants the stress arrays runs update_τ!.
// ---------------------------------------------------------------------------------------------------------
template <class T>
struct PredWin {
    static constexpr int TX = 64, TY = 16, WX = TX + 2, WY = TY + 2, NSLOT = 4, PLANE = WX * WY;
    typedef const T __attribute__((address_space(3))) *lds_ptr;
    lds_ptr L;           // [3 arrays][NSLOT][PLANE]
    int x0, y0;          // 1-based first column/row of the tile; the window starts one below
    __device__ __forceinline__ T get(int a, int i, int j, int k) const    // plane k lives in slot k mod 4
    {
        return L[(a * NSLOT + (k & (NSLOT - 1))) * PLANE + (j - (y0 - 1)) * WX + (i - (x0 - 1))];
    }
};
template <class T>
__global__ __launch_bounds__(1024) void k_predict_fused(T *__restrict__ Vxn, T *__restrict__ Vyn, T *__restrict__ Vzn,
                                                        const T *__restrict__ Vx, const T *__restrict__ Vy,
                                                        const T *__restrict__ Vz, T mu, T dt_rho, T rho_g, Geo<T> g, int nx,
                                                        int ny, int nz, int kz)
{
    typedef PredWin<T> W;
    extern __shared__ __align__(16) unsigned char predict_lds_raw[];
    T *L = reinterpret_cast<T *>(predict_lds_raw);
    const int tid = threadIdx.y * W::TX + threadIdx.x;
    const int x0 = blockIdx.x * W::TX + 1, y0 = blockIdx.y * W::TY + 1;
    const int zb = blockIdx.z * kz + 1, ze = min(zb + kz, nz + 1);          // planes iz ∈ [zb, ze), iz ≤ nz
    const int ix = x0 + threadIdx.x, iy = y0 + threadIdx.y;
    const T *const src[3] = {Vx, Vy, Vz};
    const int sxs[3] = {nx + 1, nx, nx}, sys[3] = {ny, ny + 1, ny}, szs[3] = {nz, nz, nz + 1};
    // window positions this thread fills (two per array and plane: 1 188 positions, 1 024 threads); indices outside an array
    // are clamped into it — such values only reach results that are not stored
    unsigned fbase[3][2];
    idx_t fplane[3];
    int q[2];
    bool has[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        q[h] = tid + h * 1024;
        has[h] = q[h] < W::PLANE;
        const int qq = has[h] ? q[h] : 0;
        const int gi = x0 - 1 + qq % W::WX, gj = y0 - 1 + qq / W::WX;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int ci = min(max(gi, 1), sxs[a]), cj = min(max(gj, 1), sys[a]);
            fbase[a][h] = (unsigned)(ci - 1) + (unsigned)sxs[a] * (unsigned)(cj - 1);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) fplane[a] = (idx_t)sxs[a] * sys[a];
    auto fetch = [&](int plane, T (&v)[3][2]) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const T *__restrict__ pl = src[a] + fplane[a] * (min(max(plane, 1), szs[a]) - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) v[a][h] = has[h] ? pl[fbase[a][h]] : (T)0;
        }
    };
    auto publish = [&](int plane, const T (&v)[3][2]) {
        const int slot = plane & (W::NSLOT - 1);            // plane ≥ 0
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (has[h]) L[(a * W::NSLOT + slot) * W::PLANE + q[h]] = v[a][h];
    };
    {
        T v[3][2];
        for (int pl = zb - 1; pl <= zb + 1; ++pl) { fetch(pl, v); publish(pl, v); }
    }
    __syncthreads();
    const W w{(typename W::lds_ptr)L, x0, y0};
#define VX(i_, j_, k_) w.get(0, (i_), (j_), (k_))
#define VY(i_, j_, k_) w.get(1, (i_), (j_), (k_))
#define VZ(i_, j_, k_) w.get(2, (i_), (j_), (k_))
    const T two_mu = (T)2 * mu;
    // the stresses of multi.jl:37-43 at 1-based entry (p, q, r) of their arrays
    struct Normal { T xx, yy, zz; };
    auto normal_at = [&](int p, int q_, int r) {
        const T dVx = VX(p + 1, q_, r) - VX(p, q_, r), dVy = VY(p, q_ + 1, r) - VY(p, q_, r), dVz = VZ(p, q_, r + 1) - VZ(p, q_, r);
        const T div = (DIV_X(dVx) + DIV_Y(dVy)) + DIV_Z(dVz); // @∇V()  multi.jl:15
        Normal n;
        n.xx = two_mu * (DIV_X(dVx) - DIV_3(div));
        n.yy = two_mu * (DIV_Y(dVy) - DIV_3(div));
        n.zz = two_mu * (DIV_Z(dVz) - DIV_3(div));
        return n;
    };
    auto txy_at = [&](int p, int q_, int r) {
        return mu * (DIV_Y(VX(p + 1, q_ + 1, r + 1) - VX(p + 1, q_, r + 1)) + DIV_X(VY(p + 1, q_ + 1, r + 1) - VY(p, q_ + 1, r + 1)));
    };
    auto txz_at = [&](int p, int q_, int r) {
        return mu * (DIV_Z(VX(p + 1, q_ + 1, r + 1) - VX(p + 1, q_ + 1, r)) + DIV_X(VZ(p + 1, q_ + 1, r + 1) - VZ(p, q_ + 1, r + 1)));
    };
    auto tyz_at = [&](int p, int q_, int r) {
        return mu * (DIV_Z(VY(p + 1, q_ + 1, r + 1) - VY(p + 1, q_ + 1, r)) + DIV_Y(VZ(p + 1, q_ + 1, r + 1) - VZ(p + 1, q_, r + 1)));
    };
    const bool in = ix <= nx && iy <= ny;
    const int cix = min(ix, nx), ciy = min(iy, ny);
    const unsigned oVx = (unsigned)(cix - 1) + (unsigned)(nx + 1) * (unsigned)(ciy - 1), oVy = (unsigned)(cix - 1) + (unsigned)nx * (unsigned)(ciy - 1);
    const unsigned oVz = oVy;
    const idx_t pVx = (idx_t)(nx + 1) * ny, pVy = (idx_t)nx * (ny + 1), pVz = (idx_t)nx * ny;
    // what the first plane of this chunk needs from the plane below it (planes zb−1 and zb are in the window)
    T tzz_lo = normal_at(ix, iy, zb - 1).zz;
    T txz_a_lo = txz_at(ix - 1, iy - 1, zb - 1), txz_b_lo = txz_at(ix, iy - 1, zb - 1);
    T tyz_a_lo = tyz_at(ix - 1, iy - 1, zb - 1), tyz_b_lo = tyz_at(ix - 1, iy, zb - 1);
    for (int iz = zb; iz < ze; ++iz) {
        T nxt[3][2];
        fetch(iz + 2, nxt);                                 // in flight behind this plane's arithmetic
        if (in) {
            const Normal n0 = normal_at(ix, iy, iz);
            const T txx_w = normal_at(ix - 1, iy, iz).xx, tyy_s = normal_at(ix, iy - 1, iz).yy;
            const T txy_a = txy_at(ix - 1, iy, iz - 1), txy_b = txy_at(ix - 1, iy - 1, iz - 1), txy_c = txy_at(ix, iy - 1, iz - 1);
            const T txz_a = txz_at(ix - 1, iy - 1, iz), txz_b = txz_at(ix, iy - 1, iz);
            const T tyz_a = tyz_at(ix - 1, iy - 1, iz), tyz_b = tyz_at(ix - 1, iy, iz);
            T *__restrict__ const Vxp = Vxn + pVx * (iz - 1), *__restrict__ const Vyp = Vyn + pVy * (iz - 1), *__restrict__ const Vzp = Vzn + pVz * (iz - 1);
            const T vx = VX(ix, iy, iz), vy = VY(ix, iy, iz), vz = VZ(ix, iy, iz);
            T ox = vx, oy = vy, oz = vz;
            if (ix >= 2 && iy >= 2 && iy <= ny - 1 && iz >= 2 && iz <= nz - 1) {         // @inn(Vx)  multi.jl:51
                const T a = DIV_X(n0.xx - txx_w), b = DIV_Y(txy_a - txy_b), c = DIV_Z(txz_a - txz_a_lo);
                ox = vx + dt_rho * ((a + b) + c);
            }
            if (ix >= 2 && ix <= nx - 1 && iy >= 2 && iz >= 2 && iz <= nz - 1) {         // @inn(Vy)  multi.jl:52
                const T a = DIV_Y(n0.yy - tyy_s), b = DIV_X(txy_c - txy_b), c = DIV_Z(tyz_a - tyz_a_lo);
                oy = vy + dt_rho * ((a + b) + c);
            }
            if (ix >= 2 && ix <= nx - 1 && iy >= 2 && iy <= ny - 1 && iz >= 2) {         // @inn(Vz)  multi.jl:53 (iz ≤ nz here)
                const T a = DIV_Z(n0.zz - tzz_lo), b = DIV_X(txz_b_lo - txz_a_lo), c = DIV_Y(tyz_b_lo - tyz_a_lo);
                oz = vz + dt_rho * (((a + b) + c) - rho_g);
            }
            Vxp[oVx] = ox; Vyp[oVy] = oy; Vzp[oVz] = oz;
            // the far faces of the staggered arrays: never predicted, written through
            if (ix == nx) Vxp[oVx + 1] = VX(ix + 1, iy, iz);
            if (iy == ny) Vyp[oVy + nx] = VY(ix, iy + 1, iz);
            if (iz == nz) (Vzp + pVz)[oVz] = VZ(ix, iy, iz + 1);
            tzz_lo = n0.zz; txz_a_lo = txz_a; txz_b_lo = txz_b; tyz_a_lo = tyz_a; tyz_b_lo = tyz_b;
        }
        publish(iz + 2, nxt);
        __syncthreads();
    }
#undef VX
#undef VY
#undef VZ
}
template <class T>
hipError_t predict_fused(hipStream_t s, T *Vxn, T *Vyn, T *Vzn, const T *Vx, const T *Vy, const T *Vz, double mu, double rho,
                         double gg, double dt, double dx, double dy, double dz, int nx, int ny, int nz)
{
    typedef PredWin<T> W;
    const size_t lds = (size_t)3 * W::NSLOT * W::PLANE * sizeof(T);
    hipError_t ea = hipFuncSetAttribute((const void *)k_predict_fused<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;
    const T dt_rho = (T)dt / (T)rho, rho_g = (T)rho * (T)gg;
    static const int kz_env = std::getenv("NS3D_PREDICT_KZ") ? std::atoi(std::getenv("NS3D_PREDICT_KZ")) : 0;   // A/B
    const int kz = kz_env > 0 ? kz_env : window_kz(((nx + W::TX - 1) / W::TX) * ((ny + W::TY - 1) / W::TY), nz, nz >= 128 ? 64 : 32);
    const dim3 blk(W::TX, W::TY, 1);
    const dim3 grd((unsigned)((nx + W::TX - 1) / W::TX), (unsigned)((ny + W::TY - 1) / W::TY), (unsigned)((nz + kz - 1) / kz));
    hipLaunchKernelGGL(k_predict_fused<T>, grd, blk, lds, s, Vxn, Vyn, Vzn, Vx, Vy, Vz, (T)mu, dt_rho, rho_g, make_geo<T>(dx, dy, dz), nx,
                       ny, nz, kz);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// update_∇V!  multi.jl:61-64 / gpu.jl:194-197
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_update_divV(T *__restrict__ divV, const T *__restrict__ Vx,
                                                     const T *__restrict__ Vy, const T *__restrict__ Vz, Geo<T> g,
                                                     int nx, int ny, int nz)
{
    TID3
    if (i >= nx || j >= ny || k >= nz) return;
    const T dVx = Vx[IX3(i + 1, j, k, nx + 1, ny)] - Vx[IX3(i, j, k, nx + 1, ny)];
    const T dVy = Vy[IX3(i, j + 1, k, nx, ny + 1)] - Vy[IX3(i, j, k, nx, ny + 1)];
    const T dVz = Vz[IX3(i, j, k + 1, nx, ny)] - Vz[IX3(i, j, k, nx, ny)];
    divV[IX3(i, j, k, nx, ny)] = (DIV_X(dVx) + DIV_Y(dVy)) + DIV_Z(dVz);
}
template <class T>
hipError_t update_divV(hipStream_t s, T *divV, const T *Vx, const T *Vy, const T *Vz, double dx, double dy,
                       double dz, int nx, int ny, int nz)
{
    hipLaunchKernelGGL(k_update_divV<T>, grid3(nx, ny, nz, BLK3), BLK3, 0, s, divV, Vx, Vy, Vz,
                       make_geo<T>(dx, dy, dz), nx, ny, nz);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// The Poisson right-hand side shared by update_dPrdτ! and compute_res!  (multi.jl:71,89):
//   @d2_xi(Pr)/dx/dx + @d2_yi(Pr)/dy/dy + @d2_zi(Pr)/dz/dz − ρ/dt*@inn(∇V)
// ---------------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T poisson_rhs(T c, T w, T e, T s, T n, T b, T t, T dv, T rho_dt, const Geo<T> &g)
{
    const T d2x = (e - c) - (c - w);
    const T d2y = (n - c) - (c - s);
    const T d2z = (t - c) - (c - b);
    const T lap = (DIV_XX(d2x) + DIV_YY(d2y)) + DIV_ZZ(d2z);
    return lap - rho_dt * dv;
}

// Two rows of an fp32 thread at once (k_pt_sweepN with two rows per thread, builds whose divisions are multiplications): the same
// operations in the same order on both lanes of v_pk_add_f32 / v_pk_mul_f32 — IEEE per lane, so the bits are those of poisson_rhs.
#if NS3D_FASTMATH || defined(NS3D_POW2_RECIP)
#define NS3D_VEC2 1
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (templated on the geometry type and fed through casts only so that the fp64 instantiations of the kernels, which never reach it, parse)
template <class A, class B> __device__ __forceinline__ f32x2 mk2(A a, B b) { return f32x2{(float)a, (float)b}; }
template <class G>
__device__ __forceinline__ f32x2 poisson_rhs_v2(f32x2 c, f32x2 w, f32x2 e, f32x2 s, f32x2 n, f32x2 b, f32x2 t, f32x2 dv, float rho_dt, const G &g)
{
    const f32x2 d2x = (e - c) - (c - w);
    const f32x2 d2y = (n - c) - (c - s);
    const f32x2 d2z = (t - c) - (c - b);
    const f32x2 lap = (d2x * (float)g.rdx2 + d2y * (float)g.rdy2) + d2z * (float)g.rdz2;
    return lap - rho_dt * dv;
}
#else
#define NS3D_VEC2 0
#endif
// Hot-kernel form: the same value as poisson_rhs, evaluated without branches.  In the exact-reciprocal STRICT build
// every x/d/d is two divisor-known-in-advance divisions in straight-line code; `ok` is cleared for the lanes whose
// dividend is outside the range in which that sequence is proven exact (|x| ∉ (2^-700, 2^700), Inf, NaN — zeros are
// exact and keep their sign), and the caller then re-evaluates those lanes with plain divisions (poisson_rhs_slow).
// One rare branch per stage instead of one per division keeps the independent columns interleavable.
#if defined(NS3D_EXACT_RECIP)
template <class T> struct Div2Lim;
template <> struct Div2Lim<double> { static constexpr double lo = 0x1p-700, hi = 0x1p700; };
template <> struct Div2Lim<float> { static constexpr float lo = 0x1p-60f, hi = 0x1p70f; };
// Value-based form of the same guard (k_pt_sweep2): if every P in a stencil is zero or has 2^-640 < |P| < 2^690, every
// second difference x = (e−c)−(c−w) is zero or lies in Div2Lim's range — |x| ≤ 4·2^690, and a non-zero x is a multiple
// of the smallest ulp among its operands, ≥ 2^-640·2^-52.  One test per VALUE (as it is loaded or produced) replaces
// three tests per stencil.  fp32: 2^-35 < |P| < 2^68 with 24-bit significands.
template <class T> struct ValLim;
template <> struct ValLim<double> { static constexpr double lo = 0x1p-640, hi = 0x1p690; };
template <> struct ValLim<float> { static constexpr float lo = 0x1p-35f, hi = 0x1p68f; };
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double abs_(double a) { return __builtin_fabs(a); }
__device__ __forceinline__ float abs_(float a) { return __builtin_fabsf(a); }
template <class T>
__device__ __forceinline__ T div2_known(T x, T d, T r, bool &ok)
{
    T q = x * r;
    T e = fma_(-q, d, x);
    q = fma_(e, r, q);              // RN(x/d)
    T q2 = q * r;
    e = fma_(-q2, d, q);
    q2 = fma_(e, r, q2);            // RN(RN(x/d)/d)
    const T ax = abs_(x);
    const bool z = (x == (T)0);
    ok = ok & (z | ((ax > Div2Lim<T>::lo) & (ax < Div2Lim<T>::hi)));   // bitwise on purpose: no short-circuit branches
    return z ? x : q2;
}
__device__ __forceinline__ double copysign_(double a, double b) { return __builtin_copysign(a, b); }
__device__ __forceinline__ float copysign_(float a, float b) { return __builtin_copysignf(a, b); }
template <class T>
__device__ __forceinline__ bool val_ok(T v)
{
    const T av = abs_(v);
    return (v == (T)0) | ((av > ValLim<T>::lo) & (av < ValLim<T>::hi));
}
// x/d/d for a dividend already known to be zero or inside Div2Lim's range; d > 0, so the quotient carries x's sign
// (the copysign only matters for x = −0, where the FMA sequence would give +0)
template <class T>
__device__ __forceinline__ T div2_known_nochk(T x, T d, T r)
{
    T q = x * r;
    T e = fma_(-q, d, x);
    q = fma_(e, r, q);
    T q2 = q * r;
    e = fma_(-q2, d, q);
    q2 = fma_(e, r, q2);
    return copysign_(q2, x);
}
template <class T>
__device__ __forceinline__ T poisson_rhs_nochk(T c, T w, T e, T s, T n, T b, T t, T dv, T rho_dt, const Geo<T> &g)
{
    const T d2x = (e - c) - (c - w);
    const T d2y = (n - c) - (c - s);
    const T d2z = (t - c) - (c - b);
    const T lap = (div2_known_nochk<T>(d2x, g.dx, g.rdx) + div2_known_nochk<T>(d2y, g.dy, g.rdy)) +
                  div2_known_nochk<T>(d2z, g.dz, g.rdz);
    return lap - rho_dt * dv;
}
template <class T>
__device__ __forceinline__ T poisson_rhs_ok(T c, T w, T e, T s, T n, T b, T t, T dv, T rho_dt, const Geo<T> &g, bool &ok)
{
    const T d2x = (e - c) - (c - w);
    const T d2y = (n - c) - (c - s);
    const T d2z = (t - c) - (c - b);
    const T lap = (div2_known<T>(d2x, g.dx, g.rdx, ok) + div2_known<T>(d2y, g.dy, g.rdy, ok)) +
                  div2_known<T>(d2z, g.dz, g.rdz, ok);
    return lap - rho_dt * dv;
}
template <class T>
__device__ __forceinline__ T poisson_rhs_slow(T c, T w, T e, T s, T n, T b, T t, T dv, T rho_dt, const Geo<T> &g)
{
    const T d2x = (e - c) - (c - w);
    const T d2y = (n - c) - (c - s);
    const T d2z = (t - c) - (c - b);
    const T lap = ((d2x / g.dx / g.dx) + (d2y / g.dy / g.dy)) + (d2z / g.dz / g.dz);
    return lap - rho_dt * dv;
}
#define NS3D_HAS_SLOW_PATH 1
#else
template <class T>
__device__ __forceinline__ T poisson_rhs_ok(T c, T w, T e, T s, T n, T b, T t, T dv, T rho_dt, const Geo<T> &g, bool &)
{
    return poisson_rhs<T>(c, w, e, s, n, b, t, dv, rho_dt, g);
}
template <class T>
__device__ __forceinline__ T poisson_rhs_slow(T c, T w, T e, T s, T n, T b, T t, T dv, T rho_dt, const Geo<T> &g)
{
    return poisson_rhs<T>(c, w, e, s, n, b, t, dv, rho_dt, g);
}
template <class T>
__device__ __forceinline__ T poisson_rhs_nochk(T c, T w, T e, T s, T n, T b, T t, T dv, T rho_dt, const Geo<T> &g)
{
    return poisson_rhs<T>(c, w, e, s, n, b, t, dv, rho_dt, g);
}
template <class T>
__device__ __forceinline__ bool val_ok(T) { return true; }
#define NS3D_HAS_SLOW_PATH 0
#endif

template <class T>
__device__ __forceinline__ void bad_or(bool &bad, T v)
{
#if NS3D_HAS_SLOW_PATH
    bad |= !val_ok<T>(v);
#else
    (void)bad; (void)v;
#endif
}

// Drop-in, signature-preserving (unfused) PT kernels: one thread per interior cell.
template <class T, int MODE> // MODE 0: update_dPrdτ!   1: compute_res!
__global__ __launch_bounds__(256) void k_pt_unfused(const T *__restrict__ Pr, T *__restrict__ out,
                                                    const T *__restrict__ divV, T rho_dt, T dtau, T one_m_damp,
                                                    Geo<T> g, int nx, int ny, int nz)
{
    TID3
    if (i >= nx - 2 || j >= ny - 2 || k >= nz - 2) return;
    const idx_t p = IX3(i + 1, j + 1, k + 1, nx, ny);
    const idx_t sy = nx, sz = (idx_t)nx * ny;
    const T r = poisson_rhs<T>(Pr[p], Pr[p - 1], Pr[p + 1], Pr[p - sy], Pr[p + sy], Pr[p - sz], Pr[p + sz], divV[p],
                               rho_dt, g);
    const idx_t d = IX3(i, j, k, nx - 2, ny - 2);
    if (MODE == 0) out[d] = out[d] * one_m_damp + dtau * r;
    else out[d] = r;
}
template <class T>
hipError_t update_dPrdtau(hipStream_t s, const T *Pr, T *dPrdtau, const T *divV, double rho, double dt, double dtau,
                          double damp, double dx, double dy, double dz, int nx, int ny, int nz)
{
    hipLaunchKernelGGL((k_pt_unfused<T, 0>), grid3(nx - 2, ny - 2, nz - 2, BLK3), BLK3, 0, s, Pr, dPrdtau, divV,
                       (T)rho / (T)dt, (T)dtau, (T)1.0 - (T)damp, make_geo<T>(dx, dy, dz), nx, ny, nz);
    return hipGetLastError();
}
template <class T>
hipError_t compute_res(hipStream_t s, T *Rp, const T *Pr, const T *divV, double rho, double dt, double dx, double dy,
                       double dz, int nx, int ny, int nz)
{
    hipLaunchKernelGGL((k_pt_unfused<T, 1>), grid3(nx - 2, ny - 2, nz - 2, BLK3), BLK3, 0, s, Pr, Rp, divV,
                       (T)rho / (T)dt, (T)0, (T)0, make_geo<T>(dx, dy, dz), nx, ny, nz);
    return hipGetLastError();
}

// update_Pr!  multi.jl:79-82 / gpu.jl:204-207
template <class T>
__global__ __launch_bounds__(256) void k_update_Pr(T *__restrict__ Pr, const T *__restrict__ dPrdtau, T dtau, int nx,
                                                   int ny, int nz)
{
    TID3
    if (i >= nx - 2 || j >= ny - 2 || k >= nz - 2) return;
    const idx_t p = IX3(i + 1, j + 1, k + 1, nx, ny);
    Pr[p] = Pr[p] + dtau * dPrdtau[IX3(i, j, k, nx - 2, ny - 2)];
}
template <class T>
hipError_t update_Pr(hipStream_t s, T *Pr, const T *dPrdtau, double dtau, int nx, int ny, int nz)
{
    hipLaunchKernelGGL(k_update_Pr<T>, grid3(nx - 2, ny - 2, nz - 2, BLK3), BLK3, 0, s, Pr, dPrdtau, (T)dtau, nx, ny,
                       nz);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// NaN-propagating max|A| (= Julia maximum(abs.(A)), multi.jl:466): IEEE bit patterns of non-negative doubles
// are monotone as unsigned integers and the canonical NaN 0x7FF8… sorts above +Inf, so the reduction is an
// unsigned max: wave64 butterfly (__shfl_xor) → one LDS slot per wave → one atomicMax per block.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long abs_key(double a)
{
    return (a != a) ? 0x7FF8000000000000ull : (unsigned long long)__double_as_longlong(fabs(a));
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}
__device__ __forceinline__ void block_max_to_global(unsigned long long key, unsigned long long *out)
{
    __shared__ unsigned long long wmax[16];
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    const int nw = (blockDim.x * blockDim.y * blockDim.z + 63) >> 6;
    key = wave_max_u64(key);
    if ((tid & 63) == 0) wmax[tid >> 6] = key;
    __syncthreads();
    if (tid < 64) {
        unsigned long long v = tid < nw ? wmax[tid] : 0ull;
        v = wave_max_u64(v);
        if (tid == 0 && v != 0ull) atomicMax(out, v);
    }
}
template <class T>
__global__ __launch_bounds__(256) void k_max_abs(const T *__restrict__ A, long n, unsigned long long *out)
{
    unsigned long long key = 0ull;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        unsigned long long u = abs_key((double)A[q]);
        key = u > key ? u : key;
    }
    block_max_to_global(key, out);
}
template <class T>
hipError_t max_abs_key(hipStream_t s, const T *A, long n, unsigned long long *key_dev)
{
    hipError_t e = hipMemsetAsync(key_dev, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    if (n <= 0) return hipSuccess;
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_max_abs<T>, dim3((unsigned)blocks), dim3(256), 0, s, A, n, key_dev);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// correct_V!  multi.jl:97-102 / gpu.jl:214-219
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_correct_V(T *__restrict__ Vx, T *__restrict__ Vy, T *__restrict__ Vz,
                                                   const T *__restrict__ Pr, T dt_rho, Geo<T> g, int nx, int ny,
                                                   int nz)
{
    TID3
    if (i >= nx - 1 || j >= ny - 1 || k >= nz - 1) return;
    const T c = Pr[IX3(i + 1, j + 1, k + 1, nx, ny)];
    if (j < ny - 2 && k < nz - 2) {
        const idx_t v = IX3(i + 1, j + 1, k + 1, nx + 1, ny);
        Vx[v] = Vx[v] - DIV_X(dt_rho * (c - Pr[IX3(i, j + 1, k + 1, nx, ny)]));
    }
    if (i < nx - 2 && k < nz - 2) {
        const idx_t v = IX3(i + 1, j + 1, k + 1, nx, ny + 1);
        Vy[v] = Vy[v] - DIV_Y(dt_rho * (c - Pr[IX3(i + 1, j, k + 1, nx, ny)]));
    }
    if (i < nx - 2 && j < ny - 2) {
        const idx_t v = IX3(i + 1, j + 1, k + 1, nx, ny);
        Vz[v] = Vz[v] - DIV_Z(dt_rho * (c - Pr[IX3(i + 1, j + 1, k, nx, ny)]));
    }
}
template <class T>
hipError_t correct_V(hipStream_t s, T *Vx, T *Vy, T *Vz, const T *Pr, double dt, double rho, double dx, double dy,
                     double dz, int nx, int ny, int nz)
{
    hipLaunchKernelGGL(k_correct_V<T>, grid3(nx - 1, ny - 1, nz - 1, BLK3), BLK3, 0, s, Vx, Vy, Vz, Pr,
                       (T)dt / (T)rho, make_geo<T>(dx, dy, dz), nx, ny, nz);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Boundary-plane kernels on an array of extents (sx,sy,sz): bc_x!/bc_y!/bc_z! (multi.jl:108-132),
// bc_zV! (gpu.jl:239-243), bc_xhydstatic! (gpu.jl:257-261), bc_x_Vx! (multi.jl:138-141),
// bc_x_Pr! (multi.jl:147-150).  One thread per point of the 2-D index range the reference launches.
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_bc_plane(int which, T *__restrict__ A, int sx, int sy, int sz, T a, T b, T c,
                                                  int nz_arg)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y * blockDim.y + threadIdx.y;
    switch (which) {
    case 0: // bc_x!  (iy,iz)
        if (u < sy && v < sz) {
            A[IX3(0, u, v, sx, sy)] = A[IX3(1, u, v, sx, sy)];
            A[IX3(sx - 1, u, v, sx, sy)] = A[IX3(sx - 2, u, v, sx, sy)];
        }
        break;
    case 1: // bc_y!  (ix,iz)
        if (u < sx && v < sz) {
            A[IX3(u, 0, v, sx, sy)] = A[IX3(u, 1, v, sx, sy)];
            A[IX3(u, sy - 1, v, sx, sy)] = A[IX3(u, sy - 2, v, sx, sy)];
        }
        break;
    case 2: // bc_z!  (ix,iy)
        if (u < sx && v < sy) {
            A[IX3(u, v, 0, sx, sy)] = A[IX3(u, v, 1, sx, sy)];
            A[IX3(u, v, sz - 1, sx, sy)] = A[IX3(u, v, sz - 2, sx, sy)];
        }
        break;
    case 3: // bc_zV!
        if (u < sx && v < sy) {
            A[IX3(u, v, 0, sx, sy)] = (T)0.0;
            A[IX3(u, v, sz - 1, sx, sy)] = A[IX3(u, v, sz - 2, sx, sy)];
        }
        break;
    case 4: // bc_xhydstatic!(A,dz,nz,g,ρ): a = ρ*g, b = dz ; iz = v+1
        if (u < sy && v < sz) {
            const T h = (a * ((T)(nz_arg - (v + 1)) + (T)0.5)) * b;
            A[IX3(0, u, v, sx, sy)] = h + (T)100;
            A[IX3(sx - 1, u, v, sx, sy)] = h;
        }
        break;
    case 5: // bc_x_Vx!(A,V): a = V
        if (u < sy && v < sz) A[IX3(0, u, v, sx, sy)] = a;
        break;
    case 6: // bc_x_Pr!(A,val): a = val
        if (u < sy && v < sz) A[IX3(sx - 1, u, v, sx, sy)] = a;
        break;
    }
    (void)c;
}
template <class T>
hipError_t bc_plane(hipStream_t s, int which, T *A, int sx, int sy, int sz, double a, double b, double c, int nz_arg)
{
    int U, V;
    switch (which) {
    case 0: case 4: case 5: case 6: U = sy; V = sz; break;
    case 1: U = sx; V = sz; break;
    default: U = sx; V = sy; break;
    }
    const dim3 blk(64, 4, 1);
    hipLaunchKernelGGL(k_bc_plane<T>, grid3(U, V, 1, blk), blk, 0, s, which, A, sx, sy, sz, (T)a, (T)b, (T)c, nz_arg);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// set_bc_Vel! / set_bc_Pr! as ONE launch (round 4).  The reference applies bc_x!, bc_y!, bc_z! (+ a constant or hydrostatic plane) one
// after the other (multi.jl:156-169, 175-181; gpu.jl:264-286): 8-9 launches of a few µs each per call, 12 per time step.  Every rule
// is a copy from the neighbouring cell or a constant, and the later rules read what the earlier ones wrote only on edges and corners,
// where the chain of copies ends at the interior cell with the boundary indices clamped — in the dimensions that HAVE a copy rule.  So
// the whole sequence is a gather: boundary cell (i,j,k) ← A[clamp i, clamp j, clamp k] read from cells no rule writes, then the
// overrides in the sequence's own order (bc_zV!'s zero bottom plane; Vx[1,:,:] = vin / Pr[end,:,:] = val / the hydrostatic x planes,
// which come last and cover whole planes).  Extents below 3 in a copied dimension keep the launch-per-rule path (the two faces
// would read each other).  Same values on every cell: tests/test_gpu_kernels.py compares both paths with the oracle.
// ---------------------------------------------------------------------------------------------------------
template <class T>
struct BcArr {
    T *A;
    int sx, sy, sz;
    int copy;           // bit d: dimension d has the copy rule
    int zv;             // bc_zV!: plane 0 = 0, top plane copies (gpu.jl:239-243)
    int xlo, xhi;       // 0 nothing, 1 constant, 2 hydrostatic (+100 on the low side; gpu.jl:252-261)
    T xlo_val, xhi_val, hyd_a, hyd_b;
    int hyd_nz;
};
template <class T>
struct BcArgs { BcArr<T> arr[3]; int narr; };
template <class T>
__device__ __forceinline__ void bc_cell(const BcArr<T> &b, int i, int j, int k)
{
    T v;
    const bool lo = i == 0 && b.xlo, hi = i == b.sx - 1 && b.xhi;
    if (b.zv && k == 0) v = (T)0;
    else if (lo || hi) {
        const int mode = lo ? b.xlo : b.xhi;
        if (mode == 1) v = lo ? b.xlo_val : b.xhi_val;
        else {
            const T h = (b.hyd_a * ((T)(b.hyd_nz - (k + 1)) + (T)0.5)) * b.hyd_b;
            v = lo ? h + (T)100 : h;
        }
    } else {
        const int ci = (b.copy & 1) ? min(max(i, 1), b.sx - 2) : i, cj = (b.copy & 2) ? min(max(j, 1), b.sy - 2) : j;
        const int ck = ((b.copy & 4) || b.zv) ? min(max(k, 1), b.sz - 2) : k;
        v = b.A[IX3(ci, cj, ck, b.sx, b.sy)];
    }
    b.A[IX3(i, j, k, b.sx, b.sy)] = v;
}
template <class T>
__global__ __launch_bounds__(256) void k_bc_fused(BcArgs<T> g)
{
    const int a = blockIdx.z / 3, d = blockIdx.z % 3;
    const BcArr<T> &b = g.arr[a];
    const bool applies = d == 0 ? ((b.copy & 1) || b.xlo || b.xhi) : d == 1 ? (b.copy & 2) != 0 : ((b.copy & 4) || b.zv);
    if (!applies) return;
    const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y * blockDim.y + threadIdx.y;
    if (d == 0) {
        if (u >= b.sy || v >= b.sz) return;
        if ((b.copy & 1) || b.xlo) bc_cell<T>(b, 0, u, v);
        if ((b.copy & 1) || b.xhi) bc_cell<T>(b, b.sx - 1, u, v);
    } else if (d == 1) {
        if (u >= b.sx || v >= b.sz) return;
        bc_cell<T>(b, u, 0, v);
        bc_cell<T>(b, u, b.sy - 1, v);
    } else {
        if (u >= b.sx || v >= b.sy) return;
        bc_cell<T>(b, u, v, 0);
        bc_cell<T>(b, u, v, b.sz - 1);
    }
}
// the fields of set_bc_Vel! (narr = 3: Vx, Vy, Vz) or set_bc_Pr! (narr = 1); hipErrorInvalidValue: an extent the gather form does not cover
template <class T>
hipError_t bc_fused(hipStream_t s, int what, int bc_kind, T *A0, T *A1, T *A2, int nx, int ny, int nz, int owns, double val, double rho_g,
                    double dz, int nz_arg)
{
    BcArgs<T> g;
    std::memset((void *)&g, 0, sizeof g);
    auto set = [&](int q, T *A, int sx, int sy, int sz, int copy, int zv) {
        g.arr[q].A = A; g.arr[q].sx = sx; g.arr[q].sy = sy; g.arr[q].sz = sz; g.arr[q].copy = copy; g.arr[q].zv = zv;
    };
    if (what == 0) {                    // set_bc_Vel!
        g.narr = 3;
        if (bc_kind == NS3D_BC_MULTI) { // multi.jl:157-166
            set(0, A0, nx + 1, ny, nz, 7, 0); set(1, A1, nx, ny + 1, nz, 5, 0); set(2, A2, nx, ny, nz + 1, 3, 0);
            if (owns) { g.arr[0].xlo = 1; g.arr[0].xlo_val = (T)val; }
        } else {                        // gpu.jl:265-276
            set(0, A0, nx + 1, ny, nz, 3, 1); set(1, A1, nx, ny + 1, nz, 3, 1); set(2, A2, nx, ny, nz + 1, 3, 1);
        }
    } else {                            // set_bc_Pr!
        g.narr = 1;
        if (bc_kind == NS3D_BC_MULTI) { // multi.jl:176-181
            set(0, A0, nx, ny, nz, 7, 0);
            if (owns) { g.arr[0].xhi = 1; g.arr[0].xhi_val = (T)val; }
        } else {                        // gpu.jl:282-284
            set(0, A0, nx, ny, nz, 6, 0);
            g.arr[0].xlo = g.arr[0].xhi = 2; g.arr[0].hyd_a = (T)rho_g; g.arr[0].hyd_b = (T)dz; g.arr[0].hyd_nz = nz_arg;
        }
    }
    int mu = 0, mv = 0;
    for (int q = 0; q < g.narr; ++q) {
        const BcArr<T> &b = g.arr[q];
        if (((b.copy & 1) && b.sx < 3) || ((b.copy & 2) && b.sy < 3) || (((b.copy & 4) || b.zv) && b.sz < 3)) return hipErrorInvalidValue;
        mu = max(mu, max(b.sx, b.sy)); mv = max(mv, max(b.sy, b.sz));
    }
    const dim3 blk(64, 4, 1);
    hipLaunchKernelGGL(k_bc_fused<T>, dim3((unsigned)((mu + 63) / 64), (unsigned)((mv + 3) / 4), (unsigned)(3 * g.narr)), blk, 0, s, g);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// set_cylinder!  multi.jl:249-281 (global-coordinate form) / gpu.jl:336-368 (local form, yc = yv + dx/2 sic)
// Thread range (nx+1,ny+1,nz+1) = element-wise max of the argument sizes.
// ---------------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ bool in_ellipse(T xq, T yq, T ox, T oy, T sinb, T cosb, T a2, T b2, T thr)
{
    const T xr = (xq - ox) * cosb - (yq - oy) * sinb;
    const T yr = (xq - ox) * sinb + (yq - oy) * cosb;
    return (xr * xr / a2 + yr * yr / b2) < thr;
}
// The obstacle is a cylinder along z: the four tests depend on (i, j) only (eight fp64 divisions and ≈150 instructions).  One
// thread per column and chunk of planes evaluates them once and stores down its chunk where a flag is set — nearly every
// thread leaves after the tests.  One thread per (i, j, k) cost 0.70 ms per call at 512³, twice per time step; 32 planes per
// thread: 0.038 ms (NS3D_CYL_KZ=1 / 8 / 16 / 32 / 64: 0.695 / 0.099 / 0.056 / 0.038 / 0.037).
template <class T>
__global__ __launch_bounds__(256) void k_set_cylinder(T *__restrict__ C, T *__restrict__ Vx, T *__restrict__ Vy,
                                                      T *__restrict__ Vz, T a2, T b2, T ox, T oy, T sinb, T cosb,
                                                      int local_form, T xco, T yco, T lx, T ly, T dx, T dy, int nx,
                                                      int ny, int nz, int kz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i > nx || j > ny) return;
    T xc, yc, xv, yv;
    if (!local_form) {
        xc = xco + (T)i * dx; yc = yco + (T)j * dy;
        xv = xc - dx / (T)2;  yv = yc - dy / (T)2;
    } else {
        xv = (T)i * dx - lx / (T)2; yv = (T)j * dy - ly / (T)2;
        xc = xv + dx / (T)2;        yc = yv + dx / (T)2;
    }
    const bool fC = i < nx && j < ny && in_ellipse<T>(xc, yc, ox, oy, sinb, cosb, a2, b2, (T)1.05);
    const bool fX = j < ny && in_ellipse<T>(xv, yc, ox, oy, sinb, cosb, a2, b2, (T)1.0);
    const bool fY = i < nx && in_ellipse<T>(xc, yv, ox, oy, sinb, cosb, a2, b2, (T)1.0);
    const bool fZ = i < nx && j < ny && in_ellipse<T>(xc, yc, ox, oy, sinb, cosb, a2, b2, (T)1.0);
    if (!(fC | fX | fY | fZ)) return;
    const int k0 = blockIdx.z * kz, k1 = min(k0 + kz, nz + 1);
    for (int k = k0; k < k1; ++k) {
        if (k < nz) {
            if (fC) C[IX3(i, j, k, nx, ny)] = (T)1.0;
            if (fX) Vx[IX3(i, j, k, nx + 1, ny)] = (T)0.0;
            if (fY) Vy[IX3(i, j, k, nx, ny + 1)] = (T)0.0;
        }
        if (fZ) Vz[IX3(i, j, k, nx, ny)] = (T)0.0;
    }
}
template <class T>
hipError_t set_cylinder(hipStream_t s, T *C, T *Vx, T *Vy, T *Vz, double a2, double b2, double ox, double oy,
                        double sinb, double cosb, int local_form, double xco, double yco, double lx, double ly,
                        double dx, double dy, int nx, int ny, int nz)
{
    static const int kz_env = std::getenv("NS3D_CYL_KZ") ? std::atoi(std::getenv("NS3D_CYL_KZ")) : 0;   // 1: a thread per cell, as before (A/B)
    const int kz = kz_env > 0 ? kz_env : 32;
    const dim3 blk(64, 4, 1), grd((unsigned)((nx + 1 + 63) / 64), (unsigned)((ny + 1 + 3) / 4), (unsigned)((nz + 1 + kz - 1) / kz));
    hipLaunchKernelGGL(k_set_cylinder<T>, grd, blk, 0, s, C, Vx, Vy, Vz, (T)a2, (T)b2, (T)ox, (T)oy, (T)sinb, (T)cosb, local_form,
                       (T)xco, (T)yco, (T)lx, (T)ly, (T)dx, (T)dy, nx, ny, nz, kz);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// advect! / backtrack! / lerp   multi.jl:190-243 / gpu.jl:288-334.  One thread per (ix,iy,iz) of the
// (nx+1,ny+1,nz+1) range like the reference; data-dependent 8-point gathers served by L2 / Infinity Cache.
// ix,iy,iz below are the reference's 1-based indices.
// ---------------------------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ T lerp_(T a, T b, T t) { return b * t + a * ((T)1 - t); }
__device__ __forceinline__ int clampi(long long v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); }
// δ%1 (multi.jl:196; Julia `%` = rem = C fmod): the remainder of a division by one is exact, a − trunc(a), for every finite a (±Inf and
// NaN give NaN either way); only the sign of a zero result can differ from fmod's, and the callers subtract it from 0 or 1.
__device__ __forceinline__ double fmod1(double a) { return a - __builtin_trunc(a); }
__device__ __forceinline__ float fmod1(float a) { return a - __builtin_truncf(a); }
__device__ __forceinline__ double floor_(double a) { return floor(a); }
__device__ __forceinline__ float floor_(float a) { return floorf(a); }

// koff / szg (NOT in the reference; 0 / sz reproduce it): the array is a WINDOW of a global array of szg planes that starts koff
// planes into it (ns3d_advect_wide).  The departure index is then computed from the GLOBAL plane number — `Float(iz) − δ` rounds
// differently for iz = 2 and iz = 12 when δ is below an ulp of either, so multi.jl:194 on a rank's local indices is
// decomposition-dependent even where no clamp bites — clamped to the global array and translated back.
template <class T>
__device__ __forceinline__ void backtrack(T *__restrict__ A, const T *__restrict__ A_o, T vxc, T vyc, T vzc, T dt,
                                          const Geo<T> &g, int ix, int iy, int iz, int sx, int sy, int sz, int koff = 0, int szg = 0)
{
    if (szg <= 0) szg = sz;
#if NS3D_FASTMATH
    const T ddx = dt * vxc * g.rdx, ddy = dt * vyc * g.rdy, ddz = dt * vzc * g.rdz;
#else
    const T ddx = DIV_X(dt * vxc), ddy = DIV_Y(dt * vyc), ddz = DIV_Z(dt * vzc);
#endif
    const int ix1 = clampi((long long)floor_((T)ix - ddx), 1, sx);
    const int iy1 = clampi((long long)floor_((T)iy - ddy), 1, sy);
    const int iz1g = clampi((long long)floor_((T)(iz + koff) - ddz), 1, szg);
    const int iz1 = clampi((long long)iz1g - koff, 1, sz), iz2 = clampi((long long)clampi(iz1g + 1, 1, szg) - koff, 1, sz);
    const int ix2 = clampi(ix1 + 1, 1, sx), iy2 = clampi(iy1 + 1, 1, sy);
    const T wx = (ddx > (T)0 ? (T)1 : (T)0) - fmod1(ddx);
    const T wy = (ddy > (T)0 ? (T)1 : (T)0) - fmod1(ddy);
    const T wz = (ddz > (T)0 ? (T)1 : (T)0) - fmod1(ddz);
#define AO(i_, j_, k_) A_o[IX3((i_)-1, (j_)-1, (k_)-1, sx, sy)]
    const T fy1z1 = lerp_<T>(AO(ix1, iy1, iz1), AO(ix2, iy1, iz1), wx);
    const T fy1z2 = lerp_<T>(AO(ix1, iy1, iz2), AO(ix2, iy1, iz2), wx);
    const T fy2z1 = lerp_<T>(AO(ix1, iy2, iz1), AO(ix2, iy2, iz1), wx);
    const T fy2z2 = lerp_<T>(AO(ix1, iy2, iz2), AO(ix2, iy2, iz2), wx);
#undef AO
    const T fz1 = lerp_<T>(fy1z1, fy2z1, wy);
    const T fz2 = lerp_<T>(fy1z2, fy2z2, wy);
    A[IX3(ix - 1, iy - 1, iz - 1, sx, sy)] = lerp_<T>(fz1, fz2, wz);
}

template <class T>
__global__ __launch_bounds__(256) void k_advect(T *__restrict__ Vx, const T *__restrict__ Vx_o, T *__restrict__ Vy,
                                                const T *__restrict__ Vy_o, T *__restrict__ Vz,
                                                const T *__restrict__ Vz_o, T *__restrict__ C,
                                                const T *__restrict__ C_o, T dt, Geo<T> g, int nx, int ny, int nz,
                                                int flags, int koff, int nzg)
{
    const int faithful = flags & 1, through = flags & 2;     // through: see advect() below
    const int gz0 = nzg > 0 ? nzg : 0, gz1 = nzg > 0 ? nzg + 1 : 0;         // global extents of the nz- / (nz+1)-plane arrays (0: local)
    const int ix = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int iy = blockIdx.y * blockDim.y + threadIdx.y + 1;
    const int iz = blockIdx.z * blockDim.z + threadIdx.z + 1;
    if (ix > nx + 1 || iy > ny + 1 || iz > nz + 1) return;
#define VXO(i_, j_, k_) Vx_o[IX3((i_)-1, (j_)-1, (k_)-1, nx + 1, ny)]
#define VYO(i_, j_, k_) Vy_o[IX3((i_)-1, (j_)-1, (k_)-1, nx, ny + 1)]
#define VZO(i_, j_, k_) Vz_o[IX3((i_)-1, (j_)-1, (k_)-1, nx, ny)]
    T vxc, vyc, vzc;
    // Interior threads take all four branches: straight-line code, so that the four independent
    // load → δ → gather → lerp chains overlap instead of queueing behind each other's guards (same values, same stores;
    // the second Vy store — the reference's own quirk, multi.jl:234 — still comes after the first in program order).
    if (ix > 1 && iy > 1 && iz > 1 && ix <= nx && iy <= ny && iz <= nz) {
        const T vx000 = VXO(ix, iy, iz), vx100 = VXO(ix + 1, iy, iz), vy000 = VYO(ix, iy, iz), vy010 = VYO(ix, iy + 1, iz);
        const T vz000 = VZO(ix, iy, iz), vz001 = VZO(ix, iy, iz + 1);
        const T ax = vx000;                                                                                        // :219
        const T ay = (T)0.25 * (((VYO(ix - 1, iy, iz) + VYO(ix - 1, iy + 1, iz)) + vy000) + vy010);                // :220
        const T az = (T)0.25 * (((VZO(ix - 1, iy, iz) + VZO(ix - 1, iy, iz + 1)) + vz000) + vz001);                // :221
        const T bx = (T)0.25 * (((VXO(ix, iy - 1, iz) + VXO(ix + 1, iy - 1, iz)) + vx000) + vx100);                // :225
        const T by = vy000;                                                                                        // :226
        const T bz = (T)0.25 * (((VZO(ix, iy - 1, iz) + VZO(ix, iy - 1, iz + 1)) + vz000) + vz001);                // :227
        const T cx = (T)0.25 * (((VXO(ix, iy, iz - 1) + VXO(ix + 1, iy, iz - 1)) + vx000) + vx100);                // :231
        const T cy = (T)0.25 * (((VYO(ix, iy, iz - 1) + VYO(ix, iy + 1, iz - 1)) + vy000) + vy010);                // :232
        const T cz = vz000;                                                                                        // :233
        const T ex = (T)0.5 * (vx000 + vx100), ey = (T)0.5 * (vy000 + vy010), ez = (T)0.5 * (vz000 + vz001);      // :237-239
        backtrack<T>(Vx, Vx_o, ax, ay, az, dt, g, ix, iy, iz, nx + 1, ny, nz, koff, gz0);
        backtrack<T>(Vy, Vy_o, bx, by, bz, dt, g, ix, iy, iz, nx, ny + 1, nz, koff, gz0);
        if (faithful) backtrack<T>(Vy, Vy_o, cx, cy, cz, dt, g, ix, iy, iz, nx, ny + 1, nz, koff, gz0);
        else backtrack<T>(Vz, Vz_o, cx, cy, cz, dt, g, ix, iy, iz, nx, ny, nz + 1, koff, gz1);
        backtrack<T>(C, C_o, ex, ey, ez, dt, g, ix, iy, iz, nx, ny, nz, koff, gz0);
        if (through && faithful && Vz != Vz_o) Vz[IX3(ix - 1, iy - 1, iz - 1, nx, ny)] = VZO(ix, iy, iz);
        return;
    }
    bool wx_ = false, wy_ = false, wz_ = false;
    if (ix > 1 && ix < nx + 1 && iy <= ny && iz <= nz) { // multi.jl:218-223
        wx_ = true;
        vxc = VXO(ix, iy, iz);
        vyc = (T)0.25 * (((VYO(ix - 1, iy, iz) + VYO(ix - 1, iy + 1, iz)) + VYO(ix, iy, iz)) + VYO(ix, iy + 1, iz));
        vzc = (T)0.25 * (((VZO(ix - 1, iy, iz) + VZO(ix - 1, iy, iz + 1)) + VZO(ix, iy, iz)) + VZO(ix, iy, iz + 1));
        backtrack<T>(Vx, Vx_o, vxc, vyc, vzc, dt, g, ix, iy, iz, nx + 1, ny, nz, koff, gz0);
    }
    if (iy > 1 && iy < ny + 1 && ix <= nx && iz <= nz) { // multi.jl:224-229
        wy_ = true;
        vxc = (T)0.25 * (((VXO(ix, iy - 1, iz) + VXO(ix + 1, iy - 1, iz)) + VXO(ix, iy, iz)) + VXO(ix + 1, iy, iz));
        vyc = VYO(ix, iy, iz);
        vzc = (T)0.25 * (((VZO(ix, iy - 1, iz) + VZO(ix, iy - 1, iz + 1)) + VZO(ix, iy, iz)) + VZO(ix, iy, iz + 1));
        backtrack<T>(Vy, Vy_o, vxc, vyc, vzc, dt, g, ix, iy, iz, nx, ny + 1, nz, koff, gz0);
    }
    if (iz > 1 && iz < nz + 1 && ix <= nx && iy <= ny) { // multi.jl:230-235
        if (faithful) wy_ = true; else wz_ = true;
        vxc = (T)0.25 * (((VXO(ix, iy, iz - 1) + VXO(ix + 1, iy, iz - 1)) + VXO(ix, iy, iz)) + VXO(ix + 1, iy, iz));
        vyc = (T)0.25 * (((VYO(ix, iy, iz - 1) + VYO(ix, iy + 1, iz - 1)) + VYO(ix, iy, iz)) + VYO(ix, iy + 1, iz));
        vzc = VZO(ix, iy, iz);
        // multi.jl:234 / gpu.jl:325 call backtrack!(Vy,Vy_o,…) here (sic): the SAME lane issued the branch-2
        // store to Vy[ix,iy,iz] above, so the two stores are ordered and this one wins, as in the reference.
        if (faithful) backtrack<T>(Vy, Vy_o, vxc, vyc, vzc, dt, g, ix, iy, iz, nx, ny + 1, nz, koff, gz0);
        else backtrack<T>(Vz, Vz_o, vxc, vyc, vzc, dt, g, ix, iy, iz, nx, ny, nz + 1, koff, gz1);
    }
    if (ix <= nx && iy <= ny && iz <= nz) { // multi.jl:236-241
        vxc = (T)0.5 * (VXO(ix, iy, iz) + VXO(ix + 1, iy, iz));
        vyc = (T)0.5 * (VYO(ix, iy, iz) + VYO(ix, iy + 1, iz));
        vzc = (T)0.5 * (VZO(ix, iy, iz) + VZO(ix, iy, iz + 1));
        backtrack<T>(C, C_o, vxc, vyc, vzc, dt, g, ix, iy, iz, nx, ny, nz, koff, gz0);
    }
    if (through) {      // the entries no branch stores keep their old values: written through, so that the outputs are complete
        if (!wx_ && iy <= ny && iz <= nz) Vx[IX3(ix - 1, iy - 1, iz - 1, nx + 1, ny)] = VXO(ix, iy, iz);
        if (!wy_ && ix <= nx && iz <= nz) Vy[IX3(ix - 1, iy - 1, iz - 1, nx, ny + 1)] = VYO(ix, iy, iz);
        if (!wz_ && Vz != Vz_o && ix <= nx && iy <= ny) Vz[IX3(ix - 1, iy - 1, iz - 1, nx, ny)] = VZO(ix, iy, iz);
    }
#undef VXO
#undef VYO
#undef VZO
}
// clamp(floor(·) as Int, 1, hi) of multi.jl:192-194 with the clamp applied BEFORE the conversion: the same integer for every
// finite value (the conversion is exact inside [1, hi]), the saturating behaviour of the device's float→int64 conversion for
// huge values (→ hi / 1), and 1 for NaN (fmax drops it) — what clampi((long long)floor(·)) gives on this device, without the
// multi-instruction 64-bit conversion
__device__ __forceinline__ int clampf_i(double t, int hi) { return (int)__builtin_fmin(__builtin_fmax(t, 1.0), (double)hi); }
__device__ __forceinline__ int clampf_i(float t, int hi) { return (int)__builtin_fminf(__builtin_fmaxf(t, 1.0f), (float)hi); }

// ---- advect!, LDS-windowed --------------------------------------------------------------------------------------
// k_advect reads 68 values per cell through L1/L2 (8-point gathers of four arrays plus the staggered velocity averages): at 512³
// that is ≈70 GB of cache traffic for 7.5 GB of algorithmic bytes.  Here a workgroup marches a tile of columns in z with a ring
// of xy-planes of the four OLD fields in LDS; the plane two steps ahead is fetched into registers while the current one is
// computed, and every gather and every velocity average reads LDS.  The round-2 form (64×8 columns, a four-plane window of
// 141 KB for 512 threads) kept 2 waves per SIMD and spent 44 % of its wave cycles waiting (profiles/r3_advect_sq.log): 505 VALU
// instructions per wave and plane behind four exec-masked branches that each wait for their own loads, 4.45 ms at 512³.  Now
//  * the workgroup owns 64×16 columns (1024 threads, 4 waves per SIMD) and the window holds THREE planes (iz−1 … iz+1: what a
//    departure point with |δz| < 1 and the staggered averages reach) plus the slot the next plane is fetched into: 68×18×4
//    slots × 4 arrays = 153 KB;
//  * the window reaches two columns below and one above in x (the stream runs in +x at CFL_adv = 1, multi.jl:342: departure
//    points lie up to 1+ε cells upstream), one row either side in y;
//  * a lane inside the array (ix, iy, iz > 1) runs the back-tracks as straight-line code with the staggered averages
//    sharing their loads (one back-track at a time: holding the departure cells of all four to issue their 32 gathers
//    together spilled 35-58 registers at 128 per lane);
//  * a plane lives in ring slot k mod 4, computed, not looked up (the round-2 ring of six kept a table that a per-lane k
//    turned into a scratch array: two scratch loads per back-track);
//  * in the faithful form the second branch's store to Vy[ix,iy,iz] is overwritten by the third's from the same lane
//    (multi.jl:234, App. B1): where the third branch runs, the second is a dead store and is not computed;
//  * the grid covers (nx, ny, nz) and the lanes at ix = nx, iy = ny, iz = nz write the far faces through (the reference's
//    (nx+1, ny+1, nz+1) range does nothing else there) — 8×32 tiles at 512³ instead of 9×65;
//  * loads and stores use a 32-bit in-plane offset per lane on a workgroup-uniform plane base instead of 64-bit index
//    arithmetic per access.
// 4.45 → 3.33 ms at 512³ (faithful; 4.56 → 3.8 fixed), 5.26 → 3.27 ms with the cylinder case's stream (departure points 0.7-1.3
// cells upstream, which the round-2 window did not hold); 417 VALU instructions per wave and plane, SIMDs 68 % busy.
// A lane whose departure point leaves the window takes the global gather for that back-track; same values, same expression
// trees, same final stores: bit-identical to k_advect for every δ.
template <class T>
struct AdvWin2 {
    static constexpr int TX = 64, TY = 16, XLO = 2, YLO = 1, WX = TX + 4 /* 67 used */, WY = TY + 2, NSLOT = 4, PLANE = WX * WY;
    typedef const T __attribute__((address_space(3))) *lds_ptr;
    lds_ptr L;           // [4 arrays][NSLOT][PLANE]
    int x0, y0, iz;      // 1-based first column/row of the tile, current plane
    __device__ __forceinline__ void set_plane(int iz_) { iz = iz_; }
    // plane k lives in slot k mod 4 (no table: a per-lane k would turn one into a scratch array)
    __device__ __forceinline__ int slot_off(int k) const { return (k & (NSLOT - 1)) * PLANE; }
    __device__ __forceinline__ T get(int a, int i, int j, int k) const
    {
        return L[a * (NSLOT * PLANE) + slot_off(k) + (j - (y0 - YLO)) * WX + (i - (x0 - XLO))];
    }
};
// the departure cell of one back-track (multi.jl:190-197): the index arithmetic of backtrack() with the clamp on the float side
template <class T> struct AdvDep { int i1, i2, j1, j2, k1, k2; T wx, wy, wz; };
template <class T>
__device__ __forceinline__ AdvDep<T> adv_departure(T vxc, T vyc, T vzc, T dt, const Geo<T> &g, int ix, int iy, int iz, int sx, int sy,
                                                   int sz, int koff, int szg)
{
    if (szg <= 0) szg = sz;
#if NS3D_FASTMATH
    const T ddx = dt * vxc * g.rdx, ddy = dt * vyc * g.rdy, ddz = dt * vzc * g.rdz;
#else
    const T ddx = DIV_X(dt * vxc), ddy = DIV_Y(dt * vyc), ddz = DIV_Z(dt * vzc);
#endif
    AdvDep<T> d;
    d.i1 = clampf_i(floor_((T)ix - ddx), sx);
    d.j1 = clampf_i(floor_((T)iy - ddy), sy);
    const int k1g = clampf_i(floor_((T)(iz + koff) - ddz), szg);
    d.k1 = min(max(k1g - koff, 1), sz);
    d.k2 = min(max(min(k1g + 1, szg) - koff, 1), sz);
    d.i2 = min(d.i1 + 1, sx);
    d.j2 = min(d.j1 + 1, sy);
    d.wx = (ddx > (T)0 ? (T)1 : (T)0) - fmod1(ddx);
    d.wy = (ddy > (T)0 ? (T)1 : (T)0) - fmod1(ddy);
    d.wz = (ddz > (T)0 ? (T)1 : (T)0) - fmod1(ddz);
    return d;
}
template <class T>
__device__ __forceinline__ int adv_holds(const AdvWin2<T> &w, const AdvDep<T> &d)
{
    typedef AdvWin2<T> W;
    return (int)(d.i1 >= w.x0 - W::XLO) & (int)(d.i2 <= w.x0 + W::TX) & (int)(d.j1 >= w.y0 - W::YLO) & (int)(d.j2 <= w.y0 + W::TY) &
           (int)(d.k1 >= w.iz - 1) & (int)(d.k2 <= w.iz + 1);
}
template <class T>
__device__ __forceinline__ T adv_lerp8(T v111, T v211, T v121, T v221, T v112, T v212, T v122, T v222, const AdvDep<T> &d)
{
    const T fy1z1 = lerp_<T>(v111, v211, d.wx);
    const T fy1z2 = lerp_<T>(v112, v212, d.wx);
    const T fy2z1 = lerp_<T>(v121, v221, d.wx);
    const T fy2z2 = lerp_<T>(v122, v222, d.wx);
    const T fz1 = lerp_<T>(fy1z1, fy2z1, d.wy);
    const T fz2 = lerp_<T>(fy1z2, fy2z2, d.wy);
    return lerp_<T>(fz1, fz2, d.wz);
}
template <class T>
__device__ __forceinline__ T adv_from_window(const AdvWin2<T> &w, int a, const AdvDep<T> &d)
{
    typedef AdvWin2<T> W;
    const int base = a * (W::NSLOT * W::PLANE) + (d.j1 - (w.y0 - W::YLO)) * W::WX + (d.i1 - (w.x0 - W::XLO));
    const int ex = d.i2 - d.i1, ey = (d.j2 - d.j1) * W::WX;                 // 0 where the clamp at the array end bites
    typename W::lds_ptr q1 = w.L + base + w.slot_off(d.k1), q2 = w.L + base + w.slot_off(d.k2);
    return adv_lerp8<T>(q1[0], q1[ex], q1[ey], q1[ey + ex], q2[0], q2[ex], q2[ey], q2[ey + ex], d);
}
template <class T>
__device__ __forceinline__ T adv_from_global(const T *__restrict__ A_o, const AdvDep<T> &d, int sx, int sy)
{
#define AO(i_, j_, k_) A_o[IX3((i_)-1, (j_)-1, (k_)-1, sx, sy)]
    return adv_lerp8<T>(AO(d.i1, d.j1, d.k1), AO(d.i2, d.j1, d.k1), AO(d.i1, d.j2, d.k1), AO(d.i2, d.j2, d.k1), AO(d.i1, d.j1, d.k2),
                        AO(d.i2, d.j1, d.k2), AO(d.i1, d.j2, d.k2), AO(d.i2, d.j2, d.k2), d);
#undef AO
}
template <class T>
__device__ __forceinline__ T adv_value(const AdvWin2<T> &w, int a, const T *__restrict__ A_o, const AdvDep<T> &d, int sx, int sy)
{
    return adv_holds<T>(w, d) ? adv_from_window<T>(w, a, d) : adv_from_global<T>(A_o, d, sx, sy);
}

template <class T, bool faithful>
__global__ __launch_bounds__(1024) void k_advect_win2(T *__restrict__ Vx, const T *__restrict__ Vx_o, T *__restrict__ Vy,
                                                      const T *__restrict__ Vy_o, T *__restrict__ Vz, const T *__restrict__ Vz_o,
                                                      T *__restrict__ C, const T *__restrict__ C_o, T dt, Geo<T> g, int nx, int ny,
                                                      int nz, int flags, int kz, int koff, int nzg)
{
    typedef AdvWin2<T> W;
    const bool through = flags & 2;
    const int gz0 = nzg > 0 ? nzg : 0, gz1 = nzg > 0 ? nzg + 1 : 0;
    extern __shared__ __align__(16) unsigned char advect_lds_raw[];
    T *L = reinterpret_cast<T *>(advect_lds_raw);
    const int tid = threadIdx.y * W::TX + threadIdx.x;
    const int x0 = blockIdx.x * W::TX + 1, y0 = blockIdx.y * W::TY + 1;
    const int zb = blockIdx.z * kz + 1, ze = min(zb + kz, nz + 1);          // planes iz ∈ [zb, ze), iz ≤ nz
    const int ix = x0 + threadIdx.x, iy = y0 + threadIdx.y;
    const T *const src[4] = {Vx_o, Vy_o, Vz_o, C_o};
    const int sxs[4] = {nx + 1, nx, nx, nx}, sys[4] = {ny, ny + 1, ny, ny}, szs[4] = {nz, nz, nz + 1, nz};
    // window positions this thread fills (two per array and plane: 1 224 positions, 1 024 threads), clamped like the gathers;
    // in-plane offsets fit 32 bits, the plane's base is workgroup-uniform
    unsigned fbase[4][2];
    idx_t fplane[4];
    int q[2];
    bool has[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        q[h] = tid + h * 1024;
        has[h] = q[h] < W::PLANE;
        const int qq = has[h] ? q[h] : 0;
        const int gi = x0 - W::XLO + qq % W::WX, gj = y0 - W::YLO + qq / W::WX;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int ci = min(max(gi, 1), sxs[a]), cj = min(max(gj, 1), sys[a]);
            fbase[a][h] = (unsigned)(ci - 1) + (unsigned)sxs[a] * (unsigned)(cj - 1);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) fplane[a] = (idx_t)sxs[a] * sys[a];
    auto fetch = [&](int plane, T (&v)[4][2]) {            // plane: 1-based, may lie outside the arrays
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const T *__restrict__ pl = src[a] + fplane[a] * (min(max(plane, 1), szs[a]) - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) v[a][h] = has[h] ? pl[fbase[a][h]] : (T)0;
        }
    };
    auto publish = [&](int plane, const T (&v)[4][2]) {
        const int slot = plane & (W::NSLOT - 1);            // plane ≥ 0
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (has[h]) L[(a * W::NSLOT + slot) * W::PLANE + q[h]] = v[a][h];
    };
    {
        T v[4][2];
        for (int pl = zb - 1; pl <= zb + 1; ++pl) { fetch(pl, v); publish(pl, v); }
    }
    __syncthreads();
    W w{(typename W::lds_ptr)L, x0, y0, zb};
    const bool in = ix <= nx && iy <= ny;
    const int cix = min(ix, nx), ciy = min(iy, ny);
    // stores: the in-plane offset of (ix, iy) per thread (32 bits), the plane's base workgroup-uniform
    const unsigned oVx = (unsigned)(cix - 1) + (unsigned)(nx + 1) * (unsigned)(ciy - 1), oVy = (unsigned)(cix - 1) + (unsigned)nx * (unsigned)(ciy - 1);
    const unsigned oC = oVy;                                // C and Vz: rows of nx like Vy's
    const idx_t pVx = (idx_t)(nx + 1) * ny, pVy = (idx_t)nx * (ny + 1), pC = (idx_t)nx * ny;
    const bool vz_distinct = Vz != Vz_o;
#define VXO(i_, j_, k_) w.get(0, (i_), (j_), (k_))
#define VYO(i_, j_, k_) w.get(1, (i_), (j_), (k_))
#define VZO(i_, j_, k_) w.get(2, (i_), (j_), (k_))
    for (int iz = zb; iz < ze; ++iz) {
        T nxt[4][2];
        fetch(iz + 2, nxt);                                 // in flight behind this plane's arithmetic
        w.set_plane(iz);
        T *__restrict__ const Vxp = Vx + pVx * (iz - 1), *__restrict__ const Vyp = Vy + pVy * (iz - 1), *__restrict__ const Vzp = Vz + pC * (iz - 1),
                              *__restrict__ const Cp = C + pC * (iz - 1);
        if (in) {
            const T vx000 = VXO(ix, iy, iz), vx100 = VXO(ix + 1, iy, iz), vy000 = VYO(ix, iy, iz), vy010 = VYO(ix, iy + 1, iz);
            const T vz000 = VZO(ix, iy, iz), vz001 = VZO(ix, iy, iz + 1);
            bool wx_ = false, wy_ = false, wz_ = false;
            if (ix > 1 && iy > 1 && iz > 1) {
                // all four branches of multi.jl:218-241 apply
                const T ax = vx000;                                                                                        // :219
                const T ay = (T)0.25 * (((VYO(ix - 1, iy, iz) + VYO(ix - 1, iy + 1, iz)) + vy000) + vy010);                // :220
                const T az = (T)0.25 * (((VZO(ix - 1, iy, iz) + VZO(ix - 1, iy, iz + 1)) + vz000) + vz001);                // :221
                const T cx = (T)0.25 * (((VXO(ix, iy, iz - 1) + VXO(ix + 1, iy, iz - 1)) + vx000) + vx100);                // :231
                const T cy = (T)0.25 * (((VYO(ix, iy, iz - 1) + VYO(ix, iy + 1, iz - 1)) + vy000) + vy010);                // :232
                const T cz = vz000;                                                                                        // :233
                const T ex = (T)0.5 * (vx000 + vx100), ey = (T)0.5 * (vy000 + vy010), ez = (T)0.5 * (vz000 + vz001);      // :237-239
                wx_ = true; wy_ = true;
                Vxp[oVx] = adv_value<T>(w, 0, Vx_o, adv_departure<T>(ax, ay, az, dt, g, ix, iy, iz, nx + 1, ny, nz, koff, gz0), nx + 1, ny);
                if (faithful) {
                    // the third branch back-tracks Vy (sic) and its store lands on the second's: only the third is computed
                    Vyp[oVy] = adv_value<T>(w, 1, Vy_o, adv_departure<T>(cx, cy, cz, dt, g, ix, iy, iz, nx, ny + 1, nz, koff, gz0), nx, ny + 1);
                } else {
                    const T bx = (T)0.25 * (((VXO(ix, iy - 1, iz) + VXO(ix + 1, iy - 1, iz)) + vx000) + vx100);            // :225
                    const T by = vy000;                                                                                    // :226
                    const T bz = (T)0.25 * (((VZO(ix, iy - 1, iz) + VZO(ix, iy - 1, iz + 1)) + vz000) + vz001);            // :227
                    wz_ = true;
                    Vyp[oVy] = adv_value<T>(w, 1, Vy_o, adv_departure<T>(bx, by, bz, dt, g, ix, iy, iz, nx, ny + 1, nz, koff, gz0), nx, ny + 1);
                    Vzp[oC] = adv_value<T>(w, 2, Vz_o, adv_departure<T>(cx, cy, cz, dt, g, ix, iy, iz, nx, ny, nz + 1, koff, gz1), nx, ny);
                }
                Cp[oC] = adv_value<T>(w, 3, C_o, adv_departure<T>(ex, ey, ez, dt, g, ix, iy, iz, nx, ny, nz, koff, gz0), nx, ny);
            } else {
                // cells on the low faces: the branches one by one, as the reference guards them
                T vxc, vyc, vzc;
                if (ix > 1) { // multi.jl:218-223
                    wx_ = true;
                    vxc = vx000;
                    vyc = (T)0.25 * (((VYO(ix - 1, iy, iz) + VYO(ix - 1, iy + 1, iz)) + vy000) + vy010);
                    vzc = (T)0.25 * (((VZO(ix - 1, iy, iz) + VZO(ix - 1, iy, iz + 1)) + vz000) + vz001);
                    Vxp[oVx] = adv_value<T>(w, 0, Vx_o, adv_departure<T>(vxc, vyc, vzc, dt, g, ix, iy, iz, nx + 1, ny, nz, koff, gz0), nx + 1, ny);
                }
                if (iy > 1) { // multi.jl:224-229
                    wy_ = true;
                    vxc = (T)0.25 * (((VXO(ix, iy - 1, iz) + VXO(ix + 1, iy - 1, iz)) + vx000) + vx100);
                    vyc = vy000;
                    vzc = (T)0.25 * (((VZO(ix, iy - 1, iz) + VZO(ix, iy - 1, iz + 1)) + vz000) + vz001);
                    Vyp[oVy] = adv_value<T>(w, 1, Vy_o, adv_departure<T>(vxc, vyc, vzc, dt, g, ix, iy, iz, nx, ny + 1, nz, koff, gz0), nx, ny + 1);
                }
                if (iz > 1) { // multi.jl:230-235 (sic: back-tracks Vy again in the reference; this store comes second and wins)
                    vxc = (T)0.25 * (((VXO(ix, iy, iz - 1) + VXO(ix + 1, iy, iz - 1)) + vx000) + vx100);
                    vyc = (T)0.25 * (((VYO(ix, iy, iz - 1) + VYO(ix, iy + 1, iz - 1)) + vy000) + vy010);
                    vzc = vz000;
                    if (faithful) {
                        wy_ = true;
                        Vyp[oVy] = adv_value<T>(w, 1, Vy_o, adv_departure<T>(vxc, vyc, vzc, dt, g, ix, iy, iz, nx, ny + 1, nz, koff, gz0), nx, ny + 1);
                    } else {
                        wz_ = true;
                        Vzp[oC] = adv_value<T>(w, 2, Vz_o, adv_departure<T>(vxc, vyc, vzc, dt, g, ix, iy, iz, nx, ny, nz + 1, koff, gz1), nx, ny);
                    }
                }
                { // multi.jl:236-241
                    vxc = (T)0.5 * (vx000 + vx100);
                    vyc = (T)0.5 * (vy000 + vy010);
                    vzc = (T)0.5 * (vz000 + vz001);
                    Cp[oC] = adv_value<T>(w, 3, C_o, adv_departure<T>(vxc, vyc, vzc, dt, g, ix, iy, iz, nx, ny, nz, koff, gz0), nx, ny);
                }
            }
            if (through) {      // entries no branch stores: the old value written through (complete outputs, see advect())
                if (!wx_) Vxp[oVx] = vx000;
                if (!wy_) Vyp[oVy] = vy000;
                if (!wz_ && vz_distinct) Vzp[oC] = vz000;
                // the far faces of the staggered arrays (the reference's range reaches them and does nothing there)
                if (ix == nx) Vxp[oVx + 1] = vx100;
                if (iy == ny) Vyp[oVy + nx] = vy010;
                if (iz == nz && vz_distinct) (Vzp + pC)[oC] = vz001;
            }
        }
        publish(iz + 2, nxt);
        __syncthreads();
    }
#undef VXO
#undef VYO
#undef VZO
}

// faithful: bit 0 = the reference's third branch (back-tracks Vy again, never Vz); bit 1 = WRITE-THROUGH: every entry of Vx, Vy,
// C (and of Vz unless Vz == Vz_o) is stored — the entries advect! leaves alone with the old field's value — so that
// {X_o .= X; advect!} (multi.jl:475-476) becomes ONE pass with the roles of the buffers swapped afterwards instead of four
// copies plus one pass ("avoidable by pointer swap").  Same values bit for bit.
template <class T>
hipError_t advect(hipStream_t s, T *Vx, const T *Vx_o, T *Vy, const T *Vy_o, T *Vz, const T *Vz_o, T *C,
                  const T *C_o, double dt, double dx, double dy, double dz, int nx, int ny, int nz, int faithful, int koff, int nzg)
{
    const char *glob = std::getenv("NS3D_ADVECT_GLOBAL");       // once per time step: read per call (A/B and tests flip it)
    const bool windowed = !(glob && *glob == '1');
    if (!windowed) {                                        // the one-thread-per-cell global gather (A/B, fallback)
        hipLaunchKernelGGL(k_advect<T>, grid3(nx + 1, ny + 1, nz + 1, BLK3), BLK3, 0, s, Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, C,
                           C_o, (T)dt, make_geo<T>(dx, dy, dz), nx, ny, nz, faithful, koff, nzg);
        return hipGetLastError();
    }
    typedef AdvWin2<T> W;
    const size_t lds = (size_t)4 * W::NSLOT * W::PLANE * sizeof(T);
    hipError_t ea = hipFuncSetAttribute((faithful & 1) ? (const void *)k_advect_win2<T, true> : (const void *)k_advect_win2<T, false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;
    static const int kz_env = std::getenv("NS3D_ADVECT_KZ") ? std::atoi(std::getenv("NS3D_ADVECT_KZ")) : 0;
    // planes per workgroup (three more are the window's lead-in)
    const int kz = kz_env > 0 ? kz_env : window_kz(((nx + W::TX - 1) / W::TX) * ((ny + W::TY - 1) / W::TY), nz, nz >= 128 ? 64 : 32);
    const dim3 blk(W::TX, W::TY, 1);
    const dim3 grd((unsigned)((nx + W::TX - 1) / W::TX), (unsigned)((ny + W::TY - 1) / W::TY), (unsigned)((nz + kz - 1) / kz));
    if (faithful & 1)
        hipLaunchKernelGGL((k_advect_win2<T, true>), grd, blk, lds, s, Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, C, C_o, (T)dt, make_geo<T>(dx, dy, dz),
                           nx, ny, nz, faithful, kz, koff, nzg);
    else
        hipLaunchKernelGGL((k_advect_win2<T, false>), grd, blk, lds, s, Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, C, C_o, (T)dt, make_geo<T>(dx, dy, dz),
                           nx, ny, nz, faithful, kz, koff, nzg);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// A[2:end-1,2:end-1,2:end-1] packed (what gather! receives per rank, multi.jl:399-403): one thread per inner cell
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_strip_inner(const T *__restrict__ A, T *__restrict__ out, int sx, int sy, int sz)
{
    TID3
    if (i >= sx - 2 || j >= sy - 2 || k >= sz - 2) return;
    out[IX3(i, j, k, sx - 2, sy - 2)] = A[IX3(i + 1, j + 1, k + 1, sx, sy)];
}
template <class T>
hipError_t strip_inner(hipStream_t s, const T *A, T *out, int sx, int sy, int sz)
{
    if (sx <= 2 || sy <= 2 || sz <= 2) return hipSuccess;
    hipLaunchKernelGGL(k_strip_inner<T>, grid3(sx - 2, sy - 2, sz - 2, BLK3), BLK3, 0, s, A, out, sx, sy, sz);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// One x- or y-face of a packed column-major array ↔ a packed message buffer (update_halo! of a 3-D Cartesian topology:
// z faces are contiguous planes and travel as they lie, x / y faces are strided and go through this kernel on both ends).
// dim 0: the face A[idx,:,:] (sy·sz elements, stride sx) ; dim 1: A[:,idx,:] (sx·sz, rows of sx contiguous elements).
// ---------------------------------------------------------------------------------------------------------
template <class T, bool UNPACK>
__global__ __launch_bounds__(256) void k_face_copy(T *__restrict__ A, T *__restrict__ buf, int sx, int sy, int sz, int dim, int idx)
{
    const int na = dim == 0 ? sy : sx;
    const long n = (long)na * sz, t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int a = (int)(t % na), k = (int)(t / na);
    const size_t ia = dim == 0 ? IX3(idx, a, k, sx, sy) : IX3(a, idx, k, sx, sy);
    if (UNPACK) A[ia] = buf[t];
    else buf[t] = A[ia];
}
template <class T>
hipError_t face_copy(hipStream_t s, T *A, T *buf, int sx, int sy, int sz, int dim, int idx, int unpack)
{
    const long n = (long)(dim == 0 ? sy : sx) * sz;
    if (n <= 0) return hipSuccess;
    if (dim < 0 || dim > 1 || idx < 0 || idx >= (dim == 0 ? sx : sy)) return hipErrorInvalidValue;
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (unpack) hipLaunchKernelGGL((k_face_copy<T, true>), dim3(nb), dim3(256), 0, s, A, buf, sx, sy, sz, dim, idx);
    else hipLaunchKernelGGL((k_face_copy<T, false>), dim3(nb), dim3(256), 0, s, A, buf, sx, sy, sz, dim, idx);
    return hipGetLastError();
}

// Up to NS3D_SUBBOX_MAX cx·cy·cz blocks, each between two column-major arrays of different pitches (row pitch dpx / spx elements,
// plane pitch dpl / spl; dst and src point at a block's first element), in ONE launch.  The deep-ghost state of a Cartesian
// topology (ns3d_mgpu.cpp, solve_box) moves through it: local arrays ↔ ghost-extended box, and the layers next to an x or y seam
// of every array and both sides ↔ packed message buffers.  A block's (i, j) plane is flattened over the threads, so that a block
// four cells wide (x layers) still fills its waves.
template <class T>
__global__ __launch_bounds__(256) void k_subbox_copy(ns3d_subbox_batch<T> b)
{
    int q = 0;
    unsigned blk = blockIdx.x;
#pragma unroll 1
    while (q + 1 < b.n && blk >= b.p[q].blocks) { blk -= b.p[q].blocks; ++q; }
    const ns3d_subbox<T> &s = b.p[q];
    const unsigned per_plane = (unsigned)(((long)s.cx * s.cy + 255) / 256);
    const int k = (int)(blk / per_plane);
    const long t = (long)(blk % per_plane) * 256 + threadIdx.x;
    if (k >= s.cz || t >= (long)s.cx * s.cy) return;
    const int i = (int)(t % s.cx), j = (int)(t / s.cx);
    s.dst[i + j * s.dpx + k * s.dpl] = s.src[i + j * s.spx + k * s.spl];
}
template <class T>
hipError_t subbox_copy(hipStream_t st, const ns3d_subbox_batch<T> &batch)
{
    ns3d_subbox_batch<T> b = batch;
    unsigned long total = 0;
    int m = 0;
    for (int q = 0; q < b.n; ++q) {
        ns3d_subbox<T> &s = b.p[q];
        if (s.cx <= 0 || s.cy <= 0 || s.cz <= 0) continue;
        s.blocks = (unsigned)((((long)s.cx * s.cy + 255) / 256) * s.cz);
        total += s.blocks;
        b.p[m++] = s;
    }
    b.n = m;
    if (m == 0) return hipSuccess;
    if (total > 0x7fffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_subbox_copy<T>, dim3((unsigned)total), dim3(256), 0, st, b);
    return hipGetLastError();
}

// =========================================================================================================
// The fused pseudo-transient sweep  —  THE hot kernel (≥98 % of all bytes moved).
//
// One launch = one PT iteration  { update_dPrdτ! ; update_Pr! ; set_bc_Pr! }  (multi.jl:459-463):
//     d⁺[c]  = d[c]·(1−damp) + dτ·(∇²P[c] − ρ/dt·∇V[c])                 interior cells c
//     P⁺[c]  = P[c] + dτ·d⁺[c]
//     P⁺[face/edge/corner] = boundary rule applied to P⁺ in the reference's order
// reading P (7-point), ∇V, d and writing d⁺, P⁺ into a SECOND pressure buffer (ping-pong; Jacobi sweeps read
// only the previous iterate).  Algorithmic traffic: 5 × 8 B = 40 B per cell per iteration (fp64).
//
// Boundary folding.  multi.jl applies bc_x!, bc_y!, bc_z! in that order on the whole planes, so afterwards
// every face/edge/corner cell equals the nearest interior cell (index clamped to [2,n−1] in each direction),
// then bc_x_Pr! overwrites the whole outlet plane.  gpu.jl applies bc_y!, bc_z!, then sets BOTH x planes to
// the hydrostatic profile for every (iy,iz).  Hence the thread that produces interior cell (i,j,k) also stores
// the boundary cells that clamp onto it; x planes get the copy / outlet value / hydrostatic value.
// z planes that are inter-slab halos are left to the halo exchange.
//
// Thread mapping (variant Z: z-marching register pipeline).  x is the unit-stride dimension: one wave64 owns 64
// consecutive x of RY consecutive rows and marches KZ planes in z holding P[k−1],P[k],P[k+1] of its RY rows in
// registers, so every P value is loaded once per block (plus y/x halos that hit in L2).  x±1 neighbours move
// between lanes (DPP wave shifts), the two edge lanes take one extra predicated load; y±1 neighbours are
// register rows, the first/last row take one halo-row load.  All loads for plane k+1 are issued before plane k
// is computed (software pipelining; the compiler's s_waitcnt lands at first use, one iteration later).
// =========================================================================================================
template <class T>
struct SweepArgs {
    const T *__restrict__ Pin;
    T *__restrict__ Pout;
    T *__restrict__ D;         // dPrdτ (updated in place by the single sweep; OUTPUT of the two-iteration sweep)
    const T *__restrict__ Din; // dPrdτ input of the two-iteration sweep (tiles overlap, so it cannot update in place)
    const T *__restrict__ RHS; // ∇V
    Geo<T> g;
    T rho_dt, dtau, one_m_damp;
    T outlet_val, rho_g;
    int nx, ny, nz;
    int bc_kind, owns_outlet, zlo_halo, zhi_halo;
    int k0, k1; // interior planes [k0,k1) handled by this launch
    int kz;     // planes per block
    int tx0, ty0; // origin of the launch's tile window in the plane's tile grid (0, 0: the whole grid)
    const ns3d_tile_window *win;  // host side only: the window asked for / the geometry query (ns3d_launch.h); nullptr: everything
    int cus_off;  // host side only: compute units the stream's CU mask leaves out (ns3d_reserve_cus) — the z-chunking counts the rest
    int no_faces; // NS3D_PASS_SKIP_FACES: no k_pt_faces launch behind the sweep (a split pass completes the boundary cells itself: box_pass)
};

// value stored on the x planes for target plane kk (0-based)
template <class T>
__device__ __forceinline__ T xface_val(const SweepArgs<T> &a, bool hi, T copy, int kk)
{
    if (a.bc_kind == NS3D_BC_GPU) { // gpu.jl:258-259, iz = kk+1
        const T h = (a.rho_g * ((T)(a.nz - (kk + 1)) + (T)0.5)) * a.g.dz;
        return hi ? h : h + (T)100;
    }
    return (hi && a.owns_outlet) ? a.outlet_val : copy; // multi.jl:109-110,148
}

template <class T, bool NT> __device__ __forceinline__ T ld_stream(const T *p);
template <class T, bool NT> __device__ __forceinline__ void st_stream(T *p, T v);
// store P⁺(i,j,k)=v and every boundary cell that maps onto it
template <class T, bool NT = false>
__device__ __forceinline__ void store_with_bc(const SweepArgs<T> &a, int i, int j, int k, T v)
{
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const bool xlo = (i == 1), xhi = (i == nx - 2);
    const bool ylo = (j == 1), yhi = (j == ny - 2);
    const bool zlo = (k == 1) && !a.zlo_halo, zhi = (k == nz - 2) && !a.zhi_halo;
    T *__restrict__ P = a.Pout;
    st_stream<T, NT>(&P[IX3(i, j, k, nx, ny)], v);
    if (!(xlo | xhi | ylo | yhi | zlo | zhi)) return;
#pragma unroll
    for (int zz = 0; zz < 3; ++zz) {
        if ((zz == 1 && !zlo) || (zz == 2 && !zhi)) continue;
        const int kk = zz == 0 ? k : (zz == 1 ? 0 : nz - 1);
#pragma unroll
        for (int yy = 0; yy < 3; ++yy) {
            if ((yy == 1 && !ylo) || (yy == 2 && !yhi)) continue;
            const int jj = yy == 0 ? j : (yy == 1 ? 0 : ny - 1);
            if (zz | yy) P[IX3(i, jj, kk, nx, ny)] = v;
            if (xlo) P[IX3(0, jj, kk, nx, ny)] = xface_val<T>(a, false, v, kk);
            if (xhi) P[IX3(nx - 1, jj, kk, nx, ny)] = xface_val<T>(a, true, v, kk);
        }
    }
}

// ---- variant N: one thread per interior cell, neighbours straight from global/L2 (simple baseline) ------
template <class T>
__global__ __launch_bounds__(256) void k_pt_sweep_naive(SweepArgs<T> a)
{
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = a.k0 + blockIdx.z * blockDim.z + threadIdx.z;
    if (i > a.nx - 2 || j > a.ny - 2 || k >= a.k1) return;
    const Geo<T> &g = a.g;
    const idx_t p = IX3(i, j, k, a.nx, a.ny);
    const idx_t sy = a.nx, sz = (idx_t)a.nx * a.ny;
    const T *__restrict__ P = a.Pin;
    const T c = P[p];
    const T r = poisson_rhs<T>(c, P[p - 1], P[p + 1], P[p - sy], P[p + sy], P[p - sz], P[p + sz], a.RHS[p], a.rho_dt, g);
    const idx_t d = IX3(i - 1, j - 1, k - 1, a.nx - 2, a.ny - 2);
    const T dn = a.D[d] * a.one_m_damp + a.dtau * r;
    a.D[d] = dn;
    store_with_bc<T>(a, i, j, k, c + a.dtau * dn);
}

// ---- lane shifts: value of lane−1 / lane+1 within the wave64 ---------------------------------------------
template <class T> __device__ __forceinline__ T lane_prev(T v) { return __shfl_up(v, 1, 64); }
template <class T> __device__ __forceinline__ T lane_next(T v) { return __shfl_down(v, 1, 64); }

// ---- variant Z: z-marching register pipeline -------------------------------------------------------------
template <class T, int RY, int BY>
__global__ __launch_bounds__(64 * BY) void k_pt_sweep_zmarch(SweepArgs<T> a)
{
    const int nx = a.nx, ny = a.ny;
    const Geo<T> &g = a.g;
    const int lane = threadIdx.x;
    const int i = 1 + blockIdx.x * 64 + lane;                     // P index of this lane's column
    const int j0 = 1 + (blockIdx.y * BY + threadIdx.y) * RY;      // first row of this thread
    const int kb = a.k0 + blockIdx.z * a.kz;
    const int ke = min(kb + a.kz, a.k1);
    if (j0 > ny - 2 || kb >= ke) return;                          // wave-uniform
    const bool lane_act = (i <= nx - 2);
    const int ic = min(i, nx - 1);                                // clamped load column (face column is a valid neighbour)
    const bool edge = (lane == 0) | (lane == 63);
    const int ih = lane == 0 ? i - 1 : min(i + 1, nx - 1);        // x-halo column fetched by the two edge lanes
    const idx_t sy = nx, sz = (idx_t)nx * ny;
    const idx_t dsy = nx - 2, dsz = (idx_t)(nx - 2) * (ny - 2);
    const T *__restrict__ P = a.Pin;
    const T *__restrict__ RHS = a.RHS;
    T *__restrict__ D = a.D;

    int jr[RY];       // clamped row index for P/∇V loads
    bool ract[RY];    // row is an interior row
    idx_t drow[RY];   // row offset into dPrdτ (clamped)
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const int j = j0 + r;
        ract[r] = (j <= ny - 2);
        jr[r] = min(j, ny - 1);
        drow[r] = (idx_t)(min(j, ny - 2) - 1) * dsy + (min(i, nx - 2) - 1);
    }
    const int jlo = j0 - 1, jhi = min(j0 + RY, ny - 1);

    T pm[RY], pc[RY], pp[RY]; // planes k-1, k, k+1
    T hx[RY];                 // x-halo of plane k (edge lanes only)
    T dv[RY], rv[RY];         // dPrdτ and ∇V of plane k
    T ylo, yhi;               // y-halo rows of plane k
    // prologue: planes kb-1 and kb, and everything plane kb needs
    {
        const idx_t zc = (idx_t)kb * sz, zm = zc - sz;
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            pm[r] = P[zm + (idx_t)jr[r] * sy + ic];
            pc[r] = P[zc + (idx_t)jr[r] * sy + ic];
            hx[r] = edge ? P[zc + (idx_t)jr[r] * sy + ih] : (T)0;
            dv[r] = D[(idx_t)(kb - 1) * dsz + drow[r]];
            rv[r] = RHS[zc + (idx_t)min(jr[r], ny - 2) * sy + min(i, nx - 2)];
        }
        ylo = P[zc + (idx_t)jlo * sy + ic];
        yhi = P[zc + (idx_t)jhi * sy + ic];
    }
    for (int k = kb; k < ke; ++k) {
        // ---- issue the loads of plane k+1 (consumed in the next iteration) ----
        const idx_t zn = (idx_t)(k + 1) * sz;
        const bool more = (k + 1 < ke);
        T hxn[RY], dvn[RY], rvn[RY], ylon = (T)0, yhin = (T)0;
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            pp[r] = P[zn + (idx_t)jr[r] * sy + ic];
            hxn[r] = (T)0; dvn[r] = (T)0; rvn[r] = (T)0;
            if (more) {
                hxn[r] = edge ? P[zn + (idx_t)jr[r] * sy + ih] : (T)0;
                dvn[r] = D[(idx_t)k * dsz + drow[r]];
                rvn[r] = RHS[zn + (idx_t)min(jr[r], ny - 2) * sy + min(i, nx - 2)];
            }
        }
        if (more) {
            ylon = P[zn + (idx_t)jlo * sy + ic];
            yhin = P[zn + (idx_t)jhi * sy + ic];
        }
        // ---- compute plane k ----
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const T c = pc[r];
            T w = lane_prev<T>(c), e = lane_next<T>(c);
            w = lane == 0 ? hx[r] : w;
            e = lane == 63 ? hx[r] : e;
            const T s = r == 0 ? ylo : pc[r - 1 < 0 ? 0 : r - 1];
            const T n = r == RY - 1 ? yhi : pc[r + 1 > RY - 1 ? RY - 1 : r + 1];
            const T res = poisson_rhs<T>(c, w, e, s, n, pm[r], pp[r], rv[r], a.rho_dt, g);
            const T dn = dv[r] * a.one_m_damp + a.dtau * res;
            if (lane_act && ract[r]) {
                D[(idx_t)(k - 1) * dsz + drow[r]] = dn;
                store_with_bc<T>(a, i, j0 + r, k, c + a.dtau * dn);
            }
        }
        // ---- rotate the pipeline ----
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            pm[r] = pc[r]; pc[r] = pp[r];
            hx[r] = hxn[r]; dv[r] = dvn[r]; rv[r] = rvn[r];
        }
        ylo = ylon; yhi = yhin;
    }
}

template <class T, int RY, int BY>
static hipError_t launch_zmarch(hipStream_t s, SweepArgs<T> &a, int kz)
{
    a.kz = kz;
    const int nxi = a.nx - 2, nyi = a.ny - 2, nk = a.k1 - a.k0;
    dim3 blk(64, BY, 1);
    dim3 grd((unsigned)((nxi + 63) / 64), (unsigned)((nyi + RY * BY - 1) / (RY * BY)), (unsigned)((nk + kz - 1) / kz));
    hipLaunchKernelGGL((k_pt_sweep_zmarch<T, RY, BY>), grd, blk, 0, s, a);
    return hipGetLastError();
}

// accessors of the data touched once per sweep (dPrdτ, ∇V, new Pr).  Nontemporal hints looked right on paper and cost
// 6–11 % on MI355X (512³ strict 220 000 → 235 000 Mcells·iter/s, 255×153×153 177 000 → 197 000 with plain accesses; HBM
// traffic 6.07 → 5.65 GB per pass): the rows that neighbouring tiles share are then re-fetched from HBM instead of hitting
// the L2, and the next launch finds less of its input in the Infinity Cache.  -DNS3D_NONTEMPORAL restores the hints (A/B).
template <class T, bool NT> __device__ __forceinline__ T ld_stream(const T *p)
{
#ifdef NS3D_NONTEMPORAL
    if (NT) return __builtin_nontemporal_load(p);
#endif
    return *p;
}
template <class T, bool NT> __device__ __forceinline__ void st_stream(T *p, T v)
{
#if defined(NS3D_NONTEMPORAL) || defined(NS3D_NT_STORES)
    if (NT) { __builtin_nontemporal_store(v, p); return; }
#endif
    *p = v;
}

// ---- variant P: z-marching register pipeline, whole-row workgroups, XCD-aware tile order -----------------
// Same arithmetic as variant Z.  A workgroup is WX waves wide in x (for nx ≤ 64·WX+2 it owns whole rows, so the x-halo
// lines are hits in its own CU); the grid is 1-D and the hardware's round-robin block→XCD dealing (block b → XCD b mod 8)
// is undone so that every XCD sweeps one contiguous range of tiles (y-neighbour tiles share an L2 and run together);
// dPrdτ / ∇V / new Pr go through ld_stream / st_stream (plain accesses unless built with -DNS3D_NONTEMPORAL).
// Every load of the loop body is unconditional (clamped addresses instead of `if`s; the two x-halo lanes share one
// instruction whose other lanes all read lane 0's address, i.e. one extra cache line), which keeps the loop body one
// basic block up to the stores so that the waitcnt pass can count outstanding loads instead of draining them.
// (hipcc refuses to runtime-unroll a loop that contains cross-lane operations, so the plane-ring rotation costs a few
// v_mov per step; the UNR parameter is kept for experiments only.)
template <class T, int RY, int WX, int WY, bool NT, int UNR, int MINW>
__global__ __launch_bounds__(64 * WX * WY, MINW) void k_pt_sweep_pipe(SweepArgs<T> a, int ntx, int nty)
{
    const int nx = a.nx, ny = a.ny;
    const Geo<T> &g = a.g;
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, rem = nb & 7, xcd = b & 7, loc = b >> 3;
    const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
    const int tx = tile % ntx, ty = (tile / ntx) % nty, tz = tile / (ntx * nty);

    const int lane = threadIdx.x & 63;
    const int iw = 1 + tx * (64 * WX) + (int)(threadIdx.x & ~63u); // first column of this wave
    const int i = iw + lane;
    const int j0 = 1 + (ty * WY + threadIdx.y) * RY;
    const int kb = a.k0 + tz * a.kz;
    const int ke = min(kb + a.kz, a.k1);
    if (j0 > ny - 2 || kb >= ke || iw > nx - 2) return;          // wave-uniform
    const bool lane_act = (i <= nx - 2);
    const int ic = min(i, nx - 1);
    // x-halo: lane 63 reads its east neighbour, every other lane reads the wave's west neighbour column iw-1
    const int ih = lane == 63 ? min(i + 1, nx - 1) : iw - 1;
    const int sy = nx;
    const idx_t sz = (idx_t)nx * ny;
    const int dsy = nx - 2;
    const idx_t dsz = (idx_t)(nx - 2) * (ny - 2);

    int prow[RY], hrow[RY], rrow[RY], drow[RY];
    bool ract[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const int j = j0 + r;
        ract[r] = (j <= ny - 2) && lane_act;
        const int jr = min(j, ny - 1);
        prow[r] = jr * sy + ic;
        hrow[r] = jr * sy + ih;
        rrow[r] = min(jr, ny - 2) * sy + min(i, nx - 2);
        drow[r] = (min(j, ny - 2) - 1) * dsy + (min(i, nx - 2) - 1);
    }
    const int ylo_off = (j0 - 1) * sy + ic, yhi_off = min(j0 + RY, ny - 1) * sy + ic;

    const T *__restrict__ Pk = a.Pin + (idx_t)kb * sz;
    const T *__restrict__ Rk = a.RHS + (idx_t)kb * sz;
    T *__restrict__ Dk = a.D + (idx_t)(kb - 1) * dsz;

    T pm[RY], pc[RY], pp[RY], hx[RY], dv[RY], rv[RY], ylo, yhi;
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        pm[r] = Pk[prow[r] - sz];
        pc[r] = Pk[prow[r]];
        hx[r] = Pk[hrow[r]];
        dv[r] = ld_stream<T, NT>(Dk + drow[r]);
        rv[r] = ld_stream<T, NT>(Rk + rrow[r]);
    }
    ylo = Pk[ylo_off];
    yhi = Pk[yhi_off];
    const int klast = ke - 1;
    for (int k = kb; k < ke; ++k) { // (hipcc cannot runtime-unroll a loop with cross-lane operations; UNR is unused)

        // plane k+1 (the auxiliary streams re-read plane klast on the last trip: in bounds, results unused)
        const T *__restrict__ Pn = Pk + sz;
        const idx_t adv = (k < klast) ? 1 : 0;
        const T *__restrict__ Pa = Pk + adv * sz;
        const T *__restrict__ Ra = Rk + adv * sz;
        T *__restrict__ Da = Dk + adv * dsz;
        T hxn[RY], dvn[RY], rvn[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            pp[r] = Pn[prow[r]];
            hxn[r] = Pa[hrow[r]];
            dvn[r] = ld_stream<T, NT>(Da + drow[r]);
            rvn[r] = ld_stream<T, NT>(Ra + rrow[r]);
        }
        const T ylon = Pa[ylo_off];
        const T yhin = Pa[yhi_off];
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const T c = pc[r];
            T w = lane_prev<T>(c), e = lane_next<T>(c);
            w = lane == 0 ? hx[r] : w;
            e = lane == 63 ? hx[r] : e;
            const T s = r == 0 ? ylo : pc[r - 1 < 0 ? 0 : r - 1];
            const T n = r == RY - 1 ? yhi : pc[r + 1 > RY - 1 ? RY - 1 : r + 1];
            bool ok = true;
            T res = poisson_rhs_ok<T>(c, w, e, s, n, pm[r], pp[r], rv[r], a.rho_dt, g, ok);
            if (NS3D_HAS_SLOW_PATH && __builtin_expect(!ok, 0)) res = poisson_rhs_slow<T>(c, w, e, s, n, pm[r], pp[r], rv[r], a.rho_dt, g);
            const T dn = dv[r] * a.one_m_damp + a.dtau * res;
            if (ract[r]) {
                st_stream<T, NT>(Dk + drow[r], dn);
                store_with_bc<T, NT>(a, i, j0 + r, k, c + a.dtau * dn);
            }
        }
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            pm[r] = pc[r]; pc[r] = pp[r];
            hx[r] = hxn[r]; dv[r] = dvn[r]; rv[r] = rvn[r];
        }
        ylo = ylon; yhi = yhin;
        Pk = Pn; Rk += sz; Dk += dsz;
    }
}

template <class T, int RY, int WX, int WY, bool NT, int UNR, int MINW>
static hipError_t launch_pipe(hipStream_t s, SweepArgs<T> &a, int kz)
{
    a.kz = kz;
    const int nxi = a.nx - 2, nyi = a.ny - 2, nk = a.k1 - a.k0;
    const int ntx = (nxi + 64 * WX - 1) / (64 * WX), nty = (nyi + RY * WY - 1) / (RY * WY), ntz = (nk + kz - 1) / kz;
    hipLaunchKernelGGL((k_pt_sweep_pipe<T, RY, WX, WY, NT, UNR, MINW>), dim3((unsigned)(ntx * nty * ntz)),
                       dim3(64 * WX, WY, 1), 0, s, a, ntx, nty);
    return hipGetLastError();
}
template <class T, int RY, int WY, bool NT, int UNR, int MINW>
static hipError_t launch_pipe_auto(hipStream_t s, SweepArgs<T> &a, int kz)
{
    const int nxi = a.nx - 2;
    int wx = (nxi > 256 ? 8 : nxi > 128 ? 4 : nxi > 64 ? 2 : 1);
    if (wx * WY > 8) wx = 8 / WY;
    switch (wx) {
    case 8: return launch_pipe<T, RY, (WY == 1 ? 8 : 1), WY, NT, UNR, MINW>(s, a, kz);
    case 4: return launch_pipe<T, RY, (WY <= 2 ? 4 : 1), WY, NT, UNR, MINW>(s, a, kz);
    case 2: return launch_pipe<T, RY, (WY <= 4 ? 2 : 1), WY, NT, UNR, MINW>(s, a, kz);
    default: return launch_pipe<T, RY, 1, WY, NT, UNR, MINW>(s, a, kz);
    }
}

// =========================================================================================================
// TWO PT iterations per pass over memory (temporal blocking)  —  k_pt_sweep2
//
// A single sweep is pinned to its 40 B/cell of HBM traffic.  Two consecutive Jacobi sweeps
//     level 1:  d¹ = d⁰(1−damp) + dτ(∇²P⁰ − ρ/dt ∇V),  P¹ = P⁰ + dτ d¹,  faces of P¹ by the boundary rule
//     level 2:  d² = d¹(1−damp) + dτ(∇²P¹ − ρ/dt ∇V),  P² = P¹ + dτ d²,  faces of P² by the boundary rule
// need P⁰ on a radius-2 diamond but only write d² and P², so a workgroup that keeps level 1 on chip moves
// ≈(24·overlap + 16) B per cell per TWO iterations.  Every value is computed with exactly the arithmetic of the
// single sweep (same expression tree per cell), so the result is bit-identical to two k_pt_sweep launches.
//
// Workgroup = WX×WY waves (512 or 1024 threads); tile = TX×TY columns (TX = 64·WX in x, TY = CPT·WY in y, CPT consecutive
// rows per thread), marched in z.  Per z-step s the workgroup produces level 1 of plane k1 and level 2 of plane
// k2 = k1−1:
//   registers (per column): P⁰[k1−1], P⁰[k1], P⁰[k1+1]; P¹[k2−1], P¹[k2]; d¹[k2], ∇V[k2]           (own-column z rings)
//   LDS (double-buffered):  plane k1 of P⁰ incl. a one-cell halo ring (x/y neighbours for level 1),
//                           plane k2 of P¹                              (x/y neighbours for level 2)
//   one __syncthreads() per step; the loads of step s+1 (P⁰[k1+2] and the halo ring of plane k1+2 between the two
//   levels, d⁰[k1+1] and ∇V[k1+1] after level 2 — or all of them before level 1, see EARLY) are in flight behind arithmetic.
// Level 1 is evaluated on every interior column of the tile (its x/y neighbours outside the tile come from the
// halo ring, loaded straight from P⁰), level 2 on the columns whose four neighbours are in the tile or are domain
// faces; tiles therefore overlap by two columns/rows, z-chunks by two planes.  Faces of P¹ are never materialised:
// where a level-2 stencil touches a face the boundary rule is substituted (Neumann: the cell's own P¹; outlet:
// outlet_val; gpu.jl x planes: hydrostatic value).  Faces of P² are stored by the producing thread as in the single
// sweep, or (SEPF) only the x-face cell beside an interior cell, the y/z faces following in k_pt_faces_*.  z planes that
// are inter-slab halos are not supported here: z-slab ranks pass buffers extended by a second ghost plane (slab.py).
// =========================================================================================================
// ---- the boundary cells a tile can form from its OWN output (round 4: "fold", epilogue form) ------------------------------------
// k_pt_faces is a 5 µs launch behind a 48 µs sweep on the reference's 255×153×153 grid (9 % of every pass).  Storing the boundary
// cells from inside the z-march cost more than that (SGPR pressure in the hot loop, profiles/r4_pass_chain_ab.log).  Here the march is
// untouched: AFTER it, a workgroup whose tile touches a y face or whose z-chunk holds plane 1 / nz−2 waits for its own stores
// (barrier: workgroup-scope release/acquire) and writes the y-face rows of its planes and the z-face plane over its columns — the
// same gather as k_pt_faces (face_value_own: index clamp, outlet / hydrostatic planes), whose source cells are all cells this
// workgroup stored; they are read past the L1 (agent-scope loads), which never held these lines.
__device__ __forceinline__ double ld_own(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float ld_own(const float *p)
{
    return __uint_as_float(__hip_atomic_load((const unsigned *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <class T>
__device__ __forceinline__ T face_value_own(const SweepArgs<T> &a, int i, int j, int k)
{
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    if (a.bc_kind == NS3D_BC_GPU) {
        if (i == 0) return xface_val<T>(a, false, (T)0, k);
        if (i == nx - 1) return xface_val<T>(a, true, (T)0, k);
    } else if (i == nx - 1 && a.owns_outlet) return a.outlet_val;
    const int ci = min(max(i, 1), nx - 2), cj = min(max(j, 1), ny - 2), ck = min(max(k, 1), nz - 2);
    return ld_own(&a.Pout[IX3(ci, cj, ck, nx, ny)]);
}
// x0…x1, y0…y1: the interior cells the tile stored (inclusive); kb…ke−1 its planes; tid / nthr: the workgroup's threads
template <class T>
__device__ __forceinline__ void fold_faces(const SweepArgs<T> &a, int x0, int x1, int y0, int y1, int kb, int ke, int tid, int nthr)
{
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const bool ylo = y0 == 1, yhi = y1 == ny - 2;
    const bool zlo = kb == 1 && !a.zlo_halo, zhi = ke == nz - 1 && !a.zhi_halo;
    if (!(ylo || yhi || zlo || zhi)) return;                // workgroup-uniform
    __syncthreads();
    const int xa = x0 == 1 ? 0 : x0, xb = x1 == nx - 2 ? nx - 1 : x1;     // with the x-face cells where the tile holds their neighbour
    const int wx = xb - xa + 1;
    // eight cells per thread and trip: the gathers are issued together, then the stores (one load latency per trip, not one per cell)
    constexpr int U = 8;
    if (ylo || yhi) {
        const int rows = (ylo ? 1 : 0) + (yhi ? 1 : 0), nk = ke - kb, n = wx * nk * rows;
        for (int base = tid; base < n; base += nthr * U) {
            T v[U];
            idx_t at[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = base + u * nthr;
                if (q < n) {
                    const int i = xa + q % wx, r = q / wx, k = kb + r % nk;
                    const int j = (rows == 2 ? r / nk == 0 : ylo) ? 0 : ny - 1;
                    at[u] = IX3(i, j, k, nx, ny);
                    v[u] = face_value_own<T>(a, i, j, k);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (base + u * nthr < n) a.Pout[at[u]] = v[u];
        }
    }
    if (zlo || zhi) {
        const int ya = y0 == 1 ? 0 : y0, yb = y1 == ny - 2 ? ny - 1 : y1, wy_ = yb - ya + 1;
        const int planes = (zlo ? 1 : 0) + (zhi ? 1 : 0), n = wx * wy_ * planes;
        for (int base = tid; base < n; base += nthr * U) {
            T v[U];
            idx_t at[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = base + u * nthr;
                if (q < n) {
                    const int i = xa + q % wx, r = q / wx, j = ya + r % wy_;
                    const int k = (planes == 2 ? r / wy_ == 0 : zlo) ? 0 : nz - 1;
                    at[u] = IX3(i, j, k, nx, ny);
                    v[u] = face_value_own<T>(a, i, j, k);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (base + u * nthr < n) a.Pout[at[u]] = v[u];
        }
    }
}

// boundary cells formed by the tiles themselves (fold_faces) where a separate k_pt_faces launch is a visible share of the pass:
// NS3D_FOLD_FACES=1/0 forces; default: grids up to NS3D_FOLD_MAX_CELLS — whole-grid launches only (no tile window, faces wanted)
template <class T>
static bool fold_wanted(const SweepArgs<T> &a)
{
    static const int fold_env = std::getenv("NS3D_FOLD_FACES") ? std::atoi(std::getenv("NS3D_FOLD_FACES")) : -1;
    if (a.no_faces == 1 || a.win != nullptr) return false;
    return fold_env >= 0 ? fold_env == 1 : (long)a.nx * a.ny * a.nz <= NS3D_FOLD_MAX_CELLS;
}

template <class T, int WX, int WY, int CPT, bool NT, int MINW = 1, bool SEPF = false>
__global__ __launch_bounds__(64 * WX * WY, MINW) void k_pt_sweep2(SweepArgs<T> a, int ntx, int nty)
{
    constexpr int TX = 64 * WX, TY = CPT * WY, PX = TX + 2;
    // measured on MI355X (512³, 384³; fp64 and fp32): only the fp64 256×8 separate-faces kernel gains from the early
    // loads; fp32 would drop from two workgroups per CU to one (142 instead of 124 VGPRs)
    constexpr bool EARLY = sizeof(T) == 8 && WX == 4 && WY == 2 && CPT == 4 && SEPF;
    __shared__ T L0[2][(TY + 2) * PX]; // P⁰ plane with halo ring: element (lx+1, lr+1)
    __shared__ T L1[2][TY * TX];       // P¹ plane
#if NS3D_HAS_SLOW_PATH
    // exact-reciprocal STRICT build: `Lbad` becomes (and stays) non-zero once any value that entered this tile's
    // stencils fails val_ok; from then on the tile evaluates its stencils with plain IEEE divisions.  Values are tested
    // once, by the thread that loads or produces them, and the flag travels through the same barrier as the value.
    __shared__ int Lbad;
#endif
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const Geo<T> &g = a.g;
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, rem = nb & 7, xcd = b & 7, loc = b >> 3;
    const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
    const int tx_t = a.tx0 + tile % ntx, ty_t = a.ty0 + (tile / ntx) % nty, tz_t = tile / (ntx * nty);     // ntx × nty: the launch's tile window
    const int ox = 1 + tx_t * (TX - 2), oy = 1 + ty_t * (TY - 2);
    const int kb = a.k0 + tz_t * a.kz;
    const int ke = min(kb + a.kz, a.k1);
    if (kb >= ke) return; // workgroup-uniform, before any barrier

    const int lx = threadIdx.x, wy = threadIdx.y;
    const int tid = wy * TX + lx;
    const int gi = ox + lx;
    const int ci = min(gi, nx - 1), cii = min(gi, nx - 2);
    const idx_t sz = (idx_t)nx * ny;
    const idx_t dsz = (idx_t)(nx - 2) * (ny - 2);
    const bool xlo_adj = (gi == 1), xhi_adj = (gi == nx - 2);
    const bool x_s1 = (gi <= nx - 2);
    // does this tile touch an x or y face of the domain at all?  (scalar: same for the whole workgroup)
    const bool tile_on_xy_face = (ox <= 1) || (ox + TX - 1 >= nx - 2) || (oy <= 1) || (oy + TY - 1 >= ny - 2);
    const bool tile_on_x_face = (ox <= 1) || (ox + TX - 1 >= nx - 2);
    const bool x_out = x_s1 && (lx >= 1 || xlo_adj) && (lx <= TX - 2 || xhi_adj);

    int poff[CPT], roff[CPT], doff[CPT];
    bool outc[CPT];
#pragma unroll
    for (int r = 0; r < CPT; ++r) {
        const int lr = wy * CPT + r, gj = oy + lr;
        const int cj = min(gj, ny - 1), cjj = min(gj, ny - 2);
        poff[r] = cj * nx + ci;
        roff[r] = cjj * nx + cii;
        doff[r] = (cjj - 1) * (nx - 2) + (cii - 1);
        outc[r] = x_out && (gj <= ny - 2) && (lr >= 1 || gj == 1) && (lr <= TY - 2 || gj == ny - 2);
    }
    // halo ring duties (P⁰ only): A = row below the tile, B = row above, C = the two columns beside it
    const bool hasA = (wy == 0), hasB = (wy == WY - 1), hasC = (tid < 2 * TY);
    const int offA = (oy - 1) * nx + ci;
    const int offB = min(oy + TY, ny - 1) * nx + ci;
    const int cside = tid / TY, crow = tid % TY;
    const int offC = min(oy + crow, ny - 1) * nx + (cside ? min(ox + TX, nx - 1) : ox - 1);
    const int ldsA = 0 * PX + (lx + 1), ldsB = (TY + 1) * PX + (lx + 1);
    const int ldsC = (crow + 1) * PX + (cside ? TX + 1 : 0);

    const T *__restrict__ P = a.Pin;
    const T *__restrict__ RHS = a.RHS;
    const T *__restrict__ Din = a.Din;
    T *__restrict__ D = a.D;

    T p0m[CPT], p0c[CPT], p0p[CPT];     // P⁰ planes k1-1, k1, k1+1
    T p1m[CPT], p1c[CPT];               // P¹ planes k2-1, k2
    T d1c[CPT], r1c[CPT];               // d¹[k2], ∇V[k2]
    T d0[CPT], r0[CPT];                 // d⁰[k1], ∇V[k1]
    T hA = (T)0, hB = (T)0, hC = (T)0;  // halo ring values of plane k1+1 (to be published at the end of the step)
#if NS3D_HAS_SLOW_PATH
    // boundary values substituted into level-2 stencils (outlet value, hydrostatic x planes) are launch constants
    bool bad = false;
    if (a.bc_kind == NS3D_BC_GPU) {
        const T hmin = (a.rho_g * (T)1.5) * g.dz, hmax = (a.rho_g * ((T)(nz - 2) + (T)0.5)) * g.dz;
        // |h| is monotone in k, and a non-zero h+100 is at least half an ulp of 100: the end values decide
        bad = !(val_ok<T>(hmin) && val_ok<T>(hmax) && val_ok<T>(hmin + (T)100) && val_ok<T>(hmax + (T)100));
    } else if (a.owns_outlet) bad = !val_ok<T>(a.outlet_val);
    if (tid == 0) Lbad = 0;
    __syncthreads();
#endif

    // ---- prologue: planes kb-2 (clamped), kb-1, kb of P⁰; streams of plane kb-1; publish plane kb-1 ----
    {
        const int k1 = kb - 1;
        const T *__restrict__ Pm = P + (idx_t)max(k1 - 1, 0) * sz;
        const T *__restrict__ Pc = P + (idx_t)k1 * sz;
        const T *__restrict__ Pp = P + (idx_t)(k1 + 1) * sz;
        const int ka = min(max(k1, 1), nz - 2);
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
            p0m[r] = Pm[poff[r]]; p0c[r] = Pc[poff[r]]; p0p[r] = Pp[poff[r]];
            d0[r] = ld_stream<T, NT>(Din + (idx_t)(ka - 1) * dsz + doff[r]);
            r0[r] = ld_stream<T, NT>(RHS + (idx_t)ka * sz + roff[r]);
            p1m[r] = p1c[r] = d1c[r] = r1c[r] = (T)0;
            L0[0][(wy * CPT + r + 1) * PX + lx + 1] = p0c[r];
#if NS3D_HAS_SLOW_PATH
            bad |= !val_ok<T>(p0m[r]); bad |= !val_ok<T>(p0c[r]);       // p0p is tested when it is first used (step 0)
#endif
        }
#if NS3D_HAS_SLOW_PATH
        { // halo ring of the first published plane
            T v;
            if (hasA) { v = Pc[offA]; L0[0][ldsA] = v; bad |= !val_ok<T>(v); }
            if (hasB) { v = Pc[offB]; L0[0][ldsB] = v; bad |= !val_ok<T>(v); }
            if (hasC) { v = Pc[offC]; L0[0][ldsC] = v; bad |= !val_ok<T>(v); }
            if (bad) Lbad = 1;
        }
#else
        if (hasA) L0[0][ldsA] = Pc[offA];
        if (hasB) L0[0][ldsB] = Pc[offB];
        if (hasC) L0[0][ldsC] = Pc[offC];
#endif
        if (hasA) hA = Pp[offA];
        if (hasB) hB = Pp[offB];
        if (hasC) hC = Pp[offC];
    }
    __syncthreads();

    const int nsteps = (ke - kb) + 2;
    int cur = 0;
    // EDGE steps: level 2 not yet active (s < 2) or its plane next to a z face; the bulk steps compile neither test in
    auto step = [&](const int s, auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const int k1 = kb - 1 + s, k2 = k1 - 1;
        const T *__restrict__ l0 = L0[cur];
        const T *__restrict__ l1 = L1[cur];
        // ---------------- the loads of the next step: plane k1+2 of P⁰ (+ halo ring), and with EARLY also d⁰/∇V of
        // plane k1+1 into registers of their own.  EARLY issues them before level 1, so that when step s+1 waits for
        // them the stores of this step (same in-order counter) are a whole level old; it costs 4·CPT VGPRs.  Otherwise
        // P⁰ is fetched between the levels and d⁰/∇V after level 2, straight into d0/r0. ----------------
        T p0n[CPT], d0n[EARLY ? CPT : 1], r0n[EARLY ? CPT : 1], hAn = (T)0, hBn = (T)0, hCn = (T)0;
        auto issue_next = [&]() {
            const int kp = min(k1 + 2, nz - 1);
            const int ka = min(max(k1 + 1, 1), nz - 2);
            const T *__restrict__ Pn = P + (idx_t)kp * sz;
#pragma unroll
            for (int r = 0; r < CPT; ++r) {
                p0n[r] = Pn[poff[r]];
                if constexpr (EARLY) {
                    d0n[r] = ld_stream<T, NT>(Din + (idx_t)(ka - 1) * dsz + doff[r]);
                    r0n[r] = ld_stream<T, NT>(RHS + (idx_t)ka * sz + roff[r]);
                }
            }
            (void)ka;
            if (hasA) hAn = Pn[offA];
            if (hasB) hBn = Pn[offB];
            if (hasC) hCn = Pn[offC];
        };
        if constexpr (EARLY) issue_next();
        // ---------------- level 1 at plane k1 (interior planes only) ----------------
        T p1p[CPT], d1n[CPT];
#if NS3D_HAS_SLOW_PATH
#pragma unroll
        for (int r = 0; r < CPT; ++r) bad |= !val_ok<T>(p0p[r]);  // plane k1+1 of P⁰, first use
        // wave-uniform: the tile's sticky flag (published with the values by the last barrier) or a lane's own finding
        const bool slow1 = (__builtin_amdgcn_readfirstlane(Lbad) != 0) || (__builtin_amdgcn_ballot_w64(bad) != 0);
#endif
        // one (wave-uniform) branch per level, not per cell: the CPT independent cells stay interleavable
        auto level1 = [&](auto slow_tag) {
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
            const int lr = wy * CPT + r;
            const T c = p0c[r];
            const T w = l0[(lr + 1) * PX + lx], e = l0[(lr + 1) * PX + lx + 2];
            const T sv = r == 0 ? l0[lr * PX + lx + 1] : p0c[r - 1 < 0 ? 0 : r - 1];
            const T nv = r == CPT - 1 ? l0[(lr + 2) * PX + lx + 1] : p0c[r + 1 > CPT - 1 ? CPT - 1 : r + 1];
            const T res = decltype(slow_tag)::value
                              ? poisson_rhs_slow<T>(c, w, e, sv, nv, p0m[r], p0p[r], r0[r], a.rho_dt, g)
                              : poisson_rhs_nochk<T>(c, w, e, sv, nv, p0m[r], p0p[r], r0[r], a.rho_dt, g);
            d1n[r] = d0[r] * a.one_m_damp + a.dtau * res;
            p1p[r] = c + a.dtau * d1n[r];
        }
        };
#if NS3D_HAS_SLOW_PATH
        if (__builtin_expect(slow1, 0)) level1(std::true_type{});
        else level1(std::false_type{});
#pragma unroll
        for (int r = 0; r < CPT; ++r) bad |= !val_ok<T>(p1p[r]);  // P¹ of plane k1: top neighbour of level 2 below
        const bool slow2 = slow1 || (__builtin_amdgcn_ballot_w64(bad) != 0);
#else
        level1(std::false_type{});
#endif
        if constexpr (!EARLY) issue_next();
        // ---------------- level 2 at plane k2 ----------------
        if (!EDGE || s >= 2) {
            const bool zlo = EDGE && (k2 == 1), zhi = EDGE && (k2 == nz - 2);
            const bool plain_k = !(zlo || zhi);
            T *__restrict__ Dk = D + (idx_t)(k2 - 1) * dsz;
            auto level2 = [&](auto slow_tag) {
#pragma unroll
            for (int r = 0; r < CPT; ++r) {
                const int lr = wy * CPT + r;
                const T c = p1c[r];
                T w = l1[lr * TX + max(lx - 1, 0)], e = l1[lr * TX + min(lx + 1, TX - 1)];
                T sv = r == 0 ? l1[max(lr - 1, 0) * TX + lx] : p1c[r - 1 < 0 ? 0 : r - 1];
                T nv = r == CPT - 1 ? l1[min(lr + 1, TY - 1) * TX + lx] : p1c[r + 1 > CPT - 1 ? CPT - 1 : r + 1];
                // boundary rule substituted where the stencil touches a face of P¹ (workgroup-uniform test first:
                // interior tiles skip the per-lane selects altogether)
                T bv = p1m[r], tv = p1p[r];
                if (tile_on_xy_face) {
                    const int gjf = oy + lr;
                    if (xlo_adj) w = xface_val<T>(a, false, c, k2);
                    if (xhi_adj) e = xface_val<T>(a, true, c, k2);
                    if (gjf == 1) sv = c;
                    if (gjf == ny - 2) nv = c;
                }
                if constexpr (EDGE) {
                    if (zlo) bv = c;
                    if (zhi) tv = c;
                }
                const T res = decltype(slow_tag)::value
                                  ? poisson_rhs_slow<T>(c, w, e, sv, nv, bv, tv, r1c[r], a.rho_dt, g)
                                  : poisson_rhs_nochk<T>(c, w, e, sv, nv, bv, tv, r1c[r], a.rho_dt, g);
                const T dn = d1c[r] * a.one_m_damp + a.dtau * res;
                const T pn = c + a.dtau * dn;
                if (outc[r]) {
                    st_stream<T, NT>(Dk + doff[r], dn);
                    const int gj = oy + lr;
                    // SEPF: boundary cells of P² are written by k_pt_faces_* after this kernel (keeps the rare store
                    // paths — and ≈70 VGPRs of their live state — out of the hot kernel)
                    if (SEPF || (plain_k && !tile_on_xy_face)) { // scalar test: interior tiles and planes store P² only
                        T *__restrict__ po = a.Pout + (idx_t)k2 * sz + gj * nx + gi;
                        st_stream<T, NT>(po, pn);
                        if (SEPF && tile_on_x_face) {   // the x-face cell beside it shares its cache line: one more store
                            if (xlo_adj) po[-1] = xface_val<T>(a, false, pn, k2);
                            if (xhi_adj) po[1] = xface_val<T>(a, true, pn, k2);
                        }
                    } else
                        store_with_bc<T, NT>(a, gi, gj, k2, pn);
                }
            }
            };
#if NS3D_HAS_SLOW_PATH
            if (__builtin_expect(slow2, 0)) level2(std::true_type{});
            else level2(std::false_type{});
#else
            level2(std::false_type{});
#endif
        }
        // ---------------- streams of plane k1+1 for the next level 1 (∇V[k1] becomes level 2's ∇V[k2]) ----------------
        {
            const int ka = min(max(k1 + 1, 1), nz - 2);
#pragma unroll
            for (int r = 0; r < CPT; ++r) {
                r1c[r] = r0[r];
                if constexpr (EARLY) {
                    d0[r] = d0n[r];
                    r0[r] = r0n[r];
                } else {
                    d0[r] = ld_stream<T, NT>(Din + (idx_t)(ka - 1) * dsz + doff[r]);
                    r0[r] = ld_stream<T, NT>(RHS + (idx_t)ka * sz + roff[r]);
                }
            }
        }
        // ---------------- publish plane k1+1 of P⁰ and plane k1 of P¹ into the other LDS buffers ----------------
        T *__restrict__ n0 = L0[cur ^ 1];
        T *__restrict__ n1 = L1[cur ^ 1];
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
            const int lr = wy * CPT + r;
            n0[(lr + 1) * PX + lx + 1] = p0p[r];
            n1[lr * TX + lx] = p1p[r];
        }
        if (hasA) n0[ldsA] = hA;
        if (hasB) n0[ldsB] = hB;
        if (hasC) n0[ldsC] = hC;
#if NS3D_HAS_SLOW_PATH
        if (hasA) bad |= !val_ok<T>(hA);
        if (hasB) bad |= !val_ok<T>(hB);
        if (hasC) bad |= !val_ok<T>(hC);
        if (bad) Lbad = 1;
#endif
        // ---------------- rotate the z rings ----------------
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
            p0m[r] = p0c[r]; p0c[r] = p0p[r]; p0p[r] = p0n[r];
            p1m[r] = p1c[r]; p1c[r] = p1p[r];
            d1c[r] = d1n[r];
        }
        hA = hAn; hB = hBn; hC = hCn;
        __syncthreads();
        cur ^= 1;
    };
    // the step body written out four times per trip for the two-rows-per-thread shapes (see k_pt_sweepN: the ring rotations
    // inside a trip are renamed away); the four- and six-row shapes have no registers to spare for it
    constexpr int UNR = NS3D_STEP_UNROLL > 0 ? NS3D_STEP_UNROLL : (CPT <= 2 ? 4 : 1);
    int s = 0;
    if constexpr (UNR > 1) {
        const int hot_lo = max(2, 4 - kb), hot_hi = min(nsteps, nz - kb);       // bulk: s ≥ 2 and plane k2 = kb−2+s in [2, nz−3]
        const int h0 = min(nsteps, (hot_lo + 1) & ~1);                          // an even number of edge steps first (parity)
        for (; s + 2 <= h0; s += 2) {
            step(s, std::true_type{});
            step(s + 1, std::true_type{});
        }
        if (s == h0)
            for (; s + UNR <= hot_hi; s += UNR) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) step(s + u, std::false_type{});
            }
    }
    for (; s < nsteps; ++s) step(s, std::true_type{});
    if constexpr (SEPF) {
        if (a.no_faces == NS3D_FACES_FOLDED) {
            // the interior cells this tile stored: columns lx = 1 … TX−2 (and the one beside an x face), rows lr = 1 … TY−2 likewise
            const int x0 = ox <= 1 ? 1 : ox + 1, x1 = min(ox + TX - 2 + (ox + TX - 1 == nx - 2 ? 1 : 0), nx - 2);
            const int y0 = oy <= 1 ? 1 : oy + 1, y1 = min(oy + TY - 2 + (oy + TY - 1 == ny - 2 ? 1 : 0), ny - 2);
            if (x0 <= x1 && y0 <= y1) fold_faces<T>(a, x0, x1, y0, y1, kb, ke, tid, 64 * WX * WY);
        }
    }
}

// ---- y- and z-face cells of P² as separate launches (used with the SEPF form of k_pt_sweep2) ----------------------
// Same rule as store_with_bc, evaluated per boundary cell: nearest interior cell (bc_x!, bc_y!, bc_z! in that order ≡ index
// clamp), outlet plane / hydrostatic x planes on top (multi.jl:176-181, gpu.jl:282-284).
template <class T>
__device__ __forceinline__ T face_value(const SweepArgs<T> &a, int i, int j, int k)
{
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    if (a.bc_kind == NS3D_BC_GPU) {
        if (i == 0) return xface_val<T>(a, false, (T)0, k);
        if (i == nx - 1) return xface_val<T>(a, true, (T)0, k);
    } else if (i == nx - 1 && a.owns_outlet) return a.outlet_val;
    const int ci = min(max(i, 1), nx - 2), cj = min(max(j, 1), ny - 2), ck = min(max(k, 1), nz - 2);
    return a.Pout[IX3(ci, cj, ck, nx, ny)];
}
// y-face rows (j = 0 and j = ny-1, corners included) of the interior planes [k0,k1) — the x-face cells of rows 1…ny-2 are
// stored by the sweep kernel itself, next to the interior cell they copy — and the whole z face planes (plane 0 when the launch
// contains plane 1, plane nz-1 when it contains plane nz-2): one thread per cell,
// both in ONE launch (they do not depend on each other: every boundary cell reads interior cells only), which saves a kernel
// boundary per pass — ≈2 µs, 3–4 % of a pass on the reference's 255×153×153 grid
template <class T>
__global__ __launch_bounds__(256) void k_pt_faces(SweepArgs<T> a, int ring_blocks, int nring, int lo)
{
    const int nx = a.nx, ny = a.ny;
    int b = blockIdx.x;
    if (b < nring) {                                   // y-face rows of plane k0 + b / ring_blocks
        const int q = (b % ring_blocks) * 256 + threadIdx.x;
        const int k = a.k0 + b / ring_blocks;
        if (q >= 2 * nx) return;
        const int i = q < nx ? q : q - nx, j = q < nx ? 0 : ny - 1;
        a.Pout[IX3(i, j, k, nx, ny)] = face_value<T>(a, i, j, k);
        return;
    }
    b -= nring;                                        // whole z face planes
    const int zb = (int)(((long)nx * ny + 255) / 256);
    const long q = (long)(b % zb) * 256 + threadIdx.x;
    if (q >= (long)nx * ny) return;
    const int i = (int)(q % nx), j = (int)(q / nx);
    const int k = (b / zb == 0 && lo) ? 0 : a.nz - 1;
    a.Pout[IX3(i, j, k, nx, ny)] = face_value<T>(a, i, j, k);
}
template <class T>
static hipError_t launch_faces(hipStream_t s, const SweepArgs<T> &a)
{
    const int ring_blocks = (2 * a.nx + 255) / 256, nring = ring_blocks * (a.k1 - a.k0);
    const int lo = (a.k0 == 1 && !a.zlo_halo) ? 1 : 0, hi = (a.k1 == a.nz - 1 && !a.zhi_halo) ? 1 : 0;
    const int zb = (int)(((long)a.nx * a.ny + 255) / 256);
    hipLaunchKernelGGL(k_pt_faces<T>, dim3((unsigned)(nring + zb * (lo + hi))), dim3(256), 0, s, a, ring_blocks, nring, lo);
    return hipGetLastError();
}

// The boundary cells k_pt_faces writes (y-face rows of the interior planes, whole z-face planes), restricted by where their SOURCE cell —
// the interior cell they clamp onto — lies: inside the box [c0, c1) of cells (want_core) or outside it.  A pass that is split into
// shells and a core (ns3d_mgpu.cpp box_pass) completes the boundary cells in two launches this way, each reading only cells its own
// sweeps have written, and the second one touching nothing an unpack has filled in the meantime.
template <class T>
__global__ __launch_bounds__(256) void k_pt_faces_region(SweepArgs<T> a, int cx0, int cy0, int cz0, int cx1, int cy1, int cz1, int want_core)
{
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const long nrows = 2l * nx * (nz - 2), nplanes = 2l * nx * ny;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nrows + nplanes) return;
    int i, j, k;
    if (q < nrows) { i = (int)(q % nx); const long r = q / nx; j = (r & 1) ? ny - 1 : 0; k = 1 + (int)(r >> 1); }
    else { const long r = q - nrows; i = (int)(r % nx); const long t = r / nx; j = (int)(t % ny); k = (t / ny) ? nz - 1 : 0; }
    if ((k == 0 && a.zlo_halo) || (k == nz - 1 && a.zhi_halo)) return;
    const int ci = min(max(i, 1), nx - 2), cj = min(max(j, 1), ny - 2), ck = min(max(k, 1), nz - 2);
    const int in_core = (ci >= cx0) & (ci < cx1) & (cj >= cy0) & (cj < cy1) & (ck >= cz0) & (ck < cz1);
    if (in_core != (want_core != 0)) return;
    a.Pout[IX3(i, j, k, nx, ny)] = face_value<T>(a, i, j, k);
}
template <class T>
hipError_t pt_faces_region(hipStream_t s, T *Pout, const ns3d_pt_params &p, const int c0[3], const int c1[3], int want_core)
{
    SweepArgs<T> a;
    a.Pin = Pout; a.Pout = Pout; a.D = nullptr; a.Din = nullptr; a.RHS = nullptr;
    a.g = make_geo<T>(p.dx, p.dy, p.dz);
    a.rho_dt = (T)p.rho / (T)p.dt; a.dtau = (T)p.dtau; a.one_m_damp = (T)1.0 - (T)p.damp;
    a.outlet_val = (T)p.outlet_val; a.rho_g = (T)p.rho * (T)p.g;
    a.nx = p.nx; a.ny = p.ny; a.nz = p.nz;
    a.bc_kind = p.bc_kind; a.owns_outlet = p.owns_outlet; a.zlo_halo = p.z_lo_is_halo; a.zhi_halo = p.z_hi_is_halo;
    a.k0 = 1; a.k1 = p.nz - 1; a.kz = 1; a.no_faces = 0; a.cus_off = 0; a.win = nullptr; a.tx0 = a.ty0 = 0;
    if (p.nx < 3 || p.ny < 3 || p.nz < 3) return hipErrorInvalidValue;
    const long cells = 2l * p.nx * (p.nz - 2) + 2l * p.nx * p.ny;
    hipLaunchKernelGGL(k_pt_faces_region<T>, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, a, c0[0], c0[1], c0[2], c1[0], c1[1], c1[2],
                       want_core);
    return hipGetLastError();
}

// The tile window of a launch: (ntx, nty) come in as the plane's full tile grid and go out as the window's extent; false: nothing to launch
// (an empty window, or a geometry query that has been answered).
template <class T>
static bool apply_tile_window(SweepArgs<T> &a, int TX, int TY, int OV, int &ntx, int &nty)
{
    a.tx0 = a.ty0 = 0;
    if (!a.win) return true;
    if (a.win->geom) { *a.win->geom = ns3d_tile_geom{TX, TY, OV, ntx, nty}; return false; }
    if (a.win->x1 > a.win->x0) {
        const int x0 = max(0, min(a.win->x0, ntx)), x1 = max(x0, min(a.win->x1, ntx));
        const int y0 = max(0, min(a.win->y0, nty)), y1 = max(y0, min(a.win->y1, nty));
        a.tx0 = x0; a.ty0 = y0; ntx = x1 - x0; nty = y1 - y0;
    }
    return ntx > 0 && nty > 0;
}

// Workgroups of a kernel that one CU holds at a time (registers, LDS, wave slots), and the CUs of the current device.
static int device_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n;
        else { (void)hipGetLastError(); cus = 256; }
    }
    return cus;
}
static int workgroups_per_cu(const void *kernel, int threads)
{
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, kernel) != hipSuccess) { (void)hipGetLastError(); return 1; }
    const int vg = ((at.numRegs > 0 ? at.numRegs : 128) + 7) / 8 * 8;          // VGPRs are allocated in blocks of 8
    const int waves_per_simd = min(8, 512 / vg), wg_waves = (threads + 63) / 64;
    const int by_waves = waves_per_simd * 4 / wg_waves;
    const int by_lds = at.sharedSizeBytes > 0 ? (int)((160u * 1024u) / at.sharedSizeBytes) : 64;
    return max(1, min(by_waves, by_lds));
}

// kz: planes per z-chunk.  1…89 literal.  0 and 91…99 choose the chunk length so that the launch fills whole "rounds" of
// the chip: with S = CUs × workgroups per CU slots and t tiles per chunk, c = ⌊m·S/t⌋ chunks occupy the slots m times over
// with no ragged last round (512³, 256×8 tiles: 2 720 workgroups of 32 planes = 10.6 rounds → 203 000 Mcells·iter/s;
// 1 020 of 85 planes = 3.98 rounds → 217 000).  9m asks for m rounds, 0 for two; m grows until the fill reaches 95 %.
template <class T, int WX, int WY, int CPT, bool NT, int MINW = 1, bool SEPF = false>
static hipError_t launch_sweep2(hipStream_t s, SweepArgs<T> &a, int kz)
{
    constexpr int TX = 64 * WX, TY = CPT * WY;
    const int nk = a.k1 - a.k0;
    int ntx = max(1, (a.nx - 4 + (TX - 2) - 1) / (TX - 2)), nty = max(1, (a.ny - 4 + (TY - 2) - 1) / (TY - 2));
    if (!apply_tile_window<T>(a, TX, TY, 2, ntx, nty)) return hipSuccess;
    if (kz <= 0 || kz > 90) {
        static const int per_cu = workgroups_per_cu((const void *)k_pt_sweep2<T, WX, WY, CPT, NT, MINW, SEPF>, 64 * WX * WY);
        const long slots = (long)max(8, device_cus() - a.cus_off) * per_cu, tiles = (long)ntx * nty;
        const int want = kz > 90 ? kz - 90 : 2;
        const int cmax = max(1, nk / 8);                     // chunks shorter than 8 planes are mostly pipeline fill
        long best_c = 1;
        double best_fill = 0.0;
        for (int m = want; m <= want + 12; ++m) {
            long c = m * slots / tiles;
            c = c < 1 ? 1 : (c > cmax ? cmax : c);
            const int kzc = (int)((nk + c - 1) / c);
            const long wgs = tiles * ((nk + kzc - 1) / kzc);
            const double fill = (double)wgs / (double)(((wgs + slots - 1) / slots) * slots);
            if (fill > best_fill + 1e-9) { best_fill = fill; best_c = c; }
            if (fill >= 0.95 || c == cmax) break;
        }
        kz = (int)((nk + best_c - 1) / best_c);
    }
    a.kz = kz;
    const int ntz = (nk + kz - 1) / kz;
    if (SEPF && a.no_faces != 1) a.no_faces = fold_wanted<T>(a) ? NS3D_FACES_FOLDED : 0;
    hipLaunchKernelGGL((k_pt_sweep2<T, WX, WY, CPT, NT, MINW, SEPF>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX, WY, 1), 0, s, a, ntx, nty);
    hipError_t e = hipGetLastError();
    if (SEPF && e == hipSuccess && !a.no_faces) e = launch_faces<T>(s, a);
    return e;
}

// =========================================================================================================
// NL PT iterations per pass over memory  —  k_pt_sweepN  (NL = 3, 4, in fp32 also 5; NL = 2 is kept for cross-checks against k_pt_sweep2)
//
// A two-iteration pass already sits at the HBM ceiling of its 3-read/2-write traffic mix: the only way
// up is fewer bytes per ITERATION.  NL chained Jacobi sweeps
//     level ℓ:  dˡ = dˡ⁻¹(1−damp) + dτ(∇²Pˡ⁻¹ − ρ/dt ∇V),   Pˡ = Pˡ⁻¹ + dτ dˡ        ℓ = 1 … NL
// read P⁰, d⁰, ∇V and write only d^NL, P^NL: ≈40 B per cell per NL iterations.  Same structure as k_pt_sweep2, one z-march
// with every level one plane behind the previous one: at step s level ℓ works on plane kℓ = k1 − (ℓ−1), k1 = kb−(NL−1)+s.
//   registers (per column): P⁰[k1−1..k1+1]; for ℓ = 1…NL−1: Pˡ[kℓ₊₁−1], Pˡ[kℓ₊₁], dˡ[kℓ₊₁]; ∇V[k1 … k_NL]; d⁰[k1]
//   LDS (double-buffered):  plane k1 of P⁰ with its halo ring; plane kℓ₊₁ of Pˡ for ℓ = 1…NL−1   (x/y neighbours)
//   one __syncthreads() per step.
// Level ℓ is exact on the columns at least ℓ−1 away from the tile edge (or next to a domain face, where the boundary rule
// is substituted as in k_pt_sweep2); tiles therefore overlap by 2(NL−1) columns/rows and z-chunks by 2(NL−1) planes.  Every
// value is computed with exactly the arithmetic of the single sweep: bit-identical to NL k_pt_sweep launches.
// Boundary cells of P^NL: the x-face cell beside an interior cell is stored here, y/z faces by k_pt_faces_* (SEPF form).
// =========================================================================================================
// PF: where the loads of the next step are issued — 0: between and after the levels; 1 (EARLY): before level 1.  (Loads two
// steps ahead through a second set of staging registers were measured and dropped: no gain, 16 more registers.)
template <class T, int NL, int WX, int WY, int CPT, int PF, int MINW = 1, bool FOLD = false>
__global__ __launch_bounds__(64 * WX * WY, MINW) void k_pt_sweepN(SweepArgs<T> a, int ntx, int nty)
{
    static_assert(NL >= 2 && NL <= 5, "levels");
    static_assert(PF == 0 || PF == 1 || PF == 2, "PF");
    constexpr bool EARLY = PF >= 1;
    // PF == 2 (round 4, fp32 A/B shape 25): the loads run TWO z-steps ahead — a second set of staging registers (q-set) holds what the
    // next step consumes while this step's loads fill the n-set; profiles/r4_ablation_512.log: in fp32 arithmetic and memory are equal
    // and overlap badly with one step (1.9 µs) of distance
    constexpr bool AHEAD2 = PF == 2;
    constexpr int QN = AHEAD2 ? CPT : 1;
    T p0q[QN], d0q[QN], r0q[QN], hAq = (T)0, hBq = (T)0, hCq = (T)0;
    constexpr int TX = 64 * WX, TY = CPT * WY, PX = TX + 2, OV = 2 * (NL - 1);
    static_assert(TX > OV + 2 && TY > OV + 2, "tile too small for this many levels");
    __shared__ T L0[2][(TY + 2) * PX];       // P⁰ plane with halo ring: element (lx+1, lr+1)
    __shared__ T LN[NL - 1][2][TY * TX];     // planes of P¹ … P^{NL−1}
#if NS3D_HAS_SLOW_PATH
    __shared__ int Lbad;                     // sticky: a value outside the exact-reciprocal range entered this tile (see k_pt_sweep2)
#endif
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const Geo<T> &g = a.g;
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, rem = nb & 7, xcd = b & 7, loc = b >> 3;
    const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
#if defined(NS3D_TILE_ORDER) && NS3D_TILE_ORDER == 1     // A/B: y-neighbouring tiles next to each other in an XCD's range
    const int ty_t = tile % nty, tx_t = (tile / nty) % ntx, tz_t = tile / (ntx * nty);
#elif defined(NS3D_TILE_ORDER) && NS3D_TILE_ORDER >= 2   // A/B: strips NS3D_TILE_ORDER tiles wide in x, y-fastest inside a strip
    const int tz_t = tile / (ntx * nty), t2 = tile % (ntx * nty);
    const int strip = t2 / (NS3D_TILE_ORDER * nty), ins = t2 % (NS3D_TILE_ORDER * nty);
    const int sw = min(NS3D_TILE_ORDER, ntx - strip * NS3D_TILE_ORDER);
    const int tx_t = strip * NS3D_TILE_ORDER + ins % sw, ty_t = ins / sw;
#else
    const int tx_t = a.tx0 + tile % ntx, ty_t = a.ty0 + (tile / ntx) % nty, tz_t = tile / (ntx * nty);     // ntx × nty: the launch's tile window
#endif
    const int ox = 1 + tx_t * (TX - OV), oy = 1 + ty_t * (TY - OV);
    const int kb = a.k0 + tz_t * a.kz;
    const int ke = min(kb + a.kz, a.k1);
    if (kb >= ke) return; // workgroup-uniform, before any barrier

    const int lx = threadIdx.x, wy = threadIdx.y;
    const int tid = wy * TX + lx;
    const int gi = ox + lx;
    const int ci = min(gi, nx - 1), cii = min(gi, nx - 2);
    const idx_t sz = (idx_t)nx * ny;
    const idx_t dsz = (idx_t)(nx - 2) * (ny - 2);
    const bool xlo_adj = (gi == 1), xhi_adj = (gi == nx - 2);
    // does this tile touch an x or y face of the domain at all?  (scalar: same for the whole workgroup)
    const bool tile_x_lo = (ox <= 1), tile_x_hi = (ox + TX - 1 >= nx - 2);
    const bool tile_y_lo = (oy <= 1), tile_y_hi = (oy + TY - 1 >= ny - 2);
    const bool tile_on_xy_face = tile_x_lo || tile_x_hi || tile_y_lo || tile_y_hi;
    const bool tile_on_x_face = tile_x_lo || tile_x_hi;
    // columns on which level NL is exact: NL−1 away from the tile edge, unless that edge is a face of the domain
    const bool x_out = (gi <= nx - 2) && (lx >= NL - 1 || tile_x_lo) && (lx <= TX - NL || tile_x_hi);

    int poff[CPT], roff[CPT], doff[CPT];
    bool outc[CPT];
#pragma unroll
    for (int r = 0; r < CPT; ++r) {
        const int lr = wy * CPT + r, gj = oy + lr;
        const int cj = min(gj, ny - 1), cjj = min(gj, ny - 2);
        poff[r] = cj * nx + ci;
        roff[r] = cjj * nx + cii;
        doff[r] = (cjj - 1) * (nx - 2) + (cii - 1);
        outc[r] = x_out && (gj <= ny - 2) && (lr >= NL - 1 || tile_y_lo) && (lr <= TY - NL || tile_y_hi);
    }
    // Wave-uniform level skipping (round 4, VERDICT r3 #1a; -DNS3D_LEVEL_SKIP=1).  Level ℓ is exact on the rows ℓ−1 … TY−ℓ of a tile
    // (all rows on a side that is a face of the domain); a wave none of whose CPT rows lies in that range computes nothing any
    // valid row ever reads, so it may skip level ℓ's arithmetic and the publish of its Pˡ: with 12 waves × 2 rows, waves 0 and 11
    // skip levels 3 and 4 of an interior tile, 4 of 48 wave-levels.  Same bits (138 sweepN tests) — and the same time: the pass
    // is not bound by the instructions it issues (profiles/r4_levelskip_dma_ab.log), so the default build leaves it out.
// DIAGNOSTIC builds only (WRONG results; tools/ab/ablate.sh): what part of a pass is whose — bit 1: the step's global loads replaced by
// register values, bit 2: its stores behind a condition that never holds, bit 4: no LDS publishes, neighbours read from registers,
// bit 8: no barrier per z-step
#ifndef NS3D_ABL
#define NS3D_ABL 0
#endif
#ifndef NS3D_PACK_F32
#define NS3D_PACK_F32 1      /* fp32 two-row shapes of k_pt_sweepN: both rows through v_pk_* (A/B: -DNS3D_PACK_F32=0) */
#endif
#ifndef NS3D_LEVEL_SKIP
#define NS3D_LEVEL_SKIP 0
#endif
    bool act[NL + 1];
    {
        const int wyu = __builtin_amdgcn_readfirstlane(wy);
        act[0] = act[1] = true;
#pragma unroll
        for (int l = 2; l <= NL; ++l)
            act[l] = !NS3D_LEVEL_SKIP || ((wyu * CPT + CPT - 1 >= l - 1 || tile_y_lo) && (wyu * CPT <= TY - l || tile_y_hi));
    }
    // halo ring duties (P⁰ only): A = row below the tile, B = row above, C = the two columns beside it
    const bool hasA = (wy == 0), hasB = (wy == WY - 1), hasC = (tid < 2 * TY);
    const int offA = (oy - 1) * nx + ci;
    const int offB = min(oy + TY, ny - 1) * nx + ci;
    const int cside = tid / TY, crow = tid % TY;
    const int offC = min(oy + crow, ny - 1) * nx + (cside ? min(ox + TX, nx - 1) : ox - 1);
    const int ldsA = 0 * PX + (lx + 1), ldsB = (TY + 1) * PX + (lx + 1);
    const int ldsC = (crow + 1) * PX + (cside ? TX + 1 : 0);

    const T *__restrict__ P = a.Pin;
    const T *__restrict__ RHS = a.RHS;
    const T *__restrict__ Din = a.Din;
    T *__restrict__ D = a.D;

    T p0m[CPT], p0c[CPT], p0p[CPT];           // P⁰ planes k1-1, k1, k1+1
    T pm[NL - 1][CPT], pc[NL - 1][CPT];       // Pˡ planes kℓ₊₁-1, kℓ₊₁      (index ℓ-1)
    T dc[NL - 1][CPT];                        // dˡ[kℓ₊₁]                    (index ℓ-1)
    T rr[NL][CPT];                            // ∇V[k1], ∇V[k2], …, ∇V[k_NL]
    T d0[CPT];                                // d⁰[k1]
    T hA = (T)0, hB = (T)0, hC = (T)0;        // halo ring values of plane k1+1 (published at the end of the step)
    bool bad = false;
#if NS3D_HAS_SLOW_PATH
    if (a.bc_kind == NS3D_BC_GPU) {
        const T hmin = (a.rho_g * (T)1.5) * g.dz, hmax = (a.rho_g * ((T)(nz - 2) + (T)0.5)) * g.dz;
        bad = !(val_ok<T>(hmin) && val_ok<T>(hmax) && val_ok<T>(hmin + (T)100) && val_ok<T>(hmax + (T)100));
    } else if (a.owns_outlet) bad = !val_ok<T>(a.outlet_val);
    if (tid == 0) Lbad = 0;
    __syncthreads();
#endif
    const int kfirst = kb - (NL - 1);         // plane of level 1 at step 0
    // ---- prologue: planes kfirst-1 (clamped), kfirst, kfirst+1 of P⁰; streams of plane kfirst; publish plane kfirst ----
    {
        const int k1 = kfirst;
        const T *__restrict__ Pm = P + (idx_t)min(max(k1 - 1, 0), nz - 1) * sz;
        const T *__restrict__ Pc = P + (idx_t)min(max(k1, 0), nz - 1) * sz;
        const T *__restrict__ Pp = P + (idx_t)min(max(k1 + 1, 0), nz - 1) * sz;
        const int ka = min(max(k1, 1), nz - 2);
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
            p0m[r] = Pm[poff[r]]; p0c[r] = Pc[poff[r]]; p0p[r] = Pp[poff[r]];
            d0[r] = ld_stream<T, true>(Din + (idx_t)(ka - 1) * dsz + doff[r]);
            rr[0][r] = ld_stream<T, true>(RHS + (idx_t)ka * sz + roff[r]);
#pragma unroll
            for (int l = 0; l < NL - 1; ++l) { pm[l][r] = pc[l][r] = dc[l][r] = (T)0; rr[l + 1][r] = (T)0; }
            L0[0][(wy * CPT + r + 1) * PX + lx + 1] = p0c[r];
#if NS3D_HAS_SLOW_PATH
            bad |= !val_ok<T>(p0m[r]); bad |= !val_ok<T>(p0c[r]);       // p0p is tested when it is first used (step 0)
#endif
        }
        {
            T v;
            if (hasA) { v = Pc[offA]; L0[0][ldsA] = v; bad_or(bad, v); }
            if (hasB) { v = Pc[offB]; L0[0][ldsB] = v; bad_or(bad, v); }
            if (hasC) { v = Pc[offC]; L0[0][ldsC] = v; bad_or(bad, v); }
#if NS3D_HAS_SLOW_PATH
            if (bad) Lbad = 1;
#endif
        }
        if (hasA) hA = Pp[offA];
        if (hasB) hB = Pp[offB];
        if (hasC) hC = Pp[offC];
        if constexpr (AHEAD2) {                 // what step 0 would have loaded: plane kfirst+2 of P⁰, d⁰/∇V of plane kfirst+1
            const T *__restrict__ Pn = P + (idx_t)min(max(k1 + 2, 0), nz - 1) * sz;
            const int kq = min(max(k1 + 1, 1), nz - 2);
#pragma unroll
            for (int r = 0; r < CPT; ++r) {
                p0q[r] = Pn[poff[r]];
                d0q[r] = ld_stream<T, true>(Din + (idx_t)(kq - 1) * dsz + doff[r]);
                r0q[r] = ld_stream<T, true>(RHS + (idx_t)kq * sz + roff[r]);
            }
            if (hasA) hAq = Pn[offA];
            if (hasB) hBq = Pn[offB];
            if (hasC) hCq = Pn[offC];
        }
    }
    __syncthreads();

    const int nsteps = (ke - kb) + OV;
    int cur = 0;
    // EDGE steps: the pipeline is still filling (s < 2(ℓ−1) for some level) or some level's plane touches a z face of the domain
    // (the boundary rule replaces the neighbour beyond it); everywhere else — the bulk of a chunk — neither test is compiled in
    auto step = [&](const int s, auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const int k1 = kfirst + s;
        // ---------------- the loads of the next step: plane k1+2 of P⁰ (+ halo ring), d⁰/∇V of plane k1+1 ----------------
        T p0n[CPT], d0n[CPT], r0n[CPT], hAn = (T)0, hBn = (T)0, hCn = (T)0;
        auto issue_next = [&]() {
            const int kp = min(max(k1 + 2 + (AHEAD2 ? 1 : 0), 0), nz - 1);
            const int ka = min(max(k1 + 1 + (AHEAD2 ? 1 : 0), 1), nz - 2);
            const T *__restrict__ Pn = P + (idx_t)kp * sz;
#ifdef NS3D_LOAD_PRIO       // A/B: the waves that are about to issue the next step's loads go first
            __builtin_amdgcn_s_setprio(3);
#endif
#if NS3D_ABL & 1
            (void)Pn; (void)ka;
#pragma unroll
            for (int r = 0; r < CPT; ++r) { p0n[r] = p0m[r]; d0n[r] = d0[r]; r0n[r] = rr[0][r]; }
            hAn = hA; hBn = hB; hCn = hC;
#else
#pragma unroll
            for (int r = 0; r < CPT; ++r) {
                p0n[r] = Pn[poff[r]];
                d0n[r] = ld_stream<T, true>(Din + (idx_t)(ka - 1) * dsz + doff[r]);
                r0n[r] = ld_stream<T, true>(RHS + (idx_t)ka * sz + roff[r]);
            }
            if (hasA) hAn = Pn[offA];
            if (hasB) hBn = Pn[offB];
            if (hasC) hCn = Pn[offC];
#endif
#ifdef NS3D_LOAD_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        if constexpr (EARLY) issue_next();
        // ---------------- level 1 at plane k1 ----------------
        T fresh[CPT], dnew[CPT];                 // Pˡ and dˡ of the plane the current level has just produced
#if NS3D_HAS_SLOW_PATH
#pragma unroll
        for (int r = 0; r < CPT; ++r) bad |= !val_ok<T>(p0p[r]);  // plane k1+1 of P⁰, first use
        bool slow = (__builtin_amdgcn_readfirstlane(Lbad) != 0) || (__builtin_amdgcn_ballot_w64(bad) != 0);   // wave-uniform
#else
        const bool slow = false;
#endif
        {
            const T *__restrict__ l0 = L0[cur];
            auto level1 = [&](auto slow_tag) {
#if NS3D_VEC2
                if constexpr (NS3D_PACK_F32 && sizeof(T) == 4 && CPT == 2 && !decltype(slow_tag)::value) {
                    // fp32, two rows per thread: both rows through the packed instructions (same operations, same order per lane)
                    const int lr = wy * CPT;
                    const f32x2 c = mk2(p0c[0], p0c[1]);
                    const f32x2 w = mk2(l0[(lr + 1) * PX + lx], l0[(lr + 2) * PX + lx]), e = mk2(l0[(lr + 1) * PX + lx + 2], l0[(lr + 2) * PX + lx + 2]);
                    const f32x2 sv = mk2(l0[lr * PX + lx + 1], p0c[0]), nv = mk2(p0c[1], l0[(lr + 3) * PX + lx + 1]);
                    const f32x2 res = poisson_rhs_v2(c, w, e, sv, nv, mk2(p0m[0], p0m[1]), mk2(p0p[0], p0p[1]), mk2(rr[0][0], rr[0][1]), (float)a.rho_dt, g);
                    const f32x2 dn = mk2(d0[0], d0[1]) * (float)a.one_m_damp + (float)a.dtau * res;
                    const f32x2 fr = c + (float)a.dtau * dn;
                    dnew[0] = dn.x; dnew[1] = dn.y; fresh[0] = fr.x; fresh[1] = fr.y;
                    return;
                }
#endif
#pragma unroll
                for (int r = 0; r < CPT; ++r) {
                    const int lr = wy * CPT + r;
                    const T c = p0c[r];
#if NS3D_ABL & 4
                    (void)lr; (void)l0;
                    const T w = c * (T)0.5, e = c * (T)0.25, sv = p0m[r] * (T)0.5, nv = p0p[r] * (T)0.5;
#else
                    const T w = l0[(lr + 1) * PX + lx], e = l0[(lr + 1) * PX + lx + 2];
                    const T sv = r == 0 ? l0[lr * PX + lx + 1] : p0c[r - 1 < 0 ? 0 : r - 1];
                    const T nv = r == CPT - 1 ? l0[(lr + 2) * PX + lx + 1] : p0c[r + 1 > CPT - 1 ? CPT - 1 : r + 1];
#endif
                    const T res = decltype(slow_tag)::value
                                      ? poisson_rhs_slow<T>(c, w, e, sv, nv, p0m[r], p0p[r], rr[0][r], a.rho_dt, g)
                                      : poisson_rhs_nochk<T>(c, w, e, sv, nv, p0m[r], p0p[r], rr[0][r], a.rho_dt, g);
                    dnew[r] = d0[r] * a.one_m_damp + a.dtau * res;
                    fresh[r] = c + a.dtau * dnew[r];
                }
            };
            if (__builtin_expect(slow, 0)) level1(std::true_type{});
            else level1(std::false_type{});
        }
        if constexpr (!EARLY) issue_next();
        // ---------------- levels 2 … NL, each one plane behind the previous one ----------------
#pragma unroll
        for (int l = 2; l <= NL; ++l) {
            const int kl = k1 - (l - 1);
            T *__restrict__ npub = LN[l - 2][cur ^ 1];
            // publish Pˡ⁻¹ of the plane just produced (x/y neighbours of level l in the NEXT step)
            if (act[l - 1] && (!(NS3D_ABL & 4) || a.nx < 0)) {
#pragma unroll
                for (int r = 0; r < CPT; ++r) npub[(wy * CPT + r) * TX + lx] = fresh[r];
            }
#if NS3D_HAS_SLOW_PATH
#pragma unroll
            for (int r = 0; r < CPT; ++r) bad |= !val_ok<T>(fresh[r]);   // top neighbour of level l below
            slow = slow || (__builtin_amdgcn_ballot_w64(bad) != 0);
#endif
            T out_p[CPT], out_d[CPT];
            if ((!EDGE || s >= 2 * (l - 1)) && act[l]) {
                const T *__restrict__ ll = LN[l - 2][cur];
                const bool zlo = EDGE && (kl == 1), zhi = EDGE && (kl == nz - 2);
                auto level = [&](auto slow_tag) {
#if NS3D_VEC2
                    if constexpr (NS3D_PACK_F32 && sizeof(T) == 4 && CPT == 2 && !decltype(slow_tag)::value) {
                        // fp32, two rows per thread: operands gathered per row (boundary rule included), arithmetic on both rows at once
                        T cs[2], ws[2], es[2], ss[2], ns[2], bs[2], ts[2];
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int lr = wy * CPT + r;
                            const T c = pc[l - 2][r];
                            T w = ll[lr * TX + max(lx - 1, 0)], e = ll[lr * TX + min(lx + 1, TX - 1)];
                            T sv = r == 0 ? ll[max(lr - 1, 0) * TX + lx] : pc[l - 2][0];
                            T nv = r == 1 ? ll[min(lr + 1, TY - 1) * TX + lx] : pc[l - 2][1];
                            T bv = pm[l - 2][r], tv = fresh[r];
                            if (tile_on_xy_face) {
                                const int gjf = oy + lr;
                                if (xlo_adj) w = xface_val<T>(a, false, c, kl);
                                if (xhi_adj) e = xface_val<T>(a, true, c, kl);
                                if (gjf == 1) sv = c;
                                if (gjf == ny - 2) nv = c;
                            }
                            if constexpr (EDGE) {
                                if (zlo) bv = c;
                                if (zhi) tv = c;
                            }
                            cs[r] = c; ws[r] = w; es[r] = e; ss[r] = sv; ns[r] = nv; bs[r] = bv; ts[r] = tv;
                        }
                        const f32x2 c = mk2(cs[0], cs[1]);
                        const f32x2 res = poisson_rhs_v2(c, mk2(ws[0], ws[1]), mk2(es[0], es[1]), mk2(ss[0], ss[1]), mk2(ns[0], ns[1]),
                                                         mk2(bs[0], bs[1]), mk2(ts[0], ts[1]), mk2(rr[l - 1][0], rr[l - 1][1]), (float)a.rho_dt, g);
                        const f32x2 dn = mk2(dc[l - 2][0], dc[l - 2][1]) * (float)a.one_m_damp + (float)a.dtau * res;
                        const f32x2 pn = c + (float)a.dtau * dn;
                        out_d[0] = dn.x; out_d[1] = dn.y; out_p[0] = pn.x; out_p[1] = pn.y;
                        return;
                    }
#endif
#pragma unroll
                    for (int r = 0; r < CPT; ++r) {
                        const int lr = wy * CPT + r;
                        const T c = pc[l - 2][r];
#if NS3D_ABL & 4
                        (void)ll;
                        T w = c * (T)0.5, e = c * (T)0.25, sv = pm[l - 2][r] * (T)0.5, nv = fresh[r] * (T)0.5;
#else
                        T w = ll[lr * TX + max(lx - 1, 0)], e = ll[lr * TX + min(lx + 1, TX - 1)];
                        T sv = r == 0 ? ll[max(lr - 1, 0) * TX + lx] : pc[l - 2][r - 1 < 0 ? 0 : r - 1];
                        T nv = r == CPT - 1 ? ll[min(lr + 1, TY - 1) * TX + lx] : pc[l - 2][r + 1 > CPT - 1 ? CPT - 1 : r + 1];
#endif
                        T bv = pm[l - 2][r], tv = fresh[r];
                        if (tile_on_xy_face) {      // boundary rule substituted where the stencil touches a face of Pˡ⁻¹
                            const int gjf = oy + lr;
                            if (xlo_adj) w = xface_val<T>(a, false, c, kl);
                            if (xhi_adj) e = xface_val<T>(a, true, c, kl);
                            if (gjf == 1) sv = c;
                            if (gjf == ny - 2) nv = c;
                        }
                        if constexpr (EDGE) {
                            if (zlo) bv = c;
                            if (zhi) tv = c;
                        }
                        const T res = decltype(slow_tag)::value
                                          ? poisson_rhs_slow<T>(c, w, e, sv, nv, bv, tv, rr[l - 1][r], a.rho_dt, g)
                                          : poisson_rhs_nochk<T>(c, w, e, sv, nv, bv, tv, rr[l - 1][r], a.rho_dt, g);
                        out_d[r] = dc[l - 2][r] * a.one_m_damp + a.dtau * res;
                        out_p[r] = c + a.dtau * out_d[r];
                    }
                };
                if (__builtin_expect(slow, 0)) level(std::true_type{});
                else level(std::false_type{});
                if (l == NL) {               // the output level: d^NL and P^NL of plane kl
                    T *__restrict__ Dk = D + (idx_t)(kl - 1) * dsz;
#pragma unroll
                    for (int r = 0; r < CPT; ++r) {
                        if (outc[r] && (!(NS3D_ABL & 2) || a.nx < 0)) {
                            st_stream<T, true>(Dk + doff[r], out_d[r]);
                            const int gj = oy + wy * CPT + r;
                            T *__restrict__ po = a.Pout + (idx_t)kl * sz + gj * nx + gi;
                            st_stream<T, true>(po, out_p[r]);
                            if (tile_on_x_face) {   // the x-face cell beside it shares its cache line: one more store
                                if (xlo_adj) po[-1] = xface_val<T>(a, false, out_p[r], kl);
                                if (xhi_adj) po[1] = xface_val<T>(a, true, out_p[r], kl);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < CPT; ++r) { out_p[r] = (T)0; out_d[r] = (T)0; }
            }
            // rotate the rings of Pˡ⁻¹ / dˡ⁻¹ (consumed above), then hand level l's plane to level l+1
#pragma unroll
            for (int r = 0; r < CPT; ++r) {
                pm[l - 2][r] = pc[l - 2][r]; pc[l - 2][r] = fresh[r];
                dc[l - 2][r] = dnew[r];
                fresh[r] = out_p[r]; dnew[r] = out_d[r];
            }
        }
        // ---------------- ∇V ring, streams of plane k1+1 for the next level 1 ----------------
        {
            const int ka = min(max(k1 + 1, 1), nz - 2);
#pragma unroll
            for (int r = 0; r < CPT; ++r) {
#pragma unroll
                for (int l = NL - 1; l >= 1; --l) rr[l][r] = rr[l - 1][r];
                if constexpr (AHEAD2) {
                    d0[r] = d0q[r]; rr[0][r] = r0q[r];
                    d0q[r] = d0n[r]; r0q[r] = r0n[r];
                } else if constexpr (EARLY) {
                    d0[r] = d0n[r];
                    rr[0][r] = r0n[r];
                } else {
                    (void)d0n; (void)r0n;
#if NS3D_ABL & 1
                    (void)ka; d0[r] = dnew[r]; rr[0][r] = rr[1][r];
#else
                    d0[r] = ld_stream<T, true>(Din + (idx_t)(ka - 1) * dsz + doff[r]);
                    rr[0][r] = ld_stream<T, true>(RHS + (idx_t)ka * sz + roff[r]);
#endif
                }
            }
        }
        // ---------------- publish plane k1+1 of P⁰ ----------------
        T *__restrict__ n0 = L0[cur ^ 1];
        if (!(NS3D_ABL & 4) || a.nx < 0) {
#pragma unroll
        for (int r = 0; r < CPT; ++r) n0[(wy * CPT + r + 1) * PX + lx + 1] = p0p[r];
        if (hasA) n0[ldsA] = hA;
        if (hasB) n0[ldsB] = hB;
        if (hasC) n0[ldsC] = hC;
        }
#if NS3D_HAS_SLOW_PATH
        if (hasA) bad |= !val_ok<T>(hA);
        if (hasB) bad |= !val_ok<T>(hB);
        if (hasC) bad |= !val_ok<T>(hC);
        if (bad) Lbad = 1;
#endif
        if constexpr (AHEAD2) {
#pragma unroll
            for (int r = 0; r < CPT; ++r) { p0m[r] = p0c[r]; p0c[r] = p0p[r]; p0p[r] = p0q[r]; p0q[r] = p0n[r]; }
            hA = hAq; hB = hBq; hC = hCq;
            hAq = hAn; hBq = hBn; hCq = hCn;
        } else {
#pragma unroll
        for (int r = 0; r < CPT; ++r) { p0m[r] = p0c[r]; p0c[r] = p0p[r]; p0p[r] = p0n[r]; }
        hA = hAn; hB = hBn; hC = hCn;
        }
#if defined(NS3D_NO_STEP_BARRIER) || (NS3D_ABL & 8)     // A/B (WRONG results): what the one barrier per z-step costs a CU that holds a single workgroup
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the wave's own LDS traffic only
#else
        __syncthreads();
#endif
        cur ^= 1;
    };
    // The z rings rotate every step (≈ 16 register-pair moves per row, a quarter of a step's vector instructions); with the
    // body repeated UNR times per trip the copies inside a trip are renamed away and only the back edge shuffles: +4…7 % at
    // 512³ (profiles/r2_step_unroll_ab.log).  The compiler does not unroll a loop with a barrier and a run-time trip count by
    // itself, so the repetition is written out; four copies for the two-rows-per-thread shapes (two spilled registers), two for
    // the four-row 512-thread ones, none where three or more rows per thread already fill the register budget of a 768- or
    // 1024-thread workgroup (fp32 64×48: 34 registers would spill).
    // (4 is as good as 6 or 12; an ODD count is 15–35 % slower: the LDS double-buffer parity stops being a compile-time fact)
    constexpr int UNR = NS3D_STEP_UNROLL > 0 ? NS3D_STEP_UNROLL : ((WX * WY >= 12 && CPT >= 3) ? 1 : (CPT <= 2 ? 4 : 2));
    int s = 0;
    if constexpr (UNR > 1) {
        // bulk steps [hot_lo, hot_hi): every level active, planes 2 … nz−3 for the levels that substitute z faces (ℓ ≥ 2)
        const int hot_lo = max(OV, NL + 1 - kfirst), hot_hi = min(nsteps, nz - 1 - kfirst);
        const int h0 = min(nsteps, (hot_lo + 1) & ~1);          // an EVEN number of edge steps first: the double-buffer parity
        for (; s + 2 <= h0; s += 2) {                           // is back at 0 after every trip of every loop here
            step(s, std::true_type{});
            step(s + 1, std::true_type{});
        }
        if (s == h0)
            for (; s + UNR <= hot_hi; s += UNR) {
#pragma unroll
#ifdef NS3D_NO_EDGE_SPLIT      // A/B: the general step form in the bulk loop as well
                for (int u = 0; u < UNR; ++u) step(s + u, std::true_type{});
#else
                for (int u = 0; u < UNR; ++u) step(s + u, std::false_type{});
#endif
            }
    }
    for (; s < nsteps; ++s) step(s, std::true_type{});
    if constexpr (FOLD) {       // the boundary cells next to this tile's own output, instead of a k_pt_faces launch (fold_faces)
        const int x0 = tile_x_lo ? 1 : ox + NL - 1, x1 = tile_x_hi ? nx - 2 : ox + TX - NL;
        const int y0 = tile_y_lo ? 1 : oy + NL - 1, y1 = tile_y_hi ? ny - 2 : oy + TY - NL;
        if (x0 <= x1 && y0 <= y1) fold_faces<T>(a, x0, x1, y0, y1, kb, ke, tid, 64 * WX * WY);
    }
}

template <class T, int NL, int WX, int WY, int CPT, int PF, int MINW = 1, bool FOLD = false>
static hipError_t launch_sweepN(hipStream_t s, SweepArgs<T> &a, int kz)
{
    constexpr int TX = 64 * WX, TY = CPT * WY, OV = 2 * (NL - 1);
    constexpr size_t lds = (2ul * (TY + 2) * (TX + 2) + 2ul * (NL - 1) * TY * TX) * sizeof(T) + 64;
    if constexpr (!(TX > OV + 2 && TY > OV + 2 && lds <= 160ul * 1024)) {
        return hipErrorInvalidValue;            // tile too small for this many levels, or its planes exceed the 160 KB of LDS
    } else {
    const int nk = a.k1 - a.k0;
    int ntx = max(1, (a.nx - 2 - OV + (TX - OV) - 1) / (TX - OV)), nty = max(1, (a.ny - 2 - OV + (TY - OV) - 1) / (TY - OV));
    if (!apply_tile_window<T>(a, TX, TY, OV, ntx, nty)) return hipSuccess;
    if (kz <= 0) {
        // chunks per tile column that minimise the z-steps the slowest CU marches: rounds of workgroups × (planes per chunk + the
        // 2(NL−1) steps a chunk spends filling its pipeline) — the tail round of an unlucky tile count (1024²: 1026 tiles on 256
        // CUs) costs a whole chunk, so such grids want shorter chunks
        static const int per_cu = workgroups_per_cu((const void *)k_pt_sweepN<T, NL, WX, WY, CPT, PF, MINW, FOLD>, 64 * WX * WY);
        const long slots = (long)max(8, device_cus() - a.cus_off) * per_cu, tiles = (long)ntx * nty;
        const int cmax = max(1, nk / (6 * NL));
        long best_c = 1, best_cost = -1;
        for (long c = 1; c <= cmax && c <= 64; ++c) {
            const int kzc = (int)((nk + c - 1) / c);
            const long wgs = tiles * ((nk + kzc - 1) / kzc);
            const long cost = ((wgs + slots - 1) / slots) * (kzc + OV);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_c = c; }
        }
        kz = (int)((nk + best_c - 1) / best_c);
    } else if (kz > 90) {
        static const int per_cu = workgroups_per_cu((const void *)k_pt_sweepN<T, NL, WX, WY, CPT, PF, MINW, FOLD>, 64 * WX * WY);
        const long slots = (long)max(8, device_cus() - a.cus_off) * per_cu, tiles = (long)ntx * nty;
        const int want = kz - 90;
        const int cmax = max(1, nk / (6 * NL));              // short chunks are mostly pipeline fill (2(NL−1) steps each)
        long best_c = 1;
        double best_fill = 0.0;
        for (int m = want; m <= want + 12; ++m) {
            long c = m * slots / tiles;
            c = c < 1 ? 1 : (c > cmax ? cmax : c);
            const int kzc = (int)((nk + c - 1) / c);
            const long wgs = tiles * ((nk + kzc - 1) / kzc);
            const double fill = (double)wgs / (double)(((wgs + slots - 1) / slots) * slots);
            if (fill > best_fill + 1e-9) { best_fill = fill; best_c = c; }
            if (fill >= 0.95 || c == cmax) break;
        }
        kz = (int)((nk + best_c - 1) / best_c);
    }
    a.kz = kz;
    const int ntz = (nk + kz - 1) / kz;
    hipLaunchKernelGGL((k_pt_sweepN<T, NL, WX, WY, CPT, PF, MINW, FOLD>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX, WY, 1), 0, s, a,
                       ntx, nty);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !a.no_faces && !FOLD) e = launch_faces<T>(s, a);
    return e;
    }
}

// =========================================================================================================
// k_pt_sweepD  —  the NL-iteration pass with the planes of P⁰ staged by LDS-DMA  (round 4, VERDICT r3 #1b)
//
// Same levels, same trapezoid, same arithmetic per cell as k_pt_sweepN (bit-identical to NL k_pt_sweep launches); what changes
// is how P⁰ reaches the stencils.  k_pt_sweepN carries four planes of P⁰ per column in registers (k1−1, k1, k1+1 and the staging
// copy of k1+2) plus the halo ring twice, and republishes every plane into LDS with ds_write.  Here the whole (TY+2)×66 image of
// a plane — tile, halo rows and halo columns alike — is one lane-linear run of 4-byte words in a ring of NSL LDS slots, filled
// by `global_load_lds_dword` (each lane's source address is precomputed once: clamped row / column of the word it owns; the
// destination is wave-uniform base + 4·lane, so no register ever holds the data), and level 1 reads c, w, e, s, n of plane k1
// and t of plane k1+1 from LDS; only b = P⁰[k1−1] stays in a register (it is last step's c).  Per column that is 1 live value
// of P⁰ instead of 4 and no halo registers: the 768-thread 64×24 shape loses its spills, and fp64 fits a 1024-thread 64×32
// tile (four waves per SIMD, 73.6 % of the lanes produce output against 68 %).
//
// Ordering (cdna_hip_programming.md §5 "Pipelining across barriers"; MI355X_MICROARCH.md item 7: nothing orders a ds_read
// behind a pending LDS-DMA except the issuing wave's vmcnt and a barrier): the DMA is inline asm, invisible to hipcc's
// s_waitcnt bookkeeping (a builtin DMA makes it drain vmcnt(0) before every later ds_read).  In step s the words of plane
// k1+NSL−1 are issued AFTER level 1 — so that no hidden operation is younger than the d⁰/∇V loads whose results level 1 of the
// next step waits for with hipcc's own counted vmcnt — followed by this step's 2·CPT d⁰/∇V loads; the step ends with
//     s_waitcnt vmcnt(N) lgkmcnt(0);  s_barrier        N = 2·CPT (NSL = 3)  or  CW + 4·CPT (NSL = 4: one more step in flight)
// which retires the DMA of plane k1+2 whatever number of output stores followed it (N counts only operations that are ALWAYS
// issued; stores on top make the wait stricter, never looser), and the plane is first read in step s+1, behind that barrier.
// WAR: a slot is refilled in the step after its last readers passed lgkmcnt(0) + barrier.
// =========================================================================================================
__device__ __forceinline__ void glds_word(const void *plane, unsigned byte_off, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(lds_byte), "s"(plane)
                 : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm_lgkm_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(N) : "memory");
}

template <class T, int NL, int WY, int CPT, int NSL, int UNR, int OPT>
__global__ __launch_bounds__(64 * WY) void k_pt_sweepD(SweepArgs<T> a, int ntx, int nty)
{
    static_assert(NL >= 2 && NL <= 5, "levels");
    static_assert(NSL == 3 || NSL == 4, "slots");
    constexpr bool CLDS = (OPT & 1) != 0;    // levels ≥ 2 read their centre value from the LDS plane too: one register pair per level and row less
    constexpr int TX = 64, TY = CPT * WY, PX = TX + 2, PR = TY + 2, OV = 2 * (NL - 1);
    constexpr int W = (int)sizeof(T) / 4;                       // DMA words per element
    constexpr int NW = PR * PX * W, NCH = (NW + 63) / 64;       // words / 64-word chunks of one plane image
    constexpr int CW = (NCH + WY - 1) / WY;                     // chunks per wave (the last one may be missing: wave-uniform test)
    constexpr int SLOT = NCH * 64 / W;                          // elements per slot
    constexpr int LEAD = NSL - 1;                               // the plane issued in step s is k1 + LEAD
    constexpr int WAITN = NSL == 3 ? 2 * CPT : CW + 4 * CPT;
    static_assert(TX > OV + 2 && TY > OV + 2, "tile too small for this many levels");
    __shared__ T L0[NSL][SLOT];              // ring of P⁰ plane images: element (row + 1) * PX + (column + 1)
    __shared__ T LN[NL - 1][2][TY * TX];     // planes of P¹ … P^{NL−1}
#if NS3D_HAS_SLOW_PATH
    __shared__ int Lbad;
#endif
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const Geo<T> &g = a.g;
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, rem = nb & 7, xcd = b & 7, loc = b >> 3;
    const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
    const int tx_t = a.tx0 + tile % ntx, ty_t = a.ty0 + (tile / ntx) % nty, tz_t = tile / (ntx * nty);     // ntx × nty: the launch's tile window
    const int ox = 1 + tx_t * (TX - OV), oy = 1 + ty_t * (TY - OV);
    const int kb = a.k0 + tz_t * a.kz;
    const int ke = min(kb + a.kz, a.k1);
    if (kb >= ke) return; // workgroup-uniform, before any barrier

    const int lx = threadIdx.x, wy = threadIdx.y;
    const int wyu = __builtin_amdgcn_readfirstlane(wy);
    const int gi = ox + lx;
    const int ci = min(gi, nx - 1), cii = min(gi, nx - 2);
    const idx_t sz = (idx_t)nx * ny;
    const idx_t dsz = (idx_t)(nx - 2) * (ny - 2);
    const bool xlo_adj = (gi == 1), xhi_adj = (gi == nx - 2);
    const bool tile_x_lo = (ox <= 1), tile_x_hi = (ox + TX - 1 >= nx - 2);
    const bool tile_y_lo = (oy <= 1), tile_y_hi = (oy + TY - 1 >= ny - 2);
    const bool tile_on_xy_face = tile_x_lo || tile_x_hi || tile_y_lo || tile_y_hi;
    const bool tile_on_x_face = tile_x_lo || tile_x_hi;
    const bool x_out = (gi <= nx - 2) && (lx >= NL - 1 || tile_x_lo) && (lx <= TX - NL || tile_x_hi);

    // rows are wave-uniform (one wave = one threadIdx.y): the row part of every offset is scalar, only the column part is per lane
    int roff[CPT], doff[CPT];
    bool outr[CPT];
#pragma unroll
    for (int r = 0; r < CPT; ++r) {
        const int lr = wyu * CPT + r, gj = oy + lr;
        const int cjj = min(gj, ny - 2);
        roff[r] = cjj * nx;
        doff[r] = (cjj - 1) * (nx - 2) - 1;
        outr[r] = (gj <= ny - 2) && (lr >= NL - 1 || tile_y_lo) && (lr <= TY - NL || tile_y_hi);
    }
    bool act[NL + 1];                        // wave-uniform level skipping, as in k_pt_sweepN
    act[0] = act[1] = true;
#pragma unroll
    for (int l = 2; l <= NL; ++l) act[l] = (wyu * CPT + CPT - 1 >= l - 1 || tile_y_lo) && (wyu * CPT <= TY - l || tile_y_hi);

    // the words of a plane image this lane moves: chunk c = wy + j·WY, word c·64 + lane → element (row, col) of the image →
    // cell (oy−1+row, ox−1+col) clamped into the plane (the clamped copies are the cells no valid stencil reads)
    unsigned goff[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) {
        const int wi = (wyu + j * WY) * 64 + lx;
        const int el = min(wi / W, PR * PX - 1), h = wi % W;
        const int row = el / PX, col = el - row * PX;
        const unsigned gj = (unsigned)min(oy - 1 + row, ny - 1), gc = (unsigned)min(ox - 1 + col, nx - 1);
        goff[j] = ((gj * (unsigned)nx + gc) * (unsigned)W + (unsigned)h) * 4u;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(&L0[0][0]);
    const T *__restrict__ P = a.Pin;
    auto issue_plane = [&](int kplane, int slot) __attribute__((always_inline)) {
        const T *__restrict__ Pn = P + (idx_t)min(max(kplane, 0), nz - 1) * sz;
        const unsigned base = lds0 + (unsigned)slot * (unsigned)(SLOT * sizeof(T)) + (unsigned)wyu * 256u;
#pragma unroll
        for (int j = 0; j < CW; ++j)
            if ((j + 1) * WY <= NCH || wyu + j * WY < NCH) glds_word(Pn, goff[j], base + (unsigned)(j * WY) * 256u);
    };

    const T *__restrict__ RHS = a.RHS;
    const T *__restrict__ Din = a.Din;
    T *__restrict__ D = a.D;

    const int idx00 = (wyu * CPT + 1) * PX + lx + 1;     // this thread's first cell in a plane image; row r is r·PX further
    const int lnb = wyu * CPT * TX;                      // its first row in a plane of Pˡ
    const int lxm = max(lx - 1, 0), lxp = min(lx + 1, TX - 1);

    T p0m[CPT];                               // P⁰ plane k1−1 (last step's c)
    T pm[NL - 1][CPT], pc[NL - 1][CPT];       // Pˡ planes kℓ₊₁−1, kℓ₊₁
    T dc[NL - 1][CPT];                        // dˡ[kℓ₊₁]
    T rr[NL][CPT];                            // ∇V[k1] … ∇V[k_NL]
    T d0[CPT];                                // d⁰[k1]
    bool bad = false;
    (void)bad;
#if NS3D_HAS_SLOW_PATH
    if (a.bc_kind == NS3D_BC_GPU) {
        const T hmin = (a.rho_g * (T)1.5) * g.dz, hmax = (a.rho_g * ((T)(nz - 2) + (T)0.5)) * g.dz;
        bad = !(val_ok<T>(hmin) && val_ok<T>(hmax) && val_ok<T>(hmin + (T)100) && val_ok<T>(hmax + (T)100));
    } else if (a.owns_outlet) bad = !val_ok<T>(a.outlet_val);
    if (wy == 0 && lx == 0) Lbad = 0;
#endif
    const int kfirst = kb - (NL - 1);         // plane of level 1 at step 0
    // ---- prologue: planes kfirst … kfirst+LEAD−1 into the ring, plane kfirst−1 and the streams of plane kfirst into registers ----
    {
#pragma unroll
        for (int q0 = 0; q0 < LEAD; ++q0) issue_plane(kfirst + q0, q0);
        const T *__restrict__ Pm = P + (idx_t)min(max(kfirst - 1, 0), nz - 1) * sz;
        const int ka = min(max(kfirst, 1), nz - 2);
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
            const int gj = min(oy + wyu * CPT + r, ny - 1);
            p0m[r] = Pm[gj * nx + ci];
            d0[r] = ld_stream<T, true>(Din + ((idx_t)(ka - 1) * dsz + doff[r]) + cii);
            rr[0][r] = ld_stream<T, true>(RHS + ((idx_t)ka * sz + roff[r]) + cii);
#pragma unroll
            for (int l = 0; l < NL - 1; ++l) { pm[l][r] = pc[l][r] = dc[l][r] = (T)0; rr[l + 1][r] = (T)0; }
        }
    }
    wait_vm_lgkm_barrier<0>();

    const int nsteps = (ke - kb) + OV;
    int cur = 0;
    int sc = 0, sn = 1, s2 = NSL == 4 ? 2 : 0, sl = NSL - 1;   // slots of planes k1, k1+1, (k1+2,) and the one being filled
    auto step = [&](const int s, auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const int k1 = kfirst + s;
        T d0n[CPT], r0n[CPT];
        T fresh[CPT], dnew[CPT];
        // ---------------- level 1 at plane k1: everything of P⁰ but the plane below comes from the ring ----------------
        {
            const T *__restrict__ lc = &L0[0][0] + sc * SLOT;
            const T *__restrict__ lt = &L0[0][0] + sn * SLOT;
            T c[CPT], t[CPT], w[CPT], e[CPT];
#pragma unroll
            for (int r = 0; r < CPT; ++r) { c[r] = lc[(idx00 + r * PX)]; w[r] = lc[(idx00 + r * PX) - 1]; e[r] = lc[(idx00 + r * PX) + 1]; t[r] = lt[(idx00 + r * PX)]; }
            const T s_lo = lc[idx00 - PX], n_hi = lc[(idx00 + (CPT - 1) * PX) + PX];
#if NS3D_HAS_SLOW_PATH
#pragma unroll
            for (int r = 0; r < CPT; ++r) {
                bad |= !val_ok<T>(t[r]); bad |= !val_ok<T>(w[r]); bad |= !val_ok<T>(e[r]);
                if (EDGE) { bad |= !val_ok<T>(c[r]); bad |= !val_ok<T>(p0m[r]); }
            }
            bad |= !val_ok<T>(s_lo); bad |= !val_ok<T>(n_hi);
            bool slow = (__builtin_amdgcn_readfirstlane(Lbad) != 0) || (__builtin_amdgcn_ballot_w64(bad) != 0);   // wave-uniform
#else
            const bool slow = false;
#endif
            auto level1 = [&](auto slow_tag) {
#pragma unroll
                for (int r = 0; r < CPT; ++r) {
                    const T sv = r == 0 ? s_lo : c[r - 1 < 0 ? 0 : r - 1];
                    const T nv = r == CPT - 1 ? n_hi : c[r + 1 > CPT - 1 ? CPT - 1 : r + 1];
                    const T res = decltype(slow_tag)::value
                                      ? poisson_rhs_slow<T>(c[r], w[r], e[r], sv, nv, p0m[r], t[r], rr[0][r], a.rho_dt, g)
                                      : poisson_rhs_nochk<T>(c[r], w[r], e[r], sv, nv, p0m[r], t[r], rr[0][r], a.rho_dt, g);
                    dnew[r] = d0[r] * a.one_m_damp + a.dtau * res;
                    fresh[r] = c[r] + a.dtau * dnew[r];
                }
            };
            if (__builtin_expect(slow, 0)) level1(std::true_type{});
            else level1(std::false_type{});
#pragma unroll
            for (int r = 0; r < CPT; ++r) p0m[r] = c[r];
            // ---------------- the plane LEAD steps ahead → the free slot; d⁰ / ∇V of plane k1+1 → registers ----------------
            issue_plane(k1 + LEAD, sl);
            {
                const int ka = min(max(k1 + 1, 1), nz - 2);
#pragma unroll
                for (int r = 0; r < CPT; ++r) {
                    d0n[r] = ld_stream<T, true>(Din + ((idx_t)(ka - 1) * dsz + doff[r]) + cii);
                    r0n[r] = ld_stream<T, true>(RHS + ((idx_t)ka * sz + roff[r]) + cii);
                }
            }
            // ---------------- levels 2 … NL, each one plane behind the previous one ----------------
#pragma unroll
            for (int l = 2; l <= NL; ++l) {
                const int kl = k1 - (l - 1);
                T *__restrict__ npub = LN[l - 2][cur ^ 1];
#ifdef NS3D_SCHED_FENCE      // keep the scheduler from interleaving the levels of the branch-free bulk step (register pressure)
                __builtin_amdgcn_sched_barrier(0);
#endif
                if (act[l - 1]) {
#pragma unroll
                    for (int r = 0; r < CPT; ++r) npub[lnb + r * TX + lx] = fresh[r];
                }
#if NS3D_HAS_SLOW_PATH
#pragma unroll
                for (int r = 0; r < CPT; ++r) bad |= !val_ok<T>(fresh[r]);
                slow = slow || (__builtin_amdgcn_ballot_w64(bad) != 0);
#endif
                T out_p[CPT], out_d[CPT], ccv[CPT];
                // centre values: also in the pipeline-fill steps — the one read in the step before level l first runs is that
                // step's plane below (what it reads before the plane was ever published is overwritten before anything uses it)
#pragma unroll
                for (int r = 0; r < CPT; ++r) ccv[r] = CLDS ? (act[l] ? LN[l - 2][cur][lnb + r * TX + lx] : (T)0) : pc[l - 2][r];
                if ((!EDGE || s >= 2 * (l - 1)) && act[l]) {
                    const T *__restrict__ ll = LN[l - 2][cur];
                    const bool zlo = EDGE && (kl == 1), zhi = EDGE && (kl == nz - 2);
                    // The boundary rule, lean (round 4): the x-face value — copy, outlet value or gpu.jl's hydrostatic profile, a
                    // function of the plane only — is formed once per level; per row an interior tile pays one scalar branch; rows
                    // are wave-uniform, so the y-face tests are scalar too.
                    T xlo_v = (T)0, xhi_v = (T)0;
                    bool xlo_copy = true, xhi_copy = true;
                    if (tile_on_x_face) {
                        if (a.bc_kind == NS3D_BC_GPU) {      // gpu.jl:258-259 at plane kl (xface_val)
                            xhi_v = (a.rho_g * ((T)(nz - (kl + 1)) + (T)0.5)) * g.dz;
                            xlo_v = xhi_v + (T)100;
                            xlo_copy = xhi_copy = false;
                        } else if (a.owns_outlet) { xhi_v = a.outlet_val; xhi_copy = false; }
                    }
                    auto level = [&](auto slow_tag) {
#pragma unroll
                        for (int r = 0; r < CPT; ++r) {
                            const int lr = wyu * CPT + r;
                            const T cc = ccv[r];
                            T ww = ll[lnb + r * TX + lxm], ee = ll[lnb + r * TX + lxp];
                            T sv = r == 0 ? ll[max(lr - 1, 0) * TX + lx] : ccv[r - 1 < 0 ? 0 : r - 1];
                            T nv = r == CPT - 1 ? ll[min(lr + 1, TY - 1) * TX + lx] : ccv[r + 1 > CPT - 1 ? CPT - 1 : r + 1];
                            T bv = pm[l - 2][r], tv = fresh[r];
                            if (tile_on_xy_face) {
                                if (tile_on_x_face) {
                                    if (xlo_adj) ww = xlo_copy ? cc : xlo_v;
                                    if (xhi_adj) ee = xhi_copy ? cc : xhi_v;
                                }
                                const int gjf = oy + lr;
                                if (gjf == 1) sv = cc;
                                if (gjf == ny - 2) nv = cc;
                            }
                            if constexpr (EDGE) {
                                if (zlo) bv = cc;
                                if (zhi) tv = cc;
                            }
                            const T res = decltype(slow_tag)::value
                                              ? poisson_rhs_slow<T>(cc, ww, ee, sv, nv, bv, tv, rr[l - 1][r], a.rho_dt, g)
                                              : poisson_rhs_nochk<T>(cc, ww, ee, sv, nv, bv, tv, rr[l - 1][r], a.rho_dt, g);
                            out_d[r] = dc[l - 2][r] * a.one_m_damp + a.dtau * res;
                            out_p[r] = cc + a.dtau * out_d[r];
                        }
                    };
                    if (__builtin_expect(slow, 0)) level(std::true_type{});
                    else level(std::false_type{});
                    if (l == NL) {
                        T *__restrict__ Dk = D + (idx_t)(kl - 1) * dsz;
#pragma unroll
                        for (int r = 0; r < CPT; ++r) {
                            if (outr[r] && x_out) {
                                st_stream<T, true>(Dk + doff[r] + cii, out_d[r]);
                                const int gj = oy + wyu * CPT + r;
                                T *__restrict__ po = a.Pout + ((idx_t)kl * sz + gj * nx) + gi;
                                st_stream<T, true>(po, out_p[r]);
                                if (tile_on_x_face) {   // the x-face cell beside it shares its cache line: one more store
                                    if (xlo_adj) po[-1] = xlo_copy ? out_p[r] : xlo_v;
                                    if (xhi_adj) po[1] = xhi_copy ? out_p[r] : xhi_v;
                                }
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < CPT; ++r) { out_p[r] = (T)0; out_d[r] = (T)0; }
                }
#pragma unroll
                for (int r = 0; r < CPT; ++r) {
                    if constexpr (CLDS) {
                        // the centre value just read is next step's plane below — where the level did not run (pipeline fill, a
                        // skipped wave) nothing valid reads it
                        pm[l - 2][r] = ccv[r];
                    } else {
                        pm[l - 2][r] = pc[l - 2][r]; pc[l - 2][r] = fresh[r];
                    }
                    dc[l - 2][r] = dnew[r];
                    fresh[r] = out_p[r]; dnew[r] = out_d[r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
#pragma unroll
            for (int l = NL - 1; l >= 1; --l) rr[l][r] = rr[l - 1][r];
            d0[r] = d0n[r];
            rr[0][r] = r0n[r];
        }
#if NS3D_HAS_SLOW_PATH
        if (bad) Lbad = 1;
#endif
        wait_vm_lgkm_barrier<WAITN>();
        cur ^= 1;
        const int tfree = sc;
        if constexpr (NSL == 3) { sc = sn; sn = sl; sl = tfree; }
        else { sc = sn; sn = s2; s2 = sl; sl = tfree; }
    };
    // UNR written-out bulk steps per trip (no edge tests, ring rotations renamed away) as in k_pt_sweepN; UNR = 1 keeps the one
    // general step form for every plane: fewer registers (the 1024-thread fp64 shape: 119 and no spill, against 128 + 58 spilled)
    static_assert(UNR == 1 || UNR == 2 || UNR == 4, "UNR");
    int s = 0;
    if constexpr (UNR > 1) {
        const int hot_lo = max(OV, NL + 1 - kfirst), hot_hi = min(nsteps, nz - 1 - kfirst);
        const int h0 = min(nsteps, (hot_lo + 1) & ~1);
        for (; s + 2 <= h0; s += 2) {
            step(s, std::true_type{});
            step(s + 1, std::true_type{});
        }
        if (s == h0)
            for (; s + UNR <= hot_hi; s += UNR) {
#pragma unroll
                for (int u = 0; u < UNR; ++u)
#ifdef NS3D_D_BULK_EDGE
                    step(s + u, std::true_type{});
#else
                    step(s + u, std::false_type{});
#endif
            }
    }
    for (; s < nsteps; ++s) step(s, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA words still in flight land before the LDS is released
}

template <class T, int NL, int WY, int CPT, int NSL, int UNR, int OPT>
static hipError_t launch_sweepD(hipStream_t s, SweepArgs<T> &a, int kz)
{
    constexpr int TX = 64, TY = CPT * WY, OV = 2 * (NL - 1), W = (int)sizeof(T) / 4;
    constexpr size_t slot = (size_t)(((TY + 2) * (TX + 2) * W + 63) / 64) * 256;
    constexpr size_t lds = NSL * slot + 2ul * (NL - 1) * TY * TX * sizeof(T) + 64;
    if constexpr (!(TX > OV + 2 && TY > OV + 2 && lds <= 160ul * 1024)) {
        return hipErrorInvalidValue;
    } else {
        const int nk = a.k1 - a.k0;
            if (a.nx < 4 || a.ny < 4 || (size_t)a.nx * a.ny * sizeof(T) >= (1ull << 32)) return hipErrorInvalidValue;   // 32-bit in-plane byte offsets
        int ntx = max(1, (a.nx - 2 - OV + (TX - OV) - 1) / (TX - OV)), nty = max(1, (a.ny - 2 - OV + (TY - OV) - 1) / (TY - OV));
        if (!apply_tile_window<T>(a, TX, TY, OV, ntx, nty)) return hipSuccess;
        if (kz <= 0 || kz > 90) {
            static const int per_cu = workgroups_per_cu((const void *)k_pt_sweepD<T, NL, WY, CPT, NSL, UNR, OPT>, 64 * WY);
            const long slots = (long)max(8, device_cus() - a.cus_off) * per_cu, tiles = (long)ntx * nty;
            const int cmax = max(1, nk / (6 * NL));
            long best_c = 1, best_cost = -1;
            for (long c = 1; c <= cmax && c <= 64; ++c) {
                const int kzc = (int)((nk + c - 1) / c);
                const long wgs = tiles * ((nk + kzc - 1) / kzc);
                const long cost = ((wgs + slots - 1) / slots) * (kzc + OV);
                if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_c = c; }
                if (kz > 90) break;             // 91: one chunk per tile column
            }
            kz = (int)((nk + best_c - 1) / best_c);
        }
        a.kz = kz;
        const int ntz = (nk + kz - 1) / kz;
        hipLaunchKernelGGL((k_pt_sweepD<T, NL, WY, CPT, NSL, UNR, OPT>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX, WY, 1), 0, s, a, ntx, nty);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && !a.no_faces) e = launch_faces<T>(s, a);
        return e;
    }
}

// (Round 3 tried the opposite of overlapped tiles — tiles WITHOUT overlap whose workgroups exchange the edges of every intermediate
// plane through the L2, cooperative launch, level ℓ three planes behind level ℓ−1 so that an edge has two steps to cross the
// chip: bit-identical, 2.16 ms per four-iteration pass at 512³ against 1.45 for k_pt_sweepN, 1.36 with the exchange compiled
// out.  Not kept: profiles/r3_coop_ab.log, DESIGN.md §4.3, commit "experiment: k_pt_coop".)
// `nlev` fused PT iterations (Pin,Din) → (Pout,Dout) for the output planes [k0,k1).  variant = shape*100 + kz:
// shape 1: 64×32 columns (one wave wide, 8 high, 4 rows per thread), 2: 128×16, 6: 64×16 with 256-thread workgroups (two per
// CU); +10: loads of the next step issued before level 1 (EARLY); 22: fp32 64×48 with 768 threads; 23 / 28 (EARLY / not): 64×24
// with 768 threads and two rows per thread; 24: fp32 64×32 with 1024 threads; kz: planes per z-chunk, 0 = chosen by
// launch_sweepN, 91…99 = as many chunks as fill kz−90 rounds.  0 = built-in choice.  (Shapes that lost every A/B — 256×8, six
// rows per thread, one wave per SIMD, 64×20/64×24 with 256 threads, 64×48 with 1024 — are no longer instantiated.)
template <class T>
hipError_t pt_sweepn(hipStream_t s, int nlev, int variant, const T *Pin, T *Pout, const T *Din, T *Dout, const T *RHS,
                     const ns3d_pt_params &p, int k0, int k1, int pass_flags, const ns3d_tile_window *win)
{
    SweepArgs<T> a;
    a.win = win; a.tx0 = a.ty0 = 0;
    a.Pin = Pin; a.Pout = Pout; a.D = Dout; a.Din = Din; a.RHS = RHS;
    a.g = make_geo<T>(p.dx, p.dy, p.dz);
    a.rho_dt = (T)p.rho / (T)p.dt; a.dtau = (T)p.dtau; a.one_m_damp = (T)1.0 - (T)p.damp;
    a.outlet_val = (T)p.outlet_val; a.rho_g = (T)p.rho * (T)p.g;
    a.nx = p.nx; a.ny = p.ny; a.nz = p.nz;
    a.bc_kind = p.bc_kind; a.owns_outlet = p.owns_outlet; a.zlo_halo = 0; a.zhi_halo = 0;
    a.k0 = k0; a.k1 = k1; a.kz = 1;
    a.no_faces = (pass_flags & NS3D_PASS_SKIP_FACES) ? 1 : 0;
    a.cus_off = ((pass_flags >> 8) & 0xff) * 8;
    if (k1 <= k0) return hipSuccess;
    int shape = variant / 100, kz = variant % 100;
    // built-in: 64×32 columns, next step's loads issued before level 1 (measured best at 512³ for three levels); four levels
    // want three or four waves per SIMD: 64×24 columns with 768 threads in fp64 (loads between the levels: measured best once the
    // steps were written out), 64×32 with 1024 threads in fp32
    if (variant == 0) { shape = nlev >= 4 ? (sizeof(T) == 8 ? 28 : 24) : 11; kz = 0; }
#define NS3D_SWN(NLV, WXV, WYV, CPTV, PFV) return launch_sweepN<T, NLV, WXV, WYV, CPTV, PFV>(s, a, kz)
#define NS3D_SWN_SHAPES(NLV)                                                                                \
    switch (shape) {                                                                                        \
    case 1: NS3D_SWN(NLV, 1, 8, 4, false);                                                                  \
    case 2: NS3D_SWN(NLV, 2, 4, 4, false);                                                                  \
    case 6: NS3D_SWN(NLV, 1, 4, 4, false);   /* 256-thread workgroups: two (or three) per CU */             \
    case 16: NS3D_SWN(NLV, 1, 4, 4, true);                                                                  \
    case 22: if constexpr (sizeof(T) == 4) { NS3D_SWN(NLV, 1, 12, 4, true); } else return hipErrorInvalidValue; /* fp32: 64×48, 768 threads = three waves per SIMD */ \
    case 23: NS3D_SWN(NLV, 1, 12, 2, true);  /* 64×24, 768 threads, two rows per thread: three waves per SIMD */ \
    case 24: if constexpr (sizeof(T) == 4 || NS3D_SHAPE24_F64) { NS3D_SWN(NLV, 1, 16, 2, true); } else return hipErrorInvalidValue; /* fp32: 64×32, 1024 threads = four waves per SIMD (fp64 spills: A/B with -DNS3D_SHAPE24_F64=1) */ \
    case 25: if constexpr (sizeof(T) == 4) { NS3D_SWN(NLV, 1, 16, 2, 2); } else return hipErrorInvalidValue;   /* fp32 A/B: shape 24 with the loads two steps ahead */ \
    case 28: if (fold_wanted<T>(a)) return launch_sweepN<T, NLV, 1, 12, 2, false, 1, true>(s, a, kz);       /* small grids: boundary cells folded in */ \
             NS3D_SWN(NLV, 1, 12, 2, false);                                                                \
    case 11: NS3D_SWN(NLV, 1, 8, 4, true);                                                                  \
    case 12: NS3D_SWN(NLV, 2, 4, 4, true);                                                                  \
    default: return hipErrorInvalidValue;                                                                   \
    }
    // shapes 31…38: k_pt_sweepD — P⁰ through an LDS-DMA ring.  38: 64×32 columns / 1024 threads (four waves per SIMD), two rows per thread,
    // three slots, two written-out steps per trip, centre values of levels ≥ 2 from LDS — the planner's candidate (3800); kept beside it as
    // the A/B evidence of profiles/r4_levelskip_dma_ab.log: 31: the 64×24 / 768-thread shape of k_pt_sweepN's 28xx with DMA staging, 35: the same
    // with a four-slot ring (the DMA one more step ahead), 32: 38 with the one general step form and the centres in registers
    if (shape >= 31 && shape <= 38) {
#define NS3D_SWD(NLV)                                                                                        \
        switch (shape) {                                                                                     \
        case 31: return launch_sweepD<T, NLV, 12, 2, 3, 4, 0>(s, a, kz);                                     \
        case 32: return launch_sweepD<T, NLV, 16, 2, 3, 1, 0>(s, a, kz);                                     \
        case 35: return launch_sweepD<T, NLV, 12, 2, 4, 4, 0>(s, a, kz);                                     \
        case 38: return launch_sweepD<T, NLV, 16, 2, 3, 2, 1>(s, a, kz);                                     \
        default: return hipErrorInvalidValue;                                                                \
        }
        if (nlev == 2) { NS3D_SWD(2) } else if (nlev == 3) { NS3D_SWD(3) } else if (nlev == 4) { NS3D_SWD(4) }
        return hipErrorInvalidValue;
#undef NS3D_SWD
    }
    switch (nlev) {
    case 2: NS3D_SWN_SHAPES(2)
    case 3: NS3D_SWN_SHAPES(3)
    case 4: NS3D_SWN_SHAPES(4)
    case 5:     // a fifth level only where registers are left: fp32 on 1024-thread workgroups (128 registers, four waves per SIMD).
                // 512³ fp32: 0.918 ms per pass against 0.778 for four = +6 % per iteration, 1024³ +9.5 %; fp64 spills (57 registers:
                // 3.2–3.7 ms per pass against 1.39), six levels in fp32 too (44: 1.89 ms) — profiles/r3_pace_order_ab.log
        if constexpr (sizeof(T) == 4) {
            if (shape == 24) NS3D_SWN(5, 1, 16, 2, true);
            if (shape == 25) NS3D_SWN(5, 1, 16, 2, 2);
        }
        return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
    }
#undef NS3D_SWN_SHAPES
#undef NS3D_SWN
}

// Two fused PT iterations (Pin,Din) → (Pout,Dout) for the output planes [k0,k1) (k0 = 1, k1 = nz-1: the whole slab).
template <class T>
hipError_t pt_sweep2(hipStream_t s, int variant, const T *Pin, T *Pout, const T *Din, T *Dout, const T *RHS,
                     const ns3d_pt_params &p, int k0, int k1, int pass_flags, const ns3d_tile_window *win)
{
    SweepArgs<T> a;
    a.win = win; a.tx0 = a.ty0 = 0;
    a.Pin = Pin; a.Pout = Pout; a.D = Dout; a.Din = Din; a.RHS = RHS;
    a.g = make_geo<T>(p.dx, p.dy, p.dz);
    a.rho_dt = (T)p.rho / (T)p.dt; a.dtau = (T)p.dtau; a.one_m_damp = (T)1.0 - (T)p.damp;
    a.outlet_val = (T)p.outlet_val; a.rho_g = (T)p.rho * (T)p.g;
    a.nx = p.nx; a.ny = p.ny; a.nz = p.nz;
    a.bc_kind = p.bc_kind; a.owns_outlet = p.owns_outlet; a.zlo_halo = 0; a.zhi_halo = 0;
    a.k0 = k0; a.k1 = k1; a.kz = 1;
    a.no_faces = (pass_flags & NS3D_PASS_SKIP_FACES) ? 1 : 0;
    a.cus_off = ((pass_flags >> 8) & 0xff) * 8;
    if (k1 <= k0) return hipSuccess;
    int shape = variant / 100;
    int kz = variant % 100;
    const int nxi = p.nx - 2;
    if (variant == 0) {
        // Tile shape by grid (measured: profiles/r1b_shapes.log and later sweeps).  256-wide tiles (12 rows in fp64, 8 in fp32)
        // are fastest on large grids whose rows fill them (two columns of overlap per 256: nx = 512, …);
        // otherwise 128×8 tiles with two rows per thread.  Explicit variants (shape·100 + kz) override; the context's
        // first-use tuning (ns3d_api.cpp) normally replaces this rule by a measurement.
        const long long cells = (long long)p.nx * p.ny * (k1 - k0);      // of this launch (z-slab ranks sweep thin seam ranges)
        const int ntx256 = (p.nx - 4 + 253) / 254 > 0 ? (p.nx - 4 + 253) / 254 : 1;
        const bool rows_fill_256 = nxi >= 200 && (double)nxi / (256.0 * ntx256) >= 0.95;
        if (rows_fill_256 && cells >= 64ll * 1000 * 1000) shape = sizeof(T) == 8 ? 13 : 11;   // 256×12 (fp64) / 256×8 (fp32)
        else shape = (nxi > 64 && cells >= 8ll * 1000 * 1000) ? 8 : 7;   // 128×8 / 64×16 with two workgroups per CU; small
                                                                         // grids want the many workgroups of the narrow tile
        kz = 0;                                 // z-chunks that fill whole rounds of the chip (launch_sweep2)
    }
    switch (shape) {
    case 1: return launch_sweep2<T, 4, 2, 4, true>(s, a, kz);   // 256 x 8
    case 2: return launch_sweep2<T, 2, 4, 4, true>(s, a, kz);   // 128 x 16
    case 3: return launch_sweep2<T, 1, 8, 4, true>(s, a, kz);   //  64 x 32
    case 6: return launch_sweep2<T, 2, 2, 4, true>(s, a, kz);   // 128 x 8 (256 threads)
    // boundary cells by separate launches (SEPF): leaner hot kernel, larger tiles / more workgroups per CU
    case 7: return launch_sweep2<T, 1, 8, 2, true, 4, true>(s, a, kz);  //  64 x 16, ≤128 VGPRs: two workgroups per CU
    case 8: return launch_sweep2<T, 2, 4, 2, true, 4, true>(s, a, kz);  // 128 x 8,  ≤128 VGPRs
    case 9: return launch_sweep2<T, 1, 8, 4, true, 1, true>(s, a, kz);  //  64 x 32
    case 11: return launch_sweep2<T, 4, 2, 4, true, (sizeof(T) == 4 ? 4 : 1), true>(s, a, kz); // 256 x 8 (fp32: four waves per SIMD = ≤128 VGPRs = two workgroups per CU)
    case 12: return launch_sweep2<T, 2, 4, 6, true, 1, true>(s, a, kz); // 128 x 24
    case 13: return launch_sweep2<T, 4, 2, 6, true, 1, true>(s, a, kz); // 256 x 12
    case 19: return launch_sweep2<T, 4, 4, 2, true, 4, true>(s, a, kz); // 256 x 8, 1024 threads with two rows each: four waves per SIMD
    default: // shape by row length: the widest tile whose overlap-2 tiling wastes the fewest lanes
        if (nxi > 128) return launch_sweep2<T, 4, 2, 4, true>(s, a, kz);
        if (nxi > 64) return launch_sweep2<T, 2, 4, 4, true>(s, a, kz);
        return launch_sweep2<T, 1, 8, 4, true>(s, a, kz);
    }
}

// =========================================================================================================
// Small grids: n PT iterations in ONE launch  —  k_pt_persist  (round 3)
//
// Below ≈1 M cells an iteration is a 3–5 µs launch and nothing else (config A, 63×38×38: 4.2 µs per iteration in blocks of 37,
// 3.4 in blocks of 370, replayed HIP graph or not).  Here the whole grid stays on the chip for the n iterations of a
// residual-check block: one cell per thread, a workgroup owns 64×BY×BZ cells (P with a one-cell halo in LDS, double-buffered;
// d and ∇V in registers), the workgroups are resident together (cooperative launch, at most two to a CU) and after every
// iteration hand the cells on their faces to the neighbouring workgroups through a small exchange area in global memory:
// value/key word pairs moved with relaxed agent-scope atomics, accepted only when the pair carries the key of the iteration
// waited for — no flag, no fence, one hop (0.4–0.55 µs, tools/ab/signal_latency.hip).  Same expression tree per cell as the
// single sweep, and the first iteration reads the domain's face cells as the caller left them: bit-identical to n k_pt_sweep
// launches.  Waits are bounded: a neighbour that never arrives sets an error word, the grid drains and the result is poisoned
// with NaN (NS3D_COOP_CHECK=1: the launch returns an error instead).
//
// Measured (profiles/r3_persist_ab.log, fp64 strict, µs per iteration, launches → this kernel): 30×18×18 2.67 → 2.08,
// 40×24×24 2.81 → 2.19, 63×38×38 3.37 → 2.67, 66×50×50 4.05 → 3.25 in blocks of 370 (3.5–4.9 → 3.0–4.2 in blocks of 37); two
// workgroups across x or more than ≈170 000 cells lose (130×34×34 3.32 → 4.01, 66×66×66 4.50 → 6.91): what remains per
// iteration is the arithmetic of the CUs' own cells plus a hand-over that costs 1.5–2 µs once four neighbours' skews chain,
// not the 0.5 µs of one hop.  So the automatic choice is narrow (ns3d_api.cpp use_persist) and the gain is 14–22 %; the direct
// solve (ns3d_direct.hip) is the answer where the iteration count itself is the cost.  Smaller workgroups (64×2×2, 64×4×2) are
// tried first — they spread the cells over more CUs; runs of blocks per XCD instead of round-robin were slower.
//
// Round 4, the whole loop of multi.jl:458-471 in one launch (nchk > 0; ns3d_pt_solve where the form applies): compute_res! after
// iteration n is the right-hand side iteration n+1 starts from, so a residual check costs a reduction and no stencil.  Every nchk
// iterations each workgroup leaves max|Rp| of its cells in a pair of its own, workgroup 0 collects the pairs (one per thread),
// leaves the grid's maximum in a result pair, and one thread per workgroup takes the reference's decision (err < ε or not finite)
// from it — all workgroups read the same word, so all stop at the same iteration; the maxima and the iteration count go straight
// to pinned host memory.  No host round trip per check: 63×38×38 in blocks of 37, 3.55 → 2.91 µs per iteration; a time step of
// config A 2.39 → 1.98 ms (profiles/r4_persist_solve_ab.log).  A first version with one atomicMax + one arrival counter per check
// was SLOWER than the launches (2.57 ms): 324 read-modify-writes on one address.
// =========================================================================================================
template <class T>
struct PersistArgs {
    SweepArgs<T> a;
    unsigned long long *H;
    unsigned *err, *err_host;   // device word (read back by every thread at the end) and its pinned-host twin (read by the host)
    unsigned long long epoch;
    unsigned ticket;            // this launch's number: what an expired wait leaves in the error words
    int nbx, nby, nbz, n_iters, xcds;
    int fault;                  // test hook (NS3D_PERSIST_FAULT=1): workgroup 0 never publishes and the waits give up early
    // the whole-solve form (nchk > 0): every nchk iterations max|Rp| over the grid is formed INSIDE the launch and every workgroup
    // takes the reference's decision (multi.jl:466-469) from the same word — see k_pt_persist.  What only the checks need (ε, the
    // scaling of the error, where the host reads the maxima) sits in device memory behind `red`, read by one thread per check: as
    // kernel arguments they would be live in scalar registers across the iteration loop, which is at the 106-SGPR limit already.
    int nchk;
    unsigned long long *red;    // [2b], [2b+1]: workgroup b's maximum as a value/key pair; [2·MAXWG…]: the grid's; [2·MAXCHK…]: PersistCheck
};
struct PersistCheck {
    double eps, err_mul, err_div;
    unsigned long long *res;    // pinned host memory: [0] iterations done, [1] checks made, [2+q] key of check q
};
__device__ __forceinline__ unsigned long long xkey(unsigned long long id) { return (id + 1ull) * 0x9E3779B97F4A7C15ull; }
__device__ __forceinline__ unsigned long long xbits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned long long xbits(float v) { return (unsigned long long)__float_as_uint(v); }
__device__ __forceinline__ void xunbits(unsigned long long b, double &v) { v = __longlong_as_double((long long)b); }
__device__ __forceinline__ void xunbits(unsigned long long b, float &v) { v = __uint_as_float((unsigned)b); }
template <class T> __device__ __forceinline__ void xpublish(unsigned long long *p, T v, unsigned long long key)
{
    const unsigned long long w = xbits(v);
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, w ^ key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class PA> __device__ __forceinline__ unsigned long long xfetch_raw(const unsigned long long *p, unsigned long long key, const PA &pa)
{
    unsigned long long w1 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long w2 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    const unsigned limit = pa.fault ? (1u << 12) : (1u << 22);
    while ((w1 ^ w2) != key) {
        __builtin_amdgcn_s_sleep(1);
        w1 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w2 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spins > limit) {      // the neighbour never arrived (workgroups not resident together): this launch's result is void
            atomicMax(pa.err, pa.ticket);
            __hip_atomic_store(pa.err_host, pa.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    return w1;
}
template <class T, class PA> __device__ __forceinline__ T xfetch(const unsigned long long *p, unsigned long long key, const PA &pa)
{
    T v;
    xunbits(xfetch_raw(p, key, pa), v);
    return v;
}
__device__ __forceinline__ void xpublish_raw(unsigned long long *p, unsigned long long w, unsigned long long key)
{
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, w ^ key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class T, int BY, int BZ>
__global__ __launch_bounds__(64 * BY * BZ) void k_pt_persist(PersistArgs<T> pa)
{
    constexpr int BX = 64, PX = BX + 2, PY = BY + 2, PZ = BZ + 2, R = 4;
    constexpr int FY = BX * BZ, FZ = BX * BY, FX = BY * BZ, FACE = 2 * FY + 2 * FZ + 2 * FX;
    __shared__ T Lp[2][PX * PY * PZ];
    const SweepArgs<T> &a = pa.a;
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const Geo<T> &g = a.g;
    // A/B only (pa.xcds = 8): workgroup blockIdx.x runs on XCD blockIdx.x % 8; give each XCD a contiguous run of blocks
    int b = blockIdx.x;
    if (pa.xcds > 1) {
        const int nb = gridDim.x, x = b % pa.xcds, q = nb / pa.xcds, r = nb % pa.xcds;
        b = x * q + min(x, r) + b / pa.xcds;
    }
    const int bx = b % pa.nbx, by = (b / pa.nbx) % pa.nby, bz = b / (pa.nbx * pa.nby);
    const int lx = threadIdx.x, ly = threadIdx.y, lz = threadIdx.z;
    const int gi = 1 + bx * BX + lx, gj = 1 + by * BY + ly, gk = 1 + bz * BZ + lz;
    const bool act = gi <= nx - 2 && gj <= ny - 2 && gk <= nz - 2;
    const int ci = min(gi, nx - 2), cj = min(gj, ny - 2), ck = min(gk, nz - 2);
    const idx_t sy = nx, sz = (idx_t)nx * ny;
    const idx_t pc = IX3(ci, cj, ck, nx, ny), dc = IX3(ci - 1, cj - 1, ck - 1, nx - 2, ny - 2);
    const int ctr = ((lz + 1) * PY + (ly + 1)) * PX + lx + 1;
    const bool xlo_adj = gi == 1, xhi_adj = gi == nx - 2, ylo_adj = gj == 1, yhi_adj = gj == ny - 2, zlo_adj = gk == 1, zhi_adj = gk == nz - 2;
    // faces this thread sits on that have a neighbouring workgroup behind them
    const bool fxl = lx == 0 && bx > 0, fxh = lx == BX - 1 && bx < pa.nbx - 1;
    const bool fyl = ly == 0 && by > 0, fyh = ly == BY - 1 && by < pa.nby - 1;
    const bool fzl = lz == 0 && bz > 0, fzh = lz == BZ - 1 && bz < pa.nbz - 1;
    const int oyl = lz * BX + lx, oyh = FY + oyl, ozl = 2 * FY + ly * BX + lx, ozh = ozl + FZ, oxl = 2 * FY + 2 * FZ + lz * BY + ly, oxh = oxl + FX;
    unsigned long long *__restrict__ Hme = pa.H + 2 * (size_t)b * R * FACE;
    const int nbxy = pa.nbx * pa.nby;
    T c = a.Pin[pc], d = a.Din[dc];      // dPrdτ goes to a buffer of its own: a launch whose waits expire leaves its inputs intact
    const T rv = a.RHS[pc];
    Lp[0][ctr] = c;
    // the initial halo: the cells beyond my faces as the input array holds them
    if (fxl) Lp[0][ctr - 1] = a.Pin[pc - 1];
    if (fxh) Lp[0][ctr + 1] = a.Pin[pc + 1];
    if (fyl) Lp[0][ctr - PX] = a.Pin[pc - sy];
    if (fyh) Lp[0][ctr + PX] = a.Pin[pc + sy];
    if (fzl) Lp[0][ctr - PX * PY] = a.Pin[pc - sz];
    if (fzh) Lp[0][ctr + PX * PY] = a.Pin[pc + sz];
    // cells on the domain's faces: the first iteration reads them as the caller left them (like the single sweep does); from
    // the second on they hold what the boundary rule made of the cell beside them, which is recomputed instead of stored
    const T f_w = xlo_adj ? a.Pin[pc - 1] : (T)0, f_e = xhi_adj ? a.Pin[pc + 1] : (T)0;
    const T f_s = ylo_adj ? a.Pin[pc - sy] : (T)0, f_n = yhi_adj ? a.Pin[pc + sy] : (T)0;
    const T f_b = zlo_adj ? a.Pin[pc - sz] : (T)0, f_t = zhi_adj ? a.Pin[pc + sz] : (T)0;
    __syncthreads();
    int cur = 0, it = 0, nchecks = 0;
    const int tid = (lz * BY + ly) * BX + lx;
    __shared__ unsigned long long red_w[BY * BZ];
    __shared__ int red_flag;
    int next_check = pa.nchk > 0 ? pa.nchk : -1;
    for (;; ++it) {
        // compute_res! after iteration `it` is the right-hand side the NEXT iteration starts from: the check costs a reduction, no stencil
        const bool check = it == next_check;
        if (it == pa.n_iters && !check) break;
        const T *__restrict__ l = Lp[cur];
        T w = l[ctr - 1], e = l[ctr + 1], sv = l[ctr - PX], nv = l[ctr + PX], bv = l[ctr - PX * PY], tv = l[ctr + PX * PY];
        const bool first = it == 0;
        if (xlo_adj) w = first ? f_w : xface_val<T>(a, false, c, gk);
        if (xhi_adj) e = first ? f_e : xface_val<T>(a, true, c, gk);
        if (ylo_adj) sv = first ? f_s : c;
        if (yhi_adj) nv = first ? f_n : c;
        if (zlo_adj) bv = first ? f_b : c;
        if (zhi_adj) tv = first ? f_t : c;
        const T res = poisson_rhs<T>(c, w, e, sv, nv, bv, tv, rv, a.rho_dt, g);
        if (check) {
            // maximum(abs.(Rp)) (multi.jl:466, NaN-propagating key as k_residual_max) over the grid, with the hand-over's own
            // mechanism (value/key word pairs, no read-modify-write: 324 same-address atomics per check cost ≈40 µs): every workgroup
            // leaves its maximum in its own pair, workgroup 0 collects them — a pair per thread — and leaves the grid's maximum in the
            // result pair, which one thread per workgroup waits for (bounded) and takes the decision of :467-469 from
            next_check += pa.nchk;
            const unsigned long long ckey = xkey(pa.epoch + (unsigned long long)it) ^ 0x5DEECE66Dull;
            unsigned long long key = act ? abs_key((double)res) : 0ull;
            key = wave_max_u64(key);
            if ((tid & 63) == 0) red_w[tid >> 6] = key;
            __syncthreads();
            if (tid == 0) {
                unsigned long long v = 0ull;
#pragma unroll
                for (int q = 0; q < BY * BZ; ++q) v = red_w[q] > v ? red_w[q] : v;
                xpublish_raw(pa.red + 2 * (size_t)b, v, ckey);
            }
            unsigned long long *const result = pa.red + 2 * (size_t)NS3D_PERSIST_MAXWG;
            if (b == 0) {
                unsigned long long v = 0ull;
                for (int j = tid; j < (int)gridDim.x; j += BX * BY * BZ) {
                    const unsigned long long u = xfetch_raw(pa.red + 2 * (size_t)j, ckey, pa);
                    v = u > v ? u : v;
                }
                v = wave_max_u64(v);
                __syncthreads();                 // red_w has been read by thread 0
                if ((tid & 63) == 0) red_w[tid >> 6] = v;
                __syncthreads();
                if (tid == 0) {
                    v = 0ull;
#pragma unroll
                    for (int q = 0; q < BY * BZ; ++q) v = red_w[q] > v ? red_w[q] : v;
                    xpublish_raw(result, v, ckey);
                }
            }
            if (tid == 0) {
                unsigned long long kall = xfetch_raw(result, ckey, pa);
                // a wait that expired anywhere (the launch is void, the host redoes it): leave now instead of waiting out every
                // remaining hand-over
                const bool expired = __hip_atomic_load(pa.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pa.ticket;
                if (expired) kall = 0x7FF8000000000000ull;
                const volatile PersistCheck *pc_ = (const volatile PersistCheck *)(pa.red + 2 * NS3D_PERSIST_MAXCHK);
                const double eps = pc_->eps, err = __longlong_as_double((long long)kall) * pc_->err_mul / pc_->err_div;     // multi.jl:466
                if (b == 0 && nchecks < NS3D_PERSIST_MAXCHK)
                    __hip_atomic_store(pc_->res + 2 + nchecks, kall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                red_flag = expired ? 2 : (eps >= 0.0 && (err < eps || !__builtin_isfinite(err))) ? 1 : 0;                  // :467-469
            }
            __syncthreads();
            const int flag = red_flag;
            if (flag == 2) break;
            ++nchecks;
            if (flag == 1 || it == pa.n_iters) break;
        }
        d = d * a.one_m_damp + a.dtau * res;
        c = c + a.dtau * d;
        T *__restrict__ ln = Lp[cur ^ 1];
        ln[ctr] = c;
        // the faces are handed over unless nothing follows: neither an iteration nor a check
        const bool last = it + 1 == pa.n_iters && it + 1 != next_check;
        if (!last) {
            const unsigned long long key = xkey(pa.epoch + (unsigned long long)it);
            unsigned long long *__restrict__ hp = Hme + 2 * (size_t)((it & (R - 1)) * FACE);
            if (pa.fault && b == 0) { /* test hook: a workgroup that never hands its faces over */ } else {
            if (fyl) xpublish<T>(hp + 2 * oyl, c, key);
            if (fyh) xpublish<T>(hp + 2 * oyh, c, key);
            if (fzl) xpublish<T>(hp + 2 * ozl, c, key);
            if (fzh) xpublish<T>(hp + 2 * ozh, c, key);
            if (fxl) xpublish<T>(hp + 2 * oxl, c, key);
            if (fxh) xpublish<T>(hp + 2 * oxh, c, key);
            }
            const size_t so = 2 * (size_t)((it & (R - 1)) * FACE);
            // my low-side halo ← the neighbour's high face and vice versa
            if (fyl) ln[ctr - PX] = xfetch<T>(pa.H + 2 * (size_t)(b - pa.nbx) * R * FACE + so + 2 * oyh, key, pa);
            if (fyh) ln[ctr + PX] = xfetch<T>(pa.H + 2 * (size_t)(b + pa.nbx) * R * FACE + so + 2 * oyl, key, pa);
            if (fzl) ln[ctr - PX * PY] = xfetch<T>(pa.H + 2 * (size_t)(b - nbxy) * R * FACE + so + 2 * ozh, key, pa);
            if (fzh) ln[ctr + PX * PY] = xfetch<T>(pa.H + 2 * (size_t)(b + nbxy) * R * FACE + so + 2 * ozl, key, pa);
            if (fxl) ln[ctr - 1] = xfetch<T>(pa.H + 2 * (size_t)(b - 1) * R * FACE + so + 2 * oxh, key, pa);
            if (fxh) ln[ctr + 1] = xfetch<T>(pa.H + 2 * (size_t)(b + 1) * R * FACE + so + 2 * oxl, key, pa);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (act) {
        // a wait that expired anywhere in the grid: the result is not the iteration's — poison it, so that the residual norm of
        // the block is NaN and the caller sees a failed solve rather than a plausible field
        // (second line of defence: the host reads the error word at its next synchronisation and redoes the block by launches —
        // ns3d_api.cpp persist_failed — from the inputs, which this launch did not touch)
        if (__hip_atomic_load(pa.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pa.ticket) c = d = (T)__builtin_nan("");
        a.D[dc] = d;
        store_with_bc<T>(a, gi, gj, gk, c);
    }
    if (pa.nchk > 0 && b == 0 && tid == 0) {
        unsigned long long *res_host = ((const volatile PersistCheck *)(pa.red + 2 * NS3D_PERSIST_MAXCHK))->res;
        __hip_atomic_store(res_host, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(res_host + 1, (unsigned long long)nchecks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the exchange area and the error word live with the caller's context (ns3d_persist_state, freed with it); a launch on another
// stream than the previous one waits for that one first — two grids must not meet in the same slots
static hipError_t persist_scratch(ns3d_persist_state *st, hipStream_t s, size_t bytes)
{
    hipError_t e;
    if (!st->ev) {
        if ((e = hipEventCreateWithFlags(&st->ev, hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipHostMalloc((void **)&st->err_host, 64 + (2 + NS3D_PERSIST_MAXCHK + 8) * sizeof(unsigned long long), hipHostMallocMapped)) != hipSuccess) return e;
        *st->err_host = 0u;
        if ((e = hipHostGetDevicePointer((void **)&st->err_host_dev, st->err_host, 0)) != hipSuccess) return e;
        st->res_host = (unsigned long long *)((char *)st->err_host + 64);
        st->res_host_dev = (unsigned long long *)((char *)st->err_host_dev + 64);
    } else if ((e = hipStreamWaitEvent(s, st->ev, 0)) != hipSuccess) return e;      // the previous launch, whatever stream it ran on
    if (st->bytes < bytes) {
        if (st->H) { if ((e = hipEventSynchronize(st->ev)) != hipSuccess) return e; (void)hipFree(st->H); st->H = nullptr; st->bytes = 0; }
        const size_t red_bytes = (2 * (size_t)NS3D_PERSIST_MAXCHK + 8) * sizeof(unsigned long long);
        if ((e = hipMalloc(&st->H, bytes + 64 + red_bytes)) != hipSuccess) return e;
        if ((e = hipMemset(st->H, 0, bytes + 64 + red_bytes)) != hipSuccess) return e;
        st->err = (unsigned *)((char *)st->H + bytes);
        st->red = (unsigned long long *)((char *)st->H + bytes + 64);
        st->bytes = bytes;
    }
    return hipSuccess;
}

// n fused PT iterations (Pin, D) → (Pout, D) in one cooperative launch; hipErrorInvalidValue where the form does not apply
// (z-slab halo planes, more workgroups than the chip holds at once): the caller then takes the launch-per-iteration path
template <class T, int BY, int BZ>
static hipError_t pt_persist_shape(hipStream_t s, const SweepArgs<T> &a, int n_iters, ns3d_persist_state *ps, int nchk = 0,
                                   double eps = -1.0, double err_mul = 1.0, double err_div = 1.0)
{
    constexpr int BX = 64, R = 4, FACE = 2 * BX * BZ + 2 * BX * BY + 2 * BY * BZ;
    const int nbx = (a.nx - 2 + BX - 1) / BX, nby = (a.ny - 2 + BY - 1) / BY, nbz = (a.nz - 2 + BZ - 1) / BZ;
    const long blocks = (long)nbx * nby * nbz;
    // resident together, by the runtime's own count and never more than two to a CU (every workgroup polls its neighbours)
    static const int per_cu = [] {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k_pt_persist<T, BY, BZ>, 64 * BY * BZ, 0) != hipSuccess) { (void)hipGetLastError(); n = 1; }
        return max(1, min(2, min(n, workgroups_per_cu((const void *)k_pt_persist<T, BY, BZ>, 64 * BY * BZ))));
    }();
    if (blocks > (long)device_cus() * per_cu) return hipErrorInvalidValue;
    if (!ps) return hipSuccess;                          // probe: the form applies
    hipError_t e = persist_scratch(ps, s, (size_t)blocks * R * FACE * 2 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    // NS3D_PERSIST_XCDMAP=1: contiguous runs of blocks per XCD instead of round-robin — measured slower (profiles/r3_persist_ab.log)
    static const bool remap = std::getenv("NS3D_PERSIST_XCDMAP") && *std::getenv("NS3D_PERSIST_XCDMAP") == '1';
    PersistArgs<T> pa;
    pa.a = a; pa.H = (unsigned long long *)ps->H; pa.err = ps->err; pa.err_host = ps->err_host_dev;
    pa.epoch = (++ps->launches) << 16;
    pa.ticket = ++ps->ticket;
    if (ps->ticket == 0u) pa.ticket = ps->ticket = 1u;       // 2³² launches later: tickets start over (0 means "never failed")
    pa.nbx = nbx; pa.nby = nby; pa.nbz = nbz; pa.n_iters = n_iters; pa.xcds = remap ? 8 : 1;
    pa.nchk = nchk; pa.red = ps->red;
    if (nchk > 0) {
        const int nblk = n_iters / nchk;
        if (nblk > NS3D_PERSIST_MAXCHK) return hipErrorInvalidValue;
        if (blocks > NS3D_PERSIST_MAXWG) return hipErrorInvalidValue;
        // the checks' parameters: staged in the pinned block (the previous launch that read them has been synchronised with by whoever
        // read its results)
        PersistCheck *stage = (PersistCheck *)(ps->res_host + 2 + NS3D_PERSIST_MAXCHK);
        stage->eps = eps; stage->err_mul = err_mul; stage->err_div = err_div; stage->res = ps->res_host_dev;
        if ((e = hipMemcpyAsync(ps->red + 2 * NS3D_PERSIST_MAXCHK, stage, sizeof(PersistCheck), hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    }
    { const char *fv = std::getenv("NS3D_PERSIST_FAULT"); pa.fault = (fv && *fv == '1') ? 1 : 0; }      // read per launch: a test hook
    // An ORDINARY launch (round 4): the kernel needs its workgroups resident together, not a grid barrier — the grid is sized to fit the
    // chip (above), and a launch that nevertheless finds CUs taken ends in the bounded waits and the caller's redo, not in a wrong field.
    // hipLaunchCooperativeKernel costs ≈10 µs more per block: 63×38×38, blocks of 37: 3.78 → 3.49 µs per iteration.
    // NS3D_PERSIST_PLAIN=0: the cooperative launch (refuses instead of waiting when the grid cannot be resident).
    static const bool plain = !(std::getenv("NS3D_PERSIST_PLAIN") && *std::getenv("NS3D_PERSIST_PLAIN") == '0');
    if (plain) {
        hipLaunchKernelGGL((k_pt_persist<T, BY, BZ>), dim3((unsigned)blocks), dim3(BX, BY, BZ), 0, s, pa);
        e = hipGetLastError();
    } else {
        void *kargs[] = {(void *)&pa};
        e = hipLaunchCooperativeKernel((const void *)k_pt_persist<T, BY, BZ>, dim3((unsigned)blocks), dim3(BX, BY, BZ), kargs, 0, s);
    }
    if (e != hipSuccess) return e;
    if ((e = hipEventRecord(ps->ev, s)) != hipSuccess) return e;
    // NS3D_COOP_CHECK=1: check at once (blocking) and fail the launch instead of leaving the redo to the caller's next synchronisation
    const char *cv = std::getenv("NS3D_COOP_CHECK");
    if (cv && *cv == '1') {
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        const bool failed = *(volatile unsigned *)ps->err_host == ps->ticket;
        ps->checked = ps->ticket;
        if (failed) { ++ps->faults; return hipErrorLaunchTimeOut; }
    }
    return hipSuccess;
}

template <class T>
hipError_t pt_persist(hipStream_t s, const T *Pin, T *Pout, const T *Din, T *Dout, const T *RHS, const ns3d_pt_params &p, int n_iters,
                      ns3d_persist_state *st, int nchk, double eps, double err_mul, double err_div)
{
    if (n_iters < 1 || n_iters > 60000 || p.z_lo_is_halo || p.z_hi_is_halo) return hipErrorInvalidValue;
    SweepArgs<T> a;
    a.Pin = Pin; a.Pout = Pout; a.D = Dout; a.Din = Din; a.RHS = RHS;
    a.g = make_geo<T>(p.dx, p.dy, p.dz);
    a.rho_dt = (T)p.rho / (T)p.dt; a.dtau = (T)p.dtau; a.one_m_damp = (T)1.0 - (T)p.damp;
    a.outlet_val = (T)p.outlet_val; a.rho_g = (T)p.rho * (T)p.g;
    a.nx = p.nx; a.ny = p.ny; a.nz = p.nz;
    a.bc_kind = p.bc_kind; a.owns_outlet = p.owns_outlet; a.zlo_halo = 0; a.zhi_halo = 0;
    a.k0 = 1; a.k1 = p.nz - 1; a.kz = 1; a.no_faces = 0; a.cus_off = 0; a.win = nullptr; a.tx0 = a.ty0 = 0;
    // the iteration is arithmetic on the CUs the grid occupies plus one hand-over: the smallest workgroup the chip still holds
    // all at once spreads the cells over the most CUs.  NS3D_PERSIST_SHAPE=22|42|44 pins a shape (A/B).
    static const int pin = std::getenv("NS3D_PERSIST_SHAPE") ? std::atoi(std::getenv("NS3D_PERSIST_SHAPE")) : 0;
    if (Pin == nullptr) st = nullptr;                    // no arrays: only say whether the form applies to this grid
    if (pin == 22) return pt_persist_shape<T, 2, 2>(s, a, n_iters, st, nchk, eps, err_mul, err_div);
    if (pin == 42) return pt_persist_shape<T, 4, 2>(s, a, n_iters, st, nchk, eps, err_mul, err_div);
    if (pin == 44) return pt_persist_shape<T, 4, 4>(s, a, n_iters, st, nchk, eps, err_mul, err_div);
    if (pt_persist_shape<T, 2, 2>(s, a, n_iters, nullptr) == hipSuccess) return pt_persist_shape<T, 2, 2>(s, a, n_iters, st, nchk, eps, err_mul, err_div);
    if (pt_persist_shape<T, 4, 2>(s, a, n_iters, nullptr) == hipSuccess) return pt_persist_shape<T, 4, 2>(s, a, n_iters, st, nchk, eps, err_mul, err_div);
    return pt_persist_shape<T, 4, 4>(s, a, n_iters, st, nchk, eps, err_mul, err_div);
}

template <class T>
hipError_t pt_sweep(hipStream_t s, int variant, const T *Pin, T *Pout, T *D, const T *RHS, const ns3d_pt_params &p,
                    int k0, int k1)
{
    SweepArgs<T> a;
    a.Pin = Pin; a.Pout = Pout; a.D = D; a.Din = D; a.RHS = RHS;
    a.g = make_geo<T>(p.dx, p.dy, p.dz);
    a.rho_dt = (T)p.rho / (T)p.dt; a.dtau = (T)p.dtau; a.one_m_damp = (T)1.0 - (T)p.damp;
    a.outlet_val = (T)p.outlet_val; a.rho_g = (T)p.rho * (T)p.g;
    a.nx = p.nx; a.ny = p.ny; a.nz = p.nz;
    a.bc_kind = p.bc_kind; a.owns_outlet = p.owns_outlet; a.zlo_halo = p.z_lo_is_halo; a.zhi_halo = p.z_hi_is_halo;
    a.k0 = k0; a.k1 = k1; a.kz = 1; a.no_faces = 0; a.cus_off = 0; a.win = nullptr; a.tx0 = a.ty0 = 0;
    if (k1 <= k0) return hipSuccess;
    // variant = family*100 + kz  (kz = planes marched per block; 0 → default); variant 0 = choose by grid size:
    // grids whose four PT arrays stay resident in L2 / Infinity Cache run best with one thread per cell (neighbours are
    // cache hits); beyond that the z-marching register pipeline wins (tools/sweep_variants.py, profiles/).
    int fam = variant / 100;
    int kz = variant % 100;
    if (kz <= 0) kz = 32;
    if (variant == 0) fam = ((long long)p.nx * p.ny * p.nz <= 64ll * 1000 * 1000) ? 1 : 22;
    switch (fam) {
    case 1: {
        dim3 blk(64, 4, 1);
        hipLaunchKernelGGL(k_pt_sweep_naive<T>, grid3(a.nx - 2, a.ny - 2, k1 - k0, blk), blk, 0, s, a);
        return hipGetLastError();
    }
    case 2: return launch_zmarch<T, 2, 4>(s, a, kz);
    case 7: return launch_zmarch<T, 4, 4>(s, a, kz);
    // P family: <RY rows/thread, WY wave-rows/workgroup, nontemporal streams, unroll, min waves/SIMD>
    case 20: return launch_pipe_auto<T, 2, 1, true, 3, 1>(s, a, kz);
    case 22: return launch_pipe_auto<T, 1, 1, true, 3, 1>(s, a, kz);
    case 26: return launch_pipe_auto<T, 2, 1, false, 3, 1>(s, a, kz);
    case 27: return launch_pipe_auto<T, 1, 2, true, 3, 1>(s, a, kz);
    default: return launch_pipe_auto<T, 1, 1, true, 3, 1>(s, a, kz);
    }
}

// ---------------------------------------------------------------------------------------------------------
// compute_res! fused with maximum(abs.(Rp))  (multi.jl:465-466): no Rp round trip, 16 B/cell.
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_residual_max(const T *__restrict__ P, const T *__restrict__ RHS, T rho_dt,
                                                      Geo<T> g, int nx, int ny, int nz, int kz, unsigned long long *out)
{
    // block = 64×4 columns, marching kz planes: one atomicMax per block, ≤ a few thousand blocks per launch
    // (one atomic per 256-thread block of a one-thread-per-cell grid serialises ≈0.5 M same-address atomics)
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int kb = 1 + blockIdx.z * kz, ke = min(kb + kz, nz - 1);
    unsigned long long key = 0ull;
    if (i <= nx - 2 && j <= ny - 2) {
        const idx_t sy = nx, sz = (idx_t)nx * ny;
        idx_t p = IX3(i, j, kb, nx, ny);
        T b = P[p - sz], c = P[p];
        for (int k = kb; k < ke; ++k, p += sz) {
            const T t = P[p + sz];
            const T r = poisson_rhs<T>(c, P[p - 1], P[p + 1], P[p - sy], P[p + sy], b, t, RHS[p], rho_dt, g);
            const unsigned long long u = abs_key((double)r);
            key = u > key ? u : key;
            b = c; c = t;
        }
    }
    block_max_to_global(key, out);
}
template <class T>
hipError_t residual_max_key(hipStream_t s, const T *Pr, const T *divV, const ns3d_pt_params &p,
                            unsigned long long *key_dev)
{
    hipError_t e = hipMemsetAsync(key_dev, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    const int kz = 16;
    const dim3 blk(64, 4, 1);
    const dim3 grd((unsigned)((p.nx - 2 + 63) / 64), (unsigned)((p.ny - 2 + 3) / 4), (unsigned)((p.nz - 2 + kz - 1) / kz));
    hipLaunchKernelGGL(k_residual_max<T>, grd, blk, 0, s, Pr, divV, (T)p.rho / (T)p.dt, make_geo<T>(p.dx, p.dy, p.dz),
                       p.nx, p.ny, p.nz, kz, key_dev);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Self-test of div_by_known against the hardware's IEEE division: n pseudo-random dividends per launch (random
// significands over 120 binades — one in four over the whole guarded range and beyond it —, plus quotients planted next
// to representable numbers and rounding midpoints).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix(unsigned long long &st)
{
    unsigned long long z = (st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <class T>
__global__ __launch_bounds__(256) void k_divtest(T d, T r, long n, unsigned long long seed, unsigned long long *bad)
{
    unsigned long long st = seed + 0x632BE59BD9B4E019ull * ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long nb = 0;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const unsigned long long u = splitmix(st), v = splitmix(st);
        T x;
        if (sizeof(T) == 8) {
            const unsigned long long mant = u & 0x000FFFFFFFFFFFFFull, sign = u & 0x8000000000000000ull;
            const unsigned long long expo = ((v >> 32) & 3) ? 1023 - 60 + (v % 120) : 1023 - 720 + (v % 1440);  // 1 in 4 over the guard's whole range and beyond
            double xd = __longlong_as_double((long long)(sign | (expo << 52) | mant));
            if ((v >> 60) & 1) { // plant the quotient next to a representable number / midpoint: x ≈ qc·d
                double qc = xd, prod = qc * (double)d;
                long long pb = __double_as_longlong(prod) + (long long)((v >> 56) & 7) - 3;
                xd = __longlong_as_double(pb);
            }
            x = (T)xd;
        } else {
            const unsigned int w = (unsigned int)u;
            const unsigned int mant = w & 0x007FFFFFu, sign = w & 0x80000000u,
                               expo = ((v >> 32) & 3) ? 127 - 20 + (unsigned int)(v % 40) : 127 - 80 + (unsigned int)(v % 160);
            float xf = __uint_as_float(sign | (expo << 23) | mant);
            if ((v >> 60) & 1) xf = __uint_as_float(__float_as_uint(xf * (float)d) + (unsigned int)((v >> 56) & 7) - 3u);
            x = (T)xf;
        }
        const T a = div_by_known(x, d, r), b = x / d;
        bool same;
        if (sizeof(T) == 8) same = __double_as_longlong((double)a) == __double_as_longlong((double)b);
        else same = __float_as_uint((float)a) == __float_as_uint((float)b);
#if defined(NS3D_EXACT_RECIP)
        bool ok = true;                       // the branch-free double division of the hot kernels: x/d/d
        const T a2 = div2_known<T>(x, d, r, ok), b2 = x / d / d;
        if (ok) {
            if (sizeof(T) == 8) same = same && (__double_as_longlong((double)a2) == __double_as_longlong((double)b2));
            else same = same && (__float_as_uint((float)a2) == __float_as_uint((float)b2));
            const T a3 = div2_known_nochk<T>(x, d, r);   // k_pt_sweep2's form (guarded per value, same dividend range)
            if (sizeof(T) == 8) same = same && (__double_as_longlong((double)a3) == __double_as_longlong((double)b2));
            else same = same && (__float_as_uint((float)a3) == __float_as_uint((float)b2));
        }
#endif
        nb += same ? 0 : 1;
    }
    if (nb) atomicAdd(bad, nb);
}
template <class T>
hipError_t divtest(hipStream_t s, double d, long n, unsigned long long seed, unsigned long long *bad_dev)
{
    hipError_t e = hipMemsetAsync(bad_dev, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    const T dd = (T)d, rr = (T)1 / dd;
    hipLaunchKernelGGL(k_divtest<T>, dim3(2048), dim3(256), 0, s, dd, rr, n, seed, bad_dev);
    return hipGetLastError();
}

// ---- explicit instantiations -----------------------------------------------------------------------------
#define INST(T)                                                                                              \
    template hipError_t update_tau<T>(hipStream_t, T *, T *, T *, T *, T *, T *, const T *, const T *,       \
                                      const T *, double, double, double, double, int, int, int);             \
    template hipError_t predict_V<T>(hipStream_t, T *, T *, T *, const T *, const T *, const T *, const T *, \
                                     const T *, const T *, double, double, double, double, double, double,   \
                                     int, int, int);                                                         \
    template hipError_t predict_fused<T>(hipStream_t, T *, T *, T *, const T *, const T *, const T *, double, \
                                         double, double, double, double, double, double, int, int, int);      \
    template hipError_t set_cylinder<T>(hipStream_t, T *, T *, T *, T *, double, double, double, double,     \
                                        double, double, int, double, double, double, double, double, double, \
                                        int, int, int);                                                      \
    template hipError_t update_divV<T>(hipStream_t, T *, const T *, const T *, const T *, double, double,    \
                                       double, int, int, int);                                               \
    template hipError_t update_dPrdtau<T>(hipStream_t, const T *, T *, const T *, double, double, double,    \
                                          double, double, double, double, int, int, int);                    \
    template hipError_t update_Pr<T>(hipStream_t, T *, const T *, double, int, int, int);                    \
    template hipError_t compute_res<T>(hipStream_t, T *, const T *, const T *, double, double, double,       \
                                       double, double, int, int, int);                                       \
    template hipError_t max_abs_key<T>(hipStream_t, const T *, long, unsigned long long *);                  \
    template hipError_t correct_V<T>(hipStream_t, T *, T *, T *, const T *, double, double, double, double,  \
                                     double, int, int, int);                                                 \
    template hipError_t bc_plane<T>(hipStream_t, int, T *, int, int, int, double, double, double, int);      \
    template hipError_t bc_fused<T>(hipStream_t, int, int, T *, T *, T *, int, int, int, int, double, double, double, int); \
    template hipError_t advect<T>(hipStream_t, T *, const T *, T *, const T *, T *, const T *, T *,          \
                                  const T *, double, double, double, double, int, int, int, int, int, int);  \
    template hipError_t pt_sweep<T>(hipStream_t, int, const T *, T *, T *, const T *, const ns3d_pt_params &,\
                                    int, int);                                                               \
    template hipError_t pt_persist<T>(hipStream_t, const T *, T *, const T *, T *, const T *, const ns3d_pt_params &, int, ns3d_persist_state *, \
                                      int, double, double, double);                                             \
    template hipError_t pt_sweep2<T>(hipStream_t, int, const T *, T *, const T *, T *, const T *,            \
                                     const ns3d_pt_params &, int, int, int, const ns3d_tile_window *);       \
    template hipError_t pt_sweepn<T>(hipStream_t, int, int, const T *, T *, const T *, T *, const T *,       \
                                     const ns3d_pt_params &, int, int, int, const ns3d_tile_window *);       \
    template hipError_t pt_faces_region<T>(hipStream_t, T *, const ns3d_pt_params &, const int *, const int *, int); \
    template hipError_t residual_max_key<T>(hipStream_t, const T *, const T *, const ns3d_pt_params &,       \
                                            unsigned long long *);                                           \
    template hipError_t divtest<T>(hipStream_t, double, long, unsigned long long, unsigned long long *);   \
    template hipError_t strip_inner<T>(hipStream_t, const T *, T *, int, int, int);                          \
    template hipError_t face_copy<T>(hipStream_t, T *, T *, int, int, int, int, int, int);                  \
    template hipError_t subbox_copy<T>(hipStream_t, const ns3d_subbox_batch<T> &);
#ifdef NS3D_PROBE   // tools/ab/resources.sh: ONE kernel instance per compilation (seconds instead of minutes)
hipError_t probe_launch(hipStream_t s, SweepArgs<NS3D_PROBE_T> &a) { return NS3D_PROBE(s, a, 0); }
#else
INST(double)
INST(float)
#endif
#undef INST

} // namespace NS3D_NS
