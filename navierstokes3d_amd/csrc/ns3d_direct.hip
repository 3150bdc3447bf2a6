// ns3d_direct.hip — a DIRECT solve of the pressure-Poisson problem the pseudo-transient loop iterates on (SURVEY.md §8 f4:
// "improved solver options outside parity").  Opt-in; the parity path is the PT loop.
//
// What the PT loop (multi.jl:458-471 / gpu.jl:126-137) converges to — up to its tolerance εit = 1e-3 — is the solution of
//     ∇²_h Pr = ρ/dt·∇V   on the interior cells, with the boundary cells set by set_bc_Pr! (multi.jl:175-181 / gpu.jl:281-286):
// a constant-coefficient 7-point Laplacian on a box (the obstacle only enters through ∇V), Neumann copies on the y and z faces,
// and in x either {Neumann, Neumann} (multi.jl on a rank without the outlet), {Neumann, Pr[end,:,:] = val} (multi.jl, outlet
// rank) or {hydrostatic value + 100, hydrostatic value} (gpu.jl).  Finding: the reference's PT parameters are already the
// optimal ones of the accelerated PT / second-order Richardson scheme (damp = 2/nx against the spectral optimum ≈ 1.8/nx;
// measured: no other damp or dτ needs fewer iterations), so what is left to gain is not a parameter but the
// O(n) iteration count itself (2 280 iterations per step at 255×153×153).  The operator is a sum of three 1-D tridiagonal
// operators whose eigenvectors are known in closed form (cosines / shifted cosines / sines), so it diagonalises exactly:
//     Pr = (Vx ⊗ Vy ⊗ Vz) · [ (Vxᵀ ⊗ Vyᵀ ⊗ Vzᵀ) f  ⊘  (λx ⊕ λy ⊕ λz) ]
// — six dense fp64 matrix products and one pointwise division, independent of the iteration count, exact to rounding
// (max|Rp| drops to ~1e-12 of the right-hand side instead of εit).  The products are GEMM-shaped, so they run on the MFMA
// units: k_gemm_f64 below is a hand-written v_mfma_f64_16x16x4_f64 kernel (no library dependency); the sizes are the
// reference's odd ones (253 = 11·23, 151 prime: no FFT-friendly factors), which a dense product does not care about.
// Cost: 4·(mx+my+mz) flops per cell — 10 GFLOP at 255×153×153, 0.8 TFLOP at 512³.
// fp32 fields are solved in fp64 internally (the Laplacian's condition number ~n² eats seven digits at n = 255).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "ns3d_internal.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// C(i,j) = Σ_k A(i,k)·B(k,j) with element strides (any of the six transform steps is one call); i is the dimension that is
// contiguous in C.  A workgroup of four waves owns a 64×64 tile of C, a wave 32×32 = 2×2 MFMA tiles.  v_mfma_f64_16x16x4_f64:
// operand a: lane l holds a[row l&15][k l>>4], operand b: b[k l>>4][col l&15], result register q of lane l is
// d[row (l>>4)+4q][col l&15] (cdna_hip_programming.md, "f64 MFMA does NOT use these maps").  The MFMA's column index runs
// over the lanes, so it is given to i: stores and the loads of A are then contiguous across lanes.
struct GemmArgs {
    const double *A, *B;
    double *C;
    int M, N, K;
    long sai, sak, sbk, sbj, sci, scj;
    long batchA, batchB, batchC;
    // what the LDS-staged kernel does with a finished element instead of C(i,j) = acc (round 4: two pointwise passes folded into the
    // products that feed them):  1: C(i,j) = acc / (λx[i mod mx] + λy[i / mx] + λz[j]), 0 where the sum is 0 (k_direct_scale);
    // 2 / 3: the element is interior cell (i, j mod my, j / my) of the double / float pressure array `out` (k_direct_scatter)
    int epi;
    const double *l0, *l1, *l2;
    void *out;
    int e_mx, e_my, e_nx, e_ny;
};
__global__ __launch_bounds__(256) void k_gemm_f64(GemmArgs g)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ti = ((int)blockIdx.x * 2 + (wave & 1)) * 32, tj = ((int)blockIdx.y * 2 + (wave >> 1)) * 32;
    if (ti >= g.M || tj >= g.N) return;                       // wave-uniform
    const double *__restrict__ A = g.A + (long)blockIdx.z * g.batchA;
    const double *__restrict__ B = g.B + (long)blockIdx.z * g.batchB;
    double *__restrict__ C = g.C + (long)blockIdx.z * g.batchC;
    const int r = lane & 15, kq = lane >> 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = (f64x4){0.0, 0.0, 0.0, 0.0};
    long ai[2], bj[2];
    bool iok[2], jok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = ti + 16 * u + r, j = tj + 16 * u + r;
        iok[u] = i < g.M; jok[u] = j < g.N;
        ai[u] = (long)(iok[u] ? i : 0) * g.sai;
        bj[u] = (long)(jok[u] ? j : 0) * g.sbj;
    }
    for (int k0 = 0; k0 < g.K; k0 += 4) {
        const int k = k0 + kq;
        const bool kok = k < g.K;
        const long kk = kok ? k : 0;
        double a[2], b[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const double av = A[ai[u] + kk * g.sak], bv = B[kk * g.sbk + bj[u]];
            a[u] = (kok && iok[u]) ? av : 0.0;
            b[u] = (kok && jok[u]) ? bv : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)     // MFMA rows ← j (operand a = B's column), MFMA columns ← i (operand b = A's row)
                acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[v], a[u], acc[u][v], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = ti + 16 * u + r, j = tj + 16 * v + kq + 4 * q;
                if (i < g.M && j < g.N) C[(long)i * g.sci + (long)j * g.scj] = acc[u][v][q];
            }
}

// The same product with both operands staged through LDS (round 3, second version): a 256-thread workgroup owns 128 (i) × 64 (j)
// of C, a wave 64×32 = 4×2 MFMA tiles (32 accumulator registers of f64), K in stages of 16: while the MFMAs of a stage run, the
// next stage's 128×16 and 16×64 operand tiles are in flight from global memory into registers, then written to the other LDS
// buffer — one barrier per stage.  Tiles lie in LDS as [k][i] / [k][j] with the pitch padded by two elements: a fragment read
// (16 consecutive i or j for each of 4 consecutive k) and both store patterns (threads running along k where k is the contiguous
// direction in memory, along i / j otherwise) stay within two-way bank conflicts.  Per stage a wave issues 32 MFMAs (2 048 flop
// each) for 24 LDS reads; every element of the large operand is fetched once per workgroup column.
// Round 4: the tile is a template — WI×WJ waves, each UI×VJ MFMA tiles (TI = 16·WI·UI along i, TJ = 16·WJ·VJ along j) — and OCC asks
// the compiler for that many waves per SIMD: the 128×64 form used 118 + 64 registers = two workgroups per CU, so the 714 workgroups
// of the 253-wide transform ran as 512 + 202; at three per CU they are resident together.
template <int WI, int WJ, int UI, int VJ, int OCC>
__global__ __launch_bounds__(256, OCC) void k_gemm_f64_lds(GemmArgs g)
{
    static_assert(WI * WJ == 4, "four waves");
    constexpr int TI = 16 * WI * UI, TJ = 16 * WJ * VJ, KT = 16, PA = TI + 2, PB = TJ + 2, NA = TI * KT / 256, NB = TJ * KT / 256;
    static_assert(TI * KT % 256 == 0 && TJ * KT % 256 == 0, "whole elements per thread");
    __shared__ double As[2][KT * PA];
    __shared__ double Bs[2][KT * PB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wi = wave % WI, wj = wave / WI;
    const int ti = (int)blockIdx.x * TI, tj = (int)blockIdx.y * TJ;
    const double *__restrict__ A = g.A + (long)blockIdx.z * g.batchA;
    const double *__restrict__ B = g.B + (long)blockIdx.z * g.batchB;
    double *__restrict__ C = g.C + (long)blockIdx.z * g.batchC;
    const int r = lane & 15, kq = lane >> 4;
    // which element of a tile this thread moves (NA of A, NB of B per stage): along k where k is contiguous in memory.  Each element
    // has its pointer (advanced by a stage per fetch: two additions instead of two 64-bit multiplications — the SQ counters of round 3's
    // form showed 8.5 VALU instructions per MFMA, most of them this index arithmetic), its LDS position and a validity bit, all
    // computed once; only the last, partial stage checks k (wave-uniform branch).
    const bool a_kfast = g.sak == 1 && g.sai != 1, b_kfast = g.sbk == 1 && g.sbj != 1;
    const double *pa[NA], *pb[NB];
    int sa[NA], sb[NB];
    unsigned okm = 0u;                                        // bit q: A element q inside M; bit NA+q: B element q inside N
    auto a_k = [&](int q) { return a_kfast ? (tid & 15) : (tid + 256 * q) / TI; };
    auto b_k = [&](int q) { return b_kfast ? (tid & 15) : (tid + 256 * q) / TJ; };
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int il = a_kfast ? (tid >> 4) + 16 * q : (tid + 256 * q) % TI, kl = a_k(q);
        const bool ok = ti + il < g.M;
        okm |= (ok ? 1u : 0u) << q;
        pa[q] = A + (long)(ok ? ti + il : 0) * g.sai + (long)kl * g.sak;
        sa[q] = kl * PA + il;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int jl = b_kfast ? (tid >> 4) + 16 * q : (tid + 256 * q) % TJ, kl = b_k(q);
        const bool ok = tj + jl < g.N;
        okm |= (ok ? 1u : 0u) << (NA + q);
        pb[q] = B + (long)kl * g.sbk + (long)(ok ? tj + jl : 0) * g.sbj;
        sb[q] = kl * PB + jl;
    }
    const long stepa = (long)KT * g.sak, stepb = (long)KT * g.sbk;
    f64x4 acc[UI][VJ];
#pragma unroll
    for (int u = 0; u < UI; ++u)
#pragma unroll
        for (int v = 0; v < VJ; ++v) acc[u][v] = (f64x4){0.0, 0.0, 0.0, 0.0};
    double ra[NA], rb[NB];
    auto fetch = [&](int k0) {
        if (k0 + KT <= g.K) {                                 // a whole stage: every k is inside
#pragma unroll
            for (int q = 0; q < NA; ++q) { const double v = *pa[q]; ra[q] = (okm >> q) & 1u ? v : 0.0; pa[q] += stepa; }
#pragma unroll
            for (int q = 0; q < NB; ++q) { const double v = *pb[q]; rb[q] = (okm >> (NA + q)) & 1u ? v : 0.0; pb[q] += stepb; }
        } else {                                              // the last, partial stage: rows beyond K are zeros, never loaded
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const bool ok = ((okm >> q) & 1u) && k0 + a_k(q) < g.K;
                const double v = *(ok ? pa[q] : A);
                ra[q] = ok ? v : 0.0;
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const bool ok = ((okm >> (NA + q)) & 1u) && k0 + b_k(q) < g.K;
                const double v = *(ok ? pb[q] : B);
                rb[q] = ok ? v : 0.0;
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NA; ++q) As[buf][sa[q]] = ra[q];
#pragma unroll
        for (int q = 0; q < NB; ++q) Bs[buf][sb[q]] = rb[q];
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < g.K; k0 += KT) {
        const bool more = k0 + KT < g.K;
        if (more) fetch(k0 + KT);
        const double *__restrict__ as = As[cur], *__restrict__ bs = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < KT / 4; ++kk) {
            double av[UI], bv[VJ];
#pragma unroll
            for (int u = 0; u < UI; ++u) av[u] = as[(4 * kk + kq) * PA + wi * (16 * UI) + 16 * u + r];
#pragma unroll
            for (int v = 0; v < VJ; ++v) bv[v] = bs[(4 * kk + kq) * PB + wj * (16 * VJ) + 16 * v + r];
#pragma unroll
            for (int u = 0; u < UI; ++u)
#pragma unroll
                for (int v = 0; v < VJ; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[v], av[u], acc[u][v], 0, 0, 0);
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int u = 0; u < UI; ++u)
#pragma unroll
        for (int v = 0; v < VJ; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = ti + wi * (16 * UI) + 16 * u + r, j = tj + wj * (16 * VJ) + 16 * v + kq + 4 * q;
                if (i < g.M && j < g.N) {
                    const double a = acc[u][v][q];
                    if (g.epi == 0) C[(long)i * g.sci + (long)j * g.scj] = a;
                    else if (g.epi == 1) {
                        const double lam = g.l0[i % g.e_mx] + g.l1[i / g.e_mx] + g.l2[j];
                        C[(long)i * g.sci + (long)j * g.scj] = lam != 0.0 ? a / lam : 0.0;
                    } else {
                        const long at = (long)(i + 1) + (long)g.e_nx * ((long)(j % g.e_my + 1) + (long)g.e_ny * (j / g.e_my + 1));
                        if (g.epi == 2) ((double *)g.out)[at] = a;
                        else ((float *)g.out)[at] = (float)a;
                    }
                }
            }
}

template <int WI, int WJ, int UI, int VJ, int OCC>
static void launch_gemm_lds(hipStream_t s, const GemmArgs &g, int batch)
{
    constexpr int TI = 16 * WI * UI, TJ = 16 * WJ * VJ;
    hipLaunchKernelGGL((k_gemm_f64_lds<WI, WJ, UI, VJ, OCC>), dim3((unsigned)((g.M + TI - 1) / TI), (unsigned)((g.N + TJ - 1) / TJ), (unsigned)batch),
                       dim3(256), 0, s, g);
}

hipError_t gemm(hipStream_t s, const double *A, const double *B, double *C, int M, int N, int K, long sai, long sak, long sbk,
                long sbj, long sci, long scj, int batch = 1, long bA = 0, long bB = 0, long bC = 0, const GemmArgs *epi = nullptr)
{
    GemmArgs g{A, B, C, M, N, K, sai, sak, sbk, sbj, sci, scj, bA, bB, bC, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    if (epi) { g.epi = epi->epi; g.l0 = epi->l0; g.l1 = epi->l1; g.l2 = epi->l2; g.out = epi->out; g.e_mx = epi->e_mx; g.e_my = epi->e_my; g.e_nx = epi->e_nx; g.e_ny = epi->e_ny; }
    static const bool simple_only = std::getenv("NS3D_GEMM_SIMPLE") && *std::getenv("NS3D_GEMM_SIMPLE") == '1';     // A/B
    // NS3D_GEMM_SHAPE=0 (A/B): 128×64 as round 3 built it (118 + 64 registers: two workgroups per CU); default: the same tile at three
    // per CU (168 registers, 10 spilled outside the MFMA loop).  255×153×153 step with the direct solve 1.29 → 1.19 ms; a 128×80 tile
    // (<4,1,2,5>: 151 columns in two tiles instead of three) spilled 33 registers and lost: 1.80 ms (profiles/r4_step_cost.log).
    static const int shape_env = std::getenv("NS3D_GEMM_SHAPE") ? std::atoi(std::getenv("NS3D_GEMM_SHAPE")) : -1;
    if (M >= 32 && N >= 16 && !simple_only) {
        if (shape_env == 0) launch_gemm_lds<2, 2, 4, 2, 1>(s, g, batch);
        else launch_gemm_lds<2, 2, 4, 2, 3>(s, g, batch);
    } else {
        if (g.epi) return hipErrorInvalidValue;              // the register-only kernel has no epilogue: the caller runs the pointwise pass
        hipLaunchKernelGGL(k_gemm_f64, dim3((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64), (unsigned)batch), dim3(256), 0, s, g);
    }
    return hipGetLastError();
}

// right-hand side on the interior cells: ρ/dt·∇V minus what the KNOWN boundary cells contribute to the stencil
template <class T>
__global__ __launch_bounds__(256) void k_direct_rhs(double *__restrict__ F, const T *__restrict__ divV, double rho_dt, int nx, int ny,
                                                    int nz, int bc_kind, int owns_outlet, double outlet_val, double rho_g, double dz,
                                                    double rdx2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
    const int mx = nx - 2, my = ny - 2;
    if (i >= mx || j >= my) return;
    double f = rho_dt * (double)divV[(long)(i + 1) + (long)nx * ((long)(j + 1) + (long)ny * (k + 1))];
    if (bc_kind == NS3D_BC_GPU) {       // gpu.jl:258-259 (bc_xhydstatic!): both x planes hold given values, plane index k+1 (0-based)
        const double h = (rho_g * ((double)(nz - (k + 2)) + 0.5)) * dz;
        if (i == 0) f -= (h + 100.0) * rdx2;
        if (i == mx - 1) f -= h * rdx2;
    } else if (owns_outlet && i == mx - 1)
        f -= outlet_val * rdx2;         // multi.jl:179-180: Pr[end,:,:] = val
    F[(long)i + (long)mx * ((long)j + (long)my * k)] = f;
}
// û ← û / (λx + λy + λz); the null mode of the all-Neumann problem (λ = 0) is set to zero: the zero-mean solution
__global__ __launch_bounds__(256) void k_direct_scale(double *__restrict__ U, const double *__restrict__ lx, const double *__restrict__ ly,
                                                      const double *__restrict__ lz, int mx, int my, int mz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
    if (i >= mx || j >= my) return;
    const double lam = lx[i] + ly[j] + lz[k];
    const long q = (long)i + (long)mx * ((long)j + (long)my * k);
    U[q] = lam != 0.0 ? U[q] / lam : 0.0;
}
template <class T>
__global__ __launch_bounds__(256) void k_direct_scatter(T *__restrict__ Pr, T *__restrict__ D, const double *__restrict__ U, int nx, int ny,
                                                        int nz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
    const int mx = nx - 2, my = ny - 2;
    if (i >= mx || j >= my) return;
    const long q = (long)i + (long)mx * ((long)j + (long)my * k);
    Pr[(long)(i + 1) + (long)nx * ((long)(j + 1) + (long)ny * (k + 1))] = (T)U[q];
    D[q] = (T)0;                        // the pseudo-velocity of a converged state
}

// eigenpairs of the 1-D second-difference operator on m interior cells (columns of V, orthonormal):
//   kind 0: copies on both ends (Neumann)          v_q(i) = cos(θ(i+½)),  θ = πq/m
//   kind 1: copy below, ZERO cell above            v_q(i) = cos(θ(i+½)),  θ = π(q+½)/(m+½)
//   kind 2: zero cells on both ends                v_q(i) = sin(θ(i+1)),  θ = π(q+1)/(m+1)
// eigenvalue −(2−2cosθ)/d² = −4 sin²(θ/2)/d² (inhomogeneous boundary values go to the right-hand side)
void eig1d(int m, double d, int kind, std::vector<double> &V, std::vector<double> &lam)
{
    V.assign((size_t)m * m, 0.0);
    lam.assign(m, 0.0);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int q = 0; q < m; ++q) {
        const long double th = kind == 0 ? pi * q / m : kind == 1 ? pi * (q + 0.5L) / (m + 0.5L) : pi * (q + 1) / (m + 1);
        long double nrm = 0.0L;
        for (int i = 0; i < m; ++i) {
            const long double v = kind == 2 ? sinl(th * (i + 1)) : cosl(th * (i + 0.5L));
            V[(size_t)i + (size_t)q * m] = (double)v;
            nrm += v * v;
        }
        const long double inv = 1.0L / sqrtl(nrm);
        for (int i = 0; i < m; ++i) V[(size_t)i + (size_t)q * m] = (double)(V[(size_t)i + (size_t)q * m] * inv);
        const long double sh = sinl(th / 2);
        lam[q] = (double)(-4.0L * sh * sh / ((long double)d * d));
    }
    if (kind == 0) lam[0] = 0.0;
}

struct Eig1d { int m, kind; double d; std::vector<double> V, lam; };

struct DirectPlan {
    int nx = 0, ny = 0, nz = 0, xkind = -1;
    double dx = 0, dy = 0, dz = 0;
    double *V[3] = {nullptr, nullptr, nullptr}, *lam[3] = {nullptr, nullptr, nullptr};
    double *W[2] = {nullptr, nullptr};
    void release()
    {
        for (int d = 0; d < 3; ++d) { if (V[d]) (void)hipFree(V[d]); if (lam[d]) (void)hipFree(lam[d]); V[d] = lam[d] = nullptr; }
        for (int q = 0; q < 2; ++q) { if (W[q]) (void)hipFree(W[q]); W[q] = nullptr; }
        nx = ny = nz = 0; xkind = -1;
    }
};

void free_plan(void *p)
{
    DirectPlan *pl = (DirectPlan *)p;
    if (!pl) return;
    pl->release();
    delete pl;
}

int ensure_plan(ns3d_ctx *c, const ns3d_pt_params *p, int xkind, DirectPlan **out)
{
    DirectPlan *pl = (DirectPlan *)c->direct_plan;
    if (!pl) { pl = new DirectPlan(); c->direct_plan = pl; c->direct_free = free_plan; }
    *out = pl;
    if (pl->nx == p->nx && pl->ny == p->ny && pl->nz == p->nz && pl->xkind == xkind && pl->dx == p->dx && pl->dy == p->dy && pl->dz == p->dz)
        return NS3D_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    pl->release();
    const int m[3] = {p->nx - 2, p->ny - 2, p->nz - 2};
    const double d[3] = {p->dx, p->dy, p->dz};
    const int kind[3] = {xkind, 0, 0};
    for (int q = 0; q < 3; ++q) {
        // the eigenbases depend on (extent, spacing, boundary kind) only: computed once per process (contexts come and go, one
        // per driver call), uploaded per context
        static std::vector<Eig1d> cache;
        static std::mutex mtx;
        std::vector<double> V, lam;
        {
            std::lock_guard<std::mutex> lock(mtx);
            const Eig1d *hit = nullptr;
            for (const Eig1d &e : cache)
                if (e.m == m[q] && e.d == d[q] && e.kind == kind[q]) hit = &e;
            if (!hit) {
                if (cache.size() >= 32) cache.clear();
                cache.push_back({m[q], kind[q], d[q], {}, {}});
                eig1d(m[q], d[q], kind[q], cache.back().V, cache.back().lam);
                hit = &cache.back();
            }
            V = hit->V; lam = hit->lam;
        }
        HIPCHK(c, hipMalloc((void **)&pl->V[q], V.size() * sizeof(double)));
        HIPCHK(c, hipMalloc((void **)&pl->lam[q], lam.size() * sizeof(double)));
        HIPCHK(c, hipMemcpy(pl->V[q], V.data(), V.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(pl->lam[q], lam.data(), lam.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const size_t cells = (size_t)m[0] * m[1] * m[2];
    for (int q = 0; q < 2; ++q) HIPCHK(c, hipMalloc((void **)&pl->W[q], cells * sizeof(double)));
    pl->nx = p->nx; pl->ny = p->ny; pl->nz = p->nz; pl->xkind = xkind; pl->dx = p->dx; pl->dy = p->dy; pl->dz = p->dz;
    return NS3D_OK;
}

template <class T>
int poisson_direct(ns3d_ctx *c, T *Pr, T *D, const T *divV, const ns3d_pt_params *p)
{
    int rc = ns3d_check_pt_params(p, "ns3d_poisson_direct");
    if (rc) return rc;
    if (p->z_lo_is_halo || p->z_hi_is_halo)
        return fail(NS3D_ERR_ARG, "ns3d_poisson_direct: single-rank grids only (a z-slab rank's planes are not a closed problem)");
    if (p->nx < 4 || p->ny < 4 || p->nz < 4) return fail(NS3D_ERR_ARG, "ns3d_poisson_direct: grid %dx%dx%d too small", p->nx, p->ny, p->nz);
    const int xkind = p->bc_kind == NS3D_BC_GPU ? 2 : (p->owns_outlet ? 1 : 0);
    DirectPlan *pl = nullptr;
    if ((rc = ensure_plan(c, p, xkind, &pl))) return rc;
    hipStream_t s = c->stream;
    const int nx = p->nx, ny = p->ny, nz = p->nz, mx = nx - 2, my = ny - 2, mz = nz - 2;
    const long mxy = (long)mx * my;
    const dim3 blk(64, 4, 1), grd((unsigned)((mx + 63) / 64), (unsigned)((my + 3) / 4), (unsigned)mz);
    double *W0 = pl->W[0], *W1 = pl->W[1];
    hipLaunchKernelGGL(k_direct_rhs<T>, grd, blk, 0, s, W0, divV, p->rho / p->dt, nx, ny, nz, p->bc_kind, p->owns_outlet, p->outlet_val,
                       p->rho * p->g, p->dz, 1.0 / (p->dx * p->dx));
    hipError_t e = hipGetLastError();
    // forward: x (Vxᵀ·U), y (U_k·Vy per plane), z (U·Vz)
    if (e == hipSuccess) e = gemm(s, pl->V[0], W0, W1, mx, my * mz, mx, /*A(a,i)=Vx[i+a·mx]*/ mx, 1, /*B*/ 1, mx, /*C*/ 1, mx);
    if (e == hipSuccess) e = gemm(s, W1, pl->V[1], W0, mx, my, my, 1, mx, 1, my, 1, mx, mz, mxy, 0, mxy);
    // … the division by the eigenvalue sums rides on the store of the z product, the scatter into Pr on the store of the last one
    // (NS3D_DIRECT_FUSED=0, or extents the LDS-staged kernel does not take: the pointwise kernels as before — same values)
    static const bool fused = !(std::getenv("NS3D_DIRECT_FUSED") && *std::getenv("NS3D_DIRECT_FUSED") == '0');
    GemmArgs ep{};
    ep.epi = 1; ep.l0 = pl->lam[0]; ep.l1 = pl->lam[1]; ep.l2 = pl->lam[2]; ep.e_mx = mx; ep.e_my = my;
    bool scaled = false;
    if (e == hipSuccess && fused) {
        e = gemm(s, W0, pl->V[2], W1, (int)mxy, mz, mz, 1, mxy, 1, mz, 1, mxy, 1, 0, 0, 0, &ep);
        if (e == hipErrorInvalidValue) { (void)hipGetLastError(); e = hipSuccess; } else scaled = true;
    }
    if (e == hipSuccess && !scaled) {
        e = gemm(s, W0, pl->V[2], W1, (int)mxy, mz, mz, 1, mxy, 1, mz, 1, mxy);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_direct_scale, grd, blk, 0, s, W1, pl->lam[0], pl->lam[1], pl->lam[2], mx, my, mz);
            e = hipGetLastError();
        }
    }
    // backward: z (Û·Vzᵀ), y (Û_k·Vyᵀ), x (Vx·Û)
    if (e == hipSuccess) e = gemm(s, W1, pl->V[2], W0, (int)mxy, mz, mz, 1, mxy, /*B(c,k)=Vz[k+c·mz]*/ mz, 1, 1, mxy);
    if (e == hipSuccess) e = gemm(s, W0, pl->V[1], W1, mx, my, my, 1, mx, my, 1, 1, mx, mz, mxy, 0, mxy);
    bool scattered = false;
    if (e == hipSuccess && fused) {
        GemmArgs es{};
        es.epi = sizeof(T) == 8 ? 2 : 3; es.out = (void *)Pr; es.e_mx = mx; es.e_my = my; es.e_nx = nx; es.e_ny = ny;
        e = gemm(s, pl->V[0], W1, W0, mx, my * mz, mx, 1, mx, 1, mx, 1, mx, 1, 0, 0, 0, &es);
        if (e == hipErrorInvalidValue) { (void)hipGetLastError(); e = hipSuccess; }
        else {
            scattered = true;
            if (e == hipSuccess) e = hipMemsetAsync(D, 0, (size_t)mxy * mz * sizeof(T), s);       // the pseudo-velocity of a converged state
        }
    }
    if (e == hipSuccess && !scattered) {
        e = gemm(s, pl->V[0], W1, W0, mx, my * mz, mx, 1, mx, 1, mx, 1, mx);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_direct_scatter<T>, grd, blk, 0, s, Pr, D, W0, nx, ny, nz);
            e = hipGetLastError();
        }
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(NS3D_ERR_HIP, "ns3d_poisson_direct launch: %s", hipGetErrorString(e)); }
    return NS3D_OK;
}

} // namespace

// set_bc_Pr! (host sequence of ns3d_api.cpp) on the context's stream
extern "C" int ns3d_set_bc_Pr_f64(ns3d_ctx *, double *, int, int, double, double, int, double, double, int, int, int);
extern "C" int ns3d_set_bc_Pr_f32(ns3d_ctx *, float *, int, int, double, double, int, double, double, int, int, int);

#define NS3D_DIRECT_DEFINE(T, S)                                                                             \
    extern "C" int ns3d_poisson_direct_##S(ns3d_ctx *c, T *Pr, T *dPrdtau, const T *divV, const ns3d_pt_params *p) \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Pr, dPrdtau, divV, p);                                                      \
        int rc = poisson_direct<T>(c, Pr, dPrdtau, divV, p);                                                 \
        if (rc) return rc;                                                                                   \
        /* boundary cells as set_bc_Pr! leaves them (blocks like @parallel unless the context is asynchronous) */ \
        return ns3d_set_bc_Pr_##S(c, Pr, p->bc_kind, p->owns_outlet, p->outlet_val, p->dz, p->nz, p->g, p->rho, p->nx, p->ny, p->nz); \
    }
NS3D_DIRECT_DEFINE(double, f64)
NS3D_DIRECT_DEFINE(float, f32)
