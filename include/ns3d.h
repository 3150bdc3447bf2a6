/*
 * ns3d.h — C ABI of libns3d.so: hand-written HIP (gfx950 / MI355X) kernels for the hot path of
 * mattbuergler/NavierStokes3D, behind the reference's own kernel call signatures.
 *
 * The reference has no FFI: its "operator API" is ParallelStencil's calling convention
 *     @parallel kernel!(arrays…, scalars…)
 * used by the time loops scripts/NavierStokes3D_gpu.jl:119-142 ("gpu.jl") and
 * scripts/NavierStokes3D_multi_gpu.jl:446-477 ("multi.jl").  Every entry point below replaces one of
 * those kernels (cited per function) with the SAME positional argument order, followed by the cell grid
 * (nx,ny,nz) that ParallelStencil derives from the array sizes.  INTEGRATION.md shows the Julia `ccall`
 * stubs that bind them so the reference time loops run unchanged.
 *
 * Conventions
 *  - All field pointers are DEVICE pointers to packed column-major arrays (x fastest) with exactly the
 *    reference shapes (multi.jl:343-360):
 *        Pr,C,C_o,τxx,τyy,τzz,∇V : (nx,ny,nz)      Vx,Vx_o : (nx+1,ny,nz)   Vy,Vy_o : (nx,ny+1,nz)
 *        Vz,Vz_o : (nx,ny,nz+1)    τxy,τxz,τyz : (nx-1,ny-1,nz-1)           dPrdτ,Rp : (nx-2,ny-2,nz-2)
 *    The caller owns every field buffer.  The library owns only its context (stream, reduction scratch,
 *    lazily allocated ping-pong Pr / dPrdτ buffers for the fused PT path).
 *  - Suffix _f64 / _f32 = element type of the arrays; scalar parameters are always C double (Julia
 *    Float64 host values) and are converted to the element type on entry.
 *  - Every function returns 0 (NS3D_OK) or a non-zero status; the message is available from
 *    ns3d_last_error().  Nothing throws or aborts across the boundary.
 *  - Like `@parallel`, every call is synchronous to the caller unless the context was created with
 *    NS3D_ASYNC (then calls only enqueue on the context's stream; use ns3d_sync()).
 *  - One host thread per context (as in the reference: one thread per rank); contexts are not
 *    thread-safe.
 *  - Arithmetic modes: NS3D_STRICT reproduces the reference's operation order with correctly rounded division
 *    and no FMA contraction (bit-identical to the CPU oracle); NS3D_FAST uses reciprocal constants and FMA
 *    (≤1e-6 relative L2 at equal iteration counts).  In STRICT mode x/dx is evaluated as the correctly rounded
 *    quotient by the divisor-known-in-advance sequence q=RN(x·r), e=x−q·dx (FMA), RN(q+e·r) with r=RN(1/dx)
 *    whenever dx,dy,dz are eligible (same bits as the division instruction sequence, ≈⅓ of the instructions;
 *    ns3d_selftest_exact_div compares the two on the device); NS3D_IEEE_DIV forces the plain sequence.
 */
#ifndef NS3D_H
#define NS3D_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ns3d_ctx ns3d_ctx;

enum {
    NS3D_OK = 0,
    NS3D_ERR_ARG = 1,   /* bad argument (null pointer, extent too small, …) */
    NS3D_ERR_HIP = 2,   /* a HIP runtime call or kernel launch failed */
    NS3D_ERR_STATE = 3, /* context misuse */
    NS3D_ERR_RCCL = 4   /* RCCL could not be loaded, or an RCCL call failed */
};

enum {
    NS3D_STRICT = 0x0,
    NS3D_FAST = 0x1,
    NS3D_ASYNC = 0x2,
    NS3D_IEEE_DIV = 0x4 /* STRICT only: always use the plain IEEE division instruction sequence (see below) */
};

/* bc_kind for the pressure / velocity boundary sequences */
enum {
    NS3D_BC_MULTI = 0, /* multi.jl:175-181 / :156-166 */
    NS3D_BC_GPU = 1    /* gpu.jl:281-286 / :264-279 */
};

#define NS3D_VERSION 1
int ns3d_version(void);
const char *ns3d_last_error(void);

/* Context = device + stream + scratch.  Replaces @init_parallel_stencil (gpu.jl:4-8). */
ns3d_ctx *ns3d_create(int device, int flags);
void ns3d_destroy(ns3d_ctx *ctx);
int ns3d_flags(const ns3d_ctx *ctx);
/* Launch on an existing hipStream_t (e.g. PyTorch's current stream; NULL is HIP's null stream).  A new context
 * launches on a private non-blocking stream; ns3d_use_own_stream() returns to it. */
int ns3d_set_stream(ns3d_ctx *ctx, void *hip_stream);
int ns3d_use_own_stream(ns3d_ctx *ctx);
/* Leave `n_cus` compute units of the device OUT of this context's launches: the context gets a stream of its own created with a CU
 * mask (hipExtStreamCreateWithCUMask) and launches there from now on; 0 = back to the unmasked own stream.  Why: the interior sweep
 * of a z-slab rank holds every CU of the chip for ≈150 µs per pass (one workgroup per CU), and RCCL's send/recv kernels on the
 * communication stream need CUs to start — the reference reserves `b_width` cells "for comm / comp overlap" (multi.jl:326) and never
 * uses them; this is the knob that makes room on the device instead.  n_cus is spread over the 8 XCDs (rounded up to a multiple
 * of 8, at most half the device).  NS3D_RESERVE_CUS=n presets it for every context created afterwards.  Results do not depend
 * on it.  Cost measured on one MI355X: profiles/r4_cu_mask_ab.log.  ns3d_get_stream returns the stream (wrap it to order other
 * work with it). */
int ns3d_reserve_cus(ns3d_ctx *ctx, int n_cus);
int ns3d_reserved_cus(const ns3d_ctx *ctx);
void *ns3d_get_stream(ns3d_ctx *ctx);
int ns3d_sync(ns3d_ctx *ctx);
/* Tuning knob for the fused PT sweep (0 = default); see DESIGN.md. */
int ns3d_set_pt_variant(ns3d_ctx *ctx, int variant);
/* Temporal blocking: ns3d_pt_iterate / ns3d_pt_solve advance SEVERAL PT iterations per pass over memory where the
 * schedule allows it (same results): two (this knob), or three / four on large grids where the plan phase measures a gain
 * (ns3d_set_pt_depth, ns3d_set_ptn_variant below).  variant < 0 disables, 0 = automatic; otherwise shape*100 + kz with the tile
 * shapes of DESIGN.md §4.2 and kz = planes per z-chunk (1..89 literal; 0 or 91..99: as many chunks as fill two /
 * kz-90 whole rounds of workgroups on the chip).  The environment variable NS3D_PT2_VARIANT presets it at ns3d_create. */
int ns3d_set_pt2_variant(ns3d_ctx *ctx, int variant);
/* Tile shape of the two-iteration sweep when no explicit variant is set: with autotune on (the default) the first
 * launch on a grid of >= 1.5 M cells times the candidate shapes on the caller's own arguments (the operation is idempotent,
 * every shape gives the same bits) and the process remembers the winner per device and grid; that first call therefore
 * synchronises the stream.  Off: a built-in choice by grid size.  ns3d_last_pt2_variant: the variant of the latest two-iteration launch
 * (0 = built-in choice). */
/* N-iteration sweep (ns3d_pt_sweepn): variant = shape*100 + kz.  Shapes (columns per workgroup): 1: 64×32 (512 threads, four
 * rows per thread), 2: 128×16, 6: 64×16 with 256-thread workgroups; +10 (11, 12, 16): the next step's loads issued before level 1;
 * 22: fp32 only, 64×48 with 768 threads; 23 / 28: 64×24 with 768 threads and two rows per thread (three waves per SIMD; loads
 * before level 1 / between the levels); 24: fp32 only, 64×32 with 1024 threads.  kz as above (0: chosen per launch).  0 = built-in.
 * A shape that cannot hold `nlev` levels (tile too small, LDS) or does not exist for the element type makes the call fail. */
int ns3d_set_ptn_variant(ns3d_ctx *ctx, int variant);
/* PT iterations per pass over memory in ns3d_pt_iterate / ns3d_pt_solve: 0 = automatic, 1…4 forced (same results); 5 is
 * available with float32 fields only (k_pt_sweepN has registers for a fifth level there and nowhere else). */
int ns3d_set_pt_depth(ns3d_ctx *ctx, int depth);
int ns3d_set_autotune(ns3d_ctx *ctx, int on);
int ns3d_last_pt2_variant(const ns3d_ctx *ctx);
int ns3d_last_ptn_variant(const ns3d_ctx *ctx);   /* tile variant of the latest N-iteration launch */
int ns3d_last_pt_depth(const ns3d_ctx *ctx);      /* PT iterations of the latest multi-iteration pass; after ns3d_plan_pt: the planned depth */
/* Which of the four compilations of the kernels a call with these grid spacings runs on this context: 0 = `strict` (plain IEEE
 * divisions), 1 = `strictx` (same bits, divisions by the spacings through the correctly rounded divisor-known-in-advance sequence),
 * 3 = `strictp` (same bits; every spacing a power of two, x/d = x*(1/d) exactly), 2 = `fast` (reciprocals + FMA, NS3D_FAST).
 * The reference has no counterpart (its arithmetic is whatever Julia emits for `x/dx/dx`, multi.jl:71); reported by bench.py. */
int ns3d_arith_build(const ns3d_ctx *ctx, double dx, double dy, double dz);
/* ns3d_pt_solve replays each residual-check block (nchk iterations) as one HIP graph: -1 = automatically on
 * launch-bound grids (< 3 M cells), 0 = never, 1 = always.  Same results either way. */
int ns3d_set_graph_mode(ns3d_ctx *ctx, int mode);
/* ns3d_pt_iterate / ns3d_pt_solve on launch-bound grids: a whole block of iterations (up to the next residual check) in ONE
 * cooperative launch that keeps the grid on the chip (k_pt_persist: a cell per thread, faces handed between workgroups through
 * global memory after every iteration): -1 = automatically where it was measured to win (up to 170 000 cells, nx <= 66, no
 * explicit depth / tile / graph request) and the workgroups fit the chip together, 0 = never, 1 = wherever they fit.  Same
 * results either way.  A hand-over that never arrives (bounded wait: the workgroups were not all resident at once — another context
 * or process holding CUs) voids that launch only: it writes its outputs to buffers of its own, the library reads the launch's error
 * word where it synchronises anyway (the residual read-back of ns3d_pt_solve; ns3d_pt_iterate synchronises once for it), redoes
 * the block by launches from the untouched inputs and leaves the cooperative form off for this context; ns3d_persist_faults counts
 * such launches.  (NS3D_COOP_CHECK=1: the launch itself blocks, checks and fails with NS3D_ERR_HIP instead.)  Env NS3D_PT_PERSIST sets the default. */
int ns3d_set_persist_mode(ns3d_ctx *ctx, int mode);
int ns3d_persist_faults(const ns3d_ctx *ctx);
int ns3d_cached_graphs(const ns3d_ctx *ctx);      /* residual-check blocks this context holds as instantiated HIP graphs */

/* Parameters of the fused pseudo-transient path (ns3d_pt_iterate / ns3d_pt_solve). */
typedef struct ns3d_pt_params {
    double rho, dt, dtau, damp; /* update_dPrdτ!(…,ρ,dt,dτ,damp,…)  multi.jl:70 */
    double dx, dy, dz;
    int nx, ny, nz;             /* local cell grid */
    int bc_kind;                /* NS3D_BC_MULTI | NS3D_BC_GPU */
    int owns_outlet;            /* multi.jl:179  `xve_g == lx/2`  → Pr[end,:,:] = outlet_val */
    double outlet_val;          /* multi.jl:463  0.0 */
    double g;                   /* gpu.jl:284 bc_xhydstatic!(Pr,dz,nz,g,ρ) */
    int z_lo_is_halo;           /* z-slab seams: plane 1 / plane nz belong to the neighbour rank and are */
    int z_hi_is_halo;           /*   filled by the halo exchange instead of bc_z! (multi.jl:182)        */
} ns3d_pt_params;

/* One whole time step in ONE call (round 4): ns3d_time_step enqueues multi.jl:449-477 (one rank) or gpu.jl:121-142 — predictor,
 * set_cylinder!, update_∇V!, the pressure solve, correct_V!, set_cylinder!, set_bc_Vel!, {X_o .= X; advect!} — on the context's
 * stream in its fused form (ns3d_predict_fused, ns3d_pt_solve, ns3d_copy_advect); the residual read-backs of the pressure loop are
 * its only synchronisations.  The fields are device pointers to the reference's arrays (multi.jl:343-360); the struct is IN/OUT:
 * the fused step swaps the roles of X and X_o instead of copying (multi.jl:475 `X_o .= X` becomes a change of names), so after the
 * call `Vx` is the current field and `Vx_o` the previous one, wherever they live.  In `faithful` mode Vz is never advected
 * (App. B1): it keeps its buffer, and Vz_o is what the caller brings up to date once at the end of a run (ns3d_copy).  The stress
 * arrays may be NULL unless write_stress is set (the reference's τ after its last step).  Results: those of the same sequence
 * of single calls, bit for bit (tests/test_gpu_driver.py). */
typedef struct ns3d_step_fields {
    void *Pr, *dPrdtau, *divV;
    void *Vx, *Vy, *Vz, *Vx_o, *Vy_o, *Vz_o, *C, *C_o;
    void *txx, *tyy, *tzz, *txy, *txz, *tyz;
} ns3d_step_fields;
typedef struct ns3d_step_params {
    int script;                 /* NS3D_BC_MULTI: multi.jl:449-477 on one rank; NS3D_BC_GPU: gpu.jl:121-142 */
    int nx, ny, nz;
    double mu, rho, g, dt, dtau, damp, dx, dy, dz;
    double eps; int niter, nchk; double err_mul, err_div;    /* multi.jl:458-471: err = max|Rp|·err_mul/err_div (ly², psc) */
    double a2, b2, ox, oy, sinb, cosb;                       /* set_cylinder! */
    double xco_g, yco_g, zco_g;                              /*   multi.jl form: global coordinates of the rank's first cell centre */
    double lx, ly, lz;
    int owns_inlet, owns_outlet; double vin;                 /* multi.jl:164,179 */
    int faithful;               /* advect!'s third branch as committed (1) or re-writing Vz (0) */
    int pressure;               /* 0: the PT loop (parity); 1: ns3d_poisson_direct (outside parity) */
    int write_stress;           /* also run update_τ! into txx … tyz (the reference's state after its last step) */
} ns3d_step_params;

#define NS3D_DECL(T, S)                                                                                     \
    /* the whole step (above); iters_done / err_hist / n_checks as in ns3d_pt_solve */                          \
    int ns3d_time_step_##S(ns3d_ctx *, ns3d_step_fields *f, const ns3d_step_params *p, int *iters_done,        \
                           double *err_hist, int max_checks, int *n_checks);                                   \
    /* update_τ!(τxx,τyy,τzz,τxy,τxz,τyz,Vx,Vy,Vz,μ,dx,dy,dz)      multi.jl:36-44   gpu.jl:177-185 */         \
    int ns3d_update_tau_##S(ns3d_ctx *, T *txx, T *tyy, T *tzz, T *txy, T *txz, T *tyz, const T *Vx,         \
                            const T *Vy, const T *Vz, double mu, double dx, double dy, double dz, int nx,    \
                            int ny, int nz);                                                                 \
    /* predict_V!(Vx,Vy,Vz,τxx,τyy,τzz,τxy,τxz,τyz,ρ,g,dt,dx,dy,dz) multi.jl:50-55   gpu.jl:187-192 */        \
    int ns3d_predict_V_##S(ns3d_ctx *, T *Vx, T *Vy, T *Vz, const T *txx, const T *tyy, const T *tzz,        \
                           const T *txy, const T *txz, const T *tyz, double rho, double g, double dt,         \
                           double dx, double dy, double dz, int nx, int ny, int nz);                         \
    /* {update_τ!; predict_V!} (multi.jl:449,451 / gpu.jl:121-122) in one pass for a driver that does not look at the stress \
     * arrays: reads Vx, Vy, Vz, writes COMPLETE predicted fields into Vx_new, Vy_new, Vz_new (buffers of their own: the     \
     * entries predict_V! leaves alone are written through) — the caller swaps the names.  Bit for bit what the two calls    \
     * leave in Vx, Vy, Vz.  The stresses are not stored.  Not for ranks that exchange τ halos between the two calls with     \
     * velocities that differ across ranks (multi.jl:450). */                                                \
    int ns3d_predict_fused_##S(ns3d_ctx *, T *Vx_new, T *Vy_new, T *Vz_new, const T *Vx, const T *Vy, const T *Vz,          \
                               double mu, double rho, double g, double dt, double dx, double dy, double dz, int nx, int ny,  \
                               int nz);                                                                      \
    /* set_cylinder!(C,Vx,Vy,Vz,a2,b2,ox,oy,sinβ,cosβ,xco_g,yco_g,zco_g,lx,ly,lz,dx,dy,dz) multi.jl:249-281 */\
    int ns3d_set_cylinder_##S(ns3d_ctx *, T *C, T *Vx, T *Vy, T *Vz, double a2, double b2, double ox,        \
                              double oy, double sinb, double cosb, double xco_g, double yco_g, double zco_g, \
                              double lx, double ly, double lz, double dx, double dy, double dz, int nx,      \
                              int ny, int nz);                                                               \
    /* set_cylinder!(C,Vx,Vy,Vz,a2,b2,ox,oy,sinβ,cosβ,lx,ly,lz,dx,dy,dz)  gpu.jl:336-368 (incl. its dx-for-dy */\
    int ns3d_set_cylinder_local_##S(ns3d_ctx *, T *C, T *Vx, T *Vy, T *Vz, double a2, double b2, double ox,  \
                                    double oy, double sinb, double cosb, double lx, double ly, double lz,    \
                                    double dx, double dy, double dz, int nx, int ny, int nz);                \
    /* update_∇V!(∇V,Vx,Vy,Vz,dx,dy,dz)                             multi.jl:61-64   gpu.jl:194-197 */        \
    int ns3d_update_divV_##S(ns3d_ctx *, T *divV, const T *Vx, const T *Vy, const T *Vz, double dx,          \
                             double dy, double dz, int nx, int ny, int nz);                                  \
    /* update_dPrdτ!(Pr,dPrdτ,∇V,ρ,dt,dτ,damp,dx,dy,dz)            multi.jl:70-73   gpu.jl:199-202 */        \
    int ns3d_update_dPrdtau_##S(ns3d_ctx *, const T *Pr, T *dPrdtau, const T *divV, double rho, double dt,   \
                                double dtau, double damp, double dx, double dy, double dz, int nx, int ny,   \
                                int nz);                                                                     \
    /* update_Pr!(Pr,dPrdτ,dτ)                                     multi.jl:79-82   gpu.jl:204-207 */        \
    int ns3d_update_Pr_##S(ns3d_ctx *, T *Pr, const T *dPrdtau, double dtau, int nx, int ny, int nz);        \
    /* compute_res!(Rp,Pr,∇V,ρ,dt,dx,dy,dz)                        multi.jl:88-91   gpu.jl:209-212 */        \
    int ns3d_compute_res_##S(ns3d_ctx *, T *Rp, const T *Pr, const T *divV, double rho, double dt,           \
                             double dx, double dy, double dz, int nx, int ny, int nz);                       \
    /* maximum(abs.(A)) — NaN-propagating like Julia's maximum     multi.jl:466     gpu.jl:132 */            \
    int ns3d_max_abs_##S(ns3d_ctx *, const T *A, long n_elems, double *out_host);                            \
    /* correct_V!(Vx,Vy,Vz,Pr,dt,ρ,dx,dy,dz)                        multi.jl:97-102  gpu.jl:214-219 */        \
    int ns3d_correct_V_##S(ns3d_ctx *, T *Vx, T *Vy, T *Vz, const T *Pr, double dt, double rho, double dx,   \
                           double dy, double dz, int nx, int ny, int nz);                                    \
    /* bc_x!/bc_y!/bc_z!(A) on an array of extents (sx,sy,sz)       multi.jl:108-132 gpu.jl:221-237 */        \
    int ns3d_bc_x_##S(ns3d_ctx *, T *A, int sx, int sy, int sz);                                             \
    int ns3d_bc_y_##S(ns3d_ctx *, T *A, int sx, int sy, int sz);                                             \
    int ns3d_bc_z_##S(ns3d_ctx *, T *A, int sx, int sy, int sz);                                             \
    /* bc_zV!(A)                                                    gpu.jl:239-243 */                        \
    int ns3d_bc_zV_##S(ns3d_ctx *, T *A, int sx, int sy, int sz);                                            \
    /* bc_xhydstatic!(A,dz,nz,g,ρ)                                  gpu.jl:257-261 */                        \
    int ns3d_bc_xhydstatic_##S(ns3d_ctx *, T *A, double dz, int nz, double g, double rho, int sx, int sy,    \
                               int sz);                                                                      \
    /* bc_x_Vx!(A,V)                                                multi.jl:138-141 */                      \
    int ns3d_bc_x_Vx_##S(ns3d_ctx *, T *A, double V, int sx, int sy, int sz);                                \
    /* bc_x_Pr!(A,val)                                              multi.jl:147-150 */                      \
    int ns3d_bc_x_Pr_##S(ns3d_ctx *, T *A, double val, int sx, int sy, int sz);                              \
    /* X_o .= X                                                     multi.jl:475     gpu.jl:141 */            \
    int ns3d_copy_##S(ns3d_ctx *, T *dst, const T *src, long n_elems);                                       \
    /* advect!(Vx,Vx_o,Vy,Vy_o,Vz,Vz_o,C,C_o,dt,dx,dy,dz)           multi.jl:217-243 gpu.jl:308-334           \
     * faithful!=0 reproduces the reference (third branch back-tracks Vy, Vz never advected).            */  \
    int ns3d_advect_##S(ns3d_ctx *, T *Vx, const T *Vx_o, T *Vy, const T *Vy_o, T *Vz, const T *Vz_o, T *C,  \
                        const T *C_o, double dt, double dx, double dy, double dz, int nx, int ny, int nz,    \
                        int faithful);                                                                       \
    /* {Vx_o .= Vx; Vy_o .= Vy; Vz_o .= Vz; C_o .= C; advect!(…)}  multi.jl:475-476 / gpu.jl:141-142 in ONE pass: the caller    \
     * passes the CURRENT fields (read only) and receives COMPLETE new fields in buffers of its own — every entry is stored, the \
     * ones advect! leaves alone with the current value — and swaps the roles of the buffers afterwards (the four    \
     * copies are "avoidable by pointer swap").  Vz_new may be Vz itself in faithful mode (Vz is never advected there).  Same    \
     * values as ns3d_copy ×4 + ns3d_advect, bit for bit. */                                                 \
    int ns3d_copy_advect_##S(ns3d_ctx *, T *Vx_new, const T *Vx, T *Vy_new, const T *Vy, T *Vz_new, const T *Vz, T *C_new, \
                             const T *C, double dt, double dx, double dy, double dz, int nx, int ny, int nz, \
                             int faithful);                                                                  \
    /* DIRECT solve of what the pseudo-transient loop multi.jl:458-471 / gpu.jl:126-137 iterates towards (an   \
     * option OUTSIDE parity: the reference stops at err < 1e-3, this solves the same discrete system to rounding): Pr's      \
     * interior becomes the solution of ∇²_h Pr = ρ/dt·∇V with the boundary cells of set_bc_Pr! (p->bc_kind, p->owns_outlet,  \
     * p->outlet_val, p->g), the boundary cells are set accordingly, dPrdτ = 0.  Exact diagonalisation of the box Laplacian   \
     * (closed-form eigenvectors per direction) by six fp64 MFMA matrix products; fp32 fields are solved in fp64.  Single-rank \
     * grids (no z halo flags).  All-Neumann problems (no outlet): the zero-mean solution. */                \
    int ns3d_poisson_direct_##S(ns3d_ctx *, T *Pr, T *dPrdtau, const T *divV, const ns3d_pt_params *p);      \
    /* ---- host sequences of the reference, one call each ---- */                                           \
    /* set_bc_Pr!  multi.jl:175-181 (kind 0, without the halo update) / gpu.jl:281-286 (kind 1).  Both set_bc_* run the      \
     * reference's rule sequence as ONE gather launch (same values on every cell, edges and corners included;                \
     * NS3D_BC_FUSED=0 or an extent below 3: a launch per rule) */                                            \
    int ns3d_set_bc_Pr_##S(ns3d_ctx *, T *Pr, int bc_kind, int owns_outlet, double outlet_val, double dz,    \
                           int nz_arg, double g, double rho, int nx, int ny, int nz);                        \
    /* set_bc_Vel! multi.jl:156-166 (kind 0, without the halo update) / gpu.jl:264-279 (kind 1) */           \
    int ns3d_set_bc_Vel_##S(ns3d_ctx *, T *Vx, T *Vy, T *Vz, int bc_kind, int owns_inlet, double vin,        \
                            int nx, int ny, int nz);                                                         \
    /* ---- fused fast path (not in the reference; same results) ----                                        \
     * n_iters × { update_dPrdτ! ; update_Pr! ; set_bc_Pr! }  (multi.jl:459-463 / gpu.jl:127-129) as ONE     \
     * ping-pong sweep per iteration with the boundary planes folded in.  Pr holds the result on return. */  \
    int ns3d_pt_iterate_##S(ns3d_ctx *, T *Pr, T *dPrdtau, const T *divV, const ns3d_pt_params *p,           \
                            int n_iters);                                                                    \
    /* One sweep Pr_in → Pr_out (distinct buffers; halo planes of Pr_out are NOT written when               \
     * z_*_is_halo) restricted to interior planes k0 ≤ k < k1 (0-based Pr plane index, 1 ≤ k0, k1 ≤ nz-1).  \
     * Building block for the z-slab overlap schedule (boundary planes first, interior behind the halo     \
     * exchange). */                                                                                         \
    int ns3d_pt_sweep_##S(ns3d_ctx *, const T *Pr_in, T *Pr_out, T *dPrdtau, const T *divV,                  \
                          const ns3d_pt_params *p, int k0, int k1);                                          \
    /* TWO fused PT iterations (Pr_in,dPrdtau_in) → (Pr_out,dPrdtau_out), all four buffers distinct (tiles       \
     * overlap, so nothing is updated in place); results identical to two ns3d_pt_sweep calls with a buffer     \
     * swap, for the output planes k0 ≤ k < k1 (reads planes k0-2 … k1+1 of Pr_in).  z_*_is_halo must be 0:   \
     * z-slab ranks run it on buffers extended by a second ghost plane per seam. */              \
    int ns3d_pt_sweep2_##S(ns3d_ctx *, const T *Pr_in, T *Pr_out, const T *dPrdtau_in, T *dPrdtau_out,       \
                           const T *divV, const ns3d_pt_params *p, int k0, int k1);                          \
    /* nlev (2…4; 5 in the _f32 form) fused PT iterations in one pass over memory, otherwise as ns3d_pt_sweep2: results identical to nlev   \
     * ns3d_pt_sweep calls, output planes k0 ≤ k < k1 (reads planes k0-nlev … k1+nlev-1 of Pr_in, clamped to the grid). */ \
    int ns3d_pt_sweepn_##S(ns3d_ctx *, int nlev, const T *Pr_in, T *Pr_out, const T *dPrdtau_in, T *dPrdtau_out,  \
                           const T *divV, const ns3d_pt_params *p, int k0, int k1);                          \
    /* Plan phase of the two-iteration sweep (ns3d_set_autotune): times the tile shapes NOW on these very arguments  \
     * (idempotent: inputs and outputs are distinct buffers; every shape gives the same bits) and remembers the       \
     * winner per device, grid and plane range for the rest of the process.  Blocks.  ns3d_pt_iterate / ns3d_pt_solve \
     * plan on first use by themselves; ns3d_pt_sweep2 never measures — it looks the choice up (built-in shape when   \
     * nothing was planned).  Pr_out / dPrdtau_out hold the result of one two-iteration pass afterwards. */           \
    int ns3d_plan_pt_##S(ns3d_ctx *, const T *Pr_in, T *Pr_out, const T *dPrdtau_in, T *dPrdtau_out,         \
                         const T *divV, const ns3d_pt_params *p, int k0, int k1);                            \
    /* max|∇²Pr − ρ/dt ∇V| over the interior = maximum(abs.(Rp)) after compute_res!, without writing Rp.   \
     * NaN-propagating.  (multi.jl:465-466) */                                                               \
    int ns3d_residual_max_##S(ns3d_ctx *, const T *Pr, const T *divV, const ns3d_pt_params *p,               \
                              double *out_host);                                                             \
    /* Compares the divisor-known-in-advance division with the plain IEEE division for n pseudo-random dividends  \
     * (bitwise); *mismatches must come back 0. */                                                           \
    int ns3d_selftest_exact_div_##S(ns3d_ctx *, double d, long n, unsigned long long seed, long *mismatches);\
    /* The whole inner loop multi.jl:458-471 / gpu.jl:126-137 on one rank: at most niter iterations, every   \
     * nchk-th computes err = max|Rp|*err_mul/err_div (= maximum(abs.(Rp))*ly^2/psc, multi.jl:466), stops  \
     * on err<eps || !isfinite(err) (eps<0: never stop).                                                      \
     * err_hist (capacity max_checks) may be NULL. */                                                        \
    int ns3d_pt_solve_##S(ns3d_ctx *, T *Pr, T *dPrdtau, const T *divV, const ns3d_pt_params *p, double eps, \
                          int niter, int nchk, double err_mul, double err_div, int *iters_done,              \
                          double *err_hist, int max_checks, int *n_checks);

NS3D_DECL(double, f64)
NS3D_DECL(float, f32)
#undef NS3D_DECL

/* =====================================================================================================================
 * Multi-GPU: the implicit global grid.  Replaces what multi.jl gets from ImplicitGlobalGrid.jl + MPI.jl:
 *     init_global_grid(nx,ny,nz)   multi.jl:325      →  ns3d_mgpu_create / ns3d_mgpu_create_rank   (dims = (1,1,P): z-slabs)
 *                                                       ns3d_mgpu_create_cart / _create_rank_cart  (any dims; ns3d_dims_create =
 *                                                       the MPI_Dims_create default of init_global_grid)
 *     update_halo!(A…)             multi.jl:371,373,450,453,455,460,462,182,167,477  →  ns3d_update_halo
 *     max_g(A)                     multi.jl:21,466   →  ns3d_max_g
 *     gather!(A_inn, A_v)          multi.jl:399-403,528-532  →  ns3d_gather
 *     nz_g()                       multi.jl:328,338  →  ns3d_mgpu_nz_g            finalize_global_grid() :534 → ns3d_mgpu_destroy
 * ImplicitGlobalGrid's indexing is kept: overlap 2, halo width 1, nz_g = P·(nz−2)+2; an array with nz+s planes has overlap
 * 2+s (sends plane 2+s / size−(1+s), receives into 1 / size, 1-based); arrays with overlap < 2 have no halo; physical ends
 * are left untouched; the same rule per dimension for a Cartesian topology, dimensions in the order x, y, z so that edge and
 * corner values arrive in two / three hops.  Column-major xy-planes are contiguous, so a z message is one block as it lies;
 * x and y faces are packed / unpacked by a kernel on both ends.  ns3d_pt_solve_slab runs the loop multi.jl:458-471 on any
 * topology with deep ghosts — several iterations per pass over memory, one round of exchanges per pass: on z-slabs with the state
 * below (seam planes first, their exchange behind the interior sweep), on a grid decomposed in x or y with every rank's state in a
 * box extended by depth−1 ghost cells in each decomposed direction, the ghost layers exchanged dimension by dimension (x / y
 * layers packed by a kernel); with depth 1 (ns3d_mgpu_set_temporal) or NS3D_CART_DEEP=0 such a grid runs one fused sweep and one
 * halo update per iteration instead.  ns3d_slab_load / _iterate / _store are z-slab only.
 *
 * An ns3d_mgpu holds `nlocal` of the P ranks: all P in the one-process form (ns3d_mgpu_create; planes move by
 * hipMemcpyPeerAsync over xGMI; a device may appear several times — virtual ranks), exactly one in the one-process-per-GPU
 * form (ns3d_mgpu_create_rank; planes move by RCCL send/recv on a dedicated stream, the residual by ncclAllReduce).
 * Every per-rank argument is an array of nlocal entries in local-rank order; field lists are field-major:
 * fields[f*nlocal + l].  Kernels for local rank l run on ns3d_mgpu_ctx(m, l).  Calls are ordered with the work already
 * enqueued on those contexts' streams and block unless `flags` had NS3D_ASYNC.
 * ===================================================================================================================== */
typedef struct ns3d_mgpu ns3d_mgpu;
#define NS3D_UNIQUE_ID_BYTES 128

ns3d_mgpu *ns3d_mgpu_create(int P, const int *devices, int nx, int ny, int nz_local, int flags);
/* dims[3] = ranks per dimension; rank order is MPI_Cart's (rank = (cx·dims[1] + cy)·dims[2] + cz); devices[rank] */
ns3d_mgpu *ns3d_mgpu_create_cart(const int *dims, const int *devices, int nx, int ny, int nz, int flags);
/* MPI_Dims_create: entries > 0 of dims[3] are kept, zeros are filled with the most balanced factorisation, non-increasing
 * (P = 8 → 2,2,2; 4 → 2,2,1; 2 → 2,1,1; 12 → 3,2,2) — what init_global_grid(nx,ny,nz) uses when dimx/dimy/dimz are not given */
int ns3d_dims_create(int P, int *dims);
/* One process per GPU: rank 0 calls ns3d_mgpu_unique_id and distributes the NS3D_UNIQUE_ID_BYTES bytes (MPI.Bcast in the
 * reference's setting), then every rank calls ns3d_mgpu_create_rank (collective).  RCCL is loaded at run time. */
int ns3d_mgpu_unique_id(void *id_out);
ns3d_mgpu *ns3d_mgpu_create_rank(int P, int rank, int device, const void *unique_id, int nx, int ny, int nz_local, int flags);
ns3d_mgpu *ns3d_mgpu_create_rank_cart(const int *dims, int rank, int device, const void *unique_id, int nx, int ny, int nz,
                                      int flags);
void ns3d_mgpu_destroy(ns3d_mgpu *m);
int ns3d_mgpu_world(const ns3d_mgpu *m);               /* P */
int ns3d_mgpu_nlocal(const ns3d_mgpu *m);
int ns3d_mgpu_rank(const ns3d_mgpu *m, int local);     /* "me" of a local rank (= its z coordinate for z-slabs) */
int ns3d_mgpu_dims(const ns3d_mgpu *m, int *dims_out);                 /* 3 ints */
int ns3d_mgpu_coords(const ns3d_mgpu *m, int local, int *coords_out);  /* 3 ints: MPI_Cart_coords of a local rank */
int ns3d_mgpu_n_g(const ns3d_mgpu *m, int *n_g_out);                   /* nx_g(), ny_g(), nz_g(): dims·(n−2)+2 */
ns3d_ctx *ns3d_mgpu_ctx(ns3d_mgpu *m, int local);
int ns3d_mgpu_nz_g(const ns3d_mgpu *m);
const char *ns3d_mgpu_transport(const ns3d_mgpu *m);   /* "peer" | "rccl" */
int ns3d_mgpu_rccl_ranks(const ns3d_mgpu *m);          /* ncclCommCount of the communicator (0 in the one-process form) */
int ns3d_mgpu_sync(ns3d_mgpu *m);
int ns3d_max_g(ns3d_mgpu *m, const double *local_max, double *out);    /* NaN-propagating */
/* Most PT iterations a pass over memory may advance in ns3d_slab_* / ns3d_pt_solve_slab = ghost planes per seam + 1: 4 (default),
 * 3, 2 or 1 (plain one-plane halo, single sweeps).  Passes run two iterations until ns3d_slab_plan has measured whether three
 * or four pay on this grid; every rank uses the same depth. */
int ns3d_mgpu_set_temporal(ns3d_mgpu *m, int depth);
int ns3d_mgpu_pass_depth(const ns3d_mgpu *m);
/* Making room for the exchange's kernels while the interior sweep runs (round 4; results do not depend on either):
 * ns3d_mgpu_reserve_cus — ns3d_reserve_cus on every local rank's context (the seam sweeps and the exchange stay on the unmasked
 * high-priority communication stream); ns3d_mgpu_set_interior_chunks — the interior sweep of a z-slab pass goes out as `chunks`
 * launches over consecutive plane ranges (1 = one launch; NS3D_SLAB_INTERIOR_CHUNKS presets it), so that a pending send/recv
 * finds CUs at the latest when a chunk ends. */
int ns3d_mgpu_reserve_cus(ns3d_mgpu *m, int n_cus);
int ns3d_mgpu_set_interior_chunks(ns3d_mgpu *m, int chunks);
/* ghost planes per seam of the loaded solve state: (deepest pass allowed) - 1 after ns3d_slab_load, pass depth - 1 once
 * ns3d_slab_plan has agreed on the iterations per pass; -1 when nothing is loaded */
int ns3d_mgpu_ghost_depth(const ns3d_mgpu *m);
/* The pseudo-transient state of a z-slab rank lives in library-owned buffers extended by the ghost planes temporal
 * blocking needs: load → iterate / residual → store; ns3d_pt_solve_slab is the whole inner loop multi.jl:458-471
 * (load, plan, iterate with a global residual check every nchk iterations, store).  Iterates are bit-identical to the
 * single-device solve of the global grid.  p describes the local grid of every rank; p->owns_outlet says whether the GLOBAL
 * x-hi face carries the outlet rule (multi.jl:179) — the library applies it on the ranks that hold that face (all of them on
 * z-slabs) — and the z halo flags of p are ignored (set per rank inside).  (The seam planes of divV are exchanged at load: multi.jl:455's update_halo!(∇V)
 * may but need not have run.) */
int ns3d_slab_iterate(ns3d_mgpu *m, int n_iters);
int ns3d_slab_plan(ns3d_mgpu *m);
int ns3d_slab_residual(ns3d_mgpu *m, double *out);

#define NS3D_MGPU_DECL(T, S)                                                                                \
    /* extents: 3 ints (sx,sy,sz) per field */                                                              \
    int ns3d_update_halo_##S(ns3d_mgpu *m, T *const *fields, const int *extents, int nfields);              \
    /* halo-stripped blocks A[2:end-1,2:end-1,2:end-1] of every rank, placed side by side in rank-coordinate \
     * order, into the column-major out_host (dims[0]·(sx-2) × dims[1]·(sy-2) × dims[2]·(sz-2) elements;     \
     * rank 0's process only in the one-process-per-GPU form) */                                             \
    int ns3d_gather_##S(ns3d_mgpu *m, const T *const *A, int sx, int sy, int sz, T *out_host);              \
    int ns3d_slab_load_##S(ns3d_mgpu *m, const T *const *Pr, const T *const *dPrdtau, const T *const *divV, \
                           const ns3d_pt_params *p);                                                        \
    int ns3d_slab_store_##S(ns3d_mgpu *m, T *const *Pr, T *const *dPrdtau);                                 \
    /* {X_o .= X; advect!; update_halo!} (multi.jl:475-477) on z-slab ranks with the old fields' z halo widened to TWO planes — \
     * an option OUTSIDE the reference's multi-rank semantics: backtrack! clamps to the local array (multi.jl:192-195), so a      \
     * departure point beyond the one-plane halo is clamped on a rank where the one-rank run reads the neighbour.  Here every   \
     * rank advects on copies extended by one more plane per seam, all four new fields (C too) get their halo, and P ranks      \
     * reproduce the one-rank time step bit for bit while |δz| < 2 cells.  Per-rank pointer lists as everywhere. */            \
    int ns3d_advect_wide_##S(ns3d_mgpu *m, T *const *Vx, T *const *Vx_o, T *const *Vy, T *const *Vy_o, T *const *Vz,           \
                             T *const *Vz_o, T *const *C, T *const *C_o, double dt, double dx, double dy, double dz,          \
                             int faithful);                                                                  \
    int ns3d_pt_solve_slab_##S(ns3d_mgpu *m, T *const *Pr, T *const *dPrdtau, const T *const *divV,         \
                               const ns3d_pt_params *p, double eps, int niter, int nchk, double err_mul,    \
                               double err_div, int *iters_done, double *err_hist, int max_checks,           \
                               int *n_checks);
NS3D_MGPU_DECL(double, f64)
NS3D_MGPU_DECL(float, f32)
#undef NS3D_MGPU_DECL

#ifdef __cplusplus
}
#endif
#endif /* NS3D_H */
