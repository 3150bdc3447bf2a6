"""Host-side mirror of the reference's two drivers, running on libns3d.so (HIP, gfx950).

    run_navierstokes3D(; do_vis, do_save, do_print, nx, nt)  ↔ scripts/NavierStokes3D_multi_gpu.jl:287-536
    runme(; do_vis, do_save)                                  ↔ scripts/NavierStokes3D_gpu.jl:12-173

Same keyword names, same return contract (C_v,Pr_v,Vx_v,Vy_v,Vz_v: halo-stripped global host arrays on rank 0,
multi.jl:528-535).  Extra keyword-only options select the arithmetic mode, the fused PT path and the z-slab
process grid.  do_vis writes the reference's mid-plane heat maps as bare PNG files (vis.py: same names, slices and colour
limits; no axes — Plots.jl itself is out of scope).

The time loops below follow the reference line by line (cited); with `fused=True` the inner pseudo-transient
loop {update_dPrdτ!; update_Pr!; set_bc_Pr!}×n is replaced by ns3d_pt_solve / overlapped ns3d_pt_sweep calls
that produce the same iterates (the redundant halo updates multi.jl:460,462 are dropped).
"""
import math
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import kernels as K
from . import lib as L
from .halo import ZSlabGrid
from .slab import SlabPTSolver
from .params import gpu_params, multi_params
from .vis import save_frame_gpu, save_frame_multi


def _alloc(nx, ny, nz, dtype, device):
    """multi.jl:343-360 / gpu.jl:65-82"""
    z = lambda *s: K.zeros(s, dtype, device)
    f = SimpleNamespace()
    f.Pr = z(nx, ny, nz); f.dPrdtau = z(nx - 2, ny - 2, nz - 2)
    f.C = z(nx, ny, nz); f.C_o = z(nx, ny, nz)
    f.txx = z(nx, ny, nz); f.tyy = z(nx, ny, nz); f.tzz = z(nx, ny, nz)
    f.txy = z(nx - 1, ny - 1, nz - 1); f.txz = z(nx - 1, ny - 1, nz - 1); f.tyz = z(nx - 1, ny - 1, nz - 1)
    f.Vx = z(nx + 1, ny, nz); f.Vy = z(nx, ny + 1, nz); f.Vz = z(nx, ny, nz + 1)
    f.Vx_o = z(nx + 1, ny, nz); f.Vy_o = z(nx, ny + 1, nz); f.Vz_o = z(nx, ny, nz + 1)
    f.divV = z(nx, ny, nz); f.Rp = z(nx - 2, ny - 2, nz - 2)
    return f


def save_array(Aname, A):
    """save_array(Aname, A) (multi.jl:27-30): raw little-endian column-major dump to Aname.bin."""
    with open(Aname + ".bin", "wb") as fh:
        fh.write(np.asfortranarray(A).tobytes(order="F"))


def _inner(t):
    """Array(A)[2:end-1,2:end-1,2:end-1] (multi.jl:399)"""
    return K.to_numpy(t)[1:-1, 1:-1, 1:-1]


def _gather_all(grid, fs):
    """multi.jl:399-403 / :528-532 for the local ranks' fields `fs`"""
    if hasattr(grid, "gather_fields"):                     # C-ABI grid: halo-stripping and transport inside libns3d
        return tuple(grid.gather_fields([getattr(f, n) for f in fs]) for n in ("C", "Pr", "Vx", "Vy", "Vz"))
    return tuple(grid.gather(_inner(getattr(fs[0], n))) for n in ("C", "Pr", "Vx", "Vy", "Vz"))


def _is_root(grid):
    return grid.is_root() if hasattr(grid, "is_root") else grid.me == 0


def _save_frame(grid, fields_v, iframe):
    """multi.jl:404-413 / :515-522 — rank 0 writes Float32 raw dumps of the gathered arrays."""
    if not _is_root(grid):
        return
    os.makedirs("./out_save", exist_ok=True)
    for name, A in zip(("C", "Pr", "Vx", "Vy", "Vz"), fields_v):
        save_array("out_save/out_%s_v_%04d" % (name, iframe), A.astype(np.float32))


def pt_loop_reference(ctxs, grid, fs, ps, niter, do_print=False):
    """The inner loop exactly as written in multi.jl:458-471 (including its redundant halo updates); `fs`, `ps`, `ctxs`
    hold one entry per local rank."""
    p = ps[0]
    errs, done = [], niter
    col = lambda n: [getattr(f, n) for f in fs]
    for it in range(1, niter + 1):
        for f, c in zip(fs, ctxs):
            K.update_dPrdtau(f.Pr, f.dPrdtau, f.divV, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, ctx=c)   # :459
        grid.update_halo(col("divV"))                                                                     # :460
        for f, c in zip(fs, ctxs):
            K.update_Pr(f.Pr, f.dPrdtau, p.dtau, ctx=c)                                                   # :461
        grid.update_halo(col("Pr"))                                                                       # :462
        for f, c, q in zip(fs, ctxs, ps):
            K.set_bc_Pr_multi(f.Pr, q.owns_outlet, 0.0, ctx=c)                                            # :463 → :176-181
        grid.update_halo(col("Pr"))                                                                       # :182
        if it % p.nchk == 0:                                                                              # :464
            loc = []
            for f, c in zip(fs, ctxs):
                K.compute_res(f.Rp, f.Pr, f.divV, p.rho, p.dt, p.dx, p.dy, p.dz, ctx=c)                   # :465
                loc.append(K.max_abs(f.Rp, ctx=c))
            err = grid.max_g(loc if len(loc) > 1 else loc[0]) * (p.ly * p.ly) / p.psc                      # :466
            errs.append(err)
            if _is_root(grid) and do_print:
                print("  #iter = %d, err = %1.3e" % (it, err))                                            # :468
            if err < p.eps or not math.isfinite(err):                                                     # :469
                done = it
                break
    return done, errs


def pt_loop_fused_slab(ctx, grid, f, p, pt, niter, do_print=False, scratch=None):
    """Same iterates as pt_loop_reference on a z-slab rank: per iteration the two seam-adjacent interior planes
    are swept first, their exchange over xGMI is posted, and the interior sweep runs behind it."""
    nz = p.nz
    Pa = f.Pr
    Pb = scratch if scratch is not None else K.clone(f.Pr)
    if scratch is not None:
        K.copy(Pb, Pa, ctx=ctx)   # seeds faces/halo planes
    errs, done = [], niter
    for it in range(1, niter + 1):
        K.pt_sweep(Pa, Pb, f.dPrdtau, f.divV, pt, 1, 2, ctx=ctx)
        if nz - 2 > 1:
            K.pt_sweep(Pa, Pb, f.dPrdtau, f.divV, pt, nz - 2, nz - 1, ctx=ctx)
        work = grid.start_halo(Pb)
        if nz - 2 > 2:
            K.pt_sweep(Pa, Pb, f.dPrdtau, f.divV, pt, 2, nz - 2, ctx=ctx)
        grid.finish_halo(work)
        Pa, Pb = Pb, Pa
        if it % p.nchk == 0:
            err = grid.max_g(K.residual_max(Pa, f.divV, pt, ctx=ctx)) * (p.ly * p.ly) / p.psc
            errs.append(err)
            if grid.me == 0 and do_print:
                print("  #iter = %d, err = %1.3e" % (it, err))
            if err < p.eps or not math.isfinite(err):
                done = it
                break
    if Pa is not f.Pr:
        K.copy(f.Pr, Pa, ctx=ctx)
    return done, errs


def run_navierstokes3D(do_vis=False, do_save=False, do_print=False, nx=255, nt=10, *, mode="strict", fused=True,
                       temporal=True, dtype=torch.float64, faithful=True, grid=None, device=None, niter_cap=None,
                       return_info=False, shape=None, pressure="pt", wide_advect_halo=False, one_call=True):
    """run_navierstokes3D (multi.jl:287-536).  nx is the LOCAL streamwise size (ny = nz = ceil(0.6 nx) local).
    `grid` decides the decomposition: None = one rank; a halo.ZSlabGrid = this process is one z-slab rank of an
    initialised torch.distributed group; a mgpu.MgpuGrid = the C-ABI grid (this process drives every local rank of an
    ns3d_mgpu: all P of them in the one-process form, one under RCCL) — z-slabs, or any Cartesian topology with
    fused=False (ImplicitGlobalGrid's own default is MultiGpu.dims_create(P), e.g. (2,2,2) for 8 ranks).
    `shape` (dict: ny, nz, ly_lx, lz_lx) overrides the literals multi.jl:302-303,323-324 for grids the reference cannot
    produce without editing them (BASELINE configs[3]: 512×512×1024 global).
    pressure="direct" (OUTSIDE PARITY; one rank): the inner loop :458-471 is replaced by ns3d_poisson_direct, the
    exact solution of the discrete problem that loop stops short of by εit; info.iters is 0 per step and info.errs holds the
    residual of the solution in the reference's own measure (:466).
    wide_advect_halo=True (OUTSIDE the reference's multi-rank semantics; z-slabs on the C-ABI grid): :475-477 run as
    ns3d_advect_wide — old fields with a two-plane z halo, C's halo updated too — so that the P-rank run reproduces the one-rank
    run of the same global grid bit for bit (the reference's backtrack! clamps to the local array)."""
    if pressure not in ("pt", "direct"):
        raise L.Ns3dError("pressure = %r (\"pt\" | \"direct\")" % (pressure,))
    if device is None:
        device = torch.cuda.current_device()
    shape = dict(shape or {})
    if grid is None:
        p0 = multi_params(nx, **shape)
        grid = ZSlabGrid(p0.nx, p0.ny, p0.nz)                                                  # :325
    P = grid.P
    local_ranks = list(getattr(grid, "local_ranks", [grid.me]))
    if hasattr(grid, "contexts"):
        ctxs = list(grid.contexts)                         # contexts of the ns3d_mgpu's ranks (their devices)
    else:
        ctxs = [K.Context(device, mode, async_=True)]
    dims = tuple(getattr(grid, "dims", (1, 1, P)))
    coords = list(getattr(grid, "local_coords", [(0, 0, me) for me in local_ranks]))
    ps = [multi_params(nx, dims=dims, coords=c3, **shape) for c3 in coords]
    p = ps[0]
    if fused and (dims[0] > 1 or dims[1] > 1) and getattr(grid, "mg", None) is None:
        raise L.Ns3dError("a topology decomposed in x or y (dims = %r) needs the C-ABI grid (mgpu.MgpuGrid) or fused=False" % (dims,))
    # multi.jl:179 `xve_g == lx/2`: the outlet rule as the ranks on the last x coordinate evaluate it (the library applies it
    # on those ranks only; every z-slab rank is one of them)
    outlet_rule = multi_params(nx, dims=dims, coords=(dims[0] - 1, 0, 0), **shape).owns_outlet
    nx, ny, nz = p.nx, p.ny, p.nz
    niter = p.niter if niter_cap is None else min(p.niter, niter_cap)
    fs = [_alloc(nx, ny, nz, dtype, torch.device("cuda", c.device)) for c in ctxs]             # :343-360
    col = lambda n: [getattr(f, n) for f in fs]
    # initialization :369-373
    for f, q in zip(fs, ps):
        me = q.coords[2]
        f.Vy[0, :, :] = q.vin                                                                 # :369 (sic)
        # :370 Pr = -(z_g-dz/2)*ρ*g (+0+0): identically 0 because g = 1/Fr² = 0 (:316); evaluated for fidelity
        zg = np.array([(me * (nz - 2) + iz) * q.dz for iz in range(nz)])
        f.Pr[:, :, :] = torch.from_numpy(-(zg - q.dz / 2) * q.rho * q.g + 0.0).to(f.Pr.device, dtype)[None, None, :]
    if any(getattr(c, "_pinned", False) for c in ctxs):    # ranks on streams of their own: the torch writes above come first
        torch.cuda.synchronize()
    grid.update_halo(col("Pr"))                                                               # :371
    cyls = [(q.a2, q.b2, q.ox, q.oy, q.sinb, q.cosb, q.xco_g, q.yco_g, q.zco_g, q.lx, q.ly, q.lz, q.dx, q.dy, q.dz)
            for q in ps]
    for f, c, cyl in zip(fs, ctxs, cyls):
        K.set_cylinder(f.C, f.Vx, f.Vy, f.Vz, *cyl, ctx=c)                                    # :372
    grid.update_halo(col("C"), col("Vx"), col("Vy"), col("Vz"))                               # :373
    sync = lambda: [c.sync() for c in ctxs]
    iframe = 0
    nz_g_all = dims[2] * (nz - 2) + 2
    ny_g_all = dims[1] * (ny - 2) + 2
    if do_save or do_vis:                                                                     # :399-403
        sync()
        gathered = _gather_all(grid, fs)
        if do_save:                                                                           # :404-413
            _save_frame(grid, gathered, iframe)
        if do_vis and _is_root(grid):                                                         # :416-443
            save_frame_multi(gathered, ny_g_all, nz_g_all, iframe)
    iframe += 1
    pts = [K.pt_params(f.Pr, q.rho, q.dt, q.dtau, q.damp, q.dx, q.dy, q.dz, L.NS3D_BC_MULTI, q.owns_outlet, 0.0, q.g,
                       q.coords[2] > 0, q.coords[2] < dims[2] - 1) for f, q in zip(fs, ps)]
    pt_all = K.pt_params(fs[0].Pr, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, L.NS3D_BC_MULTI, outlet_rule, 0.0, p.g,
                         False, False)      # ns3d_pt_solve_slab: one set for every local rank, halo flags set inside
    if not temporal:
        for c in ctxs:
            c.set_pt2_variant(-1)
    mg = getattr(grid, "mg", None)
    if mg is not None:
        mg.set_temporal(4 if (temporal and nz >= 4) else 1)
    slab = None
    if fused and P > 1 and mg is None and temporal and nz >= 4:
        slab = SlabPTSolver(ctxs[0], grid, fs[0].Pr, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, L.NS3D_BC_MULTI,
                            p.owns_outlet, 0.0, p.g)
    scratch = K.clone(fs[0].Pr) if (fused and P > 1 and mg is None and slab is None) else None
    info = SimpleNamespace(iters=[], errs=[], params=p)
    nsave = nvis = 10                                                                         # :330,332
    root = _is_root(grid)
    # :450's halo update of the normal stresses may only go with the stress arrays where the interior planes two ranks both compute
    # are identical.  backtrack! clamps to the LOCAL array and rounds iz − δ rank by rank, so with the fixed third branch
    # (faithful=False: Vz IS advected) on several ranks and without the wide halo the two copies of a seam's Vz plane can differ —
    # and the reference sequence would ship one rank's τzz to the other.  That combination keeps :449-451 literal.
    fuse_predictor = fused and (P == 1 or faithful or wide_advect_halo)
    # one rank, fused: the whole step :449-477 is ONE library call (ns3d_time_step: the same entry points, enqueued from C) — a step
    # of the 255×153×153 case with the direct pressure solve is otherwise mostly this loop's ≈40 calls
    step_params = None
    if fused and one_call and P == 1 and getattr(grid, "mg", None) is None:
        q = ps[0]
        step_params = L.StepParams(script=L.NS3D_BC_MULTI, nx=nx, ny=ny, nz=nz, mu=p.mu, rho=p.rho, g=p.g, dt=p.dt, dtau=p.dtau,
                                   damp=p.damp, dx=p.dx, dy=p.dy, dz=p.dz, eps=p.eps, niter=niter, nchk=p.nchk, err_mul=p.ly * p.ly,
                                   err_div=p.psc, a2=q.a2, b2=q.b2, ox=q.ox, oy=q.oy, sinb=q.sinb, cosb=q.cosb, xco_g=q.xco_g,
                                   yco_g=q.yco_g, zco_g=q.zco_g, lx=q.lx, ly=q.ly, lz=q.lz, owns_inlet=int(bool(q.owns_inlet)),
                                   owns_outlet=int(bool(q.owns_outlet)), vin=p.vin, faithful=int(bool(faithful)),
                                   pressure=1 if pressure == "direct" else 0, write_stress=0)

    def frames(it):                                                                           # :479-525 (one frame counter)
        nonlocal iframe
        if (do_vis and it % nvis == 0) or (do_save and it % nsave == 0):
            sync()
            gathered = _gather_all(grid, fs)
            if do_vis and it % nvis == 0 and root:                                            # :486-513
                save_frame_multi(gathered, ny_g_all, nz_g_all, iframe)
            if do_save and it % nsave == 0:                                                   # :515-522
                _save_frame(grid, gathered, iframe)
            iframe += 1

    for it in range(1, nt + 1):                                                               # :446
        if step_params is not None:
            step_params.write_stress = 1 if it == nt else 0     # the stress arrays as the reference leaves them after its last step
            done, errs = K.time_step(fs[0], step_params, ctx=ctxs[0])
            if root and do_print:
                print("#it = %d" % it)                                                        # :456
                for q_, e in enumerate(errs):
                    print("  #iter = %d, err = %1.3e" % ((q_ + 1) * p.nchk, e))
            info.iters.append(done); info.errs.append(errs)
            frames(it)
            continue
        if fuse_predictor:
            # :449-451 in one pass (ns3d_predict_fused): the stresses evaluated on the fly, the predicted fields written into the
            # *_o buffers (free until :475), the names swapped.  :450's halo update of τxx, τyy, τzz moves values every rank has
            # already computed itself — the velocities it computes them from are consistent across ranks after :477 / :373 —
            # and is dropped with the arrays (the multi-rank driver tests compare with the oracle, which performs it)
            if it == nt:        # the stress arrays as the reference leaves them after its last step: same final state
                for f, c in zip(fs, ctxs):
                    K.update_tau(f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, f.Vx, f.Vy, f.Vz, p.mu, p.dx, p.dy, p.dz, ctx=c)
                grid.update_halo(col("txx"), col("tyy"), col("tzz"))
            for f, c in zip(fs, ctxs):
                _predict_swap(f, p, c)
        else:
            for f, c in zip(fs, ctxs):
                K.update_tau(f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, f.Vx, f.Vy, f.Vz, p.mu, p.dx, p.dy, p.dz, ctx=c)  # :449
            grid.update_halo(col("txx"), col("tyy"), col("tzz"))                              # :450
            for f, c in zip(fs, ctxs):
                K.predict_V(f.Vx, f.Vy, f.Vz, f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, p.rho, p.g, p.dt, p.dx, p.dy, p.dz,
                            ctx=c)                                                            # :451
        for f, c, cyl in zip(fs, ctxs, cyls):
            K.set_cylinder(f.C, f.Vx, f.Vy, f.Vz, *cyl, ctx=c)                                # :452
        grid.update_halo(col("C"), col("Vx"), col("Vy"), col("Vz"))                           # :453
        for f, c in zip(fs, ctxs):
            K.update_divV(f.divV, f.Vx, f.Vy, f.Vz, p.dx, p.dy, p.dz, ctx=c)                  # :454
        grid.update_halo(col("divV"))                                                         # :455
        if root and do_print:
            print("#it = %d" % it)                                                            # :456
        show = (lambda i, e: print("  #iter = %d, err = %1.3e" % (i, e))) if (root and do_print) else None
        if pressure == "direct":
            if P != 1:
                raise L.Ns3dError("pressure=\"direct\" solves a single rank's closed problem (P = %d ranks here)" % P)
            K.poisson_direct(fs[0].Pr, fs[0].dPrdtau, fs[0].divV, pts[0], ctx=ctxs[0])
            done, errs = 0, [K.residual_max(fs[0].Pr, fs[0].divV, pts[0], ctx=ctxs[0]) * (p.ly * p.ly) / p.psc]
        elif not fused:                                                                       # :458-471
            done, errs = pt_loop_reference(ctxs, grid, fs, ps, niter, do_print)
        elif P == 1:
            done, errs = K.pt_solve(fs[0].Pr, fs[0].dPrdtau, fs[0].divV, pts[0], p.eps, niter, p.nchk, p.ly * p.ly, p.psc,
                                    ctx=ctxs[0])
        elif mg is not None:                                    # the whole loop inside libns3d (ns3d_pt_solve_slab)
            done, errs = mg.pt_solve_slab(col("Pr"), col("dPrdtau"), col("divV"), pt_all, p.eps, niter, p.nchk,
                                          p.ly * p.ly, p.psc)
        elif slab is not None:                                  # z-slab rank over torch.distributed, deep ghosts
            slab.load(fs[0].Pr, fs[0].dPrdtau, fs[0].divV)
            done, errs = slab.solve(p.eps, niter, p.nchk, p.ly * p.ly, p.psc, show)
            slab.store(fs[0].Pr, fs[0].dPrdtau)
            show = None
        else:
            done, errs = pt_loop_fused_slab(ctxs[0], grid, fs[0], p, pts[0], niter, do_print, scratch)
            show = None
        if show is not None and fused:
            for q, e in enumerate(errs):
                show((q + 1) * p.nchk, e)
        info.iters.append(done); info.errs.append(errs)
        for f, c, cyl, q in zip(fs, ctxs, cyls, ps):
            K.correct_V(f.Vx, f.Vy, f.Vz, f.Pr, p.dt, p.rho, p.dx, p.dy, p.dz, ctx=c)          # :472
            K.set_cylinder(f.C, f.Vx, f.Vy, f.Vz, *cyl, ctx=c)                                # :473
            K.set_bc_Vel_multi(f.Vx, f.Vy, f.Vz, q.owns_inlet, p.vin, ctx=c)                   # :474 → :157-166
        grid.update_halo(col("Vx"), col("Vy"), col("Vz"))                                     # :167
        if wide_advect_halo and P > 1:                                                         # :475-477, decomposition-independent
            if mg is None or not (dims[0] == 1 and dims[1] == 1):
                raise L.Ns3dError("wide_advect_halo needs z-slab ranks on the C-ABI grid (mgpu.MgpuGrid)")
            mg.advect_wide(col("Vx"), col("Vx_o"), col("Vy"), col("Vy_o"), col("Vz"), col("Vz_o"), col("C"), col("C_o"),
                           p.dt, p.dx, p.dy, p.dz, faithful)
        for f, c in zip(fs, ctxs):
            if wide_advect_halo and P > 1:
                break
            if fused:       # :475-476 in one pass (ns3d_copy_advect): complete new fields into the *_o buffers, then the roles swap
                _copy_advect_swap(f, p, faithful, c)
                continue
            K.copy(f.Vx_o, f.Vx, ctx=c); K.copy(f.Vy_o, f.Vy, ctx=c)                          # :475
            K.copy(f.Vz_o, f.Vz, ctx=c); K.copy(f.C_o, f.C, ctx=c)
            K.advect(f.Vx, f.Vx_o, f.Vy, f.Vy_o, f.Vz, f.Vz_o, f.C, f.C_o, p.dt, p.dx, p.dy, p.dz, faithful, ctx=c)  # :476
        if not (wide_advect_halo and P > 1):
            grid.update_halo(col("Vx"), col("Vy"), col("Vz"))                                 # :477 (not C)
        frames(it)
    if fused and faithful and nt > 0 and not (wide_advect_halo and P > 1):
        for f, c in zip(fs, ctxs):
            K.copy(f.Vz_o, f.Vz, ctx=c)     # the one copy of :475 the swaps skipped (Vz is never advected): same final state
    sync()
    out = _gather_all(grid, fs)                                                               # :528-532
    info.fields = fs[0]
    info.local_fields = fs
    info.ctx = ctxs[0]
    return out + ((info,) if return_info else ())                                             # :535


def _predict_swap(f, p, ctx):
    """{update_τ!; predict_V!} (multi.jl:449,451 / gpu.jl:121-122) without the stress arrays: ns3d_predict_fused reads Vx, Vy, Vz
    and writes the complete predicted fields into the *_o buffers — idle between advect! and the next old-field copies — then
    the names swap."""
    K.predict_fused(f.Vx_o, f.Vy_o, f.Vz_o, f.Vx, f.Vy, f.Vz, p.mu, p.rho, p.g, p.dt, p.dx, p.dy, p.dz, ctx=ctx)
    f.Vx, f.Vx_o = f.Vx_o, f.Vx
    f.Vy, f.Vy_o = f.Vy_o, f.Vy
    f.Vz, f.Vz_o = f.Vz_o, f.Vz


def _copy_advect_swap(f, p, faithful, ctx):
    """{Vx_o,Vy_o,Vz_o,C_o .= Vx,Vy,Vz,C; advect!} (multi.jl:475-476 / gpu.jl:141-142) without the four copies:
    ns3d_copy_advect reads the current fields and writes COMPLETE new ones into the *_o buffers, then the names swap — X is
    the advected field and X_o the previous one, exactly what the copies leave behind.  Vz is never advected in faithful
    mode (App. B1): it stays where it is, and Vz_o is brought up to date once at the end of the run."""
    K.copy_advect(f.Vx_o, f.Vx, f.Vy_o, f.Vy, f.Vz if faithful else f.Vz_o, f.Vz, f.C_o, f.C, p.dt, p.dx, p.dy, p.dz, faithful,
                  ctx=ctx)
    f.Vx, f.Vx_o = f.Vx_o, f.Vx
    f.Vy, f.Vy_o = f.Vy_o, f.Vy
    f.C, f.C_o = f.C_o, f.C
    if not faithful:
        f.Vz, f.Vz_o = f.Vz_o, f.Vz


def gpu_initial_fields(p):
    """gpu.jl:62-63,85-88: LinRange cell centres, 1/7-power-law Vx profile, hydrostatic Pr (host setup)."""
    nx, ny, nz = p.nx, p.ny, p.nz
    t = np.arange(nz, dtype=np.float64) / (nz - 1)                    # Julia LinRange: (1-t)*a + t*b
    zc = (1 - t) * (-(p.lz - p.dz) / 2) + t * ((p.lz - p.dz) / 2)
    prof = p.vin * (7.0 / 6.0) * ((zc + p.lz / 2) / p.lz) ** (1.0 / 6.0)
    Vx = np.empty((nx + 1, ny, nz), order="F"); Vx[:, :, :] = prof[None, None, :]
    Pr = np.empty((nx, ny, nz), order="F"); Pr[:, :, :] = (-(zc - p.lz / 2) * p.rho * p.g)[None, None, :]
    return Vx, Pr


def _save_mat(path, f, p, step0):
    """matwrite("out_save/step_$it.mat", Dict("Pr","Vx","Vy","Vz","C","dx","dy","dz")) (gpu.jl:168-170); the step-0 dictionary
    of gpu.jl:89 repeats the key "Vy" — the later pair wins in a Julia Dict literal, so it holds Vz's array under "Vy" and has
    no "Vz" — reproduced.  Written as MAT v5 (scipy.io.savemat; MAT.jl writes v7.3/HDF5 by default and reads both, as do MATLAB
    and Octave): same variable names, shapes and Float64 values, different container version."""
    from scipy.io import savemat
    d = {"Pr": K.to_numpy(f.Pr), "Vx": K.to_numpy(f.Vx), "C": K.to_numpy(f.C), "dx": p.dx, "dy": p.dy, "dz": p.dz}
    if step0:
        d["Vy"] = K.to_numpy(f.Vz)
    else:
        d["Vy"] = K.to_numpy(f.Vy)
        d["Vz"] = K.to_numpy(f.Vz)
    savemat(path, d, format="5", do_compression=False, oned_as="column")


def runme(do_vis=False, do_save=False, *, nx=255, nt=10000, mode="strict", fused=True, dtype=torch.float64,
          faithful=True, device=None, niter_cap=None, do_print=False, initial=None, pressure="pt", one_call=True):
    """runme (gpu.jl:12-173): single device, gravity, hydrostatic x-planes.  Returns (fields, info).
    nx/nt are literals in the reference (gpu.jl:44,51: 255, 10000) and keyword options here.  do_save writes the MAT files
    of gpu.jl:89,168-170 (step 0 and every nsave = 10 steps); do_vis the heat maps of gpu.jl:90-117,143-167 (frame 0 and
    every nvis = 10 steps)."""
    if device is None:
        device = torch.cuda.current_device()
    dev = torch.device("cuda", device)
    ctx = K.Context(device, mode, async_=True)
    p = gpu_params(nx)
    nx, ny, nz = p.nx, p.ny, p.nz
    niter = p.niter if niter_cap is None else min(p.niter, niter_cap)
    f = _alloc(nx, ny, nz, dtype, dev)
    Vx0, Pr0 = initial if initial is not None else gpu_initial_fields(p)
    f.Vx = K.from_numpy(np.asarray(Vx0).astype(np.float64 if dtype == torch.float64 else np.float32), dev)
    f.Pr = K.from_numpy(np.asarray(Pr0).astype(np.float64 if dtype == torch.float64 else np.float32), dev)
    cyl = (p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.lx, p.ly, p.lz, p.dx, p.dy, p.dz)
    pt = K.pt_params(f.Pr, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, L.NS3D_BC_GPU, False, 0.0, p.g)
    info = SimpleNamespace(iters=[], errs=[], params=p)
    nsave = nvis = 10                                                                         # :50,52
    if do_save:                                                                               # :89
        os.makedirs("./out_save", exist_ok=True)
        ctx.sync()
        _save_mat("out_save/step_0.mat", f, p, True)
    iframe = 0
    host = lambda: {n: K.to_numpy(getattr(f, n)) for n in ("Pr", "C", "Vx", "Vy", "Vz")}
    if do_vis:                                                                                # :90-117
        ctx.sync()
        save_frame_gpu(host(), ny, nz, iframe)
        iframe += 1
    step_params = None
    if fused and one_call:      # the whole step :121-142 as ONE library call (ns3d_time_step)
        step_params = L.StepParams(script=L.NS3D_BC_GPU, nx=nx, ny=ny, nz=nz, mu=p.mu, rho=p.rho, g=p.g, dt=p.dt, dtau=p.dtau,
                                   damp=p.damp, dx=p.dx, dy=p.dy, dz=p.dz, eps=p.eps, niter=niter, nchk=p.nchk, err_mul=p.ly * p.ly,
                                   err_div=p.psc, a2=p.a2, b2=p.b2, ox=p.ox, oy=p.oy, sinb=p.sinb, cosb=p.cosb, xco_g=0.0, yco_g=0.0,
                                   zco_g=0.0, lx=p.lx, ly=p.ly, lz=p.lz, owns_inlet=0, owns_outlet=0, vin=0.0,
                                   faithful=int(bool(faithful)), pressure=1 if pressure == "direct" else 0, write_stress=0)
    for it in range(1, nt + 1):                                                               # :119
        if step_params is not None:
            step_params.write_stress = 1 if it == nt else 0
            done, errs = K.time_step(f, step_params, ctx=ctx)
            if do_print:
                print("#it = %d" % it)                                                        # :125
                for q_, e in enumerate(errs):
                    print("  #iter = %d, err = %1.3e" % ((q_ + 1) * p.nchk, e))                # :134
            info.iters.append(done); info.errs.append(errs)
            if do_vis and it % nvis == 0:                                                     # :143-167
                ctx.sync()
                save_frame_gpu(host(), ny, nz, iframe)
                iframe += 1
            if do_save and it % nsave == 0:                                                   # :168-170
                ctx.sync()
                _save_mat("out_save/step_%d.mat" % it, f, p, False)
            continue
        if fused:
            if it == nt:            # the stress arrays as the reference leaves them after its last step: same final state
                K.update_tau(f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, f.Vx, f.Vy, f.Vz, p.mu, p.dx, p.dy, p.dz, ctx=ctx)
            _predict_swap(f, p, ctx)                                                          # :121-122 in one pass
        else:
            K.update_tau(f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, f.Vx, f.Vy, f.Vz, p.mu, p.dx, p.dy, p.dz, ctx=ctx)  # :121
            K.predict_V(f.Vx, f.Vy, f.Vz, f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, p.rho, p.g, p.dt, p.dx, p.dy, p.dz,
                        ctx=ctx)                                                              # :122
        K.set_cylinder(f.C, f.Vx, f.Vy, f.Vz, *cyl, ctx=ctx)                                  # :123
        K.update_divV(f.divV, f.Vx, f.Vy, f.Vz, p.dx, p.dy, p.dz, ctx=ctx)                    # :124
        if do_print:
            print("#it = %d" % it)                                                            # :125
        if pressure == "direct":        # outside parity: the exact solution of what :126-137 iterates towards
            K.poisson_direct(f.Pr, f.dPrdtau, f.divV, pt, ctx=ctx)
            done, errs = 0, [K.residual_max(f.Pr, f.divV, pt, ctx=ctx) * (p.ly * p.ly) / p.psc]
        elif fused:
            done, errs = K.pt_solve(f.Pr, f.dPrdtau, f.divV, pt, p.eps, niter, p.nchk, p.ly * p.ly, p.psc, ctx=ctx)
        else:
            errs, done = [], niter
            for itr in range(1, niter + 1):                                                   # :126
                K.update_dPrdtau(f.Pr, f.dPrdtau, f.divV, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, ctx=ctx)  # :127
                K.update_Pr(f.Pr, f.dPrdtau, p.dtau, ctx=ctx)                                 # :128
                K.set_bc_Pr_gpu(f.Pr, p.dz, nz, p.g, p.rho, ctx=ctx)                          # :129
                if itr % p.nchk == 0:                                                         # :130
                    K.compute_res(f.Rp, f.Pr, f.divV, p.rho, p.dt, p.dx, p.dy, p.dz, ctx=ctx)  # :131
                    err = K.max_abs(f.Rp, ctx=ctx) * (p.ly * p.ly) / p.psc                      # :132
                    errs.append(err)
                    if err < p.eps or not math.isfinite(err):                                 # :135
                        done = itr
                        break
        if do_print:
            for q, e in enumerate(errs):
                print("  #iter = %d, err = %1.3e" % ((q + 1) * p.nchk, e))                    # :134
        info.iters.append(done); info.errs.append(errs)
        K.correct_V(f.Vx, f.Vy, f.Vz, f.Pr, p.dt, p.rho, p.dx, p.dy, p.dz, ctx=ctx)            # :138
        K.set_cylinder(f.C, f.Vx, f.Vy, f.Vz, *cyl, ctx=ctx)                                  # :139
        K.set_bc_Vel_gpu(f.Vx, f.Vy, f.Vz, ctx=ctx)                                           # :140
        if fused:
            _copy_advect_swap(f, p, faithful, ctx)                                            # :141-142 in one pass
        else:
            K.copy(f.Vx_o, f.Vx, ctx=ctx); K.copy(f.Vy_o, f.Vy, ctx=ctx)                      # :141
            K.copy(f.Vz_o, f.Vz, ctx=ctx); K.copy(f.C_o, f.C, ctx=ctx)
            K.advect(f.Vx, f.Vx_o, f.Vy, f.Vy_o, f.Vz, f.Vz_o, f.C, f.C_o, p.dt, p.dx, p.dy, p.dz, faithful, ctx=ctx)  # :142
        if do_vis and it % nvis == 0:                                                         # :143-167
            ctx.sync()
            save_frame_gpu(host(), ny, nz, iframe)
            iframe += 1
        if do_save and it % nsave == 0:                                                       # :168-170
            ctx.sync()
            _save_mat("out_save/step_%d.mat" % it, f, p, False)
    if fused and faithful and nt > 0:
        K.copy(f.Vz_o, f.Vz, ctx=ctx)       # the one copy of :141 the swaps skipped (Vz is never advected): same final state
    ctx.sync()
    info.ctx = ctx
    return f, info
