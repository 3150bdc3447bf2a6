"""The multi-GPU half of the C ABI (ns3d_mgpu_*, include/ns3d.h) on the one GPU of the test box: ONE process drives P
virtual z-slab ranks that all sit on device 0 — the same schedule, events and peer copies that run across xGMI on a
multi-GPU node — and everything is compared with the oracle's virtual ranks (oracle/driver_ref.py: a literal restatement
of ImplicitGlobalGrid's update_halo!/gather!) or with the single-device solve of the global grid, bit for bit.
The one-process-per-GPU form is exercised as far as one GPU allows: an RCCL communicator of ONE rank (dlopen, unique id,
ncclCommInitRank, ncclAllReduce inside the residual check and max_g)."""
import math

import numpy as np
import pytest

from util import fields, geometry

pytestmark = pytest.mark.gpu


OWN_STREAMS = False


@pytest.fixture(autouse=True, params=["torch-stream", "own-streams"])
def rank_streams(request):
    """Every test of this module runs twice: with all virtual ranks following PyTorch's current stream of device 0 (one
    compute stream for everybody: only the communication streams are asynchronous) and with a NON-BLOCKING COMPUTE STREAM
    PER RANK, where the ev_ready / ev_landed handshake of exchange_begin / exchange_end (ns3d_mgpu.cpp) is what orders one
    rank's sweeps against its neighbours' pulls — as it is between the devices of a multi-GPU node."""
    global OWN_STREAMS
    import torch
    OWN_STREAMS = request.param == "own-streams"
    if OWN_STREAMS:
        torch.cuda.synchronize()
    yield
    torch.cuda.synchronize()
    OWN_STREAMS = False


def _mg(P, nx, ny, nz, mode="strict"):
    import torch
    from navierstokes3d_amd.mgpu import MultiGpu
    torch.cuda.synchronize()            # uploads made on PyTorch's stream are complete before any rank stream reads them
    return MultiGpu.create([0] * P, nx, ny, nz, mode, own_streams=OWN_STREAMS)


@pytest.mark.parametrize("P", [2, 3])
def test_update_halo_follows_implicit_global_grid(hip, P):
    """update_halo! (multi.jl:371…477) for every stagger: cell-centred (overlap 2), Vz (overlap 3), Vx/Vy (x/y staggered,
    overlap 2), and the arrays without a z halo (τxy (n-1)³, dPrdτ (n-2)³) that must come back untouched."""
    from oracle.driver_ref import update_halo_z
    nx, ny, nz = 13, 9, 7
    kinds = ["c", "vx", "vy", "vz", "s", "i"]
    host = [dict(zip(kinds, fields(nx, ny, nz, kinds, 100 * (r + 1)))) for r in range(P)]
    mg = _mg(P, nx, ny, nz)
    assert mg.transport == "peer" and mg.P == P and mg.nlocal == P and mg.ranks == list(range(P)) and mg.nz_g() == P * (nz - 2) + 2
    dev = {k: [hip.from_numpy(h[k]) for h in host] for k in kinds}
    mg.update_halo(*[dev[k] for k in kinds])
    mg.sync()
    for k in kinds:
        update_halo_z(host, k, nz)
        for r in range(P):
            assert np.array_equal(hip.to_numpy(dev[k][r]), host[r][k]), (k, r)
    # f32 goes through the same code with 4-byte elements
    h32 = [dict(c=fields(nx, ny, nz, ["c"], 7 + r, np.float32)[0]) for r in range(P)]
    d32 = [hip.from_numpy(h["c"]) for h in h32]
    mg.update_halo(d32)
    mg.sync()
    update_halo_z(h32, "c", nz)
    for r in range(P):
        assert np.array_equal(hip.to_numpy(d32[r]), h32[r]["c"])
    mg.close()


def test_max_g_and_gather(hip):
    from oracle.driver_ref import gather_z
    P, nx, ny, nz = 3, 11, 8, 6
    mg = _mg(P, nx, ny, nz)
    assert mg.max_g([1.5, -3.0, 2.25]) == 2.25
    assert mg.max_g([-7.0, -3.0, -4.0]) == -3.0
    assert math.isnan(mg.max_g([1.0, float("nan"), 5.0]))                     # Julia's maximum propagates NaN (App. B7)
    assert mg.max_g([1.0, float("inf"), 5.0]) == float("inf")
    for kind in ("c", "vx", "vz"):
        host = [dict(a=fields(nx, ny, nz, [kind], 31 * (r + 1))[0]) for r in range(P)]
        got = mg.gather([hip.from_numpy(h["a"]) for h in host])
        assert np.array_equal(got, gather_z(host, "a")) and got.flags.f_contiguous
    mg.close()


def _global_solve(hip, Pg, Dg, Rg, g, n_iters, bc, dtype):
    import torch
    ctx = hip.Context(0, "strict")
    dP, dD = hip.from_numpy(Pg), hip.from_numpy(Dg)
    p = hip.pt_params(dP, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, *bc)
    hip.pt_iterate(dP, dD, hip.from_numpy(Rg), p, n_iters, ctx=ctx)
    torch.cuda.synchronize()
    out = hip.to_numpy(dP), hip.to_numpy(dD)
    ctx.close()
    return out


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("P,depth,shape,n_iters", [(2, 2, (40, 21, 10), 7), (3, 2, (70, 12, 6), 5), (3, 1, (24, 15, 9), 4),
                                                   (2, 2, (200, 160, 66), 7), (4, 2, (33, 9, 4), 6), (2, 3, (40, 21, 10), 8),
                                                   (3, 3, (70, 12, 6), 11), (4, 3, (33, 9, 4), 7), (3, 3, (33, 9, 5), 10), (2, 4, (40, 21, 10), 9), (3, 4, (70, 12, 6), 11),
                                                   (3, 4, (33, 9, 5), 10),
                                                   (2, 3, (200, 160, 66), 7)])
def test_slab_state_equals_global_solve(hip, P, depth, shape, n_iters, dtype):
    """Decomposition independence of the C++ deep-ghost schedule (ns3d_slab_load / iterate / store): P virtual ranks leave
    exactly the planes of the single-device solve of the global grid — every local plane, halo planes included; odd
    iteration counts mix multi-iteration and single passes; the 4-plane slabs have seams that touch each other (and get
    one ghost plane fewer than asked for).  depth 3: two ghost planes per seam and three iterations per pass (forced through
    the ranks' contexts — the planner would only choose it on much larger grids)."""
    nx, ny, nz = shape
    nz_g = P * (nz - 2) + 2
    g = geometry(nx, ny, nz_g)
    Pg, Dg, Rg = fields(nx, ny, nz_g, ["c", "i", "c"], 211, dtype)
    bc = (True, 0.25, 0.0)
    Pref, Dref = _global_solve(hip, Pg, Dg, Rg, g, n_iters, bc, dtype)
    mg = _mg(P, nx, ny, nz)
    mg.set_temporal(depth)
    assert mg.pass_depth() == min(depth, 2)
    Pr = [hip.from_numpy(Pg[:, :, r * (nz - 2):r * (nz - 2) + nz]) for r in range(P)]
    D = [hip.from_numpy(Dg[:, :, r * (nz - 2):r * (nz - 2) + nz - 2]) for r in range(P)]
    R = [hip.from_numpy(Rg[:, :, r * (nz - 2):r * (nz - 2) + nz]) for r in range(P)]
    p = hip.pt_params(Pr[0], g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, *bc)
    if depth >= 3:
        for c in mg.contexts:
            c.set_pt_depth(depth)
    mg.slab_load(Pr, D, R, p)
    planned = mg.slab_plan()
    assert planned == min(depth, max(nz - 2, 1)) if depth != 2 else planned == 2
    mg.slab_iterate(n_iters)
    res = mg.slab_residual()
    mg.slab_store(Pr, D)
    mg.sync()
    for r in range(P):
        lo = r * (nz - 2)
        assert np.array_equal(hip.to_numpy(Pr[r]), Pref[:, :, lo:lo + nz]), "Pr of rank %d" % r
        assert np.array_equal(hip.to_numpy(D[r]), Dref[:, :, lo:lo + nz - 2]), "dPrdτ of rank %d" % r
    ctx = hip.Context(0, "strict")
    pg = hip.pt_params(hip.from_numpy(Pref), g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, *bc)
    assert res == hip.residual_max(hip.from_numpy(Pref), hip.from_numpy(Rg), pg, ctx=ctx)    # max_g of the slab residuals
    ctx.close()
    mg.close()


@pytest.mark.parametrize("P,shape,pass_depth,n_iters", [(2, (40, 21, 10), 2, 9), (3, (70, 12, 7), 2, 7), (3, (33, 9, 9), 3, 10),
                                                       (4, (24, 15, 6), 0, 8)])
def test_ghost_depth_follows_the_planned_pass_depth(hip, P, shape, pass_depth, n_iters):
    """the state is loaded with ghosts for the deepest pass allowed (set_temporal(4): three ghost planes per seam);
    when the ranks then settle for fewer iterations per pass (forced here through the contexts; 0 = the planner: between ranks
    the deepest pass the ghosts allow) ns3d_slab_plan drops the outer ghost planes — thinner seam sweeps, pass_depth + pass_depth − 1 planes
    per exchange — and the iterates stay those of the single-device solve of the global grid, bit for bit."""
    nx, ny, nz = shape
    nz_g = P * (nz - 2) + 2
    g = geometry(nx, ny, nz_g)
    Pg, Dg, Rg = fields(nx, ny, nz_g, ["c", "i", "c"], 977)
    bc = (True, 0.25, 0.0)
    Pref, Dref = _global_solve(hip, Pg, Dg, Rg, g, n_iters, bc, np.float64)
    mg = _mg(P, nx, ny, nz)
    mg.set_temporal(4)
    for c in mg.contexts:
        c.set_pt_depth(pass_depth)
    Pr = [hip.from_numpy(Pg[:, :, r * (nz - 2):r * (nz - 2) + nz]) for r in range(P)]
    D = [hip.from_numpy(Dg[:, :, r * (nz - 2):r * (nz - 2) + nz - 2]) for r in range(P)]
    R = [hip.from_numpy(Rg[:, :, r * (nz - 2):r * (nz - 2) + nz]) for r in range(P)]
    p = hip.pt_params(Pr[0], g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, *bc)
    mg.slab_load(Pr, D, R, p)
    assert mg.ghost_depth() == min(4, nz - 2) - 1
    planned = mg.slab_plan()
    assert planned == (pass_depth if pass_depth else min(4, nz - 2)) and mg.ghost_depth() == planned - 1
    mg.slab_iterate(n_iters)
    assert mg.slab_plan() == planned and mg.ghost_depth() == planned - 1          # planning again changes nothing
    mg.slab_iterate(1)
    mg.slab_store(Pr, D)
    mg.sync()
    Pref, Dref = _global_solve(hip, Pg, Dg, Rg, g, n_iters + 1, bc, np.float64)
    for r in range(P):
        lo = r * (nz - 2)
        assert np.array_equal(hip.to_numpy(Pr[r]), Pref[:, :, lo:lo + nz]), "Pr of rank %d" % r
        assert np.array_equal(hip.to_numpy(D[r]), Dref[:, :, lo:lo + nz - 2]), "dPrdτ of rank %d" % r
    mg.close()


@pytest.mark.parametrize("P", [2, 3])
def test_pt_solve_slab_equals_global_pt_solve(hip, oracle, P):
    """ns3d_pt_solve_slab = the whole inner loop multi.jl:458-471 over the ranks: same iteration count, same error history
    and same fields as ns3d_pt_solve on the global grid (and hence as the oracle's unfused loop)."""
    import torch
    nx, ny, nz = 34, 20, 9
    nz_g = P * (nz - 2) + 2
    g = geometry(nx, ny, nz_g)
    Pg, Dg, Rg = fields(nx, ny, nz_g, ["c", "i", "c"], 77)
    Pg *= 1e-3; Dg *= 1e-3; Rg *= 1e-6
    eps, niter, nchk, mul, div = 1.0e-4, 400, 19, 0.36, 1000.0
    ctx = hip.Context(0, "strict")
    dP, dD = hip.from_numpy(Pg), hip.from_numpy(Dg)
    pg = hip.pt_params(dP, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.0, 0.0)
    it_ref, errs_ref = hip.pt_solve(dP, dD, hip.from_numpy(Rg), pg, eps, niter, nchk, mul, div, ctx=ctx)
    torch.cuda.synchronize()
    Pref, Dref = hip.to_numpy(dP), hip.to_numpy(dD)
    ctx.close()
    assert nchk < it_ref < niter and len(errs_ref) == it_ref // nchk          # a real early exit
    mg = _mg(P, nx, ny, nz)
    Pr = [hip.from_numpy(Pg[:, :, r * (nz - 2):r * (nz - 2) + nz]) for r in range(P)]
    D = [hip.from_numpy(Dg[:, :, r * (nz - 2):r * (nz - 2) + nz - 2]) for r in range(P)]
    R = [hip.from_numpy(Rg[:, :, r * (nz - 2):r * (nz - 2) + nz]) for r in range(P)]
    p = hip.pt_params(Pr[0], g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.0, 0.0)
    it, errs = mg.pt_solve_slab(Pr, D, R, p, eps, niter, nchk, mul, div)
    mg.sync()
    assert it == it_ref and errs == errs_ref
    for r in range(P):
        lo = r * (nz - 2)
        assert np.array_equal(hip.to_numpy(Pr[r]), Pref[:, :, lo:lo + nz])
        assert np.array_equal(hip.to_numpy(D[r]), Dref[:, :, lo:lo + nz - 2])
    # NaN in one rank's right-hand side breaks the loop at the first check on every rank (multi.jl:469)
    Rbad = [hip.clone(t) for t in R]
    Rbad[P - 1][3, 4, 4] = float("nan")
    it, errs = mg.pt_solve_slab(Pr, D, Rbad, p, eps, niter, nchk, mul, div)
    assert it == nchk and len(errs) == 1 and math.isnan(errs[0])
    mg.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("dims", [(2, 1, 1), (1, 2, 1), (2, 2, 1), (2, 2, 2), (3, 1, 2)])
def test_pt_solve_on_a_cartesian_topology_equals_global_pt_solve(hip, dims, dtype):
    """ns3d_pt_solve_slab on a grid decomposed in x and y (one fused sweep + one update_halo! per iteration inside the
    library): same iteration count, same error history and same fields — halo entries included — as ns3d_pt_solve on the
    global grid; the outlet rule only on the ranks of the last x coordinate."""
    import torch
    n = (20, 14, 10)
    N = tuple(dims[d] * (n[d] - 2) + 2 for d in range(3))
    g = geometry(*N)
    Pg, Dg, Rg = fields(*N, ["c", "i", "c"], 91, dtype)
    Pg *= 1e-3; Dg *= 1e-3; Rg *= 1e-6
    eps, niter, nchk, mul, div = (1.0e-4 if dtype == np.float64 else 1.0e-3), 300, 17, 0.36, 1000.0
    ctx = hip.Context(0, "strict")
    dP, dD = hip.from_numpy(Pg), hip.from_numpy(Dg)
    pg = hip.pt_params(dP, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    it_ref, errs_ref = hip.pt_solve(dP, dD, hip.from_numpy(Rg), pg, eps, niter, nchk, mul, div, ctx=ctx)
    torch.cuda.synchronize()
    Pref, Dref = hip.to_numpy(dP), hip.to_numpy(dD)
    ctx.close()
    assert nchk < it_ref and len(errs_ref) == it_ref // nchk
    from navierstokes3d_amd.mgpu import MultiGpu
    from oracle.driver_ref import cart_coords
    P = dims[0] * dims[1] * dims[2]
    mg = MultiGpu.create([0] * P, *n, "strict", dims=dims, own_streams=OWN_STREAMS)

    def cut(A, r, shrink):
        c = cart_coords(r, dims)
        return np.asfortranarray(A[tuple(slice(c[d] * (n[d] - 2), c[d] * (n[d] - 2) + n[d] - shrink) for d in range(3))])

    Pr = [hip.from_numpy(cut(Pg, r, 0)) for r in range(P)]
    D = [hip.from_numpy(cut(Dg, r, 2)) for r in range(P)]
    R = [hip.from_numpy(cut(Rg, r, 0)) for r in range(P)]
    p = hip.pt_params(Pr[0], g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    it, errs = mg.pt_solve_slab(Pr, D, R, p, eps, niter, nchk, mul, div)
    mg.sync()
    assert it == it_ref and errs == errs_ref
    for r in range(P):
        assert np.array_equal(hip.to_numpy(Pr[r]), cut(Pref, r, 0)), r
        assert np.array_equal(hip.to_numpy(D[r]), cut(Dref, r, 2)), r
    mg.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("dims,n,depth", [((2, 1, 1), (20, 14, 10), 4), ((1, 2, 1), (20, 14, 10), 3), ((2, 2, 1), (70, 30, 9), 4),
                                          ((2, 2, 2), (20, 14, 10), 4), ((2, 2, 2), (20, 14, 10), 2), ((3, 1, 2), (12, 9, 7), 4),
                                          ((2, 3, 1), (9, 6, 8), 4), ((1, 2, 2), (66, 12, 5), 4),
                                          ((2, 2, 1), (130, 70, 34), 4), ((1, 2, 2), (200, 40, 30), 3)])   # several tiles per box
def test_deep_ghosts_on_a_cartesian_topology_equal_the_global_pt_solve(hip, dims, n, depth, dtype, monkeypatch):
    """solve_box (ns3d_mgpu.cpp): the solve state of every rank in a box extended by depth−1 ghost cells in x, y and z, passes of
    up to `depth` iterations on the whole box, ghost layers exchanged dimension by dimension (x/y layers packed by k_subbox_copy).
    Same iteration count, error history and fields as ns3d_pt_solve on the global grid, bit for bit — with the planner's pass
    depth and with every depth forced; a thin dimension (6 cells: four own layers) caps the ghost depth; NS3D_CART_DEEP=0 takes
    the one-sweep-per-iteration path and gives the same bits."""
    import torch
    N = tuple(dims[d] * (n[d] - 2) + 2 for d in range(3))
    g = geometry(*N)
    g["dtau"] = 0.8 / np.sqrt(1.0 / g["dx"] ** 2 + 1.0 / g["dy"] ** 2 + 1.0 / g["dz"] ** 2)     # inside the iteration's stability limit
    Pg, Dg, Rg = fields(*N, ["c", "i", "c"], 17, dtype)
    Pg *= 1e-3; Dg *= 1e-3; Rg *= 1e-6
    eps, niter, nchk, mul, div = -1.0, 47, 9, 0.36, 1000.0          # fixed iteration count: blocks of 9 = 4+4+1, 3+3+3, 2+2+2+2+1
    ctx = hip.Context(0, "strict")
    dP, dD = hip.from_numpy(Pg), hip.from_numpy(Dg)
    pg = hip.pt_params(dP, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    it_ref, errs_ref = hip.pt_solve(dP, dD, hip.from_numpy(Rg), pg, eps, niter, nchk, mul, div, ctx=ctx)
    torch.cuda.synchronize()
    Pref, Dref = hip.to_numpy(dP), hip.to_numpy(dD)
    ctx.close()
    assert it_ref == niter and len(errs_ref) == niter // nchk and np.all(np.isfinite(errs_ref)) and np.all(np.isfinite(Pref))
    from navierstokes3d_amd.mgpu import MultiGpu
    from oracle.driver_ref import cart_coords
    P = dims[0] * dims[1] * dims[2]

    def cut(A, r, shrink):
        c = cart_coords(r, dims)
        return np.asfortranarray(A[tuple(slice(c[d] * (n[d] - 2), c[d] * (n[d] - 2) + n[d] - shrink) for d in range(3))])

    # overlap: the shell-first order of a pass (round 4: shells and the exchange chain on the communication stream, the core sweep
    # meanwhile; the library takes it by itself from ≈40 M cells per rank on) forced on and off
    for deep, force, overlap in (("1", 0, "1"), ("1", 0, "0"), ("1", 2, "1"), ("1", 3, "1"), ("1", 4, "1"), ("1", 4, "0"), ("0", 0, "0")):
        if force > depth:
            continue
        monkeypatch.setenv("NS3D_CART_DEEP", deep)
        monkeypatch.setenv("NS3D_BOX_OVERLAP", overlap)
        mg = MultiGpu.create([0] * P, *n, "strict", dims=dims, own_streams=OWN_STREAMS)
        mg.set_temporal(depth)
        for c in mg.contexts:
            c.set_pt_depth(force)
        Pr = [hip.from_numpy(cut(Pg, r, 0)) for r in range(P)]
        D = [hip.from_numpy(cut(Dg, r, 2)) for r in range(P)]
        R = [hip.from_numpy(cut(Rg, r, 0)) for r in range(P)]
        p = hip.pt_params(Pr[0], g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
        it, errs = mg.pt_solve_slab(Pr, D, R, p, eps, niter, nchk, mul, div)
        mg.sync()
        assert it == it_ref and errs == errs_ref, (deep, force, overlap)
        for r in range(P):
            assert np.array_equal(hip.to_numpy(Pr[r]), cut(Pref, r, 0)), (deep, force, overlap, r)
            assert np.array_equal(hip.to_numpy(D[r]), cut(Dref, r, 2)), (deep, force, overlap, r)
        if deep == "1":
            cap = min([depth] + [n[d] - 2 for d in range(3) if dims[d] > 1])
            assert 1 <= mg.pass_depth() <= cap and (force == 0 or mg.pass_depth() == min(force, cap)), (mg.pass_depth(), force, cap)
        mg.close()


@pytest.mark.parametrize("P,fused,temporal,faithful,nt", [(2, True, True, True, 2), (3, True, True, True, 2), (2, True, False, True, 2),
                                                          (2, False, False, True, 2), (2, True, True, False, 3), (3, True, True, False, 3)])
def test_driver_on_mgpu_grid_vs_oracle_virtual_ranks(hip, P, fused, temporal, faithful, nt):
    """The product driver (multi.jl:287-536) on the C-ABI grid — update_halo!, max_g, gather!, and the inner loop as
    ns3d_pt_solve_slab (fused) or as the literal per-kernel sequence — against the oracle's P virtual ranks: iteration
    counts, every local field and the gathered return arrays, bit for bit.  faithful=False on several ranks: the fixed
    third branch advects Vz with backtrack!'s rank-local clamp and rounding, so the fused driver keeps update_τ! / update_halo!(τ) /
    predict_V! literal there (three steps: the advected Vz of step 2 feeds the stresses of step 3)."""
    from navierstokes3d_amd.driver import run_navierstokes3D
    from navierstokes3d_amd.mgpu import MgpuGrid, MultiGpu
    from navierstokes3d_amd.params import multi_params
    from oracle.driver_ref import run_navierstokes3D_ref
    nx = 32
    p0 = multi_params(nx)
    mg = MultiGpu.create([0] * P, p0.nx, p0.ny, p0.nz, "strict", own_streams=OWN_STREAMS)
    out = run_navierstokes3D(nx=nx, nt=nt, mode="strict", fused=fused, temporal=temporal, faithful=faithful,
                             grid=MgpuGrid(mg, p0.nx, p0.ny, p0.nz), return_info=True)
    info = out[-1]
    ref = run_navierstokes3D_ref(nx=nx, nt=nt, dims_z=P, faithful=faithful)
    assert info.iters == ref[-1].iters and info.errs == ref[-1].errs
    for r in range(P):
        for n in ("C", "Pr", "Vx", "Vy", "Vz", "divV", "dPrdtau"):
            assert np.array_equal(hip.to_numpy(getattr(info.local_fields[r], n)), ref[-1].ranks[r][n], equal_nan=True), (r, n)
    for n, a, b in zip(("C", "Pr", "Vx", "Vy", "Vz"), out[:5], ref[:5]):
        assert np.array_equal(a, b, equal_nan=True), n
    mg.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("P,nz,cfl", [(2, 9, 0.6), (2, 9, 1.9), (3, 7, 1.5), (4, 6, 1.2), (3, 12, 1.9)])
def test_advect_wide_is_the_global_advect(hip, oracle, P, nz, cfl, dtype):
    """ns3d_advect_wide (outside the reference's multi-rank semantics): {X_o .= X; advect!; update_halo!} on P z-slab ranks with a
    two-plane halo for the old fields equals advect! on the GLOBAL arrays — every local plane of every rank, halo planes and
    physical ends included, bit for bit, for departure points up to 1.9 cells away (cfl·|v| ≤ 1.9 with |v| ≤ 1), both advection
    modes; the *_o arrays hold the old fields.  (With the reference's one-plane halo backtrack! clamps to the local array:
    the plain multi-rank advect! differs from the global one — also checked, so that the test cannot pass vacuously.)"""
    nx, ny = 21, 10
    nz_g = P * (nz - 2) + 2
    g = geometry(nx, ny, nz_g)
    glob = fields(nx, ny, nz_g, ["vx", "vy", "vz", "c"], 321, dtype)
    dt = cfl * min(g["dx"], g["dy"], g["dz"])
    cutn = {0: nz, 1: nz, 2: nz + 1, 3: nz}
    for faithful in (True, False):
        ref = [a.copy(order="F") for a in glob]
        old = [a.copy(order="F") for a in glob]
        oracle.advect(ref[0], old[0], ref[1], old[1], ref[2], old[2], ref[3], old[3], dt, g["dx"], g["dy"], g["dz"], faithful)
        mg = _mg(P, nx, ny, nz)
        loc = [[hip.from_numpy(np.asfortranarray(glob[f][:, :, r * (nz - 2):r * (nz - 2) + cutn[f]])) for r in range(P)] for f in range(4)]
        loco = [[hip.zeros(tuple(t.shape), t.dtype) for t in loc[f]] for f in range(4)]
        mg.advect_wide(loc[0], loco[0], loc[1], loco[1], loc[2], loco[2], loc[3], loco[3], dt, g["dx"], g["dy"], g["dz"], faithful)
        mg.sync()
        for f in range(4):
            for r in range(P):
                lo = r * (nz - 2)
                assert np.array_equal(hip.to_numpy(loc[f][r]), ref[f][:, :, lo:lo + cutn[f]]), (f, r, faithful)
                assert np.array_equal(hip.to_numpy(loco[f][r]), glob[f][:, :, lo:lo + cutn[f]]), ("old", f, r)
        mg.close()
    # the reference's own multi-rank semantics on the same data: local clamps change the cells next to a seam once |δz| > 1
    if cfl > 1.0:
        differs = False
        for r in range(P):
            lo = r * (nz - 2)
            new = [hip.from_numpy(np.asfortranarray(glob[f][:, :, lo:lo + cutn[f]])) for f in range(4)]
            oldl = [hip.clone(t) for t in new]
            hip.advect(new[0], oldl[0], new[1], oldl[1], new[2], oldl[2], new[3], oldl[3], dt, g["dx"], g["dy"], g["dz"], False)
            own = slice(1 if r > 0 else 0, nz - 1 if r < P - 1 else nz)
            differs |= not np.array_equal(hip.to_numpy(new[3])[:, :, own], ref[3][:, :, lo:lo + nz][:, :, own])
        assert differs


@pytest.mark.parametrize("P,nz_loc", [(2, 12), (4, 7)])
def test_driver_with_wide_advect_halo_reproduces_the_one_rank_run(hip, P, nz_loc):
    """The whole time step becomes decomposition-independent: P z-slab ranks with wide_advect_halo=True return the one-rank run's
    gathered fields and iteration counts bit for bit (same global 36×22×22 grid, a size at which the reference's iteration stays
    finite; every other kernel already is decomposition-independent: local stencils
    behind update_halo!, Jacobi sweeps).  Without the option the P-rank run is the reference's P-rank run, which is a different
    result — the oracle's virtual ranks cover that."""
    from navierstokes3d_amd.driver import run_navierstokes3D
    from navierstokes3d_amd.mgpu import MgpuGrid, MultiGpu
    from navierstokes3d_amd.params import multi_params
    nx, nt = 36, 3
    one = run_navierstokes3D(nx=nx, nt=nt, mode="strict", return_info=True)
    assert one[-1].params.nz == P * (nz_loc - 2) + 2 and all(np.isfinite(a).all() for a in one[:5])
    p0 = multi_params(nx, dims=(1, 1, P), coords=(0, 0, 0), nz=nz_loc)
    mg = MultiGpu.create([0] * P, p0.nx, p0.ny, p0.nz, "strict", own_streams=OWN_STREAMS)
    out = run_navierstokes3D(nx=nx, nt=nt, mode="strict", grid=MgpuGrid(mg, p0.nx, p0.ny, p0.nz), return_info=True, shape=dict(nz=nz_loc),
                             wide_advect_halo=True)
    assert out[-1].iters == one[-1].iters and out[-1].errs == one[-1].errs
    for n, a, b in zip(("C", "Pr", "Vx", "Vy"), out[:4], one[:4]):          # (the gathered Vz has another shape for P > 1: DESIGN §6)
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), n
    # every local plane of every rank, halo planes included, against the one-rank fields
    for r in range(P):
        lo = r * (nz_loc - 2)
        for n, extra in (("C", 0), ("Pr", 0), ("Vx", 0), ("Vy", 0), ("Vz", 1), ("divV", 0)):
            a = hip.to_numpy(getattr(out[-1].local_fields[r], n))
            b = hip.to_numpy(getattr(one[-1].fields, n))[:, :, lo:lo + nz_loc + extra]
            assert np.array_equal(a, b, equal_nan=True), (r, n)
    mg.close()


def test_dims_create_is_mpi_dims_create(hip):
    """init_global_grid's default topology (multi.jl:325 passes no dimx/dimy/dimz): MPI_Dims_create — balanced, non-increasing,
    fixed entries kept."""
    from navierstokes3d_amd import lib as L
    from navierstokes3d_amd.mgpu import MultiGpu
    want = {1: (1, 1, 1), 2: (2, 1, 1), 3: (3, 1, 1), 4: (2, 2, 1), 6: (3, 2, 1), 8: (2, 2, 2), 12: (3, 2, 2), 16: (4, 2, 2),
            18: (3, 3, 2), 24: (4, 3, 2), 36: (4, 3, 3), 64: (4, 4, 4), 7: (7, 1, 1)}
    for P, d in want.items():
        assert MultiGpu.dims_create(P) == d, P
    assert MultiGpu.dims_create(8, (1, 1, 0)) == (1, 1, 8)            # the z-slab choice of this build
    assert MultiGpu.dims_create(8, (0, 1, 0)) == (4, 1, 2) and MultiGpu.dims_create(12, (0, 3, 0)) == (2, 3, 2)
    with pytest.raises(L.Ns3dError):
        MultiGpu.dims_create(8, (3, 0, 0))


@pytest.mark.parametrize("dims", [(2, 1, 1), (1, 2, 1), (2, 2, 1), (2, 2, 2), (3, 2, 1), (1, 3, 2)])
def test_update_halo_and_gather_on_a_cartesian_topology(hip, dims):
    """update_halo! / gather! with x and y decomposition (packed strided faces) against the oracle's virtual ranks
    (oracle/driver_ref.py::update_halo_3d, pinned in tests/test_oracle.py): every stagger, arrays without a halo, several
    fields per call, f64 and f32; edges and corners arrive through the x→y→z order."""
    from navierstokes3d_amd.mgpu import MgpuGrid, MultiGpu
    from oracle.driver_ref import cart_coords, gather_3d, update_halo_3d
    P = dims[0] * dims[1] * dims[2]
    nx, ny, nz = 13, 9, 7
    kinds = ["c", "vx", "vy", "vz", "s", "i"]
    mg = MultiGpu.create([0] * P, nx, ny, nz, "strict", dims=dims, own_streams=OWN_STREAMS)
    assert mg.dims == dims and mg.P == P and mg.coords == [cart_coords(r, dims) for r in range(P)]
    assert mg.n_g() == tuple(dims[d] * ((nx, ny, nz)[d] - 2) + 2 for d in range(3))
    grid = MgpuGrid(mg, nx, ny, nz)
    assert (grid.nx_g(), grid.ny_g(), grid.nz_g()) == mg.n_g() and grid.z_slabs() == (dims[0] == 1 and dims[1] == 1)
    for dtype in (np.float64, np.float32):
        host = [dict(zip(kinds, fields(nx, ny, nz, kinds, 100 * (r + 1), dtype))) for r in range(P)]
        dev = {k: [hip.from_numpy(h[k]) for h in host] for k in kinds}
        mg.update_halo(*[dev[k] for k in kinds])              # one call, six fields
        mg.update_halo(dev["vx"])                             # and again alone: idempotent, buffer reuse
        mg.sync()
        for k in kinds:
            update_halo_3d(host, k, (nx, ny, nz), dims)
            for r in range(P):
                assert np.array_equal(hip.to_numpy(dev[k][r]), host[r][k]), (k, r, dtype)
        for k in ("c", "vx", "vy", "vz"):
            got = mg.gather(dev[k])
            assert got.flags.f_contiguous and np.array_equal(got, gather_3d(host, k, dims)), k
    mg.close()


@pytest.mark.parametrize("dims,nx,shape", [((2, 1, 1), 14, dict(ny=16, nz=16)), ((2, 2, 1), 14, dict(ny=9, nz=16)),
                                           ((2, 2, 2), 14, dict(ny=9, nz=9)), ((3, 2, 1), 10, dict(ny=9, nz=16))])
@pytest.mark.parametrize("fused", [False, True])
def test_driver_on_a_cartesian_topology_vs_oracle_virtual_ranks(hip, dims, nx, shape, fused):
    """The driver on the topologies ImplicitGlobalGrid picks by default for 2, 4 and 8 ranks (multi.jl:325 →
    MPI_Dims_create), inner loop kernel by kernel as written (multi.jl:458-471) or inside the library (ns3d_pt_solve_slab on a
    Cartesian grid: one fused sweep and one halo update per iteration): iteration counts, residual histories, every
    rank's fields and the gathered arrays against the oracle's virtual ranks, bit for bit.  Only the ranks on the inlet /
    outlet planes apply those conditions (multi.jl:164,179).  The local sizes are chosen so that the GLOBAL grid is the same
    near-isotropic 26×16×16 for every topology: dτ follows max(dx,dy,dz) (multi.jl:341), so the grids ceil(0.6 nx) gives
    under an x-only decomposition (dx ≈ dy/2) make the reference's own iteration diverge."""
    from navierstokes3d_amd import lib as L
    from navierstokes3d_amd.driver import run_navierstokes3D
    from navierstokes3d_amd.mgpu import MgpuGrid, MultiGpu
    from navierstokes3d_amd.params import multi_params
    from oracle.driver_ref import run_navierstokes3D_ref
    P = dims[0] * dims[1] * dims[2]
    assert MultiGpu.dims_create(P) == dims
    nt, cap = 2, 300
    p0 = multi_params(nx, dims=dims, **shape)
    assert (p0.nx_g, p0.ny_g, p0.nz_g) == (26, 16, 16)
    mg = MultiGpu.create([0] * P, p0.nx, p0.ny, p0.nz, "strict", dims=dims, own_streams=OWN_STREAMS)
    grid = MgpuGrid(mg, p0.nx, p0.ny, p0.nz)
    out = run_navierstokes3D(nx=nx, nt=nt, mode="strict", fused=fused, grid=grid, niter_cap=cap, return_info=True, shape=shape)
    info = out[-1]
    ref = run_navierstokes3D_ref(nx=nx, nt=nt, dims=dims, niter_cap=cap, shape=shape)
    assert info.iters == ref[-1].iters and info.errs == ref[-1].errs and info.iters[-1] < cap
    for r in range(P):
        for n in ("C", "Pr", "Vx", "Vy", "Vz", "divV", "dPrdtau"):
            assert np.array_equal(hip.to_numpy(getattr(info.local_fields[r], n)), ref[-1].ranks[r][n], equal_nan=True), (r, n)
    for n, a, b in zip(("C", "Pr", "Vx", "Vy", "Vz"), out[:5], ref[:5]):
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), n
    assert np.isfinite(out[1]).all() and 0 < np.abs(out[2]).max() < 1.0
    mg.close()


def test_rccl_communicator_of_one_rank(hip):
    """The one-process-per-GPU form as far as one GPU goes: RCCL is found by dlopen, a unique id is made, a communicator
    of one rank comes up, and the residual check / max_g run their ncclAllReduce on it; results equal the single-device
    calls.  (Send/recv between ranks needs a second GPU: bench.py --gpus N probes and reports that on the multi-GPU node.)"""
    import torch
    from navierstokes3d_amd.mgpu import MultiGpu
    nx, ny, nz = 40, 21, 12
    uid = MultiGpu.unique_id()
    assert len(uid) == 128 and any(uid)
    mg = MultiGpu.create_rank(1, 0, 0, uid, nx, ny, nz, "strict")
    assert mg.transport == "rccl" and mg.rccl_ranks() == 1 and mg.nlocal == 1 and mg.ranks == [0]
    assert mg.max_g([-2.5]) == -2.5 and math.isnan(mg.max_g([float("nan")]))
    g = geometry(nx, ny, nz)
    Pg, Dg, Rg = fields(nx, ny, nz, ["c", "i", "c"], 5)
    Pref, Dref = _global_solve(hip, Pg, Dg, Rg, g, 6, (True, 0.25, 0.0), np.float64)
    Pr, D, R = hip.from_numpy(Pg), hip.from_numpy(Dg), hip.from_numpy(Rg)
    p = hip.pt_params(Pr, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    mg.update_halo(Pr)                       # one rank: nothing to exchange, must not touch the field
    mg.slab_load(Pr, D, R, p)
    mg.slab_iterate(6)
    res = mg.slab_residual()
    mg.slab_store(Pr, D)
    mg.sync()
    assert np.array_equal(hip.to_numpy(Pr), Pref) and np.array_equal(hip.to_numpy(D), Dref)
    ctx = hip.Context(0, "strict")
    assert res == hip.residual_max(Pr, R, p, ctx=ctx)
    got = mg.gather(Pr)
    assert np.array_equal(got, Pref[1:-1, 1:-1, 1:-1])
    ctx.close()
    mg.close()
    torch.cuda.synchronize()


def test_mgpu_argument_errors(hip):
    from navierstokes3d_amd import lib as L
    from navierstokes3d_amd.mgpu import MultiGpu
    with pytest.raises(L.Ns3dError):
        MultiGpu.create([0, 0], 8, 8, 2)                      # grid too small
    with pytest.raises(L.Ns3dError):
        MultiGpu.create([0, 99], 8, 8, 8)                     # no such device
    mg = _mg(2, 12, 8, 6)
    with pytest.raises(L.Ns3dError):
        mg.slab_iterate(2)                                    # nothing loaded
    with pytest.raises(L.Ns3dError):
        mg.set_temporal(5)
    with pytest.raises(L.Ns3dError):
        mg.update_halo([hip.zeros((12, 8, 6))])               # one tensor for two local ranks
    mg.close()
    with pytest.raises(L.Ns3dError):
        MultiGpu.create([0, 0], 8, 8, 8, dims=(2, 2, 1))      # dims hold four ranks
    mg = MultiGpu.create([0, 0], 12, 8, 6, dims=(2, 1, 1))
    z = [hip.zeros((12, 8, 6)), hip.zeros((12, 8, 6))]
    d = [hip.zeros((10, 6, 4)), hip.zeros((10, 6, 4))]
    p = hip.pt_params(z[0], 1000.0, 0.01, 0.01, 0.1, 0.1, 0.1, 0.1, 0, True, 0.0, 0.0)
    with pytest.raises(L.Ns3dError, match="z-slab"):
        mg.slab_load(z, d, z, p)                              # the deep-ghost state: z-slabs only
    mg.close()
