"""Two z-slab ranks sharing the one GPU of the test box (gloo backend, host-staged halo transport): the product
driver's N>1 path end to end on HIP kernels — literal reference sequence and fused/overlapped PT schedule — against
the oracle's two virtual ranks.  The RCCL ("device") transport differs only in how the same planes travel."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, nx, nt, fused, temporal, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from navierstokes3d_amd import kernels as K
        from navierstokes3d_amd.driver import run_navierstokes3D
        from navierstokes3d_amd.halo import ZSlabGrid
        from navierstokes3d_amd.params import multi_params
        p0 = multi_params(nx)
        grid = ZSlabGrid(p0.nx, p0.ny, p0.nz)
        out = run_navierstokes3D(nx=nx, nt=nt, mode="strict", fused=fused, temporal=temporal, grid=grid, device=0,
                                 return_info=True)
        info = out[-1]
        local = {n: K.to_numpy(getattr(info.fields, n)) for n in ("C", "Pr", "Vx", "Vy", "Vz", "divV", "dPrdtau")}
        q.put((rank, info.iters, local, out[:5] if rank == 0 else None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "ERROR", traceback.format_exc(), None))


@pytest.mark.parametrize("world,fused,temporal", [(2, True, True), (3, True, True), (2, True, False), (2, False, False)])
def test_zslab_ranks_one_gpu(hip, world, fused, temporal):
    """temporal=True: two iterations per pass with two-plane-deep ghosts (slab.py); the 3-rank case has a middle rank
    with seams on both sides.  temporal=False: single sweeps with the reference's one-plane halo."""
    from oracle.driver_ref import run_navierstokes3D_ref
    nx, nt = 32, 2          # stays finite (smaller grids run into the reference's known instability)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nx, nt, fused, temporal, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = {}
    for _ in range(world):
        r = q.get(timeout=600)
        assert r[1] != "ERROR", r[2]
        results[r[0]] = r
    for pr in procs:
        pr.join(timeout=60)
    ref = run_navierstokes3D_ref(nx=nx, nt=nt, dims_z=world)
    assert all(np.isfinite(a).all() for a in ref[:5]) and ref[-1].iters[-1] > ref[-1].params.nchk   # a meaningful case
    for r in range(world):
        _, iters, local, _ = results[r]
        assert iters == ref[-1].iters
        for n, a in local.items():
            assert np.array_equal(a, ref[-1].ranks[r][n], equal_nan=True), (r, n)
    for n, a, b in zip(("C", "Pr", "Vx", "Vy", "Vz"), results[0][3], ref[:5]):
        assert np.array_equal(a, b, equal_nan=True), n


def _slab_worker(rank, world, port, shape, n_iters, q):
    try:
        import faulthandler
        faulthandler.dump_traceback_later(120, exit=True)       # a hung rank reports where it hangs instead of stalling the suite
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from navierstokes3d_amd import kernels as K, lib as L
        from navierstokes3d_amd.halo import ZSlabGrid
        from navierstokes3d_amd.slab import SlabPTSolver
        from util import fields, geometry
        nx, ny, nz = shape
        nz_g = world * (nz - 2) + 2
        g = geometry(nx, ny, nz_g)
        Pg, Dg, Rg = fields(nx, ny, nz_g, ["c", "i", "c"], 211)          # the same global fields on every rank
        lo = rank * (nz - 2)
        Pr, R = K.from_numpy(Pg[:, :, lo:lo + nz]), K.from_numpy(Rg[:, :, lo:lo + nz])
        D = K.from_numpy(Dg[:, :, lo:lo + nz - 2])
        grid = ZSlabGrid(nx, ny, nz)
        ctx = K.Context(0, "strict")
        sol = SlabPTSolver(ctx, grid, Pr, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"],
                           L.NS3D_BC_MULTI, True, 0.25, 0.0)
        sol.load(Pr, D, R)
        sol.iterate(n_iters)
        sol.store(Pr, D)
        torch.cuda.synchronize()
        q.put((rank, "OK", K.to_numpy(Pr), K.to_numpy(D)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "ERROR", traceback.format_exc(), None))


@pytest.mark.parametrize("world", [2, 3])
def test_slab_solver_equals_global_solve_large(hip, world):
    """Decomposition independence at a size where the tuned two-iteration kernels run (≈2 M cells per interior launch):
    seven PT iterations on `world` z-slab ranks (deep-ghost schedule, seam planes first) leave exactly the planes of the
    single-device solve of the global grid — every local plane, halo planes included."""
    import torch
    from util import fields, geometry
    shape, n_iters = (200, 160, 66), 7
    nx, ny, nz = shape
    nz_g = world * (nz - 2) + 2
    g = geometry(nx, ny, nz_g)
    Pg, Dg, Rg = fields(nx, ny, nz_g, ["c", "i", "c"], 211)
    ctx = hip.Context(0, "strict")
    dP, dD = hip.from_numpy(Pg), hip.from_numpy(Dg)
    p = hip.pt_params(dP, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    hip.pt_iterate(dP, dD, hip.from_numpy(Rg), p, n_iters, ctx=ctx)
    torch.cuda.synchronize()
    Pref, Dref = hip.to_numpy(dP), hip.to_numpy(dD)
    ctx.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_slab_worker, args=(r, world, port, shape, n_iters, q), daemon=True) for r in range(world)]
    for pr in procs:
        pr.start()
    got = {}
    for _ in range(world):
        r = q.get(timeout=240)
        assert r[1] == "OK", r[2]
        got[r[0]] = r
    for pr in procs:
        pr.join(timeout=60)
    for r in range(world):
        lo = r * (nz - 2)
        assert np.array_equal(got[r][2], Pref[:, :, lo:lo + nz]), "Pr of rank %d" % r
        assert np.array_equal(got[r][3], Dref[:, :, lo:lo + nz - 2]), "dPrdτ of rank %d" % r
