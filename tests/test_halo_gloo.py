"""N>1 path on CPU: world_size 2 and 3 over the gloo backend (127.0.0.1).

What is under test is the PRODUCT's z-slab implicit global grid (navierstokes3d_amd/halo.py: update_halo!, max_g,
gather!, neighbour topology, host-staged transport).  The stencil arithmetic on each rank is done here by the CPU
oracle (test infrastructure) because the HIP kernels need a GPU; every rank runs the literal per-rank sequence of
multi.jl:446-477 and the results must equal, bit for bit, the oracle's P *virtual* ranks held in one process
(oracle/driver_ref.py), including the gathered global arrays on rank 0.
"""
import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _as_torch(a):
    """Column-major (nx,ny,nz) torch view sharing memory with the Fortran-ordered numpy array."""
    return torch.from_numpy(a.T).permute(2, 1, 0)


def _worker(rank, world, port, nx, nt, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("OMP_NUM_THREADS", "2")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from navierstokes3d_amd.halo import init_global_grid
        from navierstokes3d_amd.params import multi_params
        from oracle import oracle as K
        from oracle.driver_ref import _alloc_multi, Obj

        p0 = multi_params(nx)
        me, dims, grid = init_global_grid(p0.nx, p0.ny, p0.nz)                       # multi.jl:325
        assert me == rank and dims == (1, 1, world) and grid.transport == "host"
        p = multi_params(nx, world, me)
        assert grid.nz_g() == p.nz_g
        f = _alloc_multi(Obj(nx=p.nx, ny=p.ny, nz=p.nz, dtype=np.float64))
        T = {n: _as_torch(f[n]) for n in ("Pr", "C", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "txy", "divV", "dPrdtau")}
        halo = lambda *names: grid.update_halo(*[T[n] for n in names])
        f.Vy[0, :, :] = p.vin                                                        # :369
        halo("Pr")                                                                   # :371
        cyl = (p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.xco_g, p.yco_g, p.zco_g, p.lx, p.ly, p.lz, p.dx, p.dy, p.dz)
        K.set_cylinder(f.C, f.Vx, f.Vy, f.Vz, *cyl)                                  # :372
        halo("C", "Vx", "Vy", "Vz")                                                  # :373
        halo("txy", "dPrdtau")                                                       # overlap < 2: must be a no-op
        iters = []
        for it in range(nt):
            K.update_tau(f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, f.Vx, f.Vy, f.Vz, p.mu, p.dx, p.dy, p.dz)
            halo("txx", "tyy", "tzz")                                                # :450
            K.predict_V(f.Vx, f.Vy, f.Vz, f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, p.rho, p.g, p.dt, p.dx, p.dy, p.dz)
            K.set_cylinder(f.C, f.Vx, f.Vy, f.Vz, *cyl)
            halo("C", "Vx", "Vy", "Vz")                                              # :453
            K.update_divV(f.divV, f.Vx, f.Vy, f.Vz, p.dx, p.dy, p.dz)
            halo("divV")                                                             # :455
            done = p.niter
            for itr in range(1, p.niter + 1):                                        # :458
                K.update_dPrdtau(f.Pr, f.dPrdtau, f.divV, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz)
                K.update_Pr(f.Pr, f.dPrdtau, p.dtau)
                K.set_bc_Pr(f.Pr, 0, p.owns_outlet, 0.0)                             # :176-181
                halo("Pr")                                                           # :182 (the one exchange needed)
                if itr % p.nchk == 0:
                    K.compute_res(f.Rp, f.Pr, f.divV, p.rho, p.dt, p.dx, p.dy, p.dz)
                    err = grid.max_g(K.max_abs(f.Rp)) * (p.ly * p.ly) / p.psc        # :466, :21
                    if err < p.eps or not np.isfinite(err):
                        done = itr
                        break
            iters.append(done)
            K.correct_V(f.Vx, f.Vy, f.Vz, f.Pr, p.dt, p.rho, p.dx, p.dy, p.dz)
            K.set_cylinder(f.C, f.Vx, f.Vy, f.Vz, *cyl)
            K.set_bc_Vel(f.Vx, f.Vy, f.Vz, 0, p.owns_inlet, p.vin)
            halo("Vx", "Vy", "Vz")                                                   # :167
            K.copy(f.Vx_o, f.Vx); K.copy(f.Vy_o, f.Vy); K.copy(f.Vz_o, f.Vz); K.copy(f.C_o, f.C)
            K.advect(f.Vx, f.Vx_o, f.Vy, f.Vy_o, f.Vz, f.Vz_o, f.C, f.C_o, p.dt, p.dx, p.dy, p.dz, True)
            halo("Vx", "Vy", "Vz")                                                   # :477
        nanmax = grid.max_g(float("nan") if rank == world - 1 else 1.0)             # NaN must win the all-reduce
        gathered = {n: grid.gather(f[n][1:-1, 1:-1, 1:-1]) for n in ("C", "Pr", "Vx", "Vy", "Vz")}   # :528-532
        local = {n: np.array(f[n]) for n in ("C", "Pr", "Vx", "Vy", "Vz", "divV", "dPrdtau")}
        q.put((rank, iters, local, gathered if rank == 0 else None, nanmax))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "ERROR", traceback.format_exc(), None, None))


@pytest.mark.parametrize("world", [2, 3])
def test_zslab_ranks_match_virtual_rank_oracle(world):
    from oracle.driver_ref import run_navierstokes3D_ref
    nx, nt = 32, 2          # stays finite (smaller grids run into the reference's known instability)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nx, nt, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = {}
    for _ in range(world):
        r = q.get(timeout=300)
        assert r[1] != "ERROR", r[2]
        results[r[0]] = r
    for pr in procs:
        pr.join(timeout=60)
    ref = run_navierstokes3D_ref(nx=nx, nt=nt, dims_z=world)
    assert all(np.isfinite(a).all() for a in ref[:5]) and ref[-1].iters[-1] > ref[-1].params.nchk   # a meaningful case
    info = ref[-1]
    for r in range(world):
        _, iters, local, gathered, nanmax = results[r]
        assert iters == info.iters
        assert np.isnan(nanmax)
        for n, a in local.items():
            assert np.array_equal(a, info.ranks[r][n], equal_nan=True), (r, n)
    gathered = results[0][3]
    for n, a in zip(("C", "Pr", "Vx", "Vy", "Vz"), ref[:5]):
        assert np.array_equal(gathered[n], a, equal_nan=True), n
        # rank blocks are concatenated along z with their local inner extents (Vz: nz-1 per rank)
        assert gathered[n].shape[2] == world * (info.params.nz - (1 if n == "Vz" else 2))


def test_halo_plane_indices_follow_implicit_global_grid():
    """ol = 2 + (size − nz): cell-centred arrays send planes 2 / n−1 (1-based), Vz (nz+1) sends 3 / n−2, arrays of
    extent nz−1 or nz−2 have no halo (SURVEY.md §2.4 [upstream])."""
    from navierstokes3d_amd.halo import ZSlabGrid
    g = ZSlabGrid(8, 6, 10)
    z = lambda n: torch.zeros(n, 6, 8, dtype=torch.float64).permute(2, 1, 0)
    assert g.halo_planes(z(10)) == (1, 8, 0, 9)
    assert g.halo_planes(z(11)) == (2, 8, 0, 10)
    assert g.halo_planes(z(9)) is None and g.halo_planes(z(8)) is None
    assert g.P == 1 and g.nz_g() == 10 and not g.z_lo_is_halo() and not g.z_hi_is_halo()
    A = z(10); A.permute(2, 1, 0)[3].fill_(7.0)
    assert g.plane(A, 3).is_contiguous() and float(A[5, 2, 3]) == 7.0
    g.update_halo(A)                      # single rank: no-op
    assert g.max_g(3.5) == 3.5 and np.isnan(g.max_g(float("nan")))
    assert np.array_equal(g.gather(np.ones((2, 2, 2))), np.ones((2, 2, 2)))
