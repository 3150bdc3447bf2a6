"""The ONE-PROCESS-PER-GPU arm of the multi-GPU layer (ns3d_mgpu_create_rank[_cart]: exchange_begin's ncclSend/ncclRecv group,
gather_impl's send/recv, the ncclAllReduce of the residual, of max_g and of ns3d_slab_plan's depth agreement) executed on the
one GPU of the test box: RCCL refuses two ranks on one device, so the ranks — separate processes, as under
torch.distributed.run — bind libns3d's dlopen to tests/fake_rccl (NS3D_RCCL_LIB, the library's own override), a test double that
moves the bytes between the processes through registered shared memory with real stream ordering.  Pairing, grouping, byte
counts, in-order matching, event ordering and the collectives are what the 8-GPU node will run; only xGMI is missing.
Everything is compared bit for bit with the oracle's virtual ranks or with the single-device solve of the global grid."""
import os
import socket
import subprocess
import sys
import traceback
import json

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_SRC = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")
FAKE_SO = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def build_fake():
    if not os.path.exists(FAKE_SO) or os.path.getmtime(FAKE_SO) < os.path.getmtime(FAKE_SRC):
        subprocess.check_call(["hipcc", "-O2", "-shared", "-fPIC", "-std=c++17", FAKE_SRC, "-o", FAKE_SO, "-lrt", "-lpthread"])
    return FAKE_SO


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, scenario, args, q):
    try:
        import faulthandler
        faulthandler.dump_traceback_later(150, exit=True)
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        os.environ["NS3D_RCCL_LIB"] = FAKE_SO
        os.environ.setdefault("FAKE_RCCL_ARENA_MB", "8")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from navierstokes3d_amd.mgpu import MultiGpu
        box = [MultiGpu.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        assert box[0].startswith(b"/fake_rccl_")                       # the double made this id, not the real library
        out = globals()["_sc_" + scenario](rank, world, box[0], *args)
        q.put((rank, "OK", out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "ERROR", traceback.format_exc()))


def _run(world, scenario, *args, timeout=240):
    build_fake()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, scenario, args, q), daemon=True) for r in range(world)]
    for pr in procs:
        pr.start()
    got = {}
    try:
        for _ in range(world):
            r = q.get(timeout=timeout)
            assert r[1] == "OK", r[2]
            got[r[0]] = r[2]
    finally:
        for pr in procs:
            pr.join(timeout=30)
            if pr.is_alive():
                pr.kill()
    return got


# ---- scenarios (run inside the rank processes) --------------------------------------------------------------------------
def _sc_halo(rank, world, uid, dims, n):
    """update_halo! for every stagger, max_g, gather! through RCCL-shaped calls"""
    import torch
    from navierstokes3d_amd import kernels as K
    from navierstokes3d_amd.mgpu import MultiGpu
    from util import fields
    nx, ny, nz = n
    mg = MultiGpu.create_rank(world, rank, 0, uid, nx, ny, nz, "strict", dims=dims)
    assert mg.transport == "rccl" and mg.rccl_ranks() == world and mg.nlocal == 1 and mg.ranks == [rank]
    kinds = ["c", "vx", "vy", "vz", "s", "i"]
    host = dict(zip(kinds, fields(nx, ny, nz, kinds, 100 * (rank + 1))))
    dev = {k: K.from_numpy(host[k]) for k in kinds}
    mg.update_halo(*[dev[k] for k in kinds])
    mg.sync()
    out = {k: K.to_numpy(dev[k]) for k in kinds}
    f32 = fields(nx, ny, nz, ["c"], 7 + rank, np.float32)[0]
    d32 = K.from_numpy(f32)
    mg.update_halo(d32)
    mg.sync()
    out["c32"] = K.to_numpy(d32)
    out["max_g"] = mg.max_g([float(rank) - 0.5])
    out["max_g_nan"] = mg.max_g([float("nan") if rank == world - 1 else 1.0])
    for kind in ("c", "vx"):
        g = mg.gather(K.from_numpy(fields(nx, ny, nz, [kind], 31 * (rank + 1))[0]))
        assert (g is not None) == (rank == 0)
        out["gather_" + kind] = g
    mg.close()
    return out


def _sc_slab(rank, world, uid, shape, depth, n_iters, dtype, force_depth):
    """ns3d_slab_load / _plan / _iterate / _residual / _store on one z-slab rank per process"""
    import torch
    from navierstokes3d_amd import kernels as K
    from navierstokes3d_amd.mgpu import MultiGpu
    from util import fields, geometry
    nx, ny, nz = shape
    nz_g = world * (nz - 2) + 2
    g = geometry(nx, ny, nz_g)
    Pg, Dg, Rg = fields(nx, ny, nz_g, ["c", "i", "c"], 211, dtype)
    lo = rank * (nz - 2)
    Pr, R, D = K.from_numpy(Pg[:, :, lo:lo + nz]), K.from_numpy(Rg[:, :, lo:lo + nz]), K.from_numpy(Dg[:, :, lo:lo + nz - 2])
    mg = MultiGpu.create_rank(world, rank, 0, uid, nx, ny, nz, "strict")
    mg.set_temporal(depth)
    if force_depth:
        mg.contexts[0].set_pt_depth(force_depth)
    p = K.pt_params(Pr, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    mg.slab_load(Pr, D, R, p)
    planned = mg.slab_plan()                       # ncclAllReduce: the ranks agree on the minimum
    mg.slab_iterate(n_iters)
    res = mg.slab_residual()                       # ncclAllReduce(max) of the residual keys
    mg.slab_store(Pr, D)
    mg.sync()
    out = dict(Pr=K.to_numpy(Pr), D=K.to_numpy(D), res=res, planned=planned, ghost=mg.ghost_depth())
    mg.close()
    return out


def _sc_solve(rank, world, uid, dims, n, dtype, eps, niter, nchk):
    """ns3d_pt_solve_slab: the whole inner loop with its global residual check, z-slabs (deep ghosts) or any Cartesian topology"""
    from navierstokes3d_amd import kernels as K
    from navierstokes3d_amd.mgpu import MultiGpu
    from oracle.driver_ref import cart_coords
    from util import fields, geometry
    ng = tuple(dims[d] * (n[d] - 2) + 2 for d in range(3))
    g = geometry(*ng)
    Pg, Dg, Rg = fields(ng[0], ng[1], ng[2], ["c", "i", "c"], 433, dtype)
    Rg *= 1e-3
    c = cart_coords(rank, dims)
    cut = lambda A, shrink: np.asfortranarray(A[tuple(slice(c[d] * (n[d] - 2), c[d] * (n[d] - 2) + n[d] - shrink) for d in range(3))])
    Pr, D, R = K.from_numpy(cut(Pg, 0)), K.from_numpy(cut(Dg, 2)), K.from_numpy(cut(Rg, 0))
    mg = MultiGpu.create_rank(world, rank, 0, uid, n[0], n[1], n[2], "strict", dims=dims)
    p = K.pt_params(Pr, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    it, errs = mg.pt_solve_slab(Pr, D, R, p, eps, niter, nchk, 0.36, 1000.0)
    mg.sync()
    out = dict(it=it, errs=errs, Pr=K.to_numpy(Pr), D=K.to_numpy(D))
    mg.close()
    return out


def _sc_driver(rank, world, uid, nx, nt, fused, temporal):
    """the product driver (multi.jl:287-536) with one process per rank on the C-ABI grid"""
    from navierstokes3d_amd import kernels as K
    from navierstokes3d_amd.driver import run_navierstokes3D
    from navierstokes3d_amd.mgpu import MgpuGrid, MultiGpu
    from navierstokes3d_amd.params import multi_params
    p0 = multi_params(nx)
    mg = MultiGpu.create_rank(world, rank, 0, uid, p0.nx, p0.ny, p0.nz, "strict")
    out = run_navierstokes3D(nx=nx, nt=nt, mode="strict", fused=fused, temporal=temporal, grid=MgpuGrid(mg, p0.nx, p0.ny, p0.nz),
                             return_info=True)
    info = out[-1]
    local = {n: K.to_numpy(getattr(info.fields, n)) for n in ("C", "Pr", "Vx", "Vy", "Vz", "divV", "dPrdtau")}
    res = dict(iters=info.iters, errs=info.errs, local=local, gathered=out[:5] if rank == 0 else None)
    mg.close()
    return res


# ---- tests ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world", [2, 3])
def test_update_halo_max_g_gather_one_process_per_rank(hip, world):
    from oracle.driver_ref import gather_z, update_halo_z
    from util import fields
    nx, ny, nz = 13, 9, 7
    got = _run(world, "halo", None, (nx, ny, nz))
    kinds = ["c", "vx", "vy", "vz", "s", "i"]
    host = [dict(zip(kinds, fields(nx, ny, nz, kinds, 100 * (r + 1)))) for r in range(world)]
    for k in kinds:
        update_halo_z(host, k, nz)
        for r in range(world):
            assert np.array_equal(got[r][k], host[r][k]), (k, r)
    h32 = [dict(c=fields(nx, ny, nz, ["c"], 7 + r, np.float32)[0]) for r in range(world)]
    update_halo_z(h32, "c", nz)
    for r in range(world):
        assert np.array_equal(got[r]["c32"], h32[r]["c"])
        assert got[r]["max_g"] == world - 1.5 and np.isnan(got[r]["max_g_nan"])         # NaN-propagating like Julia's maximum
    for kind in ("c", "vx"):
        ref = gather_z([dict(a=fields(nx, ny, nz, [kind], 31 * (r + 1))[0]) for r in range(world)], "a")
        assert np.array_equal(got[0]["gather_" + kind], ref)


@pytest.mark.parametrize("dims,n", [((2, 1, 1), (9, 8, 7)), ((2, 2, 1), (9, 8, 7)), ((1, 2, 2), (8, 7, 9))])
def test_update_halo_on_a_cartesian_topology_one_process_per_rank(hip, dims, n):
    """x and y faces go through k_face_copy and the packed message buffers, then through the same send/recv group"""
    from oracle.driver_ref import cart_coords, update_halo_3d
    from util import fields
    world = dims[0] * dims[1] * dims[2]
    got = _run(world, "halo", dims, n)
    kinds = ["c", "vx", "vy", "vz", "s", "i"]
    host = [dict(zip(kinds, fields(n[0], n[1], n[2], kinds, 100 * (r + 1)))) for r in range(world)]
    for k in kinds:
        update_halo_3d(host, k, n, dims)
        for r in range(world):
            assert np.array_equal(got[r][k], host[r][k]), (k, r, cart_coords(r, dims))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("world,depth,shape,n_iters,force", [(2, 2, (40, 21, 10), 7, 0), (3, 4, (70, 12, 8), 11, 4), (2, 4, (200, 160, 66), 9, 4),
                                                             (3, 3, (33, 9, 5), 10, 3), (4, 4, (24, 15, 9), 8, 0), (2, 1, (24, 15, 9), 4, 0)])
def test_slab_schedule_one_process_per_rank_equals_global_solve(hip, world, depth, shape, n_iters, force, dtype):
    """The deep-ghost slab schedule with the seam sweeps and the send/recv group on the communication stream: every rank's
    planes (halo planes included) equal the single-device solve of the global grid; the residual and the planned depth come out
    of the all-reduce the same on every rank."""
    import torch
    from util import fields, geometry
    nx, ny, nz = shape
    nz_g = world * (nz - 2) + 2
    g = geometry(nx, ny, nz_g)
    Pg, Dg, Rg = fields(nx, ny, nz_g, ["c", "i", "c"], 211, dtype)
    ctx = hip.Context(0, "strict")
    dP, dD = hip.from_numpy(Pg), hip.from_numpy(Dg)
    p = hip.pt_params(dP, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    hip.pt_iterate(dP, dD, hip.from_numpy(Rg), p, n_iters, ctx=ctx)
    res_ref = hip.residual_max(dP, hip.from_numpy(Rg), p, ctx=ctx)
    torch.cuda.synchronize()
    Pref, Dref = hip.to_numpy(dP), hip.to_numpy(dD)
    ctx.close()
    got = _run(world, "slab", shape, depth, n_iters, dtype, force)
    for r in range(world):
        lo = r * (nz - 2)
        assert np.array_equal(got[r]["Pr"], Pref[:, :, lo:lo + nz]), "Pr of rank %d" % r
        assert np.array_equal(got[r]["D"], Dref[:, :, lo:lo + nz - 2]), "dPrdτ of rank %d" % r
        assert got[r]["res"] == res_ref and got[r]["planned"] == got[0]["planned"] and got[r]["ghost"] == got[0]["ghost"]
    # between ranks the planner takes the deepest pass the ghosts allow (an exchange per pass), unless a depth is pinned
    assert got[0]["planned"] == (force if force else min(depth, nz - 2)) and got[0]["ghost"] == max(got[0]["planned"], 1) - 1


@pytest.mark.parametrize("dims,n", [((1, 1, 2), (40, 21, 12)), ((1, 1, 3), (24, 15, 9)), ((2, 1, 1), (14, 16, 16)), ((2, 2, 1), (12, 11, 16))])
def test_pt_solve_slab_one_process_per_rank_equals_global_pt_solve(hip, dims, n):
    import torch
    from util import fields, geometry
    world = dims[0] * dims[1] * dims[2]
    ng = tuple(dims[d] * (n[d] - 2) + 2 for d in range(3))
    g = geometry(*ng)
    Pg, Dg, Rg = fields(ng[0], ng[1], ng[2], ["c", "i", "c"], 433)
    Rg *= 1e-3
    eps, niter, nchk = 1.0e-4, 200, 17
    ctx = hip.Context(0, "strict")
    dP, dD = hip.from_numpy(Pg), hip.from_numpy(Dg)
    pg = hip.pt_params(dP, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    it_ref, errs_ref = hip.pt_solve(dP, dD, hip.from_numpy(Rg), pg, eps, niter, nchk, 0.36, 1000.0, ctx=ctx)
    torch.cuda.synchronize()
    Pref, Dref = hip.to_numpy(dP), hip.to_numpy(dD)
    ctx.close()
    assert len(errs_ref) >= 2
    from oracle.driver_ref import cart_coords
    got = _run(world, "solve", dims, n, np.float64, eps, niter, nchk)
    for r in range(world):
        c = cart_coords(r, dims)
        cut = lambda A, shrink: A[tuple(slice(c[d] * (n[d] - 2), c[d] * (n[d] - 2) + n[d] - shrink) for d in range(3))]
        assert got[r]["it"] == it_ref and got[r]["errs"] == errs_ref, r
        assert np.array_equal(got[r]["Pr"], cut(Pref, 0)) and np.array_equal(got[r]["D"], cut(Dref, 2)), r


@pytest.mark.parametrize("world,fused,temporal", [(2, True, True), (3, True, True), (2, False, False)])
def test_driver_one_process_per_rank_vs_oracle_virtual_ranks(hip, world, fused, temporal):
    from oracle.driver_ref import run_navierstokes3D_ref
    nx, nt = 32, 2
    got = _run(world, "driver", nx, nt, fused, temporal, timeout=400)
    ref = run_navierstokes3D_ref(nx=nx, nt=nt, dims_z=world)
    for r in range(world):
        assert got[r]["iters"] == ref[-1].iters and got[r]["errs"] == ref[-1].errs
        for n, a in got[r]["local"].items():
            assert np.array_equal(a, ref[-1].ranks[r][n], equal_nan=True), (r, n)
    for n, a, b in zip(("C", "Pr", "Vx", "Vy", "Vz"), got[0]["gathered"], ref[:5]):
        assert np.array_equal(a, b, equal_nan=True), n


@pytest.mark.parametrize("gpus", [2, 4])
def test_bench_gpus_n_over_the_rccl_arm(hip, gpus):
    """bench.py --gpus N --transport rccl with the double in place: the collective bring-up (unique id, ncclCommInitRank, the
    verified probe exchange), ns3d_slab_load/_plan/_iterate through send/recv, and the line's own verification of the schedule
    (one pass against {single sweep; update_halo!(Pr)} per iteration) — weak AND strong.  N = 4 is the most rank PROCESSES this
    pool lets share one GPU beside the test process and the launcher (its process guard stops at six GPU processes; five ranks
    were killed by it): two interior ranks with two seams each;
    the EIGHT-rank schedule of scripts/runme3D.sh:18 runs as virtual ranks in tests/test_gpu_configs.py (config E)."""
    build_fake()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NS3D_RCCL_LIB=FAKE_SO, FAKE_RCCL_ARENA_MB="16", NS3D_BENCH_RCCL_TIMEOUT="120")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--grid", "128", "--steps", "8", "--warmup", "4",
           "--transport", "rccl"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == gpus and d["config"]["transport"].startswith("RCCL") and "libfake_rccl.so" in d["config"]["transport"]
    assert d["config"]["rccl_ranks"] == gpus and d["config"]["verified"] is True and d["config"]["verify"]["bitwise"] is True
    planes = -(-126 // gpus) + 2                                    # ImplicitGlobalGrid: nz_g = P·(nz_loc − 2) + 2 ≥ 128
    assert d["strong"]["verified"] is True and d["strong"]["planes_per_rank"] == planes
    assert d["strong"]["global_grid"] == [128, 128, gpus * (planes - 2) + 2]
    # the plan phase's collective trial of the overlap knobs (CUs left to the exchange, interior chunks): one choice for all ranks
    t = d["config"]["overlap_trial"]
    assert t["chosen"] in t["candidates_cus_chunks"] and len(t["ms_per_iteration"]) == len(t["candidates_cus_chunks"]) == 3
    assert [d["config"]["reserved_cus"], d["config"]["interior_chunks"]] == t["chosen"]
