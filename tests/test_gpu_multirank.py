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


def _worker(rank, world, port, nx, nt, fused, q):
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
        out = run_navierstokes3D(nx=nx, nt=nt, mode="strict", fused=fused, grid=grid, device=0, return_info=True)
        info = out[-1]
        local = {n: K.to_numpy(getattr(info.fields, n)) for n in ("C", "Pr", "Vx", "Vy", "Vz", "divV", "dPrdtau")}
        q.put((rank, info.iters, local, out[:5] if rank == 0 else None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "ERROR", traceback.format_exc(), None))


@pytest.mark.parametrize("fused", [True, False])
def test_two_ranks_one_gpu(hip, fused):
    from oracle.driver_ref import run_navierstokes3D_ref
    world, nx, nt = 2, 20, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nx, nt, fused, q)) for r in range(world)]
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
    for r in range(world):
        _, iters, local, _ = results[r]
        assert iters == ref[-1].iters
        for n, a in local.items():
            assert np.array_equal(a, ref[-1].ranks[r][n], equal_nan=True), (r, n)
    for n, a, b in zip(("C", "Pr", "Vx", "Vy", "Vz"), results[0][3], ref[:5]):
        assert np.array_equal(a, b, equal_nan=True), n
