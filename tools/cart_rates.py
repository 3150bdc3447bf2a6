#!/usr/bin/env python3
"""PT loop of P virtual ranks on ONE GPU through the multi-GPU C ABI, by topology: z-slabs (deep-ghost passes), a Cartesian
topology with deep ghosts in x, y and z (solve_box: passes of up to four iterations on the extended box), the same topology with
one sweep + one halo update per iteration (NS3D_CART_DEEP=0) and kernel by kernel (multi.jl:458-471 as written).

    python tools/cart_rates.py [--local 130] [--iters 120] [--dims "1,1,8;2,2,2"]

The ranks share the device, so this prices the schedules (launches, pack/unpack kernels, events, copies), not xGMI.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes3d_amd import kernels as K, lib as L  # noqa: E402
from navierstokes3d_amd.driver import pt_loop_reference  # noqa: E402
from navierstokes3d_amd.mgpu import MgpuGrid, MultiGpu  # noqa: E402
from types import SimpleNamespace  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--local", type=int, default=130)
    ap.add_argument("--iters", type=int, default=120)
    ap.add_argument("--dims", default="1,1,8;2,2,2")
    ap.add_argument("--no-reference", action="store_true")
    ap.add_argument("--depth", type=int, default=0, help="force the iterations per pass (0: the planner's choice)")
    a = ap.parse_args()
    n, its = a.local, a.iters
    for dims in [tuple(int(x) for x in t.split(",")) for t in a.dims.split(";")]:
        P = dims[0] * dims[1] * dims[2]
        mg = MultiGpu.create([0] * P, n, n, n, "strict", dims=dims)
        grid = MgpuGrid(mg, n, n, n)
        for c in mg.contexts:
            c.set_pt_depth(a.depth)
        d = 2.0 ** -9       # a power of two whatever the topology: the same arithmetic build (strictp) for every row
        fs = []
        for r in range(P):
            f = SimpleNamespace(Pr=K.zeros((n, n, n)), dPrdtau=K.zeros((n - 2, n - 2, n - 2)), divV=K.zeros((n, n, n)),
                                Rp=K.zeros((n - 2, n - 2, n - 2)))
            f.divV.permute(2, 1, 0).uniform_(-1e-3, 1e-3)
            fs.append(f)
        col = lambda name: [getattr(f, name) for f in fs]
        mg.update_halo(col("divV"))
        q = SimpleNamespace(rho=1000.0, dt=d, dtau=d / 3.1 ** 0.5, damp=2.0 / n, dx=d, dy=d, dz=d, nchk=10 ** 9, ly=1.0, psc=1000.0,
                            eps=-1.0, owns_outlet=False)
        pt = K.pt_params(fs[0].Pr, q.rho, q.dt, q.dtau, q.damp, q.dx, q.dy, q.dz, L.NS3D_BC_MULTI, False, 0.0, 0.0)
        cells = P * n * n * n
        zslabs = dims[0] == 1 and dims[1] == 1
        modes = ["fused"] if zslabs else ["deep ghosts", "fused"] + ([] if a.no_reference else ["reference"])
        for mode in modes:
            os.environ["NS3D_CART_DEEP"] = "0" if mode == "fused" and not zslabs else "1"
            for rep in range(2):                                # first pass: plan / warm-up
                torch.cuda.synchronize(); mg.sync()
                t0 = time.perf_counter()
                if mode != "reference":
                    mg.pt_solve_slab(col("Pr"), col("dPrdtau"), col("divV"), pt, -1.0, its, 0, 1.0, 1.0)
                else:
                    pt_loop_reference(mg.contexts, grid, fs, [q] * P, its)
                mg.sync(); torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            print(json.dumps({"dims": dims, "local": [n, n, n], "ranks_on_one_gpu": P, "loop": mode, "iters": its,
                              "pass_depth": mg.pass_depth() if (zslabs or mode == "deep ghosts") else 1,
                              "ms_per_iteration": round(dt / its * 1e3, 4),
                              "Mcells_iter_per_s": round(cells * its / dt / 1e6)}), flush=True)
        mg.close()


if __name__ == "__main__":
    main()
