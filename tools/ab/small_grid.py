#!/usr/bin/env python3
"""PT loop of a small grid through ns3d_pt_solve (HIP-graph replay of the residual-check blocks): µs per iteration by
iterations per pass and tile shape.   python tools/ab/small_grid.py [--nx 63 --ny 38 --nz 38]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from navierstokes3d_amd import kernels as K, lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=63); ap.add_argument("--ny", type=int, default=38); ap.add_argument("--nz", type=int, default=38)
ap.add_argument("--niter", type=int, default=1480); ap.add_argument("--nchk", type=int, default=37)
ap.add_argument("--cases", default="1:0,2:0,2:716,3:600,4:600")
ap.add_argument("--persist", type=int, default=0, help="ns3d_set_persist_mode for every case (1: k_pt_persist, a whole check block per launch)")
a = ap.parse_args()
nx, ny, nz = a.nx, a.ny, a.nz
d = 1.0 / nx
for case in a.cases.split(","):
    depth, var = (int(q) for q in case.split(":"))
    ctx = K.Context(0, "strict", async_=True)
    ctx.set_persist_mode(a.persist)
    if depth == 1:
        ctx.set_pt_depth(1)
    else:
        ctx.set_pt_depth(depth)
        if depth == 2 and var < 3000:
            if var: ctx.set_pt2_variant(var)
        else:
            ctx.set_ptn_variant(var)
    Pr, D, rhs = K.zeros((nx, ny, nz)), K.zeros((nx - 2, ny - 2, nz - 2)), K.zeros((nx, ny, nz))
    rhs.permute(2, 1, 0).uniform_(-1e-3, 1e-3)
    pt = K.pt_params(Pr, 1000.0, d, d / 3.1 ** 0.5, 2.0 / nx, d, 0.6 / ny, 0.6 / nz, L.NS3D_BC_MULTI, True, 0.0, 0.0)
    best = 1e9
    for rep in range(4):
        Pr.zero_(); D.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it, errs = K.pt_solve(Pr, D, rhs, pt, -1.0, a.niter, a.nchk, 1.0, 1.0, ctx=ctx)
        ctx.sync(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("persist %d " % a.persist, end="")
    print("depth %d variant %5d: %.2f us per iteration (last depth %d, ptn %d, pt2 %d) err %.6e" % (
        depth, var, best / a.niter * 1e6, ctx.last_pt_depth(), ctx.last_ptn_variant(), ctx.last_pt2_variant(), errs[-1]), flush=True)
    ctx.close()
