#!/usr/bin/env python3
"""What the z-slab schedule costs: P virtual ranks of nx×ny×nz_local on ONE GPU through ns3d_pt_solve_slab (seam sweeps, ghost
exchange by device copies, interior sweeps — executed one rank after the other on the shared device) against ns3d_pt_solve of the
global grid.  python tools/ab/slab_overhead.py [--n 512 --nz-local 258 --ranks 2 --iters 96]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from navierstokes3d_amd import kernels as K, lib as L  # noqa: E402
from navierstokes3d_amd.mgpu import MultiGpu  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512); ap.add_argument("--nz-local", type=int, default=258)
ap.add_argument("--ranks", type=int, default=2); ap.add_argument("--iters", type=int, default=96)
ap.add_argument("--temporal", default="4")
a = ap.parse_args()
n, nzl, P, its = a.n, a.nz_local, a.ranks, a.iters
nzg = P * (nzl - 2) + 2
d = 1.0 / n
def fields(nz):
    Pr, D, R = K.zeros((n, n, nz)), K.zeros((n - 2, n - 2, nz - 2)), K.zeros((n, n, nz))
    R.permute(2, 1, 0).uniform_(-1e-3, 1e-3)
    return Pr, D, R
def timed(fn):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
ctx = K.Context(0, "strict", async_=True)
Pr, D, R = fields(nzg)
pt = K.pt_params(Pr, 1000.0, d, d / 3.1 ** 0.5, 2.0 / n, d, d, d, L.NS3D_BC_MULTI, False, 0.0, 0.0)
t1 = timed(lambda: (K.pt_solve(Pr, D, R, pt, -1.0, its, 0, 1.0, 1.0, ctx=ctx), ctx.sync()))
print("global %dx%dx%d: %.4f ms per iteration (passes of %d)" % (n, n, nzg, t1 / its * 1e3, ctx.last_pt_depth()), flush=True)
del Pr, D, R; ctx.close(); torch.cuda.empty_cache()
fs = [fields(nzl) for _ in range(P)]
col = lambda j: [f[j] for f in fs]
for temporal in [int(q) for q in a.temporal.split(",")]:
    mg = MultiGpu.create([0] * P, n, n, nzl, "strict")
    mg.set_temporal(temporal)
    ptl = K.pt_params(fs[0][0], 1000.0, d, d / 3.1 ** 0.5, 2.0 / n, d, d, d, L.NS3D_BC_MULTI, False, 0.0, 0.0)
    t2 = timed(lambda: (mg.pt_solve_slab(col(0), col(1), col(2), ptl, -1.0, its, 0, 1.0, 1.0), mg.sync()))
    print("%d slabs of %dx%dx%d on one GPU, ghost depth %d: %.4f ms per iteration (passes of %d) = %.3f x the global solve" % (
        P, n, n, nzl, temporal, t2 / its * 1e3, mg.pass_depth(), t2 / t1), flush=True)
    mg.close()
