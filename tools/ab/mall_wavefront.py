#!/usr/bin/env python3
"""A/B (round 3): do consecutive four-iteration passes gain from meeting in the Infinity Cache?  Instead of NP full passes over the
grid (each streams 5 arrays through HBM), a window of Z planes slides up the grid and the NP passes follow each other through it, each
four planes behind the previous one (pass q sweeps planes [a-4q, a+Z-4q) once pass q-1 has produced up to a+Z-4(q-1)): what pass q-1
wrote is re-read by pass q a few launches later — from the 256 MB memory-side cache if it is still there.  Uses the production kernel
through ns3d_pt_sweepn's plane ranges; the result is compared bit for bit with NP full passes.
    python tools/ab/mall_wavefront.py [--n 512 --passes 3 --windows 16,24,32,48,64]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from navierstokes3d_amd import kernels as K, lib as L  # noqa: E402
from navierstokes3d_amd.params import cavity_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512); ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--windows", default="16,24,32,48,64"); ap.add_argument("--variantn", type=int, default=2891)
a = ap.parse_args()
p = cavity_params(a.n)
nx, ny, nz = p.nx, p.ny, p.nz
ctx = K.Context(0, "strict", async_=True)
ctx.set_ptn_variant(a.variantn)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
z = lambda *s: K.zeros(s)
P = [z(nx, ny, nz) for _ in range(a.passes + 1)]
D = [z(nx - 2, ny - 2, nz - 2) for _ in range(a.passes + 1)]
R = z(nx, ny, nz); R.permute(2, 1, 0).uniform_(-1e-3, 1e-3, generator=gen)
P[0].permute(2, 1, 0).uniform_(-1, 1, generator=gen); D[0].permute(2, 1, 0).uniform_(-1, 1, generator=gen)
pt = K.pt_params(P[0], p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, L.NS3D_BC_MULTI, False, 0.0, 0.0)

def full():
    for q in range(a.passes):
        K.pt_sweepn(4, P[q], P[q + 1], D[q], D[q + 1], R, pt, ctx=ctx)

def wave(Z):
    top = [1] * a.passes                      # next plane each pass has to produce
    a0 = 1
    while top[-1] < nz - 1:
        for q in range(a.passes):
            lim = nz - 1 if (q == 0 or top[q - 1] >= nz - 1) else top[q - 1] - 4      # inputs complete up to lim+4
            hi = min(lim, a0 + Z - 4 * q, nz - 1)
            if hi > top[q]:
                K.pt_sweepn(4, P[q], P[q + 1], D[q], D[q + 1], R, pt, top[q], hi, ctx=ctx)
                top[q] = hi
        a0 += Z

def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best

full(); torch.cuda.synchronize()
ref_P, ref_D = P[-1].clone(), D[-1].clone()
t_full = timed(full)
print("%d full passes of four iterations: %.3f ms (%.0f Mcells*it/s)" % (a.passes, t_full * 1e3, nx * ny * nz * 4 * a.passes / t_full / 1e6), flush=True)
for Z in [int(q) for q in a.windows.split(",")]:
    for q in range(1, a.passes + 1):
        P[q].zero_(); D[q].zero_()
    wave(Z); torch.cuda.synchronize()
    same = torch.equal(P[-1].view(torch.int64), ref_P.view(torch.int64)) and torch.equal(D[-1].view(torch.int64), ref_D.view(torch.int64))
    t = timed(lambda: wave(Z))
    print("window of %3d planes: %.3f ms (%.0f Mcells*it/s) = %.3f x, bit-identical %s" % (Z, t * 1e3, nx * ny * nz * 4 * a.passes / t / 1e6, t_full / t, same), flush=True)
