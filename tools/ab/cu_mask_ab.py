#!/usr/bin/env python3
"""What the two "make room for the exchange" knobs cost in sweep time, on ONE GPU (round 4):
ns3d_reserve_cus (compute launches leave n CUs out: hipExtStreamCreateWithCUMask) and ns3d_mgpu_set_interior_chunks (the interior
sweep of a z-slab pass in k launches).  Part 1: the four-iteration pass at 512^3 under the mask.  Part 2: the z-slab schedule of
P virtual ranks (all on this GPU: this prices the knobs, not xGMI), bits compared with the unmasked one-chunk run.

    NS3D_RESERVE_CUS_LAYOUT=0|1 python tools/ab/cu_mask_ab.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from navierstokes3d_amd import kernels as K, lib as L  # noqa: E402
from navierstokes3d_amd.mgpu import MultiGpu  # noqa: E402
from navierstokes3d_amd.params import cavity_params  # noqa: E402


def part1():
    p = cavity_params(512, 512)
    nx, ny, nz = p.nx, p.ny, p.nz
    Pr, Pb = K.zeros((nx, ny, nz)), K.zeros((nx, ny, nz))
    D, D2, rhs = K.zeros((nx - 2, ny - 2, nz - 2)), K.zeros((nx - 2, ny - 2, nz - 2)), K.zeros((nx, ny, nz))
    rhs.permute(2, 1, 0).uniform_(-1e-3, 1e-3)
    pt = K.pt_params(Pr, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, L.NS3D_BC_MULTI, False, 0.0, 0.0)
    ctx = K.Context(0, "strict", async_=True)
    ctx.set_ptn_variant(2891)
    print("layout", os.environ.get("NS3D_RESERVE_CUS_LAYOUT", "0"), " 512^3 fp64 strict, four-iteration pass (variant 2891 / 2800), ms per pass")
    for v in (2891, 2800):
        ctx.set_ptn_variant(v)
        for n in (0, 8, 16, 32, 64, 0):
            st = ctx.reserve_cus(n)
            torch.cuda.synchronize()
            with torch.cuda.stream(st):
                for rep in range(2):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        K.pt_sweepn(4, Pr, Pb, D, D2, rhs, pt, ctx=ctx)
                        K.pt_sweepn(4, Pb, Pr, D2, D, rhs, pt, ctx=ctx)
                    e1.record()
                    torch.cuda.synchronize()
                print("  variant %d reserve %2d CUs: %.4f ms per pass" % (v, ctx.lib.ns3d_reserved_cus(ctx.handle), e0.elapsed_time(e1) / 20), flush=True)
    ctx.reserve_cus(0)
    ctx.close()


def part2(P, nz_loc, n=512, iters=48):
    p = cavity_params(n, nz_loc)
    nx, ny, nz = p.nx, p.ny, p.nz
    gen = torch.Generator(device="cuda"); gen.manual_seed(77)
    mk = lambda *s: [K.zeros(s) for _ in range(P)]
    Pr, D, R = mk(nx, ny, nz), mk(nx - 2, ny - 2, nz - 2), mk(nx, ny, nz)
    for r in R:
        r.permute(2, 1, 0).uniform_(-1e-3, 1e-3, generator=gen)
    ref = None
    print("%d z-slab ranks of %dx%dx%d on one GPU, ms per PT iteration (all ranks)" % (P, nx, ny, nz))
    for reserve, chunks in [(0, 1), (0, 2), (0, 4), (8, 1), (16, 1), (32, 1), (16, 2), (0, 1)]:
        mg = MultiGpu.create([0] * P, nx, ny, nz, "strict", own_streams=True)
        if reserve:
            mg.reserve_cus(reserve)
        mg.set_interior_chunks(chunks)
        pt = K.pt_params(Pr[0], p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, L.NS3D_BC_MULTI, False, 0.0, 0.0)
        for q in range(P):
            Pr[q].zero_(); D[q].zero_()
        torch.cuda.synchronize()
        mg.slab_load(Pr, D, R, pt)
        depth = mg.slab_plan()
        mg.slab_iterate(8)
        mg.sync()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        mg.slab_iterate(iters)
        mg.sync()
        t = (time.perf_counter() - t0) * 1e3 / iters
        mg.slab_store(Pr, D)
        mg.sync()
        torch.cuda.synchronize()
        sig = [x.clone() for x in Pr]
        same = True if ref is None else all(torch.equal(a.view(torch.int64), b.view(torch.int64)) for a, b in zip(sig, ref))
        if ref is None:
            ref = sig
        print("  reserve %2d CUs, interior in %d launch(es), %d iterations per pass: %.4f ms per iteration   same bits as the first run: %s"
              % (reserve, chunks, depth, t, same), flush=True)
        mg.close()


if __name__ == "__main__":
    part1()
    part2(8, 66)
    part2(4, 130)
