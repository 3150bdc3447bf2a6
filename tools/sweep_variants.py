#!/usr/bin/env python3
"""Interleaved A/B timing of the fused PT sweep variants in ONE process (cdna_hip_programming.md §5.4 rule 24).

    python tools/sweep_variants.py [--n 512] [--nz 512] [--rounds 3] [--iters 30] [--variants 0,100,...] [--modes strict,fast]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes3d_amd import kernels as K, lib as L  # noqa: E402
from navierstokes3d_amd.params import cavity_params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--ny", type=int, default=None)
    ap.add_argument("--nz", type=int, default=None)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--variants", default="0,100,200,700,2000,2200")
    ap.add_argument("--modes", default="strict,fast")
    ap.add_argument("--variants2", default="", help="temporal-blocking (two iterations per launch) shapes to time")
    ap.add_argument("--variantsn", default="", help="N-iteration sweeps to time, as levels:variant (e.g. 3:100,3:1100,4:200)")
    ap.add_argument("--dtype", default="f64")
    a = ap.parse_args()
    p = cavity_params(a.n, a.nz)
    nx, ny, nz = p.nx, (a.ny or p.ny), p.nz
    tdt = torch.float64 if a.dtype == "f64" else torch.float32
    isz = 8 if a.dtype == "f64" else 4
    Pr, Pb = K.zeros((nx, ny, nz), tdt), K.zeros((nx, ny, nz), tdt)
    D, rhs = K.zeros((nx - 2, ny - 2, nz - 2), tdt), K.zeros((nx, ny, nz), tdt)
    D2 = K.zeros((nx - 2, ny - 2, nz - 2), tdt)
    rhs.permute(2, 1, 0).uniform_(-1e-3, 1e-3)
    abytes = isz * (nx * ny * nz + 4 * (nx - 2) * (ny - 2) * (nz - 2))
    ctxs = {m: K.Context(0, m, async_=True) for m in a.modes.split(",")}
    variants = [int(v) for v in a.variants.split(",") if v != ""]
    res = {}
    pt = K.pt_params(Pr, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, L.NS3D_BC_MULTI, False, 0.0, 0.0)
    for rnd in range(a.rounds + 1):
        for m, ctx in ctxs.items():
            for v in variants:
                ctx.set_pt_variant(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters // 2):
                    K.pt_sweep(Pr, Pb, D, rhs, pt, 1, nz - 1, ctx=ctx)
                    K.pt_sweep(Pb, Pr, D, rhs, pt, 1, nz - 1, ctx=ctx)
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    res.setdefault((m, v), []).append(e0.elapsed_time(e1) / (2 * (a.iters // 2)))
    variants2 = [int(v) for v in a.variants2.split(",") if v != ""]
    for rnd in range(a.rounds + 1):
        for m, ctx in ctxs.items():
            for v in variants2:
                ctx.set_pt2_variant(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters // 4):
                    K.pt_sweep2(Pr, Pb, D, D2, rhs, pt, ctx=ctx)
                    K.pt_sweep2(Pb, Pr, D2, D, rhs, pt, ctx=ctx)
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:   # per ITERATION (a launch does two)
                    res.setdefault((m + "-x2", v), []).append(e0.elapsed_time(e1) / (4 * (a.iters // 4)))
    for rnd in range(a.rounds + 1):
        for m, ctx in ctxs.items():
            for item in [q for q in a.variantsn.split(",") if q]:
                nlev, v = (int(t) for t in item.split(":"))
                ctx.set_ptn_variant(v)
                try:
                    K.pt_sweepn(nlev, Pr, Pb, D, D2, rhs, pt, ctx=ctx)
                except L.Ns3dError:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = max(1, a.iters // (2 * nlev))
                e0.record()
                for _ in range(reps):
                    K.pt_sweepn(nlev, Pr, Pb, D, D2, rhs, pt, ctx=ctx)
                    K.pt_sweepn(nlev, Pb, Pr, D2, D, rhs, pt, ctx=ctx)
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:   # per ITERATION (a launch does nlev)
                    res.setdefault((m + "-x%d" % nlev, v), []).append(e0.elapsed_time(e1) / (2 * nlev * reps))
    print("grid %dx%dx%d %s  algorithmic bytes/launch %.1f MB" % (nx, ny, nz, a.dtype, abytes / 1e6))
    print("%-10s %-8s %10s %10s %12s %8s   (ms per PT iteration)" % ("mode", "variant", "min ms", "med ms", "Mcell-it/s", "%8TB/s"))
    for (m, v), ts in sorted(res.items()):
        ts = sorted(ts)
        tmin, tmed = ts[0], ts[len(ts) // 2]
        print("%-10s %-8d %10.4f %10.4f %12.0f %8.1f" % (m, v, tmin, tmed, nx * ny * nz / tmin / 1e3,
                                                       abytes / (tmin * 1e-3) / 8e12 * 100))


if __name__ == "__main__":
    main()
