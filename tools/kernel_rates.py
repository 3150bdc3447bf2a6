#!/usr/bin/env python3
"""Achieved algorithmic bandwidth of every per-time-step kernel (the ones that run once per step).

    python tools/kernel_rates.py [--n 512] [--mode strict] [--reps 20]

Times each C-ABI entry point with HIP events on random fields and prints ms per call, algorithmic GB/s (bytes of the
table in DESIGN.md §4 × cells) and the fraction of the 8 TB/s HBM peak, as JSON lines.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes3d_amd import kernels as K, lib as L  # noqa: E402
from navierstokes3d_amd.params import cavity_params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--nz", type=int, default=None)
    ap.add_argument("--mode", default="strict")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="", help="comma-separated kernel names (default: all)")
    a = ap.parse_args()
    p = cavity_params(a.n, a.nz)
    nx, ny, nz = p.nx, p.ny, p.nz
    N = nx * ny * nz
    ctx = K.Context(0, a.mode, async_=True)

    def rnd(*s):
        t = K.zeros(s)
        t.permute(2, 1, 0).uniform_(-1.0, 1.0)
        return t

    Pr, C, Co, txx, tyy, tzz, dV = (rnd(nx, ny, nz) for _ in range(7))
    Vx, Vxo = rnd(nx + 1, ny, nz), rnd(nx + 1, ny, nz)
    Vy, Vyo = rnd(nx, ny + 1, nz), rnd(nx, ny + 1, nz)
    Vz, Vzo = rnd(nx, ny, nz + 1), rnd(nx, ny, nz + 1)
    txy, txz, tyz = (rnd(nx - 1, ny - 1, nz - 1) for _ in range(3))
    D, Rp = rnd(nx - 2, ny - 2, nz - 2), rnd(nx - 2, ny - 2, nz - 2)
    for V in (Vx, Vy, Vz):
        V.mul_(1e-3)                                   # keeps the back-tracked departure points near their cells
    pt = K.pt_params(Pr, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, L.NS3D_BC_MULTI, True, 0.0, 0.0)
    cases = [
        ("update_tau", 72, lambda: K.update_tau(txx, tyy, tzz, txy, txz, tyz, Vx, Vy, Vz, p.mu, p.dx, p.dy, p.dz, ctx=ctx)),
        ("predict_V", 96, lambda: K.predict_V(Vx, Vy, Vz, txx, tyy, tzz, txy, txz, tyz, p.rho, 0.0, p.dt, p.dx, p.dy, p.dz, ctx=ctx)),
        ("predict_fused", 48, lambda: K.predict_fused(Vxo, Vyo, Vzo, Vx, Vy, Vz, p.mu, p.rho, 0.0, p.dt, p.dx, p.dy, p.dz, ctx=ctx)),
        ("update_divV", 32, lambda: K.update_divV(dV, Vx, Vy, Vz, p.dx, p.dy, p.dz, ctx=ctx)),
        ("update_dPrdtau", 32, lambda: K.update_dPrdtau(Pr, D, dV, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, ctx=ctx)),
        ("update_Pr", 24, lambda: K.update_Pr(Pr, D, 1e-9, ctx=ctx)),
        ("compute_res", 24, lambda: K.compute_res(Rp, Pr, dV, p.rho, p.dt, p.dx, p.dy, p.dz, ctx=ctx)),
        ("max_abs", 8, lambda: K.max_abs(Rp, ctx=ctx)),
        ("residual_max", 16, lambda: K.residual_max(Pr, dV, pt, ctx=ctx)),
        ("correct_V", 56, lambda: K.correct_V(Vx, Vy, Vz, Pr, p.dt, p.rho, p.dx, p.dy, p.dz, ctx=ctx)),
        ("set_bc_Pr", 0, lambda: K.set_bc_Pr_multi(Pr, True, 0.0, ctx=ctx)),
        ("set_bc_Vel", 0, lambda: K.set_bc_Vel_multi(Vx, Vy, Vz, True, 1.0, ctx=ctx)),
        ("set_cylinder", 0, lambda: K.set_cylinder(C, Vx, Vy, Vz, 0.0025, 0.0025, -0.2, 0.0, 0.0, 1.0, -0.5, -0.5, -0.5, 1.0, 1.0, 1.0,
                                                   p.dx, p.dy, p.dz, ctx=ctx)),
        ("copy", 16, lambda: K.copy(Co, C, ctx=ctx)),
        ("advect", 56, lambda: K.advect(Vx, Vxo, Vy, Vyo, Vz, Vzo, C, Co, p.dt, p.dx, p.dy, p.dz, ctx=ctx)),
        ("advect_fixed", 64, lambda: K.advect(Vx, Vxo, Vy, Vyo, Vz, Vzo, C, Co, p.dt, p.dx, p.dy, p.dz, faithful=False, ctx=ctx)),
        ("copy_advect", 56, lambda: K.copy_advect(Vxo, Vx, Vyo, Vy, Vzo, Vz, Co, C, p.dt, p.dx, p.dy, p.dz, ctx=ctx)),
    ]
    only = [q for q in a.only.split(",") if q]
    if not only or "advect_stream" in only:
        # the stream of the cylinder case at CFL_adv = 1 (multi.jl:342): departure points 0.7-1.3 cells upstream in x, within
        # 0.3 cells in y and z
        Sx, Sy, Sz = rnd(nx + 1, ny, nz), rnd(nx, ny + 1, nz), rnd(nx, ny, nz + 1)
        Sx.mul_(0.3).add_(1.0).mul_(p.dx / p.dt)
        Sy.mul_(0.3 * p.dy / p.dt)
        Sz.mul_(0.3 * p.dz / p.dt)
        cases.append(("advect_stream", 56, lambda: K.advect(Vx, Sx, Vy, Sy, Vz, Sz, C, Co, p.dt, p.dx, p.dy, p.dz, ctx=ctx)))
    for name, bpc, fn in cases:
        if only and name not in only:
            continue
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        gbs = bpc * N / (ms * 1e-3) / 1e9 if bpc else None
        print(json.dumps({"kernel": name, "grid": [nx, ny, nz], "mode": a.mode, "ms": round(ms, 4),
                          "alg_bytes_per_cell": bpc, "alg_GBps": None if gbs is None else round(gbs, 1),
                          "frac_of_8TBps": None if gbs is None else round(gbs / 8000.0, 3)}), flush=True)


if __name__ == "__main__":
    main()
