#!/usr/bin/env python3
"""Time-to-solution of the pressure solve: the reference's pseudo-transient loop (multi.jl:458-471, fused: ns3d_pt_solve) against the
direct solve (ns3d_poisson_direct, SURVEY §8 f4 — outside parity) on the reference's own second time step of the cylinder case
(255×153×153 by default: ∇V of a real predictor step, warm start as the reference has it).  One JSON line per grid."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes3d_amd import kernels as K, lib as L  # noqa: E402
from navierstokes3d_amd.driver import run_navierstokes3D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=255)
ap.add_argument("--steps", type=int, default=2)
a = ap.parse_args()
# state after `steps` reference steps; then ONE more predictor gives the right-hand side both solvers see
out = run_navierstokes3D(nx=a.nx, nt=a.steps, mode="strict", return_info=True)
info = out[-1]; f, p, ctx = info.fields, info.params, info.ctx
K.update_tau(f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, f.Vx, f.Vy, f.Vz, p.mu, p.dx, p.dy, p.dz, ctx=ctx)
K.predict_V(f.Vx, f.Vy, f.Vz, f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, p.rho, p.g, p.dt, p.dx, p.dy, p.dz, ctx=ctx)
K.set_cylinder(f.C, f.Vx, f.Vy, f.Vz, p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.xco_g, p.yco_g, p.zco_g, p.lx, p.ly, p.lz, p.dx, p.dy, p.dz, ctx=ctx)
K.update_divV(f.divV, f.Vx, f.Vy, f.Vz, p.dx, p.dy, p.dz, ctx=ctx)
pt = K.pt_params(f.Pr, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, L.NS3D_BC_MULTI, p.owns_outlet, 0.0, p.g)
P0, D0 = K.clone(f.Pr), K.clone(f.dPrdtau)
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        f.Pr.copy_(P0); f.dPrdtau.copy_(D0); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best, r
t_pt, (its, errs) = timed(lambda: K.pt_solve(f.Pr, f.dPrdtau, f.divV, pt, p.eps, p.niter, p.nchk, p.ly * p.ly, p.psc, ctx=ctx))
P_pt = K.clone(f.Pr)
t_d, _ = timed(lambda: K.poisson_direct(f.Pr, f.dPrdtau, f.divV, pt, ctx=ctx), reps=5)
res_d = K.residual_max(f.Pr, f.divV, pt, ctx=ctx) * (p.ly * p.ly) / p.psc
rel = (torch.linalg.vector_norm(f.Pr - P_pt) / torch.linalg.vector_norm(f.Pr)).item()
mx, my, mz = p.nx - 2, p.ny - 2, p.nz - 2
flops = 4.0 * (mx + my + mz) * mx * my * mz
print(json.dumps({"grid": [p.nx, p.ny, p.nz], "after_steps": a.steps, "previous_steps_pt_iterations": info.iters,
                  "pt": {"iterations": its, "err_at_exit": errs[-1] if errs else None, "eps": p.eps, "ms": t_pt * 1e3},
                  "direct": {"ms": t_d * 1e3, "err": res_d, "gemm_gflop": flops / 1e9, "gemm_tflops": flops / t_d / 1e12},
                  "speedup": t_pt / t_d, "rel_l2_pt_vs_direct": rel}))
