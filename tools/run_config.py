#!/usr/bin/env python3
"""End-to-end timing of the reference configurations on one MI355X (full Chorin steps with the real PT convergence loop).

    python tools/run_config.py --script multi --nx 255 --nt 3 [--mode strict|fast] [--compare-fast]

Prints per-step PT iteration counts, seconds per step and the PT-loop throughput; with --compare-fast also runs FAST
mode on the same case and reports the relative L2 difference of every field (BASELINE north_star: ≤1e-6) and whether the
iteration counts agree.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes3d_amd import kernels as K  # noqa: E402
from navierstokes3d_amd.driver import run_navierstokes3D, runme  # noqa: E402


def run(script, nx, nt, mode, temporal=True, pressure="pt"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if script == "multi":
        out = run_navierstokes3D(nx=nx, nt=nt, mode=mode, temporal=temporal, return_info=True, pressure=pressure)
        info, fields = out[-1], dict(zip(("C", "Pr", "Vx", "Vy", "Vz"), out[:5]))
    else:
        f, info = runme(nx=nx, nt=nt, mode=mode, pressure=pressure)
        fields = {n: K.to_numpy(getattr(f, n)) for n in ("C", "Pr", "Vx", "Vy", "Vz")}
    torch.cuda.synchronize()
    return time.perf_counter() - t0, info, fields


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--script", default="multi", choices=["multi", "gpu"])
    ap.add_argument("--nx", type=int, default=255)
    ap.add_argument("--nt", type=int, default=3)
    ap.add_argument("--mode", default="strict")
    ap.add_argument("--compare-fast", action="store_true")
    ap.add_argument("--marginal", type=int, default=0, metavar="N",
                    help="also run nt+N steps and report (wall(nt+N) - wall(nt)) / N: the cost of one more time step without set-up, "
                         "initial conditions and the final copy to the host")
    ap.add_argument("--compare-direct", action="store_true",
                    help="also run with pressure=\"direct\" (outside parity): seconds per step, and how far its fields are from the PT run's")
    a = ap.parse_args()
    run(a.script, a.nx, 1, a.mode)           # warm-up on the same grid (library load, allocator, one-time tile tuning)
    wall, info, fields = run(a.script, a.nx, a.nt, a.mode)
    p = info.params
    cells = p.nx * p.ny * p.nz
    its = sum(info.iters)
    res = {"script": a.script, "grid": [p.nx, p.ny, p.nz], "nt": a.nt, "mode": a.mode, "pt_iters_per_step": info.iters,
           "last_err_per_step": [e[-1] if e else None for e in info.errs], "wall_s": wall, "s_per_step": wall / a.nt,
           "Mcells_iter_per_s_whole_run": cells * its / wall / 1e6,
           "finite": bool(all(np.isfinite(v).all() for v in fields.values()))}
    if a.marginal > 0:
        wall_m, info_m, _ = run(a.script, a.nx, a.nt + a.marginal, a.mode)
        res["marginal"] = {"extra_steps": a.marginal, "s_per_step": (wall_m - wall) / a.marginal,
                           "pt_iters_of_the_extra_steps": info_m.iters[a.nt:]}
    if a.compare_fast:
        wall_f, info_f, fields_f = run(a.script, a.nx, a.nt, "fast")
        vn = np.sqrt(sum(np.sum(fields[n].astype(np.float64) ** 2) for n in ("Vx", "Vy", "Vz")))
        rel = {}
        for n in fields:
            den = vn if n.startswith("V") else np.sqrt(np.sum(fields[n].astype(np.float64) ** 2))
            rel[n] = float(np.sqrt(np.sum((fields_f[n] - fields[n]) ** 2)) / den) if den > 0 else 0.0
        res["fast"] = {"pt_iters_per_step": info_f.iters, "same_iteration_counts": info_f.iters == info.iters,
                       "wall_s": wall_f, "rel_l2_vs_strict": rel}
    if a.compare_direct:
        run(a.script, a.nx, 1, a.mode, pressure="direct")
        wall_d, info_d, fields_d = run(a.script, a.nx, a.nt, a.mode, pressure="direct")
        vn = np.sqrt(sum(np.sum(fields[n].astype(np.float64) ** 2) for n in ("Vx", "Vy", "Vz")))
        rel = {}
        for n in fields:
            den = vn if n.startswith("V") else np.sqrt(np.sum(fields[n].astype(np.float64) ** 2))
            rel[n] = float(np.sqrt(np.sum((fields_d[n] - fields[n]) ** 2)) / den) if den > 0 else 0.0
        res["direct"] = {"wall_s": wall_d, "s_per_step": wall_d / a.nt, "speedup_per_step": wall / wall_d,
                         "err_per_step": [e[-1] for e in info_d.errs], "rel_l2_vs_pt_run": rel,
                         "finite": bool(all(np.isfinite(v).all() for v in fields_d.values()))}
        if a.marginal > 0:
            wall_dm, _, _ = run(a.script, a.nx, a.nt + a.marginal, a.mode, pressure="direct")
            res["direct"]["marginal_s_per_step"] = (wall_dm - wall_d) / a.marginal
            res["direct"]["marginal_speedup_per_step"] = res["marginal"]["s_per_step"] / res["direct"]["marginal_s_per_step"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
