#!/usr/bin/env python3
"""Cost of one time step through ns3d_time_step (multi.jl:449-477, one rank) with the direct pressure solve: wall per step over N steps of
the 255×153×153 case, to be run under `rocprofv3 --kernel-trace --stats` for the kernels' share.   python tools/ab/step_cost.py [--steps 60]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from navierstokes3d_amd.driver import run_navierstokes3D  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=60); ap.add_argument("--nx", type=int, default=255)
ap.add_argument("--pressure", default="direct")
a = ap.parse_args()
run_navierstokes3D(nx=a.nx, nt=2, pressure=a.pressure)
ts = []
for nt in (3, 3 + a.steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_navierstokes3D(nx=a.nx, nt=nt, pressure=a.pressure)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("wall per further step: %.3f ms (%d steps)" % ((ts[1] - ts[0]) / a.steps * 1e3, a.steps))
