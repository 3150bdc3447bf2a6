"""Regenerates the golden fixtures from the CPU oracle (the reference itself is Julia and cannot run in the build
container; its own known-answer test test/test3D.jl is stale — SURVEY.md §4 — so these are ORACLE goldens, parity
unpinned against the original Julia program).

    python tests/golden/make_golden.py

Writes (all < 1 MB):
  multi_nx24.npz   multi.jl semantics, 24×15×15, gathered fields after nt = 1, 2, 5 + PT iteration counts/errs
  multi_nx63.npz   multi.jl semantics, 63×38×38, nt = 20: iteration counts, err history, fields sub-sampled by 3
  gpu_nx40.npz     gpu.jl semantics, 40×24×24, nt = 2: full local fields + counts
  kernels_17x9x5.npz  per-kernel known-answer vectors on seeded U(-1,1) inputs (seeds in tests/util.py)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as K  # noqa: E402
from oracle.driver_ref import run_navierstokes3D_ref, runme_ref  # noqa: E402
from util import fields, geometry  # noqa: E402


def main():
    out = {}
    for nt in (1, 2, 5):
        C, Pr, Vx, Vy, Vz, info = run_navierstokes3D_ref(nx=24, nt=nt)
        for n, a in zip(("C", "Pr", "Vx", "Vy", "Vz"), (C, Pr, Vx, Vy, Vz)):
            out["nt%d_%s" % (nt, n)] = a
        out["nt%d_iters" % nt] = np.array(info.iters)
        out["nt%d_lasterr" % nt] = np.array([e[-1] for e in info.errs])
    np.savez_compressed(os.path.join(HERE, "multi_nx24.npz"), **out)

    C, Pr, Vx, Vy, Vz, info = run_navierstokes3D_ref(nx=63, nt=20)
    out = {"iters": np.array(info.iters), "lasterr": np.array([e[-1] for e in info.errs])}
    for n, a in zip(("C", "Pr", "Vx", "Vy", "Vz"), (C, Pr, Vx, Vy, Vz)):
        out[n] = a[::3, ::3, ::3].copy()
        out[n + "_l2"] = np.array(np.sqrt(np.sum(a * a)))
    np.savez_compressed(os.path.join(HERE, "multi_nx63.npz"), **out)

    f, info = runme_ref(nx=40, nt=2)
    out = {"iters": np.array(info.iters), "lasterr": np.array([e[-1] for e in info.errs])}
    for n in ("C", "Pr", "Vx", "Vy", "Vz"):
        out[n] = np.asarray(f[n])
    np.savez_compressed(os.path.join(HERE, "gpu_nx40.npz"), **out)

    nx, ny, nz = 17, 9, 5
    g = geometry(nx, ny, nz)
    out = {}
    a = fields(nx, ny, nz, ["c", "c", "c", "s", "s", "s", "vx", "vy", "vz"], 1)
    K.update_tau(*a, g["mu"], g["dx"], g["dy"], g["dz"])
    for q, n in enumerate(("txx", "tyy", "tzz", "txy", "txz", "tyz")):
        out["update_tau_" + n] = a[q]
    a = fields(nx, ny, nz, ["vx", "vy", "vz", "c", "c", "c", "s", "s", "s"], 1)
    K.predict_V(*a, g["rho"], g["g"], g["dt"], g["dx"], g["dy"], g["dz"])
    for q, n in enumerate(("Vx", "Vy", "Vz")):
        out["predict_V_" + n] = a[q]
    a = fields(nx, ny, nz, ["c", "i", "c"], 1)
    K.update_dPrdtau(*a, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"])
    out["update_dPrdtau"] = a[1]
    a = fields(nx, ny, nz, ["vx", "vy", "vz", "c"], 1)
    K.correct_V(*a, g["dt"], g["rho"], g["dx"], g["dy"], g["dz"])
    for q, n in enumerate(("Vx", "Vy", "Vz")):
        out["correct_V_" + n] = a[q]
    o = fields(nx, ny, nz, ["vx", "vy", "vz", "c"], 21)
    w = [np.asfortranarray(np.zeros_like(x)) for x in o]
    K.advect(w[0], o[0], w[1], o[1], w[2], o[2], w[3], o[3], 1.7 * g["dx"], g["dx"], g["dy"], g["dz"], True)
    for q, n in enumerate(("Vx", "Vy", "Vz", "C")):
        out["advect_" + n] = w[q]
    np.savez_compressed(os.path.join(HERE, "kernels_17x9x5.npz"), **out)
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
