#!/usr/bin/env python3
"""Randomised check of ns3d_pt_solve_slab on virtual ranks of ONE GPU against ns3d_pt_solve on the global grid, bit for bit:
random topologies (z-slabs and Cartesian: solve_slab / solve_box / the depth-1 form), local extents, ghost depths, pinned pass
depths, iteration counts, residual-check intervals, element types, outlet rule on / off.

    python tools/fuzz_mgpu.py [--cases 60] [--seed 1]

Prints one line per case; exits non-zero at the first mismatch (the case's parameters are in the line)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from navierstokes3d_amd import kernels as K  # noqa: E402
from navierstokes3d_amd.mgpu import MultiGpu  # noqa: E402
from util import fields, geometry  # noqa: E402

DIMS = [(1, 1, 2), (1, 1, 3), (2, 1, 1), (1, 2, 1), (2, 2, 1), (2, 1, 2), (1, 2, 2), (2, 2, 2), (3, 1, 1), (1, 3, 1), (3, 2, 1), (1, 1, 4),
        (1, 1, 8), (1, 1, 8), (1, 1, 6), (2, 1, 4)]      # eight z-slabs: scripts/runme3D.sh:18, BASELINE configs[4]


def coords(r, dims):
    return (r // (dims[1] * dims[2]), (r // dims[2]) % dims[1], r % dims[2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    for case in range(a.cases):
        dims = DIMS[rng.integers(len(DIMS))]
        n = (int(rng.integers(5, 150)), int(rng.integers(5, 60)), int(rng.integers(5, 40)))
        if rng.random() < 0.3:
            n = tuple(int(rng.integers(5, 12)) for _ in range(3))
        depth = int(rng.integers(1, 5))
        force = int(rng.integers(0, depth + 1))
        force = 0 if force == 1 else force
        niter, nchk = int(rng.integers(1, 40)), int(rng.integers(0, 12))
        dtype = np.float64 if rng.random() < 0.7 else np.float32
        outlet = bool(rng.random() < 0.6)
        deep = "0" if rng.random() < 0.15 else "1"
        N = tuple(dims[d] * (n[d] - 2) + 2 for d in range(3))
        g = geometry(*N)
        g["dtau"] = 0.8 / np.sqrt(1.0 / g["dx"] ** 2 + 1.0 / g["dy"] ** 2 + 1.0 / g["dz"] ** 2)
        Pg, Dg, Rg = fields(*N, ["c", "i", "c"], 1000 + case, dtype)
        Pg *= 1e-3; Dg *= 1e-3; Rg *= 1e-6
        ctx = K.Context(0, "strict")
        dP, dD = K.from_numpy(Pg), K.from_numpy(Dg)
        pg = K.pt_params(dP, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, outlet, 0.25, 0.0)
        it_ref, errs_ref = K.pt_solve(dP, dD, K.from_numpy(Rg), pg, -1.0, niter, nchk, 0.36, 1000.0, ctx=ctx)
        torch.cuda.synchronize()
        Pref, Dref = K.to_numpy(dP), K.to_numpy(dD)
        ctx.close()
        P = dims[0] * dims[1] * dims[2]
        os.environ["NS3D_CART_DEEP"] = deep
        own = bool(rng.random() < 0.5)
        mg = MultiGpu.create([0] * P, *n, "strict", dims=dims, own_streams=own)
        mg.set_temporal(depth)
        for c in mg.contexts:
            c.set_pt_depth(force)

        def cut(A, r, shrink):
            c = coords(r, dims)
            return np.asfortranarray(A[tuple(slice(c[d] * (n[d] - 2), c[d] * (n[d] - 2) + n[d] - shrink) for d in range(3))])

        Pr = [K.from_numpy(cut(Pg, r, 0)) for r in range(P)]
        D = [K.from_numpy(cut(Dg, r, 2)) for r in range(P)]
        R = [K.from_numpy(cut(Rg, r, 0)) for r in range(P)]
        p = K.pt_params(Pr[0], g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, outlet, 0.25, 0.0)
        it, errs = mg.pt_solve_slab(Pr, D, R, p, -1.0, niter, nchk, 0.36, 1000.0)
        mg.sync()
        ok = it == it_ref and len(errs) == len(errs_ref) and all(x == y or (x != x and y != y) for x, y in zip(errs, errs_ref))
        for r in range(P):
            ok = ok and np.array_equal(K.to_numpy(Pr[r]), cut(Pref, r, 0), equal_nan=True)
            ok = ok and np.array_equal(K.to_numpy(D[r]), cut(Dref, r, 2), equal_nan=True)
        pd = mg.pass_depth()
        mg.close()
        print("case %3d dims %s n %s depth %d pinned %d niter %d nchk %d %s outlet %d deep %s own_streams %d -> pass depth %d: %s" % (
            case, dims, n, depth, force, niter, nchk, np.dtype(dtype).name, outlet, deep, own, pd, "ok" if ok else "MISMATCH"), flush=True)
        if not ok:
            sys.exit(1)
    print("all %d cases equal the global solve bit for bit" % a.cases)


if __name__ == "__main__":
    main()
