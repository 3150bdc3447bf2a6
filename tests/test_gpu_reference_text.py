"""The HIP path against the reference's own SOURCE TEXT, with nothing hand-written in between: tests/golden/jl_eval_*.npz hold
what oracle/jl_eval.py computed by executing the kernel definitions and the two drivers of scripts/NavierStokes3D_multi_gpu.jl
and scripts/NavierStokes3D_gpu.jl token by token (see that file for what it assumes).  Here the SAME seeded inputs go through
the C ABI in STRICT mode and every output array must equal the stored one bit for bit — kernels first, then whole runs of
run_navierstokes3D / runme (fused PT loop, temporally blocked passes, HIP-graph replay: whatever the product path picks)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_JL2K = {"update_τ!": "update_tau", "predict_V!": "predict_V", "update_∇V!": "update_divV", "update_dPrdτ!": "update_dPrdtau",
         "update_Pr!": "update_Pr", "compute_res!": "compute_res", "correct_V!": "correct_V", "bc_x!": "bc_x", "bc_y!": "bc_y",
         "bc_z!": "bc_z", "bc_x_Vx!": "bc_x_Vx", "bc_x_Pr!": "bc_x_Pr", "bc_zV!": "bc_zV", "bc_xhydstatic!": "bc_xhydstatic"}


def _dev(hip, vals):
    return [hip.from_numpy(v) if isinstance(v, np.ndarray) and v.ndim == 3 else v for v in vals]


def test_kernels_equal_the_reference_kernel_text(hip):
    import torch
    from oracle import jl_eval
    gold = np.load(os.path.join(GOLD, "jl_eval_kernels.npz"))
    ctx = hip.Context(0, "strict")
    seen = 0
    for script in jl_eval.SCRIPTS:
        for grid in jl_eval.GRIDS:
            for q, (name, vals) in enumerate(jl_eval.cases(script, grid)):
                dv = _dev(hip, list(vals.values()))
                getattr(hip, _JL2K[name])(*dv, ctx=ctx)
                torch.cuda.synchronize()
                for a, t in zip(vals, dv):
                    if isinstance(vals[a], np.ndarray) and vals[a].ndim == 3:
                        key = "%s/%dx%dx%d/%02d/%s/%s" % (script, grid[0], grid[1], grid[2], q, name, a)
                        assert np.array_equal(hip.to_numpy(t), gold[key]), key
                        seen += 1
            for q, (name, args) in enumerate(jl_eval.cases2(script, grid)):
                dv = _dev(hip, args)
                if name == "set_cylinder!":
                    hip.set_cylinder(*dv, ctx=ctx)
                elif name == "advect!":
                    hip.advect(*dv, True, ctx=ctx)
                elif name == "set_bc_Vel!" and script == "multi":
                    hip.set_bc_Vel_multi(dv[0], dv[1], dv[2], args[3] == -args[4] / 2, args[5], ctx=ctx)      # multi.jl:164
                elif name == "set_bc_Vel!":
                    hip.set_bc_Vel_gpu(dv[0], dv[1], dv[2], ctx=ctx)
                elif name == "set_bc_Pr!" and script == "multi":
                    hip.set_bc_Pr_multi(dv[0], args[1] == args[2] / 2, args[3], ctx=ctx)                     # multi.jl:179
                else:
                    hip.set_bc_Pr_gpu(dv[0], args[1], args[2], args[3], args[4], ctx=ctx)
                torch.cuda.synchronize()
                for j, t in enumerate(dv):
                    if isinstance(args[j], np.ndarray) and args[j].ndim == 3:
                        key = "%s/%dx%dx%d/p2_%02d/%s/%d" % (script, grid[0], grid[1], grid[2], q, name, j)
                        assert np.array_equal(hip.to_numpy(t), gold[key]), key
                        seen += 1
    assert seen == len(gold.files)
    ctx.close()


_F2P = {"Pr": "Pr", "dPrdτ": "dPrdtau", "C": "C", "C_o": "C_o", "τxx": "txx", "τyy": "tyy", "τzz": "tzz", "τxy": "txy", "τxz": "txz",
        "τyz": "tyz", "Vx": "Vx", "Vy": "Vy", "Vz": "Vz", "Vx_o": "Vx_o", "Vy_o": "Vy_o", "Vz_o": "Vz_o", "∇V": "divV"}


@pytest.mark.parametrize("fused", [True, False])
def test_drivers_equal_the_reference_drivers_evaluated_from_their_text(hip, fused):
    """run_navierstokes3D / runme on the GPU (fused PT path and the literal kernel-by-kernel loop) against the runs of the
    scripts' text: iterations per step, residual histories and the 17 arrays the product keeps (Rp is not materialised by the
    fused residual) after the last step, bit for bit."""
    from oracle import jl_eval
    from navierstokes3d_amd.driver import run_navierstokes3D, runme
    gold = np.load(os.path.join(GOLD, "jl_eval_drivers.npz"))
    for script, nx, nt, cap in jl_eval.DRIVER_CASES:
        pre = "%s/nx%d_nt%d/" % (script, nx, nt)
        if script == "multi":
            out = run_navierstokes3D(nx=nx, nt=nt, mode="strict", fused=fused, niter_cap=cap, return_info=True)
            info, f = out[-1], out[-1].fields
        else:
            f, info = runme(nx=nx, nt=nt, mode="strict", fused=fused, niter_cap=cap)
        assert info.iters == gold[pre + "iters"].tolist(), pre
        assert [e for es in info.errs for e in es] == gold[pre + "errs"].tolist(), pre
        for jl, name in _F2P.items():
            assert np.array_equal(hip.to_numpy(getattr(f, name)), gold[pre + "field/" + jl], equal_nan=True), pre + jl


@pytest.mark.parametrize("fused", [True, False])
def test_config_a_equals_the_reference_text(hip, fused):
    """run_navierstokes3D(nx=63, nt=3) — BASELINE configs[0]'s grid 63×38×38 — on the GPU against the digests of the same run
    evaluated from multi.jl's text (tests/golden/jl_eval_config_a.json): 37, 259, 296 PT iterations, residual histories to the
    bit, sha256 of every array the product keeps."""
    import json
    from oracle import jl_eval
    from navierstokes3d_amd.driver import run_navierstokes3D
    g = json.load(open(os.path.join(GOLD, "jl_eval_config_a.json"), encoding="utf-8"))
    out = run_navierstokes3D(nx=g["nx"], nt=g["nt"], mode="strict", fused=fused, return_info=True)
    info, f = out[-1], out[-1].fields
    assert info.iters == g["iters"]
    assert [[float(e).hex() for e in es] for es in info.errs] == g["errs_hex"]
    for jl, name in _F2P.items():
        a = hip.to_numpy(getattr(f, name))
        assert list(a.shape) == g["shape"][jl] and jl_eval.field_digest(a) == g["sha256"][jl], jl
