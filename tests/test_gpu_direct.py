"""The option OUTSIDE PARITY of SURVEY.md §8 f4 — ns3d_poisson_direct (csrc/ns3d_direct.hip: exact diagonalisation of the box
Laplacian by six fp64 MFMA matrix products) — against its NumPy twin (oracle/direct_ref.py, itself pinned to the reference's
residual definition in tests/test_oracle.py) and against the reference's own measures: compute_res! must vanish, the PT loop
started from the solution must stop at its first check, and a whole driver run with pressure="direct" must equal the oracle
driver with the same option."""
import numpy as np
import pytest

from util import fields, geometry, rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("bc", [(0, True, 0.0), (0, True, 0.75), (0, False, 0.0), (1, False, 0.0)])
@pytest.mark.parametrize("grid", [(17, 9, 6), (24, 15, 15), (70, 35, 21), (131, 66, 37)])
def test_poisson_direct_against_the_numpy_twin_and_the_reference_residual(hip, oracle, grid, bc, dtype):
    """Every tile-edge case of k_gemm_f64 (extents below, at and above multiples of 16/32/64, K not a multiple of 4), the three
    x boundary rules, fp32 fields solved in fp64."""
    import torch
    from oracle.direct_ref import poisson_direct
    nx, ny, nz = grid
    g = geometry(*grid)
    bc_kind, owns, val = bc
    rhs = fields(nx, ny, nz, ["c"], 57, dtype)[0]
    if bc_kind == 0 and not owns:
        rhs[1:-1, 1:-1, 1:-1] -= rhs[1:-1, 1:-1, 1:-1].mean(dtype=np.float64).astype(dtype)
    ref = poisson_direct(rhs.astype(np.float64), g["rho"], g["dt"], g["dx"], g["dy"], g["dz"], bc_kind, owns, val, g["g"])
    ctx = hip.Context(0, "strict")
    dP = hip.from_numpy(fields(nx, ny, nz, ["c"], 3, dtype)[0])                      # whatever was there before
    dD = hip.from_numpy(fields(nx, ny, nz, ["i"], 4, dtype)[0])
    drhs = hip.from_numpy(rhs)
    p = hip.pt_params(dP, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], bc_kind, owns, val, g["g"])
    hip.poisson_direct(dP, dD, drhs, p, ctx=ctx)
    torch.cuda.synchronize()
    got = hip.to_numpy(dP)
    tol = 1e-11 if dtype == np.float64 else 2e-6
    assert rel_l2(got, ref) < tol, rel_l2(got, ref)
    assert not hip.to_numpy(dD).any() and np.array_equal(hip.to_numpy(drhs), rhs)
    # boundary cells = what set_bc_Pr! makes of the interior (the oracle's restatement)
    P2 = got.copy(order="F")
    oracle.set_bc_Pr(P2, bc_kind, owns, val, g["dz"], nz, g["g"], g["rho"])
    assert np.array_equal(P2, got)
    if dtype == np.float64:
        # the reference's residual (compute_res!, multi.jl:88-91) of the solution: rounding level
        res = hip.residual_max(dP, drhs, p, ctx=ctx)
        scale = g["rho"] / g["dt"] * np.abs(rhs).max() + np.abs(ref).max() / min(g["dx"], g["dy"], g["dz"]) ** 2
        assert res < 1e-10 * scale, res / scale
    # a second call on the same context reuses the plan; another grid replaces it
    hip.poisson_direct(dP, dD, drhs, p, ctx=ctx)
    torch.cuda.synchronize()
    assert np.array_equal(hip.to_numpy(dP), got)
    ctx.close()


def test_pt_loop_started_from_the_direct_solution_stops_at_its_first_check(hip, oracle):
    import torch
    nx, ny, nz = 63, 38, 38
    dx = 1.0 / nx
    g = dict(dx=dx, dy=dx, dz=dx, rho=1000.0, dt=dx, dtau=dx / np.sqrt(3.1), damp=2.0 / nx, g=0.0)
    rhs = fields(nx, ny, nz, ["c"], 58)[0] * 1e-3
    ctx = hip.Context(0, "strict")
    dP, dD, drhs = hip.zeros((nx, ny, nz)), hip.zeros((nx - 2, ny - 2, nz - 2)), hip.from_numpy(rhs)
    p = hip.pt_params(dP, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.0, 0.0)
    it_cold, _ = hip.pt_solve(hip.zeros((nx, ny, nz)), hip.zeros((nx - 2, ny - 2, nz - 2)), drhs, p, 1e-3, 20000, 37, 0.36, 1000.0, ctx=ctx)
    hip.poisson_direct(dP, dD, drhs, p, ctx=ctx)
    it, errs = hip.pt_solve(dP, dD, drhs, p, 1e-3, 20000, 37, 0.36, 1000.0, ctx=ctx)
    torch.cuda.synchronize()
    assert it == 37 and errs[0] < 1e-9 and it_cold >= 5 * it, (it, errs, it_cold)
    ctx.close()


def test_errors(hip):
    from navierstokes3d_amd import lib as L
    dP, dD, dR = hip.zeros((9, 8, 7)), hip.zeros((7, 6, 5)), hip.zeros((9, 8, 7))
    p = hip.pt_params(dP, 1000.0, 0.01, 0.01, 0.1, 0.1, 0.1, 0.1, 0, True, 0.0, 0.0, True, False)      # a z-slab rank's halo flag
    with pytest.raises(L.Ns3dError):
        hip.poisson_direct(dP, dD, dR, p)
    with pytest.raises(L.Ns3dError):
        hip.poisson_direct(dP, hip.zeros((7, 6, 4)), dR, hip.pt_params(dP, 1000.0, 0.01, 0.01, 0.1, 0.1, 0.1, 0.1))


@pytest.mark.parametrize("script", ["multi", "gpu"])
def test_driver_with_direct_pressure_equals_the_oracle_driver_with_the_same_option(hip, script):
    """Whole runs with the inner loop replaced by the direct solve, product driver against oracle driver (NumPy twin): the
    pressure agrees to solver rounding, and the velocities with it (≤ 1e-9; the direct solves differ in summation order, every
    other kernel is bit-exact)."""
    from navierstokes3d_amd.driver import run_navierstokes3D, runme
    from oracle.driver_ref import run_navierstokes3D_ref, runme_ref
    if script == "multi":
        out = run_navierstokes3D(nx=36, nt=3, mode="strict", pressure="direct", return_info=True)
        ref = run_navierstokes3D_ref(nx=36, nt=3, pressure="direct")
        info, rinfo = out[-1], ref[-1]
        pairs = list(zip(("C", "Pr", "Vx", "Vy", "Vz"), out[:5], ref[:5]))
    else:
        f, info = runme(nx=20, nt=2, mode="strict", pressure="direct")
        rf, rinfo = runme_ref(nx=20, nt=2, pressure="direct")
        pairs = [(n, hip.to_numpy(getattr(f, n)), rf[n]) for n in ("C", "Pr", "Vx", "Vy", "Vz")]
    assert info.iters == rinfo.iters == [0] * len(info.iters)
    assert all(e[0] < 1e-8 for e in info.errs)            # the reference's err measure (multi.jl:466) of the direct solution
    vnorm = max(np.sqrt(np.sum(np.asarray(b, dtype=np.float64) ** 2)) for n, a, b in pairs if n.startswith("V"))
    for n, a, b in pairs:
        assert np.isfinite(a).all() and rel_l2(a, b, vnorm if n.startswith("V") else None) < 1e-9, (n, rel_l2(a, b))
