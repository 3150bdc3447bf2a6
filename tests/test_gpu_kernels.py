"""GPU parity tests proper: every libns3d.so entry point (through the C ABI) against the CPU oracle on the
same seeded inputs.  STRICT mode must be bit-identical (tolerance 0: same IEEE operations in the same order);
FAST mode (reciprocals + FMA) must stay within 1e-12 relative L2 per kernel call (fp64) — the end-to-end
1e-6 bar of BASELINE.json is checked in test_gpu_driver.py.

Grids: 17×9×5 and 24×15×15, 5×4×3 (minimum-ish), 70×6×7 (x spans >1 wave64, ragged rows).
"""
import numpy as np
import pytest

from util import fields, geometry, rel_l2, rnd

pytestmark = pytest.mark.gpu

GRIDS = [(17, 9, 5), (24, 15, 15), (5, 4, 3), (70, 6, 7)]


def _run_both(hip, oracle, name, kinds, scalars, grid, mode, out_idx, hip_name=None, seed0=1, kwargs=None):
    import torch
    nx, ny, nz = grid
    kwargs = kwargs or {}
    host = fields(nx, ny, nz, kinds, seed0)
    ref = [a.copy(order="F") for a in host]
    getattr(oracle, name)(*ref, *scalars, **kwargs)
    ctx = hip.Context(0, mode)
    dev = [hip.from_numpy(a) for a in host]
    getattr(hip, hip_name or name)(*dev, *scalars, ctx=ctx, **kwargs)
    torch.cuda.synchronize()
    for q in out_idx:
        got = hip.to_numpy(dev[q])
        if mode == "strict":
            assert np.array_equal(got, ref[q]), "%s output %d not bit-identical (max |Δ| %g)" % (
                name, q, np.abs(got - ref[q]).max())
        else:
            assert rel_l2(got, ref[q]) < 1e-12, "%s output %d rel-L2 %g" % (name, q, rel_l2(got, ref[q]))
    # inputs must be untouched
    for q in range(len(host)):
        if q not in out_idx:
            assert np.array_equal(hip.to_numpy(dev[q]), host[q])
    ctx.close()


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("grid", GRIDS)
def test_update_tau(hip, oracle, grid, mode):
    g = geometry(*grid)
    _run_both(hip, oracle, "update_tau", ["c", "c", "c", "s", "s", "s", "vx", "vy", "vz"],
              (g["mu"], g["dx"], g["dy"], g["dz"]), grid, mode, range(6))


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("grid", GRIDS)
def test_predict_V(hip, oracle, grid, mode):
    g = geometry(*grid)
    _run_both(hip, oracle, "predict_V", ["vx", "vy", "vz", "c", "c", "c", "s", "s", "s"],
              (g["rho"], g["g"], g["dt"], g["dx"], g["dy"], g["dz"]), grid, mode, range(3))


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("grid", GRIDS + [(131, 37, 70), (64, 16, 32), (65, 17, 33), (129, 3, 3), (3, 3, 3)])
def test_predict_fused_equals_update_tau_then_predict_V(hip, oracle, grid, dtype, mode):
    """ns3d_predict_fused (k_predict_fused: the stresses evaluated from an LDS window of the velocities, never stored) against
    the oracle's update_τ! followed by predict_V! (multi.jl:449,451): complete predicted fields, bit for bit in STRICT mode
    (non-power-of-two spacings: the exact-division build, DIV_3 through the known-divisor sequence), 1e-12 / 1e-5 in FAST;
    one tile, several tiles and z-chunks with ragged edges, tiles of exactly 64×16×32 cells and one cell more; the inputs
    untouched."""
    import torch
    nx, ny, nz = grid
    g = geometry(*grid)
    host = [np.asfortranarray(a.astype(dtype)) for a in fields(nx, ny, nz, ["vx", "vy", "vz", "c", "c", "c", "s", "s", "s"], 5)]
    Vx, Vy, Vz, txx, tyy, tzz, txy, txz, tyz = [a.copy(order="F") for a in host]
    oracle.update_tau(txx, tyy, tzz, txy, txz, tyz, Vx, Vy, Vz, g["mu"], g["dx"], g["dy"], g["dz"])
    oracle.predict_V(Vx, Vy, Vz, txx, tyy, tzz, txy, txz, tyz, g["rho"], g["g"], g["dt"], g["dx"], g["dy"], g["dz"])
    ctx = hip.Context(0, mode)
    dV = [hip.from_numpy(a) for a in host[:3]]
    out = [hip.from_numpy(np.full_like(a, 777.0)) for a in host[:3]]
    hip.predict_fused(*out, *dV, g["mu"], g["rho"], g["g"], g["dt"], g["dx"], g["dy"], g["dz"], ctx=ctx)
    torch.cuda.synchronize()
    for name, got, ref, src in zip("xyz", out, (Vx, Vy, Vz), host[:3]):
        got = hip.to_numpy(got)
        if mode == "strict":
            assert np.array_equal(got, ref), "V%s not bit-identical on %r (max |Δ| %g)" % (name, grid, np.abs(got - ref).max())
        else:
            assert rel_l2(got, ref) < (1e-12 if dtype == np.float64 else 1e-5), (name, rel_l2(got, ref))
    for d, a in zip(dV, host[:3]):
        assert np.array_equal(hip.to_numpy(d), a)
    with pytest.raises(Exception):
        hip.predict_fused(dV[0], out[1], out[2], *dV, g["mu"], g["rho"], g["g"], g["dt"], g["dx"], g["dy"], g["dz"], ctx=ctx)
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_predict_fused_power_of_two_spacings_and_a_larger_grid(hip, oracle, dtype):
    """The power-of-two build (x/d = x·(1/d), v/3 through the known-divisor sequence with its guard) on values that over- and
    underflow when scaled, against the oracle; and a 260×150×131 grid (5×10×3 workgroups) against the two unfused HIP kernels."""
    import torch
    grid = (70, 9, 7)
    nx, ny, nz = grid
    big = 1e300 if dtype == np.float64 else 1e36
    tiny = 1e-305 if dtype == np.float64 else 1e-42
    ctx = hip.Context(0, "strict")
    for sp in POW2:
        g = dict(geometry(*grid)); g.update(sp)
        for scale in (1.0, big, tiny):
            host = [np.asfortranarray((a * scale).astype(dtype)) for a in fields(nx, ny, nz, ["vx", "vy", "vz", "c", "c", "c", "s", "s", "s"], 11)]
            Vx, Vy, Vz, txx, tyy, tzz, txy, txz, tyz = [a.copy(order="F") for a in host]
            with np.errstate(all="ignore"):
                oracle.update_tau(txx, tyy, tzz, txy, txz, tyz, Vx, Vy, Vz, g["mu"], g["dx"], g["dy"], g["dz"])
                oracle.predict_V(Vx, Vy, Vz, txx, tyy, tzz, txy, txz, tyz, g["rho"], g["g"], g["dt"], g["dx"], g["dy"], g["dz"])
            dV = [hip.from_numpy(a) for a in host[:3]]
            out = [hip.from_numpy(np.zeros_like(a)) for a in host[:3]]
            hip.predict_fused(*out, *dV, g["mu"], g["rho"], g["g"], g["dt"], g["dx"], g["dy"], g["dz"], ctx=ctx)
            torch.cuda.synchronize()
            for got, ref in zip(out, (Vx, Vy, Vz)):
                assert np.array_equal(hip.to_numpy(got), ref, equal_nan=True), (sp, scale)
    nx, ny, nz = 260, 150, 131
    g = geometry(nx, ny, nz)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)

    def rnd_dev(*shape):
        t = hip.zeros(shape, dtype=tdt)
        t.permute(2, 1, 0).copy_(torch.rand(shape[::-1], generator=gen, device="cuda", dtype=tdt) - 0.5)
        return t
    Vx, Vy, Vz = rnd_dev(nx + 1, ny, nz), rnd_dev(nx, ny + 1, nz), rnd_dev(nx, ny, nz + 1)
    out = [hip.zeros(tuple(V.shape), dtype=tdt) for V in (Vx, Vy, Vz)]
    hip.predict_fused(*out, Vx, Vy, Vz, g["mu"], g["rho"], g["g"], g["dt"], g["dx"], g["dy"], g["dz"], ctx=ctx)
    c, s_ = (nx, ny, nz), (nx - 1, ny - 1, nz - 1)
    tau = [hip.zeros(c, dtype=tdt) for _ in range(3)] + [hip.zeros(s_, dtype=tdt) for _ in range(3)]
    hip.update_tau(*tau, Vx, Vy, Vz, g["mu"], g["dx"], g["dy"], g["dz"], ctx=ctx)
    hip.predict_V(Vx, Vy, Vz, *tau, g["rho"], g["g"], g["dt"], g["dx"], g["dy"], g["dz"], ctx=ctx)
    torch.cuda.synchronize()
    for got, ref in zip(out, (Vx, Vy, Vz)):
        assert torch.equal(got, ref)
    ctx.close()


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("grid", GRIDS)
def test_update_divV(hip, oracle, grid, mode):
    g = geometry(*grid)
    _run_both(hip, oracle, "update_divV", ["c", "vx", "vy", "vz"], (g["dx"], g["dy"], g["dz"]), grid, mode, [0])


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("grid", GRIDS)
def test_update_dPrdtau_update_Pr_compute_res(hip, oracle, grid, mode):
    g = geometry(*grid)
    _run_both(hip, oracle, "update_dPrdtau", ["c", "i", "c"],
              (g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"]), grid, mode, [1])
    _run_both(hip, oracle, "update_Pr", ["c", "i"], (g["dtau"],), grid, mode, [0])
    _run_both(hip, oracle, "compute_res", ["i", "c", "c"], (g["rho"], g["dt"], g["dx"], g["dy"], g["dz"]), grid,
              mode, [0])


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("grid", GRIDS)
def test_correct_V(hip, oracle, grid, mode):
    g = geometry(*grid)
    _run_both(hip, oracle, "correct_V", ["vx", "vy", "vz", "c"], (g["dt"], g["rho"], g["dx"], g["dy"], g["dz"]),
              grid, mode, range(3))


@pytest.mark.parametrize("grid", GRIDS)
def test_bc_planes(hip, oracle, grid):
    nx, ny, nz = grid
    g = geometry(*grid)
    for kind in ("c", "vx", "vy", "vz"):
        for name in ("bc_x", "bc_y", "bc_z", "bc_zV"):
            _run_both(hip, oracle, name, [kind], (), grid, "strict", [0])
    _run_both(hip, oracle, "bc_xhydstatic", ["c"], (g["dz"], nz, g["g"], g["rho"]), grid, "strict", [0])
    _run_both(hip, oracle, "bc_x_Vx", ["vx"], (1.0,), grid, "strict", [0])
    _run_both(hip, oracle, "bc_x_Pr", ["c"], (0.0,), grid, "strict", [0])


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("grid", GRIDS + [(6, 2, 3), (2, 5, 2), (3, 3, 3)])
def test_set_bc_sequences(hip, oracle, grid, fused, monkeypatch):
    """Order matters on edges/corners: multi.jl x→y→z→outlet, gpu.jl y→z→x-hydrostatic.  Both forms of the library:
    the whole sequence as one gather launch (k_bc_fused, round 4) and rule by rule (NS3D_BC_FUSED=0; also what extents below 3 take)."""
    import torch
    monkeypatch.setenv("NS3D_BC_FUSED", fused)
    nx, ny, nz = grid
    g = geometry(*grid)
    ctx = hip.Context(0, "strict")
    for owns in (True, False):
        Pr = rnd(5, (nx, ny, nz)); ref = Pr.copy(order="F")
        oracle.set_bc_Pr(ref, 0, owns, 0.25)
        d = hip.from_numpy(Pr); hip.set_bc_Pr_multi(d, owns, 0.25, ctx=ctx)
        assert np.array_equal(hip.to_numpy(d), ref)
        V = fields(nx, ny, nz, ["vx", "vy", "vz"], 11); ref = [a.copy(order="F") for a in V]
        oracle.set_bc_Vel(*ref, 0, owns, 1.5)
        d = [hip.from_numpy(a) for a in V]; hip.set_bc_Vel_multi(*d, owns, 1.5, ctx=ctx)
        for a, b in zip(d, ref):
            assert np.array_equal(hip.to_numpy(a), b)
    Pr = rnd(6, (nx, ny, nz)); ref = Pr.copy(order="F")
    oracle.set_bc_Pr(ref, 1, False, 0.0, g["dz"], nz, g["g"], g["rho"])
    d = hip.from_numpy(Pr); hip.set_bc_Pr_gpu(d, g["dz"], nz, g["g"], g["rho"], ctx=ctx)
    assert np.array_equal(hip.to_numpy(d), ref)
    V = fields(nx, ny, nz, ["vx", "vy", "vz"], 12); ref = [a.copy(order="F") for a in V]
    oracle.set_bc_Vel(*ref, 1)
    d = [hip.from_numpy(a) for a in V]; hip.set_bc_Vel_gpu(*d, ctx=ctx)
    for a, b in zip(d, ref):
        assert np.array_equal(hip.to_numpy(a), b)
    torch.cuda.synchronize(); ctx.close()


@pytest.mark.parametrize("grid", [(24, 15, 15), (70, 6, 7), (63, 38, 38)])
def test_set_cylinder_both_forms(hip, oracle, grid):
    nx, ny, nz = grid
    dx, dy, dz = 1.0 / nx, 0.6 / ny, 0.6 / nz
    for beta in (0.0, 0.3):
        sc = (0.0121, 0.0064, -0.1, 0.02, np.sin(beta), np.cos(beta))
        glob = sc + (-(1 - dx) / 2, -(0.6 - dy) / 2, -(0.6 - dz) / 2, 1.0, 0.6, 0.6, dx, dy, dz)
        loc = sc + (1.0, 0.6, 0.6, dx, dy, dz)
        _run_both(hip, oracle, "set_cylinder", ["c", "vx", "vy", "vz"], glob, grid, "strict", range(4))
        _run_both(hip, oracle, "set_cylinder_local", ["c", "vx", "vy", "vz"], loc, grid, "strict", range(4),
                  hip_name="set_cylinder")


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("faithful", [True, False])
@pytest.mark.parametrize("cfl", [0.3, 1.0, 2.7])
@pytest.mark.parametrize("grid", [(17, 9, 5), (24, 15, 15), (70, 6, 7)])
def test_advect(hip, oracle, grid, cfl, faithful, mode):
    """Data-dependent gather; cfl>1 exercises departure points more than one cell away and the clamps."""
    import torch
    nx, ny, nz = grid
    g = geometry(*grid)
    Vx_o, Vy_o, Vz_o, C_o = fields(nx, ny, nz, ["vx", "vy", "vz", "c"], 21)
    dt = cfl * g["dx"]
    outs = [a.copy(order="F") for a in fields(nx, ny, nz, ["vx", "vy", "vz", "c"], 31)]
    ref = [a.copy(order="F") for a in outs]
    oracle.advect(ref[0], Vx_o, ref[1], Vy_o, ref[2], Vz_o, ref[3], C_o, dt, g["dx"], g["dy"], g["dz"], faithful)
    ctx = hip.Context(0, mode)
    d = [hip.from_numpy(a) for a in outs]
    do = [hip.from_numpy(a) for a in (Vx_o, Vy_o, Vz_o, C_o)]
    hip.advect(d[0], do[0], d[1], do[1], d[2], do[2], d[3], do[3], dt, g["dx"], g["dy"], g["dz"], faithful, ctx=ctx)
    torch.cuda.synchronize()
    for q in range(4):
        got = hip.to_numpy(d[q])
        if mode == "strict":
            assert np.array_equal(got, ref[q]), "advect output %d" % q
        else:
            # FMA changes δ by an ulp; a departure index can flip only when δ is within an ulp of an integer,
            # which the seeded inputs do not hit
            assert rel_l2(got, ref[q]) < 1e-12
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("grid,cfl", [((150, 21, 70), 0.9), ((150, 21, 70), 2.7), ((64, 8, 33), 0.5), ((63, 7, 31), 1.0),
                                       ((131, 19, 38), 1.6), ((3, 3, 3), 0.8)])
def test_advect_windowed_tiles(hip, oracle, grid, cfl, dtype):
    """The LDS-windowed advect! (64×8-column workgroups marching in z, 67×11×6-plane ring of the four old fields): grids of
    several tiles in x, y and z-chunks, tile-aligned and ragged extents, departure points inside the window (cfl < 1), on
    its edge and far outside it (cfl 2.7: those lanes take the global gather) — bit-identical to the oracle either way,
    both advection modes."""
    import torch
    nx, ny, nz = grid
    g = geometry(*grid)
    Vx_o, Vy_o, Vz_o, C_o = fields(nx, ny, nz, ["vx", "vy", "vz", "c"], 21, dtype)
    dt = cfl * min(g["dx"], g["dy"], g["dz"])
    ctx = hip.Context(0, "strict")
    do = [hip.from_numpy(a) for a in (Vx_o, Vy_o, Vz_o, C_o)]
    for faithful in (True, False):
        outs = fields(nx, ny, nz, ["vx", "vy", "vz", "c"], 31, dtype)
        ref = [a.copy(order="F") for a in outs]
        oracle.advect(ref[0], Vx_o, ref[1], Vy_o, ref[2], Vz_o, ref[3], C_o, dt, g["dx"], g["dy"], g["dz"], faithful)
        d = [hip.from_numpy(a) for a in outs]
        hip.advect(d[0], do[0], d[1], do[1], d[2], do[2], d[3], do[3], dt, g["dx"], g["dy"], g["dz"], faithful, ctx=ctx)
        torch.cuda.synchronize()
        for q in range(4):
            assert np.array_equal(hip.to_numpy(d[q]), ref[q]), "advect output %d (faithful=%s)" % (q, faithful)
        for a, b in zip(do, (Vx_o, Vy_o, Vz_o, C_o)):
            assert np.array_equal(hip.to_numpy(a), b)
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("grid,cfl", [((150, 21, 70), 0.9), ((150, 21, 70), 2.7), ((64, 8, 33), 0.5), ((63, 7, 31), 1.0), ((3, 3, 3), 0.8)])
def test_copy_advect_equals_four_copies_and_advect(hip, oracle, grid, cfl, dtype, monkeypatch):
    """ns3d_copy_advect = {X_o .= X; advect!} (multi.jl:475-476) in one pass with the buffers' roles swapped afterwards: the
    outputs, pre-filled with garbage, must come back COMPLETE — the advected entries and, written through, every entry advect!
    leaves alone (Vx[1,:,:], Vx[end,:,:], Vy[:,1,1], Vy[:,end,:], all of Vz in faithful mode …) — and equal the oracle's
    copies + advect bit for bit; windowed and global-gather kernels, both advection modes, Vz passed as its own output where
    that is allowed."""
    import torch
    nx, ny, nz = grid
    g = geometry(*grid)
    cur = fields(nx, ny, nz, ["vx", "vy", "vz", "c"], 21, dtype)
    dt = cfl * min(g["dx"], g["dy"], g["dz"])
    for windowed in (True, False):
        ctx = hip.Context(0, "strict")
        dcur = [hip.from_numpy(a) for a in cur]
        for faithful in (True, False):
            old = [a.copy(order="F") for a in cur]                              # X_o .= X
            ref = [a.copy(order="F") for a in cur]
            oracle.advect(ref[0], old[0], ref[1], old[1], ref[2], old[2], ref[3], old[3], dt, g["dx"], g["dy"], g["dz"], faithful)
            new = [hip.from_numpy(np.full_like(a, 777.0)) for a in cur]          # garbage: nothing may survive
            vz_out = dcur[2] if faithful else new[2]
            if not windowed:
                hip.copy_advect(new[0], dcur[0], new[1], dcur[1], new[2], dcur[2], new[3], dcur[3], dt, g["dx"], g["dy"], g["dz"],
                                faithful, ctx=ctx)                                # Vz_new a buffer of its own in both modes
                vz_out = new[2]
            else:
                hip.copy_advect(new[0], dcur[0], new[1], dcur[1], vz_out, dcur[2], new[3], dcur[3], dt, g["dx"], g["dy"], g["dz"],
                                faithful, ctx=ctx)
            torch.cuda.synchronize()
            got = [new[0], new[1], vz_out, new[3]]
            for q in range(4):
                assert np.array_equal(hip.to_numpy(got[q]), ref[q]), "output %d (faithful=%s, windowed=%s)" % (q, faithful, windowed)
            for a, b in zip(dcur, cur):
                assert np.array_equal(hip.to_numpy(a), b)                        # the current fields are read only
        ctx.close()
        if windowed:
            monkeypatch.setenv("NS3D_ADVECT_GLOBAL", "1")                        # read once per process: see below
    from navierstokes3d_amd import lib as L
    with pytest.raises(L.Ns3dError):
        hip.copy_advect(dcur[0], dcur[0], new[1], dcur[1], new[2], dcur[2], new[3], dcur[3], dt, g["dx"], g["dy"], g["dz"], True)
    with pytest.raises(L.Ns3dError):
        hip.copy_advect(new[0], dcur[0], new[1], dcur[1], dcur[2], dcur[2], new[3], dcur[3], dt, g["dx"], g["dy"], g["dz"], False)


def test_advect_integer_cfl_edge(hip, oracle):
    """Positive integer δ: weight 1 with base floor(i−δ) (SURVEY App. A backtrack! edge case)."""
    import torch
    nx, ny, nz = 17, 9, 5
    dx = dy = dz = 0.125  # exact in binary so dt*v/dx is an exact integer
    Vx_o = np.asfortranarray(np.full((nx + 1, ny, nz), 2.0)); Vy_o = np.asfortranarray(np.full((nx, ny + 1, nz), -1.0))
    Vz_o = np.asfortranarray(np.zeros((nx, ny, nz + 1))); C_o = rnd(3, (nx, ny, nz))
    outs = [np.asfortranarray(np.zeros_like(a)) for a in (Vx_o, Vy_o, Vz_o, C_o)]
    ref = [a.copy(order="F") for a in outs]
    oracle.advect(ref[0], Vx_o, ref[1], Vy_o, ref[2], Vz_o, ref[3], C_o, dx, dx, dy, dz, True)
    ctx = hip.Context(0, "strict")
    d = [hip.from_numpy(a) for a in outs]; do = [hip.from_numpy(a) for a in (Vx_o, Vy_o, Vz_o, C_o)]
    hip.advect(d[0], do[0], d[1], do[1], d[2], do[2], d[3], do[3], dx, dx, dy, dz, True, ctx=ctx)
    torch.cuda.synchronize()
    for q in range(4):
        assert np.array_equal(hip.to_numpy(d[q]), ref[q])
    ctx.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4097, 250047])
def test_max_abs(hip, oracle, n):
    import torch
    a = rnd(n, (n, 1, 1), lo=-3.0, hi=2.0)
    d = hip.from_numpy(a)
    assert hip.max_abs(d) == oracle.max_abs(a) == float(np.abs(a).max())
    if n > 1:
        a[n // 2, 0, 0] = np.nan   # Julia's maximum propagates NaN (App. B7); HIP fmax would drop it
        assert np.isnan(hip.max_abs(hip.from_numpy(a))) and np.isnan(oracle.max_abs(a))
        a[n // 2, 0, 0] = -np.inf
        assert hip.max_abs(hip.from_numpy(a)) == np.inf
    torch.cuda.synchronize()


def test_copy_and_errors(hip):
    import torch
    from navierstokes3d_amd import lib as L
    a = rnd(1, (9, 5, 4)); d = hip.from_numpy(a); e = hip.zeros((9, 5, 4))
    hip.copy(e, d)
    assert np.array_equal(hip.to_numpy(e), a)
    with pytest.raises(L.Ns3dError):      # CPU tensors are refused: there is no CPU path
        hip.bc_x(torch.zeros(4, 4, 4, dtype=torch.float64))
    with pytest.raises(L.Ns3dError):      # row-major tensor is not the reference layout
        hip.bc_x(torch.zeros(4, 5, 6, dtype=torch.float64, device="cuda"))
    with pytest.raises(L.Ns3dError):      # shape mismatch
        hip.update_Pr(hip.zeros((8, 8, 8)), hip.zeros((5, 6, 6)), 0.1)
    with pytest.raises(L.Ns3dError):      # grid too small for the stencil: status from the C ABI
        hip.update_Pr(hip.zeros((2, 8, 8)), hip.zeros((0, 6, 6)), 0.1)


def test_exact_division_by_known_divisor(hip):
    """STRICT mode evaluates x/dx as q=RN(x·r), e=x−q·dx (FMA), RN(q+e·r), r=RN(1/dx) — the correctly rounded quotient
    (Markstein).  The device self-test compares it bit for bit with the plain IEEE division on pseudo-random dividends
    (random significands over 120 binades plus quotients planted next to representable numbers and midpoints)."""
    import numpy as np
    ctx = hip.Context(0, "strict")
    rng = np.random.Generator(np.random.MT19937(2024))
    divisors = [1.0 / 63, 0.6 / 38, 1.0 / 255, 0.6 / 153, 1.0 / 512, 0.7 / 5, 1.0 / 17, 0.6 / 9, 3.0, 0.1, 1.0 / 3,
                float(np.nextafter(2.0, 0.0)) / 4.0 * 0 + 0.0078125]
    divisors += [float(x) for x in np.exp(rng.uniform(np.log(1e-5), np.log(50.0), 60))]
    total = 0
    for q, d in enumerate(divisors):
        n = 1 << 24
        assert ctx.selftest_exact_div(d, n, seed=q + 1) == 0, "f64 divisor %r" % d
        assert ctx.selftest_exact_div(d, 1 << 22, seed=q + 1, dtype=__import__("torch").float32) == 0, "f32 divisor %r" % d
        total += n
    # a long run on the spacings of the reference configurations
    for d in (1.0 / 63, 0.6 / 38, 1.0 / 255, 0.6 / 153):
        assert ctx.selftest_exact_div(d, 1 << 30, seed=99) == 0
    ctx.close()


def test_ieee_div_flag_gives_identical_results(hip, oracle):
    """NS3D_IEEE_DIV (plain division sequence) and the default STRICT path produce the same bits."""
    import torch
    nx, ny, nz = 24, 15, 15
    g = geometry(nx, ny, nz)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 77)
    outs = []
    for ieee in (False, True):
        ctx = hip.Context(0, "strict", ieee_div=ieee)
        dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
        p = hip.pt_params(dPr, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"])
        hip.pt_iterate(dPr, dd, hip.from_numpy(rhs), p, 9, ctx=ctx)
        torch.cuda.synchronize()
        outs.append((hip.to_numpy(dPr), hip.to_numpy(dd)))
        ctx.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_c_abi_status_codes(hip):
    """The boundary never aborts: bad arguments come back as NS3D_ERR_ARG with a message, straight from the C entry points
    (raw ctypes, no Python-side checks in between)."""
    import ctypes as C
    import torch
    from navierstokes3d_amd import lib as L
    lib = L.load()
    ctx = hip.Context(0, "strict")
    h = ctx.handle
    nx, ny, nz = 12, 9, 7
    P, Q = hip.zeros((nx, ny, nz)), hip.zeros((nx, ny, nz))
    D, E = hip.zeros((nx - 2, ny - 2, nz - 2)), hip.zeros((nx - 2, ny - 2, nz - 2))
    R = hip.zeros((nx, ny, nz))
    ptr = lambda t: C.c_void_p(t.data_ptr())
    p = hip.pt_params(P, 1000.0, 0.01, 0.01, 0.1, 0.1, 0.1, 0.1, 0, True, 0.0, 0.0)
    ERR = 1
    assert lib.ns3d_update_Pr_f64(None, ptr(P), ptr(D), 0.1, nx, ny, nz) == ERR and b"null context" in lib.ns3d_last_error()
    assert lib.ns3d_update_Pr_f64(h, None, ptr(D), 0.1, nx, ny, nz) == ERR and b"null field pointer" in lib.ns3d_last_error()
    assert lib.ns3d_update_Pr_f64(h, ptr(P), ptr(D), 0.1, 2, ny, nz) == ERR and b"too small" in lib.ns3d_last_error()
    assert lib.ns3d_pt_sweep2_f64(h, ptr(P), ptr(P), ptr(D), ptr(E), ptr(R), C.byref(p), 1, nz - 1) == ERR
    assert b"must differ" in lib.ns3d_last_error()
    assert lib.ns3d_pt_sweep2_f64(h, ptr(P), ptr(Q), ptr(D), ptr(D), ptr(R), C.byref(p), 1, nz - 1) == ERR
    assert lib.ns3d_pt_sweep2_f64(h, ptr(P), ptr(Q), ptr(D), ptr(E), ptr(R), C.byref(p), 0, nz - 1) == ERR
    assert b"plane range" in lib.ns3d_last_error()
    assert lib.ns3d_pt_sweep_f64(h, ptr(P), ptr(Q), ptr(D), ptr(R), C.byref(p), 1, nz) == ERR
    assert lib.ns3d_pt_iterate_f64(h, ptr(P), ptr(D), ptr(R), None, 3) == ERR and b"null params" in lib.ns3d_last_error()
    bad = hip.pt_params(P, 1000.0, 0.01, 0.01, 0.1, 0.1, 0.1, 0.1, 0, True, 0.0, 0.0)
    bad.bc_kind = 7
    assert lib.ns3d_pt_iterate_f64(h, ptr(P), ptr(D), ptr(R), C.byref(bad), 3) == ERR and b"bc_kind" in lib.ns3d_last_error()
    halo = hip.pt_params(P, 1000.0, 0.01, 0.01, 0.1, 0.1, 0.1, 0.1, 1, False, 0.0, 9.81, True, False)
    assert lib.ns3d_pt_iterate_f64(h, ptr(P), ptr(D), ptr(R), C.byref(halo), 3) == ERR   # gpu.jl set has no z halos
    assert lib.ns3d_set_pt_variant(h, -5) == ERR and lib.ns3d_set_pt2_variant(h, 123456) == ERR
    assert lib.ns3d_set_autotune(None, 1) == ERR and lib.ns3d_last_pt2_variant(None) == -1
    # … and a good call still works afterwards
    assert lib.ns3d_pt_iterate_f64(h, ptr(P), ptr(D), ptr(R), C.byref(p), 3) == 0
    torch.cuda.synchronize()
    ctx.close()


# ---- the once-per-step kernels at BASELINE's full size -----------------------------------------------------------
def test_full_size_512_cubed_substeps_against_oracle_subslab(hip, oracle):
    """512³ (BASELINE configs[2]): every once-per-step stencil kernel runs on the full grid; all of them are local in z
    (radius ≤ 1 cell, advect! with |δ| < 1), so the oracle can check a 16-cell-thick sub-slab cut out of the middle —
    all x/y faces included — bit for bit, two planes in from each cut."""
    import torch
    from util import SHAPES
    n, a, b, m = 512, 250, 266, 2
    g = geometry(n, n, n)
    gen = torch.Generator(device="cuda"); gen.manual_seed(512512)
    ctx = hip.Context(0, "strict")

    def rnd_dev(kind, scale=1.0):
        t = hip.zeros(SHAPES[kind](n, n, n))
        t.permute(2, 1, 0).uniform_(-scale, scale, generator=gen)
        return t

    def zlen(kind):                       # z extent of the sub-slab's array of that kind
        return SHAPES[kind](n, n, b - a)[2]

    def cut(t, kind):                     # host copy of the sub-slab part of a full-size device array
        return np.asfortranarray(hip.to_numpy(t[:, :, a:a + zlen(kind)]))

    def check(name, kinds, scalars, out_idx, scales=None, kwargs=None):
        kwargs = kwargs or {}
        dev = [rnd_dev(k, (scales or {}).get(q, 1.0)) for q, k in enumerate(kinds)]
        ref = [cut(t, k) for t, k in zip(dev, kinds)]
        getattr(oracle, name)(*ref, *scalars, **kwargs)
        getattr(hip, name)(*dev, *scalars, ctx=ctx, **kwargs)
        torch.cuda.synchronize()
        for q in out_idx:
            got = cut(dev[q], kinds[q])
            assert np.array_equal(got[:, :, m:-m], ref[q][:, :, m:-m]), "%s output %d differs at 512^3" % (name, q)
        del dev

    check("update_tau", ["c", "c", "c", "s", "s", "s", "vx", "vy", "vz"], (g["mu"], g["dx"], g["dy"], g["dz"]), range(6))
    check("predict_V", ["vx", "vy", "vz", "c", "c", "c", "s", "s", "s"],
          (g["rho"], g["g"], g["dt"], g["dx"], g["dy"], g["dz"]), range(3))
    check("update_divV", ["c", "vx", "vy", "vz"], (g["dx"], g["dy"], g["dz"]), [0])
    check("correct_V", ["vx", "vy", "vz", "c"], (g["dt"], g["rho"], g["dx"], g["dy"], g["dz"]), range(3))
    check("update_dPrdtau", ["c", "i", "c"], (g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"]), [1])
    check("compute_res", ["i", "c", "c"], (g["rho"], g["dt"], g["dx"], g["dy"], g["dz"]), [0])
    # advect!(Vx,Vx_o,Vy,Vy_o,Vz,Vz_o,C,C_o,…): old velocities scaled so that |δ| = |v·dt/dx| < 0.4 cells
    vs = 0.4 * g["dx"] / g["dt"]
    check("advect", ["vx", "vx", "vy", "vy", "vz", "vz", "c", "c"], (g["dt"], g["dx"], g["dy"], g["dz"], True),
          [0, 2, 4, 6], scales={1: vs, 3: vs, 5: vs})
    ctx.close()


POW2 = [dict(dx=1.0 / 64, dy=1.0 / 32, dz=1.0 / 128), dict(dx=0.5, dy=1.0, dz=0.25), dict(dx=2.0 ** -20, dy=2.0 ** -3, dz=2.0 ** -30),
        dict(dx=0.5, dy=2.0, dz=1.0)]     # the last one has a spacing > 1: not the power-of-two build, same bits all the same


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("sp", POW2)
def test_power_of_two_spacings_strict_bitexact(hip, oracle, sp, dtype):
    """Grids whose spacings are powers of two 2⁻³⁰…1 (512³ with lx = 1: dx = 2⁻⁹) run STRICT mode in the `ns3d_strictp` build,
    where x/dx is x·(1/dx) and x/dx/dx is x·(1/dx²) — the same real numbers, hence the same roundings, always (overflow and
    subnormal operands included).  Every kernel with a division against the oracle's plain divisions, bit for bit, including values
    that under/overflow when scaled."""
    import torch
    grid = (70, 9, 7)
    nx, ny, nz = grid
    g = dict(geometry(*grid)); g.update(sp)
    ctx = hip.Context(0, "strict")

    def both(name, kinds, scalars, out_idx, seed0, scale=1.0, kwargs=None):
        kwargs = kwargs or {}
        host = [np.asfortranarray((a * scale).astype(dtype)) for a in fields(nx, ny, nz, kinds, seed0)]
        ref = [a.copy(order="F") for a in host]
        getattr(oracle, name)(*ref, *scalars, **kwargs)
        dev = [hip.from_numpy(a) for a in host]
        getattr(hip, name)(*dev, *scalars, ctx=ctx, **kwargs)
        torch.cuda.synchronize()
        for q in out_idx:
            got = hip.to_numpy(dev[q])
            assert got.tobytes(order="A") == ref[q].tobytes(order="A") or np.array_equal(got, ref[q], equal_nan=True), \
                "%s output %d differs on spacings %r (scale %g)" % (name, q, sp, scale)

    big = 1e300 if dtype == np.float64 else 1e36
    tiny = 1e-305 if dtype == np.float64 else 1e-42
    for scale in (1.0, big, tiny):
        both("update_tau", ["c", "c", "c", "s", "s", "s", "vx", "vy", "vz"], (g["mu"], g["dx"], g["dy"], g["dz"]), range(6), 1, scale)
        both("predict_V", ["vx", "vy", "vz", "c", "c", "c", "s", "s", "s"], (g["rho"], g["g"], g["dt"], g["dx"], g["dy"], g["dz"]),
             range(3), 2, scale)
        both("update_divV", ["c", "vx", "vy", "vz"], (g["dx"], g["dy"], g["dz"]), [0], 3, scale)
        both("update_dPrdtau", ["c", "i", "c"], (g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"]), [1], 4, scale)
        both("compute_res", ["i", "c", "c"], (g["rho"], g["dt"], g["dx"], g["dy"], g["dz"]), [0], 5, scale)
        both("correct_V", ["vx", "vy", "vz", "c"], (g["dt"], g["rho"], g["dx"], g["dy"], g["dz"]), range(3), 6, scale)
    cfl = 0.7 * min(g["dx"], g["dy"], g["dz"])
    both("advect", ["vx", "vx", "vy", "vy", "vz", "vz", "c", "c"], (cfl, g["dx"], g["dy"], g["dz"]), [0, 2, 6], 7,
         kwargs=dict(faithful=True))
    ctx.close()
