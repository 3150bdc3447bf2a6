"""GPU parity of the fused pseudo-transient path (the hot loop) against the oracle's UNFUSED reference sequence
update_dPrdτ! → update_Pr! → set_bc_Pr! (multi.jl:459-463 / gpu.jl:127-129).

STRICT: bit-identical after any number of sweeps, for every sweep variant, both boundary sets, with and without
the outlet plane, on ragged grids.  FAST: ≤1e-9 relative L2 after 40 sweeps (fp64).
"""
import numpy as np
import pytest

from util import fields, geometry, rel_l2, rnd

pytestmark = pytest.mark.gpu

GRIDS = [(17, 9, 5), (24, 15, 15), (5, 4, 3), (3, 3, 3), (70, 6, 7), (131, 21, 35), (63, 38, 38)]
VARIANTS = [0, 100, 200, 700, 2000, 2200, 2600, 2700, 103, 2207, 2064]


def _oracle_iters(oracle, Pr, d, rhs, g, n, bc_kind, owns_outlet, outlet_val):
    nx, ny, nz = Pr.shape
    for _ in range(n):
        oracle.update_dPrdtau(Pr, d, rhs, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"])
        oracle.update_Pr(Pr, d, g["dtau"])
        oracle.set_bc_Pr(Pr, bc_kind, owns_outlet, outlet_val, g["dz"], nz, g["g"], g["rho"])


def _params(hip, dPr, g, bc_kind, owns_outlet, outlet_val, zlo=False, zhi=False):
    return hip.pt_params(dPr, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], bc_kind,
                         owns_outlet, outlet_val, g["g"], zlo, zhi)


@pytest.mark.parametrize("bc", [(0, True, 0.0), (0, False, 0.0), (0, True, 0.75), (1, False, 0.0)])
@pytest.mark.parametrize("grid", GRIDS)
def test_pt_iterate_strict_bitexact(hip, oracle, grid, bc):
    import torch
    nx, ny, nz = grid
    g = geometry(*grid)
    bc_kind, owns, val = bc
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 41)
    ctx = hip.Context(0, "strict")
    for variant in VARIANTS:
        ctx.set_pt_variant(variant)
        for n in (1, 2, 5):
            Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
            _oracle_iters(oracle, Pr, d, rhs, g, n, bc_kind, owns, val)
            dPr, dd, drhs = hip.from_numpy(Pr0), hip.from_numpy(d0), hip.from_numpy(rhs)
            hip.pt_iterate(dPr, dd, drhs, _params(hip, dPr, g, bc_kind, owns, val), n, ctx=ctx)
            torch.cuda.synchronize()
            assert np.array_equal(hip.to_numpy(dd), d), "dPrdτ differs: variant %d n %d" % (variant, n)
            assert np.array_equal(hip.to_numpy(dPr), Pr), "Pr differs: variant %d n %d" % (variant, n)
            assert np.array_equal(hip.to_numpy(drhs), rhs)
    ctx.close()


@pytest.mark.parametrize("grid", [(24, 15, 15), (70, 6, 7), (63, 38, 38)])
def test_pt_iterate_fast_within_tolerance(hip, oracle, grid):
    import torch
    nx, ny, nz = grid
    g = geometry(*grid)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 43)
    Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
    _oracle_iters(oracle, Pr, d, rhs, g, 40, 0, True, 0.0)
    ctx = hip.Context(0, "fast")
    dPr, dd, drhs = hip.from_numpy(Pr0), hip.from_numpy(d0), hip.from_numpy(rhs)
    hip.pt_iterate(dPr, dd, drhs, _params(hip, dPr, g, 0, True, 0.0), 40, ctx=ctx)
    torch.cuda.synchronize()
    assert rel_l2(hip.to_numpy(dPr), Pr) < 1e-9 and rel_l2(hip.to_numpy(dd), d) < 1e-9
    ctx.close()


@pytest.mark.parametrize("grid", [(24, 15, 15), (70, 6, 9)])
def test_pt_sweep_plane_ranges_and_halo_planes(hip, oracle, grid):
    """Split sweeps (boundary planes first, interior later — the z-slab overlap schedule) equal one full sweep, and
    halo planes of the output buffer are never written when z_*_is_halo."""
    import torch
    nx, ny, nz = grid
    g = geometry(*grid)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 47)
    ctx = hip.Context(0, "strict")
    for zlo, zhi in ((False, False), (True, False), (False, True), (True, True)):
        p = _params(hip, hip.from_numpy(Pr0), g, 0, True, 0.0, zlo, zhi)
        full_in, full_out, dfull = hip.from_numpy(Pr0), hip.from_numpy(np.full_like(Pr0, 777.0)), hip.from_numpy(d0)
        hip.pt_sweep(full_in, full_out, dfull, hip.from_numpy(rhs), p, 1, nz - 1, ctx=ctx)
        sp_in, sp_out, dsp = hip.from_numpy(Pr0), hip.from_numpy(np.full_like(Pr0, 777.0)), hip.from_numpy(d0)
        hip.pt_sweep(sp_in, sp_out, dsp, hip.from_numpy(rhs), p, 1, 2, ctx=ctx)
        hip.pt_sweep(sp_in, sp_out, dsp, hip.from_numpy(rhs), p, nz - 2, nz - 1, ctx=ctx)
        hip.pt_sweep(sp_in, sp_out, dsp, hip.from_numpy(rhs), p, 2, nz - 2, ctx=ctx)
        torch.cuda.synchronize()
        a, b = hip.to_numpy(full_out), hip.to_numpy(sp_out)
        assert np.array_equal(a, b) and np.array_equal(hip.to_numpy(dfull), hip.to_numpy(dsp))
        assert np.all(a[:, :, 0] == 777.0) == zlo and np.all(a[:, :, -1] == 777.0) == zhi
        # against the oracle: interior + x/y faces of the interior planes
        Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
        _oracle_iters(oracle, Pr, d, rhs, g, 1, 0, True, 0.0)
        assert np.array_equal(a[:, :, 1:-1], Pr[:, :, 1:-1])
        if not zlo:
            assert np.array_equal(a[:, :, 0], Pr[:, :, 0])
        if not zhi:
            assert np.array_equal(a[:, :, -1], Pr[:, :, -1])
    ctx.close()


@pytest.mark.parametrize("bc_kind", [0, 1])
def test_residual_max_and_pt_solve(hip, oracle, bc_kind):
    """ns3d_pt_solve = the whole inner loop incl. the every-nchk residual check and the early exit: identical
    iteration count, error history and fields as the oracle's loop (multi.jl:458-471 / gpu.jl:126-137)."""
    import torch
    nx, ny, nz = 24, 15, 15
    g = geometry(nx, ny, nz)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 53)
    rhs *= 1e-3
    ctx = hip.Context(0, "strict")
    dPr = hip.from_numpy(Pr0)
    p = _params(hip, dPr, g, bc_kind, True, 0.0)
    Rp = np.zeros((nx - 2, ny - 2, nz - 2), order="F")
    oracle.compute_res(Rp, Pr0, rhs, g["rho"], g["dt"], g["dx"], g["dy"], g["dz"])
    assert hip.residual_max(dPr, hip.from_numpy(rhs), p, ctx=ctx) == oracle.max_abs(Rp)
    for eps, niter, nchk in ((-1.0, 57, 14), (5e4, 400, 14), (1e-30, 45, 7)):
        Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
        it_ref, errs_ref = oracle.pt_solve(Pr, d, rhs, Rp, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"],
                                           g["dz"], bc_kind, True, 0.0, g["g"], eps, niter, nchk, 0.36, 1000.0)
        dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
        it, errs = hip.pt_solve(dPr, dd, hip.from_numpy(rhs), p, eps, niter, nchk, 0.36, 1000.0, ctx=ctx)
        torch.cuda.synchronize()
        assert it == it_ref and errs == errs_ref
        assert np.array_equal(hip.to_numpy(dPr), Pr) and np.array_equal(hip.to_numpy(dd), d)
    ctx.close()


def test_pt_solve_nan_breaks(hip, oracle):
    """`!isfinite(err)` exit (multi.jl:469): a NaN in ∇V must stop the loop at the first check."""
    nx, ny, nz = 17, 9, 5
    g = geometry(nx, ny, nz)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 59)
    rhs[3, 3, 2] = np.nan
    ctx = hip.Context(0, "strict")
    dPr = hip.from_numpy(Pr0)
    it, errs = hip.pt_solve(dPr, hip.from_numpy(d0), hip.from_numpy(rhs), _params(hip, dPr, g, 0, True, 0.0), 1e-3, 100,
                            8, 1.0, 1.0, ctx=ctx)
    assert it == 8 and len(errs) == 1 and np.isnan(errs[0])
    ctx.close()


def test_pt_f32(hip, oracle):
    """fp32 storage/arithmetic variant (BASELINE.json configs[4] 'fp32 vs fp64 stencil'): bit-identical to the
    oracle built with REAL=float."""
    import torch
    nx, ny, nz = 24, 15, 15
    g = geometry(nx, ny, nz)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 61, dtype=np.float32)
    Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
    _oracle_iters(oracle, Pr, d, rhs, g, 3, 0, True, 0.0)
    ctx = hip.Context(0, "strict")
    dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
    hip.pt_iterate(dPr, dd, hip.from_numpy(rhs), _params(hip, dPr, g, 0, True, 0.0), 3, ctx=ctx)
    torch.cuda.synchronize()
    assert np.array_equal(hip.to_numpy(dPr), Pr) and np.array_equal(hip.to_numpy(dd), d)
    ctx.close()


# ---- temporal blocking: two PT iterations per pass over memory -------------------------------------------------
SHAPES2 = [0, 100, 200, 300, 600, 103, 207, 101, 364, 316, 700, 800, 900, 1100, 1200, 703, 904,
           1300, 1900, 1905, 1392, 891]
GRIDS2 = GRIDS + [(260, 19, 9), (131, 40, 6), (66, 70, 5)]


@pytest.mark.parametrize("bc", [(0, True, 0.0), (0, False, 0.0), (0, True, 0.75), (1, False, 0.0)])
@pytest.mark.parametrize("grid", GRIDS2)
def test_pt_sweep2_equals_two_sweeps_bitexact(hip, oracle, grid, bc):
    """One k_pt_sweep2 launch == two reference iterations, bit for bit, for every tile shape / z-chunk, both boundary
    sets, tiles larger and smaller than the grid, overlapping tiles in x, y and z."""
    import torch
    nx, ny, nz = grid
    g = geometry(*grid)
    bc_kind, owns, val = bc
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 71)
    Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
    _oracle_iters(oracle, Pr, d, rhs, g, 2, bc_kind, owns, val)
    ctx = hip.Context(0, "strict")
    for shape in SHAPES2:
        ctx.set_pt2_variant(shape)
        dPr, dout, dd, drhs = hip.from_numpy(Pr0), hip.from_numpy(np.full_like(Pr0, 555.0)), hip.from_numpy(d0), hip.from_numpy(rhs)
        ddout = hip.from_numpy(np.full_like(d0, 444.0))
        hip.pt_sweep2(dPr, dout, dd, ddout, drhs, _params(hip, dPr, g, bc_kind, owns, val), ctx=ctx)
        torch.cuda.synchronize()
        assert np.array_equal(hip.to_numpy(ddout), d), "dPrdτ differs: shape %d" % shape
        assert np.array_equal(hip.to_numpy(dout), Pr), "Pr differs: shape %d" % shape
        assert np.array_equal(hip.to_numpy(dPr), Pr0) and np.array_equal(hip.to_numpy(drhs), rhs)
        assert np.array_equal(hip.to_numpy(dd), d0)
    ctx.close()


@pytest.mark.parametrize("grid", [(24, 15, 15), (70, 6, 7), (63, 38, 38)])
def test_pt_iterate_and_solve_with_temporal_blocking(hip, oracle, grid):
    """pt_iterate / pt_solve with two-iterations-per-pass enabled: odd and even counts, residual checks on odd nchk,
    early exits — identical counts, err history and fields."""
    import torch
    nx, ny, nz = grid
    g = geometry(*grid)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 73)
    rhs *= 1e-3
    ctx = hip.Context(0, "strict")
    ctx.set_pt2_variant(300 if nx > 60 else 100)   # explicit shape: the automatic choice keeps small grids on single sweeps
    for n in (1, 2, 3, 8, 11):
        Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
        _oracle_iters(oracle, Pr, d, rhs, g, n, 0, True, 0.0)
        dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
        hip.pt_iterate(dPr, dd, hip.from_numpy(rhs), _params(hip, dPr, g, 0, True, 0.0), n, ctx=ctx)
        torch.cuda.synchronize()
        assert np.array_equal(hip.to_numpy(dPr), Pr) and np.array_equal(hip.to_numpy(dd), d), n
    Rp = np.zeros((nx - 2, ny - 2, nz - 2), order="F")
    for eps, niter, nchk in ((-1.0, 57, 14), (5e4, 400, 13), (1e-30, 45, 7), (-1.0, 30, 1)):
        Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
        it_ref, errs_ref = oracle.pt_solve(Pr, d, rhs, Rp, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"],
                                           g["dz"], 0, True, 0.0, g["g"], eps, niter, nchk, 0.36, 1000.0)
        dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
        it, errs = hip.pt_solve(dPr, dd, hip.from_numpy(rhs), _params(hip, dPr, g, 0, True, 0.0), eps, niter, nchk,
                                0.36, 1000.0, ctx=ctx)
        torch.cuda.synchronize()
        assert it == it_ref and errs == errs_ref
        assert np.array_equal(hip.to_numpy(dPr), Pr) and np.array_equal(hip.to_numpy(dd), d)
    ctx.close()


@pytest.mark.parametrize("grid", [(256, 9, 7), (257, 10, 6), (258, 12, 5), (259, 34, 5), (66, 33, 6), (67, 35, 5),
                                   (130, 17, 9), (9, 64, 6), (8, 66, 70), (320, 8, 8)])
def test_pt_sweep2_tile_edge_sizes(hip, oracle, grid):
    """Grid extents straddling the tile widths/heights (64·WX−2, 4·WY−2 outputs per tile) and the z-chunk length, every
    tile shape: the two-column / two-row / two-plane overlaps must neither drop nor duplicate a cell."""
    import torch
    nx, ny, nz = grid
    g = geometry(*grid)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 79)
    Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
    _oracle_iters(oracle, Pr, d, rhs, g, 2, 0, True, 0.5)
    ctx = hip.Context(0, "strict")
    for shape in (100, 200, 300, 600, 104, 302, 0, 700, 800, 900, 1100, 1200, 1103, 1300, 1900, 1392, 894):
        ctx.set_pt2_variant(shape)
        dPr, dout, dd = hip.from_numpy(Pr0), hip.from_numpy(np.full_like(Pr0, 555.0)), hip.from_numpy(d0)
        ddout = hip.from_numpy(np.full_like(d0, 444.0))
        hip.pt_sweep2(dPr, dout, dd, ddout, hip.from_numpy(rhs), _params(hip, dPr, g, 0, True, 0.5), ctx=ctx)
        torch.cuda.synchronize()
        assert np.array_equal(hip.to_numpy(ddout), d), "dPrdτ differs: shape %d" % shape
        assert np.array_equal(hip.to_numpy(dout), Pr), "Pr differs: shape %d" % shape
    ctx.close()


@pytest.mark.parametrize("two", [False, True])
def test_pt_solve_hip_graph_replay(hip, oracle, two):
    """ns3d_pt_solve with every residual-check block replayed as a HIP graph (forced on; automatic below 3 M cells):
    same counts, err history and fields as the eager loop and as the oracle — odd nchk, early exit, repeated calls that
    reuse the cached graphs, with and without temporal blocking."""
    import torch
    nx, ny, nz = 40, 24, 24
    g = geometry(nx, ny, nz)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 83)
    rhs *= 1e-3
    Rp = np.zeros((nx - 2, ny - 2, nz - 2), order="F")
    ctx = hip.Context(0, "strict")
    ctx.set_graph_mode(1)
    ctx.set_pt2_variant(100 if two else -1)
    for rep in range(2):   # second pass hits the graph cache
        for eps, niter, nchk in ((-1.0, 75, 15), (5e4, 400, 23), (-1.0, 64, 8), (1e-30, 50, 7)):
            Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
            it_ref, errs_ref = oracle.pt_solve(Pr, d, rhs, Rp, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"],
                                               g["dz"], 0, True, 0.0, g["g"], eps, niter, nchk, 0.36, 1000.0)
            dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
            it, errs = hip.pt_solve(dPr, dd, hip.from_numpy(rhs), _params(hip, dPr, g, 0, True, 0.0), eps, niter, nchk,
                                    0.36, 1000.0, ctx=ctx)
            torch.cuda.synchronize()
            assert it == it_ref and errs == errs_ref, (rep, eps, niter, nchk)
            assert np.array_equal(hip.to_numpy(dPr), Pr) and np.array_equal(hip.to_numpy(dd), d)
    ctx.close()


# ---- the exact-division guard: values far outside the range in which the reciprocal sequence is proven exact -----------
def _bits_equal(a, b):
    """Bit-for-bit (signed zeros included); NaNs compare equal to NaNs whatever their payload."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    u = np.uint64 if a.dtype == np.float64 else np.uint32
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((a.view(u) == b.view(u)) | nan))


def _extreme_fields(nx, ny, nz, dtype, seed, mode):
    """Pressure fields that exercise the guard of STRICT's division-by-known-divisor (the cases it must hand to the plain
    IEEE division; an unguarded reciprocal sequence mis-rounds ≈30 % of the dividends near 1e-308 / 1e-38):
      dense   — 30 % of the cells replaced by zeros, −0, subnormals, tiny and huge magnitudes
      blocks  — contiguous regions scaled to the edge of the subnormal range, so that whole stencils are tiny
      sparse0 — a zero field with a few scattered tiny values (every stencil that touches one is tiny, the rest exact
                zeros; the neighbours across tile seams must notice)"""
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], seed, dtype=dtype)
    rng = np.random.Generator(np.random.MT19937(seed + 1000))
    f64 = dtype == np.float64
    edge = (1e-308, 1e-310) if f64 else (1e-38, 1e-40)
    with np.errstate(all="ignore"):
        if mode == "dense":
            kind = rng.integers(0, 8, size=Pr0.shape)
            pick = rng.uniform(size=Pr0.shape) < 0.3
            scale = {2: 1e-320 if f64 else 1e-44, 3: 1e-305 if f64 else 1e-37, 4: 1e-250 if f64 else 1e-30,
                     5: 1e-200 if f64 else 1e-12, 6: 1e250 if f64 else 1e22, 7: 1e300 if f64 else 1e30}
            for k, sc in scale.items():
                m = pick & (kind == k)
                Pr0[m] = (Pr0[m].astype(np.float64) * sc).astype(dtype)
            Pr0[pick & (kind == 0)] = 0.0
            Pr0[pick & (kind == 1)] = -0.0
        elif mode == "blocks":
            big = Pr0.astype(np.float64)
            big[nx // 3:2 * nx // 3] *= edge[0]
            big[:nx // 3, ny // 2:] *= edge[1]
            big[:, :, nz // 2:] *= 1e-3 if f64 else 1.0
            big[2 * nx // 3:, :ny // 3] = 0.0
            Pr0 = np.asfortranarray(big.astype(dtype))
            d0 = np.asfortranarray((d0.astype(np.float64) * edge[0]).astype(dtype))   # nothing large to mask a mis-rounded ∇²P
            rhs = np.zeros_like(rhs)
        else:
            pick = rng.uniform(size=Pr0.shape) < 0.004
            pick[[1, nx // 2, 254 % nx, 255 % nx, 62, 63, 64], :, :] |= rng.uniform(size=(7, ny, nz)) < 0.05
            big = np.where(pick, Pr0.astype(np.float64) * np.where(rng.uniform(size=Pr0.shape) < 0.5, edge[0], edge[1]), 0.0)
            Pr0 = np.asfortranarray(big.astype(dtype))
            d0 = np.asfortranarray((d0.astype(np.float64) * edge[0]).astype(dtype))
            rhs = np.zeros_like(rhs)
    return Pr0, d0, rhs


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("bc", [(0, True, 0.0), (0, True, 1e-310), (1, False, 0.0)])
@pytest.mark.parametrize("mode", ["dense", "blocks", "sparse0"])
def test_pt_exact_division_guard_extreme_values(hip, oracle, dtype, bc, mode):
    """Zeros, −0, subnormals, 1e-320 … 1e300 in Pr: one and two iterations, every kernel family, stay bit-identical to
    the oracle's IEEE divisions, signed zeros included."""
    import torch
    bc_kind, owns, val = bc
    ctx = hip.Context(0, "strict")
    with np.errstate(all="ignore"):
        for grid in ((260, 19, 9), (70, 21, 13), (131, 40, 6)):
            nx, ny, nz = grid
            g = geometry(*grid)
            Pr0, d0, rhs = _extreme_fields(nx, ny, nz, dtype, 97, mode)
            ref = {}
            Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
            for n in (1, 2):
                _oracle_iters(oracle, Pr, d, rhs, g, 1, bc_kind, owns, val)
                ref[n] = (Pr.copy(order="F"), d.copy(order="F"))
            p = _params(hip, hip.from_numpy(Pr0), g, bc_kind, owns, val)
            for variant in (100, 200, 2200, 2700):
                ctx.set_pt_variant(variant)
                ctx.set_pt2_variant(-1)
                dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
                hip.pt_iterate(dPr, dd, hip.from_numpy(rhs), p, 1, ctx=ctx)
                torch.cuda.synchronize()
                assert _bits_equal(hip.to_numpy(dPr), ref[1][0]) and _bits_equal(hip.to_numpy(dd), ref[1][1]), (grid, variant)
            for shape in (100, 300, 600, 1100, 1200, 703, 1300, 1900):
                ctx.set_pt2_variant(shape)
                dPr, dout, dd = hip.from_numpy(Pr0), hip.from_numpy(np.full_like(Pr0, 555.0)), hip.from_numpy(d0)
                ddout = hip.from_numpy(np.full_like(d0, 444.0))
                hip.pt_sweep2(dPr, dout, dd, ddout, hip.from_numpy(rhs), p, ctx=ctx)
                torch.cuda.synchronize()
                assert _bits_equal(hip.to_numpy(ddout), ref[2][1]), (grid, shape)
                assert _bits_equal(hip.to_numpy(dout), ref[2][0]), (grid, shape)
    ctx.close()


def test_pt2_first_use_tuning(hip, oracle):
    """ns3d_set_autotune (default on): ns3d_plan_pt — and the first ns3d_pt_iterate / ns3d_pt_solve — on a grid of >= 1.5 M
    cells time the tile shapes on the caller's arguments; the result is the oracle's, the choice is remembered (ns3d_pt_sweep2
    itself never measures, it looks the choice up), and turning the tuner off returns to the built-in choice."""
    import torch
    nx, ny, nz = 200, 164, 130
    g = geometry(nx, ny, nz)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 101)
    Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
    _oracle_iters(oracle, Pr, d, rhs, g, 2, 0, True, 0.25)
    ctx = hip.Context(0, "strict")
    p = _params(hip, hip.from_numpy(Pr0), g, 0, True, 0.25)
    seen = []
    for rep in range(3):
        dPr, dout, dd = hip.from_numpy(Pr0), hip.from_numpy(np.full_like(Pr0, 555.0)), hip.from_numpy(d0)
        ddout = hip.from_numpy(np.full_like(d0, 444.0))
        if rep == 0:                                           # before any plan: the built-in choice, no measurement
            hip.pt_sweep2(dPr, dout, dd, ddout, hip.from_numpy(rhs), p, ctx=ctx)
            torch.cuda.synchronize()
            assert ctx.last_pt2_variant() == 0
            assert np.array_equal(hip.to_numpy(dout), Pr) and np.array_equal(hip.to_numpy(ddout), d)
            hip.plan_pt(dPr, dout, dd, ddout, hip.from_numpy(rhs), p, ctx=ctx)
        hip.pt_sweep2(dPr, dout, dd, ddout, hip.from_numpy(rhs), p, ctx=ctx)
        torch.cuda.synchronize()
        seen.append(ctx.last_pt2_variant())
        assert np.array_equal(hip.to_numpy(dout), Pr) and np.array_equal(hip.to_numpy(ddout), d)
        assert np.array_equal(hip.to_numpy(dPr), Pr0) and np.array_equal(hip.to_numpy(dd), d0)
    assert seen[0] == seen[1] == seen[2]                       # tuned once, then remembered
    # pt_iterate on the same grid reuses the choice (same plane range)
    dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
    hip.pt_iterate(dPr, dd, hip.from_numpy(rhs), p, 2, ctx=ctx)
    torch.cuda.synchronize()
    assert ctx.last_pt2_variant() == seen[0]
    assert np.array_equal(hip.to_numpy(dPr), Pr) and np.array_equal(hip.to_numpy(dd), d)
    ctx.set_autotune(False)
    dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
    hip.pt_iterate(dPr, dd, hip.from_numpy(rhs), p, 2, ctx=ctx)
    torch.cuda.synchronize()
    assert ctx.last_pt2_variant() == 0
    assert np.array_equal(hip.to_numpy(dPr), Pr) and np.array_equal(hip.to_numpy(dd), d)
    ctx.close()


# ---- deeper temporal blocking: N PT iterations per pass over memory (k_pt_sweepN) -------------------------------
SHAPESN = [0, 100, 200, 600, 1100, 1200, 1600, 2200, 103, 207, 1105, 2203, 616, 192, 94, 2300, 2400, 2800, 2305, 2391, 2891, 2807,
           # k_pt_sweepD (round 4): the planes of P⁰ through an LDS-DMA ring — every shape, chunked / one chunk / short chunks
           3100, 3191, 3105, 3200, 3207, 3500, 3591, 3506, 3800, 3807, 3891,
           # fp32 only (round 4, A/B): the 1024-thread shape with the loads two z-steps ahead
           2500, 2507]


def _sweepn_all_shapes(hip, ctx, nlev, Pr0, d0, rhs, p, Pr, d, what, k0=None, k1=None, cmp=np.array_equal):
    import torch
    from navierstokes3d_amd import lib as L
    ran = 0
    for shape in SHAPESN:
        ctx.set_ptn_variant(shape)
        dPr, dout, dd, drhs = hip.from_numpy(Pr0), hip.from_numpy(np.full_like(Pr0, 555.0)), hip.from_numpy(d0), hip.from_numpy(rhs)
        ddout = hip.from_numpy(np.full_like(d0, 444.0))
        try:
            hip.pt_sweepn(nlev, dPr, dout, dd, ddout, drhs, p, k0, k1, ctx=ctx)
        except L.Ns3dError as e:                                  # a tile too small for this many levels (256×8 with 4)
            assert "cannot run" in str(e), e
            continue
        torch.cuda.synchronize()
        ran += 1
        got_P, got_d = hip.to_numpy(dout), hip.to_numpy(ddout)
        if k0 is None:
            assert cmp(got_d, d), "dPrdτ differs: %s shape %d levels %d" % (what, shape, nlev)
            assert cmp(got_P, Pr), "Pr differs: %s shape %d levels %d" % (what, shape, nlev)
        else:                                                     # output planes [k0,k1) only (+ their x/y faces)
            assert cmp(got_d[:, :, k0 - 1:k1 - 1], d[:, :, k0 - 1:k1 - 1]), (what, shape, nlev)
            assert cmp(got_P[:, :, k0:k1], Pr[:, :, k0:k1]), (what, shape, nlev)
            assert (got_P[:, :, k1 + 1:] == 555.0).all() and (got_d[:, :, k1:] == 444.0).all()
        assert np.array_equal(hip.to_numpy(dPr), Pr0) and np.array_equal(hip.to_numpy(dd), d0)
    assert ran >= 8


@pytest.mark.parametrize("nlev", [2, 3, 4])
@pytest.mark.parametrize("bc", [(0, True, 0.0), (0, False, 0.0), (0, True, 0.75), (1, False, 0.0)])
@pytest.mark.parametrize("grid", [(24, 15, 15), (70, 6, 7), (63, 38, 38), (260, 19, 9), (131, 40, 6), (66, 70, 5)])
def test_pt_sweepn_equals_n_sweeps_bitexact(hip, oracle, grid, bc, nlev):
    """One k_pt_sweepN launch == nlev reference iterations, bit for bit: every tile shape / z-chunk length, both boundary
    sets, tiles larger and smaller than the grid, tiles overlapping by 2(nlev−1) in x, y and z, grids thinner than the
    pipeline is deep."""
    nx, ny, nz = grid
    g = geometry(*grid)
    bc_kind, owns, val = bc
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 71)
    Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
    _oracle_iters(oracle, Pr, d, rhs, g, nlev, bc_kind, owns, val)
    ctx = hip.Context(0, "strict")
    _sweepn_all_shapes(hip, ctx, nlev, Pr0, d0, rhs, _params(hip, hip.from_numpy(Pr0), g, bc_kind, owns, val), Pr, d, str(grid))
    ctx.close()


@pytest.mark.parametrize("nlev", [3, 4])
@pytest.mark.parametrize("grid", [(62, 30, 7), (63, 31, 6), (64, 32, 5), (65, 33, 9), (122, 29, 8), (123, 34, 5), (126, 14, 9),
                                   (127, 15, 6), (250, 9, 7), (254, 10, 6), (9, 64, 6), (8, 66, 40), (320, 8, 8), (5, 5, 5)])
def test_pt_sweepn_tile_edge_sizes(hip, oracle, grid, nlev):
    """Grid extents straddling the tile strides (64·WX − 2(nlev−1) columns, CPT·WY − 2(nlev−1) rows per tile) and the
    z-chunk length: the overlaps must neither drop nor duplicate a cell."""
    nx, ny, nz = grid
    g = geometry(*grid)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 79)
    Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
    _oracle_iters(oracle, Pr, d, rhs, g, nlev, 0, True, 0.5)
    ctx = hip.Context(0, "strict")
    _sweepn_all_shapes(hip, ctx, nlev, Pr0, d0, rhs, _params(hip, hip.from_numpy(Pr0), g, 0, True, 0.5), Pr, d, str(grid))
    ctx.close()


@pytest.mark.parametrize("bc", [(0, True, 0.0), (0, False, 0.0), (0, True, 0.75), (1, False, 0.0)])
@pytest.mark.parametrize("grid", [(24, 15, 15), (70, 6, 13), (63, 38, 38), (260, 19, 12), (131, 40, 11), (66, 70, 12), (57, 25, 9), (58, 26, 10)])
def test_pt_sweepn_five_levels_f32(hip, oracle, grid, bc):
    """A FIFTH level exists where k_pt_sweepN has registers left: float32 on the 1024-thread 64×32 shape (variants 24xx; tiles
    overlap by 8 columns / rows / planes).  One launch == five reference iterations bit for bit — whole grids and plane
    sub-ranges, grids straddling the tile strides (56 columns, 24 rows), both boundary sets, STRICT; FAST within tolerance;
    float64 fields and shapes without room refuse."""
    import torch
    from navierstokes3d_amd import lib as L
    nx, ny, nz = grid
    bc_kind, owns, val = bc
    g = geometry(nx, ny, nz)
    P0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 1234, np.float32)
    Pr, d = P0.copy(order="F"), d0.copy(order="F")
    _oracle_iters(oracle, Pr, d, rhs, g, 5, bc_kind, owns, val)
    for mode, cmp in (("strict", np.array_equal), ("fast", lambda a, b: rel_l2(a, b) < 1e-5)):
        ctx = hip.Context(0, mode)
        p = _params(hip, hip.from_numpy(P0), g, bc_kind, owns, val)
        for v in (0, 2400, 2405, 2491):
            ctx.set_ptn_variant(v)
            for k0, k1 in ((None, None), (1, 4), (nz - 5, nz - 1), (3, nz - 3)):
                if k0 is not None and k1 <= k0:
                    continue
                dP, dout, dd = hip.from_numpy(P0), hip.from_numpy(np.full_like(P0, 555.0)), hip.from_numpy(d0)
                ddout = hip.from_numpy(np.full_like(d0, 444.0))
                hip.pt_sweepn(5, dP, dout, dd, ddout, hip.from_numpy(rhs), p, k0, k1, ctx=ctx)
                torch.cuda.synchronize()
                gP, gd = hip.to_numpy(dout), hip.to_numpy(ddout)
                a, b = (1, nz - 1) if k0 is None else (k0, k1)
                assert cmp(gd[:, :, a - 1:b - 1], d[:, :, a - 1:b - 1]) and cmp(gP[:, :, a:b], Pr[:, :, a:b]), (mode, v, k0, k1)
                if k0 is None:
                    assert cmp(gP, Pr)
                else:
                    assert (gP[:, :, b + 1:] == 555.0).all() and (gd[:, :, b:] == 444.0).all()
        if mode == "strict":
            ctx.set_ptn_variant(2800)                                  # a shape without room for five levels
            with pytest.raises(L.Ns3dError, match="cannot run"):
                hip.pt_sweepn(5, hip.from_numpy(P0), hip.from_numpy(P0), hip.from_numpy(d0), hip.from_numpy(d0), hip.from_numpy(rhs), p, ctx=ctx)
            P64 = P0.astype(np.float64)
            with pytest.raises(L.Ns3dError, match="levels"):
                q = _params(hip, hip.from_numpy(P64), g, bc_kind, owns, val)
                hip.pt_sweepn(5, hip.from_numpy(P64), hip.from_numpy(P64.copy(order="F")), hip.from_numpy(d0.astype(np.float64)),
                              hip.from_numpy(d0.astype(np.float64)), hip.from_numpy(rhs.astype(np.float64)), q, ctx=ctx)
        ctx.close()


@pytest.mark.parametrize("grid,n_iters", [((70, 21, 23), 13), ((63, 38, 38), 16), ((131, 40, 30), 11)])
def test_pt_iterate_and_solve_with_five_per_pass_f32(hip, oracle, grid, n_iters):
    """ns3d_set_pt_depth(5) on float32 fields: pt_iterate / pt_solve schedule passes of five (13 = 5+5+3, 16 = 5+5+4+2, 11 = 5+4+2)
    and give the oracle's iterates, counts and error history; a float64 context with the same setting runs four per pass."""
    import torch
    nx, ny, nz = grid
    g = geometry(nx, ny, nz)
    P0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 777, np.float32)
    rhs *= np.float32(1e-3)
    Pr, d = P0.copy(order="F"), d0.copy(order="F")
    _oracle_iters(oracle, Pr, d, rhs, g, n_iters, 0, True, 0.25)
    ctx = hip.Context(0, "strict")
    ctx.set_pt_depth(5)
    dP, dd = hip.from_numpy(P0), hip.from_numpy(d0)
    p = _params(hip, dP, g, 0, True, 0.25)
    hip.pt_iterate(dP, dd, hip.from_numpy(rhs), p, n_iters, ctx=ctx)
    torch.cuda.synchronize()
    assert ctx.last_pt_depth() in (2, 3, 4, 5)
    assert np.array_equal(hip.to_numpy(dP), Pr) and np.array_equal(hip.to_numpy(dd), d)
    Rp = np.zeros((nx - 2, ny - 2, nz - 2), dtype=np.float32, order="F")
    Pr, d = P0.copy(order="F"), d0.copy(order="F")
    it_ref, errs_ref = oracle.pt_solve(Pr, d, rhs, Rp, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True,
                                       0.25, g["g"], -1.0, 3 * n_iters, n_iters, 0.36, 1000.0)
    dP, dd = hip.from_numpy(P0), hip.from_numpy(d0)
    it, errs = hip.pt_solve(dP, dd, hip.from_numpy(rhs), p, -1.0, 3 * n_iters, n_iters, 0.36, 1000.0, ctx=ctx)
    torch.cuda.synchronize()
    assert it == it_ref and errs == errs_ref
    assert np.array_equal(hip.to_numpy(dP), Pr) and np.array_equal(hip.to_numpy(dd), d)
    ctx.close()
    # float64: the setting is accepted and clamps to four per pass
    P64, d64, r64 = fields(nx, ny, nz, ["c", "i", "c"], 778)
    Pr, d = P64.copy(order="F"), d64.copy(order="F")
    _oracle_iters(oracle, Pr, d, r64, g, n_iters, 0, True, 0.25)
    ctx = hip.Context(0, "strict")
    ctx.set_pt_depth(5)
    dP, dd = hip.from_numpy(P64), hip.from_numpy(d64)
    hip.pt_iterate(dP, dd, hip.from_numpy(r64), _params(hip, dP, g, 0, True, 0.25), n_iters, ctx=ctx)
    torch.cuda.synchronize()
    assert ctx.last_pt_depth() <= 4
    assert np.array_equal(hip.to_numpy(dP), Pr) and np.array_equal(hip.to_numpy(dd), d)
    ctx.close()


@pytest.mark.parametrize("nlev", [3, 4])
def test_pt_sweepn_plane_ranges_f32_fast_and_extremes(hip, oracle, nlev):
    """(a) output plane sub-ranges [k0,k1) — what a z-slab rank's seam / interior launches use; (b) float32; (c) FAST mode
    within tolerance; (d) the exact-division guard on extreme values (zeros, −0, subnormals, 1e-320 … 1e300, isolated tiny
    cells): the per-tile fall-back to plain divisions keeps every bit."""
    import torch
    nx, ny, nz = 70, 21, 23
    g = geometry(nx, ny, nz)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 55)
    Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
    _oracle_iters(oracle, Pr, d, rhs, g, nlev, 0, True, 0.25)
    ctx = hip.Context(0, "strict")
    p = _params(hip, hip.from_numpy(Pr0), g, 0, True, 0.25)
    for k0, k1 in ((1, nz - 1), (1, 4), (5, 9), (nz - 4, nz - 1), (nlev, nz - nlev)):
        _sweepn_all_shapes(hip, ctx, nlev, Pr0, d0, rhs, p, Pr, d, "planes %d:%d" % (k0, k1), k0, k1)
    ctx.close()
    # (b) float32
    P32, d32, r32 = fields(nx, ny, nz, ["c", "i", "c"], 56, np.float32)
    Pr, d = P32.copy(order="F"), d32.copy(order="F")
    _oracle_iters(oracle, Pr, d, r32, g, nlev, 0, True, 0.25)
    ctx = hip.Context(0, "strict")
    _sweepn_all_shapes(hip, ctx, nlev, P32, d32, r32, _params(hip, hip.from_numpy(P32), g, 0, True, 0.25), Pr, d, "f32")
    ctx.close()
    # (c) FAST mode: reciprocal constants + FMA, within 1e-6 relative L2
    Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
    _oracle_iters(oracle, Pr, d, rhs, g, nlev, 0, True, 0.25)
    ctx = hip.Context(0, "fast")
    close = lambda a, b: rel_l2(a, b) < 1e-6
    _sweepn_all_shapes(hip, ctx, nlev, Pr0, d0, rhs, _params(hip, hip.from_numpy(Pr0), g, 0, True, 0.25), Pr, d, "fast", cmp=close)
    ctx.close()
    # (d) extreme values
    for dtype in (np.float64, np.float32):
        for mode in ("dense", "blocks", "sparse0"):
            for bc in ((0, True, 1e-310), (1, False, 0.0)):
                E0, e0, er = _extreme_fields(nx, ny, nz, dtype, 97, mode)
                Pr, d = E0.copy(order="F"), e0.copy(order="F")
                _oracle_iters(oracle, Pr, d, er, g, nlev, *bc)
                ctx = hip.Context(0, "strict")
                _sweepn_all_shapes(hip, ctx, nlev, E0, e0, er, _params(hip, hip.from_numpy(E0), g, *bc), Pr, d,
                                   "extreme %s %s" % (mode, dtype.__name__), cmp=_bits_equal)
                ctx.close()


@pytest.mark.parametrize("sp", [dict(dx=1.0 / 64, dy=1.0 / 32, dz=1.0 / 128), dict(dx=0.5, dy=1.0, dz=2.0 ** -30),
                                dict(dx=0.5, dy=2.0, dz=1.0)])
def test_pt_power_of_two_spacings_every_kernel_family(hip, oracle, sp):
    """The fused PT kernels on power-of-two spacings (STRICT → the multiplication build, no range guard needed): single
    sweeps of every family, the two-iteration kernel in every tile shape, the N-iteration kernel — against the oracle's plain
    divisions, bit for bit, on ordinary and on extreme values."""
    import torch
    nx, ny, nz = 131, 21, 13
    g = dict(geometry(nx, ny, nz)); g.update(sp)
    for fieldset in ("plain", "dense", "sparse0"):
        if fieldset == "plain":
            Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 61)
        else:
            Pr0, d0, rhs = _extreme_fields(nx, ny, nz, np.float64, 97, fieldset)
        ref = {}
        Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
        for n in (1, 2, 3, 4):
            _oracle_iters(oracle, Pr, d, rhs, g, 1, 0, True, 0.25)
            ref[n] = (Pr.copy(order="F"), d.copy(order="F"))
        ctx = hip.Context(0, "strict")
        p = _params(hip, hip.from_numpy(Pr0), g, 0, True, 0.25)
        for variant in (100, 200, 2200, 2700):
            ctx.set_pt_variant(variant)
            ctx.set_pt2_variant(-1)
            dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
            hip.pt_iterate(dPr, dd, hip.from_numpy(rhs), p, 1, ctx=ctx)
            torch.cuda.synchronize()
            assert _bits_equal(hip.to_numpy(dPr), ref[1][0]) and _bits_equal(hip.to_numpy(dd), ref[1][1]), (fieldset, variant)
        for shape in (100, 300, 703, 800, 1100, 1300, 1900):
            ctx.set_pt2_variant(shape)
            dPr, dout, dd = hip.from_numpy(Pr0), hip.from_numpy(np.full_like(Pr0, 555.0)), hip.from_numpy(d0)
            ddout = hip.from_numpy(np.full_like(d0, 444.0))
            hip.pt_sweep2(dPr, dout, dd, ddout, hip.from_numpy(rhs), p, ctx=ctx)
            torch.cuda.synchronize()
            assert _bits_equal(hip.to_numpy(ddout), ref[2][1]) and _bits_equal(hip.to_numpy(dout), ref[2][0]), (fieldset, shape)
        for nlev in (3, 4):
            _sweepn_all_shapes(hip, ctx, nlev, Pr0, d0, rhs, p, ref[nlev][0], ref[nlev][1], "pow2 " + fieldset, cmp=_bits_equal)
        if fieldset == "plain":
            assert hip.residual_max(hip.from_numpy(ref[2][0]), hip.from_numpy(rhs), p, ctx=ctx) == _res_max(oracle, ref[2][0], rhs, g)
        ctx.close()


def _res_max(oracle, Pr, rhs, g):
    Rp = np.zeros(tuple(n - 2 for n in Pr.shape), order="F")
    oracle.compute_res(Rp, Pr, rhs, g["rho"], g["dt"], g["dx"], g["dy"], g["dz"])
    return oracle.max_abs(Rp)


@pytest.mark.parametrize("depth", [3, 4])
@pytest.mark.parametrize("grid", [(24, 15, 15), (70, 6, 7), (63, 38, 38)])
def test_pt_iterate_and_solve_with_deep_temporal_blocking(hip, oracle, grid, depth):
    """pt_iterate / pt_solve with three and four iterations per pass forced (ns3d_set_pt_depth): every remainder pattern
    (3+1 → 2+2, 3+2, 4+3, …), residual checks on nchk not divisible by the depth, early exits, HIP-graph replay of the
    check blocks — identical counts, err history and fields."""
    import torch
    nx, ny, nz = grid
    g = geometry(*grid)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 73)
    rhs *= 1e-3
    ctx = hip.Context(0, "strict")
    ctx.set_pt_depth(depth)
    for n in (1, 2, 3, 4, 5, 7, 8, 11, 13):
        Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
        _oracle_iters(oracle, Pr, d, rhs, g, n, 0, True, 0.0)
        dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
        hip.pt_iterate(dPr, dd, hip.from_numpy(rhs), _params(hip, dPr, g, 0, True, 0.0), n, ctx=ctx)
        torch.cuda.synchronize()
        assert np.array_equal(hip.to_numpy(dPr), Pr) and np.array_equal(hip.to_numpy(dd), d), n
        assert n < 2 or ctx.last_pt_depth() in (2, 3, 4)
    Rp = np.zeros((nx - 2, ny - 2, nz - 2), order="F")
    for graph in (0, 1):
        ctx.set_graph_mode(graph)
        for eps, niter, nchk in ((-1.0, 57, 14), (5e4, 400, 13), (1e-30, 45, 7), (-1.0, 30, 1), (-1.0, 40, 5)):
            Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
            it_ref, errs_ref = oracle.pt_solve(Pr, d, rhs, Rp, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"],
                                               g["dz"], 0, True, 0.0, g["g"], eps, niter, nchk, 0.36, 1000.0)
            dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
            it, errs = hip.pt_solve(dPr, dd, hip.from_numpy(rhs), _params(hip, dPr, g, 0, True, 0.0), eps, niter, nchk,
                                    0.36, 1000.0, ctx=ctx)
            torch.cuda.synchronize()
            assert it == it_ref and errs == errs_ref, (graph, eps, niter, nchk)
            assert np.array_equal(hip.to_numpy(dPr), Pr) and np.array_equal(hip.to_numpy(dd), d)
    ctx.close()


def _full_size_properties(hip, oracle, n, dtype, slab_at=None):
    """A full-size BASELINE grid (n³ cells, or n = (nx, ny, nz)), where the oracle cannot sweep the whole grid in test time: (a) the planned
    two-iteration kernel (ns3d_plan_pt times the tile shapes on these very arguments), two launches of the
    one-thread-per-cell sweep and two of the z-marching sweep give the same bits on the whole grid (three independent kernels;
    compared on the device), and the three-iteration pass equals three single sweeps; (b) locality — two iterations on planes [a+2, b-2) depend only on planes [a, b) — lets the oracle
    check a 30-plane sub-slab cut out of the middle, x/y faces and outlet plane included, bit for bit."""
    import torch
    nx, ny, nz = (n, n, n) if isinstance(n, int) else n
    g = geometry(nx, ny, nz)
    gen = torch.Generator(device="cuda"); gen.manual_seed(20240512)
    tdt, bits = (torch.float64, torch.int64) if dtype == "f64" else (torch.float32, torch.int32)
    zeros = lambda shape: hip.zeros(shape, tdt)

    def rnd_dev(*shape):
        t = zeros(shape)
        t.permute(2, 1, 0).uniform_(-1.0, 1.0, generator=gen)
        return t

    P0, D0, R = rnd_dev(nx, ny, nz), rnd_dev(nx - 2, ny - 2, nz - 2), rnd_dev(nx, ny, nz)
    ctx = hip.Context(0, "strict")
    p = _params(hip, P0, g, 0, True, 0.75)
    # (a1) the two-iteration kernel: plan (times the tile shapes, keeps the winner), then the launch proper
    Pa, Da = zeros((nx, ny, nz)), zeros((nx - 2, ny - 2, nz - 2))
    hip.plan_pt(P0, Pa, D0, Da, R, p, ctx=ctx)
    Pa.zero_(); Da.zero_()
    hip.pt_sweep2(P0, Pa, D0, Da, R, p, ctx=ctx)
    # (a2/a3) two single sweeps, two different kernel families
    for variant in (100, 2200):
        ctx.set_pt_variant(variant)
        Pb, Pc, Db = zeros((nx, ny, nz)), zeros((nx, ny, nz)), hip.clone(D0)
        hip.pt_sweep(P0, Pb, Db, R, p, 1, nz - 1, ctx=ctx)
        hip.pt_sweep(Pb, Pc, Db, R, p, 1, nz - 1, ctx=ctx)
        torch.cuda.synchronize()
        assert torch.equal(Pa.view(bits), Pc.view(bits)), "Pr differs between kernels at %r" % (n,)
        assert torch.equal(Da.view(bits), Db.view(bits)), "dPrdτ differs between kernels at %r" % (n,)
        if variant == 100:          # (a4) the three-iteration pass (k_pt_sweepN) against a third single sweep
            hip.pt_sweep(Pc, Pb, Db, R, p, 1, nz - 1, ctx=ctx)
            Pn, Dn = zeros((nx, ny, nz)), zeros((nx - 2, ny - 2, nz - 2))
            hip.pt_sweepn(3, P0, Pn, D0, Dn, R, p, ctx=ctx)
            torch.cuda.synchronize()
            assert torch.equal(Pn.view(bits), Pb.view(bits)), "Pr differs after three iterations at %r" % (n,)
            assert torch.equal(Dn.view(bits), Db.view(bits)), "dPrdτ differs after three iterations at %r" % (n,)
            del Pn, Dn
        del Pb, Pc, Db
    # (b) oracle on the sub-slab of planes [a, b) (sliced on the device: only the sub-slab crosses PCIe)
    mid = nz // 2 if slab_at is None else slab_at
    a, b = mid - 16, mid + 14
    Ps = hip.to_numpy(P0[:, :, a:b])
    Ds = hip.to_numpy(D0[:, :, a:b - 2])
    Rs = hip.to_numpy(R[:, :, a:b])
    _oracle_iters(oracle, Ps, Ds, Rs, dict(g), 2, 0, True, 0.75)
    assert np.array_equal(hip.to_numpy(Pa[:, :, a + 2:b - 2]), Ps[:, :, 2:-2])
    assert np.array_equal(hip.to_numpy(Da[:, :, a + 1:b - 3]), Ds[:, :, 1:-1])          # dPrdτ index = plane − 1
    ctx.close()
    del P0, D0, R, Pa, Da
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_full_size_512_cubed_properties(hip, oracle, dtype):
    """BASELINE.json configs[2] at its full size (512³, 134 M cells)."""
    _full_size_properties(hip, oracle, 512, dtype)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_full_size_1024_cubed_properties(hip, oracle, dtype):
    """BASELINE.json configs[4]'s one-GPU point (1024³, 1.07 G cells, fp64 and fp32): 8.6 / 4.3 GB per array."""
    _full_size_properties(hip, oracle, 1024, dtype)


@pytest.mark.parametrize("dtype,shape", [("f64", (1536, 1536, 1000)), ("f32", (2048, 2048, 1100))])
def test_grids_beyond_32_bit_indexing(hip, oracle, dtype, shape):
    """Maximum sizes: one MI355X holds 288 GB, so a single rank can own grids whose element count (2.36 G cells, fp64) or whose
    element AND byte offsets (4.61 G cells, fp32: past 2³² elements from plane 1024 on) do not fit 32 bits.  Same three-kernel
    agreement on the whole grid as the BASELINE sizes; the oracle's sub-slab is cut out of the LAST planes, where every offset
    is beyond the 32-bit range (19 / 18.5 GB per array, ≈ 190 GB in the test)."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    cells = shape[0] * shape[1] * shape[2]
    need = 10.5 * cells * (8 if dtype == "f64" else 4)
    if free < need:
        pytest.skip("needs %.0f GB of free device memory, %.0f GB available" % (need / 1e9, free / 1e9))
    _full_size_properties(hip, oracle, shape, dtype, slab_at=shape[2] - 24)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("grid", [(300, 200, 150), (257, 255, 131), (513, 511, 66), (190, 770, 75)])
def test_planned_passes_on_odd_large_grids(hip, grid, dtype):
    """Grids of 8–20 M cells with extents that divide no tile stride: ns3d_pt_iterate plans by itself (two-, three- and
    four-iteration candidates, one-round and chunked variants), and its 13 iterations — three passes of four and a single sweep, or
    whatever it chose — equal 13 launches of the one-thread-per-cell sweep bit for bit on the whole grid (compared on the device)."""
    import torch
    nx, ny, nz = grid
    g = geometry(nx, ny, nz)
    g["dx"], g["dy"], g["dz"] = 2.0 ** -8, 2.0 ** -7, 2.0 ** -8            # power-of-two spacings: the deep plans are candidates
    tdt, bits = (torch.float64, torch.int64) if dtype == "f64" else (torch.float32, torch.int32)
    gen = torch.Generator(device="cuda"); gen.manual_seed(77)

    def rnd_dev(*shape):
        t = hip.zeros(shape, tdt)
        t.permute(2, 1, 0).uniform_(-1.0, 1.0, generator=gen)
        return t

    P0, D0, R = rnd_dev(nx, ny, nz), rnd_dev(nx - 2, ny - 2, nz - 2), rnd_dev(nx, ny, nz)
    ctx = hip.Context(0, "strict")
    p = _params(hip, P0, g, 0, True, 0.5)
    Pa, Da = hip.clone(P0), hip.clone(D0)
    hip.pt_iterate(Pa, Da, R, p, 13, ctx=ctx)
    depth = ctx.last_pt_depth()
    ctx.set_pt_variant(100)
    Pb, Pc, Db = hip.clone(P0), hip.zeros((nx, ny, nz), tdt), hip.clone(D0)
    for _ in range(13):
        hip.pt_sweep(Pb, Pc, Db, R, p, 1, nz - 1, ctx=ctx)
        Pb, Pc = Pc, Pb
    torch.cuda.synchronize()
    assert depth in ((2, 3, 4) if dtype == "f64" else (2, 3, 4, 5))
    assert torch.equal(Pa.view(bits), Pb.view(bits)) and torch.equal(Da.view(bits), Db.view(bits)), (grid, dtype, depth)
    ctx.close()


def _timed_instance_properties(hip, oracle, n, dtype):
    """The kernel instance bench.py TIMES, at the size it is timed on: cavity_params spacings (dx = 1/n with n a power of two →
    the `strictp` arithmetic build), four iterations per pass.  (a) ns3d_plan_pt on these arguments, then ONE pass of
    ns3d_pt_sweepn(4) with the planner's tile shape and with each built-in shape the planner may return (fp64: 2800 chunked /
    2891 one round / 2300 / 2391; fp32: 2400 / 2491) against FOUR launches of the one-thread-per-cell sweep on the whole grid,
    bit for bit, compared on the device.  (b) locality — four iterations on planes [a+4, b−4) depend only on planes [a, b) —
    lets the oracle's unfused loop (update_dPrdτ!; update_Pr!; set_bc_Pr!, multi.jl:459-463) check a 30-plane sub-slab."""
    import torch
    from navierstokes3d_amd.params import cavity_params
    c = cavity_params(n)
    nx, ny, nz = c.nx, c.ny, c.nz
    g = dict(dx=c.dx, dy=c.dy, dz=c.dz, rho=c.rho, dt=c.dt, dtau=c.dtau, damp=c.damp, g=0.0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(20260301)
    tdt, bits = (torch.float64, torch.int64) if dtype == "f64" else (torch.float32, torch.int32)
    zeros = lambda shape: hip.zeros(shape, tdt)

    def rnd_dev(scale, *shape):
        t = zeros(shape)
        t.permute(2, 1, 0).uniform_(-scale, scale, generator=gen)
        return t

    P0, D0, R = rnd_dev(1.0, nx, ny, nz), rnd_dev(1.0, nx - 2, ny - 2, nz - 2), rnd_dev(1e-3, nx, ny, nz)
    ctx = hip.Context(0, "strict")
    assert ctx.arith_build(g["dx"], g["dy"], g["dz"]) == "strictp"
    p = _params(hip, P0, g, 0, False, 0.0)                  # bench.py: all-Neumann, no outlet plane
    # reference: four single sweeps, one thread per cell
    ctx.set_pt_variant(100)
    Pb, Pc, Db = hip.clone(P0), zeros((nx, ny, nz)), hip.clone(D0)
    for _ in range(4):
        hip.pt_sweep(Pb, Pc, Db, R, p, 1, nz - 1, ctx=ctx)
        Pb, Pc = Pc, Pb
    ctx.set_pt_variant(0)
    del Pc
    Pa, Da = zeros((nx, ny, nz)), zeros((nx - 2, ny - 2, nz - 2))
    hip.plan_pt(P0, Pa, D0, Da, R, p, ctx=ctx)
    planned = ctx.last_ptn_variant()
    shapes = [planned] + ([2800, 2891, 2300, 2391] if dtype == "f64" else [2400, 2491])
    for v in shapes:
        if v <= 0:
            continue
        ctx.set_ptn_variant(v)
        Pa.zero_(); Da.zero_()
        hip.pt_sweepn(4, P0, Pa, D0, Da, R, p, ctx=ctx)
        torch.cuda.synchronize()
        assert ctx.last_ptn_variant() == v
        assert torch.equal(Pa.view(bits), Pb.view(bits)), "Pr differs after a four-iteration pass, variant %d, %r" % (v, n)
        assert torch.equal(Da.view(bits), Db.view(bits)), "dPrdτ differs after a four-iteration pass, variant %d, %r" % (v, n)
    if dtype == "f32":      # the fifth level (fp32 only, 1024-thread shape): what the planner times against four — and takes on large grids
        Pc = zeros((nx, ny, nz))
        ctx.set_pt_variant(100)
        P5, D5 = hip.clone(Pb), hip.clone(Db)
        hip.pt_sweep(P5, Pc, D5, R, p, 1, nz - 1, ctx=ctx)          # fifth single sweep: Pc, D5
        ctx.set_pt_variant(0)
        for v in sorted({planned if planned // 100 == 24 else 2400, 2400, 2491}):
            ctx.set_ptn_variant(v)
            Pa.zero_(); Da.zero_()
            hip.pt_sweepn(5, P0, Pa, D0, Da, R, p, ctx=ctx)
            torch.cuda.synchronize()
            assert torch.equal(Pa.view(bits), Pc.view(bits)) and torch.equal(Da.view(bits), D5.view(bits)), "five-iteration pass, variant %d, %r" % (v, n)
        del Pc, P5, D5
        ctx.set_ptn_variant(planned if planned > 0 else 2400)
        Pa.zero_(); Da.zero_()
        hip.pt_sweepn(4, P0, Pa, D0, Da, R, p, ctx=ctx)
        torch.cuda.synchronize()
    # (b) the oracle on planes [a, b): four iterations are exact on [a+4, b−4)
    mid = nz // 2
    a, b = mid - 16, mid + 14
    Ps, Ds, Rs = hip.to_numpy(P0[:, :, a:b]), hip.to_numpy(D0[:, :, a:b - 2]), hip.to_numpy(R[:, :, a:b])
    _oracle_iters(oracle, Ps, Ds, Rs, g, 4, 0, False, 0.0)
    assert np.array_equal(hip.to_numpy(Pa[:, :, a + 4:b - 4]), Ps[:, :, 4:-4])
    assert np.array_equal(hip.to_numpy(Da[:, :, a + 3:b - 5]), Ds[:, :, 3:-3])          # dPrdτ index = plane − 1
    ctx.close()
    del P0, D0, R, Pa, Da, Pb, Db
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n", [512, 1024])
def test_timed_kernel_instance_at_its_own_size(hip, oracle, n, dtype):
    """the exact instance behind the headline number (strictp k_pt_sweepN<T,4,…>, its tile remap and
    one-round chunking) pinned at 512³ (BASELINE configs[2]) and 1024³ (configs[4])."""
    _timed_instance_properties(hip, oracle, n, dtype)


def test_graph_cache_ignores_struct_padding(hip, oracle):
    """ns3d_pt_params has four bytes of padding after owns_outlet; a C caller's stack struct (or Julia's
    Ref(PtParams(…))) leaves them indeterminate.  The HIP-graph cache of ns3d_pt_solve must hit all the same: with different
    garbage in the padding on every call the context keeps holding the same few graphs (one per buffer parity) instead of
    re-capturing one per call."""
    import ctypes as C
    import torch
    from navierstokes3d_amd import lib as L
    nx, ny, nz = 24, 15, 15
    g = geometry(nx, ny, nz)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 91)
    rhs *= 1e-3
    ctx = hip.Context(0, "strict")
    ctx.set_graph_mode(1)
    drhs = hip.from_numpy(rhs)
    assert C.sizeof(L.PtParams) == 104 and L.PtParams.outlet_val.offset == 80 and L.PtParams.owns_outlet.offset == 72
    counts = []
    Rp = np.zeros((nx - 2, ny - 2, nz - 2), order="F")
    Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
    it_ref, errs_ref = oracle.pt_solve(Pr, d, rhs, Rp, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True,
                                       0.0, g["g"], -1.0, 40, 8, 0.36, 1000.0)
    for call in range(6):
        dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
        p = _params(hip, dPr, g, 0, True, 0.0)
        C.memset(C.addressof(p) + 76, 0x11 * (call + 1), 4)              # the padding bytes
        it, errs = hip.pt_solve(dPr, dd, drhs, p, -1.0, 40, 8, 0.36, 1000.0, ctx=ctx)
        torch.cuda.synchronize()
        assert it == it_ref and errs == errs_ref
        assert np.array_equal(hip.to_numpy(dPr), Pr) and np.array_equal(hip.to_numpy(dd), d)
        counts.append(int(ctx.lib.ns3d_cached_graphs(ctx.handle)))
        del dPr, dd
    assert counts[0] >= 1
    # fresh tensors may land on new addresses (a legitimately different key); what must NOT happen is one capture per call
    # with the SAME buffers — so repeat on fixed buffers
    dPr, dd = hip.from_numpy(Pr0), hip.from_numpy(d0)
    base = None
    for call in range(5):
        dPr.copy_(hip.from_numpy(Pr0)); dd.copy_(hip.from_numpy(d0))
        p = _params(hip, dPr, g, 0, True, 0.0)
        C.memset(C.addressof(p) + 76, 0x7F - call, 4)
        hip.pt_solve(dPr, dd, drhs, p, -1.0, 40, 8, 0.36, 1000.0, ctx=ctx)
        torch.cuda.synchronize()
        n = int(ctx.lib.ns3d_cached_graphs(ctx.handle))
        base = n if base is None else base
        assert n == base, (call, n, base)
    ctx.close()



@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("grid", [(24, 15, 15), (70, 6, 7), (63, 38, 38), (131, 21, 35), (66, 7, 6), (5, 4, 3), (3, 3, 3), (200, 66, 30)])
def test_pt_persist_equals_single_sweeps(hip, oracle, grid, dtype, monkeypatch):
    """k_pt_persist (a whole block of PT iterations in one cooperative launch, the grid resident on the chip, faces handed
    between workgroups after every iteration): forced on (ns3d_set_persist_mode 1) against forced off, pt_iterate for several
    counts and pt_solve with residual checks and early exits — identical fields, counts and residual histories; both boundary
    sets; grids of one workgroup, of several in x, y and z, with ragged edges; the 63×38×38 case also against the oracle's loop."""
    import torch
    nx, ny, nz = grid
    g = geometry(*grid)
    npdt = np.float64 if dtype == "f64" else np.float32
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 97, npdt)
    rhs *= npdt(1e-3)
    drhs = hip.from_numpy(rhs)
    on, off = hip.Context(0, "strict"), hip.Context(0, "strict")
    on.set_persist_mode(1); off.set_persist_mode(0); off.set_graph_mode(0)
    for bc in ((0, True, 0.25), (0, False, 0.0), (1, False, 0.0)):
        for n in (1, 2, 3, 8, 37):
            res = []
            for ctx in (on, off):
                dP, dD = hip.from_numpy(Pr0), hip.from_numpy(d0)
                hip.pt_iterate(dP, dD, drhs, _params(hip, dP, g, *bc), n, ctx=ctx)
                torch.cuda.synchronize()
                res.append((hip.to_numpy(dP), hip.to_numpy(dD)))
            assert np.array_equal(res[0][0], res[1][0], equal_nan=True) and np.array_equal(res[0][1], res[1][1], equal_nan=True), (grid, bc, n)
        for eps, niter, nchk in ((-1.0, 57, 14), (1e-30, 45, 7), (5e4, 400, 13)):
            # on: the whole loop in one launch (round 4: residual checks and the decision inside k_pt_persist), then a launch per
            # residual-check block (NS3D_PERSIST_SOLVE=0); off: a launch per iteration
            res = []
            for ctx, whole in ((on, "1"), (on, "0"), (off, "1")):
                monkeypatch.setenv("NS3D_PERSIST_SOLVE", whole)
                dP, dD = hip.from_numpy(Pr0), hip.from_numpy(d0)
                it, errs = hip.pt_solve(dP, dD, drhs, _params(hip, dP, g, *bc), eps, niter, nchk, 0.36, 1000.0, ctx=ctx)
                torch.cuda.synchronize()
                res.append((it, errs, hip.to_numpy(dP), hip.to_numpy(dD)))
            monkeypatch.delenv("NS3D_PERSIST_SOLVE")
            for r in res[:2]:
                assert r[0] == res[2][0] and np.array_equal(r[1], res[2][1], equal_nan=True), (grid, bc, eps)
                assert np.array_equal(r[2], res[2][2], equal_nan=True) and np.array_equal(r[3], res[2][3], equal_nan=True)
    if grid == (63, 38, 38) and dtype == "f64":
        Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
        _oracle_iters(oracle, Pr, d, rhs, g, 8, 0, True, 0.25)
        dP, dD = hip.from_numpy(Pr0), hip.from_numpy(d0)
        hip.pt_iterate(dP, dD, drhs, _params(hip, dP, g, 0, True, 0.25), 8, ctx=on)
        torch.cuda.synchronize()
        assert np.array_equal(hip.to_numpy(dP), Pr) and np.array_equal(hip.to_numpy(dD), d)
    on.close(); off.close()


def test_pt_persist_scratch_lives_with_the_context(hip):
    """k_pt_persist's exchange area belongs to the context that launched it: a hundred contexts created and closed in turn (each
    with a stream of its own) all run it, and a context that changes stream between two blocks still gets the single sweeps' bits."""
    import torch
    grid = (40, 24, 24)
    nx, ny, nz = grid
    g = geometry(*grid)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 5)
    rhs *= 1e-3
    drhs = hip.from_numpy(rhs)
    off = hip.Context(0, "strict"); off.set_persist_mode(0); off.set_graph_mode(0)
    dP, dD = hip.from_numpy(Pr0), hip.from_numpy(d0)
    hip.pt_iterate(dP, dD, drhs, _params(hip, dP, g, 0, True, 0.25), 12, ctx=off)
    torch.cuda.synchronize()
    ref = (hip.to_numpy(dP), hip.to_numpy(dD))
    for q in range(100):
        ctx = hip.Context(0, "strict", async_=True); ctx.set_persist_mode(1)
        dP, dD = hip.from_numpy(Pr0), hip.from_numpy(d0)
        p = _params(hip, dP, g, 0, True, 0.25)
        hip.pt_iterate(dP, dD, drhs, p, 5, ctx=ctx)
        if q % 10 == 0:
            with torch.cuda.stream(torch.cuda.Stream()):      # the context follows the current stream
                hip.pt_iterate(dP, dD, drhs, p, 7, ctx=ctx)
        else:
            hip.pt_iterate(dP, dD, drhs, p, 7, ctx=ctx)
        ctx.sync(); torch.cuda.synchronize()
        assert np.array_equal(hip.to_numpy(dP), ref[0]) and np.array_equal(hip.to_numpy(dD), ref[1]), q
        ctx.close()
    off.close()


def test_pt_persist_expired_hand_over_is_detected_and_the_block_redone(hip, oracle, monkeypatch):
    """a cooperative block (k_pt_persist) in which a neighbour never arrives — forced here by NS3D_PERSIST_FAULT=1: workgroup 0
    publishes nothing and the bounded waits give up early — must not come back as a plausible-looking or NaN field behind an OK status.
    The launch writes to buffers of its own; ns3d_pt_solve (at its residual read-back) and ns3d_pt_iterate (one synchronisation)
    find the launch's ticket in the error word, redo the block by launches from the untouched inputs and switch the cooperative form
    off for the context: results equal the oracle's, ns3d_persist_faults says what happened.  With NS3D_COOP_CHECK=1 (how the rest of
    the suite runs) the launch itself fails with an error code."""
    import torch
    from navierstokes3d_amd import lib as L
    grid = (63, 38, 38)                              # BASELINE configs[0]: several workgroups in y and z
    nx, ny, nz = grid
    g = geometry(*grid)
    Pr0, d0, rhs = fields(nx, ny, nz, ["c", "i", "c"], 101)
    rhs *= 1e-3
    drhs = hip.from_numpy(rhs)
    bc = (0, True, 0.25)
    Pr, d = Pr0.copy(order="F"), d0.copy(order="F")
    _oracle_iters(oracle, Pr, d, rhs, g, 9, *bc)
    monkeypatch.setenv("NS3D_PERSIST_FAULT", "1")
    # (1) the suite's default NS3D_COOP_CHECK=1: an error code, inputs untouched
    ctx = hip.Context(0, "strict"); ctx.set_persist_mode(1)
    dP, dD = hip.from_numpy(Pr0), hip.from_numpy(d0)
    with pytest.raises(L.Ns3dError):
        hip.pt_iterate(dP, dD, drhs, _params(hip, dP, g, *bc), 9, ctx=ctx)
    torch.cuda.synchronize()
    assert np.array_equal(hip.to_numpy(dP), Pr0) and np.array_equal(hip.to_numpy(dD), d0) and ctx.persist_faults() == 1
    ctx.close()
    # (2) the product's default: detected where the library synchronises, block redone by launches, same bits as the oracle
    monkeypatch.setenv("NS3D_COOP_CHECK", "0")
    ctx = hip.Context(0, "strict"); ctx.set_persist_mode(1)
    dP, dD = hip.from_numpy(Pr0), hip.from_numpy(d0)
    hip.pt_iterate(dP, dD, drhs, _params(hip, dP, g, *bc), 9, ctx=ctx)
    torch.cuda.synchronize()
    assert ctx.persist_faults() == 1
    assert np.array_equal(hip.to_numpy(dP), Pr) and np.array_equal(hip.to_numpy(dD), d)
    # … and the cooperative form stays off for this context: the next block runs by launches although the fault is still armed
    hip.pt_iterate(dP, dD, drhs, _params(hip, dP, g, *bc), 5, ctx=ctx)
    torch.cuda.synchronize()
    assert ctx.persist_faults() == 1
    ctx.close()
    # (3) ns3d_pt_solve, the whole loop as one launch (the error word is read behind the launch's one synchronisation), and (4) a launch
    # per residual-check block (the check rides on the residual read-back): counts and residual history equal a run without the
    # cooperative form
    res = []
    for persist, whole in ((1, "1"), (0, "1"), (1, "0")):
        monkeypatch.setenv("NS3D_PERSIST_SOLVE", whole)
        ctx = hip.Context(0, "strict"); ctx.set_persist_mode(persist); ctx.set_graph_mode(0)
        dP, dD = hip.from_numpy(Pr0), hip.from_numpy(d0)
        it, errs = hip.pt_solve(dP, dD, drhs, _params(hip, dP, g, *bc), -1.0, 45, 7, 0.36, 1000.0, ctx=ctx)
        torch.cuda.synchronize()
        res.append((it, errs, hip.to_numpy(dP), hip.to_numpy(dD), ctx.persist_faults()))
        ctx.close()
    assert res[0][4] == 1 and res[1][4] == 0 and res[2][4] == 1
    for r in (res[0], res[2]):
        assert r[0] == res[1][0] and r[1] == res[1][1] and np.isfinite(r[1]).all()
        assert np.array_equal(r[2], res[1][2]) and np.array_equal(r[3], res[1][3])
