"""CPU tests of the oracle itself (no GPU): the C restatement must agree BIT FOR BIT with the independent NumPy
transcription, reproduce the committed golden fixtures, and satisfy analytic properties that do not depend on
any transcription.  The reference's own known-answer vector (test/test3D.jl:8-32) is stale
 — parity with the original Julia program is therefore UNPINNED; these tests pin the oracle against
everything else that exists.
"""
import os

import numpy as np
import pytest

from util import fields, geometry, rel_l2, rnd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GRIDS = [(17, 9, 5), (24, 15, 15), (5, 4, 3), (70, 6, 7)]


@pytest.fixture(scope="module")
def N():
    from oracle import numpy_ref
    return numpy_ref


def _both(oracle, N, name, arrs, *sc, **kw):
    a = [x.copy(order="F") for x in arrs]
    b = [x.copy(order="F") for x in arrs]
    getattr(oracle, name)(*a, *sc, **kw)
    getattr(N, name)(*b, *sc, **kw)
    for q, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x, y, equal_nan=True), "%s: array %d differs (max %g)" % (name, q, np.abs(x - y).max())
    return a


@pytest.mark.parametrize("grid", GRIDS)
def test_c_oracle_equals_numpy_transcription(oracle, N, grid):
    nx, ny, nz = grid
    g = geometry(*grid)
    Vx, Vy, Vz, Pr, divV, C, d = fields(nx, ny, nz, ["vx", "vy", "vz", "c", "c", "c", "i"], 101)
    tau = fields(nx, ny, nz, ["c", "c", "c", "s", "s", "s"], 201)
    tau = _both(oracle, N, "update_tau", tau + [Vx, Vy, Vz], g["mu"], g["dx"], g["dy"], g["dz"])[:6]
    _both(oracle, N, "predict_V", [Vx, Vy, Vz] + tau, g["rho"], g["g"], g["dt"], g["dx"], g["dy"], g["dz"])
    _both(oracle, N, "update_divV", [divV, Vx, Vy, Vz], g["dx"], g["dy"], g["dz"])
    _both(oracle, N, "update_dPrdtau", [Pr, d, divV], g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"])
    _both(oracle, N, "update_Pr", [Pr, d], g["dtau"])
    _both(oracle, N, "compute_res", [d, Pr, divV], g["rho"], g["dt"], g["dx"], g["dy"], g["dz"])
    _both(oracle, N, "correct_V", [Vx, Vy, Vz, Pr], g["dt"], g["rho"], g["dx"], g["dy"], g["dz"])
    for name in ("bc_x", "bc_y", "bc_z", "bc_zV"):
        for A in (Vx, Vy, Vz, Pr):
            _both(oracle, N, name, [A])
    _both(oracle, N, "bc_xhydstatic", [Pr], g["dz"], nz, g["g"], g["rho"])
    _both(oracle, N, "bc_x_Vx", [Vx], 1.0)
    _both(oracle, N, "bc_x_Pr", [Pr], 0.0)
    sc = (0.0121, 0.0064, -0.1, 0.02, np.sin(0.3), np.cos(0.3))
    _both(oracle, N, "set_cylinder", [C, Vx, Vy, Vz], *sc, -(1 - g["dx"]) / 2, -(0.6 - g["dy"]) / 2, 0.0, 1.0, 0.6, 0.7,
          g["dx"], g["dy"], g["dz"])
    _both(oracle, N, "set_cylinder_local", [C, Vx, Vy, Vz], *sc, 1.0, 0.6, 0.7, g["dx"], g["dy"], g["dz"])
    for faithful in (True, False):
        for cfl in (0.3, 1.0, 2.7):
            _both(oracle, N, "advect", [Vx.copy(order="F"), Vx, Vy.copy(order="F"), Vy, Vz.copy(order="F"), Vz,
                                        C.copy(order="F"), C], cfl * g["dx"], g["dx"], g["dy"], g["dz"],
                  faithful=faithful)
    assert oracle.max_abs(Pr) == N.max_abs(Pr)
    q = Pr.copy(order="F"); q[1, 1, 1] = np.nan
    assert np.isnan(oracle.max_abs(q)) and np.isnan(N.max_abs(q))


def test_driver_equals_numpy_driver(oracle, N):
    """Whole time loop (multi.jl:446-477, 1 rank): C-oracle driver vs the all-NumPy driver, 3 steps at nx=20."""
    from oracle.driver_ref import run_navierstokes3D_ref
    f, iters = N.run_multi_1rank(20, 3)
    out = run_navierstokes3D_ref(nx=20, nt=3)
    r = out[-1].ranks[0]
    assert iters == out[-1].iters
    for n in ("Pr", "C", "Vx", "Vy", "Vz", "dPrdtau", "divV"):
        assert np.array_equal(f[n], r[n]), n


def test_oracle_reproduces_goldens(oracle):
    """The committed fixtures are what the oracle produces today (guards against silent oracle drift)."""
    from oracle.driver_ref import run_navierstokes3D_ref, runme_ref
    gold = np.load(os.path.join(GOLD, "multi_nx24.npz"))
    for nt in (1, 2):
        out = run_navierstokes3D_ref(nx=24, nt=nt)
        assert out[-1].iters == gold["nt%d_iters" % nt].tolist()
        for n, a in zip(("C", "Pr", "Vx", "Vy", "Vz"), out[:5]):
            assert np.array_equal(a, gold["nt%d_%s" % (nt, n)]), (nt, n)
    gold = np.load(os.path.join(GOLD, "multi_nx63.npz"))
    out = run_navierstokes3D_ref(nx=63, nt=20)
    # iteration counts also match the survey's independent NumPy probe: total 12 950
    assert out[-1].iters == gold["iters"].tolist() == [37, 259, 296, 333, 407, 481, 518, 592, 666, 740, 814, 851, 925,
                                                        962, 888, 925, 999, 925, 703, 629]
    for n, a in zip(("C", "Pr", "Vx", "Vy", "Vz"), out[:5]):
        assert np.array_equal(a[::3, ::3, ::3], gold[n]), n
    gold = np.load(os.path.join(GOLD, "gpu_nx40.npz"))
    f, info = runme_ref(nx=40, nt=2)
    assert info.iters == gold["iters"].tolist()
    for n in ("C", "Pr", "Vx", "Vy", "Vz"):
        assert np.array_equal(np.asarray(f[n]), gold[n]), n


def test_kernel_known_answer_vectors(oracle):
    gold = np.load(os.path.join(GOLD, "kernels_17x9x5.npz"))
    nx, ny, nz = 17, 9, 5
    g = geometry(nx, ny, nz)
    a = fields(nx, ny, nz, ["c", "c", "c", "s", "s", "s", "vx", "vy", "vz"], 1)
    oracle.update_tau(*a, g["mu"], g["dx"], g["dy"], g["dz"])
    for q, n in enumerate(("txx", "tyy", "tzz", "txy", "txz", "tyz")):
        assert np.array_equal(a[q], gold["update_tau_" + n])
    a = fields(nx, ny, nz, ["c", "i", "c"], 1)
    oracle.update_dPrdtau(*a, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"])
    assert np.array_equal(a[1], gold["update_dPrdtau"])


# ---- analytic properties (independent of any transcription) ---------------------------------------------------
def test_laplacian_of_quadratic_is_exact(oracle):
    """compute_res! with ∇V = 0 returns ∇²Pr; for Pr = x² + 2y² − 3z² on a dyadic grid every operation is exact."""
    nx = ny = nz = 10
    h = 0.25
    x = (np.arange(nx) * h)[:, None, None]; y = (np.arange(ny) * h)[None, :, None]; z = (np.arange(nz) * h)[None, None, :]
    Pr = np.asfortranarray(x * x + 2 * y * y - 3 * z * z + 0 * (x + y + z))
    Rp = np.zeros((nx - 2, ny - 2, nz - 2), order="F")
    oracle.compute_res(Rp, Pr, np.zeros((nx, ny, nz), order="F"), 1000.0, 0.1, h, h, h)
    assert np.all(Rp == 2 + 4 - 6)


def test_divergence_of_linear_field_is_constant(oracle):
    nx, ny, nz = 9, 7, 6
    dx, dy, dz = 0.5, 0.25, 0.125
    Vx = np.asfortranarray(np.broadcast_to((3.0 * np.arange(nx + 1) * dx)[:, None, None], (nx + 1, ny, nz)).copy())
    Vy = np.asfortranarray(np.broadcast_to((-2.0 * np.arange(ny + 1) * dy)[None, :, None], (nx, ny + 1, nz)).copy())
    Vz = np.asfortranarray(np.broadcast_to((0.5 * np.arange(nz + 1) * dz)[None, None, :], (nx, ny, nz + 1)).copy())
    divV = np.zeros((nx, ny, nz), order="F")
    oracle.update_divV(divV, Vx, Vy, Vz, dx, dy, dz)
    assert np.all(divV == 3.0 - 2.0 + 0.5)
    # a divergence-free linear field has zero deviatoric normal stress trace and zero shear
    tau = [np.zeros((nx, ny, nz), order="F") for _ in range(3)] + [np.zeros((nx - 1, ny - 1, nz - 1), order="F") for _ in range(3)]
    oracle.update_tau(*tau, Vx, Vy, Vz, 1e-3, dx, dy, dz)
    assert np.all(tau[3] == 0) and np.all(tau[4] == 0) and np.all(tau[5] == 0)
    assert np.allclose(tau[0] + tau[1] + tau[2], 0.0, atol=1e-18)


def test_advect_uniform_half_cell_shift(oracle):
    """Uniform velocity with CFL 0.5 in +x: C_new[i] = ½(C_old[i-1] + C_old[i]); the inflow cell shows the reference's
    clamp-then-increment quirk."""
    nx, ny, nz = 12, 5, 4
    h = 0.125
    C_o = rnd(9, (nx, ny, nz))
    Vx_o = np.asfortranarray(np.ones((nx + 1, ny, nz))); Vy_o = np.zeros((nx, ny + 1, nz), order="F")
    Vz_o = np.zeros((nx, ny, nz + 1), order="F")
    outs = [np.zeros_like(a, order="F") for a in (Vx_o, Vy_o, Vz_o, C_o)]
    oracle.advect(outs[0], Vx_o, outs[1], Vy_o, outs[2], Vz_o, outs[3], C_o, 0.5 * h, h, h, h, True)
    assert np.array_equal(outs[3][1:], 0.5 * C_o[1:] + 0.5 * C_o[:-1])
    # inflow cell: base index floor(1-0.5)=0 is clamped to 1, its partner is base+1=2 (multi.jl:192,195), weight ½
    assert np.array_equal(outs[3][0], 0.5 * C_o[1] + 0.5 * C_o[0])
    assert np.all(outs[0][1:-1] == 1.0)                           # a uniform field is a fixed point
    assert np.all(outs[2] == 0.0)                                 # Vz is never written (App. B1)


def test_hydrostatic_balance_gpu_mode(oracle):
    """gpu.jl: hydrostatic Pr + gravity ⇒ correct_V!∘predict_V! leaves a fluid at rest at rest (inner Vz)."""
    from oracle.driver_ref import gpu_initial_fields, gpu_params
    p = gpu_params(16)
    nx, ny, nz = p.nx, p.ny, p.nz
    _, Pr = gpu_initial_fields(p)
    Vx = np.zeros((nx + 1, ny, nz), order="F"); Vy = np.zeros((nx, ny + 1, nz), order="F"); Vz = np.zeros((nx, ny, nz + 1), order="F")
    z = lambda *s: np.zeros(s, order="F")
    tau = [z(nx, ny, nz), z(nx, ny, nz), z(nx, ny, nz), z(nx - 1, ny - 1, nz - 1), z(nx - 1, ny - 1, nz - 1), z(nx - 1, ny - 1, nz - 1)]
    oracle.predict_V(Vx, Vy, Vz, *tau, p.rho, p.g, p.dt, p.dx, p.dy, p.dz)
    assert np.all(Vz[1:-1, 1:-1, 1:-1] < 0)                       # gravity accelerates downwards …
    oracle.correct_V(Vx, Vy, Vz, Pr, p.dt, p.rho, p.dx, p.dy, p.dz)
    assert np.abs(Vz[1:-1, 1:-1, 1:-1]).max() < 1e-12 * p.g * p.dt   # … and the hydrostatic gradient cancels it
    assert np.all(Vx == 0) and np.all(Vy == 0)


def test_virtual_zslab_ranks_pt_loop_is_decomposition_independent(oracle):
    """the PT loop is a Jacobi sweep, so P virtual z-slab ranks with ImplicitGlobalGrid's
    halo semantics reproduce the 1-rank solve bit for bit on the global grid (first time step, before advect!
    introduces its local clamping)."""
    from oracle.driver_ref import run_navierstokes3D_ref
    # 1 rank with nz_g = 2*(nz-2)+2 needs the same global grid: nx=20 → nz=12 local ×2 ranks = 22 global;
    # compare rank-local states of the 2-rank run after the PT loop of step 2 with a 3-rank run instead
    a = run_navierstokes3D_ref(nx=20, nt=1, dims_z=2)
    assert a[1].shape[2] == 2 * (12 - 2)
    # overlapping interior planes of neighbouring ranks hold identical values after the final halo update
    r0, r1 = a[-1].ranks
    for n in ("Pr", "Vx", "Vy", "C", "divV"):
        assert np.array_equal(r0[n][:, :, -1], r1[n][:, :, 1]), n
        assert np.array_equal(r0[n][:, :, -2], r1[n][:, :, 0]), n
    assert np.array_equal(r0["Vz"][:, :, -1], r1["Vz"][:, :, 2]) and np.array_equal(r0["Vz"][:, :, -3], r1["Vz"][:, :, 0])


def test_update_halo_3d_fills_faces_edges_and_corners():
    """ImplicitGlobalGrid's update_halo! on a Cartesian topology (oracle/driver_ref.py::update_halo_3d): each rank's array
    holds a code of the GLOBAL index everywhere except in the halo entries that have a neighbour, which hold garbage; after
    the exchange every entry — faces, and through the x→y→z order edges and corners — holds the code of its global index
    (a garbage corner that was forwarded, or a physical end that was overwritten, would show).  All staggers."""
    from oracle.driver_ref import cart_coords, gather_3d, update_halo_3d
    dims, n = (2, 3, 2), (6, 5, 7)
    P = dims[0] * dims[1] * dims[2]
    for stag in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]:
        ext = tuple(n[d] + stag[d] for d in range(3))
        code = lambda gi, gj, gk: gi + 1000.0 * gj + 1e6 * gk
        ranks = []
        for rk in range(P):
            c = cart_coords(rk, dims)
            gi = [c[d] * (n[d] - 2) + np.arange(ext[d]) for d in range(3)]
            A = code(gi[0][:, None, None], gi[1][None, :, None], gi[2][None, None, :]).astype(np.float64)
            want = A.copy()
            for d in range(3):
                for side, has in ((0, c[d] > 0), (-1, c[d] < dims[d] - 1)):
                    idx = [slice(None)] * 3
                    idx[d] = side
                    if has:
                        A[tuple(idx)] = -1.0 - rk            # garbage in every halo entry that has a neighbour
            ranks.append(dict(a=A, want=want))
        update_halo_3d(ranks, "a", n, dims)
        for rk, r in enumerate(ranks):
            assert np.array_equal(r["a"], r["want"]), (stag, rk)
    # an array without overlap in x (τxy-like, n-1 entries) keeps its x faces, still exchanges nothing there
    ranks = [dict(a=np.full((n[0] - 1, n[1] - 1, n[2] - 1), float(rk))) for rk in range(P)]
    update_halo_3d(ranks, "a", n, dims)
    assert all(np.all(r["a"] == rk) for rk, r in enumerate(ranks))
    # gather: blocks side by side in coordinate order
    ranks = [dict(a=np.full(n, float(rk))) for rk in range(P)]
    G = gather_3d(ranks, "a", dims)
    assert G.shape == tuple(dims[d] * (n[d] - 2) for d in range(3))
    assert G[0, 0, 0] == 0 and G[-1, -1, -1] == P - 1 and G[0, 0, -1] == 1 and G[0, n[1] - 2, 0] == 2 and G[n[0] - 2, 0, 0] == 6


def test_virtual_cartesian_ranks_time_step():
    """run_navierstokes3D_ref on the topology ImplicitGlobalGrid picks by default for 4 ranks, (2,2,1): the overlapping
    entries of x and y neighbours agree after the step's last halo update, and only the ranks on the inlet / outlet planes own
    them (multi.jl:164,179)."""
    from oracle.driver_ref import run_navierstokes3D_ref
    out = run_navierstokes3D_ref(nx=12, nt=1, dims=(2, 2, 1), niter_cap=30)
    r = out[-1].ranks
    assert out[1].shape == (2 * 10, 2 * 6, 6)
    assert [bool(f.owns_inlet) for f in r] == [True, True, False, False] and [bool(f.owns_outlet) for f in r] == [False, False, True, True]
    for n in ("Pr", "Vy", "Vz"):
        assert np.array_equal(r[0][n][-1, :, :], r[2][n][1, :, :]) and np.array_equal(r[0][n][-2, :, :], r[2][n][0, :, :]), n
    for n in ("Pr", "Vx", "Vz"):
        assert np.array_equal(r[0][n][:, -1, :], r[1][n][:, 1, :]) and np.array_equal(r[0][n][:, -2, :], r[1][n][:, 0, :]), n


def test_projection_identity_div_grad_is_the_laplacian(oracle):
    """Chorin's projection as an algebraic identity of the three restated operators, for ANY pressure field:
    update_∇V!(correct_V!(V*, Pr)) = update_∇V!(V*) − dt/ρ·∇²Pr = −dt/ρ·compute_res!(Pr, ∇V*) on every cell whose six faces
    correct_V! updates.  A misread index offset in @d_xi (correct_V!), @d_xa (update_∇V!) or @d2_xi (compute_res!) breaks it at
    O(1); dyadic data make every operation exact, so the identity holds to the bit."""
    nx, ny, nz = 11, 9, 8
    h, dt, rho = 0.5, 0.25, 2.0
    rng = np.random.default_rng(5)
    dy = lambda *s: np.asfortranarray(rng.integers(-8, 9, size=s).astype(np.float64) / 4)      # multiples of 1/4
    Pr, Vx, Vy, Vz = dy(nx, ny, nz), dy(nx + 1, ny, nz), dy(nx, ny + 1, nz), dy(nx, ny, nz + 1)
    div0 = np.zeros((nx, ny, nz), order="F")
    oracle.update_divV(div0, Vx, Vy, Vz, h, h, h)
    Rp = np.zeros((nx - 2, ny - 2, nz - 2), order="F")
    oracle.compute_res(Rp, Pr, div0, rho, dt, h, h, h)
    oracle.correct_V(Vx, Vy, Vz, Pr, dt, rho, h, h, h)
    div1 = np.zeros((nx, ny, nz), order="F")
    oracle.update_divV(div1, Vx, Vy, Vz, h, h, h)
    inner = div1[1:-1, 1:-1, 1:-1]
    assert np.abs(Rp).max() > 1 and np.array_equal(inner, -(dt / rho) * Rp)
    # … and the faces correct_V! leaves alone keep the divergence of the outermost cells as it was, up to their one inner face
    assert np.array_equal(div1[0, 0, 0], div0[0, 0, 0])


def test_viscous_predictor_is_exact_for_cubic_shear_flows(oracle):
    """update_τ! → predict_V! on the six pure shear flows u_a = s_b³ (velocity component a depending on coordinate b ≠ a, cell-
    centred coordinate s_b = (index+½)·h in b): the analytic answer is Δu_a = dt/ρ·μ·6·s_b, and the second difference of a
    cubic is exact, so the staggered stencils must return exactly that at every face predict_V! updates — with s_b the
    coordinate of the row the face itself sits in.  Pairing the wrong rows (an offset misread in @d_yi/@d_xa of the shear
    terms) gives 6·(s_b ± h) instead.  All data dyadic: exact to the bit."""
    n = (9, 8, 7)
    h, dt, rho, mu = 0.5, 0.25, 2.0, 0.5
    z = lambda *s: np.zeros(s, order="F")
    for a in range(3):
        for b in range(3):
            if a == b:
                continue
            V = [z(n[0] + 1, n[1], n[2]), z(n[0], n[1] + 1, n[2]), z(n[0], n[1], n[2] + 1)]
            s_b = (np.arange(n[b]) + 0.5) * h
            shape = [1, 1, 1]; shape[b] = n[b]
            V[a][...] = (s_b ** 3).reshape(shape)
            V0 = [v.copy(order="F") for v in V]
            tau = [z(*n), z(*n), z(*n), z(n[0] - 1, n[1] - 1, n[2] - 1), z(n[0] - 1, n[1] - 1, n[2] - 1), z(n[0] - 1, n[1] - 1, n[2] - 1)]
            oracle.update_tau(*tau, V[0], V[1], V[2], mu, h, h, h)
            assert all(np.all(t == 0) for t in tau[:3])                      # no normal deviatoric stress in a shear flow
            oracle.predict_V(V[0], V[1], V[2], *tau, rho, 0.0, dt, h, h, h)
            for c in range(3):
                dV = V[c] - V0[c]
                if c != a:
                    assert np.all(dV == 0), (a, b, c)
                    continue
                want = np.zeros_like(dV)
                want[...] = ((dt / rho) * mu * 6.0 * s_b).reshape(shape)
                upd = [slice(1, -1)] * 3                                      # predict_V! updates the inner entries only
                assert np.array_equal(dV[tuple(upd)], want[tuple(upd)]), (a, b)
                outer = np.ones(dV.shape, bool); outer[tuple(upd)] = False
                assert np.all(dV[outer] == 0), (a, b)


def test_advect_reproduces_linear_fields_exactly(oracle):
    """backtrack!/lerp (multi.jl:190-215) is a trilinear interpolation at the departure point x − v·dt: a field that is linear
    in x, y, z, carried by a uniform velocity with fractional CFL numbers of both signs in the three directions, comes out as
    the same linear function shifted by v·dt — exactly, on dyadic data — wherever the departure cell lies inside the array.
    (Which of the 8 neighbours pair with which weight is what the index/weight logic of backtrack! has to get right.)"""
    nx, ny, nz = 10, 9, 8
    h, dt = 0.5, 0.25
    v = (0.5, -1.0, 1.5)                                            # CFL = v·dt/h = 0.25, −0.5, 0.75
    lin = lambda X, Y, Z: 3.0 * X - 2.0 * Y + 0.5 * Z + 1.0
    xc, yc, zc = [(np.arange(m) + 0.5) * h for m in (nx, ny, nz)]
    C_o = np.asfortranarray(lin(xc[:, None, None], yc[None, :, None], zc[None, None, :]))
    Vx_o = np.asfortranarray(np.full((nx + 1, ny, nz), v[0])); Vy_o = np.asfortranarray(np.full((nx, ny + 1, nz), v[1]))
    Vz_o = np.asfortranarray(np.full((nx, ny, nz + 1), v[2]))
    outs = [np.zeros_like(a, order="F") for a in (Vx_o, Vy_o, Vz_o, C_o)]
    oracle.advect(outs[0], Vx_o, outs[1], Vy_o, outs[2], Vz_o, outs[3], C_o, dt, h, h, h, True)
    want = lin(xc[:, None, None] - v[0] * dt, yc[None, :, None] - v[1] * dt, zc[None, None, :] - v[2] * dt)
    assert np.array_equal(outs[3][1:-1, 1:-1, 1:-1], want[1:-1, 1:-1, 1:-1])
    assert not np.array_equal(outs[3][1:-1, 1:-1, 1:-1], C_o[1:-1, 1:-1, 1:-1])
    assert np.all(outs[0][1:-1, 1:-1, 1:-1] == v[0]) and np.all(outs[1][1:-1, 1:-1, 1:-1] == v[1])      # uniform fields are fixed points


# ---- the reference's own kernel text, evaluated mechanically (oracle/jl_eval.py) --------------------------------------
_JL2C = {"update_τ!": "update_tau", "predict_V!": "predict_V", "update_∇V!": "update_divV", "update_dPrdτ!": "update_dPrdtau",
         "update_Pr!": "update_Pr", "compute_res!": "compute_res", "correct_V!": "correct_V", "bc_x!": "bc_x", "bc_y!": "bc_y",
         "bc_z!": "bc_z", "bc_x_Vx!": "bc_x_Vx", "bc_x_Pr!": "bc_x_Pr", "bc_zV!": "bc_zV", "bc_xhydstatic!": "bc_xhydstatic"}


def _jl_cases_through(fn_of):
    """run every case of oracle/jl_eval.py through `fn_of(kernel name)`; yields (key prefix, argument dict after the call)"""
    from oracle import jl_eval
    for script in jl_eval.SCRIPTS:
        for grid in jl_eval.GRIDS:
            for q, (name, vals) in enumerate(jl_eval.cases(script, grid)):
                fn_of(_JL2C[name])(*vals.values())
                yield "%s/%dx%dx%d/%02d/%s/" % (script, grid[0], grid[1], grid[2], q, name), vals


def test_c_oracle_equals_the_reference_kernel_text_evaluated_mechanically(oracle):
    """tests/golden/jl_eval_kernels.npz holds the outputs of the reference's OWN kernel definitions — the statements of
    `@parallel function update_τ! … correct_V!` and of the `@parallel_indices` boundary kernels of both scripts, read from the
    .jl files and executed token by token under ParallelStencil's macro table (oracle/jl_eval.py; no hand transcription).
    The C oracle, called with the same seeded inputs in the reference's argument order, reproduces every array bit for bit."""
    gold = np.load(os.path.join(GOLD, "jl_eval_kernels.npz"))
    seen = 0
    for prefix, vals in _jl_cases_through(lambda n: getattr(oracle, n)):
        for a, v in vals.items():
            if isinstance(v, np.ndarray) and v.ndim == 3:
                assert np.array_equal(v, gold[prefix + a]), prefix + a
                seen += 1
    assert seen >= 100


def _oracle_call_part2(oracle, script, name, args):
    """the C oracle's entry for a part-2 case (plain-Julia kernels and host BC sequences), reference argument order"""
    if name == "set_cylinder!":
        (oracle.set_cylinder if script == "multi" else oracle.set_cylinder_local)(*args)
    elif name == "advect!":
        oracle.advect(*args, True)
    elif name == "set_bc_Vel!" and script == "multi":
        Vx, Vy, Vz, xvo_g, lx, vin = args
        oracle.set_bc_Vel(Vx, Vy, Vz, 0, xvo_g == -lx / 2, vin)          # multi.jl:164: the caller evaluates the comparison
    elif name == "set_bc_Vel!":
        oracle.set_bc_Vel(args[0], args[1], args[2], 1)                  # gpu.jl:264-279 (Vprof is unused there: commented out)
    elif name == "set_bc_Pr!" and script == "multi":
        Pr, xve_g, lx, val = args
        oracle.set_bc_Pr(Pr, 0, xve_g == lx / 2, val)
    else:
        Pr, dz, nz, g, rho = args
        oracle.set_bc_Pr(Pr, 1, True, 0.0, dz, nz, g, rho)


def test_c_oracle_equals_the_plain_julia_kernels_evaluated_mechanically(oracle):
    """The same for the kernels and host functions that are plain Julia — set_cylinder! (both scripts, gpu.jl's dx-for-dy
    slip included), advect!/backtrack!/lerp at |δ| below, up to and beyond one cell (clamps, the Vy-twice/never-Vz quirk),
    set_bc_Vel! and set_bc_Pr! with the inlet/outlet comparison true and false — transpiled line by line from the scripts
    (oracle/jl_eval.py part 2) and run thread by thread: every array the C oracle produces equals them bit for bit."""
    from oracle import jl_eval
    gold = np.load(os.path.join(GOLD, "jl_eval_kernels.npz"))
    seen = 0
    for script in jl_eval.SCRIPTS:
        for grid in jl_eval.GRIDS:
            for q, (name, args) in enumerate(jl_eval.cases2(script, grid)):
                _oracle_call_part2(oracle, script, name, args)
                for j, v in enumerate(args):
                    if isinstance(v, np.ndarray) and v.ndim == 3:
                        key = "%s/%dx%dx%d/p2_%02d/%s/%d" % (script, grid[0], grid[1], grid[2], q, name, j)
                        assert np.array_equal(v, gold[key]), key
                        seen += 1
    assert seen >= 120


def test_numpy_transcription_equals_the_reference_kernel_text():
    """… and so does the independent NumPy transcription (oracle/numpy_ref.py) for the stencil kernels."""
    from oracle import numpy_ref as N
    gold = np.load(os.path.join(GOLD, "jl_eval_kernels.npz"))
    seen = 0
    for prefix, vals in _jl_cases_through(lambda n: getattr(N, n, None) or (lambda *a: None)):
        name = prefix.split("/")[3]
        if getattr(N, _JL2C[name], None) is None:
            continue
        for a, v in vals.items():
            if isinstance(v, np.ndarray) and v.ndim == 3:
                assert np.array_equal(v, gold[prefix + a]), prefix + a
                seen += 1
    assert seen >= 60


def test_kernel_text_evaluation_is_reproducible_where_the_reference_is_present():
    """Where /root/reference exists (the build container), evaluate the scripts again and compare with the committed file."""
    from oracle import jl_eval
    if not jl_eval.available():
        pytest.skip("the reference scripts are not on this machine")
    gold = np.load(os.path.join(GOLD, "jl_eval_kernels.npz"))
    res = jl_eval.evaluate_all()
    res.update(jl_eval.evaluate_all2())
    assert set(res) == set(gold.files)
    for k, v in res.items():
        assert np.array_equal(v, gold[k]), k


_F2O = {"Pr": "Pr", "dPrdτ": "dPrdtau", "C": "C", "C_o": "C_o", "τxx": "txx", "τyy": "tyy", "τzz": "tzz", "τxy": "txy", "τxz": "txz",
        "τyz": "tyz", "Vx": "Vx", "Vy": "Vy", "Vz": "Vz", "Vx_o": "Vx_o", "Vy_o": "Vy_o", "Vz_o": "Vz_o", "∇V": "divV", "Rp": "Rp"}
_S2O = {"dτ": "dtau", "sinβ": "sinb", "cosβ": "cosb", "εit": "eps"}


def test_oracle_drivers_equal_the_reference_drivers_evaluated_from_their_text():
    """tests/golden/jl_eval_drivers.npz: run_navierstokes3D (multi.jl:288-373 setup + :446-477 time loop) and runme
    (gpu.jl:13-88 + :119-142) executed FROM THE SCRIPTS' TEXT by oracle/jl_eval.py part 3 (every kernel through parts 1–2) —
    derived scalars, iterations per step, residual histories and all 18 arrays after the last step.  The hand-written oracle
    drivers (oracle/driver_ref.py — what every GPU parity test compares the HIP path with) reproduce all of it bit for bit."""
    from oracle import jl_eval
    from oracle.driver_ref import run_navierstokes3D_ref, runme_ref
    gold = np.load(os.path.join(GOLD, "jl_eval_drivers.npz"))
    for script, nx, nt, cap in jl_eval.DRIVER_CASES:
        pre = "%s/nx%d_nt%d/" % (script, nx, nt)
        if script == "multi":
            out = run_navierstokes3D_ref(nx=nx, nt=nt, niter_cap=cap)
            info, f, p = out[-1], out[-1].ranks[0], out[-1].params
        else:
            f, info = runme_ref(nx=nx, nt=nt, niter_cap=cap)
            p = info.params
        assert info.iters == gold[pre + "iters"].tolist(), pre
        flat = [e for es in info.errs for e in es]
        assert [len(es) for es in info.errs] == gold[pre + "errs_per_step"].tolist() and flat == gold[pre + "errs"].tolist(), pre
        for jl in jl_eval.FIELDS:
            assert np.array_equal(np.asarray(f[_F2O[jl]]), gold[pre + "field/" + jl], equal_nan=True), pre + jl
        for k, v in zip(jl_eval.SCALARS, gold[pre + "scalars"]):
            if k == "niter" and cap is not None:
                assert v == cap                                # the one edit of a capped run
                continue
            assert float(getattr(p, _S2O.get(k, k))) == v, (pre, k)
        assert max(info.iters) > p.nchk                       # at least one step iterates past its first residual check


def test_driver_text_evaluation_is_reproducible_where_the_reference_is_present():
    from oracle import jl_eval
    if not jl_eval.available():
        pytest.skip("the reference scripts are not on this machine")
    gold = np.load(os.path.join(GOLD, "jl_eval_drivers.npz"))
    res = jl_eval.evaluate_drivers()
    assert set(res) == set(gold.files)
    for k, v in res.items():
        assert np.array_equal(v, gold[k], equal_nan=True), k


def test_config_a_from_the_reference_text(oracle):
    """BASELINE configs[0]'s grid — `run_navierstokes3D(nx=63)`, 63×38×38 — for the first three time steps (37, 259, 296 PT
    iterations), evaluated from multi.jl's text (oracle/jl_eval.py --config-a, ≈75 s of Python; stored as sha256 digests of the
    18 arrays plus iteration counts, residual histories and scalars in hex): the oracle driver gives the same bits."""
    import json
    from oracle import jl_eval
    from oracle.driver_ref import run_navierstokes3D_ref
    g = json.load(open(os.path.join(GOLD, "jl_eval_config_a.json"), encoding="utf-8"))
    out = run_navierstokes3D_ref(nx=g["nx"], nt=g["nt"])
    info, f, p = out[-1], out[-1].ranks[0], out[-1].params
    assert info.iters == g["iters"] == [37, 259, 296]
    assert [[float(e).hex() for e in es] for es in info.errs] == g["errs_hex"]
    for k, v in g["scalars_hex"].items():
        assert float(getattr(p, _S2O.get(k, k))).hex() == v, k
    for jl, digest in g["sha256"].items():
        a = np.asarray(f[_F2O[jl]])
        assert list(a.shape) == g["shape"][jl] and jl_eval.field_digest(a) == digest, jl
    assert g["max_abs"]["Pr"] > 0 and g["max_abs"]["Vx"] > 0


def test_viscous_predictor_pairs_the_right_neighbours_in_all_three_directions(oracle):
    """The cubic shear flows again, now modulated in the other two directions: u_a = s_a·s_c·s_b³ with s_a the FACE coordinate
    along the component's own direction (index·h) and s_b, s_c cell-centre coordinates.  Analytically Δu_a = dt/ρ·μ·6·s_a·s_c·s_b
    at the updated face's own (s_a, s_b, s_c); every term that is not ∂τ_ab/∂b cancels exactly in the stencils (τ_aa is constant
    along a, τ_ac constant along c), so `update_τ! → predict_V!` must return exactly that — which they only do if the shear
    stress an update reads was built from the velocities of ITS OWN face row/column/plane in all three directions (the +1
    offsets of @d_yi/@d_zi/@d_xi against @inn and @d_ya/@d_za/@d_xa).  Dyadic data, exact to the bit."""
    n = (7, 6, 5)
    h, dt, rho, mu = 0.5, 0.25, 2.0, 0.5
    z = lambda *s: np.zeros(s, order="F")
    for a in range(3):
        for b in range(3):
            if a == b:
                continue
            c = 3 - a - b
            ext = list(n); ext[a] += 1
            coord = [None] * 3
            coord[a] = np.arange(ext[a]) * h                          # faces along a
            coord[b] = (np.arange(ext[b]) + 0.5) * h                  # centres along b and c
            coord[c] = (np.arange(ext[c]) + 0.5) * h
            grid = np.meshgrid(*coord, indexing="ij")
            V = [z(n[0] + 1, n[1], n[2]), z(n[0], n[1] + 1, n[2]), z(n[0], n[1], n[2] + 1)]
            V[a][...] = grid[a] * grid[c] * grid[b] ** 3
            V0 = V[a].copy(order="F")
            tau = [z(*n), z(*n), z(*n), z(n[0] - 1, n[1] - 1, n[2] - 1), z(n[0] - 1, n[1] - 1, n[2] - 1), z(n[0] - 1, n[1] - 1, n[2] - 1)]
            oracle.update_tau(*tau, V[0], V[1], V[2], mu, h, h, h)
            oracle.predict_V(V[0], V[1], V[2], *tau, rho, 0.0, dt, h, h, h)
            want = (dt / rho) * mu * 6.0 * grid[a] * grid[c] * grid[b]
            inner = (slice(1, -1),) * 3
            assert np.array_equal((V[a] - V0)[inner], want[inner]), (a, b)
            assert np.abs(want[inner]).min() > 0


def test_the_reference_fixture_is_stale_by_the_reference_s_own_text():
    """test/test3D.jl:12-32 samples Pr of run_navierstokes3D(nx=63, nt=1) at 4×4×4 indices and expects 0.2 … 0.6 at a hot spot.
    The committed multi.jl, evaluated from its text (jl_eval_config_a.json → test3D_nt1), gives Pr ≡ 0 after that step (the
    predictor is exactly divergence-free) — and so does the oracle: the fixture belongs to an earlier program."""
    import json
    from oracle.driver_ref import run_navierstokes3D_ref
    sys_path_ok = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    g = json.load(open(os.path.join(GOLD, "jl_eval_config_a.json"), encoding="utf-8"))["test3D_nt1"]
    assert g["iters"] == [37] and g["max_abs_Pr"] == 0.0 and not np.any(np.array(g["Pr_samples"]))
    out = run_navierstokes3D_ref(nx=63, nt=1)
    assert out[-1].iters == [37] and not np.any(out[1])
    import importlib.util
    spec = importlib.util.spec_from_file_location("fixture_probe", os.path.join(sys_path_ok, "fixture_probe.py"))
    fp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fp)
    assert fp.ref.shape == (3, 4, 4) and np.abs(fp.ref).min() > 1e-9 and fp.ref[1, 2, 0] == 0.6208831467566082


@pytest.mark.parametrize("bc", [(0, True, 0.0), (0, True, 0.75), (0, False, 0.0), (1, False, 0.0)])
@pytest.mark.parametrize("grid", [(17, 9, 6), (24, 15, 15)])
def test_direct_pressure_solve_zeroes_the_reference_residual(oracle, grid, bc):
    """The option outside parity (oracle/direct_ref.py = the NumPy twin of csrc/ns3d_direct.hip): its result makes
    the REFERENCE's residual compute_res! (multi.jl:88-91) vanish to rounding — with set_bc_Pr!'s boundary cells applied by the
    oracle's own set_bc_Pr, so a wrong boundary rule, eigenbasis or right-hand-side correction shows at O(1) — and the oracle's
    PT loop started from it stays there (first check already below 1e-9 of the right-hand side's scale)."""
    from oracle.direct_ref import poisson_direct
    nx, ny, nz = grid
    g = geometry(*grid)
    bc_kind, owns, val = bc
    rhs = fields(nx, ny, nz, ["c"], 57)[0]
    if bc_kind == 0 and not owns:                       # all-Neumann: solvable only for a right-hand side without a mean
        rhs[1:-1, 1:-1, 1:-1] -= rhs[1:-1, 1:-1, 1:-1].mean()
    Pr = poisson_direct(rhs, g["rho"], g["dt"], g["dx"], g["dy"], g["dz"], bc_kind, owns, val, g["g"])
    P2 = Pr.copy(order="F")
    oracle.set_bc_Pr(P2, bc_kind, owns, val, g["dz"], nz, g["g"], g["rho"])
    assert np.array_equal(P2, Pr)                        # the boundary cells already are what set_bc_Pr! makes of the interior
    Rp = np.zeros((nx - 2, ny - 2, nz - 2), order="F")
    oracle.compute_res(Rp, Pr, rhs, g["rho"], g["dt"], g["dx"], g["dy"], g["dz"])
    scale = g["rho"] / g["dt"] * np.abs(rhs).max() + np.abs(Pr).max() / min(g["dx"], g["dy"], g["dz"]) ** 2
    assert np.abs(Rp).max() < 1e-11 * scale, np.abs(Rp).max() / scale
    if bc_kind == 0 and not owns:
        assert abs(Pr[1:-1, 1:-1, 1:-1].mean()) < 1e-9 * np.abs(Pr).max()          # the zero-mean solution


def test_pt_loop_converges_to_the_direct_solution(oracle):
    """… and the reference's own iteration, run far below its εit, ends at the direct solution (multi.jl boundary set)."""
    from oracle.direct_ref import poisson_direct
    nx, ny, nz = 20, 12, 12
    dx = 1.0 / nx
    g = dict(dx=dx, dy=dx, dz=dx, rho=1000.0, dt=dx, dtau=dx / np.sqrt(3.1), damp=2.0 / nx, g=0.0)
    rhs = fields(nx, ny, nz, ["c"], 58)[0] * 1e-3
    ref = poisson_direct(rhs, g["rho"], g["dt"], g["dx"], g["dy"], g["dz"], 0, True, 0.0)
    Pr = np.zeros((nx, ny, nz), order="F"); d = np.zeros((nx - 2, ny - 2, nz - 2), order="F"); Rp = np.zeros_like(d)
    it, errs = oracle.pt_solve(Pr, d, rhs, Rp, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.0, 0.0,
                               1e-13, 20000, 50, 1.0, np.abs(ref).max() / dx ** 2)
    assert it < 20000 and rel_l2(Pr, ref) < 1e-9, (it, rel_l2(Pr, ref))


@pytest.mark.parametrize("P,nz_loc", [(2, 12), (4, 7)])
def test_wide_advect_halo_makes_the_oracle_s_time_step_decomposition_independent(oracle, P, nz_loc):
    """The option outside the reference's multi-rank semantics (ns3d_advect_wide; oracle/driver_ref.py::advect_wide_z): with the old
    fields' z halo two planes wide, C's halo updated, AND the departure indices computed from GLOBAL plane numbers, P virtual ranks
    reproduce the one-rank run of the same 36×22×22 grid bit for bit — every plane of every field after three steps (105 PT
    iterations in the last one).  The reference's own multi-rank run does not (second half of the test): backtrack! clamps to the
    local array, update_halo! skips C (multi.jl:477), and `Float(iz) − δ` (multi.jl:194) rounds differently for a rank's local
    iz = 2 and the global iz = 12 when δ is below an ulp — the near-zero vertical velocities of the first steps are exactly there."""
    from oracle.driver_ref import run_navierstokes3D_ref
    nx, nt = 36, 3
    one = run_navierstokes3D_ref(nx=nx, nt=nt)
    f1 = one[-1].ranks[0]
    assert one[-1].params.nz == P * (nz_loc - 2) + 2 and all(np.isfinite(a).all() for a in one[:5])

    def mismatches(out):
        bad = []
        for r in range(P):
            lo = r * (nz_loc - 2)
            for n, extra in (("C", 0), ("Pr", 0), ("Vx", 0), ("Vy", 0), ("Vz", 1), ("divV", 0), ("dPrdtau", -2)):
                if not np.array_equal(out[-1].ranks[r][n], f1[n][:, :, lo:lo + nz_loc + extra]):
                    bad.append((r, n))
        return bad

    wide = run_navierstokes3D_ref(nx=nx, nt=nt, dims_z=P, shape=dict(nz=nz_loc), wide_advect_halo=True)
    assert wide[-1].iters == one[-1].iters and wide[-1].errs == one[-1].errs and mismatches(wide) == []
    plain = run_navierstokes3D_ref(nx=nx, nt=nt, dims_z=P, shape=dict(nz=nz_loc))
    assert mismatches(plain) != []
