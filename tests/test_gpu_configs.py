"""BASELINE.json configs at their named shapes on the one GPU of the test box.

configs[3] "512×512×1024 cylinder, 2- and 4-GPU z-slab decomposition": ImplicitGlobalGrid arithmetic nz_g = P·(nz−2)+2 makes
1024 reachable for P = 2 (two slabs of 512×512×513) and not for P = 4 (257.5 planes) — the 4-slab case runs at its nearest
shape, 512×512×258 per rank = 1026 global (SURVEY.md §8d Config 4).  The ranks are virtual ranks of ONE process on device 0
(ns3d_mgpu_create): the schedule, events and peer copies are those of the multi-GPU node, only the link is not xGMI.
configs[4] "1024³, 8×MI355X": its decomposition — eight z-slabs of 1024×1024×130, fp32 and fp64 — and the 512³ strong-scaling shape
(eight of 512×512×66) run here as eight virtual ranks (round 4); its one-GPU point (1024³) is
tests/test_gpu_pt.py::test_full_size_1024_cubed_properties.
configs[1] (255×153×153) uncapped: one full second time step, 2 280 PT iterations, against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NAMES = ("C", "Pr", "Vx", "Vy", "Vz")


@pytest.mark.parametrize("P,nz_loc", [(2, 513), (4, 258)])
def test_config_D_slabs_equal_global_poisson_solve(hip, P, nz_loc):
    """The pseudo-transient loop of configs[3]'s ranks (ns3d_slab_*: deep-ghost two-iteration schedule, seam planes first,
    exchange behind the interior sweep) leaves, after 5 iterations (2+2+1), exactly the planes of the single-device solve
    of the 512×512×nz_g grid — compared on the device, every local plane, halo planes included."""
    import torch
    from navierstokes3d_amd.mgpu import MultiGpu
    from util import geometry
    nx = ny = 512
    nz_g = P * (nz_loc - 2) + 2
    assert nz_g in (1024, 1026)
    g = geometry(nx, ny, nz_g)
    gen = torch.Generator(device="cuda"); gen.manual_seed(31337)

    def rnd_dev(*shape):
        t = hip.zeros(shape, torch.float64)
        t.permute(2, 1, 0).uniform_(-1.0, 1.0, generator=gen)
        return t

    Pg, Dg, Rg = rnd_dev(nx, ny, nz_g), rnd_dev(nx - 2, ny - 2, nz_g - 2), rnd_dev(nx, ny, nz_g)
    cut = lambda A, lo, n: hip.clone(A[:, :, lo:lo + n])
    Pr = [cut(Pg, r * (nz_loc - 2), nz_loc) for r in range(P)]
    D = [cut(Dg, r * (nz_loc - 2), nz_loc - 2) for r in range(P)]
    R = [cut(Rg, r * (nz_loc - 2), nz_loc) for r in range(P)]
    ctx = hip.Context(0, "strict")
    pg = hip.pt_params(Pg, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    hip.pt_iterate(Pg, Dg, Rg, pg, 5, ctx=ctx)
    ctx.sync()
    mg = MultiGpu.create([0] * P, nx, ny, nz_loc, "strict")
    p = hip.pt_params(Pr[0], g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    mg.slab_load(Pr, D, R, p)
    mg.slab_iterate(5)
    res = mg.slab_residual()
    mg.slab_store(Pr, D)
    mg.sync()
    for r in range(P):
        lo = r * (nz_loc - 2)
        assert torch.equal(Pr[r].view(torch.int64), Pg[:, :, lo:lo + nz_loc].view(torch.int64)), "Pr of rank %d" % r
        assert torch.equal(D[r].view(torch.int64), Dg[:, :, lo:lo + nz_loc - 2].view(torch.int64)), "dPrdτ of rank %d" % r
    assert res == hip.residual_max(Pg, Rg, pg, ctx=ctx)
    ctx.close()
    mg.close()


@pytest.mark.parametrize("own_streams", [False, True])
@pytest.mark.parametrize("n,nz_loc,dtype", [(1024, 130, "f32"), (1024, 130, "f64"), (512, 66, "f64")])
def test_config_E_eight_z_slabs_equal_global_poisson_solve(hip, n, nz_loc, dtype, own_streams):
    """configs[4]'s decomposition: EIGHT z-slab ranks (scripts/runme3D.sh:18 `srun -n8`; multi.jl:325,458-471) — 8 × 1024×1024×130
    (nz_g = 8·128+2 = 1026: ImplicitGlobalGrid cannot make exactly 1024 from eight slabs, SURVEY §8d Config 5) in fp64 and fp32, and
    the 512³ strong-scaling shape 8 × 512×512×66 (nz_g = 514).  Eight virtual ranks of ONE process on device 0, each with its own
    communication stream (and, `own_streams`, its own non-blocking compute stream, so that the ready/landed events carry the ordering
    as between devices): slab_load / plan / iterate(9) / residual / store — six interior ranks with TWO seams each, the ranks'
    agreement on the pass depth, the ghost shrink after planning — against ns3d_pt_iterate on the global grid, every local plane,
    halo planes included, compared on the device (≈115 GB of HBM in the fp64 1024² case with the library's ghost-extended buffers)."""
    import torch
    from navierstokes3d_amd.mgpu import MultiGpu
    from util import geometry
    P = 8
    nx = ny = n
    nz_g = P * (nz_loc - 2) + 2
    tdt, bits = (torch.float64, torch.int64) if dtype == "f64" else (torch.float32, torch.int32)
    g = geometry(nx, ny, nz_g)
    gen = torch.Generator(device="cuda"); gen.manual_seed(8128)

    def rnd_dev(*shape):
        t = hip.zeros(shape, tdt)
        t.permute(2, 1, 0).uniform_(-1.0, 1.0, generator=gen)
        return t

    Pg, Dg, Rg = rnd_dev(nx, ny, nz_g), rnd_dev(nx - 2, ny - 2, nz_g - 2), rnd_dev(nx, ny, nz_g)
    cut = lambda A, lo, m: hip.clone(A[:, :, lo:lo + m])
    Pr = [cut(Pg, r * (nz_loc - 2), nz_loc) for r in range(P)]
    D = [cut(Dg, r * (nz_loc - 2), nz_loc - 2) for r in range(P)]
    R = [cut(Rg, r * (nz_loc - 2), nz_loc) for r in range(P)]
    ctx = hip.Context(0, "strict")
    pg = hip.pt_params(Pg, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    hip.pt_iterate(Pg, Dg, Rg, pg, 9, ctx=ctx)
    ref_res = hip.residual_max(Pg, Rg, pg, ctx=ctx)
    ctx.sync()
    ctx.close()
    torch.cuda.empty_cache()
    mg = MultiGpu.create([0] * P, nx, ny, nz_loc, "strict", own_streams=own_streams)
    p = hip.pt_params(Pr[0], g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    mg.slab_load(Pr, D, R, p)
    depth = mg.slab_plan()
    assert 1 <= depth <= (5 if dtype == "f32" else 4) and mg.ghost_depth() == depth - 1
    mg.slab_iterate(9)
    res = mg.slab_residual()
    mg.slab_store(Pr, D)
    mg.sync()
    for r in range(P):
        lo = r * (nz_loc - 2)
        assert torch.equal(Pr[r].view(bits), Pg[:, :, lo:lo + nz_loc].view(bits)), "Pr of rank %d" % r
        assert torch.equal(D[r].view(bits), Dg[:, :, lo:lo + nz_loc - 2].view(bits)), "dPrdτ of rank %d" % r
    assert res == ref_res
    mg.close()


@pytest.mark.parametrize("dims,n", [((2, 1, 1), (514, 512, 512)), ((2, 2, 1), (258, 258, 512))])
def test_default_topologies_of_init_global_grid_at_full_size(hip, dims, n):
    """What `init_global_grid(nx, ny, nz)` itself picks for 2 and 4 ranks — (2,1,1) and (2,2,1), never z-slabs — at configs[3]'s
    scale: two ranks of 514×512×512 (global 1026×512×512, 269 M cells) and four of 258×258×512.  ns3d_pt_solve_slab takes the
    deep-ghost box path there (solve_box: three ghost cells per decomposed direction, four iterations per pass); after 9
    iterations (4+4+1) with a residual check every 4 the counts, the error history and every local array — halo cells
    included — equal ns3d_pt_solve on the global grid, compared on the device."""
    import torch
    from navierstokes3d_amd.mgpu import MultiGpu
    from util import geometry
    N = tuple(dims[d] * (n[d] - 2) + 2 for d in range(3))
    g = geometry(*N)
    g["dtau"] = 0.8 / np.sqrt(1.0 / g["dx"] ** 2 + 1.0 / g["dy"] ** 2 + 1.0 / g["dz"] ** 2)
    gen = torch.Generator(device="cuda"); gen.manual_seed(4242)

    def rnd_dev(*shape):
        t = hip.zeros(shape, torch.float64)
        t.permute(2, 1, 0).uniform_(-1.0, 1.0, generator=gen)
        return t

    Pg, Dg, Rg = rnd_dev(*N), rnd_dev(N[0] - 2, N[1] - 2, N[2] - 2), rnd_dev(*N)
    P = dims[0] * dims[1] * dims[2]

    def cut(A, r, shrink):
        c = (r // (dims[1] * dims[2]), (r // dims[2]) % dims[1], r % dims[2])
        return hip.clone(A[tuple(slice(c[d] * (n[d] - 2), c[d] * (n[d] - 2) + n[d] - shrink) for d in range(3))])

    Pr = [cut(Pg, r, 0) for r in range(P)]
    D = [cut(Dg, r, 2) for r in range(P)]
    R = [cut(Rg, r, 0) for r in range(P)]
    ctx = hip.Context(0, "strict")
    pg = hip.pt_params(Pg, g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    it_ref, errs_ref = hip.pt_solve(Pg, Dg, Rg, pg, -1.0, 9, 4, 0.36, 1000.0, ctx=ctx)
    ctx.sync()
    mg = MultiGpu.create([0] * P, *n, "strict", dims=dims)
    p = hip.pt_params(Pr[0], g["rho"], g["dt"], g["dtau"], g["damp"], g["dx"], g["dy"], g["dz"], 0, True, 0.25, 0.0)
    it, errs = mg.pt_solve_slab(Pr, D, R, p, -1.0, 9, 4, 0.36, 1000.0)
    mg.sync()
    assert mg.pass_depth() == 4 and it == it_ref == 9 and errs == errs_ref and len(errs) == 2
    for r in range(P):
        c = (r // (dims[1] * dims[2]), (r // dims[2]) % dims[1], r % dims[2])
        sl = lambda shrink: tuple(slice(c[d] * (n[d] - 2), c[d] * (n[d] - 2) + n[d] - shrink) for d in range(3))
        assert torch.equal(Pr[r].view(torch.int64), Pg[sl(0)].view(torch.int64)), "Pr of rank %d" % r
        assert torch.equal(D[r].view(torch.int64), Dg[sl(2)].view(torch.int64)), "dPrdτ of rank %d" % r
    ctx.close()
    mg.close()


def test_config_D_cylinder_chorin_steps_on_two_slabs(hip):
    """configs[3] as what it says — flow around the cylinder, full Chorin steps, 512×512×1024 global on two z-slabs — through
    the explicit-shape entry of the driver (shape=…: multi.jl hard-codes ny = nz = ceil(0.6 nx)).  Two time steps (the first
    has an exactly divergence-free predictor; the second runs the PT loop, capped at two residual checks = 1 022
    iterations).  The fused path (ns3d_pt_solve_slab: two iterations per pass, deep ghosts, overlapped exchange) must equal
    the literal reference sequence (one launch per kernel, three update_halo! per iteration) bit for bit: counts, error
    history, every field.  The same entry is checked against the oracle at a size it can run in
    test_explicit_shape_entry_vs_oracle."""
    from navierstokes3d_amd.driver import run_navierstokes3D
    from navierstokes3d_amd.mgpu import MgpuGrid, MultiGpu
    shape = dict(ny=512, nz=513, ly_lx=1.0, lz_lx=2.0)
    runs = []
    for fused in (True, False):
        mg = MultiGpu.create([0, 0], 512, 512, 513, "strict")
        out = run_navierstokes3D(nx=512, nt=2, mode="strict", fused=fused, grid=MgpuGrid(mg, 512, 512, 513), shape=shape,
                                 niter_cap=1022, return_info=True)
        info = out[-1]
        assert info.params.nz_g == 1024 and info.params.nchk == 511
        runs.append((info.iters, info.errs, out[:5]))
        del out, info
        mg.close()
    assert runs[0][0] == runs[1][0] and runs[0][0][0] == 511 and 511 < runs[0][0][1] <= 1022
    assert runs[0][1] == runs[1][1] and np.isfinite(runs[0][1][1]).all() and runs[0][1][1][-1] > 0
    for n, a, b in zip(NAMES, runs[0][2], runs[1][2]):
        assert np.array_equal(a, b), n
    assert runs[0][2][1].shape == (510, 510, 1022)                       # Pr_v: halo-stripped global array (multi.jl:529)
    assert np.abs(runs[0][2][2]).max() > 0 and np.abs(runs[0][2][1]).max() > 0      # a developing flow, not zeros


@pytest.mark.parametrize("P", [1, 2])
def test_explicit_shape_entry_vs_oracle(hip, P):
    """The shape overrides (ny, nz, ly_lx, lz_lx) against the oracle driver with the same overrides: 48×30×(P·15+2), 2 steps
    (the second one runs the PT loop; a third step of this anisotropic grid blows up to 1e300 in the oracle as well, where
    the out-of-range float→int conversions of backtrack! are undefined in C and an InexactError in Julia)."""
    from navierstokes3d_amd.driver import run_navierstokes3D
    from navierstokes3d_amd.mgpu import MgpuGrid, MultiGpu
    from oracle.driver_ref import run_navierstokes3D_ref
    shape = dict(ny=30, nz=17, ly_lx=0.6, lz_lx=0.33 * P)
    ref = run_navierstokes3D_ref(nx=48, nt=2, dims_z=P, shape=shape)
    grid = None
    if P > 1:
        mg = MultiGpu.create([0] * P, 48, 30, 17, "strict")
        grid = MgpuGrid(mg, 48, 30, 17)
    out = run_navierstokes3D(nx=48, nt=2, mode="strict", grid=grid, shape=shape, return_info=True)
    assert out[-1].iters == ref[-1].iters and out[-1].errs == ref[-1].errs and out[-1].iters[-1] > out[-1].params.nchk
    assert all(np.isfinite(b).all() and np.abs(b).max() < 10 for b in ref[:5])
    for n, a, b in zip(NAMES, out[:5], ref[:5]):
        assert np.array_equal(a, b), n


def test_config_B_uncapped_second_step_vs_oracle(hip):
    """BASELINE.json configs[1] without a cap on the PT loop: 255×153×153, two time steps; the second one runs the full
    2 280 iterations to err < 1e-3.  Iteration counts, error history and all five returned fields bit-identical to the
    oracle (≈1 minute of CPU time for the oracle's 2 432 unfused iterations)."""
    from navierstokes3d_amd.driver import run_navierstokes3D
    from oracle.driver_ref import run_navierstokes3D_ref
    ref = run_navierstokes3D_ref(nx=255, nt=2)
    out = run_navierstokes3D(nx=255, nt=2, mode="strict", return_info=True)
    assert out[-1].iters == ref[-1].iters == [152, 2280]
    assert out[-1].errs == ref[-1].errs
    for n, a, b in zip(NAMES, out[:5], ref[:5]):
        assert np.array_equal(a, b), n
