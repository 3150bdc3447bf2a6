"""Time-boxed attempt (round 2) to reproduce the reference's only fixture, test/test3D.jl:8-32 (64 samples of Pr after
`run_navierstokes3D(nx=63, nt=1)`), with the oracle under plausible EARLIER variants of the script — the committed script
cannot have produced it (with the committed ICs Pr ≡ 0 after nt = 1).  TEST INFRASTRUCTURE ONLY.

Variants tried: obstacle = z-uniform cylinder | ellipsoid using the unused `c_lx` (multi.jl:306) as third axis; obstacle at
ox_lx = -0.4 (committed) | 0 (domain centre, where the fixture's hot spot x=31,y=19 sits); IC = `Vy[1,:,:]=vin` (committed) |
`Vx[1,:,:]=vin` | uniform Vx | uniform Vy; g = 0 (committed) | 9.81.  Result (python oracle/fixture_probe.py, ≈2 min): none of
the 32 comes within orders of magnitude (the fixture has O(0.2–0.6) at the hot spot with a far field of 1e-7…1e-4, i.e. a
localized source a few dozen PT iterations old; every variant with a non-zero source gives O(1–400) and a far field ≥ 1e-2).
A second family (`python oracle/fixture_probe.py radius`): sphere | cylinder of radius 0.05 … 0.12 lx at the domain centre
with uniform or inlet-plane Vx: hot spot O(2 … 2500), far field of the same size — again nothing close.
What the fixture itself says (read off its 64 numbers): the far field is uniform over x and y within a z plane
(−1.5e-4 at z=23, +1.5e-7 / +1.4e-8 at z=12 / 13), i.e. a front that left the TOP wall ≈ 20 cells ago — the damped-wave
iteration moves 1/√3.1 ≈ 0.57 cells per sweep, so the state is ≈ 37 sweeps old (= nchk, the first residual check) — plus a
compact source at the domain centre (0.2 … 0.6, decaying 2–3× per cell).  The committed program has neither a source at
the top wall nor anything at the domain centre (its obstacle sits at ox = −0.4 lx and its velocities start at zero).
The fixture therefore stays unusable and parity with the Julia program stays UNPINNED.
"""
import sys, math, itertools
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import oracle as K
from oracle.driver_ref import multi_params, _alloc_multi, _x_g

inds_x = [31, 38, 50, 51]; inds_y = [2, 5, 19, 31]; inds_z = [12, 13, 23, 23]
ref = np.array([
 [[1.533238393934448e-7,1.528864051208823e-7,1.5339073726141464e-7,1.5343984486166758e-7],
  [4.899815577977306e-7,1.5616085552454124e-7,1.5338275094223396e-7,1.534319424539255e-7],
  [0.19578294327792722,0.001261159687564842,1.5335773442293184e-7,1.534034099299091e-7],
  [1.533238393934448e-7,1.528864051208823e-7,1.5339073726141467e-7,1.5343984486166758e-7]],
 [[1.519773048824846e-8,1.3958550021722135e-8,1.4004152193940194e-8,1.4009325656876154e-8],
  [9.978308687706312e-7,2.2305446976064653e-8,1.4003073776407945e-8,1.400832708667883e-8],
  [0.6208831467566082,0.004033942968227375,1.3999179729038258e-8,1.4004434106377066e-8],
  [1.519773048824846e-8,1.3958550021722135e-8,1.4004152193940194e-8,1.4009325656876154e-8]],
 [[-0.00016058350179262487,-0.00015965105843374818,-0.00015358800129985288,-0.00015316531554535897],
  [-0.00015840028184221653,-0.00015895701351855638,-0.00015374902285328045,-0.00015330228647029593],
  [0.3830266309792465,0.0003346988596948679,-0.0001544133298812684,-0.00015388114137263316],
  [-0.0001605835017926249,-0.00015965105843374823,-0.0001535880012998529,-0.00015316531554535897]]])  # [z][y][x]

def obstacle(f, p, kind, ox, oy, oz, c2):
    nx, ny, nz = p.nx, p.ny, p.nz
    xc = f.xco_g + np.arange(nx + 1) * p.dx; yc = f.yco_g + np.arange(ny + 1) * p.dy; zc = f.zco_g + np.arange(nz + 1) * p.dz
    xv, yv, zv = xc - p.dx / 2, yc - p.dy / 2, zc - p.dz / 2
    def q(X, Y, Z):
        xr = (X - ox) * p.cosb - (Y - oy) * p.sinb; yr = (X - ox) * p.sinb + (Y - oy) * p.cosb
        v = xr * xr / p.a2 + yr * yr / p.b2
        if kind == 'sphere':
            v = v + (Z - oz) ** 2 / c2
        return v
    g = lambda a, b, c: np.meshgrid(a, b, c, indexing='ij')
    f.C[q(*g(xc[:nx], yc[:ny], zc[:nz])) < 1.05] = 1.0
    f.Vx[q(*g(xv[:nx + 1], yc[:ny], zc[:nz])) < 1.0] = 0.0
    f.Vy[q(*g(xc[:nx], yv[:ny + 1], zc[:nz])) < 1.0] = 0.0
    f.Vz[q(*g(xc[:nx], yc[:ny], zv[:nz + 1])) < 1.0] = 0.0

def run(kind='cyl', ox_lx=-0.4, oz_lx=0.0, ic='vy_plane', g=0.0, nt=1, c_lx=0.05, bc_before=False):
    p = multi_params(63)
    p.g = g
    f = _alloc_multi(p); nx, ny, nz = p.nx, p.ny, p.nz
    f.xco_g = _x_g(1, p.dx, nx, nx, 0) - (p.lx - p.dx) / 2
    f.yco_g = _x_g(1, p.dy, ny, ny, 0) - (p.ly - p.dy) / 2
    f.zco_g = _x_g(1, p.dz, nz, nz, 0) - (p.lz - p.dz) / 2
    ox, oz, c2 = ox_lx * p.lx, oz_lx * p.lx, (c_lx * p.lx) ** 2
    if ic == 'vy_plane': f.Vy[0, :, :] = p.vin
    elif ic == 'vx_plane': f.Vx[0, :, :] = p.vin
    elif ic == 'vx_all': f.Vx[:, :, :] = p.vin
    elif ic == 'vy_all': f.Vy[:, :, :] = p.vin
    for iz in range(nz):
        f.Pr[:, :, iz] = -(_x_g(iz + 1, p.dz, nz, nz, 0) - p.dz / 2) * p.rho * p.g
    obstacle(f, p, kind, ox, p.oy, oz, c2)
    for it in range(nt):
        K.update_tau(f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, f.Vx, f.Vy, f.Vz, p.mu, p.dx, p.dy, p.dz)
        K.predict_V(f.Vx, f.Vy, f.Vz, f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, p.rho, p.g, p.dt, p.dx, p.dy, p.dz)
        obstacle(f, p, kind, ox, p.oy, oz, c2)
        K.update_divV(f.divV, f.Vx, f.Vy, f.Vz, p.dx, p.dy, p.dz)
        iters, errs = K.pt_solve(f.Pr, f.dPrdtau, f.divV, f.Rp, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, 0, True, 0.0, p.g,
                                 p.eps, p.niter, p.nchk, p.err_scale_num, p.psc)
    Pv = np.asarray(f.Pr[1:-1, 1:-1, 1:-1])
    got = np.array([[[Pv[x - 1, y - 1, z - 1] for x in inds_x] for y in inds_y] for z in inds_z[:3]])
    return got, iters

def run_radius(kind, a_lx, ic):
    """second family: obstacle of radius a_lx·lx at the domain centre"""
    p = multi_params(63)
    p.a2 = p.b2 = (a_lx * p.lx) ** 2
    f = _alloc_multi(p); nx, ny, nz = p.nx, p.ny, p.nz
    f.xco_g = _x_g(1, p.dx, nx, nx, 0) - (p.lx - p.dx) / 2
    f.yco_g = _x_g(1, p.dy, ny, ny, 0) - (p.ly - p.dy) / 2
    f.zco_g = _x_g(1, p.dz, nz, nz, 0) - (p.lz - p.dz) / 2
    if ic == 'vx_plane': f.Vx[0, :, :] = p.vin
    else: f.Vx[:, :, :] = p.vin
    c2 = (a_lx * p.lx) ** 2
    obstacle(f, p, kind, 0.0, p.oy, 0.0, c2)
    K.update_tau(f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, f.Vx, f.Vy, f.Vz, p.mu, p.dx, p.dy, p.dz)
    K.predict_V(f.Vx, f.Vy, f.Vz, f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, p.rho, p.g, p.dt, p.dx, p.dy, p.dz)
    obstacle(f, p, kind, 0.0, p.oy, 0.0, c2)
    K.update_divV(f.divV, f.Vx, f.Vy, f.Vz, p.dx, p.dy, p.dz)
    iters, errs = K.pt_solve(f.Pr, f.dPrdtau, f.divV, f.Rp, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz, 0, True, 0.0, p.g,
                             p.eps, p.niter, p.nchk, p.err_scale_num, p.psc)
    Pv = np.asarray(f.Pr[1:-1, 1:-1, 1:-1])
    return np.array([[[Pv[x - 1, y - 1, z - 1] for x in inds_x] for y in inds_y] for z in inds_z[:3]]), iters


if __name__ == '__main__':
    if sys.argv[1:] == ['radius']:
        print('fixture: hot', ref[:, 2, 0], 'far', ref[:, 0, 3])
        for kind in ('sphere', 'cyl'):
            for a_lx in (0.05, 0.08, 0.1, 0.12):
                for ic in ('vx_all', 'vx_plane'):
                    got, iters = run_radius(kind, a_lx, ic)
                    print(kind, a_lx, ic, 'iters', iters, 'hot', got[:, 2, 0], 'far', got[:, 0, 3], flush=True)
        sys.exit(0)
    cases = []
    for kind in ('cyl', 'sphere'):
        for ox in (-0.4, 0.0):
            for ic in ('vy_plane', 'vx_plane', 'vx_all', 'vy_all'):
                for g in (0.0, 9.81):
                    cases.append(dict(kind=kind, ox_lx=ox, ic=ic, g=g))
    for c in cases:
        got, iters = run(**c)
        rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
        print(c, 'iters', iters, 'hot', ['%.4g' % got[z, 2, 0] for z in range(3)], 'far', '%.3g' % got[0, 0, 3], 'max rel diff %.3g' % rel.max(), flush=True)
