"""Oracle drivers: CPU restatement of the reference's two time loops.  TEST INFRASTRUCTURE ONLY.

    run_navierstokes3D_ref  follows scripts/NavierStokes3D_multi_gpu.jl:287-536 (no vis/save),
                            optionally with P *virtual* z-slab ranks held in one process and a literal
                            restatement of ImplicitGlobalGrid's update_halo!/gather! ([upstream] semantics,
                            SURVEY.md §2.4) — P=1 is what the reference's own test exercises.
    runme_ref               follows scripts/NavierStokes3D_gpu.jl:12-173.

PARITY UNPINNED — see oracle/ns3d_oracle.c.  Parameter derivation is restated here independently of
navierstokes3d_amd/params.py on purpose.
"""
import math

import numpy as np

from . import oracle as K


class Obj(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _ceil_int(x):
    return int(math.ceil(x))


# ------------------------------------------------------------------------------------------------
# multi.jl
# ------------------------------------------------------------------------------------------------
def multi_params(nx, dims_z=1, dtype=np.float64, ny=None, nz=None, ly_lx=0.6, lz_lx=0.6, dims=None):
    """multi.jl:290-341 (local nx; ny,nz local; z decomposed over dims_z ranks).  ny, nz, ly_lx, lz_lx override the
    literals multi.jl:302-303,323-324 (defaults = the reference) for grids it cannot produce unedited."""
    p = Obj()
    p.lx, p.rho, p.vin, p.mu = 1.0, 1000.0, 1.0, 0.001           # :290-293
    p.psc = p.rho * (p.vin * p.vin)                             # :296
    Fr = math.inf                                                # :301
    a_lx, b_lx, ox_lx, oy_lx = 0.05, 0.05, -0.4, 0.0            # :304-308 (ly_lx, lz_lx :302-303 are arguments)
    beta = 0 * math.pi / 6                                       # :309
    p.ly, p.lz = ly_lx * p.lx, lz_lx * p.lx                      # :312-313
    p.ox, p.oy = ox_lx * p.lx, oy_lx * p.lx                      # :314-315
    p.g = 1 / (Fr * Fr) * (p.vin * p.vin) / p.lx                # :316  (= 0.0)
    p.a2, p.b2 = (a_lx * p.lx) * (a_lx * p.lx), (b_lx * p.lx) * (b_lx * p.lx)   # :317-318
    p.sinb, p.cosb = math.sin(beta), math.cos(beta)              # :319
    p.nx = nx
    p.ny = _ceil_int(nx * ly_lx) if ny is None else int(ny)      # :323
    p.nz = _ceil_int(nx * lz_lx) if nz is None else int(nz)      # :324
    p.dims = (1, 1, dims_z) if dims is None else tuple(int(q) for q in dims)     # :325 (dims: a 3-D Cartesian topology)
    # ImplicitGlobalGrid: n_g = dims*(n-overlap)+overlap, overlap 2 [upstream]
    p.nx_g, p.ny_g, p.nz_g = (p.dims[0] * (p.nx - 2) + 2, p.dims[1] * (p.ny - 2) + 2, p.dims[2] * (p.nz - 2) + 2)
    p.eps = 1e-3                                                 # :327
    p.niter = 50 * max(p.nx_g, p.ny_g, p.nz_g)                   # :328
    p.nchk = 1 * (p.ny_g - 1)                                    # :329
    CFLtau = 1.0 / math.sqrt(3.1)                                # :333
    CFL_visc, CFL_adv = 1 / 4.1, 1.0                             # :334-335
    p.dx, p.dy, p.dz = p.lx / p.nx_g, p.ly / p.ny_g, p.lz / p.nz_g       # :338
    m = max(p.dx, p.dy, p.dz)
    p.dt = min(CFL_visc * (m * m) * p.rho / p.mu, CFL_adv * m / p.vin)   # :339
    p.damp = 2 / p.nx                                            # :340 (LOCAL nx, App. B6)
    p.dtau = CFLtau * m                                          # :341
    p.err_scale_num = p.ly * p.ly                                # err = max*ly^2/psc   :466
    p.dtype = dtype
    return p


def _x_g(i1, d, size_a, n, coord):
    """ImplicitGlobalGrid x_g/y_g/z_g [upstream]: x0 = 0.5*(n-size(A))*d ; x = (coord*(n-2) + i-1)*d + x0."""
    x0 = 0.5 * (n - size_a) * d
    return (coord * (n - 2) + (i1 - 1)) * d + x0


def _alloc_multi(p):
    nx, ny, nz, dt = p.nx, p.ny, p.nz, p.dtype
    f = Obj()
    f.Pr = K.zeros((nx, ny, nz), dt)
    f.dPrdtau = K.zeros((nx - 2, ny - 2, nz - 2), dt)
    f.C = K.zeros((nx, ny, nz), dt); f.C_o = K.zeros((nx, ny, nz), dt)
    f.txx = K.zeros((nx, ny, nz), dt); f.tyy = K.zeros((nx, ny, nz), dt); f.tzz = K.zeros((nx, ny, nz), dt)
    f.txy = K.zeros((nx - 1, ny - 1, nz - 1), dt); f.txz = K.zeros((nx - 1, ny - 1, nz - 1), dt)
    f.tyz = K.zeros((nx - 1, ny - 1, nz - 1), dt)
    f.Vx = K.zeros((nx + 1, ny, nz), dt); f.Vy = K.zeros((nx, ny + 1, nz), dt); f.Vz = K.zeros((nx, ny, nz + 1), dt)
    f.Vx_o = K.zeros((nx + 1, ny, nz), dt); f.Vy_o = K.zeros((nx, ny + 1, nz), dt)
    f.Vz_o = K.zeros((nx, ny, nz + 1), dt)
    f.divV = K.zeros((nx, ny, nz), dt)
    f.Rp = K.zeros((nx - 2, ny - 2, nz - 2), dt)
    return f


def cart_coords(rank, dims):
    """MPI_Cart_coords of a row-major Cartesian communicator (last dimension fastest), as ImplicitGlobalGrid creates it."""
    return (rank // (dims[1] * dims[2]), (rank // dims[2]) % dims[1], rank % dims[2])


def cart_rank(c, dims):
    return (c[0] * dims[1] + c[1]) * dims[2] + c[2]


def update_halo_3d(ranks, name, ncells, dims):
    """update_halo!(A) of ImplicitGlobalGrid [upstream] for a Cartesian topology `dims`, halo width 1, overlap 2 (+ stagger):
    dimension by dimension (x, then y, then z — so that edge and corner values travel in two / three hops); in dimension d an
    array with local extent n_d+s has overlap ol=2+s, sends index `ol` (1-based) to the lower neighbour and `size-(ol-1)` to the
    upper one and receives into 1 / size; arrays with ol<2 have no halo in that dimension; physical (non-periodic) ends are
    left untouched.  ranks: list of field dicts in MPI rank order; ncells = (nx,ny,nz) local cell counts."""
    P = len(ranks)
    if P == 1:
        return
    for d in range(3):
        if dims[d] == 1:
            continue
        size = ranks[0][name].shape[d]
        ol = 2 + (size - ncells[d])
        if ol < 2:
            continue
        take = lambda A, i: np.take(A, i, axis=d).copy()
        # all sends are posted from the state BEFORE this dimension's exchange (Isend/Irecv then wait)
        to_lower = [take(r[name], ol - 1) for r in ranks]
        to_upper = [take(r[name], size - ol) for r in ranks]
        for rk, r in enumerate(ranks):
            c = cart_coords(rk, dims)
            idx = [slice(None)] * 3
            if c[d] > 0:
                lo = list(c); lo[d] -= 1
                idx[d] = 0
                r[name][tuple(idx)] = to_upper[cart_rank(lo, dims)]
            if c[d] < dims[d] - 1:
                hi = list(c); hi[d] += 1
                idx[d] = size - 1
                r[name][tuple(idx)] = to_lower[cart_rank(hi, dims)]


def update_halo_z(ranks, name, nz_cells):
    """z-slab form kept for the tests written against it: dims = (1,1,P)."""
    n = ranks[0][name].shape
    # only the z extent enters for dims (1,1,P)
    update_halo_3d(ranks, name, (n[0], n[1], nz_cells), (1, 1, len(ranks)))


def gather_3d(ranks, name, dims):
    """The *_inn / gather! contract of multi.jl:399-403,528-532 for a Cartesian topology: strip one cell on every side of
    each local array and place the rank blocks side by side in rank-coordinate order (block extent = local inner extent)."""
    blocks = [np.asarray(r[name][1:-1, 1:-1, 1:-1]) for r in ranks]
    return np.concatenate([np.concatenate([np.concatenate([blocks[cart_rank((cx, cy, cz), dims)] for cz in range(dims[2])], axis=2)
                                           for cy in range(dims[1])], axis=1) for cx in range(dims[0])], axis=0)


def gather_z(ranks, name):
    return gather_3d(ranks, name, (1, 1, len(ranks)))


def advect_wide_z(ranks, p, faithful):
    """The option outside the reference's multi-rank semantics (ns3d_advect_wide): {X_o .= X; advect!; update_halo!} with the old
    fields extended by ONE MORE plane per seam (the neighbour's plane sz-ol resp. ol+1), so that departure points up to two planes
    away read the neighbour instead of being clamped to the local array (multi.jl:192-195); all four new fields get their halo."""
    P = len(ranks)
    nz = ranks[0]["C"].shape[2]
    names = ("Vx", "Vy", "Vz", "C")
    ext_old = []
    for r, f in enumerate(ranks):
        e = {}
        for n in names:
            A = f[n]
            sz = A.shape[2]; ol = 2 + (sz - nz)
            parts = []
            if r > 0:
                parts.append(ranks[r - 1][n][:, :, sz - ol - 1:sz - ol])          # 1-based plane sz-ol of the lower neighbour
            parts.append(A)
            if r < P - 1:
                parts.append(ranks[r + 1][n][:, :, ol:ol + 1])                    # 1-based plane ol+1 of the upper neighbour
            e[n] = np.asfortranarray(np.concatenate(parts, axis=2))
        ext_old.append(e)
    for r, f in enumerate(ranks):
        for n in names:
            f[n + "_o"][...] = f[n]                                               # :475
    for r, f in enumerate(ranks):
        elo = 1 if r > 0 else 0
        old = ext_old[r]
        new = {n: old[n].copy(order="F") for n in names}
        K.advect_window(new["Vx"], old["Vx"], new["Vy"], old["Vy"], new["Vz"], old["Vz"], new["C"], old["C"], p.dt, p.dx, p.dy, p.dz,
                        faithful, r * (nz - 2) - elo, P * (nz - 2) + 2)
        for n in names:
            f[n][...] = new[n][:, :, elo:elo + f[n].shape[2]]
    for n in names:
        update_halo_3d(ranks, n, (ranks[0]["C"].shape[0], ranks[0]["C"].shape[1], nz), (1, 1, P))


def run_navierstokes3D_ref(nx=63, nt=1, dims_z=1, dtype=np.float64, faithful=True, niter_cap=None,
                           record=None, shape=None, dims=None, pressure="pt", wide_advect_halo=False):
    """multi.jl:287-536 without vis/save.  Returns (C_v,Pr_v,Vx_v,Vy_v,Vz_v, info) where info holds the
    per-step PT iteration counts and err histories, and the final local states."""
    p = multi_params(nx, dims_z, dtype, dims=dims, **(shape or {}))
    nx, ny, nz = p.nx, p.ny, p.nz
    niter = p.niter if niter_cap is None else min(p.niter, niter_cap)
    dims = p.dims
    P = dims[0] * dims[1] * dims[2]
    update_halo_z = lambda rks, name, _nz: update_halo_3d(rks, name, (nx, ny, nz), dims)      # noqa: F841 (shadows the z-only form)
    gather_z = lambda rks, name: gather_3d(rks, name, dims)                                     # noqa: F841
    ranks = []
    for rk in range(P):
        cx, cy, c = cart_coords(rk, dims)
        f = _alloc_multi(p)
        f.coord = c
        f.coords = (cx, cy, c)
        # multi.jl:363-367
        f.xco_g = _x_g(1, p.dx, nx, nx, cx) - (p.lx - p.dx) / 2
        f.yco_g = _x_g(1, p.dy, ny, ny, cy) - (p.ly - p.dy) / 2
        f.zco_g = _x_g(1, p.dz, nz, nz, c) - (p.lz - p.dz) / 2
        f.xvo_g = _x_g(1, p.dx, nx + 1, nx, cx) - (p.lx - p.dx) / 2
        f.xve_g = _x_g(nx + 1, p.dx, nx + 1, nx, cx) - (p.lx - p.dx) / 2
        f.owns_inlet = f.xvo_g == -p.lx / 2          # :164  (App. B10)
        f.owns_outlet = f.xve_g == p.lx / 2          # :179
        f.Vy[0, :, :] = p.vin                        # :369  (sic — App. B3)
        # :370  Pr = -(z_g-dz/2)*ρ*g + 0 + 0 == 0 because g == 0; kept as an explicit formula
        for iz in range(nz):
            f.Pr[:, :, iz] = -(_x_g(iz + 1, p.dz, nz, nz, c) - p.dz / 2) * p.rho * p.g + 0.0
        ranks.append(f)
    update_halo_z(ranks, "Pr", nz)                                                         # :371
    for f in ranks:                                                                        # :372
        K.set_cylinder(f.C, f.Vx, f.Vy, f.Vz, p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, f.xco_g, f.yco_g,
                       f.zco_g, p.lx, p.ly, p.lz, p.dx, p.dy, p.dz)
    for n in ("C", "Vx", "Vy", "Vz"):                                                      # :373
        update_halo_z(ranks, n, nz)

    info = Obj(iters=[], errs=[], params=p)
    for it in range(1, nt + 1):                                                            # :446
        for f in ranks:                                                                    # :449
            K.update_tau(f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, f.Vx, f.Vy, f.Vz, p.mu, p.dx, p.dy, p.dz)
        for n in ("txx", "tyy", "tzz"):                                                    # :450
            update_halo_z(ranks, n, nz)
        for f in ranks:                                                                    # :451-452
            K.predict_V(f.Vx, f.Vy, f.Vz, f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, p.rho, p.g, p.dt,
                        p.dx, p.dy, p.dz)
            K.set_cylinder(f.C, f.Vx, f.Vy, f.Vz, p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, f.xco_g, f.yco_g,
                           f.zco_g, p.lx, p.ly, p.lz, p.dx, p.dy, p.dz)
        for n in ("C", "Vx", "Vy", "Vz"):                                                  # :453
            update_halo_z(ranks, n, nz)
        for f in ranks:                                                                    # :454
            K.update_divV(f.divV, f.Vx, f.Vy, f.Vz, p.dx, p.dy, p.dz)
        update_halo_z(ranks, "divV", nz)                                                   # :455
        errs, iters_done = [], niter
        if pressure == "direct":        # the option outside parity (oracle/direct_ref.py), one rank
            from oracle.direct_ref import poisson_direct
            f = ranks[0]
            f.Pr[...] = poisson_direct(f.divV, p.rho, p.dt, p.dx, p.dy, p.dz, 0, f.owns_outlet, 0.0, p.g).astype(dtype)
            f.dPrdtau[...] = 0
            K.compute_res(f.Rp, f.Pr, f.divV, p.rho, p.dt, p.dx, p.dy, p.dz)
            iters_done, errs = 0, [K.max_abs(f.Rp) * p.err_scale_num / p.psc]
        elif P == 1:
            f = ranks[0]
            iters_done, errs = K.pt_solve(f.Pr, f.dPrdtau, f.divV, f.Rp, p.rho, p.dt, p.dtau, p.damp, p.dx,
                                          p.dy, p.dz, 0, f.owns_outlet, 0.0, p.g, p.eps, niter, p.nchk,
                                          p.err_scale_num, p.psc)
        else:
            for itr in range(1, niter + 1):                                                # :458
                for f in ranks:                                                            # :459
                    K.update_dPrdtau(f.Pr, f.dPrdtau, f.divV, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, p.dz)
                update_halo_z(ranks, "divV", nz)                                           # :460
                for f in ranks:                                                            # :461
                    K.update_Pr(f.Pr, f.dPrdtau, p.dtau)
                update_halo_z(ranks, "Pr", nz)                                             # :462
                for f in ranks:                                                            # :463 → :175-181
                    K.set_bc_Pr(f.Pr, 0, f.owns_outlet, 0.0)
                update_halo_z(ranks, "Pr", nz)                                             # :182
                if itr % p.nchk == 0:                                                      # :464
                    for f in ranks:
                        K.compute_res(f.Rp, f.Pr, f.divV, p.rho, p.dt, p.dx, p.dy, p.dz)   # :465
                    loc = [K.max_abs(f.Rp) for f in ranks]
                    mx = float("nan") if any(math.isnan(v) for v in loc) else max(loc)     # :21 max_g
                    err = mx * p.err_scale_num / p.psc                                     # :466
                    errs.append(err)
                    if err < p.eps or not math.isfinite(err):                              # :469
                        iters_done = itr
                        break
        info.iters.append(iters_done)
        info.errs.append(errs)
        for f in ranks:                                                                    # :472-474
            K.correct_V(f.Vx, f.Vy, f.Vz, f.Pr, p.dt, p.rho, p.dx, p.dy, p.dz)
            K.set_cylinder(f.C, f.Vx, f.Vy, f.Vz, p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, f.xco_g, f.yco_g,
                           f.zco_g, p.lx, p.ly, p.lz, p.dx, p.dy, p.dz)
            K.set_bc_Vel(f.Vx, f.Vy, f.Vz, 0, f.owns_inlet, p.vin)
        for n in ("Vx", "Vy", "Vz"):                                                       # :167
            update_halo_z(ranks, n, nz)
        if wide_advect_halo and P > 1:
            advect_wide_z(ranks, p, faithful)
        else:
            for f in ranks:                                                                # :475-476
                K.copy(f.Vx_o, f.Vx); K.copy(f.Vy_o, f.Vy); K.copy(f.Vz_o, f.Vz); K.copy(f.C_o, f.C)
                K.advect(f.Vx, f.Vx_o, f.Vy, f.Vy_o, f.Vz, f.Vz_o, f.C, f.C_o, p.dt, p.dx, p.dy, p.dz, faithful)
            for n in ("Vx", "Vy", "Vz"):                                                   # :477 (not C)
                update_halo_z(ranks, n, nz)
        if record is not None:
            record(it, ranks, info)
    info.ranks = ranks
    out = tuple(gather_z(ranks, n) for n in ("C", "Pr", "Vx", "Vy", "Vz"))                 # :528-535
    return out + (info,)


# ------------------------------------------------------------------------------------------------
# gpu.jl
# ------------------------------------------------------------------------------------------------
def gpu_params(nx=255, dtype=np.float64):
    """gpu.jl:15-61."""
    p = Obj()
    p.lx, p.rho, p.vin, p.mu = 1.0, 1000.0, 1.0, 0.001           # :15-18
    p.psc = p.rho * (p.vin * p.vin)                             # :21
    ly_lx, lz_lx, a_lx, b_lx, ox_lx, oy_lx = 0.6, 0.6, 0.05, 0.05, -0.3, 0.0   # :25-30
    beta = 0 * math.pi / 6
    p.ly, p.lz = ly_lx * p.lx, lz_lx * p.lx
    p.ox, p.oy = ox_lx * p.lx, oy_lx * p.lx
    p.g = 9.81                                                   # :38
    p.a2, p.b2 = (a_lx * p.lx) * (a_lx * p.lx), (b_lx * p.lx) * (b_lx * p.lx)
    p.sinb, p.cosb = math.sin(beta), math.cos(beta)
    p.nx = nx                                                    # :44 (255 in the script)
    p.ny = _ceil_int(nx * ly_lx)
    p.nz = _ceil_int(nx * lz_lx)
    p.eps = 1e-3
    p.niter = 50 * max(p.ny, p.nz)                               # :48
    p.nchk = 1 * (p.ny - 1)                                      # :49
    CFLtau, CFL_visc, CFL_adv = 1.0 / math.sqrt(3.1), 1 / 4.1, 1.0
    p.dx, p.dy, p.dz = p.lx / p.nx, p.ly / p.ny, p.lz / p.nz     # :58
    m = max(p.dx, p.dy, p.dz)
    p.dt = min(CFL_visc * (m * m) * p.rho / p.mu, CFL_adv * m / p.vin)   # :59
    p.damp = 2 / p.nx                                            # :60
    p.dtau = CFLtau * m                                          # :61
    p.dtype = dtype
    return p


def _linrange(a, b, n):
    """Julia LinRange(a,b,n)[i] = (1-t)*a + t*b with t=(i-1)/(n-1) [Base.lerpi]."""
    t = np.arange(n, dtype=np.float64) / (n - 1)
    return (1 - t) * a + t * b


def gpu_initial_fields(p):
    """gpu.jl:62-88 — ICs are host-side setup, outside the hot path; both the oracle and the HIP
    driver are fed these same arrays in parity tests."""
    nx, ny, nz = p.nx, p.ny, p.nz
    zc = _linrange(-(p.lz - p.dz) / 2, (p.lz - p.dz) / 2, nz)
    prof = p.vin * (7.0 / 6.0) * ((zc + p.lz / 2) / p.lz) ** (1.0 / 6.0)      # :85-86
    Vx = np.empty((nx + 1, ny, nz), dtype=np.float64, order="F")
    Vx[:, :, :] = prof[None, None, :]
    Pr = np.empty((nx, ny, nz), dtype=np.float64, order="F")
    Pr[:, :, :] = (-(zc - p.lz / 2) * p.rho * p.g)[None, None, :]            # :87
    return Vx, Pr


def runme_ref(nx=255, nt=1, dtype=np.float64, faithful=True, niter_cap=None, pressure="pt"):
    """gpu.jl:12-173 without vis/save; returns the final local fields + per-step PT info."""
    p = gpu_params(nx, dtype)
    nx, ny, nz = p.nx, p.ny, p.nz
    f = _alloc_multi(Obj(nx=nx, ny=ny, nz=nz, dtype=dtype))
    Vx0, Pr0 = gpu_initial_fields(p)
    f.Vx[...] = Vx0.astype(dtype); f.Pr[...] = Pr0.astype(dtype)
    niter = p.niter if niter_cap is None else min(p.niter, niter_cap)
    info = Obj(iters=[], errs=[], params=p)
    for it in range(1, nt + 1):                                                            # :119
        K.update_tau(f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, f.Vx, f.Vy, f.Vz, p.mu, p.dx, p.dy, p.dz)
        K.predict_V(f.Vx, f.Vy, f.Vz, f.txx, f.tyy, f.tzz, f.txy, f.txz, f.tyz, p.rho, p.g, p.dt, p.dx, p.dy, p.dz)
        K.set_cylinder_local(f.C, f.Vx, f.Vy, f.Vz, p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.lx, p.ly, p.lz,
                             p.dx, p.dy, p.dz)
        K.update_divV(f.divV, f.Vx, f.Vy, f.Vz, p.dx, p.dy, p.dz)
        if pressure == "direct":        # the option outside parity (oracle/direct_ref.py)
            from oracle.direct_ref import poisson_direct
            f.Pr[...] = poisson_direct(f.divV, p.rho, p.dt, p.dx, p.dy, p.dz, 1, False, 0.0, p.g).astype(dtype)
            f.dPrdtau[...] = 0
            K.compute_res(f.Rp, f.Pr, f.divV, p.rho, p.dt, p.dx, p.dy, p.dz)
            iters_done, errs = 0, [K.max_abs(f.Rp) * (p.ly * p.ly) / p.psc]
        else:
            iters_done, errs = K.pt_solve(f.Pr, f.dPrdtau, f.divV, f.Rp, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy,
                                          p.dz, 1, False, 0.0, p.g, p.eps, niter, p.nchk, p.ly * p.ly, p.psc)   # :126-137
        info.iters.append(iters_done); info.errs.append(errs)
        K.correct_V(f.Vx, f.Vy, f.Vz, f.Pr, p.dt, p.rho, p.dx, p.dy, p.dz)                 # :138
        K.set_cylinder_local(f.C, f.Vx, f.Vy, f.Vz, p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.lx, p.ly, p.lz,
                             p.dx, p.dy, p.dz)                                             # :139
        K.set_bc_Vel(f.Vx, f.Vy, f.Vz, 1)                                                  # :140
        K.copy(f.Vx_o, f.Vx); K.copy(f.Vy_o, f.Vy); K.copy(f.Vz_o, f.Vz); K.copy(f.C_o, f.C)   # :141
        K.advect(f.Vx, f.Vx_o, f.Vy, f.Vy_o, f.Vz, f.Vz_o, f.C, f.C_o, p.dt, p.dx, p.dy, p.dz, faithful)  # :142
    return f, info
