"""Independent NumPy transcription of the reference kernels.  TEST INFRASTRUCTURE ONLY.

Purpose: a second, structurally different restatement (whole-array slices instead of index loops) of
scripts/NavierStokes3D_multi_gpu.jl:15-281 / scripts/NavierStokes3D_gpu.jl:175-368 that the C oracle
(oracle/ns3d_oracle.c) must match BIT FOR BIT.  NumPy evaluates one ufunc per operation (no FMA
contraction), so expression trees written in the Julia order give Julia's roundings.

Slices below are written with the FiniteDifferences3D meanings ([upstream]):
  @all(A)=A   @inn(A)=A[1:-1,1:-1,1:-1]   @d_xa(A)=A[1:]-A[:-1]   @d_xi(A)=@d_xa on A[:,1:-1,1:-1] …
each statement restricted to the extents of the array it assigns (the @parallel bounds guard).
"""
import math

import numpy as np


def _divV(Vx, Vy, Vz, dx, dy, dz):                                  # multi.jl:15
    return ((Vx[1:, :, :] - Vx[:-1, :, :]) / dx + (Vy[:, 1:, :] - Vy[:, :-1, :]) / dy) + (Vz[:, :, 1:] - Vz[:, :, :-1]) / dz


def update_tau(txx, tyy, tzz, txy, txz, tyz, Vx, Vy, Vz, mu, dx, dy, dz):     # multi.jl:36-44
    div = _divV(Vx, Vy, Vz, dx, dy, dz)
    txx[...] = (2 * mu) * ((Vx[1:, :, :] - Vx[:-1, :, :]) / dx - div / 3.0)
    tyy[...] = (2 * mu) * ((Vy[:, 1:, :] - Vy[:, :-1, :]) / dy - div / 3.0)
    tzz[...] = (2 * mu) * ((Vz[:, :, 1:] - Vz[:, :, :-1]) / dz - div / 3.0)
    nx, ny, nz = txx.shape
    # @d_yi(Vx) = Vx[ix+1,iy+1,iz+1]-Vx[ix+1,iy,iz+1]  for ix<=nx-1, iy<=ny-1, iz<=nz-1
    d_yi_Vx = Vx[1:nx, 1:ny, 1:nz] - Vx[1:nx, 0:ny - 1, 1:nz]
    d_xi_Vy = Vy[1:nx, 1:ny, 1:nz] - Vy[0:nx - 1, 1:ny, 1:nz]
    d_zi_Vx = Vx[1:nx, 1:ny, 1:nz] - Vx[1:nx, 1:ny, 0:nz - 1]
    d_xi_Vz = Vz[1:nx, 1:ny, 1:nz] - Vz[0:nx - 1, 1:ny, 1:nz]
    d_zi_Vy = Vy[1:nx, 1:ny, 1:nz] - Vy[1:nx, 1:ny, 0:nz - 1]
    d_yi_Vz = Vz[1:nx, 1:ny, 1:nz] - Vz[1:nx, 0:ny - 1, 1:nz]
    txy[...] = mu * (d_yi_Vx / dy + d_xi_Vy / dx)
    txz[...] = mu * (d_zi_Vx / dz + d_xi_Vz / dx)
    tyz[...] = mu * (d_zi_Vy / dz + d_yi_Vz / dy)


def predict_V(Vx, Vy, Vz, txx, tyy, tzz, txy, txz, tyz, rho, g, dt, dx, dy, dz):     # multi.jl:50-55
    nx, ny, nz = txx.shape
    # @inn(Vx): (nx-1, ny-2, nz-2)
    a = (txx[1:nx, 1:ny - 1, 1:nz - 1] - txx[0:nx - 1, 1:ny - 1, 1:nz - 1]) / dx            # @d_xi(τxx)
    b = (txy[0:nx - 1, 1:ny - 1, 0:nz - 2] - txy[0:nx - 1, 0:ny - 2, 0:nz - 2]) / dy        # @d_ya(τxy)
    c = (txz[0:nx - 1, 0:ny - 2, 1:nz - 1] - txz[0:nx - 1, 0:ny - 2, 0:nz - 2]) / dz        # @d_za(τxz)
    Vx[1:-1, 1:-1, 1:-1] = Vx[1:-1, 1:-1, 1:-1] + dt / rho * ((a + b) + c)
    a = (tyy[1:nx - 1, 1:ny, 1:nz - 1] - tyy[1:nx - 1, 0:ny - 1, 1:nz - 1]) / dy            # @d_yi(τyy)
    b = (txy[1:nx - 1, 0:ny - 1, 0:nz - 2] - txy[0:nx - 2, 0:ny - 1, 0:nz - 2]) / dx        # @d_xa(τxy)
    c = (tyz[0:nx - 2, 0:ny - 1, 1:nz - 1] - tyz[0:nx - 2, 0:ny - 1, 0:nz - 2]) / dz        # @d_za(τyz)
    Vy[1:-1, 1:-1, 1:-1] = Vy[1:-1, 1:-1, 1:-1] + dt / rho * ((a + b) + c)
    a = (tzz[1:nx - 1, 1:ny - 1, 1:nz] - tzz[1:nx - 1, 1:ny - 1, 0:nz - 1]) / dz            # @d_zi(τzz)
    b = (txz[1:nx - 1, 0:ny - 2, 0:nz - 1] - txz[0:nx - 2, 0:ny - 2, 0:nz - 1]) / dx        # @d_xa(τxz)
    c = (tyz[0:nx - 2, 1:ny - 1, 0:nz - 1] - tyz[0:nx - 2, 0:ny - 2, 0:nz - 1]) / dy        # @d_ya(τyz)
    Vz[1:-1, 1:-1, 1:-1] = Vz[1:-1, 1:-1, 1:-1] + dt / rho * (((a + b) + c) - rho * g)


def update_divV(divV, Vx, Vy, Vz, dx, dy, dz):                     # multi.jl:61-64
    divV[...] = _divV(Vx, Vy, Vz, dx, dy, dz)


def _res(Pr, divV, rho, dt, dx, dy, dz):                            # multi.jl:71,89 right-hand side
    c = Pr[1:-1, 1:-1, 1:-1]
    d2x = (Pr[2:, 1:-1, 1:-1] - c) - (c - Pr[:-2, 1:-1, 1:-1])
    d2y = (Pr[1:-1, 2:, 1:-1] - c) - (c - Pr[1:-1, :-2, 1:-1])
    d2z = (Pr[1:-1, 1:-1, 2:] - c) - (c - Pr[1:-1, 1:-1, :-2])
    return ((d2x / dx / dx + d2y / dy / dy) + d2z / dz / dz) - rho / dt * divV[1:-1, 1:-1, 1:-1]


def update_dPrdtau(Pr, dPrdtau, divV, rho, dt, dtau, damp, dx, dy, dz):    # multi.jl:70-73
    dPrdtau[...] = dPrdtau * (1.0 - damp) + dtau * _res(Pr, divV, rho, dt, dx, dy, dz)


def update_Pr(Pr, dPrdtau, dtau):                                   # multi.jl:79-82
    Pr[1:-1, 1:-1, 1:-1] = Pr[1:-1, 1:-1, 1:-1] + dtau * dPrdtau


def compute_res(Rp, Pr, divV, rho, dt, dx, dy, dz):                 # multi.jl:88-91
    Rp[...] = _res(Pr, divV, rho, dt, dx, dy, dz)


def correct_V(Vx, Vy, Vz, Pr, dt, rho, dx, dy, dz):                 # multi.jl:97-102
    Vx[1:-1, 1:-1, 1:-1] = Vx[1:-1, 1:-1, 1:-1] - dt / rho * (Pr[1:, 1:-1, 1:-1] - Pr[:-1, 1:-1, 1:-1]) / dx
    Vy[1:-1, 1:-1, 1:-1] = Vy[1:-1, 1:-1, 1:-1] - dt / rho * (Pr[1:-1, 1:, 1:-1] - Pr[1:-1, :-1, 1:-1]) / dy
    Vz[1:-1, 1:-1, 1:-1] = Vz[1:-1, 1:-1, 1:-1] - dt / rho * (Pr[1:-1, 1:-1, 1:] - Pr[1:-1, 1:-1, :-1]) / dz


def bc_x(A):                                                        # multi.jl:108-112
    A[0, :, :] = A[1, :, :]; A[-1, :, :] = A[-2, :, :]


def bc_y(A):                                                        # multi.jl:118-122
    A[:, 0, :] = A[:, 1, :]; A[:, -1, :] = A[:, -2, :]


def bc_z(A):                                                        # multi.jl:128-132
    A[:, :, 0] = A[:, :, 1]; A[:, :, -1] = A[:, :, -2]


def bc_zV(A):                                                       # gpu.jl:239-243
    A[:, :, 0] = 0.0; A[:, :, -1] = A[:, :, -2]


def bc_xhydstatic(A, dz, nz, g, rho):                               # gpu.jl:257-261
    iz = np.arange(1, A.shape[2] + 1)
    h = rho * g * (nz - iz + 0.5) * dz
    A[0, :, :] = (h + 100)[None, :]
    A[-1, :, :] = h[None, :]


def bc_x_Vx(A, V):                                                  # multi.jl:138-141
    A[0, :, :] = V


def bc_x_Pr(A, val):                                                # multi.jl:147-150
    A[-1, :, :] = val


def _mask(xq, yq, ox, oy, sinb, cosb, a2, b2, thr):
    xr = (xq - ox) * cosb - (yq - oy) * sinb
    yr = (xq - ox) * sinb + (yq - oy) * cosb
    return xr * xr / a2 + yr * yr / b2 < thr


def _apply_cyl(C, Vx, Vy, Vz, xc, yc, xv, yv, a2, b2, ox, oy, sinb, cosb):
    nx, ny, nz = C.shape
    XC, YC = xc[:, None], yc[None, :]
    XV, YV = xv[:, None], yv[None, :]
    m = _mask(XC[:nx], YC[:, :ny], ox, oy, sinb, cosb, a2, b2, 1.05)
    C[m, :] = 1.0
    m = _mask(XV[:nx + 1], YC[:, :ny], ox, oy, sinb, cosb, a2, b2, 1.0)
    Vx[m, :] = 0.0
    m = _mask(XC[:nx], YV[:, :ny + 1], ox, oy, sinb, cosb, a2, b2, 1.0)
    Vy[m, :] = 0.0
    m = _mask(XC[:nx], YC[:, :ny], ox, oy, sinb, cosb, a2, b2, 1.0)
    Vz[m, :] = 0.0


def set_cylinder(C, Vx, Vy, Vz, a2, b2, ox, oy, sinb, cosb, xco_g, yco_g, zco_g, lx, ly, lz, dx, dy, dz):
    nx, ny, nz = C.shape                                            # multi.jl:249-281
    xc = xco_g + np.arange(nx + 1) * dx
    yc = yco_g + np.arange(ny + 1) * dy
    _apply_cyl(C, Vx, Vy, Vz, xc, yc, xc - dx / 2, yc - dy / 2, a2, b2, ox, oy, sinb, cosb)


def set_cylinder_local(C, Vx, Vy, Vz, a2, b2, ox, oy, sinb, cosb, lx, ly, lz, dx, dy, dz):
    nx, ny, nz = C.shape                                            # gpu.jl:336-368
    xv = np.arange(nx + 1) * dx - lx / 2
    yv = np.arange(ny + 1) * dy - ly / 2
    _apply_cyl(C, Vx, Vy, Vz, xv + dx / 2, yv + dx / 2, xv, yv, a2, b2, ox, oy, sinb, cosb)   # dx/2: sic


def _lerp(a, b, t):                                                 # multi.jl:211
    return b * t + a * (1 - t)


def _backtrack(A, A_o, vxc, vyc, vzc, dt, dx, dy, dz, IX, IY, IZ):  # multi.jl:190-205 (1-based index grids)
    sx, sy, sz = A.shape
    ddx, ddy, ddz = dt * vxc / dx, dt * vyc / dy, dt * vzc / dz
    ix1 = np.clip(np.floor(IX - ddx).astype(np.int64), 1, sx)
    iy1 = np.clip(np.floor(IY - ddy).astype(np.int64), 1, sy)
    iz1 = np.clip(np.floor(IZ - ddz).astype(np.int64), 1, sz)
    ix2, iy2, iz2 = np.clip(ix1 + 1, 1, sx), np.clip(iy1 + 1, 1, sy), np.clip(iz1 + 1, 1, sz)
    wx = (ddx > 0).astype(A.dtype) - np.fmod(ddx, 1)
    wy = (ddy > 0).astype(A.dtype) - np.fmod(ddy, 1)
    wz = (ddz > 0).astype(A.dtype) - np.fmod(ddz, 1)
    g = lambda i, j, k: A_o[i - 1, j - 1, k - 1]
    fy1z1 = _lerp(g(ix1, iy1, iz1), g(ix2, iy1, iz1), wx)
    fy1z2 = _lerp(g(ix1, iy1, iz2), g(ix2, iy1, iz2), wx)
    fy2z1 = _lerp(g(ix1, iy2, iz1), g(ix2, iy2, iz1), wx)
    fy2z2 = _lerp(g(ix1, iy2, iz2), g(ix2, iy2, iz2), wx)
    fz1 = _lerp(fy1z1, fy2z1, wy)
    fz2 = _lerp(fy1z2, fy2z2, wy)
    A[IX - 1, IY - 1, IZ - 1] = _lerp(fz1, fz2, wz)


def advect(Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, C, C_o, dt, dx, dy, dz, faithful=True):   # multi.jl:217-243
    nx, ny, nz = C.shape
    def grid(xs, ys, zs):
        return np.meshgrid(np.asarray(xs), np.asarray(ys), np.asarray(zs), indexing="ij")
    o = lambda A, i, j, k: A[i - 1, j - 1, k - 1]
    # branch 1: ix in 2..nx, iy<=ny, iz<=nz
    IX, IY, IZ = grid(range(2, nx + 1), range(1, ny + 1), range(1, nz + 1))
    vxc = o(Vx_o, IX, IY, IZ)
    vyc = 0.25 * (((o(Vy_o, IX - 1, IY, IZ) + o(Vy_o, IX - 1, IY + 1, IZ)) + o(Vy_o, IX, IY, IZ)) + o(Vy_o, IX, IY + 1, IZ))
    vzc = 0.25 * (((o(Vz_o, IX - 1, IY, IZ) + o(Vz_o, IX - 1, IY, IZ + 1)) + o(Vz_o, IX, IY, IZ)) + o(Vz_o, IX, IY, IZ + 1))
    _backtrack(Vx, Vx_o, vxc, vyc, vzc, dt, dx, dy, dz, IX, IY, IZ)
    # branch 2: iy in 2..ny
    IX, IY, IZ = grid(range(1, nx + 1), range(2, ny + 1), range(1, nz + 1))
    vxc = 0.25 * (((o(Vx_o, IX, IY - 1, IZ) + o(Vx_o, IX + 1, IY - 1, IZ)) + o(Vx_o, IX, IY, IZ)) + o(Vx_o, IX + 1, IY, IZ))
    vyc = o(Vy_o, IX, IY, IZ)
    vzc = 0.25 * (((o(Vz_o, IX, IY - 1, IZ) + o(Vz_o, IX, IY - 1, IZ + 1)) + o(Vz_o, IX, IY, IZ)) + o(Vz_o, IX, IY, IZ + 1))
    _backtrack(Vy, Vy_o, vxc, vyc, vzc, dt, dx, dy, dz, IX, IY, IZ)
    # branch 3: iz in 2..nz — the reference back-tracks Vy here (multi.jl:234), executed after branch 2
    IX, IY, IZ = grid(range(1, nx + 1), range(1, ny + 1), range(2, nz + 1))
    vxc = 0.25 * (((o(Vx_o, IX, IY, IZ - 1) + o(Vx_o, IX + 1, IY, IZ - 1)) + o(Vx_o, IX, IY, IZ)) + o(Vx_o, IX + 1, IY, IZ))
    vyc = 0.25 * (((o(Vy_o, IX, IY, IZ - 1) + o(Vy_o, IX, IY + 1, IZ - 1)) + o(Vy_o, IX, IY, IZ)) + o(Vy_o, IX, IY + 1, IZ))
    vzc = o(Vz_o, IX, IY, IZ)
    if faithful:
        _backtrack(Vy, Vy_o, vxc, vyc, vzc, dt, dx, dy, dz, IX, IY, IZ)
    else:
        _backtrack(Vz, Vz_o, vxc, vyc, vzc, dt, dx, dy, dz, IX, IY, IZ)
    # branch 4: C
    IX, IY, IZ = grid(range(1, nx + 1), range(1, ny + 1), range(1, nz + 1))
    vxc = 0.5 * (o(Vx_o, IX, IY, IZ) + o(Vx_o, IX + 1, IY, IZ))
    vyc = 0.5 * (o(Vy_o, IX, IY, IZ) + o(Vy_o, IX, IY + 1, IZ))
    vzc = 0.5 * (o(Vz_o, IX, IY, IZ) + o(Vz_o, IX, IY, IZ + 1))
    _backtrack(C, C_o, vxc, vyc, vzc, dt, dx, dy, dz, IX, IY, IZ)


def max_abs(A):                                                     # multi.jl:466 — NaN-propagating
    return float(np.max(np.abs(A)))


def set_bc_Pr_multi(Pr, owns_outlet, val=0.0):                      # multi.jl:175-181
    bc_x(Pr); bc_y(Pr); bc_z(Pr)
    if owns_outlet:
        bc_x_Pr(Pr, val)


def set_bc_Pr_gpu(Pr, dz, nz, g, rho):                              # gpu.jl:281-286
    bc_y(Pr); bc_z(Pr); bc_xhydstatic(Pr, dz, nz, g, rho)


def set_bc_Vel_multi(Vx, Vy, Vz, owns_inlet, vin):                  # multi.jl:156-166
    bc_x(Vx); bc_y(Vx); bc_z(Vx); bc_x(Vy); bc_z(Vy); bc_x(Vz); bc_y(Vz)
    if owns_inlet:
        bc_x_Vx(Vx, vin)


def set_bc_Vel_gpu(Vx, Vy, Vz):                                     # gpu.jl:264-279
    bc_x(Vx); bc_y(Vx); bc_zV(Vx); bc_x(Vy); bc_y(Vy); bc_zV(Vy); bc_x(Vz); bc_y(Vz); bc_zV(Vz)


def run_multi_1rank(nx, nt):
    """multi.jl:287-536 on one rank, all NumPy (slow: use nx<=24). Returns local fields dict + iteration counts."""
    lx, rho, vin, mu = 1.0, 1000.0, 1.0, 0.001
    psc = rho * (vin * vin)
    ly, lz = 0.6 * lx, 0.6 * lx
    ox, oy = -0.4 * lx, 0.0 * lx
    g = 1 / (math.inf * math.inf) * (vin * vin) / lx
    a2, b2 = (0.05 * lx) * (0.05 * lx), (0.05 * lx) * (0.05 * lx)
    sinb, cosb = math.sin(0.0), math.cos(0.0)
    ny, nz = int(math.ceil(nx * 0.6)), int(math.ceil(nx * 0.6))
    niter, nchk, eps = 50 * max(nx, ny, nz), ny - 1, 1e-3
    dx, dy, dz = lx / nx, ly / ny, lz / nz
    m = max(dx, dy, dz)
    dt = min(1 / 4.1 * (m * m) * rho / mu, 1.0 * m / vin)
    damp, dtau = 2 / nx, 1.0 / math.sqrt(3.1) * m
    Z = lambda *s: np.zeros(s, order="F")
    Pr, dPrdtau, C, C_o = Z(nx, ny, nz), Z(nx - 2, ny - 2, nz - 2), Z(nx, ny, nz), Z(nx, ny, nz)
    txx, tyy, tzz = Z(nx, ny, nz), Z(nx, ny, nz), Z(nx, ny, nz)
    txy, txz, tyz = Z(nx - 1, ny - 1, nz - 1), Z(nx - 1, ny - 1, nz - 1), Z(nx - 1, ny - 1, nz - 1)
    Vx, Vy, Vz = Z(nx + 1, ny, nz), Z(nx, ny + 1, nz), Z(nx, ny, nz + 1)
    Vx_o, Vy_o, Vz_o = Z(nx + 1, ny, nz), Z(nx, ny + 1, nz), Z(nx, ny, nz + 1)
    divV, Rp = Z(nx, ny, nz), Z(nx - 2, ny - 2, nz - 2)
    xco_g, yco_g, zco_g = 0.0 - (lx - dx) / 2, 0.0 - (ly - dy) / 2, 0.0 - (lz - dz) / 2
    Vy[0, :, :] = vin
    cyl = lambda: set_cylinder(C, Vx, Vy, Vz, a2, b2, ox, oy, sinb, cosb, xco_g, yco_g, zco_g, lx, ly, lz, dx, dy, dz)
    cyl()
    iters = []
    for it in range(nt):
        update_tau(txx, tyy, tzz, txy, txz, tyz, Vx, Vy, Vz, mu, dx, dy, dz)
        predict_V(Vx, Vy, Vz, txx, tyy, tzz, txy, txz, tyz, rho, g, dt, dx, dy, dz)
        cyl()
        update_divV(divV, Vx, Vy, Vz, dx, dy, dz)
        done = niter
        for itr in range(1, niter + 1):
            update_dPrdtau(Pr, dPrdtau, divV, rho, dt, dtau, damp, dx, dy, dz)
            update_Pr(Pr, dPrdtau, dtau)
            set_bc_Pr_multi(Pr, True, 0.0)
            if itr % nchk == 0:
                compute_res(Rp, Pr, divV, rho, dt, dx, dy, dz)
                err = max_abs(Rp) * (ly * ly) / psc
                if err < eps or not math.isfinite(err):
                    done = itr
                    break
        iters.append(done)
        correct_V(Vx, Vy, Vz, Pr, dt, rho, dx, dy, dz)
        cyl()
        set_bc_Vel_multi(Vx, Vy, Vz, True, vin)
        Vx_o[...] = Vx; Vy_o[...] = Vy; Vz_o[...] = Vz; C_o[...] = C
        advect(Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, C, C_o, dt, dx, dy, dz)
    return dict(Pr=Pr, C=C, Vx=Vx, Vy=Vy, Vz=Vz, dPrdtau=dPrdtau, divV=divV), iters
