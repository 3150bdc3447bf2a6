"""TEST INFRASTRUCTURE (oracle/): a NumPy restatement of the direct pressure solve of navierstokes3d_amd/csrc/ns3d_direct.hip —
the OPTION OUTSIDE PARITY of SURVEY.md §8 f4 — so that the HIP version stays testable: the discrete problem the reference's
pseudo-transient loop iterates towards (multi.jl:458-471 with set_bc_Pr! multi.jl:175-181; gpu.jl:126-137 with gpu.jl:281-286),

    ∇²_h Pr = ρ/dt·∇V  on the interior cells,   boundary cells as set_bc_Pr! leaves them,

solved by exact diagonalisation: per direction the second-difference operator with its boundary rule has closed-form
eigenvectors (copy | copy: cosines; copy | zero cell: shifted cosines; zero | zero: sines).  The reference has no such solver;
what pins this file is the reference's own residual definition (compute_res!, multi.jl:88-91, through the C oracle): the result
must zero it to rounding, and the oracle's PT loop run to a tight tolerance must converge to it (tests/test_oracle.py)."""
import numpy as np


def eig1d(m, d, kind):
    """columns of V: orthonormal eigenvectors of the 1-D second difference on m interior cells; lam: eigenvalues.
    kind 0: copies on both ends; 1: copy below, zero cell above; 2: zero cells on both ends."""
    i = np.arange(m, dtype=np.longdouble)
    q = np.arange(m, dtype=np.longdouble)
    pi = np.longdouble(np.pi) if np.finfo(np.longdouble).eps >= 1e-16 else np.arccos(np.longdouble(-1))
    if kind == 0:
        th = pi * q / m
        V = np.cos(np.outer(i + 0.5, th))
    elif kind == 1:
        th = pi * (q + 0.5) / (m + np.longdouble(0.5))
        V = np.cos(np.outer(i + 0.5, th))
    else:
        th = pi * (q + 1) / (m + 1)
        V = np.sin(np.outer(i + 1, th))
    V = V / np.sqrt((V * V).sum(axis=0))[None, :]
    lam = -4 * np.sin(th / 2) ** 2 / (np.longdouble(d) ** 2)
    if kind == 0:
        lam[0] = 0
    return V.astype(np.float64), lam.astype(np.float64)


def poisson_direct(divV, rho, dt, dx, dy, dz, bc_kind=0, owns_outlet=True, outlet_val=0.0, g=0.0):
    """Returns Pr (nx,ny,nz) with the interior solved and the boundary cells of set_bc_Pr! (bc_kind 0: multi.jl, 1: gpu.jl)."""
    nx, ny, nz = divV.shape
    mx, my, mz = nx - 2, ny - 2, nz - 2
    F = (rho / dt) * np.asarray(divV[1:-1, 1:-1, 1:-1], dtype=np.float64).copy()
    hyd = None
    if bc_kind == 1:                                                   # gpu.jl:258-259, 1-based iz = plane + 1
        hyd = np.array([(rho * g * ((nz - (k + 1)) + 0.5)) * dz for k in range(nz)])
        F[0, :, :] -= (hyd[1:-1] + 100.0)[None, :] / (dx * dx)
        F[-1, :, :] -= hyd[1:-1][None, :] / (dx * dx)
        xkind = 2
    elif owns_outlet:                                                  # multi.jl:179-180
        F[-1, :, :] -= outlet_val / (dx * dx)
        xkind = 1
    else:
        xkind = 0
    Vx, lx = eig1d(mx, dx, xkind)
    Vy, ly = eig1d(my, dy, 0)
    Vz, lz = eig1d(mz, dz, 0)
    U = np.einsum("ia,ijk->ajk", Vx, F)
    U = np.einsum("jb,ajk->abk", Vy, U)
    U = np.einsum("kc,abk->abc", Vz, U)
    lam = lx[:, None, None] + ly[None, :, None] + lz[None, None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        U = np.where(lam != 0.0, U / lam, 0.0)
    U = np.einsum("kc,abc->abk", Vz, U)
    U = np.einsum("jb,abk->ajk", Vy, U)
    U = np.einsum("ia,ajk->ijk", Vx, U)
    Pr = np.zeros((nx, ny, nz), order="F")
    Pr[1:-1, 1:-1, 1:-1] = U
    if bc_kind == 1:                                                   # gpu.jl:282-284: bc_y!, bc_z!, bc_xhydstatic!
        Pr[:, 0, :] = Pr[:, 1, :]; Pr[:, -1, :] = Pr[:, -2, :]
        Pr[:, :, 0] = Pr[:, :, 1]; Pr[:, :, -1] = Pr[:, :, -2]
        Pr[0, :, :] = (hyd + 100.0)[None, :]; Pr[-1, :, :] = hyd[None, :]
    else:                                                              # multi.jl:176-181: bc_x!, bc_y!, bc_z!, outlet
        Pr[0, :, :] = Pr[1, :, :]; Pr[-1, :, :] = Pr[-2, :, :]
        Pr[:, 0, :] = Pr[:, 1, :]; Pr[:, -1, :] = Pr[:, -2, :]
        Pr[:, :, 0] = Pr[:, :, 1]; Pr[:, :, -1] = Pr[:, :, -2]
        if owns_outlet:
            Pr[-1, :, :] = outlet_val
    return np.asfortranarray(Pr)
