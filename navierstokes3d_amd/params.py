"""Parameter derivation of the two reference drivers (host scalars only).

multi_params  ↔ scripts/NavierStokes3D_multi_gpu.jl:290-341 (local grid nx×ny×nz per rank, global sizes through
                ImplicitGlobalGrid's n_g = dims·(n−2)+2; dims = (1,1,P) z-slabs by default, any topology with dims=/coords=)
gpu_params    ↔ scripts/NavierStokes3D_gpu.jl:15-61
cavity_params ↔ the synthetic 512³ "lid-driven cavity, Poisson-only" benchmark configuration
                (BASELINE.json configs[2]; SURVEY.md §8d Config 3 — not in the reference)

Expressions keep the reference's evaluation order so that the derived doubles are bit-identical to Julia's.
"""
import math
from types import SimpleNamespace


def _ceil_int(x):
    return int(math.ceil(x))


def multi_params(nx, dims_z=1, coord_z=0, ny=None, nz=None, ly_lx=0.6, lz_lx=0.6, dims=None, coords=None):
    """ny, nz, ly_lx, lz_lx: explicit-shape overrides of the literals multi.jl:302-303 and the ceil(0.6 nx) rule
    multi.jl:323-324 (defaults = the reference).  dims / coords: a Cartesian topology and this rank's MPI_Cart coordinates
    (override dims_z / coord_z)."""
    p = SimpleNamespace()
    p.lx, p.rho, p.vin, p.mu = 1.0, 1000.0, 1.0, 0.001                     # :290-293
    p.psc = p.rho * (p.vin * p.vin)                                        # :296
    Fr = math.inf                                                          # :301
    a_lx, b_lx = 0.05, 0.05                                                # :304-305
    ox_lx, oy_lx = -0.4, 0.0                                               # :307-308
    beta = 0 * math.pi / 6                                                 # :309
    p.ly, p.lz = ly_lx * p.lx, lz_lx * p.lx                                # :312-313
    p.ox, p.oy = ox_lx * p.lx, oy_lx * p.lx                                # :314-315
    p.g = 1 / (Fr * Fr) * (p.vin * p.vin) / p.lx                           # :316 → 0.0
    p.a2 = (a_lx * p.lx) * (a_lx * p.lx)                                   # :317
    p.b2 = (b_lx * p.lx) * (b_lx * p.lx)                                   # :318
    p.sinb, p.cosb = math.sin(beta), math.cos(beta)                        # :319
    p.nx = int(nx)
    p.ny = _ceil_int(nx * ly_lx) if ny is None else int(ny)                # :323
    p.nz = _ceil_int(nx * lz_lx) if nz is None else int(nz)                # :324
    p.dims = (1, 1, int(dims_z)) if dims is None else tuple(int(q) for q in dims)        # :325
    p.coords = (0, 0, int(coord_z)) if coords is None else tuple(int(q) for q in coords)
    p.nx_g = p.dims[0] * (p.nx - 2) + 2
    p.ny_g = p.dims[1] * (p.ny - 2) + 2
    p.nz_g = p.dims[2] * (p.nz - 2) + 2
    p.eps = 1e-3                                                           # :327
    p.niter = 50 * max(p.nx_g, p.ny_g, p.nz_g)                             # :328
    p.nchk = 1 * (p.ny_g - 1)                                              # :329
    CFLtau = 1.0 / math.sqrt(3.1)                                          # :333
    CFL_visc, CFL_adv = 1 / 4.1, 1.0                                       # :334-335
    p.dx, p.dy, p.dz = p.lx / p.nx_g, p.ly / p.ny_g, p.lz / p.nz_g         # :338
    m = max(p.dx, p.dy, p.dz)
    p.dt = min(CFL_visc * (m * m) * p.rho / p.mu, CFL_adv * m / p.vin)     # :339
    p.damp = 2 / p.nx                                                      # :340 (local nx)
    p.dtau = CFLtau * m                                                    # :341
    # global-coordinate origins, :363-367, with ImplicitGlobalGrid's x_g [upstream]:
    #   x_g(ix,dx,A) = (coord*(n-2) + ix-1)*dx + 0.5*(n-size(A))*dx
    def x_g(i1, d, size_a, n, coord):
        return (coord * (n - 2) + (i1 - 1)) * d + 0.5 * (n - size_a) * d
    p.xco_g = x_g(1, p.dx, p.nx, p.nx, p.coords[0]) - (p.lx - p.dx) / 2
    p.yco_g = x_g(1, p.dy, p.ny, p.ny, p.coords[1]) - (p.ly - p.dy) / 2
    p.zco_g = x_g(1, p.dz, p.nz, p.nz, p.coords[2]) - (p.lz - p.dz) / 2
    p.xvo_g = x_g(1, p.dx, p.nx + 1, p.nx, p.coords[0]) - (p.lx - p.dx) / 2
    p.xve_g = x_g(p.nx + 1, p.dx, p.nx + 1, p.nx, p.coords[0]) - (p.lx - p.dx) / 2
    p.owns_inlet = p.xvo_g == -p.lx / 2                                    # :164
    p.owns_outlet = p.xve_g == p.lx / 2                                    # :179
    return p


def gpu_params(nx=255):
    p = SimpleNamespace()
    p.lx, p.rho, p.vin, p.mu = 1.0, 1000.0, 1.0, 0.001                     # gpu.jl:15-18
    p.psc = p.rho * (p.vin * p.vin)                                        # :21
    ly_lx, lz_lx, a_lx, b_lx, ox_lx, oy_lx = 0.6, 0.6, 0.05, 0.05, -0.3, 0.0   # :25-30
    beta = 0 * math.pi / 6                                                 # :31
    p.ly, p.lz = ly_lx * p.lx, lz_lx * p.lx                                # :34-35
    p.ox, p.oy = ox_lx * p.lx, oy_lx * p.lx                                # :36-37
    p.g = 9.81                                                             # :38
    p.a2 = (a_lx * p.lx) * (a_lx * p.lx)                                   # :39
    p.b2 = (b_lx * p.lx) * (b_lx * p.lx)                                   # :40
    p.sinb, p.cosb = math.sin(beta), math.cos(beta)                        # :41
    p.nx = int(nx)                                                         # :44 (255)
    p.ny = _ceil_int(nx * ly_lx)                                           # :45
    p.nz = _ceil_int(nx * lz_lx)                                           # :46
    p.eps = 1e-3                                                           # :47
    p.niter = 50 * max(p.ny, p.nz)                                         # :48
    p.nchk = 1 * (p.ny - 1)                                                # :49
    CFLtau, CFL_visc, CFL_adv = 1.0 / math.sqrt(3.1), 1 / 4.1, 1.0         # :53-55
    p.dx, p.dy, p.dz = p.lx / p.nx, p.ly / p.ny, p.lz / p.nz               # :58
    m = max(p.dx, p.dy, p.dz)
    p.dt = min(CFL_visc * (m * m) * p.rho / p.mu, CFL_adv * m / p.vin)     # :59
    p.damp = 2 / p.nx                                                      # :60
    p.dtau = CFLtau * m                                                    # :61
    return p


def cavity_params(n=512, nz=None):
    """Synthetic Poisson-only benchmark grid: unit box, no obstacle, g=0, all-Neumann pressure BCs
    (multi.jl's set_bc_Pr! without the outlet plane), dx=1/n, dt=dx, dτ=dx/√3.1, damp=2/n."""
    p = SimpleNamespace()
    p.nx = p.ny = int(n)
    p.nz = int(nz if nz is not None else n)
    p.lx = p.ly = 1.0
    p.rho, p.mu, p.vin = 1000.0, 1e-3, 1.0
    p.dx, p.dy = p.lx / p.nx, p.ly / p.ny
    p.dz = p.dx
    p.lz = p.dz * p.nz
    p.dt = p.dx
    p.dtau = p.dx / math.sqrt(3.1)
    p.damp = 2 / p.nx
    p.g = 0.0
    p.psc = p.rho * p.vin * p.vin
    return p
