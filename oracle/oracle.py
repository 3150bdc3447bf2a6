"""ctypes binding of the CPU oracle (oracle/ns3d_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing
under navierstokes3d_amd/ does.  PARITY UNPINNED (see ns3d_oracle.c header): the reference is Julia and
cannot run here, its only known-answer test is stale.

Arrays are numpy, Fortran order (column-major, x fastest) with exactly the reference shapes
(scripts/NavierStokes3D_multi_gpu.jl:343-360).  Function names/argument order follow the reference
kernels (multi.jl:36-281); the cell grid (nx,ny,nz) is derived from the array shapes like
ParallelStencil derives the launch range from its arguments.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    if os.environ.get("NS3D_ORACLE_LIB"):          # e.g. an ASan/UBSan build of ns3d_oracle.c (CPU sanitizer runs)
        return os.environ["NS3D_ORACLE_LIB"]
    so = os.path.join(_HERE, "libns3d_oracle.so")
    src = os.path.join(_HERE, "ns3d_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        for suf in ("f64", "f32"):
            getattr(_LIB, "ns3d_ref_max_abs_" + suf).restype = C.c_double
            getattr(_LIB, "ns3d_ref_pt_solve_" + suf).restype = C.c_int
    return _LIB


def _suf(a):
    if a.dtype == np.float64:
        return "f64"
    if a.dtype == np.float32:
        return "f32"
    raise TypeError(a.dtype)


def _p(a):
    assert a.flags.f_contiguous, "oracle arrays must be column-major (order='F')"
    return a.ctypes.data_as(C.c_void_p)


def _d(*xs):
    return [C.c_double(float(x)) for x in xs]


def _i(*xs):
    return [C.c_int(int(x)) for x in xs]


def _call(name, ref, *args):
    return getattr(lib(), "ns3d_ref_%s_%s" % (name, _suf(ref)))(*args)


def zeros(shape, dtype=np.float64):
    return np.zeros(shape, dtype=dtype, order="F")


# ---- kernels (reference names; `!` dropped, τ→tau, ∇V→divV) --------------------------------------
def update_tau(txx, tyy, tzz, txy, txz, tyz, Vx, Vy, Vz, mu, dx, dy, dz):
    nx, ny, nz = txx.shape
    _call("update_tau", txx, _p(txx), _p(tyy), _p(tzz), _p(txy), _p(txz), _p(tyz), _p(Vx), _p(Vy), _p(Vz),
          *_d(mu, dx, dy, dz), *_i(nx, ny, nz))


def predict_V(Vx, Vy, Vz, txx, tyy, tzz, txy, txz, tyz, rho, g, dt, dx, dy, dz):
    nx, ny, nz = txx.shape
    _call("predict_V", Vx, _p(Vx), _p(Vy), _p(Vz), _p(txx), _p(tyy), _p(tzz), _p(txy), _p(txz), _p(tyz),
          *_d(rho, g, dt, dx, dy, dz), *_i(nx, ny, nz))


def update_divV(divV, Vx, Vy, Vz, dx, dy, dz):
    nx, ny, nz = divV.shape
    _call("update_divV", divV, _p(divV), _p(Vx), _p(Vy), _p(Vz), *_d(dx, dy, dz), *_i(nx, ny, nz))


def update_dPrdtau(Pr, dPrdtau, divV, rho, dt, dtau, damp, dx, dy, dz):
    nx, ny, nz = Pr.shape
    _call("update_dPrdtau", Pr, _p(Pr), _p(dPrdtau), _p(divV), *_d(rho, dt, dtau, damp, dx, dy, dz),
          *_i(nx, ny, nz))


def update_Pr(Pr, dPrdtau, dtau):
    nx, ny, nz = Pr.shape
    _call("update_Pr", Pr, _p(Pr), _p(dPrdtau), *_d(dtau), *_i(nx, ny, nz))


def compute_res(Rp, Pr, divV, rho, dt, dx, dy, dz):
    nx, ny, nz = Pr.shape
    _call("compute_res", Pr, _p(Rp), _p(Pr), _p(divV), *_d(rho, dt, dx, dy, dz), *_i(nx, ny, nz))


def max_abs(A):
    return float(_call("max_abs", A, _p(A), C.c_long(A.size)))


def correct_V(Vx, Vy, Vz, Pr, dt, rho, dx, dy, dz):
    nx, ny, nz = Pr.shape
    _call("correct_V", Pr, _p(Vx), _p(Vy), _p(Vz), _p(Pr), *_d(dt, rho, dx, dy, dz), *_i(nx, ny, nz))


def bc_x(A):
    _call("bc_x", A, _p(A), *_i(*A.shape))


def bc_y(A):
    _call("bc_y", A, _p(A), *_i(*A.shape))


def bc_z(A):
    _call("bc_z", A, _p(A), *_i(*A.shape))


def bc_zV(A):
    _call("bc_zV", A, _p(A), *_i(*A.shape))


def bc_xhydstatic(A, dz, nz, g, rho):
    _call("bc_xhydstatic", A, _p(A), *_d(dz), *_i(nz), *_d(g, rho), *_i(*A.shape))


def bc_x_Vx(A, V):
    _call("bc_x_Vx", A, _p(A), *_d(V), *_i(*A.shape))


def bc_x_Pr(A, val):
    _call("bc_x_Pr", A, _p(A), *_d(val), *_i(*A.shape))


def copy(dst, src):
    _call("copy", dst, _p(dst), _p(src), C.c_long(src.size))


def set_cylinder(Cf, Vx, Vy, Vz, a2, b2, ox, oy, sinb, cosb, xco_g, yco_g, zco_g, lx, ly, lz, dx, dy, dz):
    nx, ny, nz = Cf.shape
    _call("set_cylinder", Cf, _p(Cf), _p(Vx), _p(Vy), _p(Vz),
          *_d(a2, b2, ox, oy, sinb, cosb, xco_g, yco_g, zco_g, lx, ly, lz, dx, dy, dz), *_i(nx, ny, nz))


def set_cylinder_local(Cf, Vx, Vy, Vz, a2, b2, ox, oy, sinb, cosb, lx, ly, lz, dx, dy, dz):
    nx, ny, nz = Cf.shape
    _call("set_cylinder_local", Cf, _p(Cf), _p(Vx), _p(Vy), _p(Vz),
          *_d(a2, b2, ox, oy, sinb, cosb, lx, ly, lz, dx, dy, dz), *_i(nx, ny, nz))


def advect(Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, Cf, C_o, dt, dx, dy, dz, faithful=True):
    nx, ny, nz = Cf.shape
    _call("advect", Cf, _p(Vx), _p(Vx_o), _p(Vy), _p(Vy_o), _p(Vz), _p(Vz_o), _p(Cf), _p(C_o),
          *_d(dt, dx, dy, dz), *_i(nx, ny, nz, 1 if faithful else 0))


def advect_window(Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, Cf, C_o, dt, dx, dy, dz, faithful, koff, nz_glob):
    """advect! on a window of a global grid that starts koff planes into it (departure indices from GLOBAL plane numbers)"""
    nx, ny, nz = Cf.shape
    _call("advect_window", Cf, _p(Vx), _p(Vx_o), _p(Vy), _p(Vy_o), _p(Vz), _p(Vz_o), _p(Cf), _p(C_o),
          *_d(dt, dx, dy, dz), *_i(nx, ny, nz, 1 if faithful else 0, koff, nz_glob))


def set_bc_Pr(Pr, bc_kind, owns_outlet=True, outlet_val=0.0, dz=0.0, nz_arg=0, g=0.0, rho=0.0):
    nx, ny, nz = Pr.shape
    _call("set_bc_Pr", Pr, _p(Pr), *_i(bc_kind, owns_outlet), *_d(outlet_val, dz), *_i(nz_arg), *_d(g, rho),
          *_i(nx, ny, nz))


def set_bc_Vel(Vx, Vy, Vz, bc_kind, owns_inlet=True, vin=0.0):
    nx, ny, nz = Vx.shape[0] - 1, Vx.shape[1], Vx.shape[2]
    _call("set_bc_Vel", Vx, _p(Vx), _p(Vy), _p(Vz), *_i(bc_kind, owns_inlet), *_d(vin), *_i(nx, ny, nz))


def pt_solve(Pr, dPrdtau, divV, Rp, rho, dt, dtau, damp, dx, dy, dz, bc_kind, owns_outlet, outlet_val, g,
             eps, niter, nchk, err_mul, err_div):
    """multi.jl:458-471 / gpu.jl:126-137 on one rank. Returns (iters_done, [err per check])."""
    nx, ny, nz = Pr.shape
    cap = niter // max(nchk, 1) + 1
    hist = np.zeros(cap, dtype=np.float64)
    nchecks = C.c_int(0)
    it = _call("pt_solve", Pr, _p(Pr), _p(dPrdtau), _p(divV), _p(Rp), *_d(rho, dt, dtau, damp, dx, dy, dz),
               *_i(nx, ny, nz, bc_kind, owns_outlet), *_d(outlet_val, g, eps), *_i(niter, nchk), *_d(err_mul, err_div),
               hist.ctypes.data_as(C.c_void_p), C.c_int(cap), C.byref(nchecks))
    return int(it), hist[: nchecks.value].tolist()
