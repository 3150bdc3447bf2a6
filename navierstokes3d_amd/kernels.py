"""Host-side mirror of the reference's kernel layer (scripts/NavierStokes3D_multi_gpu.jl:15-281,
scripts/NavierStokes3D_gpu.jl:175-368): the same kernel names and positional argument lists, each one a
thin call into libns3d.so (hand-written HIP, include/ns3d.h).  `!` is dropped, τ→tau, ∇V→divV.

Arrays are torch CUDA tensors used purely as device-memory handles, shaped like the reference
(nx,ny,nz) and laid out column-major (x fastest) like Julia arrays: create them with `zeros`/`from_numpy`.
The launch grid (nx,ny,nz) is derived from the array shapes like ParallelStencil's `@parallel` does.

PyTorch does no arithmetic here and there is no CPU fallback: without the HIP library / a GPU every call
raises (lib.Ns3dError).
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L

_DT = {torch.float64: "f64", torch.float32: "f32"}


# ---- column-major device arrays -------------------------------------------------------------------------
def zeros(shape, dtype=torch.float64, device="cuda"):
    """@zeros(nx,ny,nz) (multi.jl:343): a (nx,ny,nz) tensor with strides (1,nx,nx*ny)."""
    sx, sy, sz = shape
    return torch.zeros((sz, sy, sx), dtype=dtype, device=device).permute(2, 1, 0)


def from_numpy(a, device="cuda"):
    """Upload a numpy array (any order) as a column-major device array of the same shape."""
    a = np.asfortranarray(a)
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 1, 0))).to(device)
    return t.permute(2, 1, 0)


def to_numpy(t):
    """Download to a Fortran-ordered numpy array of the same shape."""
    return np.asfortranarray(t.permute(2, 1, 0).contiguous().cpu().numpy().transpose(2, 1, 0))


def clone(t):
    return t.permute(2, 1, 0).contiguous().clone().permute(2, 1, 0)


def _chk(t, shape=None, name="array"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise L.Ns3dError("%s must be a CUDA/HIP tensor (navierstokes3d_amd has no CPU path)" % name)
    if t.dtype not in _DT:
        raise L.Ns3dError("%s: unsupported dtype %s" % (name, t.dtype))
    sx, sy, sz = t.shape
    want = (1, sx, sx * sy)
    if any(n > 1 and s != w for n, s, w in zip(t.shape, t.stride(), want)):   # strides of extent-1 dims are free
        raise L.Ns3dError("%s must be column-major (x fastest): use kernels.zeros / from_numpy" % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise L.Ns3dError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return C.c_void_p(t.data_ptr())


class Context:
    """ns3d_ctx wrapper = what `@init_parallel_stencil(CUDA, Float64, 3)` sets up in the reference
    (gpu.jl:4-8): device + arithmetic mode.  mode: 'strict' (bit-identical to the reference operation
    order) or 'fast' (reciprocals + FMA)."""

    _owns = True

    @classmethod
    def from_handle(cls, handle, device, mode):
        """Wrap an ns3d_ctx owned by someone else (a rank of an ns3d_mgpu): same calls, never destroyed from here."""
        self = cls.__new__(cls)
        self.lib = L.load()
        self.device, self.mode, self.handle, self._owns = int(device), mode, handle, False
        self.use_torch_stream()
        return self

    def __init__(self, device=None, mode="strict", async_=False, ieee_div=False):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.Ns3dError("no GPU visible: libns3d has no CPU path")
        if device is None:
            device = torch.cuda.current_device()
        self.device = int(device)
        flags = {"strict": L.NS3D_STRICT, "fast": L.NS3D_FAST}[mode] | (L.NS3D_ASYNC if async_ else 0)
        flags |= L.NS3D_IEEE_DIV if ieee_div else 0
        self.mode = mode
        self.handle = self.lib.ns3d_create(self.device, flags)
        if not self.handle:
            raise L.Ns3dError("ns3d_create failed: " + L.last_error())
        self.use_torch_stream()

    def use_torch_stream(self, stream=None, pin=False):
        """Launch on PyTorch's current stream so that tensor ops (uploads, copies) are ordered with kernels.  pin: stay on
        `stream` whatever PyTorch's current stream becomes (a rank of a MultiGpu created with own_streams)."""
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._pinned = bool(pin and stream is not None)
        self._stream = s.cuda_stream
        L.check(self.lib.ns3d_set_stream(self.handle, C.c_void_p(s.cuda_stream)))

    def sync(self):
        L.check(self.lib.ns3d_sync(self.handle))

    def reserve_cus(self, n):
        """Leave n compute units out of this context's launches (ns3d_reserve_cus: a stream with a CU mask, so that RCCL's kernels
        find room beside a sweep that would hold every CU).  Returns the stream as a torch stream — tensor work that must be
        ordered with the kernels belongs on it (`with torch.cuda.stream(s): …`); n = 0 goes back to PyTorch's current stream."""
        L.check(self.lib.ns3d_reserve_cus(self.handle, int(n)))
        if int(n) <= 0:
            self._ext = None
            self.use_torch_stream()
            return torch.cuda.current_stream(self.device)
        self._ext = torch.cuda.ExternalStream(int(self.lib.ns3d_get_stream(self.handle)), device=self.device)
        self._pinned, self._stream = True, self._ext.cuda_stream
        return self._ext

    def set_pt_variant(self, v):
        L.check(self.lib.ns3d_set_pt_variant(self.handle, int(v)))

    def set_graph_mode(self, mode):
        """HIP-graph replay of residual-check blocks in pt_solve: -1 auto (launch-bound grids), 0 off, 1 on."""
        L.check(self.lib.ns3d_set_graph_mode(self.handle, int(mode)))

    def set_persist_mode(self, mode):
        """pt_iterate / pt_solve on launch-bound grids: a whole block of iterations in one cooperative launch (k_pt_persist);
        -1 automatic (small grids), 0 never, 1 wherever it applies.  Same results."""
        L.check(self.lib.ns3d_set_persist_mode(self.handle, int(mode)))

    def persist_faults(self):
        """Cooperative launches (k_pt_persist) of this context in which a hand-over timed out; each was redone by launches."""
        return int(self.lib.ns3d_persist_faults(self.handle))

    def set_autotune(self, on):
        """Time the tile shapes of the two-iteration sweep on the first launch per grid (default on; same results)."""
        L.check(self.lib.ns3d_set_autotune(self.handle, int(bool(on))))

    def last_pt2_variant(self):
        """Variant (shape·100 + z-chunk) of the latest two-iteration launch; 0 = built-in choice by grid."""
        return int(self.lib.ns3d_last_pt2_variant(self.handle))

    def last_ptn_variant(self):
        return int(self.lib.ns3d_last_ptn_variant(self.handle))

    def last_pt_depth(self):
        """PT iterations of the latest multi-iteration pass (after plan_pt: the planned depth)."""
        return int(self.lib.ns3d_last_pt_depth(self.handle))

    def arith_build(self, dx, dy, dz):
        """Which compilation of the kernels these grid spacings select: 'strict' | 'strictx' | 'strictp' | 'fast'."""
        return {0: "strict", 1: "strictx", 2: "fast", 3: "strictp"}[int(self.lib.ns3d_arith_build(self.handle, dx, dy, dz))]

    def set_pt2_variant(self, v):
        """Temporal blocking (two PT iterations per pass) in pt_iterate / pt_solve: v < 0 off, 0 default tile."""
        L.check(self.lib.ns3d_set_pt2_variant(self.handle, int(v)))

    def set_ptn_variant(self, v):
        """Tile shape of the N-iteration sweep (pt_sweepn): shape*100 + kz, 0 = built-in."""
        L.check(self.lib.ns3d_set_ptn_variant(self.handle, int(v)))

    def set_pt_depth(self, depth):
        """PT iterations per pass over memory in pt_iterate / pt_solve: 0 automatic, 1…4 forced (5: float32 fields only)."""
        L.check(self.lib.ns3d_set_pt_depth(self.handle, int(depth)))

    def selftest_exact_div(self, d, n=1 << 24, seed=1, dtype=torch.float64):
        """Mismatches between the divisor-known-in-advance division and the plain IEEE division over n dividends."""
        out = C.c_long(-1)
        fn = getattr(self.lib, "ns3d_selftest_exact_div_" + _DT[dtype])
        L.check(fn(self.handle, C.c_double(d), C.c_long(n), C.c_ulonglong(seed), C.byref(out)))
        return out.value

    def close(self):
        if getattr(self, "handle", None):
            if self._owns:
                self.lib.ns3d_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, ref, *args):
        if not getattr(self, "_pinned", False) and torch.cuda.current_stream(self.device).cuda_stream != self._stream:
            self.use_torch_stream()     # follow `with torch.cuda.stream(...)` blocks
        fn = getattr(self.lib, "ns3d_%s_%s" % (name, _DT[ref.dtype]))
        L.check(fn(self.handle, *args))


_DEFAULT = {}


def init_parallel_stencil(device=None, mode="strict", async_=False):
    """Create (or replace) the default context for `device`; mirrors @init_parallel_stencil (gpu.jl:4-8)."""
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    old = _DEFAULT.pop(int(device), None)
    if old is not None:
        old.close()
    ctx = Context(device, mode, async_)
    _DEFAULT[int(device)] = ctx
    return ctx


def default_context(t=None):
    dev = t.device.index if t is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
    ctx = _DEFAULT.get(dev)
    if ctx is None:
        ctx = init_parallel_stencil(dev)
    return ctx


def _ctx(ctx, t):
    return ctx if ctx is not None else default_context(t)


def _d(*xs):
    return [C.c_double(float(x)) for x in xs]


# ---- the reference's kernels ----------------------------------------------------------------------------
def update_tau(txx, tyy, tzz, txy, txz, tyz, Vx, Vy, Vz, mu, dx, dy, dz, ctx=None):
    """update_τ!  multi.jl:36-44 / gpu.jl:177-185"""
    nx, ny, nz = txx.shape
    c = (nx, ny, nz); s = (nx - 1, ny - 1, nz - 1)
    _ctx(ctx, txx).call("update_tau", txx, _chk(txx, c, "txx"), _chk(tyy, c, "tyy"), _chk(tzz, c, "tzz"),
                        _chk(txy, s, "txy"), _chk(txz, s, "txz"), _chk(tyz, s, "tyz"),
                        _chk(Vx, (nx + 1, ny, nz), "Vx"), _chk(Vy, (nx, ny + 1, nz), "Vy"),
                        _chk(Vz, (nx, ny, nz + 1), "Vz"), *_d(mu, dx, dy, dz), nx, ny, nz)


def predict_V(Vx, Vy, Vz, txx, tyy, tzz, txy, txz, tyz, rho, g, dt, dx, dy, dz, ctx=None):
    """predict_V!  multi.jl:50-55 / gpu.jl:187-192"""
    nx, ny, nz = txx.shape
    c = (nx, ny, nz); s = (nx - 1, ny - 1, nz - 1)
    _ctx(ctx, txx).call("predict_V", txx, _chk(Vx, (nx + 1, ny, nz), "Vx"), _chk(Vy, (nx, ny + 1, nz), "Vy"),
                        _chk(Vz, (nx, ny, nz + 1), "Vz"), _chk(txx, c, "txx"), _chk(tyy, c, "tyy"),
                        _chk(tzz, c, "tzz"), _chk(txy, s, "txy"), _chk(txz, s, "txz"), _chk(tyz, s, "tyz"),
                        *_d(rho, g, dt, dx, dy, dz), nx, ny, nz)



def predict_fused(Vx_new, Vy_new, Vz_new, Vx, Vy, Vz, mu, rho, g, dt, dx, dy, dz, ctx=None):
    """{update_τ!; predict_V!} (multi.jl:449,451) in one pass: complete predicted fields into buffers of their own, the stress
    arrays neither read nor written; the caller swaps the names."""
    nx, ny, nz = Vx.shape[0] - 1, Vx.shape[1], Vx.shape[2]
    _ctx(ctx, Vx).call("predict_fused", Vx, _chk(Vx_new, (nx + 1, ny, nz), "Vx_new"), _chk(Vy_new, (nx, ny + 1, nz), "Vy_new"),
                       _chk(Vz_new, (nx, ny, nz + 1), "Vz_new"), _chk(Vx, (nx + 1, ny, nz), "Vx"), _chk(Vy, (nx, ny + 1, nz), "Vy"),
                       _chk(Vz, (nx, ny, nz + 1), "Vz"), *_d(mu, rho, g, dt, dx, dy, dz), nx, ny, nz)

def set_cylinder(Cf, Vx, Vy, Vz, a2, b2, ox, oy, sinb, cosb, *rest, ctx=None):
    """set_cylinder!  multi.jl:249-281 (19 arguments: …,xco_g,yco_g,zco_g,lx,ly,lz,dx,dy,dz) or
    gpu.jl:336-368 (16 arguments: …,lx,ly,lz,dx,dy,dz) — dispatched on the argument count like the two
    scripts' definitions."""
    nx, ny, nz = Cf.shape
    ptrs = (_chk(Cf, (nx, ny, nz), "C"), _chk(Vx, (nx + 1, ny, nz), "Vx"), _chk(Vy, (nx, ny + 1, nz), "Vy"),
            _chk(Vz, (nx, ny, nz + 1), "Vz"))
    if len(rest) == 9:
        _ctx(ctx, Cf).call("set_cylinder", Cf, *ptrs, *_d(a2, b2, ox, oy, sinb, cosb, *rest), nx, ny, nz)
    elif len(rest) == 6:
        _ctx(ctx, Cf).call("set_cylinder_local", Cf, *ptrs, *_d(a2, b2, ox, oy, sinb, cosb, *rest), nx, ny, nz)
    else:
        raise TypeError("set_cylinder: expected 19 (multi.jl) or 16 (gpu.jl) positional arguments")


def update_divV(divV, Vx, Vy, Vz, dx, dy, dz, ctx=None):
    """update_∇V!  multi.jl:61-64 / gpu.jl:194-197"""
    nx, ny, nz = divV.shape
    _ctx(ctx, divV).call("update_divV", divV, _chk(divV, None, "divV"), _chk(Vx, (nx + 1, ny, nz), "Vx"),
                         _chk(Vy, (nx, ny + 1, nz), "Vy"), _chk(Vz, (nx, ny, nz + 1), "Vz"), *_d(dx, dy, dz),
                         nx, ny, nz)


def update_dPrdtau(Pr, dPrdtau, divV, rho, dt, dtau, damp, dx, dy, dz, ctx=None):
    """update_dPrdτ!  multi.jl:70-73 / gpu.jl:199-202"""
    nx, ny, nz = Pr.shape
    _ctx(ctx, Pr).call("update_dPrdtau", Pr, _chk(Pr, None, "Pr"), _chk(dPrdtau, (nx - 2, ny - 2, nz - 2), "dPrdtau"),
                       _chk(divV, (nx, ny, nz), "divV"), *_d(rho, dt, dtau, damp, dx, dy, dz), nx, ny, nz)


def update_Pr(Pr, dPrdtau, dtau, ctx=None):
    """update_Pr!  multi.jl:79-82 / gpu.jl:204-207"""
    nx, ny, nz = Pr.shape
    _ctx(ctx, Pr).call("update_Pr", Pr, _chk(Pr, None, "Pr"), _chk(dPrdtau, (nx - 2, ny - 2, nz - 2), "dPrdtau"),
                       *_d(dtau), nx, ny, nz)


def compute_res(Rp, Pr, divV, rho, dt, dx, dy, dz, ctx=None):
    """compute_res!  multi.jl:88-91 / gpu.jl:209-212"""
    nx, ny, nz = Pr.shape
    _ctx(ctx, Pr).call("compute_res", Pr, _chk(Rp, (nx - 2, ny - 2, nz - 2), "Rp"), _chk(Pr, None, "Pr"),
                       _chk(divV, (nx, ny, nz), "divV"), *_d(rho, dt, dx, dy, dz), nx, ny, nz)


def max_abs(A, ctx=None):
    """maximum(abs.(A)) — NaN-propagating  (multi.jl:466 / gpu.jl:132)"""
    out = C.c_double(0.0)
    _ctx(ctx, A).call("max_abs", A, _chk(A, None, "A"), C.c_long(A.numel()), C.byref(out))
    return out.value


def correct_V(Vx, Vy, Vz, Pr, dt, rho, dx, dy, dz, ctx=None):
    """correct_V!  multi.jl:97-102 / gpu.jl:214-219"""
    nx, ny, nz = Pr.shape
    _ctx(ctx, Pr).call("correct_V", Pr, _chk(Vx, (nx + 1, ny, nz), "Vx"), _chk(Vy, (nx, ny + 1, nz), "Vy"),
                       _chk(Vz, (nx, ny, nz + 1), "Vz"), _chk(Pr, None, "Pr"), *_d(dt, rho, dx, dy, dz), nx, ny, nz)


def _bc(name, A, *extra, ctx=None):
    _ctx(ctx, A).call(name, A, _chk(A, None, "A"), *extra, *A.shape)


def bc_x(A, ctx=None):
    """bc_x!  multi.jl:108-112"""
    _bc("bc_x", A, ctx=ctx)


def bc_y(A, ctx=None):
    """bc_y!  multi.jl:118-122"""
    _bc("bc_y", A, ctx=ctx)


def bc_z(A, ctx=None):
    """bc_z!  multi.jl:128-132"""
    _bc("bc_z", A, ctx=ctx)


def bc_zV(A, ctx=None):
    """bc_zV!  gpu.jl:239-243"""
    _bc("bc_zV", A, ctx=ctx)


def bc_xhydstatic(A, dz, nz, g, rho, ctx=None):
    """bc_xhydstatic!  gpu.jl:257-261"""
    _bc("bc_xhydstatic", A, C.c_double(dz), C.c_int(nz), C.c_double(g), C.c_double(rho), ctx=ctx)


def bc_x_Vx(A, V, ctx=None):
    """bc_x_Vx!  multi.jl:138-141"""
    _bc("bc_x_Vx", A, C.c_double(V), ctx=ctx)


def bc_x_Pr(A, val, ctx=None):
    """bc_x_Pr!  multi.jl:147-150"""
    _bc("bc_x_Pr", A, C.c_double(val), ctx=ctx)


def copy(dst, src, ctx=None):
    """X_o .= X  (multi.jl:475 / gpu.jl:141)"""
    _ctx(ctx, dst).call("copy", dst, _chk(dst, src.shape, "dst"), _chk(src, None, "src"), C.c_long(src.numel()))


def advect(Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, Cf, C_o, dt, dx, dy, dz, faithful=True, ctx=None):
    """advect!  multi.jl:217-243 / gpu.jl:308-334.  faithful=True reproduces the reference's third branch
    (back-tracks Vy, never Vz — SURVEY.md App. B1)."""
    nx, ny, nz = Cf.shape
    _ctx(ctx, Cf).call("advect", Cf, _chk(Vx, (nx + 1, ny, nz), "Vx"), _chk(Vx_o, (nx + 1, ny, nz), "Vx_o"),
                       _chk(Vy, (nx, ny + 1, nz), "Vy"), _chk(Vy_o, (nx, ny + 1, nz), "Vy_o"),
                       _chk(Vz, (nx, ny, nz + 1), "Vz"), _chk(Vz_o, (nx, ny, nz + 1), "Vz_o"),
                       _chk(Cf, None, "C"), _chk(C_o, (nx, ny, nz), "C_o"), *_d(dt, dx, dy, dz), nx, ny, nz,
                       1 if faithful else 0)


def copy_advect(Vx_new, Vx, Vy_new, Vy, Vz_new, Vz, C_new, Cf, dt, dx, dy, dz, faithful=True, ctx=None):
    """{X_o .= X; advect!} (multi.jl:475-476 / gpu.jl:141-142) in one pass: reads the current fields, writes COMPLETE new
    fields into buffers of their own; the caller swaps the roles afterwards.  Vz_new may be Vz in faithful mode."""
    nx, ny, nz = Cf.shape
    _ctx(ctx, Cf).call("copy_advect", Cf, _chk(Vx_new, (nx + 1, ny, nz), "Vx_new"), _chk(Vx, (nx + 1, ny, nz), "Vx"),
                       _chk(Vy_new, (nx, ny + 1, nz), "Vy_new"), _chk(Vy, (nx, ny + 1, nz), "Vy"),
                       _chk(Vz_new, (nx, ny, nz + 1), "Vz_new"), _chk(Vz, (nx, ny, nz + 1), "Vz"),
                       _chk(C_new, (nx, ny, nz), "C_new"), _chk(Cf, None, "C"), *_d(dt, dx, dy, dz), nx, ny, nz,
                       1 if faithful else 0)


# ---- the reference's host sequences, one library call each ----------------------------------------------
def set_bc_Pr_multi(Pr, owns_outlet, val=0.0, ctx=None):
    """set_bc_Pr!(Pr, xve_g, lx, val) of multi.jl:175-181 with `xve_g == lx/2` passed as a flag; the trailing
    update_halo!(Pr) (multi.jl:182) is the caller's."""
    nx, ny, nz = Pr.shape
    _ctx(ctx, Pr).call("set_bc_Pr", Pr, _chk(Pr, None, "Pr"), L.NS3D_BC_MULTI, int(bool(owns_outlet)),
                       *_d(val, 0.0), 0, *_d(0.0, 0.0), nx, ny, nz)


def set_bc_Pr_gpu(Pr, dz, nz_arg, g, rho, ctx=None):
    """set_bc_Pr!(Pr, dz, nz, g, ρ) of gpu.jl:281-286"""
    nx, ny, nz = Pr.shape
    _ctx(ctx, Pr).call("set_bc_Pr", Pr, _chk(Pr, None, "Pr"), L.NS3D_BC_GPU, 0, *_d(0.0, dz), int(nz_arg),
                       *_d(g, rho), nx, ny, nz)


def set_bc_Vel_multi(Vx, Vy, Vz, owns_inlet, vin, ctx=None):
    """set_bc_Vel!(Vx,Vy,Vz,xvo_g,lx,vin) of multi.jl:156-166 (flag instead of `xvo_g == -lx/2`; halo update
    multi.jl:167 is the caller's)."""
    nx, ny, nz = Vx.shape[0] - 1, Vx.shape[1], Vx.shape[2]
    _ctx(ctx, Vx).call("set_bc_Vel", Vx, _chk(Vx, None, "Vx"), _chk(Vy, (nx, ny + 1, nz), "Vy"),
                       _chk(Vz, (nx, ny, nz + 1), "Vz"), L.NS3D_BC_MULTI, int(bool(owns_inlet)), *_d(vin), nx, ny, nz)


def set_bc_Vel_gpu(Vx, Vy, Vz, ctx=None):
    """set_bc_Vel!(Vx,Vy,Vz,Vprof) of gpu.jl:264-279 (Vprof is unused there)."""
    nx, ny, nz = Vx.shape[0] - 1, Vx.shape[1], Vx.shape[2]
    _ctx(ctx, Vx).call("set_bc_Vel", Vx, _chk(Vx, None, "Vx"), _chk(Vy, (nx, ny + 1, nz), "Vy"),
                       _chk(Vz, (nx, ny, nz + 1), "Vz"), L.NS3D_BC_GPU, 0, *_d(0.0), nx, ny, nz)


# ---- fused pseudo-transient path ------------------------------------------------------------------------
def pt_params(Pr, rho, dt, dtau, damp, dx, dy, dz, bc_kind=L.NS3D_BC_MULTI, owns_outlet=True, outlet_val=0.0,
              g=0.0, z_lo_is_halo=False, z_hi_is_halo=False):
    nx, ny, nz = Pr.shape
    return L.PtParams(rho, dt, dtau, damp, dx, dy, dz, nx, ny, nz, int(bc_kind), int(bool(owns_outlet)),
                      outlet_val, g, int(bool(z_lo_is_halo)), int(bool(z_hi_is_halo)))


def pt_iterate(Pr, dPrdtau, divV, p, n_iters, ctx=None):
    """n_iters × {update_dPrdτ!; update_Pr!; set_bc_Pr!} (multi.jl:459-463 / gpu.jl:127-129) as fused sweeps."""
    nx, ny, nz = Pr.shape
    _ctx(ctx, Pr).call("pt_iterate", Pr, _chk(Pr, None, "Pr"), _chk(dPrdtau, (nx - 2, ny - 2, nz - 2), "dPrdtau"),
                       _chk(divV, (nx, ny, nz), "divV"), C.byref(p), int(n_iters))


def poisson_direct(Pr, dPrdtau, divV, p, ctx=None):
    """OUTSIDE PARITY: the discrete pressure-Poisson problem the PT loop multi.jl:458-471 / gpu.jl:126-137 iterates
    towards, solved directly (exact diagonalisation, six fp64 MFMA matrix products): Pr ← solution with set_bc_Pr!'s boundary
    cells, dPrdτ ← 0."""
    nx, ny, nz = Pr.shape
    _ctx(ctx, Pr).call("poisson_direct", Pr, _chk(Pr, None, "Pr"), _chk(dPrdtau, (nx - 2, ny - 2, nz - 2), "dPrdtau"),
                       _chk(divV, (nx, ny, nz), "divV"), C.byref(p))


def pt_sweep(Pr_in, Pr_out, dPrdtau, divV, p, k0, k1, ctx=None):
    """One fused sweep Pr_in → Pr_out over interior planes k0 ≤ k < k1 (0-based)."""
    nx, ny, nz = Pr_in.shape
    _ctx(ctx, Pr_in).call("pt_sweep", Pr_in, _chk(Pr_in, None, "Pr_in"), _chk(Pr_out, (nx, ny, nz), "Pr_out"),
                          _chk(dPrdtau, (nx - 2, ny - 2, nz - 2), "dPrdtau"), _chk(divV, (nx, ny, nz), "divV"),
                          C.byref(p), int(k0), int(k1))


def pt_sweep2(Pr_in, Pr_out, dPrdtau_in, dPrdtau_out, divV, p, k0=None, k1=None, ctx=None):
    """TWO fused PT iterations (Pr_in, dPrdtau_in) → (Pr_out, dPrdtau_out) in one pass over memory (same result
    as two pt_sweep calls); all four buffers distinct."""
    nx, ny, nz = Pr_in.shape
    _ctx(ctx, Pr_in).call("pt_sweep2", Pr_in, _chk(Pr_in, None, "Pr_in"), _chk(Pr_out, (nx, ny, nz), "Pr_out"),
                          _chk(dPrdtau_in, (nx - 2, ny - 2, nz - 2), "dPrdtau_in"),
                          _chk(dPrdtau_out, (nx - 2, ny - 2, nz - 2), "dPrdtau_out"), _chk(divV, (nx, ny, nz), "divV"),
                          C.byref(p), 1 if k0 is None else int(k0), nz - 1 if k1 is None else int(k1))


def pt_sweepn(nlev, Pr_in, Pr_out, dPrdtau_in, dPrdtau_out, divV, p, k0=None, k1=None, ctx=None):
    """nlev (2…4; float32 also 5) fused PT iterations (Pr_in, dPrdtau_in) → (Pr_out, dPrdtau_out) in one pass over memory (same result as
    nlev pt_sweep calls); all four buffers distinct."""
    nx, ny, nz = Pr_in.shape
    _ctx(ctx, Pr_in).call("pt_sweepn", Pr_in, int(nlev), _chk(Pr_in, None, "Pr_in"), _chk(Pr_out, (nx, ny, nz), "Pr_out"),
                          _chk(dPrdtau_in, (nx - 2, ny - 2, nz - 2), "dPrdtau_in"),
                          _chk(dPrdtau_out, (nx - 2, ny - 2, nz - 2), "dPrdtau_out"), _chk(divV, (nx, ny, nz), "divV"),
                          C.byref(p), 1 if k0 is None else int(k0), nz - 1 if k1 is None else int(k1))


def plan_pt(Pr_in, Pr_out, dPrdtau_in, dPrdtau_out, divV, p, k0=None, k1=None, ctx=None):
    """Plan phase of the two-iteration sweep: time the tile shapes now, on these arguments, and remember the winner
    (ns3d_plan_pt).  pt_iterate / pt_solve plan by themselves on first use; pt_sweep2 only looks the choice up."""
    nx, ny, nz = Pr_in.shape
    _ctx(ctx, Pr_in).call("plan_pt", Pr_in, _chk(Pr_in, None, "Pr_in"), _chk(Pr_out, (nx, ny, nz), "Pr_out"),
                          _chk(dPrdtau_in, (nx - 2, ny - 2, nz - 2), "dPrdtau_in"),
                          _chk(dPrdtau_out, (nx - 2, ny - 2, nz - 2), "dPrdtau_out"), _chk(divV, (nx, ny, nz), "divV"),
                          C.byref(p), 1 if k0 is None else int(k0), nz - 1 if k1 is None else int(k1))


def residual_max(Pr, divV, p, ctx=None):
    """maximum(abs.(Rp)) after compute_res! (multi.jl:465-466) without materialising Rp."""
    out = C.c_double(0.0)
    _ctx(ctx, Pr).call("residual_max", Pr, _chk(Pr, None, "Pr"), _chk(divV, tuple(Pr.shape), "divV"), C.byref(p),
                       C.byref(out))
    return out.value


_STEP_SWAP = ("Vx", "Vy", "Vz", "Vx_o", "Vy_o", "Vz_o", "C", "C_o")


def time_step(f, sp, ctx=None):
    """One whole time step in one library call (ns3d_time_step: multi.jl:449-477 on one rank / gpu.jl:121-142 in its fused form).
    f: a namespace of the reference's arrays as tensors (Pr, dPrdtau, divV, Vx, …, C_o, txx … tyz); sp: lib.StepParams.  The step
    swaps the roles of X and X_o instead of copying: on return f.Vx is the current field and f.Vx_o the previous one.  Returns
    (iters_done, [err …]) like pt_solve."""
    nx, ny, nz = f.Pr.shape
    if (sp.nx, sp.ny, sp.nz) != (nx, ny, nz):
        raise L.Ns3dError("time_step: params grid %r differs from Pr's %r" % ((sp.nx, sp.ny, sp.nz), (nx, ny, nz)))
    shapes = {"Pr": (nx, ny, nz), "dPrdtau": (nx - 2, ny - 2, nz - 2), "divV": (nx, ny, nz), "Vx": (nx + 1, ny, nz),
              "Vy": (nx, ny + 1, nz), "Vz": (nx, ny, nz + 1), "Vx_o": (nx + 1, ny, nz), "Vy_o": (nx, ny + 1, nz),
              "Vz_o": (nx, ny, nz + 1), "C": (nx, ny, nz), "C_o": (nx, ny, nz), "txx": (nx, ny, nz), "tyy": (nx, ny, nz),
              "tzz": (nx, ny, nz), "txy": (nx - 1, ny - 1, nz - 1), "txz": (nx - 1, ny - 1, nz - 1), "tyz": (nx - 1, ny - 1, nz - 1)}
    sf = L.StepFields()
    for n, shp in shapes.items():
        t = getattr(f, n, None)
        setattr(sf, n, None if t is None else _chk(t, shp, n))
    by_ptr = {getattr(f, n).data_ptr(): getattr(f, n) for n in _STEP_SWAP}
    cap = sp.niter // max(sp.nchk, 1) + 1
    hist = (C.c_double * cap)()
    it, nchecks = C.c_int(0), C.c_int(0)
    _ctx(ctx, f.Pr).call("time_step", f.Pr, C.byref(sf), C.byref(sp), C.byref(it), hist, cap, C.byref(nchecks))
    for n in _STEP_SWAP:
        setattr(f, n, by_ptr[getattr(sf, n)])
    return it.value, list(hist[: nchecks.value])


def pt_solve(Pr, dPrdtau, divV, p, eps, niter, nchk, err_mul, err_div, ctx=None):
    """The inner loop multi.jl:458-471 / gpu.jl:126-137 on one rank; err = max|Rp|*err_mul/err_div
    (= maximum(abs.(Rp))*ly^2/psc). Returns (iters_done, [err …])."""
    nx, ny, nz = Pr.shape
    cap = niter // max(nchk, 1) + 1
    hist = (C.c_double * cap)()
    it, nchecks = C.c_int(0), C.c_int(0)
    _ctx(ctx, Pr).call("pt_solve", Pr, _chk(Pr, None, "Pr"), _chk(dPrdtau, (nx - 2, ny - 2, nz - 2), "dPrdtau"),
                       _chk(divV, (nx, ny, nz), "divV"), C.byref(p), C.c_double(eps), int(niter), int(nchk),
                       C.c_double(err_mul), C.c_double(err_div), C.byref(it), hist, cap, C.byref(nchecks))
    return it.value, list(hist[: nchecks.value])
