"""The multi-GPU half of the C ABI (include/ns3d.h, ns3d_mgpu_*): ImplicitGlobalGrid's init_global_grid / update_halo! /
max_g / gather! / finalize_global_grid (scripts/NavierStokes3D_multi_gpu.jl:325, :371…, :21, :399-403, :534) for z-slabs
(dims = (1,1,P), the default here) or any Cartesian topology (`dims=`; dims_create(P) = init_global_grid's own default), and
the pseudo-transient loop of a z-slab rank (multi.jl:458-471) with its halo traffic behind the interior sweep.

Two forms:
  MultiGpu.create(devices, …)          one process drives P devices (a device may repeat: virtual ranks on one GPU); planes
                                       move by hipMemcpyPeerAsync over xGMI
  MultiGpu.create_rank(P, rank, …)     one process per GPU; planes move by RCCL send/recv inside libns3d.so; the
                                       communicator's unique id travels through torch.distributed (any backend) or MPI

Per-rank arguments are lists with one entry per LOCAL rank (a bare tensor is accepted when there is one local rank).
"""
import ctypes as C

import numpy as np
import torch

from . import kernels as K
from . import lib as L

_SUF = {torch.float64: "f64", torch.float32: "f32"}
_FLAGS = {"strict": L.NS3D_STRICT, "fast": L.NS3D_FAST}


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


class MultiGpu:
    def __init__(self, handle, mode):
        self.lib = L.load()
        self.handle = handle
        self.mode = mode
        self.P = self.lib.ns3d_mgpu_world(handle)
        self.nlocal = self.lib.ns3d_mgpu_nlocal(handle)
        self.ranks = [self.lib.ns3d_mgpu_rank(handle, l) for l in range(self.nlocal)]
        self.transport = self.lib.ns3d_mgpu_transport(handle).decode()
        d3 = (C.c_int * 3)()
        L.check(self.lib.ns3d_mgpu_dims(handle, d3))
        self.dims = tuple(d3)
        self.coords = []
        for l in range(self.nlocal):
            L.check(self.lib.ns3d_mgpu_coords(handle, l, d3))
            self.coords.append(tuple(d3))
        self.contexts = []
        self._devices = []

    # ---- init_global_grid (multi.jl:325) ---------------------------------------------------------------------
    @staticmethod
    def dims_create(P, dims=(0, 0, 0)):
        """MPI.Dims_create!(nprocs, dims) as init_global_grid calls it: zeros are filled, balanced and non-increasing."""
        d3 = (C.c_int * 3)(*[int(q) for q in dims])
        L.check(L.load().ns3d_dims_create(int(P), d3))
        return tuple(d3)

    @classmethod
    def create(cls, devices, nx, ny, nz, mode="strict", async_=True, dims=None, own_streams=False):
        """dims=None: z-slabs over len(devices) ranks; dims=(Px,Py,Pz): a Cartesian topology, devices[rank] in MPI_Cart
        rank order (last dimension fastest).  own_streams: every rank computes on a non-blocking stream of its own (as the
        ranks of different devices always do) instead of following PyTorch's current stream of its device — with several
        virtual ranks on ONE device this is what makes the ready/landed events between the ranks carry the ordering; the
        caller then orders its own tensor work with torch.cuda.synchronize() before and MultiGpu.sync() after."""
        lib = L.load()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        flags = _FLAGS[mode] | (L.NS3D_ASYNC if async_ else 0)
        if dims is None:
            h = lib.ns3d_mgpu_create(len(devices), devs, int(nx), int(ny), int(nz), flags)
        else:
            if len(dims) != 3 or int(dims[0]) * int(dims[1]) * int(dims[2]) != len(devices):
                raise L.Ns3dError("dims %r do not hold %d ranks" % (tuple(dims), len(devices)))
            h = lib.ns3d_mgpu_create_cart((C.c_int * 3)(*[int(q) for q in dims]), devs, int(nx), int(ny), int(nz), flags)
        if not h:
            raise L.Ns3dError("ns3d_mgpu_create failed: " + L.last_error())
        self = cls(h, mode)
        self._devices = [int(d) for d in devices]
        self._wrap_contexts()
        if own_streams:
            self._own = [torch.cuda.Stream(device=d) for d in self._devices]
            for c, st in zip(self.contexts, self._own):
                c.use_torch_stream(st, pin=True)
        return self

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(L.NS3D_UNIQUE_ID_BYTES)
        L.check(L.load().ns3d_mgpu_unique_id(buf))
        return buf.raw

    @classmethod
    def create_rank(cls, P, rank, device, unique_id, nx, ny, nz, mode="strict", async_=True, dims=None):
        lib = L.load()
        if len(unique_id) != L.NS3D_UNIQUE_ID_BYTES:
            raise L.Ns3dError("unique id must be %d bytes" % L.NS3D_UNIQUE_ID_BYTES)
        flags = _FLAGS[mode] | (L.NS3D_ASYNC if async_ else 0)
        if dims is None:
            h = lib.ns3d_mgpu_create_rank(int(P), int(rank), int(device), bytes(unique_id), int(nx), int(ny), int(nz), flags)
        else:
            if len(dims) != 3 or int(dims[0]) * int(dims[1]) * int(dims[2]) != int(P):
                raise L.Ns3dError("dims %r do not hold %d ranks" % (tuple(dims), P))
            h = lib.ns3d_mgpu_create_rank_cart((C.c_int * 3)(*[int(q) for q in dims]), int(rank), int(device), bytes(unique_id),
                                               int(nx), int(ny), int(nz), flags)
        if not h:
            raise L.Ns3dError("ns3d_mgpu_create_rank failed: " + L.last_error())
        self = cls(h, mode)
        self._devices = [int(device)]
        self._wrap_contexts()
        return self

    def _wrap_contexts(self):
        self.contexts = [K.Context.from_handle(self.lib.ns3d_mgpu_ctx(self.handle, l), self._devices[l], self.mode)
                         for l in range(self.nlocal)]

    def rccl_ranks(self):
        return int(self.lib.ns3d_mgpu_rccl_ranks(self.handle))

    def nz_g(self):
        return int(self.lib.ns3d_mgpu_nz_g(self.handle))

    def n_g(self):
        d3 = (C.c_int * 3)()
        L.check(self.lib.ns3d_mgpu_n_g(self.handle, d3))
        return tuple(d3)

    # ---- finalize_global_grid (multi.jl:534) -----------------------------------------------------------------
    def close(self):
        if getattr(self, "handle", None):
            for c in self.contexts:
                c.close()
            self.lib.ns3d_mgpu_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        L.check(self.lib.ns3d_mgpu_sync(self.handle))

    def _follow_torch_streams(self):
        if getattr(self, "_own", None):       # pinned rank streams: what PyTorch's streams hold so far is complete first
            for d in set(self._devices):
                torch.cuda.current_stream(d).synchronize()
            return
        for c in self.contexts:
            if torch.cuda.current_stream(c.device).cuda_stream != c._stream:
                c.use_torch_stream()

    def _ptrs(self, per_field_lists):
        """field-major array of device pointers: fields[f*nlocal + l]"""
        flat = []
        for lst in per_field_lists:
            lst = _as_list(lst)
            if len(lst) != self.nlocal:
                raise L.Ns3dError("expected one tensor per local rank (%d), got %d" % (self.nlocal, len(lst)))
            flat += [K._chk(t, None, "field").value for t in lst]
        return (C.c_void_p * len(flat))(*flat)

    def _typed(self, name, ref):
        return getattr(self.lib, "ns3d_%s_%s" % (name, _SUF[ref.dtype]))

    # ---- update_halo! (multi.jl:371,373,450,453,455,460,462,182,167,477) --------------------------------------
    def update_halo(self, *fields):
        if not fields:
            return
        self._follow_torch_streams()
        lists = [_as_list(f) for f in fields]
        ref = lists[0][0]
        if any(t.dtype != ref.dtype for lst in lists for t in lst):
            raise L.Ns3dError("update_halo: all fields of one call must have one element type")
        ext = []
        for lst in lists:
            shp = tuple(lst[0].shape)
            if any(tuple(t.shape) != shp for t in lst):
                raise L.Ns3dError("update_halo: a field has different shapes on different local ranks")
            ext += list(shp)
        L.check(self._typed("update_halo", ref)(self.handle, self._ptrs(lists), (C.c_int * len(ext))(*ext), len(lists)))

    # ---- max_g (multi.jl:21) ----------------------------------------------------------------------------------
    def max_g(self, local_max):
        vals = [float(v) for v in _as_list(local_max)]
        out = C.c_double(0.0)
        L.check(self.lib.ns3d_max_g(self.handle, (C.c_double * len(vals))(*vals), C.byref(out)))
        return out.value

    # ---- gather! (multi.jl:399-403, 528-532) ------------------------------------------------------------------
    def gather(self, A):
        """Halo-stripped blocks of every rank, side by side in rank-coordinate order: a Fortran-ordered numpy array on the
        process that holds rank 0, None elsewhere."""
        self._follow_torch_streams()
        lst = _as_list(A)
        sx, sy, sz = lst[0].shape
        npdt = np.float64 if lst[0].dtype == torch.float64 else np.float32
        is_root = 0 in self.ranks
        shp = (self.dims[0] * (sx - 2), self.dims[1] * (sy - 2), self.dims[2] * (sz - 2))
        out = np.empty(shp, dtype=npdt, order="F") if is_root else None
        L.check(self._typed("gather", lst[0])(self.handle, self._ptrs([lst]), sx, sy, sz,
                                              out.ctypes.data_as(C.c_void_p) if is_root else None))
        return out

    # ---- {X_o .= X; advect!; update_halo!} with a two-plane z halo (outside the reference's multi-rank semantics) ------
    def advect_wide(self, Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, Cf, C_o, dt, dx, dy, dz, faithful=True):
        """multi.jl:475-477 on z-slab ranks, decomposition-independent: the old fields' z halo is widened to two planes, so that
        departure points up to two planes away read the neighbour instead of being clamped to the local array; all four new
        fields (C too) get their halo.  P ranks then reproduce the one-rank time step bit for bit (|δz| < 2 cells)."""
        self._follow_torch_streams()
        ref = _as_list(Cf)[0]
        L.check(self._typed("advect_wide", ref)(self.handle, self._ptrs([Vx]), self._ptrs([Vx_o]), self._ptrs([Vy]), self._ptrs([Vy_o]),
                                                self._ptrs([Vz]), self._ptrs([Vz_o]), self._ptrs([Cf]), self._ptrs([C_o]),
                                                C.c_double(dt), C.c_double(dx), C.c_double(dy), C.c_double(dz), 1 if faithful else 0))

    # ---- pseudo-transient loop of the z-slab ranks -------------------------------------------------------------
    def reserve_cus(self, n):
        """ns3d_mgpu_reserve_cus: every local rank's compute launches leave n compute units to the exchange's kernels.  Returns the
        ranks' compute streams (torch streams) — order tensor work with them."""
        return [c.reserve_cus(n) for c in self.contexts]

    def set_interior_chunks(self, chunks):
        """ns3d_mgpu_set_interior_chunks: the interior sweep of a z-slab pass in `chunks` launches (1: one)."""
        L.check(self.lib.ns3d_mgpu_set_interior_chunks(self.handle, int(chunks)))

    def set_temporal(self, depth):
        L.check(self.lib.ns3d_mgpu_set_temporal(self.handle, int(depth)))

    def slab_load(self, Pr, dPrdtau, divV, p):
        self._follow_torch_streams()
        ref = _as_list(Pr)[0]
        L.check(self._typed("slab_load", ref)(self.handle, self._ptrs([Pr]), self._ptrs([dPrdtau]), self._ptrs([divV]),
                                              C.byref(p)))

    def slab_plan(self):
        """Measure tile shapes and iterations per pass for the loaded state's interior sweeps (same depth on every rank)."""
        L.check(self.lib.ns3d_slab_plan(self.handle))
        return self.pass_depth()

    def pass_depth(self):
        return int(self.lib.ns3d_mgpu_pass_depth(self.handle))

    def ghost_depth(self):
        """Ghost planes per seam of the loaded solve state (pass depth − 1 once ns3d_slab_plan has run)."""
        return int(self.lib.ns3d_mgpu_ghost_depth(self.handle))

    def slab_iterate(self, n):
        L.check(self.lib.ns3d_slab_iterate(self.handle, int(n)))

    def slab_residual(self):
        out = C.c_double(0.0)
        L.check(self.lib.ns3d_slab_residual(self.handle, C.byref(out)))
        return out.value

    def slab_store(self, Pr, dPrdtau):
        ref = _as_list(Pr)[0]
        L.check(self._typed("slab_store", ref)(self.handle, self._ptrs([Pr]), self._ptrs([dPrdtau])))

    def pt_solve_slab(self, Pr, dPrdtau, divV, p, eps, niter, nchk, err_mul, err_div):
        """The inner loop multi.jl:458-471 on every local rank (global residual). Returns (iters_done, [err …])."""
        self._follow_torch_streams()
        ref = _as_list(Pr)[0]
        cap = niter // max(nchk, 1) + 1
        hist = (C.c_double * cap)()
        it, nchecks = C.c_int(0), C.c_int(0)
        L.check(self._typed("pt_solve_slab", ref)(self.handle, self._ptrs([Pr]), self._ptrs([dPrdtau]), self._ptrs([divV]),
                                                  C.byref(p), C.c_double(eps), int(niter), int(nchk), C.c_double(err_mul),
                                                  C.c_double(err_div), C.byref(it), hist, cap, C.byref(nchecks)))
        return it.value, list(hist[: nchecks.value])


class MgpuGrid:
    """The driver-facing grid object (same surface as halo.ZSlabGrid) on top of a MultiGpu."""

    def __init__(self, mg, nx, ny, nz):
        self.mg = mg
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.P, self.nlocal, self.local_ranks = mg.P, mg.nlocal, list(mg.ranks)
        self.me = self.local_ranks[0]
        self.dims = tuple(mg.dims)
        self.local_coords = list(mg.coords)          # MPI_Cart_coords per local rank
        self.coords = self.local_coords[0]
        self.transport = mg.transport
        self.contexts = mg.contexts

    def nx_g(self):
        return self.dims[0] * (self.nx - 2) + 2

    def ny_g(self):
        return self.dims[1] * (self.ny - 2) + 2

    def nz_g(self):
        return self.dims[2] * (self.nz - 2) + 2

    def is_root(self):
        return 0 in self.local_ranks

    def z_slabs(self):
        return self.dims[0] == 1 and self.dims[1] == 1

    def z_lo_is_halo(self, l=0):
        return self.local_coords[l][2] > 0

    def z_hi_is_halo(self, l=0):
        return self.local_coords[l][2] < self.dims[2] - 1

    def update_halo(self, *fields):
        self.mg.update_halo(*fields)

    def max_g(self, local_max):
        return self.mg.max_g(local_max)

    def gather_fields(self, A):
        return self.mg.gather(A)

    def barrier(self):
        self.mg.sync()
