"""z-slab implicit global grid: the subset of ImplicitGlobalGrid.jl the reference driver uses
(scripts/NavierStokes3D_multi_gpu.jl:325 init_global_grid, :371… update_halo!, :21 max_g, :399 gather!,
:534 finalize_global_grid), re-designed for one node of MI355X: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI), 1-D decomposition along z so that every halo message is one contiguous
xy-plane of the packed column-major arrays (no pack/unpack kernels).

Semantics kept from ImplicitGlobalGrid [upstream, SURVEY.md §2.4]: overlap 2, halo width 1; an array whose
local z extent is nz+s has overlap ol = 2+s, sends plane `ol` (1-based) to the lower neighbour and plane
`size-(ol-1)` to the upper one and receives into planes 1 / size; arrays with ol < 2 have no halo; physical
(non-periodic) ends are left untouched; global size nz_g = P·(nz−2)+2.

Transports: "device" hands device planes straight to RCCL (xGMI peer-to-peer);  "host" stages planes through
host memory (what ImplicitGlobalGrid does without IGG_CUDAAWARE_MPI, scripts/runme3D.sh:15-16) and is what the
gloo backend uses — it makes the N>1 path testable on CPU-only machines and on a single-GPU box.
"""
import numpy as np
import torch
import torch.distributed as dist


class ZSlabGrid:
    def __init__(self, nx, ny, nz, group=None, transport="auto"):
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.me = dist.get_rank(group)
            self.P = dist.get_world_size(group)
            self.backend = dist.get_backend(group)
        else:
            self.me, self.P, self.backend = 0, 1, None
        self.dims = (1, 1, self.P)
        self.coords = (0, 0, self.me)
        self.nlocal, self.local_ranks = 1, [self.me]     # one rank per process (mgpu.MgpuGrid may hold several)
        self.lower = self.me - 1 if self.me > 0 else None
        self.upper = self.me + 1 if self.me < self.P - 1 else None
        if transport == "auto":
            transport = "device" if self.backend == "nccl" else "host"
        self.transport = transport
        self._host_bufs = {}

    # ---- ImplicitGlobalGrid accessors --------------------------------------------------------------------
    def nx_g(self):
        return self.nx

    def ny_g(self):
        return self.ny

    def nz_g(self):
        return self.P * (self.nz - 2) + 2

    def z_lo_is_halo(self):
        return self.lower is not None

    def z_hi_is_halo(self):
        return self.upper is not None

    def _global_rank(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    # ---- planes -------------------------------------------------------------------------------------------
    @staticmethod
    def plane(A, k):
        """Contiguous xy-plane k (0-based) of a column-major (sx,sy,sz) tensor, as a (sy,sx) row-major view."""
        return A.permute(2, 1, 0)[k]

    def halo_planes(self, A):
        """(send_to_lower, send_to_upper, recv_from_lower, recv_from_upper) 0-based plane indices, or None
        when the array has no halo in z (overlap < 2)."""
        sz = A.shape[2]
        ol = 2 + (sz - self.nz)
        if ol < 2:
            return None
        return ol - 1, sz - ol, 0, sz - 1

    # ---- update_halo! -------------------------------------------------------------------------------------
    def update_halo(self, *fields):
        """update_halo!(A…) (multi.jl:371,373,450,453,455,460,462,182,167,477): blocking for the caller's
        stream semantics (the exchanged planes are ready for the next kernel on the current stream)."""
        fields = [f[0] if isinstance(f, (list, tuple)) else f for f in fields]   # per-local-rank lists of one
        work = self.start_halo(*fields)
        self.finish_halo(work)

    def start_halo(self, *fields):
        """Post the plane exchange of `fields` and return a handle for finish_halo (lets the caller overlap
        interior work with the exchange)."""
        if self.P == 1:
            return None
        if self.transport == "device":
            ops = []
            for A in fields:
                hp = self.halo_planes(A)
                if hp is None:
                    continue
                s_lo, s_hi, r_lo, r_hi = hp
                if self.lower is not None:
                    ops.append(dist.P2POp(dist.isend, self.plane(A, s_lo), self._global_rank(self.lower), self.group))
                    ops.append(dist.P2POp(dist.irecv, self.plane(A, r_lo), self._global_rank(self.lower), self.group))
                if self.upper is not None:
                    ops.append(dist.P2POp(dist.isend, self.plane(A, s_hi), self._global_rank(self.upper), self.group))
                    ops.append(dist.P2POp(dist.irecv, self.plane(A, r_hi), self._global_rank(self.upper), self.group))
            return ("device", dist.batch_isend_irecv(ops) if ops else [])
        # host-staged transport
        reqs, unpack = [], []
        for idx, A in enumerate(fields):
            hp = self.halo_planes(A)
            if hp is None:
                continue
            s_lo, s_hi, r_lo, r_hi = hp
            for nb, s_k, r_k, tag in ((self.lower, s_lo, r_lo, 0), (self.upper, s_hi, r_hi, 1)):
                if nb is None:
                    continue
                src = self.plane(A, s_k)
                sbuf = src.detach().to("cpu", copy=True).contiguous()
                rbuf = torch.empty_like(sbuf)
                # tags: direction-coded so that the two messages between a pair never cross
                reqs.append(dist.isend(sbuf, self._global_rank(nb), self.group, tag=2 * idx + tag))
                reqs.append(dist.irecv(rbuf, self._global_rank(nb), self.group, tag=2 * idx + (1 - tag)))
                unpack.append((self.plane(A, r_k), rbuf, sbuf))
        return ("host", reqs, unpack)

    def planes(self, A, k, n):
        """n consecutive xy-planes starting at plane k: one contiguous (n,sy,sx) row-major block."""
        return A.permute(2, 1, 0)[k:k + n]

    def start_exchange(self, to_lower=(), from_lower=(), to_upper=(), from_upper=()):
        """Post a neighbour exchange of arbitrary contiguous blocks (used by the two-plane-deep ghost exchange of the
        temporally blocked PT loop).  The i-th block sent to a neighbour lands in that neighbour's i-th receive block
        from this side.  Returns a handle for finish_halo."""
        if self.P == 1:
            return None
        sides = []
        if self.lower is not None:
            sides.append((self.lower, list(to_lower), list(from_lower), 0))
        if self.upper is not None:
            sides.append((self.upper, list(to_upper), list(from_upper), 1))
        if self.transport == "device":
            ops = []
            for nb, sends, recvs, _ in sides:
                for t in sends:
                    ops.append(dist.P2POp(dist.isend, t, self._global_rank(nb), self.group))
                for t in recvs:
                    ops.append(dist.P2POp(dist.irecv, t, self._global_rank(nb), self.group))
            return ("device", dist.batch_isend_irecv(ops) if ops else [])
        reqs, unpack = [], []
        for nb, sends, recvs, side in sides:
            for idx, t in enumerate(sends):
                sbuf = t.detach().to("cpu", copy=True).contiguous()
                reqs.append(dist.isend(sbuf, self._global_rank(nb), self.group, tag=1000 + 2 * idx + side))
                unpack.append((None, None, sbuf))
            for idx, t in enumerate(recvs):
                rbuf = torch.empty(t.shape, dtype=t.dtype, device="cpu")
                reqs.append(dist.irecv(rbuf, self._global_rank(nb), self.group, tag=1000 + 2 * idx + (1 - side)))
                unpack.append((t, rbuf, None))
        return ("host", reqs, unpack)

    def finish_halo(self, work):
        if work is None:
            return
        if work[0] == "device":
            for w in work[1]:
                w.wait()
            return
        _, reqs, unpack = work
        for r in reqs:
            r.wait()
        for dst, rbuf, _keep in unpack:
            if dst is not None:
                dst.copy_(rbuf)

    # ---- max_g --------------------------------------------------------------------------------------------
    def max_g(self, local_max):
        """max_g(A) (multi.jl:21): MPI.Allreduce(max_l, MAX).  NaN-propagating like the local maximum."""
        if isinstance(local_max, (list, tuple)):
            local_max = local_max[0]
        if self.P == 1:
            return float(local_max)
        dev = "cuda" if (self.backend == "nccl") else "cpu"
        isnan = 1.0 if local_max != local_max else 0.0
        t = torch.tensor([0.0 if isnan else float(local_max), isnan], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        v = t.cpu()
        return float("nan") if v[1].item() > 0 else float(v[0].item())

    # ---- gather! ------------------------------------------------------------------------------------------
    def gather(self, A_inn):
        """gather!(A_inn, A_v) (multi.jl:399-403, 528-532): rank 0 receives every rank's halo-stripped block
        and concatenates along z; other ranks get None.  A_inn: numpy array (Fortran order)."""
        A_inn = np.asfortranarray(A_inn)
        if self.P == 1:
            return A_inn
        t = torch.from_numpy(np.ascontiguousarray(A_inn.transpose(2, 1, 0)))
        if self.backend == "nccl":
            t = t.cuda()
        if self.me == 0:
            parts = [torch.empty_like(t) for _ in range(self.P)]
            dist.gather(t, parts, dst=self._global_rank(0), group=self.group)
            return np.asfortranarray(torch.cat([q.cpu() for q in parts], dim=0).numpy().transpose(2, 1, 0))
        dist.gather(t, None, dst=self._global_rank(0), group=self.group)
        return None

    def barrier(self):
        if self.P > 1:
            dist.barrier(group=self.group)


def init_global_grid(nx, ny, nz, group=None, transport="auto"):
    """init_global_grid(nx,ny,nz) (multi.jl:325) for dims=(1,1,P). Returns (me, dims, grid)."""
    g = ZSlabGrid(nx, ny, nz, group, transport)
    return g.me, g.dims, g
