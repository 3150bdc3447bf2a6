"""Temporally blocked pseudo-transient loop on a z-slab rank (one process per GPU).

The single-GPU fast path advances TWO PT iterations per pass over memory (`ns3d_pt_sweep2`).  Level 2 of the first
interior plane needs level 1 of the seam halo plane, which needs the previous iterate one plane further out — so a
z-slab rank keeps its pressure-solve state in buffers **extended by a second ghost plane per seam**:

    original plane k  ↔  extended plane k + glo          (glo/ghi = 1 where the lower/upper z neighbour exists)
    Pr_ext  (nx,ny,nz+glo+ghi)    dPrdτ_ext (nx-2,ny-2,nz+glo+ghi-2)    ∇V_ext (nx,ny,nz+glo+ghi)

Each launch computes level 1 redundantly on the seam halo plane (bit-identical on both ranks: same inputs, same
arithmetic) and level 2 on the rank's own interior planes; afterwards the two outermost own planes of `Pr` and the
outermost own plane of `dPrdτ` travel to the neighbour's ghost planes (one contiguous block each, RCCL over xGMI).
That is 3 planes per side per two iterations instead of the reference's ≥2 exchanges per single iteration
(multi.jl:460-463, 182).  The seam-adjacent output planes are swept first, their exchange is posted, and the interior
sweep runs behind it.  The iterates are identical to the reference's per-iteration sequence (Jacobi sweeps are
decomposition independent).
"""
import math

from . import kernels as K
from . import lib as L


class SlabPTSolver:
    def __init__(self, ctx, grid, Pr, rho, dt, dtau, damp, dx, dy, dz, bc_kind=L.NS3D_BC_MULTI, owns_outlet=True,
                 outlet_val=0.0, g=0.0):
        self.ctx, self.grid = ctx, grid
        nx, ny, nz = Pr.shape
        if nz < 4:
            raise L.Ns3dError("SlabPTSolver needs at least two interior planes per rank")
        self.nx, self.ny, self.nz = nx, ny, nz
        self.glo, self.ghi = int(grid.z_lo_is_halo()), int(grid.z_hi_is_halo())
        self.nze = nz + self.glo + self.ghi
        z = lambda *s: K.zeros(s, Pr.dtype, Pr.device)
        self.P = [z(nx, ny, self.nze), z(nx, ny, self.nze)]
        self.D = [z(nx - 2, ny - 2, self.nze - 2), z(nx - 2, ny - 2, self.nze - 2)]
        self.R = z(nx, ny, self.nze)
        self.pt = K.pt_params(self.P[0], rho, dt, dtau, damp, dx, dy, dz, bc_kind, owns_outlet, outlet_val, g)
        self.k0, self.k1 = 1 + self.glo, nz - 1 + self.glo          # own interior planes in extended indices
        self.ip, self.id = 0, 0                                    # current Pr / dPrdτ buffer
        # the exchanged blocks are fixed views of the four buffers: build them once (keeps the per-launch host work small)
        gr, g, nze = grid, self.glo, self.nze
        self._blk = {}
        for ib, Pq in enumerate(self.P):
            self._blk["P", ib] = (gr.planes(Pq, 1 + g, 2), gr.planes(Pq, 0, 2), gr.planes(Pq, nz - 3 + g, 2),
                                  gr.planes(Pq, nze - 2, 2))
        for ib, Dq in enumerate(self.D):
            self._blk["D", ib] = (gr.planes(Dq, g, 1), gr.planes(Dq, 0, 1), gr.planes(Dq, g + nz - 3, 1),
                                  gr.planes(Dq, nze - 3, 1))

    # ---- state in / out -----------------------------------------------------------------------------------
    def load(self, Pr, dPrdtau, divV):
        g, nz = self.glo, self.nz
        self.ip = self.id = 0
        self.P[0][:, :, g:g + nz] = Pr
        self.D[0][:, :, g:g + nz - 2] = dPrdtau
        self.R[:, :, g:g + nz] = divV
        self._exchange(0, 0, wait=True)                            # deep ghosts of the incoming state

    def store(self, Pr, dPrdtau):
        g, nz = self.glo, self.nz
        Pr[:, :, :] = self.P[self.ip][:, :, g:g + nz]
        dPrdtau[:, :, :] = self.D[self.id][:, :, g:g + nz - 2]

    # ---- ghost exchange: 2 planes of Pr + 1 plane of dPrdτ per seam ------------------------------------------
    def _exchange(self, ip, idd, wait):
        """Own planes (1,2 | nz-3,nz-2 of Pr; first | last of dPrdτ) → the neighbours' ghost planes, buffers ip / idd."""
        gr = self.grid
        ps_lo, pr_lo, ps_hi, pr_hi = self._blk["P", ip]
        ds_lo, dr_lo, ds_hi, dr_hi = self._blk["D", idd]
        work = gr.start_exchange(
            to_lower=[ps_lo, ds_lo] if self.glo else [], from_lower=[pr_lo, dr_lo] if self.glo else [],
            to_upper=[ps_hi, ds_hi] if self.ghi else [], from_upper=[pr_hi, dr_hi] if self.ghi else [])
        if wait:
            gr.finish_halo(work)
            return None
        return work

    def _launch(self, two):
        """One pass: two PT iterations (ns3d_pt_sweep2) or one (ns3d_pt_sweep), seam planes first."""
        src, dst = self.P[self.ip], self.P[self.ip ^ 1]
        dsrc = self.D[self.id]
        idd_out = self.id ^ 1 if two else self.id
        ddst = self.D[idd_out]
        k0, k1 = self.k0, self.k1
        lo_end = min(k0 + 2, k1) if self.glo else k0
        hi_beg = max(k1 - 2, lo_end) if self.ghi else k1

        def sweep(a, b):
            if b <= a:
                return
            if two:
                K.pt_sweep2(src, dst, dsrc, ddst, self.R, self.pt, a, b, ctx=self.ctx)
            else:
                K.pt_sweep(src, dst, dsrc, self.R, self.pt, a, b, ctx=self.ctx)

        sweep(k0, lo_end)
        sweep(hi_beg, k1)
        work = self._exchange(self.ip ^ 1, idd_out, wait=False)
        sweep(lo_end, hi_beg)
        self.grid.finish_halo(work)
        self.ip ^= 1
        if two:
            self.id ^= 1

    def iterate(self, n):
        """n PT iterations {update_dPrdτ!; update_Pr!; set_bc_Pr!; update_halo!(Pr)} (multi.jl:459-463)."""
        it = 0
        while it < n:
            two = (self.ctx_two and it + 2 <= n)
            self._launch(two)
            it += 2 if two else 1

    @property
    def ctx_two(self):
        return getattr(self, "_two", True)

    def set_temporal_blocking(self, on):
        self._two = bool(on)

    def residual(self):
        """max_g(abs.(Rp)) of the current iterate (multi.jl:465-466,21).  The extended slab includes the seam halo
        planes, whose residuals are the neighbour's own values — the global maximum is unchanged."""
        return self.grid.max_g(K.residual_max(self.P[self.ip], self.R, self.pt, ctx=self.ctx))

    def solve(self, eps, niter, nchk, err_mul, err_div, on_check=None):
        """The inner loop multi.jl:458-471 on this rank. Returns (iters_done, [err …])."""
        errs, done, it = [], niter, 0
        while it < niter:
            to_check = nchk - it % nchk if nchk > 0 else niter - it
            two = self.ctx_two and to_check >= 2 and it + 2 <= niter
            self._launch(two)
            it += 2 if two else 1
            if nchk > 0 and it % nchk == 0:                                             # :464
                err = self.residual() * err_mul / err_div                                # :466
                errs.append(err)
                if on_check is not None:
                    on_check(it, err)
                if err < eps or not math.isfinite(err):                                  # :469
                    done = it
                    break
        return done, errs
