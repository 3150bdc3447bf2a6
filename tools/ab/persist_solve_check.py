import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from navierstokes3d_amd import kernels as K, lib as L
nx, ny, nz = 63, 38, 38
d = 1.0 / nx
res = {}
for mode in ("1", "0"):
    os.environ["NS3D_PERSIST_SOLVE"] = mode
    ctx = K.Context(0, "strict", async_=True)
    ctx.set_persist_mode(1); ctx.set_pt_depth(1)
    torch.manual_seed(1)
    Pr, D, rhs = K.zeros((nx, ny, nz)), K.zeros((nx - 2, ny - 2, nz - 2)), K.zeros((nx, ny, nz))
    rhs.permute(2, 1, 0).uniform_(-1e-3, 1e-3)
    pt = K.pt_params(Pr, 1000.0, d, d / 3.1 ** 0.5, 2.0 / nx, d, 0.6 / ny, 0.6 / nz, L.NS3D_BC_MULTI, True, 0.0, 0.0)
    it, errs = K.pt_solve(Pr, D, rhs, pt, -1.0, 370, 37, 1.0, 1.0, ctx=ctx)
    ctx.sync(); torch.cuda.synchronize()
    res[mode] = (it, list(errs), Pr.clone(), D.clone())
    ctx.close()
a, b = res["1"], res["0"]
print(a[0], b[0], len(a[1]), len(b[1]))
for x, y in zip(a[1], b[1]): print(x, y, x == y)
print("P equal", torch.equal(a[2], b[2]), "D equal", torch.equal(a[3], b[3]))
