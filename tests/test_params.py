"""Host logic: parameter derivation of the two drivers (navierstokes3d_amd/params.py) against the oracle's independent
restatement and against literal values of the reference scripts."""
import math


def test_multi_params_match_oracle_restatement():
    from navierstokes3d_amd.params import multi_params
    from oracle.driver_ref import multi_params as ref_params
    for nx, P, c in ((63, 1, 0), (255, 1, 0), (24, 1, 0), (20, 3, 1), (512, 8, 7)):
        a, b = multi_params(nx, P, c), ref_params(nx, P)
        for k in ("nx", "ny", "nz", "nz_g", "niter", "nchk", "dx", "dy", "dz", "dt", "dtau", "damp", "g", "a2", "b2",
                  "ox", "oy", "sinb", "cosb", "rho", "mu", "vin", "psc", "eps", "lx", "ly", "lz"):
            assert getattr(a, k) == b[k], (nx, P, k)


def test_multi_params_literals():
    from navierstokes3d_amd.params import multi_params
    p = multi_params(63)
    assert (p.nx, p.ny, p.nz) == (63, 38, 38)                 # ceil(63*0.6) (multi.jl:323-324)
    assert p.niter == 3150 and p.nchk == 37                   # multi.jl:328-329
    assert p.dt == p.dx == 1.0 / 63                           # CFL_adv*max(dx)/vin wins (multi.jl:339)
    assert p.dtau == (1.0 / math.sqrt(3.1)) * p.dx and p.damp == 2 / 63
    assert p.g == 0.0 and p.owns_inlet and p.owns_outlet      # multi.jl:316,164,179 (App. B10)
    p = multi_params(255)
    assert (p.ny, p.nz, p.niter, p.nchk) == (153, 153, 12750, 152)   # README.md:59 "255x153x153"
    p = multi_params(130, 8, 3)
    assert p.nz_g == 8 * (78 - 2) + 2 and p.niter == 50 * p.nz_g and p.damp == 2 / 130
    assert p.dz == p.lz / p.nz_g


def test_gpu_params():
    from navierstokes3d_amd.params import gpu_params
    from oracle.driver_ref import gpu_params as ref_params
    a, b = gpu_params(255), ref_params(255)
    for k in ("nx", "ny", "nz", "niter", "nchk", "dx", "dy", "dz", "dt", "dtau", "damp", "g", "ox"):
        assert getattr(a, k) == b[k], k
    assert a.niter == 50 * 153 and a.g == 9.81 and a.ox == -0.3      # gpu.jl:48,38,29


def test_gpu_initial_fields_match_oracle():
    import numpy as np
    from navierstokes3d_amd.driver import gpu_initial_fields
    from navierstokes3d_amd.params import gpu_params
    from oracle.driver_ref import gpu_initial_fields as ref_fields, gpu_params as ref_params
    Vx, Pr = gpu_initial_fields(gpu_params(40))
    Vx_r, Pr_r = ref_fields(ref_params(40))
    assert np.array_equal(Vx, Vx_r) and np.array_equal(Pr, Pr_r)
    assert Pr[0, 0, -1] > 0 and Pr[0, 0, 0] > Pr[0, 0, -1]           # hydrostatic: heavier at the bed


def test_algorithmic_bytes():
    import bench
    # 5 318.6 MB per PT iteration at 512³ fp64, 232.4 MB at 255×153×153
    assert round(bench.algorithmic_bytes(512, 512, 512, 8) / 1e6, 1) == 5318.6
    assert round(bench.algorithmic_bytes(255, 153, 153, 8) / 1e6, 1) == 232.4


def _read_png(path):
    """minimal PNG reader for the files vis.write_png makes (8-bit RGB, filter 0): returns (h, w, 3) uint8"""
    import struct
    import zlib
    import numpy as np
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
        chunks.append((tag, data))
        pos += 12 + n
    assert [t for t, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[0][1][:10])
    assert (depth, ctype) == (8, 2)
    rows = np.frombuffer(zlib.decompress(chunks[1][1]), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert not rows[:, 0].any()
    return rows[:, 1:].reshape(h, w, 3)


def test_heatmap_png_files(tmp_path):
    """vis.py: valid PNG (signature, chunk CRCs, inflatable rows); orientation of heatmap(x, y, A'): x to the right, y upwards;
    colour limits clamp; NaN is grey; the frame writers use the reference's file names (multi.jl:435-443, gpu.jl:101-115)."""
    import numpy as np
    from navierstokes3d_amd import vis
    A = np.zeros((6, 4))
    A[5, 0] = 1.0                     # largest x, smallest y → bottom right
    A[0, 3] = np.nan                  # smallest x, largest y → top left
    shp = vis.heatmap_png(str(tmp_path / "a.png"), A, (0.0, 1.0), min_side=8)
    img = _read_png(tmp_path / "a.png")
    assert img.shape[:2] == shp == (8, 12)                     # 4×6 cells drawn as 2×2 blocks
    assert tuple(img[-1, -1]) == (252, 255, 164) and tuple(img[0, 0]) == (128, 128, 128) and tuple(img[0, -1]) == (0, 0, 4)
    vis.heatmap_png(str(tmp_path / "b.png"), A * 10 - 5, (0.0, 1.0), min_side=1)          # clamped at both ends
    img = _read_png(tmp_path / "b.png")
    assert img.shape == (4, 6, 3) and tuple(img[-1, -1]) == (252, 255, 164) and tuple(img[-1, 0]) == (0, 0, 4)
    rng = np.random.default_rng(1)
    fv = [rng.random(s) for s in ((10, 6, 6), (10, 6, 6), (11, 6, 6), (10, 7, 6), (10, 6, 7))]
    names = vis.save_frame_multi(fv, 8, 8, 3, str(tmp_path / "viz3D_out"))
    assert sorted(n.split("/")[-1] for n in names) == sorted("3D_NavierStokes_%s_%s_0003.png" % (t, f) for t in ("xy", "xz")
                                                             for f in ("C", "Pr", "Vx", "Vy", "Vz"))
    assert _read_png(names[0]).shape[2] == 3
    g = dict(Pr=fv[0], C=fv[1], Vx=fv[2], Vy=fv[3], Vz=fv[4])
    names = vis.save_frame_gpu(g, 6, 6, 0, str(tmp_path / "viz3D_out"))
    assert sorted(n.split("/")[-1] for n in names) == sorted("3D_NavierStokes_%s%s_0000.png" % (t, f) for t in ("", "long_")
                                                             for f in ("C", "Pr", "Vx", "Vy", "Vz"))
