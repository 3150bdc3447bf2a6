"""Mid-plane heat maps of the drivers' `do_vis` branches (scripts/NavierStokes3D_multi_gpu.jl:416-443,486-513,
scripts/NavierStokes3D_gpu.jl:90-115,143-165) as plain PNG files — same directory, file names, slices and colour limits;
no axes, labels or colour bar (the reference draws them with Plots.jl, which is outside the hot path and absent here), and
gpu.jl's residual-history plot (`3D_NavierStokes_iter_*.png`) is not drawn.  Pure Python + NumPy: a PNG is zlib-compressed
rows of RGB bytes in four chunks.
"""
import math
import os
import struct
import zlib

import numpy as np

# anchor colours of the "inferno" map (Plots.jl's default gradient), linearly interpolated
_INFERNO = np.array([[0, 0, 4], [40, 11, 84], [101, 21, 110], [159, 42, 99], [212, 72, 66], [245, 125, 21], [250, 193, 39],
                     [252, 255, 164]], dtype=np.float64)


def write_png(path, rgb):
    """rgb: uint8 array (height, width, 3), row 0 on top."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, c = rgb.shape
    if c != 3:
        raise ValueError("write_png: expected (h, w, 3)")
    raw = np.empty((h, 1 + 3 * w), dtype=np.uint8)
    raw[:, 0] = 0                                   # filter type 0 (none) on every row
    raw[:, 1:] = rgb.reshape(h, 3 * w)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n")
        fh.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))     # 8-bit RGB
        fh.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        fh.write(chunk(b"IEND", b""))


def colour_map(t):
    """t in [0,1] (NaN → grey) → uint8 RGB"""
    t = np.asarray(t, dtype=np.float64)
    nan = ~np.isfinite(t)
    u = np.clip(np.where(nan, 0.0, t), 0.0, 1.0) * (len(_INFERNO) - 1)
    i = np.minimum(u.astype(np.int64), len(_INFERNO) - 2)
    f = (u - i)[..., None]
    rgb = _INFERNO[i] * (1.0 - f) + _INFERNO[i + 1] * f
    rgb[nan] = (128.0, 128.0, 128.0)
    return np.rint(rgb).astype(np.uint8)


def heatmap_png(path, A, clims=None, min_side=256):
    """heatmap(x, y, A') of the reference: A[ix, iy] with x to the right and y UPWARDS; clims = (lo, hi) or the finite range
    of A.  Cells are drawn as k×k pixel blocks so that the shorter side has at least min_side pixels."""
    A = np.asarray(A, dtype=np.float64)
    if clims is None:
        fin = A[np.isfinite(A)]
        lo, hi = (float(fin.min()), float(fin.max())) if fin.size else (0.0, 1.0)
    else:
        lo, hi = float(clims[0]), float(clims[1])
    if not hi > lo:
        hi = lo + 1.0
    img = colour_map((A.T[::-1, :] - lo) / (hi - lo))          # rows: y from top (max) to bottom
    k = max(1, int(math.ceil(min_side / max(1, min(img.shape[0], img.shape[1])))))
    if k > 1:
        img = np.repeat(np.repeat(img, k, axis=0), k, axis=1)
    write_png(path, img)
    return img.shape[:2]


_MULTI_CLIMS = {"Pr": (-1.5, 1.5), "C": (0.0, 1.0), "Vx": (-0.25, 1.5), "Vy": (-1.0, 1.0), "Vz": (-1.0, 1.0)}   # multi.jl:424-434


def _mid(n):
    return int(math.ceil(n / 2)) - 1            # ceil(Int, n/2), 1-based → 0-based


def save_frame_multi(fields_v, ny_g, nz_g, iframe, outdir="viz3D_out"):
    """multi.jl:424-443 / :487-512: x-y planes at index ceil(nz_g()/2) and x-z planes at ceil(ny_g()/2) of the gathered,
    halo-stripped arrays C_v, Pr_v, Vx_v, Vy_v, Vz_v (in that order in `fields_v`), fixed colour limits."""
    os.makedirs(outdir, exist_ok=True)
    out = []
    for name, A in zip(("C", "Pr", "Vx", "Vy", "Vz"), fields_v):
        kz, jy = min(_mid(nz_g), A.shape[2] - 1), min(_mid(ny_g), A.shape[1] - 1)
        for tag, S in (("xy", A[:, :, kz]), ("xz", A[:, jy, :])):
            path = os.path.join(outdir, "3D_NavierStokes_%s_%s_%04d.png" % (tag, name, iframe))
            heatmap_png(path, S, _MULTI_CLIMS[name])
            out.append(path)
    return out


def save_frame_gpu(fields, ny, nz, iframe, outdir="viz3D_out"):
    """gpu.jl:96-115 / :144-165: planes at ceil(nz/2) (`3D_NavierStokes_<F>_%04d.png`) and ceil(ny/2) (`…_long_<F>_…`) of the
    full device arrays Pr, C, Vx, Vy, Vz (dict or namespace), colour range = the data's."""
    os.makedirs(outdir, exist_ok=True)
    out = []
    for name in ("Pr", "C", "Vx", "Vy", "Vz"):
        A = fields[name] if isinstance(fields, dict) else getattr(fields, name)
        kz, jy = min(_mid(nz), A.shape[2] - 1), min(_mid(ny), A.shape[1] - 1)
        for tag, S in (("", A[:, :, kz]), ("long_", A[:, jy, :])):
            path = os.path.join(outdir, "3D_NavierStokes_%s%s_%04d.png" % (tag, name, iframe))
            heatmap_png(path, S)
            out.append(path)
    return out
