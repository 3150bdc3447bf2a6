"""Shared helpers for the parity tests (seeded synthetic fields, comparison metrics)."""
import numpy as np

SHAPES = {
    "c": lambda nx, ny, nz: (nx, ny, nz),
    "vx": lambda nx, ny, nz: (nx + 1, ny, nz),
    "vy": lambda nx, ny, nz: (nx, ny + 1, nz),
    "vz": lambda nx, ny, nz: (nx, ny, nz + 1),
    "s": lambda nx, ny, nz: (nx - 1, ny - 1, nz - 1),
    "i": lambda nx, ny, nz: (nx - 2, ny - 2, nz - 2),
}


def rnd(seed, shape, dtype=np.float64, lo=-1.0, hi=1.0):
    """U(lo,hi) from a seeded Mersenne Twister (SURVEY.md §8d 'value distributions')."""
    rng = np.random.Generator(np.random.MT19937(seed))
    return np.asfortranarray(rng.uniform(lo, hi, size=shape).astype(dtype))


def fields(nx, ny, nz, kinds, seed0=1, dtype=np.float64):
    return [rnd(seed0 + q, SHAPES[k](nx, ny, nz), dtype) for q, k in enumerate(kinds)]


def rel_l2(a, b, den=None):
    """‖a−b‖₂ / ‖b‖₂ (or / den when a reference norm is given, e.g. the norm of the whole velocity vector for one
    component that is pure round-off by symmetry)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    den = np.sqrt(np.sum(b * b)) if den is None else float(den)
    num = np.sqrt(np.sum((a - b) ** 2))
    return float(num / den) if den > 0 else float(num)


def geometry(nx, ny, nz):
    """Non-trivial, non-power-of-two spacings so that divisions are inexact."""
    return dict(dx=1.0 / nx, dy=0.6 / ny, dz=0.7 / nz, mu=1e-3, rho=1000.0, g=9.81, dt=0.013, dtau=0.009,
                damp=2.0 / nx)
