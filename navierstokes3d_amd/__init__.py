"""navierstokes3d_amd — MI355X-native (gfx950) hot path of mattbuergler/NavierStokes3D.

Hand-written HIP kernels behind a C ABI (include/ns3d.h, libns3d.so) plus the host-side mirror of the reference's
kernel layer (`kernels`), drivers (`driver`) and z-slab implicit global grid (`halo`).  No CPU fallback exists.
"""
from . import lib  # noqa: F401

__all__ = ["lib", "kernels", "driver", "halo", "params", "build"]
