"""Build libns3d.so (gfx950 only) in-tree with hipcc.

    python -m navierstokes3d_amd.build [--force]

The kernel translation unit is compiled four times: STRICT (-ffp-contract=off: reference operation order, IEEE
division, no FMA — bit-identical to the CPU oracle), STRICT with exact division-by-known-divisor (same results,
fewer instructions), STRICT for power-of-two spacings (same results: x/d ≡ x·(1/d) exactly) and FAST
(-ffp-contract=fast + reciprocal constants).
hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libns3d.so")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
UNITS = [
    # (source, object, extra flags)
    ("ns3d_kernels.hip", "ns3d_kernels_strict.o", ["-DNS3D_MODE_STRICT", "-ffp-contract=off"]),
    ("ns3d_kernels.hip", "ns3d_kernels_strictx.o", ["-DNS3D_MODE_STRICT", "-DNS3D_EXACT_RECIP", "-ffp-contract=off"]),
    ("ns3d_kernels.hip", "ns3d_kernels_strictp.o", ["-DNS3D_MODE_STRICT", "-DNS3D_POW2_RECIP", "-ffp-contract=off"]),
    ("ns3d_kernels.hip", "ns3d_kernels_fast.o", ["-DNS3D_MODE_FAST", "-ffp-contract=fast"]),
    ("ns3d_direct.hip", "ns3d_direct.o", []),
    ("ns3d_api.cpp", "ns3d_api.o", ["-x", "hip"]),
    ("ns3d_mgpu.cpp", "ns3d_mgpu.o", ["-x", "hip"]),
]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libns3d.so cannot be built (there is no CPU fallback)")
    return exe


def _sources():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    out.append(os.path.join(HERE, "..", "include", "ns3d.h"))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _sources())


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "ns3d.h")]


def _deps_of(obj, fallback):
    """the repo's own files a unit was compiled from (its -MD dependency file); every header when there is none yet"""
    try:
        words = open(obj + ".d").read().replace("\\\n", " ").split()
    except OSError:
        return fallback
    root = os.path.realpath(os.path.join(HERE, ".."))
    deps = [w for w in words[1:] if os.path.realpath(w).startswith(root)]
    return deps if deps and all(os.path.exists(d) for d in deps) else fallback


def build(force=False, verbose=False, extra_flags=()):
    """Compile every HIP translation unit for gfx950 and link libns3d.so. Returns the library path.  Units whose object is
    newer than their source and every header are not recompiled (the kernel unit is compiled four times: ≈2 min)."""
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    objs = []
    procs = []
    stamp = os.path.join(bdir, "flags.txt")
    flags_now = " ".join(COMMON + list(extra_flags))
    same_flags = os.path.exists(stamp) and open(stamp).read() == flags_now
    for src, obj, flags in UNITS:
        o = os.path.join(bdir, obj)
        objs.append(o)
        deps = _deps_of(o, [os.path.join(CSRC, src)] + _headers())
        if not force and same_flags and os.path.exists(o) and all(os.path.getmtime(d) <= os.path.getmtime(o) for d in deps):
            continue
        cmd = [hipcc] + COMMON + list(flags) + list(extra_flags) + ["-MD", "-MF", o + ".d", "-c", os.path.join(CSRC, src), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    with open(stamp, "w") as f:
        f.write(flags_now)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
