"""CPU tests of the boundary: libns3d.so builds for gfx950, loads, exports every symbol include/ns3d.h declares, and
refuses to run without a GPU (no CPU fallback, no oracle in the product path).  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from navierstokes3d_amd import build, lib
    build.build()
    return lib


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "ns3d.h")).read()
    names = set(re.findall(r"\b(ns3d_[a-zA-Z_0-9]+?)(?:_##S)?\s*\(", src))
    names.discard("ns3d_ctx")
    out = set()
    for n in names:
        base = n
        # names inside the NS3D_DECL macro carry the _##S suffix in the text
        if re.search(r"\b%s_##S\s*\(" % re.escape(n), src):
            out.update({base + "_f64", base + "_f32"})
        else:
            out.add(base)
    return out


def test_header_and_binding_agree(L):
    assert _header_symbols() == set(L.exported_symbols())


def test_library_exports_every_declared_symbol(L):
    lib = L.load()
    missing = [s for s in sorted(_header_symbols()) if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.ns3d_version() == 1


def test_code_object_is_gfx950_only(L):
    """The fat binary must carry gfx950 code objects and nothing else (no CUDA path, no dual build)."""
    blob = open(L.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets
    assert b"nvptx" not in blob and b"sm_" + b"90" not in blob


def test_no_gpu_fails_loudly(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the GPU-less build container")
    lib = L.load()
    h = lib.ns3d_create(0, 0)
    assert not h
    assert "no CPU path" in L.last_error()
    from navierstokes3d_amd import kernels
    with pytest.raises(L.Ns3dError):
        kernels.Context(0)
    with pytest.raises(L.Ns3dError):      # CPU tensors are refused before any library call
        kernels.bc_x(torch.zeros(4, 4, 4, dtype=torch.float64))


def test_null_context_is_an_error_not_a_crash(L):
    lib = L.load()
    assert lib.ns3d_sync(None) == 1       # NS3D_ERR_ARG
    assert "null context" in L.last_error()
    p = L.PtParams()
    assert lib.ns3d_pt_iterate_f64(None, None, None, None, ctypes.byref(p), 1) == 1


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "navierstokes3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), fn
                assert "ns3d_ref_" not in txt and "libns3d_oracle" not in txt, fn
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle", bench)]
    body = bench[bench.index("def cpu_baseline"):bench.index("def main")]
    assert len(uses) == 1 and "from oracle" in body


def test_multi_gpu_layer_fails_loudly_without_gpu(L):
    """ns3d_mgpu_create on a machine without a GPU: a NULL handle and a message, never a crash or a CPU path; the RCCL
    loader either finds librccl (this image ships it) or reports NS3D_ERR_RCCL — also without a GPU."""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check")
    lib = L.load()
    devs = (C.c_int * 2)(0, 0)
    assert not lib.ns3d_mgpu_create(2, devs, 16, 16, 8, 0)
    assert b"no HIP device" in lib.ns3d_last_error() or b"HIP" in lib.ns3d_last_error()
    assert not lib.ns3d_mgpu_create(0, devs, 16, 16, 8, 0) and b"P = 0" in lib.ns3d_last_error()
    buf = C.create_string_buffer(L.NS3D_UNIQUE_ID_BYTES)
    rc = lib.ns3d_mgpu_unique_id(buf)
    assert rc in (L.NS3D_OK, L.NS3D_ERR_RCCL)
    assert lib.ns3d_mgpu_unique_id(None) == L.NS3D_ERR_ARG
    for fn in ("ns3d_mgpu_world", "ns3d_mgpu_nlocal", "ns3d_mgpu_nz_g", "ns3d_mgpu_pass_depth"):
        assert getattr(lib, fn)(None) == -1
    assert lib.ns3d_slab_iterate(None, 1) == L.NS3D_ERR_ARG and lib.ns3d_mgpu_sync(None) == L.NS3D_ERR_ARG


def build_c_host(tmp_path):
    """tests/c_host/ns3d_c_host.c with gcc as C11, warnings as errors, against include/ns3d.h and the in-tree libns3d.so"""
    import subprocess
    exe = str(tmp_path / "ns3d_c_host")
    libdir = os.path.join(ROOT, "navierstokes3d_amd")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I",
           "/opt/rocm/include", os.path.join(ROOT, "tests", "c_host", "ns3d_c_host.c"), "-L", libdir, "-lns3d", "-L", "/opt/rocm/lib",
           "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_a_c_host_links(L, tmp_path):
    """include/ns3d.h is a C header (the boundary the reference's `ccall` binds): a C11 translation unit that uses the context,
    the reference-signature kernels, the fused path and the error channel compiles without a warning and links against
    libns3d.so alone (+ the HIP runtime for its own allocations).  Without a GPU it reports the library's message and exits 3."""
    import subprocess
    exe = build_c_host(tmp_path)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 3 and "ns3d_create failed" in r.stderr and "no CPU path" in r.stderr


def test_fake_rccl_double_exports_what_load_rccl_resolves():
    """tests/fake_rccl (the test double behind tests/test_gpu_fake_rccl.py) builds here and exports exactly the ten nccl* symbols
    ns3d_mgpu.cpp's load_rccl() resolves with dlsym — read from the source, so the two cannot drift apart."""
    import ctypes
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "navierstokes3d_amd", "csrc", "ns3d_mgpu.cpp")).read()
    m = re.search(r"SYM\(GetUniqueId\)(.*?)#undef SYM", src, re.S)
    names = ["ncclGetUniqueId"] + ["nccl" + n for n in re.findall(r"SYM\((\w+)\)", m.group(1))]
    assert len(names) == 10
    so = os.path.join(root, "tests", "fake_rccl", "libfake_rccl.so")
    cpp = os.path.join(root, "tests", "fake_rccl", "fake_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(cpp):
        subprocess.check_call(["hipcc", "-O2", "-shared", "-fPIC", "-std=c++17", cpp, "-o", so, "-lrt", "-lpthread"])
    import torch  # noqa: F401 — the HIP runtime the double links against is the one PyTorch ships
    lib = ctypes.CDLL(so)
    for n in names:
        assert hasattr(lib, n), n
    buf = ctypes.create_string_buffer(128)
    assert lib.ncclGetUniqueId(buf) == 0 and buf.raw.startswith(b"/fake_rccl_")
    assert lib.fake_rccl_marker() == 1
