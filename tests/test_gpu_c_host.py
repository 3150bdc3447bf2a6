"""A compiled C host (tests/c_host/ns3d_c_host.c, gcc -std=c11) drives libns3d.so on the GPU with no Python in between: the
kernel-by-kernel sequence of multi.jl:459-463 and ns3d_pt_iterate must agree bit for bit inside the C program."""
import subprocess

import pytest

from test_abi import build_c_host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("iters", [1, 2, 11, 37])
def test_c_host_runs_the_pt_loop(tmp_path, iters):
    exe = build_c_host(tmp_path)
    r = subprocess.run([exe, str(iters)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert r.stdout.startswith("C HOST OK: %d iterations" % iters)
