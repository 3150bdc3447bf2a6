"""bench.py keeps its contract: one JSON line on stdout with the driver's keys plus `roofline` and `cpu_baseline`
(small grid, a handful of steps; the numbers themselves are not judged here)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--mode", "fast", "--dtype", "f32", "--no-cpu-baseline", "--no-traffic"],
                                   ["--steps", "5", "--no-cpu-baseline", "--no-traffic"]])
def test_bench_prints_one_json_line(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--grid", "96", "--steps", "6", "--warmup", "2", "--cpu-iters", "2"] + extra
    out = subprocess.run(cmd, cwd=ROOT, check=True, capture_output=True, text=True, timeout=600).stdout
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["config"]["finite"] is True and "workload" in d["config"]
    # every line certifies itself: one pass of the timed kernel instance against single sweeps, on the device
    v = d["config"]["verify"]
    assert d["config"]["verified"] is True and v["iterations"] == d["config"]["pt_depth"] and v["rel_l2"] <= 1e-6
    assert v["bitwise"] is True or "fast" in extra
    assert d["config"]["arith_build"] == ("fast" if "fast" in extra else "strictx")          # dx = 1/96: not a power of two
    assert d["strong"]["value"] == d["value"] and d["strong"]["global_grid"] == d["config"]["global_grid"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0 < r["frac"] <= 1.0                                   # a physical fraction: bytes one launch must move
    assert (r["traffic"] is None) == (r["traffic_source"] == "none")
    if "--no-traffic" not in extra:
        # measured live: two short rocprofv3 --pmc child runs of the timed kernel instance (FETCH_SIZE x2 + WRITE_SIZE).  What the
        # memory side moved for one pass over a 96^3 grid: at least the two output arrays, at most a few passes' worth
        assert r["traffic_source"].startswith("measured on this box"), r["traffic_source"]
        must = r["bytes_per_launch"]
        assert 0.3 * must < r["traffic"] < 6 * must and r["hbm_gbps_measured"] > 0
    else:
        assert r["traffic_source"].startswith("profiles lookup") or r["traffic_source"] == "none"
    assert r["frac"] * (1 - 1e-9) <= r["effective_frac"] <= r["pt_iterations_per_launch"] * r["frac"] * (1 + 1e-9)
    assert abs(d["hbm_gbps_algorithmic"] - r["effective_gbps"]) < 1e-6 * r["effective_gbps"]
    # the reference's own grid beside the synthetic one (north_star: "throughput on the 255×153×153 cylinder case …"): STRICT takes the
    # exact-division build there, FAST is the product mode; both self-verified
    b = d["config_b"]
    assert b["grid"] == [255, 153, 153] and "error" not in b
    assert b["strict"]["arith_build"] == "strictx" and b["fast"]["arith_build"] == "fast"
    for m in ("strict", "fast"):
        assert b[m]["value"] > 0 and b[m]["verified"] is True and b[m]["finite"] is True and 0 < b[m]["roofline_frac"] <= 1.0
    if "--no-cpu-baseline" not in extra:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_bench_gpus_2_as_typed_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks (torch.distributed.run as a child,
    before any GPU call) and relays rank 0's line.  On this one-GPU box the two ranks share the device, so the collective
    transport choice is the host-staged one over gloo — and the line says so; the RCCL data plane needs the multi-GPU node."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "96", "--steps", "6", "--warmup", "2"]
    out = subprocess.run(cmd, cwd=ROOT, check=True, capture_output=True, text=True, timeout=900, env=env).stdout
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["finite"] is True
    assert d["config"]["global_grid"] == [96, 96, 2 * 94 + 2] and d["config"]["decomposition"] == "z-slabs x2"
    assert "transport" in d["config"] and "rccl_ranks" in d["config"]
    import torch
    if torch.cuda.device_count() < 2:
        assert d["config"]["transport"].startswith("host-staged over gloo") and d["config"]["rccl_ranks"] == 0
    else:
        assert d["config"]["transport"].startswith("RCCL") and d["config"]["rccl_ranks"] == 2
    assert 0 < d["roofline"]["frac"] <= 1.0 and "cpu_baseline" not in d
    # the schedule certifies itself (one pass of the slab schedule against {single sweep; update_halo!(Pr)} per iteration) …
    assert d["config"]["verified"] is True and d["config"]["verify"]["bitwise"] is True
    # … and the same GLOBAL grid split in z is measured next to the weak headline (BASELINE: "512³ grid, 1/2/4/8 MI355X")
    st = d["strong"]
    assert st["scaling"] == "strong" and st["planes_per_rank"] == 49 and st["global_grid"] == [96, 96, 96]
    assert st["value"] > 0 and st["verified"] is True and st["finite"] is True


def test_bench_exits_nonzero_when_the_self_check_fails():
    """config.verified is load-bearing: with NS3D_BENCH_SABOTAGE=1 the checker perturbs one value of the reference side and the
    run must print verified: false and exit with a non-zero status."""
    env = dict(os.environ, NS3D_BENCH_SABOTAGE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--grid", "96", "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
           "--no-traffic"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["verified"] is False and d["config"]["verify"]["bitwise"] is False
