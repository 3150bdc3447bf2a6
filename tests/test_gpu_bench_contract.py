"""bench.py keeps its contract: one JSON line on stdout with the driver's keys plus `roofline` and `cpu_baseline`
(small grid, a handful of steps; the numbers themselves are not judged here)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--mode", "fast", "--dtype", "f32", "--no-cpu-baseline"], ["--steps", "5", "--no-cpu-baseline"]])
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
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if "--no-cpu-baseline" not in extra:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
