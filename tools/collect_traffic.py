#!/usr/bin/env python3
"""HBM traffic per k_pt_sweep2 launch from the memory-side counters, the way MI355X_MICROARCH.md §HBM prescribes:
`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in SEPARATE passes of the same bench command (kernel-trace only),
FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B), both counters are in KiB.

    python tools/collect_traffic.py --out gpurun_out/traffic.json [--runs 2:1392,3:100] [--modes strict,fast]     (runs = depth:variant)

Writes {"<nx>x<ny>x<nz>_<dtype>_<mode>_x<depth>_v<variant>": {...}}; copy the result into profiles/pt_sweep_traffic.json.
The profiler gets `python3 bench.py …` itself after `--` (no shell / env hop).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counter, bench_args, workdir):
    shutil.rmtree(workdir, ignore_errors=True)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-f", "csv", "-d", workdir, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    acc = {}
    for f in glob.glob(os.path.join(workdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"].split("(")[0]
            s, n = acc.get(k, (0.0, 0))
            acc[k] = (s + float(row["Counter_Value"]), n + 1)
    return {k: s / n for k, (s, n) in acc.items()}      # mean per dispatch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--runs", default="2:1392", help="depth:variant[,depth:variant…]")
    ap.add_argument("--modes", default="strict")
    ap.add_argument("--dtype", default="f64")
    a = ap.parse_args()
    res = {}
    for mode in a.modes.split(","):
        for run in a.runs.split(","):
            depth, v = run.split(":")
            args = ["--steps", str(6 * int(depth)), "--warmup", depth, "--no-cpu-baseline", "--no-traffic", "--grid", str(a.grid), "--mode", mode,
                    "--dtype", a.dtype, "--depth", depth, "--variant2" if depth == "2" else "--variantn", v]
            per = {c: one_pass(c, args, "/tmp/ns3d_pmc_%s" % c) for c in ("FETCH_SIZE", "WRITE_SIZE")}
            fetch = write = 0.0
            names = []
            for k in per["FETCH_SIZE"]:
                main = "k_pt_sweep2" if depth == "2" else "k_pt_sweepN<%s, %s," % ("double" if a.dtype == "f64" else "float", depth)
                if main in k or "k_pt_faces" in k:
                    fetch += 2.0 * 1024.0 * per["FETCH_SIZE"][k]
                    write += 1024.0 * per["WRITE_SIZE"].get(k, 0.0)
                    names.append(k.replace("void ", ""))
            res["%dx%dx%d_%s_%s_x%s_v%s" % (a.grid, a.grid, a.grid, a.dtype, mode, depth, v)] = {
                "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write,
                "kernels": sorted(names),
                "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950 correction); one "
                        "sweep launch + its boundary-cell launches = `depth` PT iterations"}
            print(v, mode, res[list(res)[-1]]["hbm_bytes_per_launch"] / 1e9, "GB", flush=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
