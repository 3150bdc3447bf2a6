#!/usr/bin/env python3
"""SQ counters of the k_pt_sweep2 / k_pt_sweepN launches of one bench run (one rocprofv3 --pmc pass, kernel-trace only).

    python tools/collect_sq.py --out gpurun_out/sq.json [--runs 2:1392,3:100] [--modes strict,fast]      (runs = depth:variant)

WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ≈ WAVE_CYCLES (MI355X_MICROARCH.md, counter table).
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
COUNTERS = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
            "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"]
# --set insts: dynamic instruction counts per wave (round 4: what an interior tile really executes)
COUNTERS_INSTS = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
                  "SQ_LDS_BANK_CONFLICT"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--runs", default="2:1392", help="depth:variant[,depth:variant…]")
    ap.add_argument("--modes", default="strict,fast")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--grid", default="512")
    ap.add_argument("--set", default="time", choices=["time", "insts"])
    a = ap.parse_args()
    counters = COUNTERS if a.set == "time" else COUNTERS_INSTS
    res = {}
    for mode, run in [(m, r) for m in a.modes.split(",") for r in a.runs.split(",")]:
        depth, variant = run.split(":")
        wd = "/tmp/ns3d_sq"
        shutil.rmtree(wd, ignore_errors=True)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["-f", "csv", "-d", wd, "-o", "p", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--steps", str(8 * int(depth)), "--warmup", depth, "--no-cpu-baseline", "--no-traffic", "--mode", mode,
               "--dtype", a.dtype, "--grid", a.grid, "--depth", depth, "--variant2" if depth == "2" else "--variantn", variant]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
        acc = {}
        for f in glob.glob(os.path.join(wd, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if not any(kn in row["Kernel_Name"] for kn in (("k_pt_sweep2",) if depth == "2" and int(variant) < 3000 else ("k_pt_sweepN", "k_pt_sweepD"))):
                    continue
                s, n = acc.get(row["Counter_Name"], (0.0, 0))
                acc[row["Counter_Name"]] = (s + float(row["Counter_Value"]), n + 1)
        m = {k: s / n for k, (s, n) in acc.items()}
        if a.set == "time":
            wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            m["fractions_of_wave_cycles"] = {k: round(v / wc, 4) for k, v in m.items() if k.startswith("SQ_") and k != "SQ_WAVE_CYCLES"}
            print(mode, run, m["fractions_of_wave_cycles"], flush=True)
        else:
            w = m.get("SQ_WAVES", 0.0) or 1.0
            m["per_wave"] = {k: round(v / w, 1) for k, v in m.items() if k.startswith("SQ_") and k != "SQ_WAVES"}
            print(mode, run, "waves", w, m["per_wave"], flush=True)
        res["%s_x%s_v%s_%s" % (mode, depth, variant, a.dtype)] = m
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
