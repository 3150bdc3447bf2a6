#!/bin/bash
# Does the power-of-two row / plane pitch of the 512^3 grid cost bandwidth?  The same tiling (9 x 29 tiles of 64x24, one round,
# 512 planes) on grids n x n x 512 with n around 512, FAST mode (no arithmetic difference between spacings) and fp64:
#   gpurun -- 'bash tools/ab/pitch_probe.sh'
export NS3D_BENCH_NO_TRAFFIC=1 NS3D_BENCH_NO_CONFIG_B=1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for n in 506 510 512 514 518 520 522; do
python bench.py --grid $n --grid-nz 512 --mode fast --depth 4 --variantn 2891 --steps 200 --warmup 20 --no-cpu-baseline --no-strong > gpurun_out/pitch_${n}_$rep.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/pitch_${n}_$rep.json").read().strip().splitlines()[-1]); r=d["roofline"]
n=$n
print("n=%d  %7.0f Mcells*it/s  %.4f ms/pass  must-move %.3f GB  %.0f GB/s physical  verified=%s" % (n, d["value"], r["kernel_ms"], r["bytes_per_launch"]/1e9, r["achieved"], d["config"]["verified"]))
PY
done
done
