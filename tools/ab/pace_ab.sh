#!/bin/bash
# A/B of the pacing hint (k_pt_sweepN built with -DNS3D_PACE=<slack>, tools/ab/build_variant.sh) on ONE box:
#   gpurun -- 'bash tools/ab/pace_ab.sh "--variantn 2891" "--variantn 2800"'
# every tools/ab/libns3d_*.so, interleaved, twice per argument set; prints value, ms per pass, verified.
export NS3D_BENCH_NO_TRAFFIC=1   # these runs are timed or profiled themselves: no nested rocprofv3 --pmc child runs (bench.py --no-traffic)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/pace_*.json
k=0
for args in "$@"; do
k=$((k+1))
for i in 1 2; do
for l in tools/ab/libns3d_*.so; do
n=$(basename $l .so)
NS3D_LIB=$PWD/$l timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-strong $args > gpurun_out/pace_${k}_${n}_$i.json
done
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/pace_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value']), 'ms/pass', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'v', d['config'].get('ptn_variant'), 'verified', d['config'].get('verified'))
PY
