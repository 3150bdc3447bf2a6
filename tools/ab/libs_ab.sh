#!/bin/bash
# A/B of library builds (tools/ab/build_variant.sh NAME "-D…") on ONE box, several bench argument sets per call:
#   gpurun -- 'bash tools/ab/libs_ab.sh "--depth 4 --variantn 2891" "--depth 4 --variantn 2800" "--dtype f32"'
# every tools/ab/libns3d_*.so, interleaved, twice per argument set; prints value, ms per pass, verified.
# (profiles/r3_pace_order_ab.log: pacing hint, tile order, halo-ring duties, compiler scheduling strategies)
export NS3D_BENCH_NO_TRAFFIC=1 NS3D_BENCH_NO_CONFIG_B=${NS3D_BENCH_NO_CONFIG_B:-1}   # these runs are timed or profiled themselves: no nested rocprofv3 --pmc child runs (bench.py --no-traffic)
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
    cb = d.get('config_b') or {}
    print(f, round(d['value']), 'ms/pass', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'v', d['config'].get('ptn_variant'), 'verified', d['config'].get('verified'),
          'config_b', {m: (round(v['value']), v.get('verified')) for m, v in cb.items() if isinstance(v, dict) and 'value' in v})
PY
