#!/bin/bash
# Pass chaining (NS3D_PASS_SKIP_FACES / NS3D_PASS_INPUT_OBEYS_BC between the passes of a block) on and off, one box, interleaved:
#   gpurun -- 'bash tools/ab/chain_ab.sh "" "--mode fast" "--grid 510"'      (each argument set: bench.py default + config_b)
export NS3D_BENCH_NO_TRAFFIC=1
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/chain_*.json
k=0
for args in "$@"; do
k=$((k+1))
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --pass-chain $args > gpurun_out/chain_${k}_on_$i.json
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline $args > gpurun_out/chain_${k}_off_$i.json
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/chain_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    cb = d.get('config_b') or {}
    print(f, round(d['value']), 'ms/pass', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'depth', d['config'].get('pt_depth'), 'v', d['config'].get('ptn_variant'), 'verified', d['config'].get('verified'),
          'config_b', {m: (round(v['value']), v.get('pt_depth'), v.get('verified')) for m, v in cb.items() if isinstance(v, dict) and 'value' in v})
PY
