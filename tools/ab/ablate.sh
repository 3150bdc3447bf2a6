#!/bin/bash
# DIAGNOSTIC: how long the 512^3 four-iteration pass takes with parts of it compiled out (-DNS3D_ABL bits, WRONG results, --no-verify):
#   1 no global loads in the z-march   2 no stores   4 no LDS traffic   8 no barrier per z-step
# Build here (no GPU needed):  for v in 1 2 3 4 8 12 7 15; do tools/ab/build_variant.sh abl$v "-DNS3D_ABL=$v"; done; tools/ab/build_variant.sh abl0 ""
# Run on the box:              gpurun -- 'bash tools/ab/ablate.sh'
export NS3D_BENCH_NO_TRAFFIC=1 NS3D_BENCH_NO_CONFIG_B=1
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for l in tools/ab/libns3d_abl*.so; do
  n=$(basename $l .so)
  NS3D_LIB=$PWD/$l timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-strong --no-verify ${ABL_ARGS:---depth 4 --variantn 2800} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', 'run $i', 'ms per pass', round(d['roofline']['kernel_ms'], 4), 'Mcells*iter/s', round(d['value']))"
done
done
