#!/bin/bash
# A/B of library builds on ONE GPU box (box-to-box spread is larger than most kernel changes):
#   cp navierstokes3d_amd/libns3d.so tools/ab/libns3d_old.so;  …edit, rebuild…;  cp navierstokes3d_amd/libns3d.so tools/ab/libns3d_new.so
#   gpurun -- 'BENCH_ARGS="--variant2 1392" bash tools/ab/ab.sh'
# Every tools/ab/libns3d_*.so is benchmarked twice, interleaved, through the NS3D_LIB override of navierstokes3d_amd/lib.py.
export NS3D_BENCH_NO_TRAFFIC=1 NS3D_BENCH_NO_CONFIG_B=1   # these runs are timed or profiled themselves: no nested rocprofv3 --pmc child runs (bench.py --no-traffic)
set -e
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_pt.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
fi
rm -f gpurun_out/ab_*.json
for i in 1 2; do
for l in tools/ab/libns3d_*.so; do
n=$(basename $l .so)
NS3D_LIB=$PWD/$l python bench.py --steps 60 --warmup 10 --no-cpu-baseline $BENCH_ARGS > gpurun_out/ab_${n}_$i.json
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],3))
PY
