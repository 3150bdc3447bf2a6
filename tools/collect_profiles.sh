#!/bin/bash
# Everything profiles/ holds for one round, on ONE GPU box (box-to-box spread is ±5 %):
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r3'
# Outputs land in gpurun_out/<tag>_*; copy the ones to be judged into profiles/.
# The profiler gets `python3 bench.py …` itself after `--` (no shell / env hop); PMC passes are kernel-trace only.
export NS3D_BENCH_NO_TRAFFIC=1 NS3D_BENCH_NO_CONFIG_B=1   # these runs are timed or profiled themselves: no nested rocprofv3 --pmc child runs (bench.py --no-traffic)
set -u
TAG=${1:-r3}
cd $GRAFT_REPO_ROOT
O=gpurun_out
b() { name=$1; shift; python3 bench.py "$@" > $O/${TAG}_bench_$name.json 2> $O/${TAG}_bench_$name.err; echo "$name: $(python3 -c "import json,sys; d=json.loads(open('$O/${TAG}_bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), 'Mcells*it/s', 'depth', d['config']['pt_depth'], 'frac', round(r['frac'],3), 'eff', round(r['effective_frac'],3), 'kernel_ms', round(r['kernel_ms'],4))" 2>&1)"; }
NS3D_BENCH_NO_TRAFFIC=0 NS3D_BENCH_NO_CONFIG_B=0 b strict                     # the headline line measures roofline.traffic live
NS3D_BENCH_NO_TRAFFIC=0 b fast --mode fast --no-cpu-baseline
NS3D_BENCH_NO_TRAFFIC=0 b f32_strict --dtype f32 --no-cpu-baseline
b f32_fast --dtype f32 --mode fast --no-cpu-baseline
b strict_depth2 --depth 2 --no-cpu-baseline
b strict_depth3 --depth 3 --no-cpu-baseline
b 1024cubed_strict --grid 1024 --steps 60 --warmup 6 --no-cpu-baseline
b 1024cubed_f32_strict --grid 1024 --dtype f32 --steps 60 --warmup 6 --no-cpu-baseline
b 255x255x153_strict --grid 255 --grid-nz 153 --no-cpu-baseline
b 255x255x153_fast --grid 255 --grid-nz 153 --mode fast --no-cpu-baseline
# the headline rides on power-of-two spacings (arith_build strictp); what other spacings get, STRICT (exact-division build) and FAST
b 510cubed_strict --grid 510 --no-cpu-baseline
b 510cubed_fast --grid 510 --mode fast --no-cpu-baseline
b 2ranks_one_gpu --gpus 2 --grid 256 --steps 40 --warmup 4
# kernel trace + stats of the default bench command
rm -rf /tmp/ns3d_kt; rocprofv3 --kernel-trace --stats -f csv -d /tmp/ns3d_kt -o kt -- python3 bench.py --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_rocprof.err
f=$(find /tmp/ns3d_kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats_strict_512.csv && head -6 $O/${TAG}_kernel_stats_strict_512.csv | cut -c1-220
rm -rf /tmp/ns3d_kt; rocprofv3 --kernel-trace --stats -f csv -d /tmp/ns3d_kt -o kt -- python3 bench.py --no-cpu-baseline --mode fast > $O/${TAG}_bench_under_rocprof_fast.json 2>> $O/${TAG}_rocprof.err
f=$(find /tmp/ns3d_kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats_fast_512.csv
rm -rf /tmp/ns3d_kt; rocprofv3 --kernel-trace --stats -f csv -d /tmp/ns3d_kt -o kt -- python3 bench.py --no-cpu-baseline --dtype f32 > $O/${TAG}_bench_under_rocprof_f32.json 2>> $O/${TAG}_rocprof.err
f=$(find /tmp/ns3d_kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats_f32_512.csv
python3 tools/collect_sq.py --out $O/${TAG}_pmc_sq_512.json --runs 4:2891,4:2300,3:1100,2:1392 --modes strict,fast 2>&1 | tail -4
python3 tools/collect_traffic.py --out $O/${TAG}_traffic_512.json --runs 4:2300,4:2391,4:2800,4:2891,3:1100,2:1392 --modes strict,fast 2>&1 | tail -4
python3 tools/collect_traffic.py --out $O/${TAG}_traffic_512_f32.json --runs 5:2400,4:2400,4:2200,3:100,2:1100 --modes strict --dtype f32 2>&1 | tail -3
python3 tools/kernel_rates.py > $O/${TAG}_kernel_rates_512.jsonl 2>/dev/null; grep -c kernel $O/${TAG}_kernel_rates_512.jsonl
python3 tools/run_config.py --script multi --nx 63 --nt 20 > $O/${TAG}_config_a_63x38x38.json 2>/dev/null; tail -c 300 $O/${TAG}_config_a_63x38x38.json; echo
python3 tools/run_config.py --script multi --nx 255 --nt 3 --marginal 40 --compare-fast --compare-direct > $O/${TAG}_config_b_multi_255x153x153.json 2>/dev/null; tail -c 300 $O/${TAG}_config_b_multi_255x153x153.json; echo
python3 tools/run_config.py --script gpu --nx 255 --nt 3 > $O/${TAG}_config_b_gpujl_255x153x153.json 2>/dev/null; tail -c 300 $O/${TAG}_config_b_gpujl_255x153x153.json; echo
python3 tools/cart_rates.py > $O/${TAG}_cart_rates.jsonl 2> $O/${TAG}_cart_rates.err; cat $O/${TAG}_cart_rates.jsonl
# pressure solve: the reference's PT loop against the direct solve (outside parity), the reference's own grids
python3 tools/direct_rates.py --nx 63 > $O/${TAG}_direct_vs_pt.jsonl 2> $O/${TAG}_direct.err; python3 tools/direct_rates.py --nx 255 >> $O/${TAG}_direct_vs_pt.jsonl 2>> $O/${TAG}_direct.err; cat $O/${TAG}_direct_vs_pt.jsonl
# what the z-slab schedule costs in compute (virtual ranks of one GPU against the global solve)
for nzl in 66 130 258; do python3 tools/ab/slab_overhead.py --nz-local $nzl --ranks $((512/(nzl-2))) --iters 96 --temporal 4; done > $O/${TAG}_slab_overhead.log 2>&1; grep slabs $O/${TAG}_slab_overhead.log
# the RCCL arm on one GPU (tests/fake_rccl): two ranks, weak + strong, self-verified
NS3D_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so FAKE_RCCL_ARENA_MB=64 python3 bench.py --gpus 2 --grid 256 --steps 40 --warmup 4 --transport rccl > $O/${TAG}_bench_2ranks_fake_rccl.json 2> $O/${TAG}_bench_2ranks_fake_rccl.err; cut -c1-400 $O/${TAG}_bench_2ranks_fake_rccl.json
# SQ counters of the two windowed once-per-step kernels (two --pmc passes each over tools/kernel_rates.py)
{ bash tools/ab/kernel_sq.sh k_advect_win2; bash tools/ab/kernel_sq.sh k_predict_fused; } > $O/${TAG}_window_kernels_sq.log 2>&1; cat $O/${TAG}_window_kernels_sq.log
# a further time step of the two reference cases (wall, >= 40 steps: 5-step differences scatter), and the kernels of the direct-solve step
{ python3 tools/ab/step_cost.py --steps 100; python3 tools/ab/step_cost.py --steps 40 --pressure pt --nx 63; python3 tools/ab/step_cost.py --steps 10 --pressure pt; } > $O/${TAG}_step_cost_wall.log 2>&1; cat $O/${TAG}_step_cost_wall.log
rm -rf /tmp/ns3d_kt; rocprofv3 --kernel-trace --stats -f csv -d /tmp/ns3d_kt -o kt -- python3 tools/ab/step_cost.py --steps 100 > /dev/null 2>> $O/${TAG}_rocprof.err
f=$(find /tmp/ns3d_kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${TAG}_step_cost_kernel_stats.csv
