set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out; TAG=r3
python3 bench.py --dtype f32 --no-cpu-baseline > $O/${TAG}_bench_f32_strict.json 2> $O/f32.err
export NS3D_BENCH_NO_TRAFFIC=1
python3 bench.py --grid 1024 --dtype f32 --steps 60 --warmup 6 --no-cpu-baseline > $O/${TAG}_bench_1024cubed_f32_strict.json 2>> $O/f32.err
python3 bench.py --dtype f32 --mode fast --no-cpu-baseline > $O/${TAG}_bench_f32_fast.json 2>> $O/f32.err
rm -rf /tmp/ns3d_kt; rocprofv3 --kernel-trace --stats -f csv -d /tmp/ns3d_kt -o kt -- python3 bench.py --no-cpu-baseline --dtype f32 > $O/${TAG}_bench_under_rocprof_f32.json 2>> $O/f32.err
f=$(find /tmp/ns3d_kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats_f32_512.csv && head -3 $O/${TAG}_kernel_stats_f32_512.csv | cut -c1-200
python3 tools/collect_traffic.py --out $O/${TAG}_traffic_512_f32.json --runs 5:2400,4:2400,4:2200,3:100,2:1100 --modes strict --dtype f32 2>&1 | tail -5
for f in $O/${TAG}_bench_f32_strict.json $O/${TAG}_bench_1024cubed_f32_strict.json $O/${TAG}_bench_f32_fast.json; do python3 -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', round(d['value']), 'depth', d['config']['pt_depth'], 'ms/pass', round(r['kernel_ms'],4), 'frac', round(r['frac'],3), 'traffic', r['traffic'], d['config']['verified'])"; done
