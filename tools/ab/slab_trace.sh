#!/bin/bash
# kernel trace (start/end timestamps) of the z-slab schedule on virtual ranks of one GPU: where a pass spends its time
#   gpurun -- 'bash tools/ab/slab_trace.sh 66 8'
cd $GRAFT_REPO_ROOT
NZL=${1:-66}; P=${2:-8}
for ser in 1 0; do
rm -rf /tmp/ns3d_st
NS3D_SLAB_SERIAL_SEAMS=$ser rocprofv3 --kernel-trace -f csv -d /tmp/ns3d_st -o st -- python3 tools/ab/slab_overhead.py --nz-local $NZL --ranks $P --iters 24 --temporal 4 > gpurun_out/slab_trace_${NZL}_ser$ser.log 2>&1
f=$(find /tmp/ns3d_st -name '*kernel_trace.csv' | head -1)
python3 - "$f" $ser $NZL <<'PY'
import csv, sys
f, ser, nzl = sys.argv[1:4]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 400 kernels = the final timed repetition of the slab solve
tail = rows[-400:]
t0 = int(tail[0]["Start_Timestamp"])
out = open("gpurun_out/slab_trace_%s_ser%s.txt" % (nzl, ser), "w")
prev_end = t0
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"][:60]
    out.write("%9.1f us  dur %7.1f  gap %6.1f  q%s  %s grid %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), name, r.get("Grid_Size", "?")))
    prev_end = max(prev_end, e)
out.close()
PY
done
head -60 gpurun_out/slab_trace_${NZL}_ser0.txt
