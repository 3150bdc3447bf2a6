#!/bin/bash
# SQ counters of one of the once-per-step kernels at 512^3:  tools/ab/kernel_sq.sh <kernel name substring>  (two rocprofv3 --pmc
# passes, kernel-trace only).  Prints time shares relative to wave cycles and dynamic instructions per wave.
K=${1:-k_advect}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ksq1 /tmp/ksq2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -f csv -d /tmp/ksq1 -o p -- python3 $GRAFT_REPO_ROOT/tools/kernel_rates.py --reps 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d /tmp/ksq2 -o p -- python3 $GRAFT_REPO_ROOT/tools/kernel_rates.py --reps 3 > /dev/null 2>&1
KSUB="$K" python3 - <<'PY'
import csv, glob, os
k = os.environ["KSUB"]
for d in ('/tmp/ksq1', '/tmp/ksq2'):
    acc = {}
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if k not in row['Kernel_Name']: continue
            s, n = acc.get(row['Counter_Name'], (0.0, 0)); acc[row['Counter_Name']] = (s + float(row['Counter_Value']), n + 1)
    m = {c: s / n for c, (s, n) in acc.items()}
    if 'SQ_WAVE_CYCLES' in m:
        wc = m['SQ_WAVE_CYCLES']
        print(k, "share of wave cycles:", {c: round(v / wc, 4) for c, v in m.items()})
    elif 'SQ_WAVES' in m:
        w = m['SQ_WAVES']
        print(k, "per wave:", {c: round(v / w, 1) for c, v in m.items()}, "waves", w)
PY
