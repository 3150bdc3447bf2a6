#!/bin/bash
# SQ counters of k_gemm_f64_lds inside the direct-solve time step (tools/ab/step_cost.py): two rocprofv3 --pmc passes, kernel-trace only.
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gsq1 /tmp/gsq2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -f csv -d /tmp/gsq1 -o p -- python3 $GRAFT_REPO_ROOT/tools/ab/step_cost.py --steps 10 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d /tmp/gsq2 -o p -- python3 $GRAFT_REPO_ROOT/tools/ab/step_cost.py --steps 10 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for d in ('/tmp/gsq1', '/tmp/gsq2'):
    acc = {}
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'k_gemm_f64_lds' not in row['Kernel_Name']: continue
            s, n = acc.get(row['Counter_Name'], (0.0, 0)); acc[row['Counter_Name']] = (s + float(row['Counter_Value']), n + 1)
    m = {c: s / n for c, (s, n) in acc.items()}
    if 'SQ_WAVE_CYCLES' in m:
        wc = m['SQ_WAVE_CYCLES']
        print("k_gemm_f64_lds share of wave cycles:", {c: round(v / wc, 4) for c, v in m.items()}, "wave cycles", wc)
    else:
        print("k_gemm_f64_lds per launch:", {c: round(v, 1) for c, v in m.items()})
PY
