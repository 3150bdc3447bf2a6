#!/bin/bash
# SQ counters of k_advect_win at 512^3 (one rocprofv3 --pmc pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/adv_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -f csv -d /tmp/adv_sq -o p -- python3 $GRAFT_REPO_ROOT/tools/kernel_rates.py --reps 3 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
acc = {}
for f in glob.glob('/tmp/adv_sq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_advect' not in row['Kernel_Name']: continue
        s, n = acc.get(row['Counter_Name'], (0.0, 0)); acc[row['Counter_Name']] = (s + float(row['Counter_Value']), n + 1)
m = {k: s / n for k, (s, n) in acc.items()}
wc = m.get('SQ_WAVE_CYCLES', 1.0)
print({k: round(v / wc, 4) for k, v in m.items()})
PY
