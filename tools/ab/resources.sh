#!/bin/bash
# Register / LDS / spill figures of the kernels in ONE arithmetic build of ns3d_kernels.hip (default: the power-of-two STRICT unit),
# straight from the compiler (-Rpass-analysis=kernel-resource-usage), filtered by a kernel-name substring:
#   tools/ab/resources.sh k_pt_sweepD ["-DNS3D_…"] [strictp|strictx|strict|fast]
set -e
cd "$(dirname "$0")/../.."
K=${1:-k_pt_sweepN}
case "${3:-strictp}" in
  strictp) F="-DNS3D_MODE_STRICT -DNS3D_POW2_RECIP -ffp-contract=off" ;;
  strictx) F="-DNS3D_MODE_STRICT -DNS3D_EXACT_RECIP -ffp-contract=off" ;;
  strict) F="-DNS3D_MODE_STRICT -ffp-contract=off" ;;
  fast) F="-DNS3D_MODE_FAST -ffp-contract=fast" ;;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-fast-math -Wall -Wno-unused-function $F $2 \
  -Rpass-analysis=kernel-resource-usage -c navierstokes3d_amd/csrc/ns3d_kernels.hip -o /tmp/nsb/res_probe.o 2> /tmp/nsb/res_probe.txt || { tail -30 /tmp/nsb/res_probe.txt; exit 1; }
python3 - "$K" <<'PY'
import re, sys, subprocess
k = sys.argv[1]
txt = open('/tmp/nsb/res_probe.txt').read()
blocks = re.split(r'(?=remark: [^\n]*Function Name: )', txt)
for b in blocks:
    m = re.search(r'Function Name: (\S+)', b)
    if not m: continue
    name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    if k not in name: continue
    g = lambda key: (re.search(key + r': (\S+)', b) or [None, '?'])[1]
    print("%-70s VGPR %s AGPR %s spill %s SGPR %s sspill %s scratch %s LDS %s occ %s" % (
        name[name.find('k_'):][:70], g('VGPRs'), g('AGPRs'), g('VGPRs Spill'), g('TotalSGPRs'), g('SGPRs Spill'),
        g(r'ScratchSize \[bytes/lane\]'), g(r'LDS Size \[bytes/block\]'), g(r'Occupancy \[waves/SIMD\]')))
PY
