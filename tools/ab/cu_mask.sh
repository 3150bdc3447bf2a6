#!/bin/bash
# profiles/r4_cu_mask_ab.log: both readings of the CU numbering, then bench.py --gpus 2 through the one-GPU fall-back with the knobs set
cd $GRAFT_REPO_ROOT
for l in 0 1; do NS3D_RESERVE_CUS_LAYOUT=$l timeout -k 10 500 python tools/ab/cu_mask_ab.py 2>&1 | grep -v amdgpu.ids; done
