#!/bin/bash
# Round 4: which depth / tile shape wins on mid-size grids (the reference's 255×153×153 and neighbours), k_pt_sweepD included.
set -e
cd $GRAFT_REPO_ROOT
VN="3:2800,3:2300,3:1100,3:100,3:600,3:3100,3:3500,3:3800,3:3200,4:2800,4:3800,4:3100"
for g in "255 153 153" "192 192 192" "256 256 128" "127 77 77" "320 192 192"; do
set -- $g
timeout -k 10 300 python tools/sweep_variants.py --n $1 --ny $2 --nz $3 --rounds 3 --iters 60 --variants "" --modes strict,fast --variantsn "$VN" --variants2 "0,700,800,1300" 2>&1 | grep -v amdgpu.ids
done
