#!/bin/bash
# k_pt_persist A/B on the small grids: shapes, XCD mapping, check-block size
for grid in "63 38 38" "40 24 24" "100 60 60"; do
  set -- $grid
  for nchk in 37 370; do
    NS3D_PT_PERSIST=0 timeout -k 10 100 python tools/ab/small_grid.py --nx $1 --ny $2 --nz $3 --cases 1:0 --persist 0 --nchk $nchk 2>&1 | grep "^persist" | sed "s/^/$1x$2x$3 nchk $nchk launches: /"
    for shape in 22 42 44; do
      NS3D_PERSIST_SHAPE=$shape timeout -k 10 100 python tools/ab/small_grid.py --nx $1 --ny $2 --nz $3 --cases 1:0 --persist 1 --nchk $nchk 2>&1 | grep "^persist" | sed "s/^/$1x$2x$3 nchk $nchk shape $shape: /"
    done
    NS3D_PERSIST_XCDMAP=1 timeout -k 10 100 python tools/ab/small_grid.py --nx $1 --ny $2 --nz $3 --cases 1:0 --persist 1 --nchk $nchk 2>&1 | grep "^persist" | sed "s/^/$1x$2x$3 nchk $nchk auto shape, XCD map: /"
  done
done
