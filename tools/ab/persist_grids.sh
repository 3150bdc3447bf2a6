#!/bin/bash
for grid in "66 50 50" "66 66 66" "130 34 34" "130 50 50" "30 18 18"; do
  set -- $grid
  for nchk in 37 370; do
    for pm in 0 1; do
    timeout -k 10 100 python tools/ab/small_grid.py --nx $1 --ny $2 --nz $3 --cases 1:0 --persist $pm --nchk $nchk 2>&1 | grep "^persist" | sed "s/^/$1x$2x$3 nchk $nchk: /"
    done
  done
done
