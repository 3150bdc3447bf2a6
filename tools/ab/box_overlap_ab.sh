#!/bin/bash
# Shell-first box pass (exchange chain on the communication stream, core sweep meanwhile) against sweep-then-exchange, virtual ranks on ONE GPU:
# this prices the extra launches; what the overlap buys needs one GPU per rank.   gpurun -- 'bash tools/ab/box_overlap_ab.sh'
cd $GRAFT_REPO_ROOT
for cfg in "130 2,2,2" "258 2,2,2" "258 2,2,1" "258 2,1,1" "386 2,2,1" "512 2,1,1"; do
set -- $cfg
for ov in 1 0; do
echo "local $1 dims $2 NS3D_BOX_OVERLAP=$ov: $(NS3D_BOX_OVERLAP=$ov timeout -k 10 300 python tools/cart_rates.py --local $1 --dims "$2" --iters 48 --no-reference 2>/dev/null | grep 'deep ghosts')"
done
done
