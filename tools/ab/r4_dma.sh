#!/bin/bash
# Round 4: k_pt_sweepD (LDS-DMA staging of P⁰) and wave-uniform level skipping against k_pt_sweepN, one box, one process per grid.
#   gpurun -- 'bash tools/ab/r4_dma.sh'
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pt.py -x -q -m gpu -k "sweepn or timed_kernel" > gpurun_out/r4_dma_tests.log 2>&1 || { tail -40 gpurun_out/r4_dma_tests.log; exit 1; }
tail -3 gpurun_out/r4_dma_tests.log
V4="4:2800,4:2891,4:3100,4:3500,4:3200,4:3800,4:3807"   # (33xx, 34xx, 36xx, 37xx, 39xx were measured with this script and then removed: profiles/r4_levelskip_dma_ab.log)
timeout -k 10 600 python tools/sweep_variants.py --n 512 --rounds 3 --iters 40 --variants "" --modes strict,fast --variantsn "$V4,3:2800,3:3800" > gpurun_out/r4_dma_512.log 2>&1
cat gpurun_out/r4_dma_512.log
timeout -k 10 600 python tools/sweep_variants.py --n 512 --rounds 3 --iters 40 --variants "" --modes strict --dtype f32 --variantsn "4:2400,5:2400,4:3800,4:3200" > gpurun_out/r4_dma_512_f32.log 2>&1
cat gpurun_out/r4_dma_512_f32.log
if [ -n "$SQ_RUNS" ]; then
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 python3 $GRAFT_REPO_ROOT/tools/collect_sq.py --out $GRAFT_REPO_ROOT/gpurun_out/r4_sq_insts.json --set insts --modes strict --runs "$SQ_RUNS" 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r4_sq_insts.log
timeout -k 10 600 python3 $GRAFT_REPO_ROOT/tools/collect_sq.py --out $GRAFT_REPO_ROOT/gpurun_out/r4_sq_time.json --set time --modes strict --runs "$SQ_RUNS" 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r4_sq_time.log
fi
