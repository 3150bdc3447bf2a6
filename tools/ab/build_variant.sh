#!/bin/bash
# A/B library with extra -D flags in the power-of-two STRICT kernel unit only (what bench.py's 512^3 default runs):
#   tools/ab/build_variant.sh NAME "-DNS3D_COOP_EXP=1"   ->  tools/ab/libns3d_NAME.so   (use with NS3D_LIB=… or tools/ab/ab.sh)
set -e
cd "$(dirname "$0")/../.."
B=navierstokes3d_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-fast-math -Wall -Wno-unused-function -DNS3D_MODE_STRICT -DNS3D_POW2_RECIP -ffp-contract=off $2 \
  -c navierstokes3d_amd/csrc/ns3d_kernels.hip -o /tmp/ns3d_strictp_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libns3d_$1.so $B/ns3d_kernels_strict.o $B/ns3d_kernels_strictx.o /tmp/ns3d_strictp_$1.o \
  $B/ns3d_kernels_fast.o $B/ns3d_direct.o $B/ns3d_api.o $B/ns3d_mgpu.o -ldl
ls -la tools/ab/libns3d_$1.so
