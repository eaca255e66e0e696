#!/bin/bash
# A/B variant of the 16x16x32 conv object only, linked against the tree's other objects (run the tree build first):
#   tools/k32_variant.sh <tag> [-D...]  ->  fastdiffsr_amd/csrc/ab/libfdsr_hip_<tag>.so   (FDSR_LIB=<that path>)
set -e
TAG=$1; shift
R=$(cd $(dirname $0)/.. && pwd); C=$R/fastdiffsr_amd/csrc; O=$C/ab; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-result -O3 -fno-slp-vectorize "$@" -c $C/fdsr_conv_k32.hip -o $O/k32_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/fdsr_kernels_hip.o $C/fdsr_conv_h_hip.o $C/fdsr_conv_up2_hip.o $C/fdsr_conv_wino_hip.o \
  $O/k32_$TAG.o $C/fdsr_train_hip.o $C/fdsr_engine_cpp.o $C/fdsr_train_cpp.o -o $O/libfdsr_hip_$TAG.so
rm -f $O/k32_$TAG.o
echo built $O/libfdsr_hip_$TAG.so
