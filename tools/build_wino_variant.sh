#!/bin/bash
# A/B variant of the library that differs only in fdsr_conv_wino.hip: tools/build_wino_variant.sh <tag> [extra hipcc flags...]
# -> fastdiffsr_amd/csrc/ab/libfdsr_hip_<tag>.so (run with FDSR_LIB=<that path>); the other objects come from the tree's build.
set -e
TAG=$1; shift
R=$(cd $(dirname $0)/.. && pwd); C=$R/fastdiffsr_amd/csrc; O=$C/ab; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-result -O3 -fno-slp-vectorize "$@" -c $C/fdsr_conv_wino.hip -o $O/wino_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/fdsr_kernels_hip.o $C/fdsr_conv_h_hip.o $C/fdsr_conv_up2_hip.o $O/wino_$TAG.o $C/fdsr_train_hip.o \
  $C/fdsr_engine_cpp.o $C/fdsr_train_cpp.o -o $O/libfdsr_hip_$TAG.so
rm -f $O/wino_$TAG.o
echo built $O/libfdsr_hip_$TAG.so
