#!/bin/bash
# A/B or diagnostic variant of the library that differs in ONE kernel file: tools/build_obj_variant.sh <tag> <file.hip> [extra hipcc flags...]
# -> fastdiffsr_amd/csrc/ab/libfdsr_hip_<tag>.so (run with FDSR_LIB=<that path>); the other objects come from the tree's build.
set -e
TAG=$1; F=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd); C=$R/fastdiffsr_amd/csrc; O=$C/ab; mkdir -p $O
B=$(basename $F .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-result -O3 -fno-slp-vectorize "$@" -c $C/$B.hip -o $O/${B}_$TAG.o
OBJS=""
for f in fdsr_kernels fdsr_conv_h fdsr_conv_up2 fdsr_conv_wino fdsr_train; do
  if [ $f = $B ]; then OBJS="$OBJS $O/${B}_$TAG.o"; else OBJS="$OBJS $C/${f}_hip.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $C/fdsr_engine_cpp.o $C/fdsr_train_cpp.o -o $O/libfdsr_hip_$TAG.so
rm -f $O/${B}_$TAG.o
echo built $O/libfdsr_hip_$TAG.so
