#!/bin/bash
# A/B or diagnostic variant of the library that differs in ONE kernel file: tools/build_obj_variant.sh <tag> <file.hip> [extra hipcc flags...]
# -> fastdiffsr_amd/csrc/ab/libfdsr_hip_<tag>.so (run with FDSR_LIB=<that path>); every other object comes from the tree's build
# (python -m fastdiffsr_amd.build first).  The object list is whatever the tree has (csrc/*_hip.o, *_cpp.o), and the link refuses
# undefined symbols, so a kernel file added later cannot silently drop out of the variant.
set -e
TAG=$1; F=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd); C=$R/fastdiffsr_amd/csrc; O=$C/ab; mkdir -p $O
B=$(basename $F .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-result -O3 -fno-slp-vectorize "$@" -c $C/$B.hip -o $O/${B}_$TAG.o
OBJS=""
for o in $C/*_hip.o $C/*_cpp.o; do
  if [ "$(basename $o)" = "${B}_hip.o" ]; then OBJS="$OBJS $O/${B}_$TAG.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs $OBJS -o $O/libfdsr_hip_$TAG.so
rm -f $O/${B}_$TAG.o
echo built $O/libfdsr_hip_$TAG.so
