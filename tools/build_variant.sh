#!/bin/bash
# Build an A/B variant of the library: tools/build_variant.sh <tag> [extra hipcc flags for the conv kernels...]
# -> fastdiffsr_amd/csrc/ab/libfdsr_hip_<tag>.so  (run with FDSR_LIB=<that path>)
set -e
TAG=$1; shift
R=$(cd $(dirname $0)/.. && pwd); C=$R/fastdiffsr_amd/csrc; O=$C/ab; mkdir -p $O/$TAG
COMMON="--offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-result"
/opt/rocm/bin/hipcc $COMMON -O3 -munsafe-fp-atomics "$@" -c $C/fdsr_kernels.hip -o $O/$TAG/k.o &
/opt/rocm/bin/hipcc $COMMON -O3 "$@" -c $C/fdsr_train.hip -o $O/$TAG/t.o &
/opt/rocm/bin/hipcc $COMMON -O2 "$@" -c $C/fdsr_train.cpp -o $O/$TAG/tc.o &
/opt/rocm/bin/hipcc $COMMON -O3 -fno-slp-vectorize "$@" -c $C/fdsr_conv_h.hip -o $O/$TAG/h.o &
/opt/rocm/bin/hipcc $COMMON -O3 -fno-slp-vectorize "$@" -c $C/fdsr_conv_up2.hip -o $O/$TAG/u.o &
/opt/rocm/bin/hipcc $COMMON -O3 -fno-slp-vectorize "$@" -c $C/fdsr_conv_wino.hip -o $O/$TAG/w.o &
/opt/rocm/bin/hipcc $COMMON -O3 -fno-slp-vectorize "$@" -c $C/fdsr_conv_k32.hip -o $O/$TAG/k32.o &
/opt/rocm/bin/hipcc $COMMON -O3 "$@" -c $C/fdsr_val.hip -o $O/$TAG/v.o &
/opt/rocm/bin/hipcc $COMMON -O2 -DFDSR_SRC_SHA256=\"variant-$TAG\" "$@" -c $C/fdsr_engine.cpp -o $O/$TAG/e.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs $O/$TAG/v.o $O/$TAG/k.o $O/$TAG/h.o $O/$TAG/u.o $O/$TAG/w.o $O/$TAG/k32.o $O/$TAG/e.o $O/$TAG/t.o $O/$TAG/tc.o -o $O/libfdsr_hip_$TAG.so
rm -rf $O/$TAG
echo built $O/libfdsr_hip_$TAG.so
