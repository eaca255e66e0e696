#!/bin/bash
# A/B variant of the 16x16x32 conv object only: tools/k32_variant.sh <tag> [-D...]  ->  fastdiffsr_amd/csrc/ab/libfdsr_hip_<tag>.so
set -e
TAG=$1; shift
exec $(dirname $0)/build_obj_variant.sh $TAG fdsr_conv_k32.hip "$@"
