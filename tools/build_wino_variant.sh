#!/bin/bash
# A/B variant of the library that differs only in fdsr_conv_wino.hip: tools/build_wino_variant.sh <tag> [extra hipcc flags...]
set -e
TAG=$1; shift
exec $(dirname $0)/build_obj_variant.sh $TAG fdsr_conv_wino.hip "$@"
