#!/bin/bash
# Same-box A/B of two library builds: tools/lib_ab.sh <variant.so> "<opts A (tree library)>" "<opts B (variant library)>" [bench args]
# (variant libraries: tools/build_obj_variant.sh / k32_variant.sh -> fastdiffsr_amd/csrc/ab/; loaded through FDSR_LIB).  A B A B, images/s.
V="$1"; A="$2"; B="$3"; shift 3
one() {  # lib ("" = the tree's), opts, bench args...
  lib="$1"; o="$2"; shift 2
  args=""; for kv in $o; do args="$args --debug-option $kv"; done
  if [ -n "$lib" ]; then export FDSR_LIB="$lib"; else unset FDSR_LIB; fi
  v=$(python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sub-records $args "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('%.2f img/s  conv frac %.4f' % (r['value'], r['roofline']['frac']))")
  echo "[${lib:+variant }$o] $v"
  unset FDSR_LIB
}
for rep in 1 2; do
  one "" "$A" "$@"
  one "$V" "$B" "$@"
done
