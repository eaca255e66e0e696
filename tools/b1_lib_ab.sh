#!/bin/bash
# Same-box A/B of library variants (tools/build_obj_variant.sh) at small batches: tools/b1_lib_ab.sh <tag> [<tag> ...]   ("tree" = the tree's own build)
cd $GRAFT_REPO_ROOT
for b in 1 2 4; do
  for rep in 1 2; do
    for t in "$@"; do
      if [ "$t" = tree ]; then unset FDSR_LIB; else export FDSR_LIB=$GRAFT_REPO_ROOT/fastdiffsr_amd/csrc/ab/libfdsr_hip_$t.so; fi
      v=$(python bench.py --precision f16x3 --batch $b --graph --steps 10 --warmup 3 --no-cpu-baseline --no-sub-records --no-profile 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('%8.2f img/s  %8.2f ms per batch  parity %s' % (r['value'], r['ms_per_step'], (r.get('parity_check') or {}).get('max_abs_diff_image')))")
      echo "f16x3 B=$b [$t]  $v"
    done
  done
done
