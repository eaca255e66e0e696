#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b1; mkdir -p $O
cd $R
for o in "sk_target=256" "sk_target=128" "sk_target=192" "sk_target=384" "sk_target=512" "sk_target=256 th_min_wgs=128" "sk_target=256 th_min_wgs=512"; do
  args=""; for kv in $o; do args="$args --debug-option $kv"; done
  v=$(python bench.py --batch 1 --graph --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --no-profile $args 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('%.2f img/s  %.2f ms/image' % (r['value'], r['ms_per_step']))")
  echo "[$o] $v"
done 2>&1 | tee $O/sweep.txt
