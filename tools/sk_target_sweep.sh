#!/bin/bash
# split-K depth at small batches with the consumer-side GroupNorm in (round 6): sk_target (split the K loop until a launch has this many
# workgroups; default 256) x th_min_wgs, f16x3, hipGraph replay, A B ... on one box:   bash tools/sk_target_sweep.sh > gpurun_out/sk_target_sweep.txt
cd $GRAFT_REPO_ROOT
for b in 1 2 4; do
  for rep in 1 2; do
    for o in "sk_target=256" "sk_target=128" "sk_target=192" "sk_target=320" "sk_target=128 th_min_wgs=128" "splitk=0"; do
      args=""; for kv in $o; do args="$args --debug-option $kv"; done
      v=$(python bench.py --precision f16x3 --batch $b --graph --steps 10 --warmup 3 --no-cpu-baseline --no-sub-records --no-profile $args 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('%8.2f img/s  %8.2f ms per batch' % (r['value'], r['ms_per_step']))")
      echo "f16x3 B=$b [$o]  $v"
    done
  done
done
