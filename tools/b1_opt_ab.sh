#!/bin/bash
# B=1 + hipGraph images/s under debug-option sets: tools/b1_opt_ab.sh "" "th_min_wgs=128" ...
O=gpurun_out/k32; mkdir -p $O
B=${B1_BATCH:-1}
for rep in 1 2; do
  for set in "$@"; do
    args=""; for o in ${set//,/ }; do args="$args --debug-option $o"; done
    python bench.py --batch $B --graph --steps 24 --warmup 3 --no-cpu-baseline --no-sub-records --no-profile $args 2>>$O/err.txt | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f16x3 B=$B graph [$set]', round(d['value'],2), 'img/s', round(d['ms_per_step'],2), 'ms')" | tee -a $O/b1_opt_ab_summary.txt
  done
done
