#!/bin/bash
# same-box A/B of debug-option sets on the headline workload: tools/opt_ab.sh "wino=0,th_min_wgs=128" "" ...   ("" = defaults)
O=gpurun_out/k32; mkdir -p $O
for rep in 1 2; do
  for set in "$@"; do
    args=""; for o in ${set//,/ }; do args="$args --debug-option $o"; done
    python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-sub-records --no-profile $args 2>>$O/err.txt | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f16x3 B=16 [$set]', round(d['value'],2), 'img/s', round(d['ms_per_step'],1), 'ms')" | tee -a $O/opt_ab_summary.txt
  done
done
