#!/bin/bash
# the consumer-side GroupNorm (gn_consumer, round 6) over small batches, both 16-bit modes, hipGraph replay, A B A B on one box:
#   bash tools/gn_consumer_sweep.sh > gpurun_out/gn_consumer_sweep.txt
cd $GRAFT_REPO_ROOT
for prec in f16x3 bf16; do
  for b in 1 2 4 8; do
    for rep in 1 2; do
      for g in 0 1; do
        v=$(python bench.py --precision $prec --batch $b --graph --steps 10 --warmup 3 --no-cpu-baseline --no-sub-records --no-profile --debug-option gn_consumer=$g 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('%8.2f img/s  %8.2f ms per batch' % (r['value'], r['ms_per_step']))")
        echo "$prec B=$b gn_consumer=$g  $v"
      done
    done
  done
done
