#!/bin/bash
# images/s and ms per batch over batch sizes, both 16-bit modes, hipGraph replay: bash tools/batch_sweep.sh > gpurun_out/batch_sweep.txt
cd $GRAFT_REPO_ROOT
for prec in f16x3 bf16; do
  for b in 1 2 4 8 16 32 64; do
    if [ $b -ge 32 ]; then st=4; else st=8; fi
    v=$(python bench.py --precision $prec --batch $b --graph --steps $st --warmup 2 --no-cpu-baseline --no-sub-records 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('%8.2f img/s  %8.2f ms per batch  conv frac %.3f' % (r['value'], r['ms_per_step'], r['roofline']['frac']))")
    echo "$prec B=$b  $v"
  done
done
