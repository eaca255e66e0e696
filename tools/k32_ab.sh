#!/bin/bash
# same-box A/B of the 16x16x32 conv form: bash tools/k32_ab.sh [reps]   (f16x3 B=16 eager, bf16 B=64 hipGraph)
O=gpurun_out/k32; mkdir -p $O
run() {  # precision batch extra option steps
  python bench.py --precision $1 --batch $2 $3 --steps $5 --warmup 1 --no-cpu-baseline --no-sub-records --no-profile --debug-option $4 2>>$O/err.txt | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 B=$2 $3 [$4]', round(d['value'],2), 'img/s', round(d['ms_per_step'],1), 'ms')" | tee -a $O/ab_summary.txt
}
for rep in $(seq 1 ${1:-2}); do
  for o in k32=1 k32=0; do run f16x3 16 "" $o 5; done
  for o in k32=3 k32=0; do run bf16 64 --graph $o 3; done
done
