#!/bin/bash
# bf16 B=64 (hipGraph) and B=16 images/s under debug-option settings: tools/bf16_ab.sh "bf16_nw16=1" "bf16_nw16=0" ...
run() {
  python bench.py --precision bf16 --batch $1 $2 --steps $4 --warmup 1 --no-cpu-baseline --no-sub-records --no-profile --debug-option $3 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 B=$1 [$3]', round(d['value'],2), 'img/s', round(d['ms_per_step'],1), 'ms')"
}
for rep in 1 2; do for o in "$@"; do run 64 --graph "$o" 3; done; done
for o in "$@"; do run 16 "" "$o" 4; done
