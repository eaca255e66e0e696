#!/bin/bash
# same-box A/B of two builds at B=1 (graph), B=2 and B=4: tools/b1_lib_ab.sh <variant .so>
V=$1
run() { # $1 = label, $2 = lib ('' = tree), $3 = batch, $4 flags, $5 steps
  FDSR_LIB=${2:-$PWD/fastdiffsr_amd/csrc/libfdsr_hip.so} python bench.py --batch $3 $4 --steps $5 --warmup 3 --no-cpu-baseline --no-sub-records --no-profile 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$3 [$1]', round(d['value'],2), 'img/s', round(d['ms_per_step'],2), 'ms')"
}
for rep in 1 2 3; do run tree "" 1 --graph 24; run variant $V 1 --graph 24; done
run tree "" 4 "" 8; run variant $V 4 "" 8; run tree "" 2 "" 12; run variant $V 2 "" 12
