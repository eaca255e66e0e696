#!/bin/bash
# same-box A/B of the tree's library against a variant .so on the bench regimes: tools/lib_ab.sh <variant .so> [regimes: bf16 f16x3 b1 train]
V=$1; shift; REG="${@:-bf16 f16x3}"
T=${FDSR_AB_BASE:-$PWD/fastdiffsr_amd/csrc/libfdsr_hip.so}   # FDSR_AB_BASE: another base than the tree's build
run() { # label lib args...
  local l=$1 lib=$2; shift 2
  FDSR_LIB=$lib python bench.py "$@" --no-cpu-baseline --no-sub-records --no-profile 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', round(d['value'],2), d['unit'], round(d['ms_per_step'],1), 'ms')"
}
for rep in 1 2; do
  for r in $REG; do
    case $r in
      bf16)  A="--precision bf16 --batch 64 --graph --steps 3 --warmup 1";;
      f16x3) A="--steps 6 --warmup 2";;
      b1)    A="--batch 1 --graph --steps 24 --warmup 3";;
      train) A="--train --precision f16x3 --steps 4 --warmup 1";;
    esac
    run "$r tree   " $T $A; run "$r variant" $V $A
  done
done
