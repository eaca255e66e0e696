#!/bin/bash
# same-box A/B of k32 variant libraries (tools/k32_variant.sh): tools/k32_lib_ab.sh "<regimes: f16x3 bf16>" tagA tagB ...   ('tree' = the tree's build)
REG=$1; shift
O=gpurun_out/k32; mkdir -p $O
run() {  # tag precision batch extra steps
  local lib=$PWD/fastdiffsr_amd/csrc/ab/libfdsr_hip_$1.so; [ $1 = tree ] && lib=$PWD/fastdiffsr_amd/csrc/libfdsr_hip.so
  FDSR_LIB=$lib python bench.py --precision $2 --batch $3 $4 --steps $5 --warmup 1 --no-cpu-baseline --no-sub-records --no-profile ${K32_OPT:+--debug-option $K32_OPT} 2>>$O/err.txt | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 B=$3 $4 [$1]', round(d['value'],2), 'img/s', round(d['ms_per_step'],1), 'ms')" | tee -a $O/lib_ab_summary.txt
}
for rep in 1 2; do
  for t in "$@"; do
    for r in $REG; do
      case $r in
        f16x3) run $t f16x3 16 "" 5;;
        bf16)  run $t bf16 64 --graph 3;;
      esac
    done
  done
done
