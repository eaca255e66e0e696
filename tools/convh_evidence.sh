#!/bin/bash
# Phase timelines (s_memtime stamps) of the direct conv kernel and the no-in-loop-staging bound, from the variant builds of
# tools/build_obj_variant.sh -> gpurun_out/convh_phase_timeline.txt
A=$PWD/fastdiffsr_amd/csrc/ab; O=$PWD/gpurun_out/convh_phase_timeline.txt; mkdir -p gpurun_out; : > $O
for v in "st_bf16_64 bf16 64" "st_bf16_256 bf16 64" "st_f16_64 f16x3 16" "st_f16_128 f16x3 16"; do
  set -- $v
  echo "== $1 (python tools/convh_stamps.py $2 $3)" >> $O
  FDSR_LIB=$A/libfdsr_hip_$1.so python tools/convh_stamps.py $2 $3 >> $O 2>/dev/null
done
echo "== no in-loop staging (perf-only, wrong results) against the tree's build, same box" >> $O
FDSR_AB_BASE=$PWD/fastdiffsr_amd/csrc/libfdsr_hip.so tools/lib_ab.sh $A/libfdsr_hip_ko_bf16.so bf16 >> $O 2>&1
for l in $PWD/fastdiffsr_amd/csrc/libfdsr_hip.so $A/libfdsr_hip_ko_f16.so $PWD/fastdiffsr_amd/csrc/libfdsr_hip.so $A/libfdsr_hip_ko_f16.so; do
  FDSR_LIB=$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sub-records --no-profile --debug-option sat_guard=0 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f16x3 B=16 $(basename $l)', round(d['value'],2), 'img/s')" >> $O
done
cat $O
