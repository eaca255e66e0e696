#!/bin/bash
# per-kernel time with the 16x16x32 form on and off (rocprofv3 kernel stats of the same bench command): bash tools/k32_stats.sh [precision batch extra]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/k32; mkdir -p $O
P=${1:-f16x3}; B=${2:-16}; X=$3
cd /tmp && export TMPDIR=/tmp
for o in 1 0; do
  [ $P = bf16 ] && [ $o = 1 ] && o=3
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${P}_$o -o s -- python3 $R/bench.py --precision $P --batch $B $X --steps 3 --warmup 1 \
    --no-cpu-baseline --no-sub-records --no-profile --debug-option k32=$o > $O/stats_${P}_$o.log 2>&1 < /dev/null
  f=$(find $O/stats_${P}_$o -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_${P}_k32_$o.csv
  rm -rf $O/stats_${P}_$o
done
cd $R
python tools/k32_stats_cmp.py $O/kernel_stats_${P}_k32_*.csv | tee $O/stats_cmp_$P.txt
