#!/bin/bash
# per-kernel time under two launcher-option sets (rocprofv3 kernel stats of the same bench command):
#   bash tools/opt_stats.sh "<opts A>" "<opts B>" [bench args]      e.g. "k32=27" "k32=59 k32_stagger=3" --precision bf16 --batch 64
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/optstats; mkdir -p $O
A="$1"; B="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for o in "$A" "$B"; do
  i=$((i+1)); args=""; for kv in $o; do args="$args --debug-option $kv"; done
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/run$i -o s -- python3 $R/bench.py "$@" --steps 3 --warmup 1 \
    --no-cpu-baseline --no-sub-records --no-profile $args > $O/run$i.log 2>&1 < /dev/null
  f=$(find $O/run$i -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$i.csv
  rm -rf $O/run$i
done
cd $R
python tools/k32_stats_cmp.py $O/kernel_stats_1.csv $O/kernel_stats_2.csv | tee $O/cmp.txt
