#!/bin/bash
# kernel stats of the training step under a debug option: tools/train_stats.sh <tag> [NAME=VALUE]  -> gpurun_out/train_stats_<tag>.csv
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; TAG=$1; OPT=""; [ -n "$2" ] && OPT="--debug-option $2"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ts_$TAG -o t -- python3 $R/bench.py --train --precision f16x3 --steps 2 --warmup 1 --no-cpu-baseline $OPT > $O/ts_$TAG.log 2>&1 < /dev/null
f=$(find $O/ts_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/train_stats_$TAG.csv
rm -rf $O/ts_$TAG
head -16 $O/train_stats_$TAG.csv | cut -d, -f1-4 | cut -c1-170
