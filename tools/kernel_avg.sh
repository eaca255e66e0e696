#!/bin/bash
# per-kernel average time of chosen kernels under several library builds / option sets (rocprofv3 kernel stats of tools/lib_probe.py):
#   bash tools/kernel_avg.sh <prec> <batch> <kernel-name regex> <label>=<lib.so or ''>[:opt=val,opt=val] ...
# e.g. bash tools/kernel_avg.sh bf16 64 'conv_strip|conv_k32_kernel<8, 2' tree= pin0=fastdiffsr_amd/csrc/ab/libfdsr_hip_pin0.so wgs1024=:strip_min_wgs=1024
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kavg; mkdir -p $O
PREC=$1; B=$2; RE="$3"; shift 3
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  label=${spec%%=*}; rest=${spec#*=}; lib=${rest%%:*}; opts=""
  if [[ "$rest" == *:* ]]; then opts=${rest#*:}; fi
  if [ -n "$lib" ]; then export FDSR_LIB=$R/$lib; else unset FDSR_LIB; fi
  export PROBE_OPTS="$opts" REPS=1
  rm -rf $O/run_$label
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/run_$label -o s -- python3 $R/tools/lib_probe.py $PREC $B > $O/$label.log 2>&1 < /dev/null
  f=$(find $O/run_$label -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then
    cp $f $O/stats_$label.csv
    python3 - "$f" "$RE" "$label" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows:
    if re.search(sys.argv[2], r['Name']):
        print('%-10s %-70s calls %5s avg %9.1f us  share %5.1f%%' % (sys.argv[3], r['Name'][:70].replace('void fdsr::', ''), r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print('%-10s total kernel time %.1f ms;  %s' % (sys.argv[3], tot / 1e6, open(sys.argv[1].rsplit('/run_', 1)[0] + '/' + sys.argv[3] + '.log').read().strip().splitlines()[-1][:100]))
PY
  else
    echo "$label: no stats"; tail -5 $O/$label.log
  fi
  rm -rf $O/run_$label
done
