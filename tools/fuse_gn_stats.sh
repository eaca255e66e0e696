#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fusegn; mkdir -p $O
cd $R
bash tools/kernel_avg.sh f16x3 1 'splitk_reduce|gn_finalize' off=:fuse_gn=0 on=:fuse_gn=1 2>&1 | tee $O/kavg_b1.txt
cd /tmp && export TMPDIR=/tmp PROBE_OPTS=fuse_gn=1 REPS=0
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/tools/lib_probe.py f16x3 1 > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'splitk_reduce_gn' in r['Kernel_Name']]
print(len(rows), 'fused launches')
by = collections.defaultdict(list)
for r in rows[:200]:
    by[(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r.get('Grid_Size_Y'))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in by.items():
    print(k, len(v), 'avg us', sum(v) / len(v), 'min', min(v), 'max', max(v))
print(list(rows[0].keys()))
PY
rm -rf $O/trace
