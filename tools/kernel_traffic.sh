#!/bin/bash
# HBM bytes per launch of chosen kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of tools/lib_probe.py; gfx950: FETCH x 2):
#   bash tools/kernel_traffic.sh <prec> <batch> <kernel-name regex>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ktraffic; mkdir -p $O
PREC=$1; B=$2; RE="$3"
cd /tmp && export TMPDIR=/tmp
export REPS=0
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o t -- python3 $R/tools/lib_probe.py $PREC $B > $O/$c.log 2>&1 < /dev/null
done
python3 - "$O" "$RE" <<'PY'
import collections, csv, glob, re, sys
O, RE = sys.argv[1], sys.argv[2]
tab = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for r in csv.DictReader(open(glob.glob(O + '/' + c + '/**/*counter_collection.csv', recursive=True)[0])):
        if r['Counter_Name'] == c and re.search(RE, r['Kernel_Name']):
            k = r['Kernel_Name'].split('(')[0].replace('void fdsr::', '')
            tab[k][c] += float(r['Counter_Value']) * 1024.0
            n[k].add((c, r['Dispatch_Id']))
for k in tab:
    nl = max(1, len([1 for c, d in n[k] if c == 'FETCH_SIZE']))
    print('%-60s launches %4d  read %8.1f MB (FETCH x 2)  write %8.1f MB  per launch' % (k[:60], nl, 2 * tab[k]['FETCH_SIZE'] / nl / 1e6, tab[k]['WRITE_SIZE'] / nl / 1e6))
PY
find $O -name "*counter_collection.csv" -delete 2>/dev/null
