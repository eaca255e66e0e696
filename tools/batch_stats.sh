#!/bin/bash
# per-kernel time PER IMAGE at several batch sizes (rocprofv3 kernel stats of the same bench command): does a layer get cheaper when
# its tensors fit the 256 MiB Infinity Cache?   bash tools/batch_stats.sh "4 8 16" [bench args]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/batchstats; mkdir -p $O
BS="$1"; shift
cd /tmp && export TMPDIR=/tmp
for b in $BS; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/run$b -o s -- python3 $R/bench.py "$@" --batch $b --steps 3 --warmup 1 \
    --no-cpu-baseline --no-sub-records --no-profile > $O/run$b.log 2>&1 < /dev/null
  f=$(find $O/run$b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_b$b.csv
  rm -rf $O/run$b
done
cd $R
python - "$BS" <<'PY'
import csv, re, sys, os
bs = sys.argv[1].split()
O = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'batchstats')
def short(name):
    m = re.match(r'.*?(conv_\w+kernel)<(.*)>', name)
    if m:
        return m.group(1).replace('_kernel', '') + '<' + m.group(2).replace(' ', '').replace('(bool)', '').replace('(fdsr::Precision)', 'P') + '>'
    return re.sub(r'\(.*', '', name).replace('void fdsr::', '')[:60]
tabs = {}
for b in bs:
    t = {}
    for r in csv.DictReader(open(f'{O}/kernel_stats_b{b}.csv')):
        k = short(r['Name']); c, d = t.get(k, (0, 0.0)); t[k] = (c + int(r['Calls']), d + float(r['TotalDurationNs']) / 1e3)
    tabs[b] = t
keys = sorted(set().union(*tabs.values()), key=lambda k: -max(t.get(k, (0, 0))[1] for t in tabs.values()))
print('microseconds per IMAGE-forward (total kernel time / (batch x 80 forwards))')
print('%-50s' % 'kernel' + ''.join('%16s' % ('B=' + b) for b in bs))
for k in keys[:26]:
    print('%-50s' % k[:50] + ''.join('%16.2f' % (tabs[b].get(k, (0, 0.0))[1] / (int(b) * 80)) for b in bs))
print('%-50s' % 'TOTAL' + ''.join('%16.2f' % (sum(v[1] for v in tabs[b].values()) / (int(b) * 80)) for b in bs))
PY
