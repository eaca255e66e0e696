"""Side-by-side per-kernel totals of rocprofv3 kernel_stats.csv files (conv kernels by template arguments)."""
import csv, re, sys


def short(name):
    m = re.match(r'.*?(conv_\w+kernel)<(.*)>', name)
    if m:
        return m.group(1).replace('_kernel', '') + '<' + m.group(2).replace(' ', '').replace('(bool)', '').replace('(fdsr::Precision)', 'P') + '>'
    return re.sub(r'\(.*', '', name).replace('void fdsr::', '')[:60]


tabs = []
for f in sys.argv[1:]:
    t = {}
    for r in csv.DictReader(open(f)):
        k = short(r['Name'])
        c, d = t.get(k, (0, 0.0))
        t[k] = (c + int(r['Calls']), d + float(r['TotalDurationNs']) / 1e6)
    tabs.append(t)
keys = sorted(set().union(*tabs), key=lambda k: -max(t.get(k, (0, 0))[1] for t in tabs))
print('%-64s' % 'kernel' + ''.join('%22s' % f.split('/')[-1][-20:] for f in sys.argv[1:]))
for k in keys[:40]:
    print('%-64s' % k[:64] + ''.join('%8d x %9.2f ms' % t.get(k, (0, 0.0)) for t in tabs))
print('%-64s' % 'TOTAL' + ''.join('%22.2f' % sum(v[1] for v in t.values()) for t in tabs))
