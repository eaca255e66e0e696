"""Per-(kernel, grid) summary of a rocprofv3 --kernel-trace run (CSV directory or rocpd .db file)."""
import collections
import csv
import glob
import sqlite3
import sys


def rows_csv(d):
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    for r in csv.DictReader(open(f)):
        yield (r['Kernel_Name'], int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) // int(r['Workgroup_Size_X']),
               int(r['Start_Timestamp']), int(r['End_Timestamp']))


def rows_db(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    q = ('select s.display_name, d.grid_size_x*d.grid_size_y/d.workgroup_size_x/d.workgroup_size_y, d.start, d.end '
         'from %s d join %s s on d.kernel_id = s.id order by d.start' % (kd, ks))
    for r in db.execute(q):
        yield r


def main(d, top=30):
    rows = list(rows_db(d) if d.endswith('.db') else rows_csv(d))
    agg = collections.OrderedDict()
    t0, t1, busy = None, None, 0
    for name, wgs, a, b in rows:
        if 'fdsr::' not in name:
            continue
        short = name.split('fdsr::')[1].split('(')[0][:48]
        agg.setdefault((short, wgs), []).append(b - a)
        t0 = a if t0 is None else min(t0, a)
        t1 = b if t1 is None else max(t1, b)
    tot = sum(sum(v) for v in agg.values())
    print('total fdsr kernel time %.1f ms over a %.1f ms span, %d launches' % (tot / 1e6, (t1 - t0) / 1e6, sum(len(v) for v in agg.values())))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print('%-50s wgs=%-6d n=%4d avg=%8.1f us total=%7.1f ms (%4.1f%%)' % (k[0], k[1], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, 100 * sum(v) / tot))


if __name__ == '__main__':
    main(sys.argv[1])
