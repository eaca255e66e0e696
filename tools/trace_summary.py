"""Per-(kernel, grid) summary of a rocprofv3 --kernel-trace CSV."""
import collections
import csv
import glob
import sys


def main(d, top=30):
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if 'fdsr' not in name:
            continue
        short = name.split('fdsr::')[1].split('(')[0][:48]
        key = (short, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
        agg.setdefault(key, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    tot = sum(sum(v) for v in agg.values())
    print('total fdsr kernel time %.1f ms' % (tot / 1e6))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print('%-50s wgs=%-6d n=%4d avg=%8.1f us total=%7.1f ms (%4.1f%%)' % (k[0], k[1], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, 100 * sum(v) / tot))


if __name__ == '__main__':
    main(sys.argv[1])
