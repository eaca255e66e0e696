"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name (sum over dispatches)."""
import collections
import csv
import glob
import sys


def main(d):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        short = name.split('(')[0].replace('void ', '').replace('fdsr::', '')
        agg[short][r['Counter_Name']] += float(r['Counter_Value'])
        key = (r['Dispatch_Id'], short)
        if key not in seen:
            seen.add(key)
            n[short] += 1
    for k in sorted(agg, key=lambda k: -agg[k].get('SQ_WAVE_CYCLES', agg[k].get('GRBM_GUI_ACTIVE', 0))):
        print(f'{k}  dispatches={n[k]}')
        for c, v in sorted(agg[k].items()):
            print(f'    {c:32s} {v:.4g}')


if __name__ == '__main__':
    main(sys.argv[1])
