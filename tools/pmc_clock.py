"""From a rocprofv3 --pmc + --kernel-trace run: per kernel name, sum of counters, total time and
effective clock = GRBM_GUI_ACTIVE / 8 XCDs / time."""
import collections, csv, glob, sys
d = sys.argv[1]
cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(cc)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('fdsr::', '')
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen and 'Start_Timestamp' in r:
        seen.add(r['Dispatch_Id'])
        dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
for k in sorted(agg, key=lambda k: -dur.get(k, 0))[:int(sys.argv[2]) if len(sys.argv) > 2 else 6]:
    a = agg[k]
    t = dur.get(k, 0)
    print(k, 'time=%.1f ms' % (t * 1e3))
    if t and 'GRBM_GUI_ACTIVE' in a:
        clk = a['GRBM_GUI_ACTIVE'] / 8 / t
        print('   effective clock %.3f GHz' % (clk / 1e9))
        if 'SQ_INSTS_VALU_MFMA_MOPS_F16' in a:
            print('   MFMA F16 MOPS', a['SQ_INSTS_VALU_MFMA_MOPS_F16'])
    for c, v in sorted(a.items()):
        print('     %-34s %.4g' % (c, v))
