"""MFMA / VALU / LDS utilisation per kernel from two rocprofv3 --pmc --kernel-trace passes (see
tools/refresh_profiles.sh).  MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs);
effective clock = GRBM_GUI_ACTIVE / 8 / kernel time (MI355X_MICROARCH.md, DVFS give-back)."""
import collections, csv, glob, sys


def load(d):
    cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur, n, seen = collections.defaultdict(float), collections.Counter(), set()
    for r in csv.DictReader(open(cc)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('fdsr::', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
            n[k] += 1
    return agg, dur, n


a, dur, n = load(sys.argv[1])
b, _, _ = load(sys.argv[2])
print('%-46s %6s %8s %6s %9s %9s %8s %9s' % ('kernel', 'n', 'ms', 'GHz', 'MFMA busy', 'VALU/MFMA', 'LDS/MFMA', 'LDS confl'))
for k in sorted(dur, key=lambda k: -dur[k])[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
    x, y, t = a[k], b.get(k, {}), dur[k]
    simd_cycles = x['GRBM_GUI_ACTIVE'] / 8 * 1024
    mfma = x.get('SQ_INSTS_MFMA', 0)
    print('%-46s %6d %8.1f %6.2f %8.1f%% %9.2f %8.2f %8.2f%%' % (
        k[:46], n[k], t * 1e3, x['GRBM_GUI_ACTIVE'] / 8 / t / 1e9, 100 * x.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd_cycles,
        x.get('SQ_INSTS_VALU', 0) / mfma if mfma else float('nan'), x.get('SQ_INSTS_LDS', 0) / mfma if mfma else float('nan'),
        100 * y.get('SQ_LDS_BANK_CONFLICT', 0) / (y.get('GRBM_GUI_ACTIVE', 1) / 8 * 256)))
