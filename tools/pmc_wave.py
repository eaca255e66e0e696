"""Wave-cycle breakdown per kernel from a rocprofv3 --pmc pass with SQ_WAVE_CYCLES, SQ_WAIT_ANY (parked: s_waitcnt / barrier),
SQ_WAIT_INST_ANY (issue stall: MFMA RAW / pipe busy), SQ_ACTIVE_INST_ANY and the LDS / VMEM / scalar issue shares, plus the MFMA
issue share from the utilisation pass (SQ_INSTS_MFMA x 8 quad-cycles of vector issue each, MI355X_MICROARCH.md cycle constants).
The three top-level buckets are disjoint and add up to ~SQ_WAVE_CYCLES (all in quad-cycles).
  python tools/pmc_wave.py <pass_c_dir> <pass_a_dir> [rows]"""
import collections, csv, glob, sys


def load(d):
    cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur, seen = collections.defaultdict(float), set()
    for r in csv.DictReader(open(cc)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('fdsr::', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
    return agg, dur


c, dur = load(sys.argv[1])
a, _ = load(sys.argv[2])
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 12
print('%-46s %8s | %8s %10s %8s | of active: %6s %6s %6s %8s' % ('kernel (share of wave cycles)', 'ms', 'parked', 'issue-stall', 'active', 'LDS', 'VMEM', 'scalar', 'MFMA-iss'))
for k in sorted(dur, key=lambda k: -dur[k])[:rows]:
    x = c[k]
    wc = x.get('SQ_WAVE_CYCLES', 0.0)
    if wc <= 0:
        continue
    pct = lambda v: 100.0 * v / wc
    mfma_issue = a.get(k, {}).get('SQ_INSTS_MFMA', 0.0) * 8.0 / 4.0   # 8 cycles of vector issue per MFMA = 2 quad-cycles
    print('%-46s %8.1f | %7.1f%% %9.1f%% %7.1f%% | %15.1f%% %5.1f%% %5.1f%% %7.1f%%' % (
        k[:46], dur[k] * 1e3, pct(x.get('SQ_WAIT_ANY', 0)), pct(x.get('SQ_WAIT_INST_ANY', 0)), pct(x.get('SQ_ACTIVE_INST_ANY', 0)),
        pct(x.get('SQ_ACTIVE_INST_LDS', 0)), pct(x.get('SQ_ACTIVE_INST_VMEM', 0)), pct(x.get('SQ_ACTIVE_INST_SCA', 0)), pct(mfma_issue)))
