"""Per-launch durations of the 3x3 conv kernels of the LAST forward in a rocprofv3 kernel trace made with tools/wino_probe.py:
python tools/wino_trace_summary.py <dir> [label]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
conv = [r for r in rows if ('conv_wino' in r['Kernel_Name'] or 'conv_mfma_h_kernel<3, 1, false' in r['Kernel_Name'])]
n = 44
last = conv[-n:]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in last]
lab = sys.argv[2] if len(sys.argv) > 2 else ''
print(lab, 'sum_us', round(sum(d), 1), ' '.join(str(round(v)) for v in d))
