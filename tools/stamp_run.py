import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET
from fastdiffsr_amd.engine import Engine
from fastdiffsr_amd.synth import synth_state_dict
cfg = UNetConfig(**FASTDIFFSR_UNET)
eng = Engine(cfg); eng.load_state_dict(synth_state_dict(cfg, 0)); eng.set_precision('f16x3')
x = torch.randn(16, 6, 256, 256, device='cuda'); nl = torch.full((16,), 0.5, device='cuda')
for _ in range(3):
    eng.unet_forward(x, nl)
torch.cuda.synchronize()
import csv
rows = list(csv.DictReader(open('gpurun_out/stamps.csv')))
import statistics as st
pro = [int(r['loop']) - int(r['start']) for r in rows]
main = [int(r['epi']) - int(r['loop']) for r in rows]
epi = [int(r['end']) - int(r['epi']) for r in rows]
tot = [int(r['end']) - int(r['start']) for r in rows]
rt = [(int(r['rt1']) - int(r['rt0'])) * 10 for r in rows]   # ns (100 MHz)
print('wgs', len(rows))
for n, v in (('prologue', pro), ('main', main), ('epilogue', epi), ('total', tot)):
    print(f'{n:9s} cycles median {st.median(v):9.0f} mean {st.mean(v):9.0f} max {max(v)}')
print('wall ns median', st.median(rt), ' => clock GHz', st.median(tot) / st.median(rt))
starts = sorted(int(r['rt0']) for r in rows); ends = sorted(int(r['rt1']) for r in rows)
print('kernel span us', (ends[-1] - starts[0]) / 100.0, 'first-wave start spread us', (starts[255] - starts[0]) / 100.0)
