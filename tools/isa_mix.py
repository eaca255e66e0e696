"""Instruction mix of one kernel of a hipcc -S listing, per basic block: MFMA / VALU (transcendental counted apart) / SALU / LDS /
VMEM / waits / barriers, and an estimate of the vector-issue cycles per block (MI355X_MICROARCH.md: 4 cycles per plain VALU
instruction, 8 per transcendental, an MFMA holds the issue port 8 cycles).  Usage: isa_mix.py listing.s <mangled-name-substring> [min_insts]"""
import re
import sys
import collections

path, key = sys.argv[1], sys.argv[2]
min_insts = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith('E') and ':' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
TRANS = ('v_exp_', 'v_log_', 'v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_sin_', 'v_cos_')


def cls(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith(TRANS):
        return 'trans'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


blocks, cur, name = [], collections.Counter(), 'entry'
ops = collections.Counter()
for l in lines[start + 1:end]:
    s = l.strip()
    if not s or s.startswith((';', '.')) and not s.startswith('.LBB'):
        continue
    if s.startswith('.LBB') or re.match(r'^[\w.]+:', s):
        if sum(cur.values()):
            blocks.append((name, cur))
        name, cur = s.split(':')[0], collections.Counter()
        continue
    op = s.split()[0]
    c = cls(op)
    cur[c] += 1
    if c in ('valu', 'trans'):
        ops[op] += 1
if sum(cur.values()):
    blocks.append((name, cur))
tot = collections.Counter()
print('%-12s %6s %6s %6s %6s %6s %6s %6s %6s | VALU/MFMA  issue-cyc/MFMA' % ('block', 'mfma', 'valu', 'trans', 'salu', 'lds', 'vmem', 'wait', 'barr'))
for name, c in blocks:
    tot.update(c)
    n = sum(c.values())
    if n < min_insts:
        continue
    m = c['mfma']
    v = c['valu'] + c['trans']
    cyc = 4 * c['valu'] + 8 * c['trans'] + 8 * m
    print('%-12s %6d %6d %6d %6d %6d %6d %6d %6d | %6.2f  %8.1f' % (name, m, c['valu'], c['trans'], c['salu'], c['lds'], c['vmem'], c['wait'], c['barrier'],
                                                                    v / m if m else float('nan'), cyc / m if m else float('nan')))
print('total', dict(tot))
print('top VALU ops:', ops.most_common(25))
for l in lines[end:end + 60]:
    if any(k in l for k in ('vgpr_count', 'spill', 'sgpr_count', 'NumVgprs', 'ScratchSize', 'Occupancy', 'NumAgprs', 'LDSByteSize')):
        print(l.strip())
