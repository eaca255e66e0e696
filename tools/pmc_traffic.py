"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), and for the launch set
bench.py's roofline brackets with HIP events: every 3x3 convolution op (conv_mfma_h_kernel<3,...>, conv_k32_kernel, conv_strip_kernel,
conv_up2_h_kernel, conv_mfma_f32_kernel<3,...> and the split-K reduce that finishes such an op).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of wide coalesced
reads -> doubled; WRITE_SIZE is exact for 16-B streaming stores.  Units: KiB.
  python tools/pmc_traffic.py <fetch_dir> <write_dir> > profiles/rNN_pmc_hbm_traffic_<prec>_b<B>.json
The output records the SHA-256 of the kernel sources it was measured on; bench.py quotes `traffic` from it
only while that hash matches the tree."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    return name.split('<')[0].replace('void ', '').replace('fdsr::', '').split('(')[0]


def is_3x3(name):
    k = short(name)
    if k in ('conv_up2_h_kernel', 'conv_up2_k32_kernel', 'conv_k32_kernel', 'conv_strip_kernel', 'conv_in8_kernel', 'conv_out3_kernel'):
        return True
    if k in ('conv_mfma_h_kernel', 'conv_mfma_f32_kernel'):
        m = re.search(r'<\s*(\d+)\s*,', name)
        return bool(m) and m.group(1) == '3'
    return False


out = {}
fam = {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0, 'launches': 0}
for d, ctr in ((sys.argv[1], 'FETCH_SIZE'), (sys.argv[2], 'WRITE_SIZE')):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    tot = collections.defaultdict(float); n = collections.Counter(); seen = set(); nf = 0
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != ctr: continue
        k = short(r['Kernel_Name'])
        v = float(r['Counter_Value']) * 1024.0
        tot[k] += v
        first = (r['Dispatch_Id'], k) not in seen
        if first:
            seen.add((r['Dispatch_Id'], k)); n[k] += 1
        if is_3x3(r['Kernel_Name']):
            fam[ctr] += v
            nf += 1 if first else 0
    fam['launches'] = nf
    for k in tot:
        out.setdefault(k, {})[ctr] = tot[k]; out[k]['launches'] = n[k]
res = {}
for k, v in out.items():
    rd = 2.0 * v.get('FETCH_SIZE', 0.0); wr = v.get('WRITE_SIZE', 0.0)
    res[k] = {'launches': v['launches'], 'read_bytes_corrected': rd, 'write_bytes': wr,
              'hbm_bytes_per_launch': (rd + wr) / max(v['launches'], 1)}
from bench import kernel_source_hash
frd, fwr = 2.0 * fam['FETCH_SIZE'], fam['WRITE_SIZE']
res = {'kernel_source_hash': kernel_source_hash(),
       'family_3x3': {'launches': fam['launches'], 'read_bytes_corrected': frd, 'write_bytes': fwr,
                      'hbm_bytes_per_launch': (frd + fwr) / max(fam['launches'], 1),
                      'note': 'split-K reduce kernels (small grids) are not in this set; none runs at B>=16 but one stride-2 op'},
       'per_kernel': res}
print(json.dumps(res, indent=1))
