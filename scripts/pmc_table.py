"""Prints one frame's per-dispatch counters from a rocprofv3 --pmc --kernel-trace csv directory."""
import csv, collections, re, sys
d0 = sys.argv[1]
rows = list(csv.DictReader(open(f'{d0}/p_counter_collection.csv')))
kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(f'{d0}/p_kernel_trace.csv'))}
d = collections.OrderedDict()
for r in rows:
    k = r['Dispatch_Id']
    e = d.setdefault(k, {'name': re.sub(r'\(.*', '', r['Kernel_Name'])[:38], 'vgpr': r['VGPR_Count'], 'lds': r['LDS_Block_Size'], 'grid': r['Grid_Size']})
    e[r['Counter_Name']] = float(r['Counter_Value'])
keys = list(d.keys())
last = [i for i, k in enumerate(keys) if 'k_raygen' in d[k]['name']]
a, b = last[-2], last[-1]
names = sorted({c for k in keys[a:b] for c in d[k] if c not in ('name', 'vgpr', 'lds', 'grid')})
for k in keys[a:b]:
    v = d[k]
    if 'rocclr' in v['name']:
        continue
    t = kt.get(k)
    dur = (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3 if t else 0
    print(f"{v['name']:38s} {dur:7.1f}us grid {v['grid']:>8s} vgpr {v['vgpr']:>3s} lds {v['lds']:>6s} " + " ".join(f"{n[3:]}={v.get(n, 0):.3g}" for n in names))
