"""Sums rocprofv3 --pmc counters per kernel over the counter_collection CSVs under the given directories.
Usage: python tools/pmc_summary.py out.json dir1 [dir2 ...]   (one directory per PMC pass; counters of all passes are merged)"""
import csv, glob, json, os, re, sys

def short(name):
    if name.startswith('_ZN'):                                  # still mangled: keep the kernel's own identifier + template digits
        m = re.search(r'N_1\d+([A-Za-z_0-9]*?_kernel)(I\S*?E)?Ev', name)
        if m: return (m.group(1) + (m.group(2) or ''))[:80]
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0][:80]

def main(out, dirs):
    tab = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            seen = {}
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                e = tab.setdefault(k, {})
                c = r['Counter_Name']
                e[c] = e.get(c, 0.0) + float(r['Counter_Value'])
                seen.setdefault((k, c), set()).add(r['Dispatch_Id'])
            for (k, c), s in seen.items():
                tab[k]['launches:' + c] = tab[k].get('launches:' + c, 0) + len(s)
    res = {}
    for k, e in tab.items():
        wc = e.get('SQ_WAVE_CYCLES')
        row = {c: v for c, v in e.items() if not c.startswith('launches:')}
        row['launches'] = max([v for c, v in e.items() if c.startswith('launches:')] or [0])
        if wc:
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS'):
                if c in e: row['frac_wave_cycles:' + c] = round(e[c] / wc, 4)
        if e.get('SQ_LDS_IDX_ACTIVE'):
            row['lds_conflict_frac'] = round(e.get('SQ_LDS_BANK_CONFLICT', 0.0) / e['SQ_LDS_IDX_ACTIVE'], 4)
        if e.get('SQ_BUSY_CU_CYCLES') and 'SQ_VALU_MFMA_BUSY_CYCLES' in e:
            row['mfma_busy_per_cu_busy_cycle'] = round(e['SQ_VALU_MFMA_BUSY_CYCLES'] / e['SQ_BUSY_CU_CYCLES'], 4)
        res[k] = row
    res = dict(sorted(res.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', kv[1].get('SQ_BUSY_CU_CYCLES', 0))))
    json.dump(res, open(out, 'w'), indent=1)
    print(f'{len(res)} kernels -> {out}')

if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
