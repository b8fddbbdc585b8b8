"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes).
usage: python tools/hbm_traffic.py out.json <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>"""
import csv, glob, json, os, re, sys

def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0][:64]

def per_kernel(d, counter):
    tot, n = {}, {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter: continue
            k = short(r['Kernel_Name'])
            tot[k] = tot.get(k, 0.0) + float(r['Counter_Value']); n[k] = n.get(k, 0) + 1
    return tot, n

def main(out, dfetch, dwrite):
    f, nf = per_kernel(dfetch, 'FETCH_SIZE'); w, nw = per_kernel(dwrite, 'WRITE_SIZE')
    ks = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0))):
        n = max(nf.get(k, 0), nw.get(k, 0))
        fk, wk = f.get(k, 0.0) / max(nf.get(k, 1), 1), w.get(k, 0.0) / max(nw.get(k, 1), 1)
        ks[k] = {'launches': n, 'fetch_size_kb': round(fk, 1), 'write_size_kb': round(wk, 1), 'hbm_bytes_per_launch': int(2 * fk * 1024 + wk * 1024)}
    about = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline` '
             '(config 3, B = 64), averaged per launch.  Counter unit: KB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of '
             'the bytes of wide coalesced reads, so hbm_bytes_per_launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.')
    json.dump({'_about': about, 'kernels': ks}, open(out, 'w'), indent=1)
    print(f'{len(ks)} kernels -> {out}')
    for k in list(ks)[:6]: print(k, ks[k])

if __name__ == '__main__':
    main(*sys.argv[1:4])
