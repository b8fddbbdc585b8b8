"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes).
usage: python tools/hbm_traffic.py out.json <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>"""
import csv, glob, json, os, re, sys

def short(name):
    if name.startswith('_ZN'):                                  # still mangled: keep the kernel's own identifier + template digits
        m = re.search(r'N_1\d+([A-Za-z_0-9]*?_kernel)(I\S*?E)?Ev', name)
        if m: return (m.group(1) + (m.group(2) or ''))[:64]
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

# Kernels whose reads are NOT wide (16 B per lane) coalesced streams: 4-byte gathers / per-lane scalar reads.  The guide's gfx950
# correction (FETCH_SIZE tallies 128-B requests at 64 B -> double it) is calibrated for wide streaming reads only; for these the raw
# counter is reported (doubling it made ctc_gather_kernel "read" 7.5 TB/s in round 1).
NARROW_READS = ('ctc_gather_kernel', 'ctc_alphabeta_kernel', 'brn_finalize_kernel', 'brn_bwd_finalize_kernel', 'brn_bump_kernel',
                'norm_bwd_reduce_kernel', 'splitk_reduce_kernel')

def main(out, dfetch, dwrite, batch='128'):
    f, nf = per_kernel(dfetch, 'FETCH_SIZE'); w, nw = per_kernel(dwrite, 'WRITE_SIZE')
    ks = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0))):
        n = max(nf.get(k, 0), nw.get(k, 0))
        fk, wk = f.get(k, 0.0) / max(nf.get(k, 1), 1), w.get(k, 0.0) / max(nw.get(k, 1), 1)
        narrow = any(k.startswith(x) for x in NARROW_READS)
        ks[k] = {'launches': n, 'fetch_size_kb': round(fk, 1), 'write_size_kb': round(wk, 1), 'fetch_correction': 1 if narrow else 2,
                 'hbm_bytes_per_launch': int((1 if narrow else 2) * fk * 1024 + wk * 1024)}
    # family rows (one instantiation per epilogue kind): launch-weighted mean over the members, under the name bench.py looks up
    for fam, pats in (('gemm256_kernel<false, 2, *>', ('gemm256_kernelILb0ELi2ELi', 'gemm256_kernel<false, 2, ')), ('gemm192_kernel<*>', ('gemm192_kernelILi', 'gemm192_kernel<'))):
        mem = [k for k in ks if k.startswith(pats)]
        if mem:
            n = sum(ks[k]['launches'] for k in mem)
            ks[fam] = {'launches': n, 'members': mem, 'fetch_correction': 2,
                       'fetch_size_kb': round(sum(ks[k]['fetch_size_kb'] * ks[k]['launches'] for k in mem) / n, 1),
                       'write_size_kb': round(sum(ks[k]['write_size_kb'] * ks[k]['launches'] for k in mem) / n, 1),
                       'hbm_bytes_per_launch': int(sum(ks[k]['hbm_bytes_per_launch'] * ks[k]['launches'] for k in mem) / n)}
    about = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline` '
             f'(config 3, per-GPU batch {batch}), averaged per launch.  Counter unit: KB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports '
             'half of the bytes of WIDE coalesced reads (16 B per lane), so hbm_bytes_per_launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 for the '
             'streaming kernels; kernels with narrow / gathered reads (fetch_correction = 1: ' + ', '.join(NARROW_READS) + ') keep the raw counter, '
             'which the guide calls uncalibrated for such accesses.  The counters sit on the L2 memory side: Infinity-Cache hits are included.')
    json.dump({'_about': about, 'kernels': ks}, open(out, 'w'), indent=1)
    print(f'{len(ks)} kernels -> {out}')
    for k in list(ks)[:6]: print(k, ks[k])

if __name__ == '__main__':
    main(*sys.argv[1:5])
