"""Per-kernel statistics (calls, total / average / min / max duration, share) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace -d DIR -o NAME -- ...` writes DIR/NAME_results.db on this image).  Usage: rocpd_stats.py results.db [out.csv]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                   f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
for n, c, t, a, mn, mx in rows:
    lines.append(f'"{n}",{c},{t},{a:.1f},{100.0 * t / tot:.4f},{mn},{mx}')
# family rows: the 256-row NT GEMM kernels exist as one instantiation per epilogue kind; bench.py's roofline line is about the family
fams = {'gemm256_kernel<false, 2, *> (NT 256x256, all epilogue instantiations)': '14gemm256_kernelILb0ELi2ELi',
        'gemm192_kernel<*> (NT 256x192, all epilogue instantiations)': '14gemm192_kernelILi'}
for label, pat in fams.items():
    sel = [r for r in rows if pat in r[0]]
    if sel:
        c, t = sum(r[1] for r in sel), sum(r[2] for r in sel)
        lines.append(f'"FAMILY {label}",{c},{t},{t / c:.1f},{100.0 * t / tot:.4f},{min(r[4] for r in sel)},{max(r[5] for r in sel)}')
out = '\n'.join(lines) + '\n'
if len(sys.argv) > 2: open(sys.argv[2], 'w').write(out)
for l in lines[:45] + [l for l in lines[45:] if l.startswith('"FAMILY')]: print(l[:170])
print('total kernel ns', tot)
