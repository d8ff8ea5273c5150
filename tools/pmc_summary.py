#!/usr/bin/env python
"""Per-kernel PMC counter sums from a rocprofv3 --pmc rocpd database.  usage: pmc_summary.py <db|dir> [name-filter]"""
import glob, os, re, sqlite3, sys
src = sys.argv[1]
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, '**', '*.db'), recursive=True))[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
con = sqlite3.connect(src)
cols = [r[1] for r in con.execute('pragma table_info(counters_collection)')]
q = ('select kernel_name, grid_size_x, grid_size_y, counter_name, sum(value), count(*) from counters_collection '
     'group by kernel_name, grid_size_x, grid_size_y, counter_name') if 'grid_size_x' in cols else None
if q is None:
    print(cols)
    sys.exit()
acc = {}
for name, gx, gy, cname, val, n in con.execute(q):
    name = re.sub(r'\(.*$', '', name.replace('(anonymous namespace)::', '')).replace('void ', '')
    if flt and flt not in name:
        continue
    acc.setdefault((name, gx, gy), {})[cname] = (val, n)
for (name, gx, gy), d in sorted(acc.items()):
    n = max(v[1] for v in d.values())
    print(f'{name[:70]} grid=({gx},{gy}) dispatches={n}')
    print('   ' + '  '.join(f'{k}={v[0] / v[1]:.3g}' for k, v in sorted(d.items())))
