#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db or *_kernel_stats.csv) into a
small CSV for profiles/: kernel, calls, total_ms, avg_us, min_us, max_us, pct.   usage: rocprof_summary.py <db|dir> <out.csv>"""
import csv
import glob
import os
import re
import sqlite3
import sys


def main(src, dst):
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, '**', '*.db'), recursive=True))[0]
    con = sqlite3.connect(src)
    rows = list(con.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
                            'from kernels group by name order by 3 desc'))
    tot = sum(r[2] for r in rows)
    t0, t1 = con.execute('select min(start), max(end) from kernels').fetchone()
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['# rocprofv3 --kernel-trace --stats summary', f'total_kernel_ms={tot / 1e6:.3f}',
                    f'launches={sum(r[1] for r in rows)}', f'first_to_last_kernel_ms={(t1 - t0) / 1e6:.3f}'])
        w.writerow(['kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'pct'])
        for name, n, s, a, mn, mx in rows:
            name = re.sub(r'\(.*$', '', name.replace('(anonymous namespace)::', '')).replace('void ', '')
            w.writerow([name, n, f'{s / 1e6:.3f}', f'{a / 1e3:.2f}', f'{mn / 1e3:.2f}', f'{mx / 1e3:.2f}', f'{100 * s / tot:.2f}'])
    print('wrote', dst)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
