#!/usr/bin/env python
"""End of a training step per HIP stream in a rocprofv3 --kernel-trace run (rocpd .db; steps delimited by adamw_clip_kernel): when does each
stream issue its last kernel before the optimiser, how long is it busy, and what does the launch stream wait for at the join?
usage: stream_tail.py <db|dir>"""
import collections, glob, os, re, sqlite3, sys
src = sys.argv[1]
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, '**', '*.db'), recursive=True))[0]
con = sqlite3.connect(src)
rows = list(con.execute('select start, end, stream_id, name from kernels order by start'))
short = lambda n: re.sub(r'\(.*$', '', n.replace('(anonymous namespace)::', '')).replace('void ', '')[:60]
idx = [i for i, r in enumerate(rows) if 'adamw' in r[3]]
for a, b in zip(idx[-3:-1], idx[-2:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], rows[b][0]
    print(f'step: {(rows[b][1] - rows[a][1]) / 1e6:.3f} ms adamw-to-adamw, {len(seg)} launches')
    by = collections.defaultdict(list)
    for r in seg[:-1]:
        by[r[2]].append(r)
    main = max(by, key=lambda k: len(by[k]))
    for st, rs in sorted(by.items(), key=lambda kv: -len(kv[1])):
        busy = sum(r[1] - r[0] for r in rs) / 1e6
        print(f'  stream {st}{" (launch)" if st == main else ""}: {len(rs)} kernels, busy {busy:.3f} ms, first start +{(rs[0][0] - t0) / 1e6:.3f} ms, '
              f'last end {-(t1 - max(r[1] for r in rs)) / 1e6:.3f} ms before adamw; last: {short(rs[-1][3])}')
    # the last 12 kernels before the optimiser, all streams
    for r in seg[-14:-1]:
        print(f'    {"M" if r[2] == main else "s"} start {-(t1 - r[0]) / 1e3:9.1f} us  dur {(r[1] - r[0]) / 1e3:7.1f} us  {short(r[3])}')
