#!/usr/bin/env python
"""Idle gaps of the launch stream in a rocprofv3 --kernel-trace run (rocpd .db): per training step (delimited by adamw_clip_kernel)
busy / span / idle time, and the largest gaps with the kernels on both sides.  usage: stream_gaps.py <db|dir> [min_gap_us=15]"""
import collections, glob, os, re, sqlite3, sys
src = sys.argv[1]
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, '**', '*.db'), recursive=True))[0]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
con = sqlite3.connect(src)
rows = list(con.execute('select start, end, stream_id, name from kernels order by start'))
short = lambda n: re.sub(r'\(.*$', '', n.replace('(anonymous namespace)::', '')).replace('void ', '')[:70]
main = collections.Counter(r[2] for r in rows).most_common(1)[0][0]
mr = [r for r in rows if r[2] == main]
side = [r for r in rows if r[2] != main]
idx = [i for i, r in enumerate(mr) if 'adamw' in r[3]]
print(f'{len(idx)} steps; streams: main {len(mr)} kernels, others {len(side)}')
for a, b in zip(idx[:-1], idx[1:]):
    seg = mr[a + 1:b + 1]
    busy = sum(r[1] - r[0] for r in seg) / 1e6
    span = (seg[-1][1] - mr[a][1]) / 1e6
    sb = sum(r[1] - r[0] for r in side if mr[a][1] <= r[0] < seg[-1][1]) / 1e6
    print(f'step: {len(seg)} launches, main busy {busy:.2f} ms, span {span:.2f} ms, idle {span - busy:.2f} ms, side-stream busy {sb:.2f} ms')
a, b = idx[-2], idx[-1]
seg = mr[a:b + 1]
gaps = [((seg[i + 1][0] - seg[i][1]) / 1e3, short(seg[i][3]), short(seg[i + 1][3])) for i in range(len(seg) - 1)]
small = sum(g[0] for g in gaps if g[0] < min_gap)
print(f'last step: {len(gaps)} gaps, sum of gaps < {min_gap} us: {small / 1e3:.2f} ms (avg {small / max(1, sum(1 for g in gaps if g[0] < min_gap)):.2f} us)')
print(f'gaps >= {min_gap} us:')
for g in sorted(gaps, key=lambda g: -g[0]):
    if g[0] >= min_gap:
        print(f'  {g[0]:8.1f} us   after {g[1]:70s} before {g[2]}')
