#!/usr/bin/env python
"""Where the runtime's copy kernels (hipMemcpyAsync device-to-device -> __amd_rocclr_copyBuffer) and ATen elementwise launches sit in a
training step: for every such dispatch of the last step of a rocprofv3 --kernel-trace run, the kernels before and after it on its stream.
usage: copy_sites.py <db|dir> [name-substring=copyBuffer]"""
import collections, glob, os, re, sqlite3, sys
src = sys.argv[1]
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, '**', '*.db'), recursive=True))[0]
sub = sys.argv[2] if len(sys.argv) > 2 else 'copyBuffer'
con = sqlite3.connect(src)
rows = list(con.execute('select start, end, stream_id, name from kernels order by start'))
short = lambda n: re.sub(r'\(.*$', '', n.replace('(anonymous namespace)::', '')).replace('void ', '')[:60]
idx = [i for i, r in enumerate(rows) if 'adamw' in r[3]]
seg = rows[idx[-2] + 1: idx[-1] + 1]
by_stream = collections.defaultdict(list)
for r in seg:
    by_stream[r[2]].append(r)
sites = collections.Counter()
dur = collections.defaultdict(float)
for st, rs in by_stream.items():
    for i, r in enumerate(rs):
        if sub in r[3]:
            prev = short(rs[i - 1][3]) if i else '-'
            nxt = short(rs[i + 1][3]) if i + 1 < len(rs) else '-'
            sites[(prev, nxt)] += 1
            dur[(prev, nxt)] += (r[1] - r[0]) / 1e3
print(f'{sum(sites.values())} dispatches matching "{sub}" in the last step, {sum(dur.values()):.1f} us')
for k, v in sites.most_common(60):
    print(f'{v:4d} x {dur[k] / v:7.1f} us   after {k[0]:60s} before {k[1]}')
