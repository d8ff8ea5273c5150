#!/usr/bin/env python
"""Kernels of the LAST training step of a rocprofv3 --kernel-trace run (rocpd .db; steps delimited by adamw_clip_kernel), grouped by
name: calls, total us, avg us, stream count -- the per-step picture without the warm-up / set-up launches that a whole-run
summary mixes in.  usage: last_step_kernels.py <db|dir> [out.csv]"""
import collections, glob, os, re, sqlite3, sys
src = sys.argv[1]
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, '**', '*.db'), recursive=True))[0]
con = sqlite3.connect(src)
rows = list(con.execute('select start, end, stream_id, name from kernels order by start'))
short = lambda n: re.sub(r'\(.*$', '', n.replace('(anonymous namespace)::', '')).replace('void ', '')[:110]
idx = [i for i, r in enumerate(rows) if 'adamw' in r[3]]
a, b = idx[-2], idx[-1]
seg = rows[a + 1:b + 1]
acc = collections.OrderedDict()
for s, e, st, n in seg:
    d = acc.setdefault(short(n), [0, 0.0, set()])
    d[0] += 1; d[1] += (e - s) / 1e3; d[2].add(st)
tot = sum(v[1] for v in acc.values())
lines = [f'# last step: {len(seg)} launches, {tot / 1e3:.3f} ms of kernel time, span {(seg[-1][1] - rows[a][1]) / 1e6:.3f} ms, streams {len(set(r[2] for r in seg))}',
         'kernel,calls,total_us,avg_us,streams']
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    lines.append(f'"{k}",{v[0]},{v[1]:.1f},{v[1] / v[0]:.2f},{len(v[2])}')
out = '\n'.join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(out + '\n')
print(out)
