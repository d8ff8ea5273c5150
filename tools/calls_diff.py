#!/usr/bin/env python
"""Per-family and per-shape comparison of two bench.py --dump-calls tables (second probe step of each).
usage: python tools/calls_diff.py a.txt b.txt [min_us_delta=15]"""
import collections
import sys


def load(p):
    rows = []
    for l in open(p):
        us, name, ints = l.split(None, 2)
        rows.append((float(us), name, ints.strip()))
    return rows[len(rows) // 2:]


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    thr = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
    fa, fb = collections.Counter(), collections.Counter()
    for us, n, i in a:
        fa[n] += us
    for us, n, i in b:
        fb[n] += us
    print(f'total {sum(fa.values()):9.1f} -> {sum(fb.values()):9.1f} us ({len(a)} / {len(b)} calls)')
    for n in sorted(set(fa) | set(fb), key=lambda n: -(abs(fb[n] - fa[n]))):
        if abs(fb[n] - fa[n]) >= thr:
            print(f'  {n:<36s} {fa[n]:9.1f} -> {fb[n]:9.1f}  ({fb[n] - fa[n]:+.1f})')
    sa, sb = collections.defaultdict(list), collections.defaultdict(list)
    for us, n, i in a:
        sa[(n, i)].append(us)
    for us, n, i in b:
        sb[(n, i)].append(us)
    print('per shape:')
    for k in sorted(set(sa) & set(sb), key=lambda k: -abs(sum(sb[k]) - sum(sa[k]))):
        d = sum(sb[k]) - sum(sa[k])
        if abs(d) >= thr:
            print(f'  {k[0]:<32s} {k[1]:<44s} {sum(sa[k]) / len(sa[k]):8.1f} -> {sum(sb[k]) / len(sb[k]):8.1f} x{len(sb[k])}')


if __name__ == '__main__':
    main()
