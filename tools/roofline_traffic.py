#!/usr/bin/env python
"""HBM traffic of the roofline kernel FAMILY from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE and WRITE_SIZE are collected in
separate runs, MI355X_MICROARCH.md).  Corrections applied as that guide prescribes for gfx950: counters are in KiB; FETCH_SIZE of wide
coalesced reads reports half the bytes -> x2.

The family is delimited by marker dispatches (LEOD_FAMILY_MARKERS=1: the library launches ``leod_family_marker_kernel`` in front of and
behind every ``leod_linear_wgrad`` / ``leod_linear_wgrad_gelu16`` call; single-stream eager run, so dispatch order = program order): the
bytes of every dispatch between a begin and an end marker are summed and divided by the number of marker pairs -- the same launches the
HIP-event probe of bench.py brackets (36 per training step), whatever kernels they resolve to (round 3 divided the bytes of the name
pattern wgrad_wide|wgradw by 102 launches of which 30 were 1x1-conv weight gradients: 208 MB reported, 284 MB true).
usage: roofline_traffic.py <fetch.db|dir> <write.db|dir> <out.json> <out.csv>"""
import collections, csv, glob, json, os, re, sqlite3, sys

MARK = 'leod_family_marker_kernel'


def load(src, counter):
    """-> (bytes-per-kernel-name dict of lists, number of marker pairs) for the dispatches inside marker pairs"""
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, '**', '*.db'), recursive=True))[0]
    con = sqlite3.connect(src)
    rows = list(con.execute('select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name = ? '
                            'group by dispatch_id, kernel_name order by dispatch_id', (counter,)))
    inside, pairs = False, 0
    per = collections.defaultdict(list)
    for _, name, val in rows:
        short = re.sub(r'\(.*$', '', name.replace('(anonymous namespace)::', '')).replace('void ', '')
        if MARK in short:
            if inside:
                pairs += 1
            inside = not inside
            continue
        if inside:
            per[short].append(val)
    return per, pairs


def main(fdb, wdb, out_json, out_csv):
    f, nf = load(fdb, 'FETCH_SIZE')
    w, nw = load(wdb, 'WRITE_SIZE')
    tot_f = sum(sum(v) for v in f.values())
    tot_w = sum(sum(v) for v in w.values())
    with open(out_csv, 'w', newline='') as fh:
        wr = csv.writer(fh)
        wr.writerow(['# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) over bench.py; KiB as reported; dispatches inside '
                     f'family markers only: {nf} / {nw} leod_linear_wgrad* calls'])
        wr.writerow(['kernel', 'dispatches', 'FETCH_SIZE_KiB_sum', 'WRITE_SIZE_KiB_sum'])
        for name in sorted(set(f) | set(w)):
            wr.writerow([name, len(f.get(name, [])), f'{sum(f.get(name, [])):.1f}', f'{sum(w.get(name, [])):.1f}'])
    rd = 2 * 1024 * tot_f / max(nf, 1)
    wt = 1024 * tot_w / max(nw, 1)
    json.dump({'kernel': 'every dispatch inside leod_linear_wgrad / leod_linear_wgrad_gelu16 (family markers)', 'launches_sampled': [nf, nw],
               'hbm_bytes_per_launch': round(rd + wt), 'read_bytes_per_launch': round(rd), 'write_bytes_per_launch': round(wt),
               'source': os.path.relpath(out_csv)}, open(out_json, 'w'), indent=1)
    print(open(out_json).read())


if __name__ == '__main__':
    main(*sys.argv[1:5])
