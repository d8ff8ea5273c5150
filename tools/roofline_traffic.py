#!/usr/bin/env python
"""HBM traffic of the roofline kernel from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE and WRITE_SIZE are
collected in separate runs, MI355X_MICROARCH.md).  Corrections applied as that guide prescribes for gfx950:
counters are in KiB; FETCH_SIZE of wide coalesced reads reports half the bytes -> x2.
usage: roofline_traffic.py <fetch.db|dir> <write.db|dir> <kernel-regex> <out.json> <out.csv> [launch-regex]
The bytes of every dispatch matching kernel-regex are summed; they are divided by the dispatches matching launch-regex (default: the
same) -- a weight-gradient "launch" of precision mode bf16 is wgrad_wide_bf16_kernel plus its wgrad_wide_reduce_kernel."""
import csv, glob, json, os, re, sqlite3, sys


def load(src, counter, sub):
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, '**', '*.db'), recursive=True))[0]
    con = sqlite3.connect(src)
    rows = {}
    for name, disp, val in con.execute('select kernel_name, dispatch_id, sum(value) from counters_collection '
                                       'where counter_name = ? group by kernel_name, dispatch_id', (counter,)):
        name = re.sub(r'\(.*$', '', name).replace('void ', '')
        if re.search(sub, name):
            rows.setdefault(name, []).append(val)
    return rows


def main(fdb, wdb, sub, out_json, out_csv, launch_sub=None):
    f, w = load(fdb, 'FETCH_SIZE', sub), load(wdb, 'WRITE_SIZE', sub)
    launch_sub = launch_sub or sub
    tot_f = tot_w = n_f = n_w = 0
    with open(out_csv, 'w', newline='') as fh:
        wr = csv.writer(fh)
        wr.writerow(['# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) over bench.py; KiB as reported'])
        wr.writerow(['kernel', 'dispatches', 'FETCH_SIZE_KiB_sum', 'WRITE_SIZE_KiB_sum', 'read_bytes_per_launch(x2 corrected)', 'write_bytes_per_launch'])
        for name in sorted(set(f) | set(w)):
            fv, wv = f.get(name, []), w.get(name, [])
            tot_f += sum(fv); tot_w += sum(wv)
            if re.search(launch_sub, name):
                n_f += len(fv); n_w += len(wv)
            wr.writerow([name, len(fv), f'{sum(fv):.1f}', f'{sum(wv):.1f}',
                         f'{2 * 1024 * sum(fv) / max(len(fv), 1):.0f}', f'{1024 * sum(wv) / max(len(wv), 1):.0f}'])
    rd = 2 * 1024 * tot_f / max(n_f, 1)
    wt = 1024 * tot_w / max(n_w, 1)
    json.dump({'kernel': sub, 'launches_sampled': [n_f, n_w], 'hbm_bytes_per_launch': round(rd + wt),
               'read_bytes_per_launch': round(rd), 'write_bytes_per_launch': round(wt), 'source': os.path.relpath(out_csv)},
              open(out_json, 'w'), indent=1)
    print(open(out_json).read())


if __name__ == '__main__':
    main(*sys.argv[1:7])
