#!/usr/bin/env python
"""Per-kernel MFMA utilisation and LDS bank-conflict ratio from rocprofv3 --pmc passes over bench.py.
pass A: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE   pass B: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) x 1024 SIMDs) (MI355X_MICROARCH.md: BUSY_CYCLES counts
cycles, summed over the 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE comes once per XCD and is summed over the 8 XCDs here); LDS conflict ratio = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
usage: pmc_util_summary.py <passA db|dir> <passB db|dir> <out.csv>"""
import csv, glob, os, re, sqlite3, sys


def load(src):
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, '**', '*.db'), recursive=True))[0]
    con = sqlite3.connect(src)
    acc = {}
    for name, cname, val, n in con.execute('select kernel_name, counter_name, sum(value), count(distinct dispatch_id) '
                                           'from counters_collection group by kernel_name, counter_name'):
        name = re.sub(r'\(.*$', '', name.replace('(anonymous namespace)::', '')).replace('void ', '')
        acc.setdefault(name, {})[cname] = (val, n)
    return acc


def main(a, b, out):
    A, B = load(a), load(b)
    rows = []
    for name in sorted(set(A) | set(B)):
        da, db = A.get(name, {}), B.get(name, {})
        busy, gui = da.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0)), da.get('GRBM_GUI_ACTIVE', (0, 1))
        conf, act = db.get('SQ_LDS_BANK_CONFLICT', (0, 0)), db.get('SQ_LDS_IDX_ACTIVE', (0, 0))
        util = busy[0] / (gui[0] / 8 * 1024) if gui[0] else 0.0
        rows.append((gui[0], name, gui[1], busy[0], gui[0], util, conf[0], act[0], conf[0] / act[0] if act[0] else 0.0))
    rows.sort(reverse=True)
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['# rocprofv3 --pmc (separate passes) over bench.py --steps 2 --warmup 1; sums over all dispatches of the kernel'])
        w.writerow(['kernel', 'dispatches', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'mfma_util(busy/((gui/8)*1024))',
                    'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'lds_conflict_ratio'])
        for _, name, n, busy, gui, util, conf, act, ratio in rows:
            w.writerow([name, n, f'{busy:.0f}', f'{gui:.0f}', f'{util:.4f}', f'{conf:.0f}', f'{act:.0f}', f'{ratio:.4f}'])
    print('wrote', out)


if __name__ == '__main__':
    main(*sys.argv[1:4])
