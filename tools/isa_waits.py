#!/usr/bin/env python
"""Full-drain waits in the gfx950 code of one object file: per kernel the number of global loads / stores, of `s_waitcnt vmcnt(0)`, and of
those drains that follow a global store within a few instructions (a store whose acknowledgement the wave then waits for -- what hipcc
emits at control-flow joins inside unrolled epilogues).  usage: python tools/isa_waits.py leod_amd/csrc/build/k_linear.o [name filter]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def main():
    obj, flt = os.path.abspath(sys.argv[1]), (sys.argv[2] if len(sys.argv) > 2 else '')
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, 'k.o')
        os.symlink(obj, local)
        subprocess.run([f'{LLVM}/llvm-objdump', '--offloading', local], cwd=td, capture_output=True)
        co = [f for f in os.listdir(td) if 'gfx950' in f][0]
        dis = subprocess.run([f'{LLVM}/llvm-objdump', '-d', '--no-show-raw-insn', os.path.join(td, co)], capture_output=True, text=True).stdout
    cur, stats = None, {}
    window = 0
    for line in dis.split('\n'):
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
        if m:
            cur = m.group(1)
            stats[cur] = dict(loads=0, stores=0, drains=0, store_drains=0, mfma=0, lines=0)
            window = 0
            continue
        if cur is None:
            continue
        t = line.strip().split()
        if not t:
            continue
        op = t[0]
        s = stats[cur]
        s['lines'] += 1
        if op.startswith('global_load') or op.startswith('buffer_load'):
            s['loads'] += 1
        elif op.startswith('global_store') or op.startswith('buffer_store') or op.startswith('global_atomic'):
            s['stores'] += 1
            window = 12
        elif op.startswith('v_mfma'):
            s['mfma'] += 1
        elif op == 's_waitcnt' and 'vmcnt(0)' in line:
            s['drains'] += 1
            if window > 0:
                s['store_drains'] += 1
        if window > 0 and not op.startswith('global_store'):
            window -= 1
    names = list(stats)
    dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    rows = [(stats[n], d) for n, d in zip(names, dem) if flt in d and stats[n]['lines'] > 50]
    rows.sort(key=lambda r: -r[0]['store_drains'])
    print(f'{"loads":>6} {"stores":>6} {"vmcnt0":>6} {"st->0":>6} {"mfma":>5} {"instr":>6}  kernel')
    for s, d in rows:
        print(f'{s["loads"]:6d} {s["stores"]:6d} {s["drains"]:6d} {s["store_drains"]:6d} {s["mfma"]:5d} {s["lines"]:6d}  {d[:150]}')


if __name__ == '__main__':
    main()
