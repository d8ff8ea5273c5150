#!/usr/bin/env python
"""Register / LDS / scratch use of the kernels of one object file (gfx950 code object metadata).
usage: python tools/kernel_regs.py leod_amd/csrc/build/k_linear.o [name filter]"""
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
        co = [f for f in os.listdir(td) if 'gfx950' in f]
        notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', os.path.join(td, co[0])], capture_output=True, text=True).stdout
        dem = 'c++filt'
        for ent in notes.split('- .agpr_count:')[1:]:
            m = re.search(r'\.name:\s+(\S+)', ent)
            if not m:
                continue
            name = subprocess.run([dem, m.group(1)], capture_output=True, text=True).stdout.strip()
            if flt not in name:
                continue
            g = lambda k: (re.findall(k + r':\s+(\d+)', ent) or ['?'])[0]  # noqa
            print(f"vgpr {g(r'.vgpr_count'):>4} agpr {ent.split()[0]:>3} spill {g(r'.vgpr_spill_count'):>3} scratch {g(r'.private_segment_fixed_size'):>4} "
                  f"lds {g(r'.group_segment_fixed_size'):>6}  {name[:150]}")


if __name__ == '__main__':
    main()
