#!/usr/bin/env python
"""Headroom measurement by ablation: runs bench.py with selected C entry points turned into no-ops (the results of the step are then
WRONG -- this only answers "how much of ms_per_step would a perfect fusion / a free kernel of this family give back?" before a kernel is
written).  GPU box only; never part of a reported number.

    python tools/ablate_bench.py "linear_wgrad:M<=60000,bn_silu_fwd" --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-second-dtype

Each item: <entry point without the leod_ prefix>[:M<=N | :M>N]  (the row filter applies to the first integer argument >= 4096 of the call).
"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)


class _Ablated:
    def __init__(self, lib, spec):
        self._lib, self._cache, self.rules, self.skipped = lib, {}, {}, {}
        for item in filter(None, spec.split(',')):
            name, _, cond = item.partition(':')
            self.rules['leod_' + name] = cond

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            real = getattr(self._lib, name)
            cond = self.rules.get(name)
            if cond is None:
                fn = real
            else:
                def fn(*a, _real=real, _cond=cond, _name=name):
                    if _cond:
                        m = next((x for x in a if isinstance(x, int) and not isinstance(x, bool) and x >= 4096), 0)
                        op, lim = ('<=', int(_cond.split('<=')[1])) if '<=' in _cond else ('>', int(_cond.split('>')[1]))
                        if not ((m <= lim) if op == '<=' else (m > lim)):
                            return _real(*a)
                    self.skipped[_name] = self.skipped.get(_name, 0) + 1
                    return 0
            self._cache[name] = fn
        return fn


if __name__ == '__main__':
    spec = sys.argv[1]
    sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
    import leod_amd.ops as ops
    ops._LIB = _Ablated(ops._l(), spec)
    import bench
    bench.main()
    print('ABLATED', spec, ops._LIB.skipped, file=sys.stderr)
