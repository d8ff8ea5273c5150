"""CPU oracle for the LEOD hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Everything under ``oracle/`` is a plain PyTorch-CPU / numpy restatement of the
reference algorithm (Wuziyi616/LEOD) for the path named in BASELINE.json.  Each
function cites the reference ``file:line`` it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it -- and only as the checker / the reported CPU baseline, never as something the
product (``leod_amd``) routes through.  ``leod_amd`` never imports this package.

Pinning: the oracle is checked against golden vectors produced by importing the
reference itself in the build container (``tests/golden/make_golden.py`` with the
stand-in third-party modules under ``tests/golden/ref_stubs``); see
``tests/test_oracle_golden.py``.  One boundary is *parity-unpinned*: torchvision's
``nms``/``batched_nms`` (torchvision is neither installed here nor vendored by the
reference), which ``oracle/nms.py`` restates from torchvision 0.15's published
semantics.
"""
