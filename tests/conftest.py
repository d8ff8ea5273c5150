import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
# The parity tests pin the fp32 mode of the library (bit-tight against the fp32 oracle); the reference's configs say precision=16,
# which the modules map to the bf16 mode -- tests of that mode opt in explicitly (tests/test_bf16_gpu.py).
os.environ.setdefault('LEOD_PRECISION', 'f32')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def manifest():
    import json
    return json.load(open(os.path.join(GOLDEN, 'g11_manifest.json')))
