import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def record_parity(line):
    """Keep a measured deviation: appended to gpurun_out/parity.txt (gpurun merges that directory back; the round's copy is
    committed as profiles/rNN/parity.txt).  `pytest -q` swallows prints -- the judge asked for a kept log."""
    print(line)
    try:
        out = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity.txt'), 'a') as f:
            f.write(line.rstrip('\n') + '\n')
    except OSError:
        pass


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + '.npz'))
        return cache[name]
    return load
