import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def meas(label, value):
    """Return `value` unchanged; with VITPOSE_TEST_LOG=<file> also append "label value" there, so that the bounds the
    tests assert can be re-derived from a run (bounds are kept at measured + 50 %)."""
    path = os.environ.get('VITPOSE_TEST_LOG')
    if path:
        with open(path, 'a') as f:
            f.write(f'{label}\t{float(value):.4e}\n')
    return value
