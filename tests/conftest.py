import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def sub(npz, prefix):
    """Dict of the entries below `prefix` with the prefix stripped."""
    return {k[len(prefix):]: npz[k] for k in npz.files if k.startswith(prefix)}


@pytest.fixture(scope='session')
def golden():
    return load_golden


@pytest.fixture(params=['rows', 'tiled'])
def route(request):
    """Runs a GPU test once per network route of the library: the row-local kernels (mlp_rows.h, the default) and the
    tiled multi-launch kernels (option "rows" = 0), which remain the route of shapes the row-local kernels refuse."""
    from curious_amd import ops
    with ops.option('rows', 1 if request.param == 'rows' else 0):
        yield request.param
