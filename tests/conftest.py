import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
PKG_NAME = 'joint-regressor-refinement_amd'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope='session')
def pkg():
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope='session')
def smpl_model_np():
    m = importlib.import_module(PKG_NAME + '.smpl_model')
    return m.synthetic_smpl(1234)


@pytest.fixture(scope='session')
def j_h36m_np():
    m = importlib.import_module(PKG_NAME + '.smpl_model')
    t = load_golden('j_regressor_triplets.npz')
    return m.j_regressor_from_triplets(t['rows'], t['cols'], t['vals'])
