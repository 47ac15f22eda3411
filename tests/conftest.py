import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _pkg(name):
    return importlib.import_module("3dbodyanimation_amd." + name)


@pytest.fixture(scope="session")
def synth():
    return _pkg("synth")


@pytest.fixture(scope="session")
def api():
    return _pkg("api")


@pytest.fixture(scope="session")
def model(synth):
    return synth.make_model(0)


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def omodel(oracle_mod, model):
    return oracle_mod.OracleModel(model)


@pytest.fixture(scope="session")
def gpu_model(api, model):
    if api.device_count() < 1:
        pytest.fail("no GPU visible: -m gpu tests need an MI355X (there is no CPU fallback)")
    return api.Model(model, device=0)


def random_params(rng, F, pose_sigma=0.3, nJ=24):
    x = np.zeros((F, 76))
    x[:, 0] = rng.uniform(0.7, 1.4, F)
    x[:, 1:4] = rng.normal(scale=0.3, size=(F, 3))
    x[:, 4:7] = np.array([0.0, 0.0, 3.0]) + rng.normal(scale=0.2, size=(F, 3))
    x[:, 7:] = rng.normal(scale=pose_sigma, size=(F, 3 * (nJ - 1)))
    return x
