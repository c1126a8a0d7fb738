import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def report_dir():
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return d


@pytest.fixture(scope="session")
def ckpts():
    """seeded synthetic state_dicts (encoder, gan, sr) — oracle/synth.py, ~7 s once per session"""
    from oracle import synth
    return synth.make_encoder_state_dict(), synth.make_gan_state_dict(), synth.make_sr_state_dict()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz")))
