import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    out = {}
    for name in ("golden_v1.npz", "golden_k_v1.npz", "golden_k0_v1.npz", "golden_dconv_grad_v1.npz"):     # model / cell goldens, other diffusion orders, dconv gradients
        with np.load(os.path.join(ROOT, "tests", "golden", name)) as z:
            out.update({k: z[k] for k in z.files})
    return out


@pytest.fixture(scope="session")
def adj3d():
    return np.load(os.path.join(ROOT, "tests", "golden", "adj_mx_3d.npy"))


@pytest.fixture(scope="session")
def golden_train():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_train_v1.npz"))


@pytest.fixture(scope="session")
def golden_dropout():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_dropout_v1.npz"))


@pytest.fixture(scope="session")
def golden_fft():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_fft_v1.npz"))
