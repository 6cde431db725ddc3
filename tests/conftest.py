"""Shared fixtures.  `-m gpu` tests need a MI355X; everything else runs on CPU."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd"
for p in (str(ROOT), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class PackedCases:
    """Reader for the packed .npz layouts written by tests/golden/make_golden.py."""

    def __init__(self, path):
        self.z = np.load(path, allow_pickle=False)
        self.labels = [str(s) for s in self.z["labels"]]

    def __len__(self):
        return len(self.labels)

    def case(self, k):
        z = self.z
        n = int(z["n"][k])
        ov, om = int(z["off_vec"][k]), int(z["off_mat"][k])
        out = dict(label=self.labels[k], n=n, C=z["C"][om:om + n * n].reshape(n, n).copy(),
                   ret=int(z["ret"][k]), x=z["x"][ov:ov + n], y=z["y"][ov:ov + n])
        if "u" in z.files:
            out.update(u=z["u"][ov:ov + n].copy(), v=z["v"][ov:ov + n].copy(), eps=float(z["eps"][k]))
        return out


@pytest.fixture(scope="session")
def seeded_cases():
    return PackedCases(GOLDEN / "seeded_cases.npz")


@pytest.fixture(scope="session")
def cold_cases():
    return PackedCases(GOLDEN / "cold_cases.npz")


@pytest.fixture(scope="session")
def features_cases():
    return np.load(GOLDEN / "features_cases.npz", allow_pickle=False)


@pytest.fixture(scope="session")
def onegnn_cases():
    return np.load(GOLDEN / "onegnn_cases.npz", allow_pickle=False)
