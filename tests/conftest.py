import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")

MESH_KEYS = ("vertices", "vertex_markers", "triangles", "edges", "edge_markers", "neighbors")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False) as data:
        return {k: data[k] for k in data.files}


def mesh_from_golden(data):
    return {k[3:]: v for k, v in data.items() if k.startswith("in_") and k[3:] in MESH_KEYS}


def scaled_error(got, want):
    """max(|got-want|)/max(|want|) and ||got-want||_F/||want||_F, whichever is larger.

    Element-wise rtol is ill-posed here: structured meshes produce exact-zero
    entries by cancellation.
    """
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    scale_max = max(np.abs(want).max(), 1e-300)
    scale_fro = max(np.linalg.norm(want.ravel()), 1e-300)
    return max(
        np.abs(got - want).max() / scale_max,
        np.linalg.norm((got - want).ravel()) / scale_fro,
    )


@pytest.fixture
def golden():
    return load_golden
