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


def rowwise_error(got, want, rowptr=None, scale=None):
    """Entry-wise bound, scaled per ROW: max_i max_j |got_ij - want_ij| / s_i.

    The tolerance north_star states (rtol 1e-12) is entry-wise; a plain entry-wise ratio is
    ill-posed where entries cancel to zero, a norm-wise bound (scaled_error) lets a large row hide
    an error in a small one.  Here every entry is measured against the scale of its own row:
    s_i = max_j |want_ij| for CSR values (`rowptr` given; the row's diagonal for a stiffness
    matrix), or the caller's `scale` (vectors: the sum of the magnitudes of the element shares
    that make up entry i).  Rows with s_i = 0 must be reproduced exactly."""
    got = np.asarray(got, dtype=np.float64).ravel()
    want = np.asarray(want, dtype=np.float64).ravel()
    assert got.shape == want.shape, (got.shape, want.shape)
    diff = np.abs(got - want)
    if rowptr is not None:
        rowptr = np.asarray(rowptr, dtype=np.int64)
        starts = rowptr[:-1][np.diff(rowptr) > 0]
        row_of = np.repeat(np.arange(rowptr.size - 1), np.diff(rowptr))
        s = np.zeros(rowptr.size - 1)
        s[np.diff(rowptr) > 0] = np.maximum.reduceat(np.abs(want), starts)
        s = s[row_of]
    else:
        s = np.abs(np.asarray(scale, dtype=np.float64).ravel())
        assert s.shape == want.shape
    zero = s == 0.0
    if zero.any() and diff[zero].max() != 0.0:
        return float("inf")
    if zero.all():
        return 0.0
    return float((diff[~zero] / s[~zero]).max())


@pytest.fixture
def golden():
    return load_golden
