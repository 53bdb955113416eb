"""The committed fixture recipe runs and reproduces the committed fixtures bit for bit.

Build-container test: needs the reference under /root/reference (absent on the GPU box, where
this test is skipped).  tests/golden/tools/make_golden.py is run as a script into a scratch
directory; every array of every committed .npz must come out identical (dtype, shape, bytes).
"""

import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
REFERENCE = os.environ.get("TFEM_REFERENCE_ROOT", "/root/reference")

pytestmark = pytest.mark.skipif(
    not os.path.isdir(os.path.join(REFERENCE, "torch_fem")), reason="the reference is not on this machine"
)


def test_recipe_reproduces_the_committed_fixtures(tmp_path):
    script = os.path.join(GOLDEN, "tools", "make_golden.py")
    done = subprocess.run(
        [sys.executable, script, "--out", str(tmp_path)], cwd=str(tmp_path),
        capture_output=True, text=True, timeout=900,
    )
    assert done.returncode == 0, done.stdout + done.stderr
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert committed, "no fixtures"
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz")) == committed
    for name in committed:
        have, want = np.load(os.path.join(tmp_path, name)), np.load(os.path.join(GOLDEN, name))
        assert sorted(have.files) == sorted(want.files), name
        for key in want.files:
            a, b = have[key], want[key]
            assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), (name, key)
