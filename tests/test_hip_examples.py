"""The repository's examples (reference flows restated on synthetic meshes) run as scripts on the
GPU (-m gpu): exit code 0; their own assertions are the checks."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,arg", [("poisson_assembly.py", "48"), ("fractures_fem.py", "8"),
                                        ("poisson_large_cg.py", "120")])
def test_example_runs(script, arg):
    done = subprocess.run([sys.executable, os.path.join(REPO, "examples", script), arg],
                          capture_output=True, text=True, timeout=600, cwd=REPO)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    assert done.stdout.strip(), "the example prints what it computed"
