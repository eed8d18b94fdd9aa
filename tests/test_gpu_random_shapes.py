"""Shapes no fixed-size test names: random chain lengths of every manifold, and the BASELINE factor mixes at random small sizes,
through the default plan against the oracle (scripts/stress_sizes.py: 3 Gauss-Newton iterations at 1e-9, then 3 Levenberg-Marquardt iterations in lock step; scripts/stress_mixes.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,count,seed", [("stress_sizes.py", 10, 11), ("stress_mixes.py", 8, 5)])
def test_random_shapes_agree_with_the_oracle(script, count, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), str(count), str(seed)], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "agree with the oracle" in r.stdout
