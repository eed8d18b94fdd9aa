"""Regenerates tests/golden/jac_steps.json: the reference test cases whose analytic Jacobian meets the reference's tolerance only
at a central-difference step larger than the reference's own (rounding-limited differences, see tests/test_oracle_golden.py:_jac_ok),
with the smallest step of the ladder 1e-5, 1e-4, 1e-3 at which they do.  Every case NOT in the table has to pass at the reference step.
    python tests/golden/make_jac_steps.py
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
out = os.path.join(HERE, "jac_steps.json")
env = dict(os.environ, GPSLAM_JAC_DISCOVER=out)
sys.exit(subprocess.call([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-q"], env=env, cwd=ROOT))
