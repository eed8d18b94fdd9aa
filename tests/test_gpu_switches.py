"""The A/B switches select whole kernel families (read once per process): the parity suites are re-run in child processes
with the switch set, so that the path a default run does NOT take at the test sizes stays covered."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rerun(env, files, k=None):
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + [os.path.join(ROOT, "tests", f) for f in files]
    if k:
        cmd += ["-k", k]
    r = subprocess.run(cmd, env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout, r.stdout[-500:]


def test_block6_chains_through_the_unfused_row_layout_kernels():
    """Chains of block size 6 take k_fused_level0<.., 6> up to 131072 states and k_assemble_ghost + k_chunk_forward_rows<6> beyond
    (api_impl.inc: fused_kernel_applies); GPSLAM_FUSE_B6=0 sends the small parity cases through the latter."""
    _rerun({"GPSLAM_FUSE_B6": "0"}, ["test_gpu_parity.py", "test_gpu_upper.py"])


def test_round2_hierarchy_and_column_layout_kernels_still_agree_with_the_oracle():
    """GPSLAM_UPPER=0 (one launch per level of chunks of four) + GPSLAM_FWD_ROWS=0 (column-layout level 0): the fallbacks of
    chains with landmark columns, kept honest on the plain chains as well."""
    _rerun({"GPSLAM_UPPER": "0", "GPSLAM_FWD_ROWS": "0"}, ["test_gpu_parity.py"])
