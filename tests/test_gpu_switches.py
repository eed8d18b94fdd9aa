"""gpslam_hip_config.reserved[6] (GPSLAM_PLAN_*) selects whole kernel families per handle: the parity suites are re-run in child
processes whose ChainSolver mirror ORs the bits into every handle it creates (GPSLAM_PY_DEFAULT_PLAN, gpslam_amd/chain.py -- the
library itself reads no environment), so that the path a default run does NOT take at the test sizes stays covered."""
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
    (api_impl.inc: fused_kernel_applies); GPSLAM_PLAN_UNFUSED_LEVEL0 sends the small parity cases through the latter -- where the GP
    priors of the d = 3 manifolds travel as structured records (round 4)."""
    _rerun({"GPSLAM_PY_DEFAULT_PLAN": "1"}, ["test_gpu_parity.py", "test_gpu_upper.py"])


def test_round2_hierarchy_and_column_layout_kernels_still_agree_with_the_oracle():
    """GPSLAM_PLAN_LEVELS_OF_FOUR (one launch per level of chunks of four) + GPSLAM_PLAN_COLUMN_LEVEL0 (column-layout level 0):
    the fallbacks of chains with landmark columns, kept honest on the plain chains as well."""
    _rerun({"GPSLAM_PY_DEFAULT_PLAN": str(2 + 4)}, ["test_gpu_parity.py"])


def test_plain_row_path_of_every_chain_still_agrees_with_the_oracle():
    """GPSLAM_PLAN_GP_ROWS: the GP priors as plain Jacobian rows (what chains with several distinct Qc, world-frame velocities or
    fp32 rows run), with and without the fused level-0 kernel."""
    _rerun({"GPSLAM_PY_DEFAULT_PLAN": "16"}, ["test_gpu_parity.py"])
    _rerun({"GPSLAM_PY_DEFAULT_PLAN": str(16 + 1)}, ["test_gpu_parity.py"], k="pose2 or rot3 or linear3 or lock_step")


def test_plan_bits_are_validated():
    import gpslam_amd
    with pytest.raises(gpslam_amd.GpslamHipError):
        gpslam_amd.ChainSolver(gpslam_amd.POSE3, plan=1 << 9)
