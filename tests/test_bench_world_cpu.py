"""bench.py's world > 1 branch, end to end on CPU: launched exactly as the driver launches it
(python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N), with the two phases of every rank played by
tests/segment_model.py over gloo (GPSLAM_BENCH_BACKEND=segment_model).  Checked: the one JSON line appears on rank 0 only,
carries ranks / collective / value, converges like the unsharded oracle -- and a mismatch between --gpus and the world size
is refused."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, gpus, states=20, steps=2, warmup=1, timeout=600):
    env = dict(os.environ, GPSLAM_BENCH_BACKEND="segment_model", OMP_NUM_THREADS="1")
    port = 29600 + (os.getpid() % 1500) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", str(steps), "--warmup", str(warmup),
           "--states", str(states)]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.parametrize("world", [2, 4, 8])      # 8: the node the driver's scaling run uses
def test_bench_line_of_a_multi_rank_run(world):
    out = _launch(world, world)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # rank 0 prints ONE line, the other ranks none
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["ranks"] == world and d["steps"] == 2 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["unit"] == "state-iterations/s"
    assert d["config"]["total_states"] == 20 * world and d["config"]["states_per_gpu"] == 20
    assert d["value"] > 0 and abs(d["value"] - d["config"]["total_states"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
    assert d["collective"]["bytes_per_rank"] == (2 * 144 + 12 + 144 + 12) * 8 and d["collective"]["ms_per_iteration"] > 0
    # the sharded iteration converges as the chain does on one rank (tests/test_sharded_cpu.py holds it to the oracle)
    assert d["iters_to_convergence"] <= 8 and d["delta_inf_at_convergence"] < 1e-6
    assert d["data"] == "model" and d["roofline"] is None      # nothing here is a measurement


def test_world_size_must_equal_gpus():
    out = _launch(2, 4)
    assert out.returncode != 0
    assert "AssertionError" in out.stderr or "assert" in out.stderr
