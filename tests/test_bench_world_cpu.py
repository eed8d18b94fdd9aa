"""bench.py's world > 1 branch, end to end on CPU: launched as the task's contract says the driver launches
it (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N) AND bare (python bench.py --gpus N: the script
starts its own ranks), with the two phases of every rank played by
tests/segment_model.py over gloo (GPSLAM_BENCH_BACKEND=segment_model).  Checked: the one JSON line appears on rank 0 only,
carries ranks / collective / value, converges like the unsharded oracle -- and a mismatch between --gpus and the world size
is refused."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, GPSLAM_BENCH_BACKEND="segment_model", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _launch(world, gpus, states=20, steps=2, warmup=1, timeout=600, extra=("--strong-states", "0")):
    port = 29600 + (os.getpid() % 1500) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", str(steps), "--warmup", str(warmup),
           "--states", str(states)] + list(extra)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=_env(), cwd=ROOT)


def _line(out):
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # rank 0 prints ONE line, the other ranks none
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 4, 8])      # 8: the node the driver's scaling run uses
def test_bench_line_of_a_multi_rank_run(world):
    d = _line(_launch(world, world))
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


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no torchrun in front and no WORLD_SIZE in the environment re-executes
    itself under torch.distributed.run and prints the one line; the multi-rank line carries the strong-scaling block (ONE
    chain of --strong-states states cut two ways) beside the weak headline."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--states", "20",
           "--strong-states", "60"]
    d = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT))
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["scaling"] == "weak" and d["config"]["total_states"] == 40
    ss = d["strong_scaling"]
    assert ss["scaling"] == "strong" and ss["total_states"] == 60 and ss["n_gpus"] == 2 and ss["states_per_gpu"] == 30
    assert ss["iters_to_convergence"] <= 8 and ss["delta_inf_at_convergence"] < 1e-6
    assert ss["ms_per_iteration"] > 0 and ss["seconds_to_convergence"] > 0
    assert abs(ss["state_iterations_per_sec"] - 60 * ss["steps"] / (ss["ms_per_iteration"] * 1e-3 * ss["steps"])) <= 1e-6 * ss["state_iterations_per_sec"]


def test_total_states_makes_the_headline_a_strong_scaling_line():
    """--total-states T: ONE chain of T states cut `world` ways is the headline itself ("scaling": "strong"); an odd T leaves
    segments that differ by one state."""
    d = _line(_launch(2, 2, extra=("--total-states", "41")))
    assert d["scaling"] == "strong" and d["config"]["total_states"] == 41 and d["config"]["states_per_gpu"] == 21
    assert d["strong_scaling"] is None
    assert d["iters_to_convergence"] <= 8 and d["delta_inf_at_convergence"] < 1e-6
    assert abs(d["value"] - 41 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
