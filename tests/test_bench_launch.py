"""`python bench.py --gpus N` from a PLAIN python command: with no WORLD_SIZE in the environment bench.py re-executes
itself under torch.distributed.run (one rank per GPU, 127.0.0.1) and rank 0 prints ONE JSON line.  Exercised on CPU through `--dry`
(gloo, a stand-in forward): launch, per-rank batches, the overlapped all-gather, barrier + max-over-ranks timing and the line itself
are bench.py's own code; only the HIP forward is replaced."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_plain_python_gpus2_self_launches_and_prints_one_line():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry"])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["dry"] is True
    assert out["scaling"] == "weak" and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 2 * out["config"]["per_gpu_batch"]
    rk = out["ranks"]
    assert rk["world_size"] == 2 and len(rk["per_rank_images_per_s"]) == 2 and rk["gathered_equals_single_process"] is True
    # value = the units ALL ranks processed / the slowest rank's time: never above the sum of the per-rank rates
    assert out["value"] <= sum(rk["per_rank_images_per_s"]) * 1.001
    assert abs(out["value"] - out["config"]["global_batch"] / (out["ms_per_step"] / 1e3)) <= 0.02 * out["value"]


def test_plain_python_gpus1_dry_needs_no_launcher():
    out = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--dry"])
    assert out["n_gpus"] == 1 and out["config"]["collective"] == "none" and out["ranks"]["world_size"] == 1
