"""bench.py --gpus N must start N ranks by itself (the reference spawns one process per GPU, src/train.py:294-297)
and must never mislabel a run: a world size that differs from --gpus, or fewer GPUs than requested, is an error.
The launcher / rendezvous / barrier / max-over-ranks / all-reduce path runs here with 2 gloo ranks on a CPU stub
step (`--stub-cpu`); the GPU leg is the same code with backend nccl (= RCCL) and the HIP forward."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, env=env, timeout=600)


def test_gpus_2_launches_two_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["stub"] is True
    # the training leg's collective shape (a flat gradient bucket averaged over the ranks) ran on both ranks
    assert d["train_step"]["stub"] is True and d["train_step"]["grad_mean_ok"] is True
    assert d["collective"]["ranks_seen"] == 2 and d["collective"]["backend"] == "gloo"
    assert len(d["per_rank_frames_per_s"]) == 2
    assert d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    # whole-job value over the max-over-ranks time can never beat the sum of the per-rank rates
    assert 0 < d["value"] <= sum(d["per_rank_frames_per_s"]) * 1.0001


def test_single_process_stub_line():
    r = _run(["--steps", "2", "--warmup", "0", "--stub-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["collective"]["ranks_seen"] == 1


def test_world_size_mismatch_is_a_hard_error():
    r = _run(["--gpus", "4", "--stub-cpu"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "refusing to mislabel" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_more_gpus_than_visible_fails_loudly_instead_of_falling_back():
    import torch
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode == 2 and "refusing to fall back" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
