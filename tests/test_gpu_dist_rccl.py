"""The "nccl" (= RCCL) branch of the data-parallel glue on ONE GPU ("nothing in the tree has
ever executed the nccl branch").  A gpurun box has a single MI355X, and RCCL refuses two ranks on one device, so the
self-test builds a ONE-rank process group (CMGAN_DIST_FORCE_INIT=1): communicator set-up over the box's fabric
settings (HSA_ENABLE_IPC_MODE_LEGACY=0), the per-step all-reduce of the loss scalars, the barrier bracket of the timed
region, the per-rank gather, and - in the training leg - the rank-0 broadcasts, the flat-bucket gradient all-reduces
and the MIN all-reduce all run through RCCL on device tensors.  The multi-rank arithmetic is covered by the 2-rank gloo
tests (tests/test_dist_gloo.py, tests/test_bench_launcher.py); the scaling curve is the driver's to measure."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_runs_its_collectives_through_rccl_on_a_one_rank_group():
    env = dict(os.environ)
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               CMGAN_DIST_FORCE_INIT="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-f32", "--no-f16x1", "--no-extra", "--train-batch", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])            # the JSON line is the LAST line of stdout (after RCCL's banner)
    assert line["n_gpus"] == 1 and line["collective"]["backend"].startswith("nccl"), line["collective"]
    assert line["collective"]["ranks_seen"] == 1
    assert line["value"] > 0 and len(line["per_rank_frames_per_s"]) == 1
    ts = line["train_step"]
    assert "error" not in ts, ts
    assert ts["ms_per_step"] > 0 and all(map(lambda v: v == v, (ts["loss"], ts["disc_loss"])))      # finite, not NaN
