"""Round-6 GPU tests (through the C ABI, against the oracle): the N > 1 bench protocol on RCCL (two real GPUs when the box
has them), the resumable Sokoban solver behind the ready mask, device-side target resampling at auto-reset."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pcgrl_oracle as po  # noqa: E402  (checker only)
from conftest import GOLDEN, ROOT  # noqa: E402

REW_TOL = 1e-6


def _vec(*a, **k):
    from control_pcgrl_amd import VecPcgrlEnv
    return VecPcgrlEnv(*a, **k)


def _bench(*argv, env=None, timeout=900):
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert lines and r.stdout.strip().splitlines()[-1] == lines[-1], "the JSON line is the LAST line of stdout:\n" + r.stdout[-2000:]
    return json.loads(lines[-1])


# ------------------------------------------------------------------------------------- N > 1 on RCCL (SURVEY 8 e)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the first real multi-rank RCCL run of the path")
def test_bench_two_gpus_on_rccl():
    """`python bench.py --gpus 2 --steps 20 --warmup 5` (the driver's command line at N = 2), one rank per GPU over RCCL:
    shard seeds, per-rank episode counts, the overlapped exchange, and rank 0's line as the LAST line of stdout."""
    out = _bench("--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--rollout-steps", "0", "--rllib-adapter", "0",
                 "--closed-loop-steps", "0", "--sub-batches", "")
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5
    assert out["config"]["envs_per_gpu"] == 4096 and out["config"]["global_envs"] == 8192
    assert out["config"]["seed_ranges"] == [[0x5EED, 0x5EED + 4095], [0x5EED + 4096, 0x5EED + 8191]]
    pr = out["per_rank"]
    assert pr["collective"].startswith("nccl all-gather") and "side stream" in pr["collective"]
    assert len(pr["env_steps_per_s"]) == 2 and pr["episodes"] == [0.0, 0.0]
    tr = out["timed_region"]
    assert tr["exchange"] == "overlap" and tr["protocol_efficiency_bound"] >= 0.95, tr
    # a run long enough for every env of both shards to finish one episode per interval
    out = _bench("--gpus", "2", "--envs", "256", "--steps", "800", "--warmup", "800", "--no-cpu-baseline", "--rollout-steps", "0",
                 "--rllib-adapter", "0", "--closed-loop-steps", "0", "--sub-batches", "")
    assert out["per_rank"]["episodes"] == [256.0, 256.0] and out["episodes"]["episodes"] == 512.0
    assert out["episodes"]["mean_length"] == 770.0
    assert out["timed_region"]["previous_interval_sums"][2] == 512.0
