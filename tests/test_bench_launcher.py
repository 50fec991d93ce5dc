"""bench.py's N-rank launcher: `--gpus N` must start N ranks (one process per GPU, the reference's
`torch.distributed.launch --nproc_per_node=N`: /root/reference/README.md:138, basicsr/utils/dist_util.py:11-30) or fail
loudly -- never print `n_gpus: 1` for a request of N.  On CPU the launcher is driven with `--dry-run --backend gloo`
(rendezvous, the product's GradSync plan over the real parameter inventory, barrier + max-over-ranks timing, JSON)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_two_gloo_ranks():
    r = _run(["--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["dry_run"] is True
    assert out["value"] is None and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 16


def test_strong_scaling_shards_the_global_batch():
    r = _run(["--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--scaling", "strong"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 8
    assert "batch 4/GPU" in out["config"]["workload"]
    r = _run(["--gpus", "3", "--dry-run", "--backend", "gloo", "--scaling", "strong"])
    assert r.returncode != 0 and "not divisible" in r.stderr


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--dry-run", "--backend", "gloo"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "must match" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_no_gpu_no_silent_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-container check")
    for n in ("1", "2"):
        r = _run(["--gpus", n])
        assert r.returncode != 0 and "needs a ROCm GPU" in r.stderr, (n, r.stderr[-500:])
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_more_ranks_than_devices_is_an_error():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "refusing to run fewer ranks" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
