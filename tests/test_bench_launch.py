"""bench.py launches its own ranks: `python bench.py --gpus N` without torchrun must not quietly run on one GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_gt_visible_devices_fails_loudly():
    import torch
    nd = torch.cuda.device_count()
    r = _run(["--gpus", str(nd + 1 if nd else 2), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {})
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stderr + r.stdout)


def test_world_size_mismatch_fails():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_self_launch_two_ranks_on_one_gpu_reports_breakdown():
    """The driver's N > 1 command without a launcher: bench.py spawns the ranks itself (gloo rehearsal of the
    multi-rank path on one GPU) and rank 0's line carries every rank's panel / update / broadcast-wait times."""
    r = _run(["--gpus", "2", "--n-obs", "2200", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
             {"CK_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and len(out["per_rank"]) == 2
    for pr in out["per_rank"]:
        assert pr["update_ms"] > 0 and pr["panel_ms"] > 0 and pr["bcast_wait_ms"] >= 0
    assert out["comm"]["panels"] == 9          # N = 4 400 -> 9 panels of 512: look-ahead over >= 8 panels
    assert out["value"] > 0
    # the N > 1 line is complete (VERDICT r03 missing #1b): roofline with its traffic source, stage block with rank 0's
    # covariance assembly against the HBM peak
    assert out["roofline"]["frac"] > 0 and out["roofline"]["traffic_source"]
    ca = out["stages"]["cov_assembly"]
    assert ca["K1_sigma_rank0"]["GBs"] > 0 and ca["K2_c0_rank0"]["GBs"] > 0 and out["stages"]["rank0"]["update_ms"] > 0
