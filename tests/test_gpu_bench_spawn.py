"""`python bench.py --gpus N` with NO torchrun environment must start its N ranks itself: run it as the
driver would, with two ranks sharing the one GPU of the box (NA_DIST_BACKEND=gloo), and compare with the N = 1 run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(gpus, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-other-configs", "--no-traffic"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks():
    one = _bench(1)
    two = _bench(2, {"NA_DIST_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert len(two["per_rank_kernel_ms"]) == 2 and all(x > 0 for x in two["per_rank_kernel_ms"])
    assert two["config"]["frame_checksum"] == one["config"]["frame_checksum"]  # the gathered frame is the N = 1 frame
    assert two["roofline"]["traffic"] is None and two["roofline"]["traffic_source"] is None
    # the N = 1 line names the profile its HBM bytes come from (or carries none): never an unlabelled constant
    assert (one["roofline"]["traffic"] is None) == (one["roofline"]["traffic_source"] is None)
    assert one["roofline"]["traffic_source"] is None or one["roofline"]["traffic_source"].startswith("profiles/")
    assert two["scaling"] == "strong" and two["gather_ms"] >= 0
    # the process group's first collective counted the ranks on the backend the line names
    assert (one["backend"], one["rccl_ranks"], one["gather_impl"]) == ("none", 1, "none")
    assert (two["backend"], two["rccl_ranks"], two["gather_impl"]) == ("gloo", 2, "gather")
    # N > 1 explains itself: one GPU's time for the whole frame in the same run, and the efficiency of the split.  (Two ranks on
    # ONE GPU time-share it, so the number itself only has to be sane here: ~0.5 when the launches serialise, up to 1.)
    assert "scaling_efficiency" not in one
    assert two["single_gpu_full_frame_kernel_ms"] > 0 and 0.2 < two["scaling_efficiency"] <= 1.1
