#!/usr/bin/env python3
"""Run GPU tests in a process whose caching allocator hands out blocks full of a chosen bit pattern: a kernel that reads memory it
never wrote (an output row it skips, an uninitialised workspace) gives different answers under different patterns, where a fresh
process (zero pages) hides it.  Used in round 4 to rule out uninitialised reads behind a failure that only appeared when the
whole suite ran in one process (the cause was the f16x range guard's per-schedule launch ids).

    python tools/garbage_run.py nan|1e30|-1e4 [pytest args ...]      (GPU box; default: the VolSDF tiled-frame tests)
"""
import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    pattern = float(sys.argv[1]) if len(sys.argv) > 1 else float("nan")
    blocks = [torch.full((1 << 28,), pattern, device="cuda") for _ in range(24)]  # 24 GiB of the pattern
    torch.cuda.synchronize()
    del blocks
    args = sys.argv[2:] or [os.path.join(REPO, "tests", "test_gpu_wholeframe.py"), "-q", "-m", "gpu", "-k", "volsdf and mlp", "-x"]
    return pytest.main(args)


if __name__ == "__main__":
    sys.exit(main())
