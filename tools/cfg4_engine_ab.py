#!/usr/bin/env python3
"""Same-call A/B of config 4's deformation engines (round 6): bench.other_configs rows `4` and `4-plv` with
config.deformation_engine = "ls-bf16x3" (one bf16x3 launch of the layer-synchronous engine, the default) against "generic" (the
register engine), twice.      python tools/cfg4_engine_ab.py        (GPU box)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_atlas_amd import config
dev = torch.device("cuda", 0)
orig = config.set_deformation_engine
for rep in range(2):
    for eng in ("ls-bf16x3", "generic"):
        # other_configs sets the engine itself: patch the default it restores / selects
        def patched(e, eng=eng):
            orig(eng if e == "ls-bf16x3" else e)
        config.set_deformation_engine = patched
        rows, secs, outside = bench.other_configs(dev, precisions=("f16x", "bf16x3"), iters=5, only={"4", "4-plv"})
        config.set_deformation_engine = orig
        for r in rows:
            print(eng, r["config"][:12], r["dtype"], r["Msamples_s"], r["frac"], flush=True)
