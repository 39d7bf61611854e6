#!/usr/bin/env python3
"""Run-to-run identity of lsfw on a recipe's shapes (a race shows as a differing run).  python tools/fwd_repeat.py [N] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 55296
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for (in0, in1, out, act) in [(38, 0, 256, "none"), (256, 0, 256, "leaky_relu"), (256, 38, 256, "leaky_relu"), (256, 0, 19, "leaky_relu"),
                             (256, 0, 65, "leaky_relu"), (69, 0, 256, "none"), (256, 0, 256, "sin"), (256, 69, 256, "sin"), (256, 0, 3, "sin")]:
    torch.manual_seed(in0 + in1 + out)
    x0 = torch.randn(N, in0, device="cuda"); x1 = torch.randn(N, in1, device="cuda") if in1 else None
    W = torch.randn(out, in0 + in1, device="cuda") / (in0 + in1) ** 0.5
    b = torch.randn(out, device="cuda")
    (pf,) = ops.train_pack_many([(W, False)])
    ref = ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=True)   # streaming kernel
    ndiff = 0; worst = 0.0
    junk = [torch.randn(N, 256, device="cuda") for _ in range(3)]
    for r in range(reps):
        y = ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=True, packed=pf)
        z = junk[r % 3] * 1.0001  # other traffic in between
        d = float((y - ref).abs().max())
        if d != 0.0: ndiff += 1; worst = max(worst, d)
    print(f"in {in0}+{in1} out {out} {act}: {ndiff} of {reps} runs differ from the streaming kernel, worst {worst:.3e}")
