#!/usr/bin/env python3
"""Outputs of the one-pass backward (csrc/train_bwd.hip) and the register-resident forward (csrc/train_fwd.hip) on fixed inputs,
saved for bit comparison across builds (tools/sched_fuzz.py).    python tools/train_fused_dump.py OUT.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import ops
out = {}
for i, (N, in1, o, act) in enumerate([(8192 + 37, 0, 256, "leaky_relu"), (12000, 38, 256, "sin"), (9000, 0, 65, "leaky_relu"), (8192, 69, 256, "none")]):
    torch.manual_seed(i)
    x0 = torch.randn(N, 256, device="cuda"); x1 = torch.randn(N, in1, device="cuda") if in1 else None
    W = torch.randn(o, 256 + in1, device="cuda") / 16; b = torch.randn(o, device="cuda"); gy = torch.randn(N, o, device="cuda")
    pf, pt = ops.train_pack_many([(W, False), (W, True)])
    g0, dW, db = ops.linear_bwd_fused(gy, x0, act, pt, in1=in1)
    res = [g0, dW[:, :256].contiguous(), db]
    if in1:
        g1, _, _ = ops.linear_bwd_fused(gy, x1, act, pt, dW=dW, col0=256)
        res += [g1, dW.clone()]
    if o == 256:
        res.append(ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=True, packed=pf))
    out[f"case{i}"] = [t.cpu() for t in res]
torch.save(out, sys.argv[1])
