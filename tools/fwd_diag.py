#!/usr/bin/env python3
"""lsfw (packed entry) against the streaming kernels (unpacked entry) on the shapes of a training recipe; where do they differ?
    python tools/fwd_diag.py [N=55296]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 55296
for (in0, in1, out, act, bias) in [(38, 0, 256, "none", True), (256, 0, 256, "leaky_relu", True), (256, 38, 256, "leaky_relu", True), (256, 0, 19, "leaky_relu", True),
                                   (256, 0, 65, "leaky_relu", True), (69, 0, 256, "none", True), (256, 0, 256, "sin", True), (256, 69, 256, "sin", True), (256, 0, 3, "sin", True)]:
    torch.manual_seed(in0 + in1 + out)
    x0 = torch.randn(N, in0, device="cuda"); x1 = torch.randn(N, in1, device="cuda") if in1 else None
    W = torch.randn(out, in0 + in1, device="cuda") / (in0 + in1) ** 0.5
    b = torch.randn(out, device="cuda") if bias else None
    (pf,) = ops.train_pack_many([(W, False)])
    y = ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=True, packed=pf)
    y2 = ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=True)
    bad = (y - y2).abs() > 1e-4
    print(f"in {in0}+{in1} out {out} {act}: max diff {float((y - y2).abs().max()):.3e}, wrong elements {int(bad.sum())}", end="")
    if bad.any():
        r, c = bad.nonzero(as_tuple=True)
        print(f"  rows {int(r.min())}..{int(r.max())} (stages {int(r.min()) // 32}..{int(r.max()) // 32}), cols {sorted(set(c.tolist()))[:8]}", end="")
    print()
