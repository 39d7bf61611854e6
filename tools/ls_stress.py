#!/usr/bin/env python3
"""Stress the layer-synchronous renderer for call-to-call interference (recycled workspaces / ray buffers): renders a
slab and a band of it with changing cameras and compares the band rows bit for bit, N times.
    python tools/ls_stress.py [N]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_atlas_amd import nerf, config, cameras

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
bad = 0
for prec in ("bf16x3", "bf16"):
    config.set_precision(prec)
    m = nerf.PlainNeRF(steps=128, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").cuda().eval()
    with torch.no_grad():
        for i in range(n):
            ang = 0.05 * i
            c2w = torch.tensor([[[math.cos(ang), 0, math.sin(ang), 0.3 * math.sin(ang)], [0, 1, 0, 0.01 * i],
                                 [-math.sin(ang), 0, math.cos(ang), 4.0]]])
            cam = cameras.NeRFCamera(cam_to_world=c2w, focal=0.5 * 800 / math.tan(0.5 * 0.6911)).cuda()
            rows = 24 + (i % 5) * 8
            slab = cam.sample_positions((300, 0, rows, 800), size=800)
            full = m(slab)
            r0 = 8 + (i % 3) * 4
            band = m(slab[:, r0:r0 + 8].contiguous())
            if not torch.equal(band, full[:, r0:r0 + 8]):
                bad += 1
                print(prec, "iteration", i, "band != full rows, max diff", float((band - full[:, r0:r0 + 8]).abs().max()))
print(f"{2 * n} comparisons, {bad} mismatches")
sys.exit(1 if bad else 0)
