#!/usr/bin/env python3
"""VolSDF (mlp SDF + View) forward in a loop on a 200 x 200 x 128 slab, for rocprofv3 --kernel-trace --stats (tools/volsdf_case.py [bf16|bf16x3])."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import config, ops, nerf, sdf, refl
config.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
torch.manual_seed(0)
under = sdf.sdf_kinds["mlp"](intermediate_size=64)
r = refl.View(latent_size=64, act="upshifted", out_features=3)
m = nerf.VolSDF(sdf=sdf.SDF(under, r, isect=None, t_near=0.3, t_far=1.8), steps=128, t_near=0.3, t_far=1.8, sigmoid_kind="upshifted").cuda().eval()
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device="cuda")
rays = ops.raygen(c2w, 0.5 * 800 / math.tan(0.5 * 0.6911), 800, (300, 300, 200, 200))
with torch.no_grad():
    for _ in range(6): out = m(rays)
torch.cuda.synchronize()
print(float(out.sum()))
