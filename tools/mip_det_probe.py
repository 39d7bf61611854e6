#!/usr/bin/env python3
"""Run-to-run reproducibility of the one-launch mip renderer (render_ls.hip MODEL 6): the same 96 x 800 band N times, mismatching
pixels against the first run.  NA_LIB_PATH selects an experiment library (tools/ls_variant.py build NAME -DNA_LS_MIP_ABLATE=..)."""
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import nerf_atlas_amd.nerf as nerf
    from nerf_atlas_amd import config, ops
    from nerf_atlas_amd.utils import CylinderGaussian
    from oracle.procedural import proc_param
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    SIZE, T = 800, 128
    m = nerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted", mip=CylinderGaussian()).cuda().eval()
    for k, v in m.state_dict().items():
        if k.endswith("primes") or v.numel() == 0:
            continue
        v.copy_(torch.from_numpy(proc_param(k, tuple(v.shape))))
    focal = 0.5 * SIZE / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1.0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]]).cuda()
    config.set_precision("f16x")
    with torch.no_grad():
        rays = ops.raygen(c2w, focal, SIZE, (300, 0, 96, SIZE))
        first = m(rays).clone()
        w0 = m.weights.clone()
        bad, worst, badw = 0, 0.0, 0
        for _ in range(n):
            o = m(rays)
            d = (o - first).abs()
            bad += int((d.amax(-1) > 0).sum())
            worst = max(worst, float(d.max()))
            badw += int(((m.weights - w0).abs() > 0).sum())
    print(f"{os.environ.get('NA_LIB_PATH', 'shipped')}: {n} repeats, {bad} mismatching pixels (worst {worst:.2e}), {badw} mismatching weights")


if __name__ == "__main__":
    main()
