#!/usr/bin/env python3
"""NA_PREC_F16X on the GPU box: L-inf of the layer-synchronous PlainNeRF(view) renderer against the CPU oracle (golden
procedural weights and the bench's random-init weights), next to bf16x3 / f16, and the full-frame rate of each mode.

    python tools/f16x_check.py [--time]
"""
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import oracle as O
    from conftest import load_golden, golden_params
    from test_gpu_render_ls import pack_ls
    from nerf_atlas_amd import ops
    import bench
    size, T = 800, 128
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]])
    ts, _ = ops.compute_ts(2.0, 6.0, T, "cuda")
    h = load_golden("g11_plain_view_b1")
    pg = golden_params(h)
    model = bench.build_model(torch.device("cuda", 0))
    pb = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for name, p in (("golden weights", pg), ("bench weights", pb)):
        crop = (380, 390, 40, 40)
        rays = ops.raygen(c2w.cuda(), focal, size, crop)
        aux = {}
        ref = O.plain_nerf(p, rays.cpu(), 2.0, 6.0, T, "view", act="upshifted", aux=aux)
        for prec in ("bf16x3", "f16x", "f16", "bf16"):
            packed, tables = pack_ls(ops, p, prec)
            out, alpha, w = ops.render_plain_view_ls(rays, ts, tables, packed, prec, "upshifted", "black", want_weights=True)
            torch.cuda.synchronize()
            print(f"{name:15s} {prec:7s} L-inf rgb {float((out.cpu() - ref).abs().max()):.3e}  alpha {float((alpha.cpu() - aux['alpha']).abs().max()):.3e}"
                  f"  weights {float((w.cpu() - aux['weights']).abs().max()):.3e}  finite {bool(torch.isfinite(out).all())}", flush=True)
    if "--time" in sys.argv:
        rays = ops.raygen(c2w.cuda(), focal, size, (0, 0, size, size))
        for prec in ("f16x", "bf16x3", "f16", "bf16", "f16x"):
            packed, tables = pack_ls(ops, pb, prec)
            f = lambda: ops.render_plain_view_ls(rays, ts, tables, packed, prec, "upshifted", "black")
            f(); f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            n = size * size * T
            print(f"full frame {prec:7s} {ms:8.2f} ms  {n / ms / 1e3:8.1f} Msamples/s  {n * 1192960 / (ms * 1e-3) / 2.5e15:6.1%} of the bf16 MFMA peak", flush=True)


if __name__ == "__main__":
    main()
