"""probe: the one-launch PlainNeRF renderer (bf16x3 / f16x) at the TRAINING step's size (64 x 64 rays x 64 steps, and 128 x 128)"""
import math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nerf_atlas_amd.nerf as nerf
from nerf_atlas_amd import ops, config
dev = torch.device("cuda", 0)
torch.manual_seed(0)
size = 800
focal = 0.5 * size / math.tan(0.5 * 0.6911)
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
for crop in (64, 128):
    m = nerf.PlainNeRF(steps=64, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").to(dev).eval()
    rays = ops.raygen(c2w, focal, size, ((size - crop) // 2, (size - crop) // 2, crop, crop))
    ts, _ = ops.compute_ts(2.0, 6.0, 64, "cuda")
    for p in ("bf16x3", "f16x"):
        for aw in (False, True):
            config.set_precision(p)
            with torch.no_grad():
                for _ in range(3): out = m._render_fused(rays, ts, aw)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20): out = m._render_fused(rays, ts, aw)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            print(f"crop {crop} {p} alpha/weights {aw}: {dt * 1e3:.3f} ms per call, {crop * crop * 64 / dt / 1e6:.1f} Msamples/s", flush=True)
