import sys, math, torch
sys.path.insert(0, "/root/repo")
from nerf_atlas_amd import ops
dev = torch.device("cuda", 0)
side, T = 400, 128
size = 800
focal = 0.5 * size / math.tan(0.5 * 0.6911)
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
rays = ops.raygen(c2w, focal, size, (0, 0, side, side))
ts = ops.compute_ts(2.0, 6.0, T, dev)[0]
density = torch.randn(T, 1, side, side, device=dev)
rgb = torch.rand(T, 1, side, side, 3, device=dev)
def timed(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
N = side * side * T
for name, fn, b in (("composite (+alpha,weights)", lambda: ops.composite(density, rgb, ts, rays, softplus=True, bg="black"), 24),
                    ("composite (no aux)", lambda: ops.composite(density, rgb, ts, rays, softplus=True, bg="black", want_weights=False), 16)):
    us = timed(fn)
    print(f"{name:28s} {us:8.1f} us  {N * b / us / 1e6:6.2f} TB/s ({N * b / us / 1e6 / 8:5.1%} of 8 TB/s)")
