"""Timing of the one-launch training forward alone (ops.train_plain_view_ls, 262 144 and 1 048 576 samples); NA_LIB_PATH picks a variant
library built by tools/ls_variant.py (e.g. `build tr_nostore --prec bf16x3 -DNA_LS_TRAIN_EXP=1`): the ablation table of DESIGN 3d."""
import math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_atlas_amd.nerf as nerf
from nerf_atlas_amd import ops
from nerf_atlas_amd.nerf import compute_pts_ts
dev = torch.device("cuda", 0)
torch.manual_seed(0)
size = 800
focal = 0.5 * size / math.tan(0.5 * 0.6911)
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
for crop in (64, 128):
    m = nerf.PlainNeRF(steps=64, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").to(dev).eval()
    rays = ops.raygen(c2w, focal, size, ((size - crop) // 2, (size - crop) // 2, crop, crop))
    pts, ts, r_o, r_d, _ = compute_pts_ts(rays, 2.0, 6.0, 64, perturb=0)
    packed = m.packed_ls("bf16x3")
    with torch.no_grad():
        for _ in range(3): o = ops.train_plain_view_ls(rays.reshape(-1, 6), ts, pts, m.first.enc.tables(), packed, "upshifted")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): o = ops.train_plain_view_ls(rays.reshape(-1, 6), ts, pts, m.first.enc.tables(), packed, "upshifted")
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{os.environ.get('NA_LIB_PATH', 'shipped').split('/')[-1]} crop {crop}: {dt * 1e3:.3f} ms per call", flush=True)
