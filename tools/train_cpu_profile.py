#!/usr/bin/env python3
"""Where does the HOST time of a training step go?  cProfile over 30 steps of tools/train_bench.py's loop (no per-step sync).
    python tools/train_cpu_profile.py [fused_adam=0|1]"""
import cProfile, pstats, sys, time, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_atlas_amd.nerf as nerf
from nerf_atlas_amd import ops
fused = len(sys.argv) > 1 and sys.argv[1] == "1"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
size = 800; focal = 0.5 * size / math.tan(0.5 * 0.6911)
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
m = nerf.PlainNeRF(steps=64, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").to(dev); m.eval()
target = torch.rand(1, 64, 64, 3, device=dev)
opt = torch.optim.Adam(m.parameters(), lr=2e-4, fused=fused)
def step():
    rays = ops.raygen(c2w, focal, size, (368, 368, 64, 64))
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.mse_loss(m(rays), target)
    loss.backward(); opt.step(); return loss
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(30): step()
t1 = time.perf_counter()
pr.disable(); torch.cuda.synchronize()
print(f"CPU dispatch {1e3*(t1-t0)/30:.2f} ms/step (under cProfile)")
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
