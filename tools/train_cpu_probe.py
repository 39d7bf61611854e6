import sys, time, math, torch
sys.path.insert(0, '/root/repo')
import nerf_atlas_amd.nerf as nerf
from nerf_atlas_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
size = 800; focal = 0.5 * size / math.tan(0.5 * 0.6911)
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
m = nerf.PlainNeRF(steps=64, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").to(dev); m.eval()
target = torch.rand(1, 64, 64, 3, device=dev)
for fused in (False, True):
    opt = torch.optim.Adam(m.parameters(), lr=2e-4, fused=fused)
    def step():
        rays = ops.raygen(c2w, focal, size, (368, 368, 64, 64))
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(m(rays), target)
        loss.backward(); opt.step(); return loss
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"fused_adam={fused}: CPU dispatch {1e3*(t1-t0)/30:.2f} ms/step, wall {1e3*(t2-t0)/30:.2f} ms/step")
    t0 = time.perf_counter()
    for _ in range(30): float(step().detach())
    torch.cuda.synchronize(); print(f"   with a sync per step: {1e3*(time.perf_counter()-t0)/30:.2f} ms/step")
