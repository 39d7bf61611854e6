#!/usr/bin/env python3
"""Host time of every nerf_atlas_amd.ops call inside a training step (forward AND backward: the autograd worker thread is
invisible to cProfile), plus the C-ABI calls underneath.    python tools/train_host_time.py"""
import sys, time, math, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_atlas_amd.nerf as nerf
from nerf_atlas_amd import ops, _lib
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(mod, name, tag):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter()
        try: return f(*a, **k)
        finally:
            e = acc[tag + name]; e[0] += 1; e[1] += time.perf_counter() - t
    setattr(mod, name, g)
for n in dir(ops):
    if callable(getattr(ops, n)) and not n.startswith("_") and getattr(getattr(ops, n), "__module__", "") == ops.__name__:
        wrap(ops, n, "ops.")
lib = _lib.load()
class LibProxy:
    def __getattr__(self, n):
        f = getattr(lib, n)
        def g(*a):
            t = time.perf_counter()
            try: return f(*a)
            finally:
                e = acc["C." + n]; e[0] += 1; e[1] += time.perf_counter() - t
        return g
_lib.load = lambda: LibProxy()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
size = 800; focal = 0.5 * size / math.tan(0.5 * 0.6911)
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
m = nerf.PlainNeRF(steps=64, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").to(dev); m.eval()
target = torch.rand(1, 64, 64, 3, device=dev)
opt = torch.optim.Adam(m.parameters(), lr=2e-4)
tt = collections.defaultdict(float)
def step():
    t0 = time.perf_counter()
    rays = ops.raygen(c2w, focal, size, (368, 368, 64, 64))
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.mse_loss(m(rays), target)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    tt["forward"] += t1 - t0; tt["backward"] += t2 - t1; tt["adam"] += t3 - t2
for _ in range(5): step()
torch.cuda.synchronize(); acc.clear(); tt.clear()
N = 30
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host {1e3*(t1-t0)/N:.2f} ms/step, wall {1e3*(t2-t0)/N:.2f} ms/step;  " + ", ".join(f"{k} {1e3*v/N:.2f}" for k, v in tt.items()))
for k, (c, s) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:36]:
    print(f"  {k:44s} {c/N:6.1f} calls/step  {1e6*s/N:8.1f} us/step  {1e6*s/max(c,1):7.1f} us/call")
