import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import ops
torch.manual_seed(0)
for N, in0, out in ((4096, 38, 256), (4096, 40, 256), (4096, 38, 65)):
    x0 = torch.randn(N, in0, device="cuda"); W = torch.randn(out, in0, device="cuda") * 0.1; gy = torch.randn(N, out, device="cuda")
    ref = (gy.double() @ W.double()) * torch.where(x0 > 0, 1.0, 0.01).double()
    g0, _ = ops.linear_dgrad(gy, W, x0, "leaky_relu")
    err = (g0.double() - ref).abs()
    print(N, in0, out, "max err", float(err.max()))
    bad = (err > 1e-3).nonzero()
    print(" bad count", bad.shape[0], "of", err.numel(), "; first", bad[:8].tolist(), "; bad cols", sorted(set(bad[:, 1].tolist()))[:40], "; bad rows mod 64", sorted(set((bad[:, 0] % 64).tolist()))[:70])
    r = err > 1e-3
    if bad.shape[0]:
        i, j = bad[0].tolist()
        print(" sample", g0[i, j].item(), ref[i, j].item(), "ratio", g0[i, j].item() / ref[i, j].item())
N, in0, out = 4096, 38, 256
x0 = torch.randn(N, in0, device="cuda"); W = torch.randn(out, in0, device="cuda") * 0.1; gy = torch.randn(N, out, device="cuda")
u = (gy.double() @ W.double())
d = torch.where(x0 > 0, 1.0, 0.01).double()
ref = u * d
g0, _ = ops.linear_dgrad(gy, W, x0, "leaky_relu")
print("got ", [round(v, 4) for v in g0[0, :12].tolist()])
print("ref ", [round(v, 4) for v in ref[0, :12].tolist()])
print("u   ", [round(v, 4) for v in u[0, :12].tolist()])
print("d   ", d[0, :12].tolist())
ratio = (g0.double() / u)[0, :12]
print("got/u", [round(v, 4) for v in ratio.tolist()])
gn, _ = ops.linear_dgrad(gy, W, x0, "none")
print("none err", float((gn.double() - u).abs().max()))
