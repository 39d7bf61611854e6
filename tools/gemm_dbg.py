import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import ops
torch.manual_seed(0)
for N, in0, out in ((8192, 256, 256), (32768, 256, 256), (262144, 256, 256)):
    x0 = torch.randn(N, in0, device="cuda"); W = torch.randn(out, in0, device="cuda") * 0.1; gy = torch.randn(N, out, device="cuda")
    u = (gy.double() @ W.double())
    d = torch.where(x0 > 0, 1.0, 0.01).double()
    ref = u * d
    g0, _ = ops.linear_dgrad(gy, W, x0, "leaky_relu")
    err = (g0.double() - ref).abs()
    print(N, in0, out, "max err", float(err.max()))
    bad = (err > 1e-3).nonzero()
    print(" bad count", bad.shape[0], "of", err.numel(), "; bad cols mod 64", sorted(set((bad[:, 1] % 64).tolist()))[:70], "; bad rows mod 64", sorted(set((bad[:, 0] % 64).tolist()))[:70])
    if bad.shape[0]:
        i, j = bad[0].tolist()
        print(" first bad", i, j, "got", g0[i, j].item(), "ref", ref[i, j].item(), "u", u[i, j].item(), "d", d[i, j].item())
        r = (g0.double() / u)
        print(" got/u at first bad row, cols j..j+8:", [round(v, 3) for v in r[i, j:j + 8].tolist()], " d:", d[i, j:j + 8].tolist())
        print(" bad tiles (row // 64) first 20:", sorted(set((bad[:, 0] // 64).tolist()))[:20], " count of bad tiles", len(set((bad[:, 0] // 64).tolist())))
    gn, _ = ops.linear_dgrad(gy, W, x0, "none")
    print(" act none: max err", float((gn.double() - u).abs().max()))
