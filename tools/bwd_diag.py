#!/usr/bin/env python3
"""Where does the fused backward differ from the two-launch path?  python tools/bwd_diag.py [N] [out] [act]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
out = int(sys.argv[2]) if len(sys.argv) > 2 else 256
act = sys.argv[3] if len(sys.argv) > 3 else "none"
torch.manual_seed(1)
x0 = torch.randn(N, 256, device="cuda"); W = torch.randn(out, 256, device="cuda") / 16; gy = torch.randn(N, out, device="cuda")
(pt,) = ops.train_pack_many([(W, True)])
g0, dW, db = ops.linear_bwd_fused(gy, x0, act, pt)
h0, _ = ops.linear_dgrad(gy, W, x0, act, packed_t=pt)
dW2, db2 = ops.linear_wgrad(x0, gy, act, split_bf16=True)
bad = (g0 - h0).abs() > 1e-3
print("g_x wrong elements:", int(bad.sum()), "of", bad.numel(), " dW maxdiff", float((dW - dW2).abs().max()), " db maxdiff", float((db - db2).abs().max()))
if bad.any():
    r, c = bad.nonzero(as_tuple=True)
    print("rows mod 32:", sorted(set((r % 32).tolist()))[:40])
    print("stage index (row // 32) first 20:", sorted(set((r // 32).tolist()))[:20], " count", len(set((r // 32).tolist())))
    print("cols // 4 mod 32:", sorted(set(((c // 4) % 32).tolist())))
    print("col halves:", sorted(set((c // 128).tolist())))
    i = int(r[0]); j = int(c[0])
    print("example", i, j, float(g0[i, j]), float(h0[i, j]), " same value elsewhere in row block? ", (h0[i - i % 32:i - i % 32 + 32, :] == g0[i, j]).nonzero()[:4].tolist())
