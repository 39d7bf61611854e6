#!/usr/bin/env python3
"""Where does the LS renderer differ from the register engine?  Error per pass index (8 blocks of 32 samples)."""
import math, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_golden, golden_params
from nerf_atlas_amd import ops
from test_gpu_render_ls import pack_ls
from test_gpu_render import pack_plain

h = load_golden("g11_plain_view_b1"); p = golden_params(h)
size, T = 800, 128
focal = 0.5 * size / math.tan(0.5 * 0.6911)
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]]).cuda()
rays = ops.raygen(c2w, focal, size, (380, 390, 37, 41))
ts, _ = ops.compute_ts(2.0, 6.0, T, "cuda")
for prec in ("bf16x3",):
    packed, tables = pack_ls(ops, p, prec)
    a, al, wa = ops.render_plain_view_ls(rays, ts, tables, packed, prec, "upshifted", "black", want_weights=True)
    pf, pv, _ = pack_plain(ops, p, prec)
    b, bl, wb = ops.render_plain_view(rays, ts, tables, pf, pv, prec, "upshifted", "black", want_weights=True)
    err = (a - b).abs().amax(-1).reshape(-1)          # per ray
    R = err.numel(); nb = (T + 31) // 32
    print(prec, "max err", float(err.max()), "rays wrong", int((err > 1e-5).sum()), "of", R)
    aerr = (al - bl).abs().reshape(T, -1)             # [T, R] alpha error per sample
    blk_err = aerr.reshape(nb, 32, R).amax(1)         # [nb, R]
    item_err = blk_err.t().reshape(-1)                # item = ray*nb + tb
    npass = (item_err.numel() + 7) // 8
    pad = torch.zeros(npass * 8, device=item_err.device); pad[:item_err.numel()] = item_err
    pe = pad.reshape(npass, 8)
    bad = (pe > 1e-5)
    print("passes with a wrong block:", int(bad.any(1).sum()), "of", npass)
    idx = bad.any(1).nonzero().flatten().tolist()
    print("first bad passes:", idx[:20], " pass % 256:", sorted(set(i % 256 for i in idx))[:20], " pass // 256:", sorted(set(i // 256 for i in idx)))
    print("bad block slots (0-3 group0, 4-7 group1):", bad.sum(0).tolist())
    print("alpha err max", float(aerr.max()))
