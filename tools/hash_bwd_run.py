#!/usr/bin/env python3
"""Time na_hash_encode_backward of the shipped library and of experiment builds (tools/ls_variant.py build-unit backward.hip NAME
flags) on the training bench's sample positions: 64 x 64 rays x 64 samples of the 800^2 camera.  python tools/hash_bwd_run.py shipped [NAME...]"""
import ctypes as C
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from nerf_atlas_amd import _lib, ops

dev = torch.device("cuda", 0)
focal = 0.5 * bench.SIZE / math.tan(0.5 * bench.FOV)
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
rays = ops.raygen(c2w, focal, bench.SIZE, (368, 368, 64, 64))
ts, _ = ops.compute_ts(bench.NEAR, bench.FAR, int(os.environ.get("HB_STEPS", "64")), dev)
pts = ops.compute_pts(rays, ts).reshape(-1, 3).contiguous()
N = pts.shape[0]
torch.manual_seed(0)
g = torch.randn(N, 35, device=dev)
ref = None
for name in sys.argv[1:]:
    path = os.path.join(REPO, "nerf_atlas_amd", "libnerf_atlas_amd.so") if name == "shipped" else os.path.join(REPO, "gpurun_ablate", f"lib_var_{name}.so")
    lib = C.CDLL(path)
    fn = lib.na_hash_encode_backward
    fn.restype, fn.argtypes = _lib.SIGNATURES["na_hash_encode_backward"]
    st = torch.cuda.current_stream().cuda_stream
    for det in (0,):  # (fp32 atomics, the mode the training bench runs in)
        tg = torch.zeros(8, 65536, 4, device=dev)
        def f():
            assert fn(pts.data_ptr(), N, g.data_ptr(), 1, tg.data_ptr(), st) == 0
        f()
        torch.cuda.synchronize()
        out = tg.clone()
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 20 * 1e6
        if ref is None:
            ref = out
        err = float((out - ref).abs().max() / ref.abs().max())
        print(f"{name:10s} deterministic={det} N={N}: {us:7.1f} us   rel L-inf vs the first run {err:.2e}", flush=True)
