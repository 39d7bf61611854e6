#!/usr/bin/env python3
"""The training step of tools/train_bench.py eager and as ONE captured HIP graph (torch.cuda.graph; Adam capturable=True; the
training GEMMs' hipMallocAsync / hipFreeAsync are captured as they are).  DESIGN.md 9a: at 65 536 samples the replay takes 3.94 ms
against 3.8-5.0 ms eager -- a graph node costs what its launch cost.      python tools/train_graph_try.py [steps_per_ray=16]"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_atlas_amd.nerf as nerf
from nerf_atlas_amd import ops


def main():
    dev = torch.device("cuda", 0)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    size = 800
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    torch.manual_seed(0)
    m = nerf.PlainNeRF(steps=steps, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").to(dev)
    m.eval()
    opt = torch.optim.Adam(m.parameters(), lr=2e-4, capturable=True)
    target = torch.rand(1, 64, 64, 3, device=dev)
    loss_buf = torch.zeros((), device=dev)

    def step():
        rays = ops.raygen(c2w, focal, size, (368, 368, 64, 64))
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(m(rays), target)
        loss.backward()
        opt.step()
        loss_buf.copy_(loss.detach())

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    print(f"eager  {64 * 64 * steps} samples: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step, loss {float(loss_buf):.6f}")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f"graph  {64 * 64 * steps} samples: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step, loss {float(loss_buf):.6f}")


if __name__ == "__main__":
    main()
