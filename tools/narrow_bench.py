#!/usr/bin/env python3
"""Times the narrow-output training GEMMs (csrc/train_gemm.hip nrw: 256 -> 65 / 3 forward, 256 -> 38 / 69 input gradient) and, for
comparison, the same shapes through the unpacked entry points (the three-role kernel) and a plain read of the same tensor.
    [NA_LIB_PATH=gpurun_ablate/lib_var_X.so] python tools/narrow_bench.py [N=262144]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
torch.manual_seed(0)
dev = "cuda"


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = torch.randn(N, 256, device=dev)
gy = torch.randn(N, 256, device=dev)
print(f"lib {os.environ.get('NA_LIB_PATH', 'shipped')}  N = {N}")
print(f"  read of one [N,256] tensor (x.sum()): {timed(lambda: x.sum()):7.1f} us;  copy: {timed(lambda: x.clone()):7.1f} us")
for out, act in ((65, "leaky_relu"), (3, "sin"), (128, "none")):
    W = torch.randn(out, 256, device=dev) * 0.06
    b = torch.randn(out, device=dev)
    (pk,) = ops.train_pack_many([(W, False)])
    a = timed(lambda: ops.linear_f32(x, W, b, pre_act=act, split_bf16=True, packed=pk))
    c = timed(lambda: ops.linear_f32(x, W, b, pre_act=act, split_bf16=True))
    print(f"  forward 256 -> {out:3d} ({act:10s}): packed (row-stream kernel) {a:7.1f} us, unpacked (three-role kernel) {c:7.1f} us")
for inn, act in ((38, "none"), (69, "none"), (38, "leaky_relu"), (69, "sin")):
    W = torch.randn(256, inn, device=dev) * 0.1
    x0 = torch.randn(N, inn, device=dev)
    (pt,) = ops.train_pack_many([(W, True)])
    a = timed(lambda: ops.linear_dgrad(gy, W, x0, act, packed_t=pt))
    c = timed(lambda: ops.linear_dgrad(gy, W, x0, act))
    print(f"  input gradient 256 -> {inn:3d} ({act:10s}): packed {a:7.1f} us, unpacked {c:7.1f} us")
for inn, act in ((38, "leaky_relu"), (69, "sin")):
    W = torch.randn(256, 256 + inn, device=dev) * 0.06
    x0, x1 = torch.randn(N, 256, device=dev), torch.randn(N, inn, device=dev)
    (pt,) = ops.train_pack_many([(W, True)])
    a = timed(lambda: ops.linear_dgrad(gy, W, x0, act, x1=x1, packed_t=pt))
    c = timed(lambda: ops.linear_dgrad(gy, W, x0, act, x1=x1))
    print(f"  skip layer input gradient [256 | {inn}] ({act}): packed (plain layer + row-stream) {a:7.1f} us, unpacked (two passes) {c:7.1f} us")
