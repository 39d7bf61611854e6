import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import ops
N = 262144
x = torch.randn(N, 256, device="cuda"); W = torch.randn(256, 256, device="cuda") * 0.06; b = torch.zeros(256, device="cuda"); gy = torch.randn(N, 256, device="cuda")
for act in ("leaky_relu", "sin"):
    for _ in range(5):
        ops.linear_f32(x, W, b, pre_act=act, split_bf16=True)
        ops.linear_dgrad(gy, W, x, act)
        ops.linear_wgrad(x, gy, act, split_bf16=True)
torch.cuda.synchronize()
