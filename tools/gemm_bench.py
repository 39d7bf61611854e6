import torch, time, sys
sys.path.insert(0, '/root/repo')
from nerf_atlas_amd import ops
N=262144
x=torch.randn(N,256,device='cuda'); W=torch.randn(256,256,device='cuda')*0.06; b=torch.zeros(256,device='cuda'); gy=torch.randn(N,256,device='cuda')
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
for act in ("none","sin"):
    print(act, "fwd %.0f us" % t(lambda: ops.linear_f32(x,W,b,pre_act=act,split_bf16=True)),
          "dgrad %.0f us" % t(lambda: ops.linear_dgrad(gy,W,x,act)),
          "wgrad %.0f us" % t(lambda: ops.linear_wgrad(x,gy,act,split_bf16=True)))
print("copy 268MB: %.0f us" % t(lambda: x.clone()))
