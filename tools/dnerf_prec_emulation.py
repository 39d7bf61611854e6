"""Which Linears of D-NeRF's deformation network carry the f16x error on the reference golden g9?  CPU emulation with the operand
formats of tools/prec_search.py (Scheme: products per layer, regex overrides), end to end through oracle.dynamic_nerf_spline.
Round 4: all f16x 1.19e-4 (the GPU kernel: 1.0-1.6e-4); init in f16 hi/lo 0.88e-4, init + skip layers 0.82e-4, out 0.97e-4, the whole
network in hi/lo 0.23e-4; canonical model alone in f16x 0.20e-4 -- no single layer dominates, the error is the network's.
    python tools/dnerf_prec_emulation.py"""
import sys, os, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tools')
import oracle as O, oracle.nerf_oracle as NO
from conftest import load_golden, golden_params
from prec_search import Scheme
import torch.nn.functional as F
h = load_golden("g9_dnerf_spline6"); p = golden_params(h)
rays, times = h["rays"], h["times"]
T = int(h["steps"])
def run(): return O.dynamic_nerf_spline(p, rays, times, float(h["near"]), float(h["far"]), T, 6, act="upshifted")
ref = run()
names = {id(v): k for k, v in p.items()}
X = "f16*f16+e2m3lo(f16)*e2m3+e2m3*e2m3lo(f16)"
H = "f16*f16+f16lo*f16+f16*f16lo"
specs = {"all f16x": X,
         "deform.init hi/lo": f"{X};delta_estim\\.init={H}",
         "deform.init+skips hi/lo": f"{X};delta_estim\\.(init|layers\\.0|layers\\.3)={H}",
         "deform.out hi/lo": f"{X};delta_estim\\.out={H}",
         "deform last two hi/lo": f"{X};delta_estim\\.(out|layers\\.4)={H}",
         "deform all hi/lo": f"{X};delta_estim={H}",
         "canonical exact, deform f16x": f"f32*f32;delta_estim={X}",
         "deform exact, canonical f16x": f"{X};delta_estim=f32*f32"}
real = F.linear
for k, spec in specs.items():
    sch = Scheme(spec, names)
    NO.F.linear = lambda x, w, b=None: sch.linear(x, w, b)
    try: out = run()
    finally: NO.F.linear = real
    print(f"{k:34s} RGB L-inf {float((out-ref).abs().max()):.3e}", flush=True)
