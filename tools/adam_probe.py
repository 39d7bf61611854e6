#!/usr/bin/env python3
"""Which float operations does torch's foreach Adam perform per element (round 6: na_adam_step reproduces them bit for bit)?
Each ATen op of torch/optim/adam.py::_multi_tensor_adam against candidate formulas evaluated in float64 and rounded once or twice.
    python tools/adam_probe.py          (GPU box)"""
import torch

torch.manual_seed(0)
N = 1 << 20
dev = "cuda"
f32 = lambda t: t.to(torch.float32)
f64 = lambda t: t.to(torch.float64)
m = torch.randn(N, device=dev) * 10.0 ** torch.randint(-6, 2, (N,), device=dev).float()
g = torch.randn(N, device=dev) * 10.0 ** torch.randint(-6, 2, (N,), device=dev).float()
v = (torch.randn(N, device=dev) * 10.0 ** torch.randint(-8, 2, (N,), device=dev).float()).abs()
p = torch.randn(N, device=dev)


def rate(name, got, cands):
    print(name, {k: f"{float((got == c).float().mean()):.6f}" for k, c in cands.items()})


# 1. lerp
w = 1 - 0.9
wf = torch.tensor(w, dtype=torch.float64).float().item()
got = torch._foreach_lerp([m.clone()], [g], w)[0]
diff = g - m
rate("lerp", got, {"fma(w, g - m, m)": f32(f64(m) + wf * f64(diff)), "m + rnd(w (g - m))": m + f32(wf * f64(diff)),
                   "w as double": f32(f64(m) + w * f64(diff))})
# 2. mul by beta2
b2 = 0.999
got = torch._foreach_mul([v.clone()], b2)[0]
b2f = torch.tensor(b2, dtype=torch.float64).float().item()
rate("mul", got, {"v * f32(b2)": f32(f64(v) * b2f), "v * double(b2)": f32(f64(v) * b2)})
# 3. addcmul
c = 1 - b2
cf = torch.tensor(c, dtype=torch.float64).float().item()
got = torch._foreach_addcmul([v.clone()], [g], [g], c)[0]
cg = f32(cf * f64(g))
rate("addcmul", got, {"fma(rnd(c g), g, v)": f32(f64(v) + f64(cg) * f64(g)), "v + rnd(rnd(c g) g)": v + f32(f64(cg) * f64(g)),
                      "v + rnd(c rnd(g g))": v + f32(cf * f64(f32(f64(g) * f64(g)))), "fma(c, rnd(g g), v)": f32(f64(v) + cf * f64(f32(f64(g) * f64(g)))),
                      "exact triple": f32(f64(v) + cf * f64(g) * f64(g))})
# 4. sqrt
got = torch._foreach_sqrt([v])[0]
rate("sqrt", got, {"correctly rounded": f32(f64(v).sqrt())})
# 5. div by scalar list
s = (1 - 0.999 ** 7) ** 0.5
sf = torch.tensor(s, dtype=torch.float64).float().item()
d0 = v.sqrt()
got = torch._foreach_div([d0.clone()], [s])[0]
rate("div", got, {"correctly rounded by f32(s)": f32(f64(d0) / sf), "x * rnd(1 / s)": d0 * torch.tensor(1.0 / sf, dtype=torch.float64).float(),
                  "by double s": f32(f64(d0) / s), "x * f32(1/double s)": f32(f64(d0) * float(torch.tensor(1.0 / s, dtype=torch.float64).float()))})
# 6. add eps
eps = 1e-7
got = torch._foreach_add([d0.clone()], eps)[0]
rate("add", got, {"x + f32(eps)": f32(f64(d0) + float(torch.tensor(eps, dtype=torch.float64).float())), "x + double eps": f32(f64(d0) + eps)})
# 7. addcdiv
a = -5e-4 / (1 - 0.9 ** 7)
af = torch.tensor(a, dtype=torch.float64).float().item()
den = d0 + 1e-7
got = torch._foreach_addcdiv([p.clone()], [m], [den], [a])[0]
q = f32(f64(m) / f64(den))
rate("addcdiv", got, {"fma(a, rnd(m / d), p)": f32(f64(p) + af * f64(q)), "p + rnd(a rnd(m / d))": p + f32(af * f64(q)),
                      "p + rnd(rnd(a m) / d)": p + f32(f64(f32(af * f64(m))) / f64(den)), "exact": f32(f64(p) + af * f64(m) / f64(den)),
                      "fma(a m ... )": f32(f64(p) + f64(f32(af * f64(m))) / f64(den))})
got2 = torch._foreach_addcdiv([p.clone()], [m], [den], a)[0]
print("addcdiv scalar == scalarlist:", bool(torch.equal(got, got2)))
