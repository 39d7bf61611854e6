"""Round-6 bring-up of the one-launch PlainNeRF + Positional / PosLinearView renderers (MODEL 7 / 8 of render_ls_kernel):
the fused f16x launch against (i) the reference goldens g11 and (ii) the unfused bf16x3 operator chain on bigger random batches,
with and without explicit points / refl_latent.  Prints L-inf errors; exit code 1 if any is over 1e-4.

    python tools/head_check.py [--big]
"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from conftest import load_golden, golden_params  # noqa: E402
import nerf_atlas_amd.nerf as nerf  # noqa: E402
import nerf_atlas_amd.refl as refl  # noqa: E402
from nerf_atlas_amd import config, ops  # noqa: E402


def build(kind, n_rl=0, steps=16, near=2.0, far=6.0, bg="black", seed=0):
    torch.manual_seed(seed)
    m = nerf.PlainNeRF(steps=steps, t_near=near, t_far=far, intermediate_size=64, sigmoid_kind="upshifted", bg=bg)
    m.set_refl(refl.refl_kinds[kind](latent_size=64 + n_rl, act="upshifted", out_features=3))
    return m.cuda().eval()


def md(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max())


def main():
    bad = []
    calls = {"pos": 0, "plv": 0}
    for name in ("pos", "plv"):
        fn = getattr(ops, f"render_plain_{name}_ls")

        def wrap(*a, _fn=fn, _n=name, **k):
            calls[_n] += 1
            return _fn(*a, **k)
        setattr(ops, f"render_plain_{name}_ls", wrap)
    with torch.no_grad():
        # (i) the reference goldens
        for kind in ("pos", "pos-linear-view"):
            for B in (1, 2):
                h = load_golden(f"g11_plain_{kind}_b{B}")
                m = build(kind, steps=int(h["steps"]), near=float(h["near"]), far=float(h["far"]), bg=str(h["bg"]))
                sd = m.state_dict()
                for k, v in golden_params(h).items():
                    sd[k].copy_(v)
                for prec in ("bf16x3", "f16x"):
                    config.set_precision(prec)
                    out = m(h["rays"].cuda())
                    e = (md(out, h["out"]), md(m.alpha, h["alpha"]), md(m.weights, h["weights"]))
                    print(f"golden {kind:16s} B={B} {prec:7s} out {e[0]:.2e} alpha {e[1]:.2e} weights {e[2]:.2e}", flush=True)
                    if max(e) > 1e-4:
                        bad.append((kind, B, prec, e))
        # (ii) random weights, bigger batches, against the unfused bf16x3 chain
        shapes = [((1, 7, 9), 48), ((2, 16, 16), 128), ((1, 33, 31), 70)] + ([((1, 200, 200), 128)] if "--big" in sys.argv else [])
        for kind, n_rl in (("pos", 0), ("pos-linear-view", 0), ("pos-linear-view", 1), ("pos-linear-view", 3)):
            for shp, T in shapes:
                m = build(kind, n_rl=n_rl, steps=T, seed=3)
                g = torch.Generator().manual_seed(5)
                o = torch.tensor([0.1, -0.2, 4.0]) + 0.05 * torch.randn(shp + (3,), generator=g)
                d = torch.nn.functional.normalize(torch.tensor([0.0, 0.05, -1.0]) + 0.15 * torch.randn(shp + (3,), generator=g), dim=-1) * 1.1
                rays = torch.cat([o, d], dim=-1).cuda()
                rl = (0.5 * torch.randn((T,) + shp + (n_rl,), generator=g)).cuda() if n_rl else None
                for with_pts in (False, True):
                    if rl is not None and not with_pts:
                        continue
                    res = {}
                    for prec in ("bf16x3", "f16x"):
                        config.set_precision(prec)
                        if with_pts:
                            pts, ts, r_o, r_d, _ = nerf.compute_pts_ts(rays, m.t_near, m.t_far, m.steps)
                            pts = (pts + 0.01 * torch.sin(pts * 3.0)).contiguous()
                            out = m.from_pts(pts, ts, r_o, r_d, refl_latent=rl, rays=rays)
                        else:
                            out = m(rays)
                        torch.cuda.synchronize()
                        res[prec] = (out.clone(), m.alpha.clone(), m.weights.clone())
                    e = tuple(md(a, b) for a, b in zip(res["f16x"], res["bf16x3"]))
                    nan = bool(torch.isnan(res["f16x"][0]).any())
                    print(f"random {kind:16s} n_rl={n_rl} {shp} T={T} pts={int(with_pts)} out {e[0]:.2e} alpha {e[1]:.2e} weights {e[2]:.2e}"
                          f"{' NaN' if nan else ''}", flush=True)
                    if nan or max(e) > 1e-4:
                        bad.append((kind, n_rl, shp, T, with_pts, e))
    print("fused launches:", calls)
    if calls["pos"] == 0 or calls["plv"] == 0:
        bad.append(("fused path not taken", calls))
    if bad:
        print("FAILED:", *bad, sep="\n  ")
        sys.exit(1)
    print("OK")


if __name__ == "__main__":
    main()
