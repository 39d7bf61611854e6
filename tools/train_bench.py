#!/usr/bin/env python3
"""Training-step timing of the N1 path (forward + backward HIP kernels + torch.optim.Adam), PlainNeRF(view),
the reference's `make original`-like recipe at a crop: --crop 64 --steps 64 --batch 1 (SURVEY 8(d)(iii) measured the
reference itself at ~2.9 s/iteration for 73 728 samples on 8 CPU cores).

    python tools/train_bench.py [--crop 64] [--steps 64] [--iters 20] [--cpu-oracle]
"""
import argparse, json, math, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crop", type=int, default=64)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cpu-oracle", action="store_true")
    a = ap.parse_args()
    import nerf_atlas_amd.nerf as nerf
    from nerf_atlas_amd import ops
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    size = 800
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    m = nerf.PlainNeRF(steps=a.steps, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").to(dev)
    m.eval()  # deterministic sampling; gradients still flow (training noise terms are plumbing)
    import types
    from nerf_atlas_amd import train
    opt = train.load_optim(types.SimpleNamespace(opt_kind="adam", learning_rate=2e-4, decay=0), m.parameters())  # (the product's Adam)
    target = torch.rand(1, a.crop, a.crop, 3, device=dev)
    losses = []
    def step():
        rays = ops.raygen(c2w, focal, size, ((size - a.crop) // 2, (size - a.crop) // 2, a.crop, a.crop))
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(m(rays), target)
        loss.backward()
        opt.step()
        return loss
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters): losses.append(float(step().detach()))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    n = a.crop * a.crop * a.steps
    res = {"what": "PlainNeRF(view) training step: fwd + bwd (HIP kernels, split-bf16 GEMMs) + Adam", "samples_per_step": n,
           "ms_per_step": round(dt * 1e3, 2), "Msamples_per_s": round(n / dt / 1e6, 3), "loss_first": losses[0], "loss_last": losses[-1]}
    if a.cpu_oracle:
        import oracle as O
        p = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
        rays = O.nerf_camera_rays(O.pixel_grid(size, ((size - a.crop) // 2, (size - a.crop) // 2, a.crop, a.crop)), c2w.cpu(), focal, size)
        tgt = target.cpu()
        t0 = time.perf_counter()
        loss = torch.nn.functional.mse_loss(O.plain_nerf(p, rays, 2.0, 6.0, a.steps, "view", act="upshifted"), tgt)
        loss.backward()
        res["cpu_oracle_fwd_bwd_s"] = round(time.perf_counter() - t0, 2)
        res["cpu_threads"] = torch.get_num_threads()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
