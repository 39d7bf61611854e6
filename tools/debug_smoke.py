import math, sys, torch
sys.path.insert(0, '/root/repo')
import oracle as O
import nerf_atlas_amd.nerf as nerf
from nerf_atlas_amd import config, cameras, render, ops
dev = torch.device('cuda', 0)
def run(seed, size, T, crop, prec, init='default'):
    torch.manual_seed(seed)
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]])
    model = nerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind='upshifted').to(dev).eval()
    cam = cameras.NeRFCamera(cam_to_world=c2w, focal=focal).to(dev)
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rays_ref = O.nerf_camera_rays(O.pixel_grid(size, crop), c2w, focal, size)
    aux = {}
    ref = O.plain_nerf(params, rays_ref, 2.0, 6.0, T, 'view', act='upshifted', aux=aux)
    config.set_precision(prec)
    out, rays = render.render(model, cam, crop, size, with_noise=False)
    err = float((out.cpu() - ref).abs().max())
    # unfused path
    pts, ts, r_o, r_d, _ = nerf.compute_pts_ts(rays, 2.0, 6.0, T)
    un = model.from_pts(pts, ts, r_o, r_d, rays=rays)
    err2 = float((un.cpu() - ref).abs().max())
    werr = float((model.weights.cpu() - aux['weights']).abs().max())
    # density via first MLP
    fo = model.first(pts)
    dref = aux['density']
    derr = float((fo[..., 0].cpu() - dref).abs().max())
    print(f'seed {seed} size {size} T {T} crop {crop} {prec}: fused err {err:.3e} unfused err {err2:.3e} w_err(unfused) {werr:.3e} density err {derr:.3e} |density| {float(dref.abs().max()):.2f}')
for prec in ('bf16x3', 'bf16'):
    run(0, 64, 32, (28, 28, 8, 8), prec)
    run(2, 64, 32, (28, 28, 8, 8), prec)
    run(0, 800, 128, (380, 390, 8, 8), prec)
    run(0, 64, 16, (28, 28, 8, 8), prec)
    run(0, 64, 32, (0, 0, 8, 8), prec)
