"""Component timing of config 3 (PlainNeRF + mip) with HIP events (tools/mip_profile.py [bf16|bf16x3])."""
import sys, types, torch
sys.path.insert(0, "/root/repo")
from nerf_atlas_amd import nerf, config, ops
from nerf_atlas_amd.utils import load_mip
from nerf_atlas_amd.nerf import compute_pts_ts, cat_not_none
config.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
torch.manual_seed(0)
m = nerf.PlainNeRF(intermediate_size=64, mip=load_mip(types.SimpleNamespace(mip="cylinder")), steps=128, t_near=2.0, t_far=6.0, sigmoid_kind="upshifted").cuda().eval()
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]]).cuda()
rays = ops.raygen(c2w, 0.5 * 800 / 0.36, 800, (200, 200, 200, 200))
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out
with torch.no_grad():
    t_all, _ = timed(lambda: m(rays))
    pts, ts, r_o, r_d, _ = compute_pts_ts(rays, 2.0, 6.0, 128, perturb=0)
    lat = m.mip_latent(rays, ts)
    t_first, first_out = timed(lambda: m.first(pts, lat))
    inter = first_out[..., 1:]
    view = r_d.unsqueeze(0).expand_as(pts)
    rl = cat_not_none(lat, inter)
    t_refl, rgb = timed(lambda: m.refl(x=pts, view=view, latent=rl))
    t_mlp, _ = timed(lambda: m.refl.mlp(torch.zeros(pts.shape[:-1] + (5,), device="cuda"), rl))
    dens = first_out[..., 0].contiguous()
    t_comp, _ = timed(lambda: m._composite(dens, rgb, ts, rays))
    t_pts, _ = timed(lambda: compute_pts_ts(rays, 2.0, 6.0, 128, perturb=0))
    t_lat, _ = timed(lambda: ops.mip_encode(rays, ts, "cylinder", 6.03, 0, 16))
print(f"all {t_all:.2f} ms | pts {t_pts:.2f} first {t_first:.2f} refl {t_refl:.2f} (mlp+zeros {t_mlp:.2f}) composite {t_comp:.2f} | standalone mip_encode {t_lat:.2f}")
