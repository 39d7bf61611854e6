#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box).  What is
committed are small fixtures: inputs + expected outputs (+ parameter name/shape lists so the
tests can regenerate the procedural weights with oracle/procedural.py).  No reference source
is copied; the reference modules are imported in-process with the shims of SURVEY.md App. A.

    python tools/gen_golden.py            # rewrites every fixture
"""
import os
import sys
import types
import math

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.procedural import proc_param, proc_uniform  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"


# ------------------------------------------------------------------ shims
def _stub(name, subs=()):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    for s in subs:
        sm = types.ModuleType(f"{name}.{s}")
        sm.__path__ = []
        sys.modules[f"{name}.{s}"] = sm
        setattr(m, s, sm)


_stub("torchvision", ["models", "transforms", "io"])
_tf = types.ModuleType("torchvision.transforms.functional")
sys.modules["torchvision.transforms.functional"] = _tf
sys.modules["torchvision.transforms"].functional = _tf
_stub("imageio")
sys.path.insert(0, REF)
import src.nerf as rnerf  # noqa: E402
import src.utils as rutils  # noqa: E402
import src.cameras as rcam  # noqa: E402
import src.neural_blocks as rnb  # noqa: E402
import src.refl as rrefl  # noqa: E402
import src.sdf as rsdf  # noqa: E402

nn.Module.cuda = lambda self, *a, **k: self
rnerf.with_transmission = False
rutils.git_hash = lambda: "nogit"
torch.set_grad_enabled(False)
torch.set_num_threads(8)


def save(name, **kw):
    out = {}
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def fill_procedural(module: nn.Module, sigma_by_suffix=None):
    """Overwrite every parameter/buffer with oracle.procedural.proc_param(name, shape).
    Returns (names, shapes) in state_dict order."""
    names, shapes = [], []
    sd = module.state_dict()
    for name, t in sd.items():
        if name.endswith("primes") or t.numel() == 0:
            continue
        if name == "scale" or name.endswith(".scale"):
            continue  # VolSDF beta stays at its init (0.1)
        v = torch.from_numpy(proc_param(name, tuple(t.shape)))
        if name.endswith("basis"):
            sigma = 16.0 if "sdf" in name or "underlying" in name else 32.0
            if sigma_by_suffix is not None:
                sigma = sigma_by_suffix
            v = v * sigma
        t.copy_(v.to(t.dtype))
        names.append(name)
        shapes.append(list(t.shape))
    return names, shapes


def spec(names, shapes):
    flat = []
    for s in shapes:
        flat.append(",".join(str(d) for d in s))
    return dict(param_names=np.array(names), param_shapes=np.array(flat))


POSES = torch.tensor([
    [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]],
    [[0.8, -0.36, 0.48, 1.9], [0.0, 0.8, 0.6, 2.4], [-0.6, -0.48, 0.64, 2.6]],
    [[-0.28, 0.0, -0.96, -3.8], [0.576, 0.8, -0.168, -0.7], [0.768, -0.6, -0.224, -0.9]],
], dtype=torch.float)


def cam(poses=POSES[:1], size=16, fov=0.6911):
    focal = 0.5 * size / math.tan(0.5 * fov)
    return rcam.NeRFCamera(cam_to_world=poses.clone(), focal=focal), focal


def ref_pixel_grid(size, crop):
    ii, jj = torch.meshgrid(torch.arange(size, dtype=torch.float), torch.arange(size, dtype=torch.float),
                            indexing="ij")
    positions = torch.stack([ii.transpose(-1, -2), jj.transpose(-1, -2)], dim=-1)
    t, l, h, w = crop
    return positions[t:t + h, l:l + w, :]


# ------------------------------------------------------------------ G1 cameras
def g1():
    size = 16
    c, focal = cam(POSES, size)
    crops = [(0, 0, 16, 16), (3, 5, 6, 7), (12, 12, 8, 8)]  # full, interior, ragged edge (slice clips)
    kw = dict(c2w=POSES, focal=np.float64(focal), size=size, crops=np.array(crops))
    for i, crop in enumerate(crops):
        pos = ref_pixel_grid(size, crop)
        kw[f"pos{i}"] = pos
        kw[f"rays{i}"] = c.sample_positions(pos, size=size, with_noise=False)
    # jittered rays: pin the global RNG draw order (u first, then v)
    torch.manual_seed(11)
    pos = ref_pixel_grid(size, crops[1])
    ru = torch.rand_like(pos[..., :1])
    rv = torch.rand_like(pos[..., :1])
    torch.manual_seed(11)
    kw["noise"] = torch.cat([ru, rv], dim=-1)
    kw["rays_noise"] = c.sample_positions(pos, size=size, with_noise=0.1)
    save("g1_nerf_camera", **kw)

    # DTU camera with synthetic pose/intrinsics
    size = 12
    pose = torch.eye(4).repeat(2, 1, 1)
    pose[0, :3, :4] = torch.tensor([[0.9, 0.1, -0.42, 0.5], [-0.05, 0.99, 0.12, -0.2], [0.43, -0.09, 0.9, -1.1]])
    pose[1, :3, :4] = torch.tensor([[0.6, -0.3, 0.74, -0.8], [0.2, 0.95, 0.22, 0.1], [-0.77, 0.02, 0.63, -0.9]])
    intr = torch.eye(4).repeat(2, 1, 1)
    intr[:, 0, 0] = 2892.0
    intr[:, 1, 1] = 2883.0
    intr[:, 0, 2] = 800.0
    intr[:, 1, 2] = 600.0
    intr[1, 0, 1] = 1.5
    d = rcam.DTUCamera(pose=pose.clone(), intrinsic=intr.clone())
    pos = ref_pixel_grid(size, (2, 1, 7, 9))
    save("g1_dtu_camera", pose=pose, intrinsic=intr, size=size, pos=pos,
         rays=d.sample_positions(pos, size=size))


# ------------------------------------------------------------------ G2 sampling
def g2():
    rays = torch.zeros(1, 2, 2, 6)
    kw = {}
    for tag, (near, far, T, lindisp) in {"lin": (2.0, 6.0, 16, False), "disp": (0.3, 1.8, 16, True),
                                         "lin128": (2.0, 6.0, 128, False)}.items():
        _, _, ts, _ = rnerf.compute_ts(rays, near, far, T, lindisp=lindisp)
        kw[f"ts_{tag}"] = ts
        kw[f"cfg_{tag}"] = np.array([near, far, T, int(lindisp)], dtype=np.float64)
    torch.manual_seed(5)
    rand = torch.rand(16)
    torch.manual_seed(5)
    _, _, ts, mids = rnerf.compute_ts(rays, 2.0, 6.0, 16, perturb=1.0)
    kw.update(rand=rand, ts_perturb=ts, mids=mids)
    c, focal = cam(POSES[1:2], 8)
    r = c.sample_positions(ref_pixel_grid(8, (0, 0, 8, 8)), size=8)
    pts, ts, r_o, r_d, _ = rnerf.compute_pts_ts(r, 2.0, 6.0, 16)
    kw.update(rays=r, pts=pts)
    save("g2_sampling", **kw)


# ------------------------------------------------------------------ G3 compositing
def g3():
    T, B, H, W = 16, 2, 3, 4
    density = torch.from_numpy(proc_uniform((T, B, H, W), 101, 4.0))
    density[3, 0, 0, 0] = 60.0  # saturating: alpha == 1
    density[5, 1, 2, 3] = 1e4
    density[7, 0, 1, 1] = -50.0
    rgb = torch.from_numpy(proc_uniform((T, B, H, W, 3), 102, 0.5)) + 0.5
    r_d = torch.from_numpy(proc_uniform((B, H, W, 3), 103, 1.0))
    ts = torch.linspace(2, 6, T)
    ts_zero = ts.clone()
    ts_zero[5] = ts_zero[4]  # zero-length interval -> clamp 1e-5
    kw = dict(density=density, rgb=rgb, r_d=r_d, ts=ts, ts_zero=ts_zero)
    for tag, (t, sp) in {"softplus": (ts, True), "relu": (ts, False), "zero": (ts_zero, True)}.items():
        alpha, weights = rnerf.alpha_from_density(density, t, r_d, softplus=sp)
        out = rnerf.volumetric_integrate(weights, rgb)
        kw[f"alpha_{tag}"] = alpha
        kw[f"weights_{tag}"] = weights
        kw[f"out_{tag}"] = out
        kw[f"white_{tag}"] = out + rnerf.white(None, weights)
    # KAT from SURVEY 8(c)
    a, w = rnerf.alpha_from_density(torch.tensor([0.5, 1, 2, -1.0]).reshape(4, 1, 1, 1), torch.linspace(2, 6, 4),
                                    torch.tensor([0, 0, -1.0]).reshape(1, 1, 1, 3))
    kw.update(kat_alpha=a.reshape(-1), kat_weights=w.reshape(-1))
    save("g3_composite", **kw)


# ------------------------------------------------------------------ G4 hash
def hash_inputs():
    x = torch.from_numpy(proc_uniform((61, 3), 201, 3.0))
    special = torch.tensor([[0.3, -1.7, 2.2], [0.0, 0.0, 0.0], [1.0, -1.0, 2.0], [-3.0, 3.0, -0.5],
                            [0.0625, -0.0625, 0.125], [5.99, -5.99, 4.0], [-40.0, 37.5, 55.25]])
    return torch.cat([special, x], dim=0)


def g4():
    enc = rnb.HashEncoder()
    names, shapes = fill_procedural(enc)
    x = hash_inputs()
    idx = []
    for i in range(enc.levels):
        N_l = enc.low_reso * (enc.scale ** i)
        l = (x * N_l).floor().long()
        lx, ly, lz = l.split([1, 1, 1], dim=-1)
        h = l + 1
        hx, hy, hz = h.split([1, 1, 1], dim=-1)
        cat = lambda a, b, c: torch.cat([a, b, c], dim=-1)
        vs = [l, cat(lx, ly, hz), cat(lx, hy, lz), cat(lx, hy, hz), cat(hx, ly, lz), cat(hx, ly, hz),
              cat(hx, hy, lz), h]
        idx.append(torch.stack([(enc.hash_fn(v) % enc.emb_size).squeeze(-1) for v in vs], dim=0))
    save("g4_hash", x=x, idx=torch.stack(idx, 0), feats=enc(x),
         resolutions=np.array([enc.low_reso * (enc.scale ** i) for i in range(enc.levels)], dtype=np.float64),
         **spec(names, shapes))


# ------------------------------------------------------------------ G5 fourier / positional
def g5():
    kw = {}
    for sigma in (16, 32):
        enc = rnb.FourierEncoder(input_dims=3, sigma=sigma)
        basis = torch.from_numpy(proc_param("basis", (3, 128))) * sigma
        enc.basis.copy_(basis)
        x = torch.from_numpy(proc_uniform((48, 3), 300 + sigma, 3.0))
        kw[f"x_{sigma}"] = x
        kw[f"out_{sigma}"] = enc(x)
        kw[f"out64_{sigma}"] = rutils.fourier(x.double(), basis.double())
    pe = rnb.PositionalEncoder(input_dims=3, max_freq=6.0, N=8)
    x = torch.from_numpy(proc_uniform((20, 3), 333, 2.0))
    kw.update(pe_x=x, pe_bands=pe.bands, pe_out=pe(x))
    save("g5_fourier", **kw)


# ------------------------------------------------------------------ G6 SkipConnMLP
MLP_CASES = {
    # tag: (in, enc, latent, layers, out, act, init)
    "tiny": (3, None, 0, 6, 4, "leaky_relu", "xavier"),
    "first": (3, "hash", 0, 4, 65, "leaky_relu", None),
    "view": (5, None, 64, 4, 3, "sin", "siren"),
    "posrefl": (3, "hash", 64, 5, 3, "leaky_relu", None),
    "delta6": (3, "hash", 0, 5, 19, "leaky_relu", "xavier"),
    "sdfmlp": (3, "fourier16", 0, 6, 65, "leaky_relu", "xavier"),
    "siren": (3, None, 0, 5, 65, "sin", "siren"),
    "mipfirst": (3, "hash", 96, 4, 65, "leaky_relu", None),
    "plv_view": (6, None, 128, 2, 1, "sin", "siren"),  # hidden 128
    "plv_pos": (3, "hash", 0, 2, 67, "leaky_relu", None),
}


def make_mlp(case):
    in_size, enc, latent, layers, out, act, init = MLP_CASES[case]
    e = None
    if enc == "hash":
        e = rnb.HashEncoder()
    elif enc == "fourier16":
        e = rnb.FourierEncoder(input_dims=in_size, sigma=16)
    hidden = 128 if case == "plv_view" else 256
    kwargs = dict(num_layers=layers, hidden_size=hidden, in_size=in_size, out=out, latent_size=latent, enc=e, init=init)
    if act == "sin":
        kwargs["activation"] = torch.sin
    return rnb.SkipConnMLP(**kwargs)


def g6():
    for case in MLP_CASES:
        in_size, enc, latent, layers, out, act, init = MLP_CASES[case]
        m = make_mlp(case)
        names, shapes = fill_procedural(m, sigma_by_suffix=16.0 if enc == "fourier16" else None)
        N = 40
        p = torch.from_numpy(proc_uniform((N, in_size), 400, 2.5))
        lat = torch.from_numpy(proc_uniform((N, latent), 401, 1.0)) if latent else None
        inter = []
        hooks = [m.init.register_forward_hook(lambda mod, i, o: inter.append(o.clone()))]
        for l in m.layers:
            hooks.append(l.register_forward_hook(lambda mod, i, o: inter.append(o.clone())))
        y = m(p, lat)
        for h in hooks:
            h.remove()
        kw = dict(p=p, y=y, act=act, enc=str(enc), layers=layers, out=out, latent_size=latent, **spec(names, shapes))
        if lat is not None:
            kw["latent"] = lat
        for i, t in enumerate(inter):
            kw[f"inter{i}"] = t
        save(f"g6_mlp_{case}", **kw)


# ------------------------------------------------------------------ G7 elaz + heads
def g7():
    d = torch.from_numpy(proc_uniform((29, 3), 500, 1.0))
    d = torch.cat([d, torch.tensor([[0, 0, 1.0], [0, 0, -1.0], [0, 0, -5.0], [1e-4, -1e-4, 3.0], [-0.08, 0.08, -1.0],
                                    [2.0, 0, 0], [0, -3.0, 0]])], dim=0)
    kw = dict(dirs=d, elaz=rutils.dir_to_elev_azim(d))
    v = torch.from_numpy(proc_uniform((33,), 501, 6.0))
    for k in ["normal", "thin", "fat", "tanh", "upshifted", "relu", "sin", "leaky_relu", "upshifted_softplus",
              "upshifted_relu", "cyclic"]:
        kw[f"sig_{k}"] = rutils.load_sigmoid(k)(v)
    kw["sig_in"] = v
    save("g7_elaz_sigmoid", **kw)
    N = 24
    x = torch.from_numpy(proc_uniform((N, 3), 510, 2.5))
    view = torch.from_numpy(proc_uniform((N, 3), 511, 1.0))
    lat = torch.from_numpy(proc_uniform((N, 64), 512, 1.0))
    for kind, cons in {"view": rrefl.View, "pos": rrefl.Positional, "pos-linear-view": rrefl.PosLinearView}.items():
        for act in (["thin", "upshifted"] if kind == "view" else ["thin"]):
            r = cons(latent_size=64, act=act, out_features=3)
            names, shapes = fill_procedural(r)
            save(f"g7_refl_{kind}_{act}", x=x, view=view, latent=lat, rgb=r(x=x, view=view, latent=lat), act=act,
                 **spec(names, shapes))


# ------------------------------------------------------------------ G8 mip primitives
def g8():
    x = torch.from_numpy(proc_uniform((10, 3), 600, 3.0))
    var = torch.from_numpy(proc_uniform((10, 3), 601, 0.5)).abs()
    y, yv = rutils.expected_sin(x, var)
    kw = dict(x=x, var=var, es_y=y, es_var=yv, ipe=rutils.integrated_pos_enc_diag(x, var, 0, 16))
    c, focal = cam(POSES[:2], 16)
    for i, crop in enumerate([(0, 0, 16, 16), (3, 5, 6, 7), (10, 2, 6, 9)]):
        rays = c.sample_positions(ref_pixel_grid(16, crop), size=16)
        kw[f"rd{i}"] = rays[..., 3:]
        kw[f"radii{i}"] = rutils.radii_x(rays[..., 3:])
    t0 = torch.linspace(2, 5.75, 16)
    t1 = t0 + 0.25
    rad = torch.tensor(0.003)
    # scalar moments, taken from the reference by probing lift_gaussian with a unit z direction:
    # mean_z = t_mean, cov_zz = t_var, cov_xx = r_var
    rd = torch.tensor([[0.0, 0.0, 1.0]])
    mean, cov = rutils.cylinder_to_gaussian(rd, t0, t1, rad)
    kw.update(t0=t0, t1=t1, rad=rad, cyl_tmean=mean[:, 0, 2], cyl_cov=cov)
    mean, cov = rutils.conical_frustrum_to_gaussian(rd, t0, t1, float(rad))
    kw.update(cone_tmean=mean[:, 0, 2], cone_cov=cov)
    save("g8_mip", **kw)


# ------------------------------------------------------------------ G9 bezier + D-NeRF
def g9():
    kw = {}
    t = torch.from_numpy(proc_uniform((7, 1), 700, 0.5)) + 0.5
    for n in range(2, 7):
        co = torch.from_numpy(proc_uniform((n, 7, 3), 701 + n, 1.0))
        kw[f"coeffs{n}"] = co
        kw[f"dc{n}"] = rnerf.de_casteljau(co, t, n)
    kw["cubic"] = rnerf.cubic_bezier(kw["coeffs4"], t, 4)
    kw["t"] = t
    save("g9_bezier", **kw)
    for spline in (6, 4):
        size, T = 6, 8
        canon = rnerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted")
        m = rnerf.DynamicNeRF(canonical=canon, spline=spline)
        m.eval()
        names, shapes = fill_procedural(m)
        c, focal = cam(POSES[:2], size)
        rays = c.sample_positions(ref_pixel_grid(size, (0, 0, size, size)), size=size)
        times = torch.tensor([0.25, 0.8])
        out = m((rays, times))
        save(f"g9_dnerf_spline{spline}", rays=rays, times=times, out=out, steps=T, near=2.0, far=6.0,
             rigidity=m.rigidity, dp=m.dp, weights=canon.weights, **spec(names, shapes))
    # `make dnerf` as shipped (makefile:106-114): --spline 6 --dyn-refl-latent 3 --refl-kind pos-linear-view; + the View head and
    # the cubic spline with the same latent (runner.py:1169-1211 builds the head with latent_size = model.intermediate_size)
    for spline, kind, rl in ((6, "pos-linear-view", 3), (6, "view", 3), (4, "pos-linear-view", 2)):
        size, T = 6, 8
        canon = rnerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted")
        m = rnerf.DynamicNeRF(canonical=canon, spline=spline, refl_latent=rl)
        m.set_refl(rrefl.refl_kinds[kind](latent_size=m.intermediate_size, act="upshifted", out_features=3))
        m.eval()
        names, shapes = fill_procedural(m)
        c, focal = cam(POSES[:2], size)
        rays = c.sample_positions(ref_pixel_grid(size, (0, 0, size, size)), size=size)
        times = torch.tensor([0.25, 0.8])
        captured = {}
        orig = canon.from_pts
        canon.from_pts = lambda pts, ts, r_o, r_d, enc=None: (captured.update(enc=enc), orig(pts, ts, r_o, r_d, enc))[1]
        out = m((rays, times))
        tag = {"pos-linear-view": "plv", "view": "view"}[kind]
        save(f"g9_dnerf_spline{spline}_rl{rl}_{tag}", rays=rays, times=times, out=out, steps=T, near=2.0, far=6.0,
             rigidity=m.rigidity, dp=m.dp, weights=canon.weights, refl_latent=captured["enc"], refl_kind=kind, n_rl=rl,
             **spec(names, shapes))


# ------------------------------------------------------------------ G10 laplace + VolSDF
def g10():
    s = torch.from_numpy(proc_uniform((40,), 800, 0.6))
    s[0] = 0.0
    kw = dict(sdf=s)
    for sc in (0.1, 0.02, 1.5):
        kw[f"cdf_{sc}"] = rutils.laplace_cdf(s, torch.tensor(sc))
    save("g10_laplace", **kw)
    for kind in ("mlp", "siren"):
        size, T = 6, 8
        under = rsdf.sdf_kinds[kind](intermediate_size=64)
        refl = rrefl.View(latent_size=64, act="upshifted", out_features=3)
        sdf = rsdf.SDF(under, refl, isect=None, t_near=0.3, t_far=1.8)
        m = rnerf.VolSDF(sdf=sdf, steps=T, t_near=0.3, t_far=1.8, sigmoid_kind="upshifted")
        m.eval()
        names, shapes = fill_procedural(m)
        c, focal = cam(POSES[:2], size)
        rays = c.sample_positions(ref_pixel_grid(size, (0, 0, size, size)), size=size)
        rays = torch.cat([rays[..., :3] * 0.2, torch.nn.functional.normalize(rays[..., 3:], dim=-1)], dim=-1)
        out = m(rays)
        save(f"g10_volsdf_{kind}", rays=rays, out=out, steps=T, near=0.3, far=1.8, scale=m.scale,
             weights=m.weights, alpha=m.alpha, **spec(names, shapes))


# ------------------------------------------------------------------ G11 PlainNeRF
def g11():
    for kind in ("view", "pos", "pos-linear-view"):
        for B in (1, 2):
            size, T = 6, 16
            m = rnerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted",
                                bg="white" if B == 2 else "black")
            if kind != "view":
                r = rrefl.refl_kinds[kind](latent_size=64, act="upshifted", out_features=3)
                m.set_refl(r)
            m.eval()
            names, shapes = fill_procedural(m)
            c, focal = cam(POSES[:B], size)
            rays = c.sample_positions(ref_pixel_grid(size, (0, 0, size, size)), size=size)
            out = m(rays)
            save(f"g11_plain_{kind}_b{B}", rays=rays, out=out, steps=T, near=2.0, far=6.0,
                 bg="white" if B == 2 else "black", ts=m.ts, alpha=m.alpha, weights=m.weights, **spec(names, shapes))
    # fp64 tie-breaker for the headline model
    import copy
    m = rnerf.PlainNeRF(steps=16, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted")
    m.eval()
    names, shapes = fill_procedural(m)
    c, focal = cam(POSES[:1], 6)
    rays = c.sample_positions(ref_pixel_grid(6, (0, 0, 6, 6)), size=6)
    m64 = copy.deepcopy(m).double()
    save("g11_plain_view_fp64", rays=rays, out32=m(rays), out64=m64(rays.double()), **spec(names, shapes))


# ------------------------------------------------------------------ G12 tiled frame (runner.render + test()-style tiling)
def g12():
    import runner  # the reference's own render()
    runner.device = "cpu"
    size, T, cs = 20, 8, 8  # ragged: 20 = 8+8+4
    m = rnerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted")
    m.eval()
    names, shapes = fill_procedural(m)
    c, focal = cam(POSES[1:2], size)
    args = types.SimpleNamespace(data_kind="original")
    got = torch.zeros(size, size, 3)
    n = math.ceil(size / cs)
    for x in range(n):
        c0 = x * cs
        for y in range(n):
            c1 = y * cs
            out, rays = runner.render(m, c, (c0, c1, cs, cs), size=size, with_noise=False, times=None, args=args)
            got[c0:c0 + cs, c1:c1 + cs, :] = out.squeeze(0)
    exp = torch.from_numpy(proc_uniform((size, size, 3), 1200, 0.5)) + 0.5
    mse = torch.nn.functional.mse_loss(got, exp)
    save("g12_tiled_frame", c2w=POSES[1:2], focal=np.float64(focal), size=size, steps=T, crop_size=cs, near=2.0, far=6.0,
         frame=got, exp=exp, psnr=rutils.mse2psnr(mse), **spec(names, shapes))


# ------------------------------------------------------------------ tiny (reference primitives composed)
def g13():
    size, T = 8, 8
    mlp = rnb.SkipConnMLP(in_size=3, out=4, num_layers=6, hidden_size=256, init="xavier")
    sd_names, sd_shapes = [], []
    for name, t in mlp.state_dict().items():
        full = "estim." + name
        t.copy_(torch.from_numpy(proc_param(full, tuple(t.shape))))
        sd_names.append(full)
        sd_shapes.append(list(t.shape))
    c, focal = cam(POSES[:1], size)
    rays = c.sample_positions(ref_pixel_grid(size, (0, 0, size, size)), size=size)
    pts, ts, r_o, r_d, _ = rnerf.compute_pts_ts(rays, 2.0, 6.0, T)
    o = mlp(pts)
    alpha, w = rnerf.alpha_from_density(o[..., 0], ts, r_d)
    out = rnerf.volumetric_integrate(w, rutils.upshifted_sigmoid(o[..., 1:]))
    save("g13_tiny", rays=rays, out=out, steps=T, near=2.0, far=6.0, weights=w, **spec(sd_names, sd_shapes))


# ------------------------------------------------------------------ G14 loaders (N2)
def g14():
    """Reference loaders (src/loaders.py:74-150) on the analytic scene of tools/make_scene.py, at the file size and
    through PIL's resize; labels stored as uint16 fixed point of the exact k/255 grid where possible."""
    import tempfile
    import src.loaders as rload
    from tools.make_scene import make_scene
    kw = {}
    with tempfile.TemporaryDirectory() as td:
        d = make_scene(os.path.join(td, "static"), size=24, n_train=4, n_test=2) + "/"
        for training in (True, False):
            for size in (24, 16):
                for white in (False, True):
                    labels, c, _ = rload.original(d, normalize=False, training=training, size=size, white_bg=white)
                    key = f"orig_{'train' if training else 'test'}_{size}_{'w' if white else 'b'}"
                    kw[key + "_labels"] = labels
                    kw[key + "_c2w"] = c.cam_to_world
                    kw[key + "_focal"] = np.float64(c.focal)
        labels, c, _ = rload.original(d, normalize=True, training=True, size=24, with_mask=True)
        kw["orig_norm_mask_labels"], kw["orig_norm_mask_c2w"] = labels, c.cam_to_world
        dd = make_scene(os.path.join(td, "dyn"), size=24, n_train=5, n_test=2, dynamic=True) + "/"
        # shuffle the frame order and stretch the times so that sorting and normalisation are exercised
        import json
        tf = json.load(open(dd + "transforms_train.json"))
        tf["frames"] = [tf["frames"][i] for i in (3, 0, 4, 1, 2)]
        for fr in tf["frames"]:
            fr["time"] = fr["time"] * 3.0 - 0.5
        json.dump(tf, open(dd + "transforms_train.json", "w"))
        for gamma in (False, True):
            (labels, times), c, _ = rload.dnerf(dd, training=True, size=24, time_gamma=gamma, white_bg=False)
            kw[f"dnerf_g{int(gamma)}_labels"], kw[f"dnerf_g{int(gamma)}_times"] = labels, times
            kw[f"dnerf_g{int(gamma)}_c2w"] = c.cam_to_world
        (labels, times), c, _ = rload.dnerf(dd, training=False, size=16, time_gamma=False, white_bg=True)
        kw["dnerf_test_labels"], kw["dnerf_test_times"], kw["dnerf_test_focal"] = labels, times, np.float64(c.focal)
    save("g14_loaders", **kw)


# ------------------------------------------------------------------ G15 SDF marching (N4)
def g15():
    """Reference src/march.py on (a) an analytic two-sphere SDF (pins the marching logic exactly) and (b) the reference
    SIREN SDF model with procedural weights (pins it through a network)."""
    import random
    import src.march as rmarch
    c2w = POSES[1:2]
    camr, _ = cam(c2w, 12)
    rays = camr.sample_positions(ref_pixel_grid(12, (0, 0, 12, 12)), size=12, with_noise=False)[0]
    r_o, r_d = rays[..., :3].contiguous(), torch.nn.functional.normalize(rays[..., 3:], dim=-1)

    def analytic(p):
        a = torch.linalg.norm(p - torch.tensor([0.1, -0.2, 0.0]), dim=-1) - 1.1
        b = torch.linalg.norm(p - torch.tensor([0.9, 0.6, 0.3]), dim=-1) - 0.5
        return torch.minimum(a, b).unsqueeze(-1)
    siren = rsdf.SIREN(intermediate_size=0)
    names, shapes = fill_procedural(siren)
    kw = dict(r_o=r_o, r_d=r_d, **spec(names, shapes))
    for tag, fn, near, far in (("an", analytic, 1.0, 6.0), ("nn", siren, 0.5, 5.0)):
        pts, hits, dist, _ = rmarch.sphere_march(fn, r_o, r_d, iters=24, eps=1e-3, near=near, far=far)
        kw.update({f"{tag}_sm_pts": pts, f"{tag}_sm_hits": hits, f"{tag}_sm_dist": dist})
        random.seed(7)
        jit = random.random()
        random.seed(7)
        tput, best, lastp, firstn = rmarch.throughput_with_sign_change(fn, r_o, r_d, near, far, batch_size=40)
        kw.update({f"{tag}_tp": tput, f"{tag}_best": best, f"{tag}_last": lastp, f"{tag}_first": firstn})
        random.seed(7)
        pts, hits, best2, tput2 = rmarch.bisect(fn, r_o, r_d, iters=40, near=near, far=far)
        kw.update({f"{tag}_bi_pts": pts, f"{tag}_bi_hits": hits, f"{tag}_bi_tput": tput2})
        kw[f"{tag}_near"], kw[f"{tag}_far"] = np.float64(near), np.float64(far)
    kw["jitter"] = np.float64(jit)
    save("g15_march", **kw)


# ------------------------------------------------------------------ G16 occlusion models (N4)
def g16():
    """Reference src/renderers.py:29-163 occlusion kinds over a src/lights.py Point light and the reference SIREN SDF's
    intersect_mask (src/sdf.py:123-135), all with procedural weights; points on a [2,6,8] grid, some masked out."""
    import random
    import src.renderers as rrend
    import src.lights as rlights
    siren = rsdf.SIREN(intermediate_size=0)
    names, shapes = fill_procedural(siren)
    sdf = rsdf.SDF(siren, rrefl.View(latent_size=0), None, t_near=0.5, t_far=5.0)
    sdf.eval()
    # the procedural SIREN is negative somewhere along every ray: lift its output bias so about half the points see the light
    shift = 0.9
    siren.siren.out.bias.data += shift
    light0 = rlights.Point(center=[1.5, 2.0, -1.0], intensity=[30.0])
    light = next(iter(light0.iter()))  # how src/renderers.py:198 hands lights to the occlusion model
    gen = torch.Generator().manual_seed(16)
    pts = torch.rand(2, 6, 8, 3, generator=gen) * 2.4 - 1.2
    mask = torch.rand(2, 6, 8, generator=gen) > 0.25
    kw = dict(pts=pts, mask=mask, sdf_out_bias_shift=np.float64(shift), center=light.center.detach(), intensity=light.intensity.detach(),
              sdf_param_names=np.array(names), sdf_param_shapes=np.array([",".join(map(str, sh)) for sh in shapes]))
    seen = {}

    def isect(r_o, r_d, near=None, far=None, eps=1e-3):
        vis, tput, _ = sdf.intersect_mask(r_o, r_d, near=near, far=far, eps=eps)
        seen["tput"], seen["far"] = tput, far
        return vis, tput, None
    cases = [("hard", True, {}), ("learned", True, {}), ("learned-const", False, {}),
             ("all-learned", True, {"all_learned_occ_kind": "pos-elaz"}), ("all-learned-pos", False, {"all_learned_occ_kind": "pos"}),
             ("joint-all-const", False, {})]
    for tag, use_mask, extra in cases:
        kind = "all-learned" if tag.startswith("all-learned") else tag
        args = types.SimpleNamespace(occ_kind=kind, **extra)
        occ = rrend.load_occlusion_kind(args, kind, 0)
        occ.eval()
        onames, oshapes = fill_procedural(occ, sigma_by_suffix=4.0)
        for n_, p_ in occ.named_parameters():
            if n_.endswith("alpha"):
                p_.data.fill_(0.3)
        seen.clear()
        random.seed(16)
        d, s = occ(pts, light, isect, mask=mask if use_mask else None, latent=None)
        t = tag.replace("-", "_")
        kw.update({f"{t}_dir": d, f"{t}_spectrum": s, f"{t}_param_names": np.array(onames),
                   f"{t}_param_shapes": np.array([",".join(map(str, sh)) for sh in oshapes])})
        if "tput" in seen:
            kw[f"{t}_tput"], kw[f"{t}_far"] = seen["tput"], np.float64(seen["far"])
        if hasattr(occ, "all_learned_occ"):
            kw[f"{t}_raw_att"] = occ.all_learned_occ.raw_att
    random.seed(16)
    kw["jitter"] = np.float64(random.random())
    # no-shadow lighting (src/renderers.py:29-31)
    d, s = rrend.lighting_wo_isect(pts, light, None, mask=mask)
    kw["none_dir"], kw["none_spectrum"] = d, s
    save("g16_occlusion", **kw)


# ------------------------------------------------------------------ G17 FFJORD divergence estimate (N1)
def g17():
    """Reference utils.div_approx (src/utils.py:467-478) on DynamicNeRF.rigid_dp exactly as runner.py:697-700 calls it:
    training-mode forward (perturbed samples, pts.requires_grad_()), e = randn_like(rigid_dp).  The estimate carries no
    graph (torch.autograd.grad without create_graph): `term_requires_grad` records that the loss term is a constant."""
    size, T, spline = 6, 8, 6
    canon = rnerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted")
    m = rnerf.DynamicNeRF(canonical=canon, spline=spline)
    m.train()
    names, shapes = fill_procedural(m)
    c, focal = cam(POSES[:2], size)
    rays = c.sample_positions(ref_pixel_grid(size, (0, 0, size, size)), size=size)
    times = torch.tensor([0.25, 0.8])
    with torch.enable_grad():
        torch.manual_seed(17)
        out = m((rays, times))
        torch.manual_seed(18)
        div = rutils.div_approx(m.pts, m.rigid_dp)
        torch.manual_seed(18)
        e = torch.randn_like(m.rigid_dp)
        term = (m.canonical.alpha.detach() * div.abs().square()).mean()
    save("g17_ffjord", rays=rays, times=times, pts=m.pts, e=e, div=div, rigid_dp=m.rigid_dp, alpha=m.canonical.alpha,
         term=term, term_requires_grad=np.bool_(term.requires_grad), steps=T, near=2.0, far=6.0, **spec(names, shapes))


# ------------------------------------------------------------------ G18 auxiliary maps (N3: runner.py:511-538, 894-913)
def g18():
    """The reference's OWN visualisation functions on the last forward of DynamicNeRF(spline 6): runner.depth_vis / flow_vis /
    rigidity_vis (runner.py:511-538) and the raw maps of the test() loop (runner.py:894-913), utils.depth_to_normals
    (src/utils.py:421-427).  `depth_vis` as written divides by `(args.far - args.near).clamp(min=0, max=1)` -- the clamp binds
    to the denominator and only exists for tensors -- so it is called with tensor near / far chosen such that far - near = 1,
    where the written expression and the evident intent ((raw - near) / (far - near), clamped to [0, 1]) coincide wherever
    the result lies in [0, 1]; the fixture keeps the raw map too."""
    import argparse
    import runner as rrunner
    size, T, spline = 6, 8, 6
    canon = rnerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted")
    m = rnerf.DynamicNeRF(canonical=canon, spline=spline)
    m.eval()
    names, shapes = fill_procedural(m)
    c, focal = cam(POSES[:2], size)
    rays = c.sample_positions(ref_pixel_grid(size, (0, 0, size, size)), size=size)
    times = torch.tensor([0.25, 0.8])
    out = m((rays, times))
    w = canon.weights
    raw_depth = rnerf.volumetric_integrate(w, canon.ts[:, None, None, None, None])
    args = argparse.Namespace(near=torch.tensor(2.5), far=torch.tensor(3.5), normals_from_depth=True)
    dvis = rrunner.depth_vis(m, args)
    fvis = rrunner.flow_vis(m, args)
    rvis = rrunner.rigidity_vis(m, args)
    save("g18_aux_maps", rays=rays, times=times, out=out, steps=T, near=2.0, far=6.0, vis_near=2.5, vis_far=3.5,
         weights=w, raw_depth=raw_depth, depth_vis=dvis[0], depth_normal_vis=dvis[1],
         depth_normals=rutils.depth_to_normals(raw_depth[0]),
         flow_raw=rnerf.volumetric_integrate(w, m.rigid_dp), flow_vis=fvis[0],
         rigidity_raw=rnerf.volumetric_integrate(w, m.rigidity), rigidity_vis=rvis[0],
         acc=w[:-1].sum(dim=0), **spec(names, shapes))


# ------------------------------------------------------------------ G19 --bg random (src/nerf.py:99-103)
def g19():
    """PlainNeRF(view) with bg = "random": one uniform draw PER RAY (rand_like of the [..., 1] remainder), broadcast over the
    colour channels, times 1 - sum(weights[:-1]).  The draw is replayed from the same seed and stored (the Q13 convention:
    stochastic inputs are explicit tensors on the build's side)."""
    size, T = 6, 8
    m = rnerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted", bg="random")
    m.eval()
    names, shapes = fill_procedural(m)
    c, focal = cam(POSES[:2], size)
    rays = c.sample_positions(ref_pixel_grid(size, (0, 0, size, size)), size=size)
    torch.manual_seed(19)
    out = m(rays)
    summed = 1 - m.weights[:-1].sum(dim=0).unsqueeze(-1)
    torch.manual_seed(19)
    rand = torch.rand_like(summed)
    m.set_bg("black")
    out_black = m(rays)
    assert torch.allclose(out, out_black + rand * summed, atol=1e-6)
    # the bare function on synthetic weights (g3's)
    g3w = rnerf.alpha_from_density(torch.from_numpy(proc_uniform((16, 2, 3, 4), 101, 4.0)), torch.linspace(2, 6, 16),
                                   torch.from_numpy(proc_uniform((2, 3, 4, 3), 103, 1.0)))[1]
    torch.manual_seed(20)
    sky = rnerf.random_color(None, g3w)
    torch.manual_seed(20)
    rand3 = torch.rand_like(sky)
    save("g19_bg_random", rays=rays, out=out, out_black=out_black, rand=rand, steps=T, near=2.0, far=6.0,
         fn_weights=g3w, fn_rand=rand3, fn_sky=sky, **spec(names, shapes))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19"]
    for g in which:
        globals()[g]()
    with open(os.path.join(OUT, "PROVENANCE.txt"), "w") as f:
        f.write("Generated by tools/gen_golden.py from the reference at /root/reference (JulianKnodt/nerf_atlas)\n")
        f.write(f"torch {torch.__version__}, CPU fp32, threads={torch.get_num_threads()}\n")
        f.write(torch.__config__.show())
