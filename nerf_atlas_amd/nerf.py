"""Model-variant plugins of the hot path with the reference's protocol (src/nerf.py):
TinyNeRF, PlainNeRF, VolSDF, DynamicNeRF(spline) + the registries model_kinds / dyn_model_kinds / load_nerf /
load_dyn.  Every forward runs HIP kernels only; PlainNeRF with the View head takes the fully fused renderer.

Protocol kept for callers (runner.py): forward(rays) / forward((rays, times)), from_pts(...), .nerf, .refl,
.set_refl, .intermediate_size, .total_latent_size(), .set_bg, .set_sigmoid, .steps/.t_near/.t_far and, after a
forward, .ts, .alpha, .weights (+ .pts/.dp/.rigidity/.rigid_dp for dynamic models, .scale_post_act for VolSDF).
"""
import functools
import os

import torch
import torch.nn as nn

from . import autograd as ag
from . import config, ops
from . import refl
from .neural_blocks import HashEncoder, SkipConnMLP
from . import utils
from .utils import load_mip, load_sigmoid


# ------------------------------------------------------------------------------------------------- operators
_f16x_depth = 0


def _f16x_policy(fn):
    """config.f16x_on_saturation == "rerender_bf16x3": a forward whose f16x launch was flagged by the range guard
    (ops.F16xSaturated; csrc/render_ls.hip g_lsx_saturated) is rendered again in bf16x3, whose operands have fp32's range.  Only
    the outermost decorated call catches (DynamicNeRF.forward -> canonical.from_pts is one render)."""
    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        global _f16x_depth
        if _f16x_depth or config.f16x_on_saturation != "rerender_bf16x3" or config.precision != "f16x":
            return fn(self, *args, **kwargs)
        _f16x_depth += 1
        try:
            try:
                return fn(self, *args, **kwargs)
            except ops.F16xSaturated:
                with config.precision_as("bf16x3"):
                    return fn(self, *args, **kwargs)
        finally:
            _f16x_depth -= 1
    return wrapped


def compute_ts(rays, near, far, steps, lindisp=False, perturb: float = 0, rand=None):
    """src/nerf.py:29-47.  `rand` [steps] replaces the global-RNG draw; drawn here if perturb>0 and none given."""
    r_o, r_d = rays.split([3, 3], dim=-1)
    if perturb > 0 and rand is None:
        rand = utils.rand((steps,), rays.device)
    ts, mids = ops.compute_ts(near, far, steps, rays.device, lindisp, perturb, rand)
    return r_o, r_d, ts, mids


def compute_pts_ts(rays, near, far, steps, lindisp=False, perturb: float = 0, rand=None):
    """src/nerf.py:50-55."""
    r_o, r_d, ts, mids = compute_ts(rays, near, far, steps, lindisp, perturb, rand)
    pts = ops.compute_pts(rays, ts)
    return pts, ts, r_o, r_d, mids


def alpha_from_density(density, ts, r_d, softplus: bool = True):
    """src/nerf.py:60-73 (via the compositing kernel with a dummy 1-channel feature)."""
    rays = torch.cat([torch.zeros_like(r_d), r_d], dim=-1)
    feat = torch.ones(density.shape + (1,), device=density.device)
    _, alpha, weights = ops.composite(density, feat, ts, rays, softplus=softplus)
    return alpha, weights


def volumetric_integrate(weights, other):
    """src/nerf.py:79-80."""
    return ops.integrate(weights, other)


def black(_elaz_r_d, _weights): return 0
def white(_, weights): return 1 - weights[:-1].sum(dim=0).unsqueeze(-1)


def random_color(_elaz_r_d, weights, rand=None):
    """src/nerf.py:101-103: ONE uniform draw per ray (rand_like of the [..., 1] remainder) times the remainder; the draw is an
    explicit tensor here (utils.rand: the replayable random source of Q13) and a kernel does the arithmetic."""
    shape = tuple(weights.shape[1:]) + (1,)
    if rand is None: rand = utils.rand(shape, weights.device)
    return ops.sky_random(weights, rand, torch.zeros(shape, device=weights.device))


# src/nerf.py:104-109 (same keys; "mlp" is broken in the reference: Q12)
sky_kinds = {"black": black, "white": white, "mlp": "MLP_MARKER", "random": random_color}


def cat_not_none(a, b, dim=-1):
    if isinstance(a, utils.MipLatent):  # lazy IPE latent: stays lazy, the other columns ride along
        return a if b is None else a.with_rest(b)
    return a if b is None else (b if a is None else torch.cat([a, b], dim=dim))


# ------------------------------------------------------------------------------------------------- CommonNeRF
class CommonNeRF(utils.PackedCacheMixin, nn.Module):
    """src/nerf.py:147-276."""

    def __init__(self, r=None, steps: int = 64, fine_steps: int = 32, t_near: float = 0, t_far: float = 1,
                 density_std: float = 0.01, noise_std: int = 1e-2, mip=None, instance_latent_size: int = 0,
                 per_pixel_latent_size: int = 0, per_point_latent_size: int = 0, intermediate_size: int = 32,
                 sigmoid_kind: str = "thin", bg: str = "black"):
        super().__init__()
        self.t_near, self.t_far, self.steps, self.fine_steps = t_near, t_far, steps, fine_steps
        self.mip = mip
        assert instance_latent_size == 0 and per_pixel_latent_size == 0 and per_point_latent_size == 0, \
            "instance/per-pixel/per-point latents are outside the 5 configs of the hot path"
        self.per_pixel_latent_size = self.instance_latent_size = self.per_pt_latent_size = 0
        try: self.intermediate_size = intermediate_size
        except AttributeError: ...
        self.alpha = self.weights = self.ts = self.ts_ray = None
        self.noise_std = 0.2
        self._init_packed_hooks()
        self.set_bg(bg)
        if r is not None: self.refl = r(self.total_latent_size())
        self.set_sigmoid(sigmoid_kind)

    def set_bg(self, bg="black"):
        if bg not in sky_kinds: raise NotImplementedError(bg)
        if bg == "mlp":
            raise NotImplementedError("bg 'mlp' is broken in the reference (SURVEY Q12); black|white|random")
        self.bg = bg
        self.sky_color = sky_kinds[bg]
        self.bg_rand = None  # bg "random": the per-ray draw of the last forward ([..., 1])

    def set_sigmoid(self, kind="thin"):
        act = load_sigmoid(kind)
        self.sigmoid_kind = kind
        self.feat_act = act
        if hasattr(self, "refl") and self.refl is not None:
            self.refl.act = act
            self.refl.act_kind = kind  # (the name the fused paths go by: it used to keep the constructor's default here)

    def total_latent_size(self) -> int: return self.mip_size()

    def set_refl(self, r):
        if hasattr(self, "refl"): self.refl = r

    @property
    def nerf(self): return self

    def mip_size(self): return 0 if self.mip is None else self.mip.size() * 6

    def mip_latent(self, rays, ts):
        """the IPE latent as a lazy handle (utils.MipLatent): generated inside the fused MLP kernels"""
        return None if self.mip is None else self.mip.lazy(rays, ts)

    def mip_encoding(self, rays, ts):
        """src/nerf.py:256-261, intended layout; rays [B,H,W,6] of one crop."""
        return None if self.mip is None else self.mip(rays, ts)

    def _perturb(self): return 1 if self.training else 0

    def _kernel_bg(self):
        """what the fused one-kernel renderers composite against: the random background is added behind them (`_finish_sky`)"""
        return "black" if self.bg == "random" else self.bg

    def _finish_sky(self, out):
        """bg "random" behind a fused renderer that composited against black and kept its weights (src/nerf.py:99-103)"""
        if self.bg == "random":
            self.bg_rand = utils.rand(tuple(out.shape[:-1]) + (1,), out.device)
            ops.sky_random(self.weights, self.bg_rand, out)
        return out

    def _composite(self, density, feat, ts, rays, softplus=True, with_sky=True):
        bg = self.bg if with_sky else "black"
        rand = None
        if bg == "random":
            rand = self.bg_rand = utils.rand(tuple(rays.shape[:-1]) + (1,), rays.device)
        if torch.is_grad_enabled() and (density.requires_grad or feat.requires_grad):
            from .autograd import CompositeFn
            out, self.alpha, self.weights = CompositeFn.apply(density.contiguous(), feat.contiguous(), ts, rays, softplus, bg, rand)
            return out
        out, self.alpha, self.weights = ops.composite(density, feat, ts, rays, softplus=softplus, bg=bg, rand=rand)
        return out


# ------------------------------------------------------------------------------------------------- TinyNeRF
class TinyNeRF(CommonNeRF):
    """src/nerf.py:278-305 with the intended semantics density = estim[...,0] (the reference class cannot be
    constructed at HEAD: SURVEY header table)."""

    def __init__(self, out_features: int = 3, **kwargs):
        super().__init__(**kwargs)
        self.estim = SkipConnMLP(in_size=3, out=1 + out_features, latent_size=self.total_latent_size(), num_layers=6,
                                 hidden_size=256, init="xavier")

    def _fusable(self):
        """one-kernel inference on the layer-synchronous engine (csrc/render_ls.hip, MODEL 1)"""
        wants_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        return (config.engine == "ls" and self.mip is None and self.total_latent_size() == 0
                and self.estim.out.out_features == 4 and not self.training and not wants_grad)

    def packed_ls(self, precision: str):
        """Weight stream of the layer-synchronous renderer (cached, re-packed when any parameter changed)."""
        lin = self.estim._linears()
        stamp = utils.pack_stamp(lin)
        cache = self.__dict__.setdefault("_packed_ls", {})
        hit = cache.get(precision)
        if hit is None or stamp is None or hit[0] != stamp:
            cache[precision] = (stamp, ops.render_tiny_ls_pack(precision, [l.weight.data for l in lin], [l.bias.data for l in lin]))
        return cache[precision][1]

    @_f16x_policy
    def forward(self, rays, want_weights: bool = True):
        if self._fusable():
            _, _, self.ts, _ = compute_ts(rays, self.t_near, self.t_far, self.steps)
            prec = config.kernel_precision(has_f16x=True)
            out, self.alpha, self.weights = ops.render_tiny_ls(rays, self.ts, self.packed_ls(prec), prec, self.sigmoid_kind,
                                                                self._kernel_bg(), want_weights or self.bg == "random")
            return self._finish_sky(out)
        pts, self.ts, r_o, r_d, _ = compute_pts_ts(rays, self.t_near, self.t_far, self.steps, perturb=self._perturb())
        return self.from_pts(pts, self.ts, r_o, r_d, rays=rays)

    def from_pts(self, pts, ts, r_o, r_d, refl_latent=None, rays=None):
        if rays is None: rays = torch.cat([r_o, r_d], dim=-1)
        if self._fusable() and refl_latent is None and not ag.needs_grad(pts):
            # explicit sample positions (a deformation field in front of TinyNeRF) through the same kernel
            prec = config.kernel_precision(has_f16x=True)
            out, self.alpha, self.weights = ops.render_tiny_ls(rays.contiguous(), ts, self.packed_ls(prec), prec, self.sigmoid_kind,
                                                                self._kernel_bg(), True, pts=pts.contiguous())
            return self._finish_sky(out)
        latent = self.mip_encoding(rays, ts)
        o = self.estim(pts, latent)
        density, feats = o[..., 0].contiguous(), o[..., 1:].contiguous()
        return self._composite(density, self.feat_act(feats), ts, rays)


# ------------------------------------------------------------------------------------------------- PlainNeRF
class PlainNeRF(CommonNeRF):
    """src/nerf.py:310-361."""

    def __init__(self, out_features: int = 3, **kwargs):
        super().__init__(r=lambda ls: refl.View(out_features=out_features, latent_size=ls + self.intermediate_size),
                         **kwargs)
        self.first = SkipConnMLP(in_size=3, out=1 + self.intermediate_size, latent_size=self.total_latent_size(),
                                 enc=HashEncoder(), num_layers=4, hidden_size=256)

    def _fusable(self, refl_latent=None):
        wants_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        return (isinstance(self.refl, refl.View) and self.mip is None and refl_latent is None
                and self.intermediate_size == 64 and self.refl.out_features == 3 and not self.training
                and not wants_grad)

    def _fusable_head(self, refl_latent=None):
        """PlainNeRF + refl.Positional / refl.PosLinearView as ONE launch of the layer-synchronous engine (csrc/ls_sched_plain_pos.inc,
        ls_sched_plain_plv.inc: MODEL 7 / 8; f16x only): "pos" | "plv" | None.  PosLinearView takes up to three refl_latent columns
        (DynamicNeRF, --dyn-refl-latent)."""
        wants_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if (config.engine != "ls" or config.precision != "f16x" or self.mip is not None or self.intermediate_size != 64
                or self.training or wants_grad or self.refl.out_features != 3 or getattr(self.refl, "act_kind", None) not in ops.SIGMOID):
            return None
        n_rl = 0 if refl_latent is None else refl_latent.shape[-1]
        if type(self.refl) is refl.Positional and n_rl == 0 and self.refl.latent_size == 64:
            return "pos"
        if (type(self.refl) is refl.PosLinearView and n_rl <= 3 and self.refl.latent_size == 64 + n_rl and self.refl.im == 64
                and self.refl.act_kind in ("normal", "thin", "fat", "upshifted")):  # (the kinds MODEL 8 applies to its 67 rows in the kernel)
            return "plv"
        return None

    def packed_head_ls(self, kind: str, precision: str, n_rl: int = 0):
        r = self.refl
        heads = r.mlp._linears() if kind == "pos" else r.pos._linears() + r.view._linears()
        lin = self.first._linears() + heads
        stamp = utils.pack_stamp(lin)
        cache = self.__dict__.setdefault("_packed_head_ls", {})
        key = (kind, precision, n_rl)
        hit = cache.get(key)
        if hit is None or stamp is None or hit[0] != stamp:
            wb = lambda ls: ([l.weight.data for l in ls], [l.bias.data for l in ls])
            if kind == "pos":
                packed = ops.render_plain_pos_ls_pack(precision, wb(self.first._linears()), wb(heads))
            else:
                packed = ops.render_plain_plv_ls_pack(precision, wb(self.first._linears()), wb(heads), n_rl)
            cache[key] = (stamp, packed)
        return cache[key][1]

    def _render_head(self, kind, rays, ts, want_weights, pts=None, refl_latent=None):
        want_weights = want_weights or self.bg == "random"
        r = self.refl
        if kind == "pos":
            return ops.render_plain_pos_ls(rays, ts, self.first.enc.tables(), r.mlp.enc.tables(), self.packed_head_ls("pos", "f16x"),
                                           "f16x", self.sigmoid_kind, self._kernel_bg(), want_weights, pts=pts)
        n_rl = 0 if refl_latent is None else refl_latent.shape[-1]
        return ops.render_plain_plv_ls(rays, ts, self.first.enc.tables(), r.pos.enc.tables(), self.packed_head_ls("plv", "f16x", n_rl),
                                       "f16x", self.sigmoid_kind, self._kernel_bg(), want_weights, pts=pts, refl_latent=refl_latent)

    def _fusable_mip(self, rays):
        """mip + f16x on the layer-synchronous engine (csrc/render_ls.hip MODEL 6): whole crops [B,H,W,6], 16 IPE degrees."""
        wants_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        return (config.engine == "ls" and config.precision == "f16x" and self.mip is not None
                and self.mip.max_deg - self.mip.min_deg == 16 and rays.dim() == 4 and rays.shape[1] >= 2
                and isinstance(self.refl, refl.View) and self.intermediate_size == 64 and self.refl.out_features == 3
                and self.total_latent_size() == 96 and not self.training and not wants_grad)

    def packed_mip_ls(self, precision: str):
        lin = self.first._linears() + self.refl.mlp._linears()
        stamp = utils.pack_stamp(lin)
        cache = self.__dict__.setdefault("_packed_mip_ls", {})
        hit = cache.get(precision)
        if hit is None or stamp is None or hit[0] != stamp:
            wb = lambda m: ([l.weight.data for l in m._linears()], [l.bias.data for l in m._linears()])
            cache[precision] = (stamp, ops.render_plain_mip_ls_pack(precision, wb(self.first), wb(self.refl.mlp)))
        return cache[precision][1]

    def packed_ls(self, precision: str):
        """Weight stream of the layer-synchronous renderer (both MLPs in one buffer; cached, re-packed when any
        parameter changed)."""
        lin = self.first._linears() + self.refl.mlp._linears()
        stamp = utils.pack_stamp(lin)
        cache = self.__dict__.setdefault("_packed_ls", {})
        hit = cache.get(precision)
        if hit is None or stamp is None or hit[0] != stamp:
            wb = lambda m: ([l.weight.data for l in m._linears()], [l.bias.data for l in m._linears()])
            cache[precision] = (stamp, ops.render_ls_pack(precision, wb(self.first), wb(self.refl.mlp)))
        return cache[precision][1]

    def _render_fused(self, rays, ts, want_weights, pts=None):
        """sample -> hash -> first -> elaz -> View -> sigmoid -> composite in ONE kernel (config.engine picks the
        layer-synchronous engine or the register-resident one)."""
        prec = config.kernel_precision(has_f16x=True)
        want_weights = want_weights or self.bg == "random"
        if config.engine == "ls":
            return ops.render_plain_view_ls(rays, ts, self.first.enc.tables(), self.packed_ls(prec), prec,
                                            self.sigmoid_kind, self._kernel_bg(), want_weights, pts=pts)
        _, pf = self.first.packed(prec, "plain_first")
        _, pv = self.refl.mlp.packed(prec, "plain_view")
        return ops.render_plain_view(rays, ts, self.first.enc.tables(), pf, pv, prec, self.sigmoid_kind, self._kernel_bg(),
                                     want_weights, pts=pts)

    def _train_forward_ls(self, rays, ts, pts, r_d):
        """Training (round 6): both networks' forwards as ONE launch of the layer-synchronous engine in the three-product bf16 split
        (csrc/ls_kernel.h MODEL 9) instead of twelve training Linears -- every Linear's output rows are written once for the backward
        pass and never read back by the forward.  Returns (planes [10, N, 256], (density [N], the View MLP's init rows [N, 69]),
        rgb_pre [N, 3], the stacked hash tables), or None when the step does not have the shape the kernel serves (then the
        layer-by-layer forward runs: the same three-product arithmetic with another summation order).
        `config.set_train_forward("layers")` / NA_TRAIN_LS=0 switch it off."""
        N = pts.numel() // 3
        if (config.train_forward != "ls" or config.train_precision != "bf16x3" or not torch.is_grad_enabled() or not pts.is_cuda
                or type(self.refl) is not refl.View or self.mip is not None or self.intermediate_size != 64 or self.refl.out_features != 3
                or self.refl.mlp.last_layer_act or self.first.last_layer_act or self.refl.mlp.latent_size != 64
                or r_d.shape != pts.shape[1:] or ts.dim() != 1 or ts.shape[0] != pts.shape[0]
                or not (8192 <= N <= ops.TRAIN_LS_MAX_ROWS) or os.environ.get("NA_TRAIN_ROWS") == "0"
                or os.environ.get("NA_TRAIN_MLP_FN") == "0" or os.environ.get("NA_TRAIN_FUSED_BWD") == "0"
                or not all(p.requires_grad for p in self.first.parameters()) or not all(p.requires_grad for p in self.refl.mlp.parameters())):
            return None
        enc = self.first.enc
        tables = torch.stack([e.weight for e in enc.embs])  # (differentiable: the encoder's node of this step takes the same tensor)
        with torch.no_grad():
            planes, rows, density, rgb_pre, _ = ops.train_plain_view_ls(rays.reshape(-1, 6), ts, pts.detach(), tables.detach(),
                                                                        self.packed_ls("bf16x3"), self.sigmoid_kind)
        return planes, (density, rows), rgb_pre, tables

    @_f16x_policy
    def forward(self, rays, want_weights: bool = True):
        self.ts_ray = None  # (only forward_coarse_fine integrates over per-ray steps)
        if self._fusable():
            _, _, self.ts, _ = compute_ts(rays, self.t_near, self.t_far, self.steps)
            out, self.alpha, self.weights = self._render_fused(rays, self.ts, want_weights)
            return self._finish_sky(out)
        head = self._fusable_head()
        if head is not None:
            _, _, self.ts, _ = compute_ts(rays, self.t_near, self.t_far, self.steps)
            out, self.alpha, self.weights = self._render_head(head, rays.contiguous(), self.ts, want_weights)
            return self._finish_sky(out)
        if self._fusable_mip(rays):
            # config 3 under f16x: sample -> hash -> IPE -> first -> View -> composite as ONE launch
            _, _, self.ts, _ = compute_ts(rays, self.t_near, self.t_far, self.steps)
            out, self.alpha, self.weights = ops.render_plain_mip_ls(
                rays.contiguous(), self.ts, self.first.enc.tables(), self.packed_mip_ls("f16x"), "f16x", self.mip.kind, float("nan"),
                self.mip.min_deg, self.mip.max_deg, self.sigmoid_kind, self._kernel_bg(), want_weights or self.bg == "random")
            return self._finish_sky(out)
        rand = None
        pts, self.ts, r_o, r_d, _ = compute_pts_ts(rays, self.t_near, self.t_far, self.steps, perturb=self._perturb(),
                                                   rand=rand)
        return self.from_pts(pts, self.ts, r_o, r_d, rays=rays)

    @_f16x_policy
    def forward_coarse_fine(self, rays, steps_fine: int, u=None, want_weights: bool = True):
        """Coarse -> fine rendering (BASELINE config 2 "64 + 128"; INTENDED reading of the reference's dead sample_pdf /
        CoarseFineNeRF, src/nerf.py:548-580, 1745-1779 -- see csrc/basic_ops.hip resample_ts_kernel): a coarse pass over
        `self.steps` shared steps, inverse-cdf resampling of `steps_fine` new positions per ray from its weights (u: None =
        linspace, or [steps_fine, *rays.shape[:-1]] draws), and a fine pass over the union in order -- three launches of the
        layer-synchronous engine's kernels, one network for both passes like the reference's class.  Inference only."""
        if not (self._fusable() and config.engine == "ls"):
            raise NotImplementedError("coarse -> fine rendering runs on the fused layer-synchronous renderer (eval mode, View head)")
        rays = rays.contiguous()
        _, _, ts, _ = compute_ts(rays, self.t_near, self.t_far, self.steps)
        prec = config.kernel_precision(has_f16x=True)
        tables, packed = self.first.enc.tables(), self.packed_ls(prec)
        coarse, _, w = ops.render_plain_view_ls(rays, ts, tables, packed, prec, self.sigmoid_kind, self._kernel_bg(), True)
        self.coarse = coarse
        # `ts` keeps the protocol of every other forward -- the [T] shared (coarse) steps; the per-ray union the fine pass
        # integrated over is `ts_ray` [*batch, T + N] (weights / alpha rows follow IT: render.depth_map knows)
        self.ts = ts
        self.ts_ray = ops.resample_ts(ts, w, steps_fine, u)
        out, self.alpha, self.weights = ops.render_plain_view_ls_rayts(rays, self.ts_ray, tables, packed, prec, self.sigmoid_kind,
                                                                       self._kernel_bg(), want_weights or self.bg == "random")
        return self._finish_sky(out)

    def from_pts(self, pts, ts, r_o, r_d, refl_latent=None, rays=None):
        if rays is None: rays = torch.cat([r_o, r_d], dim=-1).contiguous()
        self.ts_ray = None  # (explicit positions: shared steps -- a stale per-ray row of an earlier coarse -> fine call must not reach render.depth_map)
        if self._fusable(refl_latent) and not ag.needs_grad(pts):
            # explicit sample positions (D-NeRF: spline-warped canonical points) through the same fused kernel
            out, self.alpha, self.weights = self._render_fused(rays, ts, True, pts=pts.contiguous())
            return self._finish_sky(out)
        head = self._fusable_head(refl_latent)
        if head is not None and not ag.needs_grad(pts) and (refl_latent is None or not ag.needs_grad(refl_latent)):
            # explicit sample positions (D-NeRF: warped points, + its per-sample reflectance latent) through the one-launch renderer
            out, self.alpha, self.weights = self._render_head(head, rays.contiguous(), ts, True, pts=pts.contiguous(), refl_latent=refl_latent)
            return self._finish_sky(out)
        if not self.training and not ag.needs_grad(pts, *self.parameters()) and refl_latent is None and self.mip is None:
            utils.note_fallback(f"plain-unfused-{type(self.refl).__name__}-{self.intermediate_size}",
                                f"PlainNeRF with {type(self.refl).__name__} reflectance / intermediate size {self.intermediate_size} is not one of the "
                                "fused renderer's schedules (View head, intermediate 64, 3 channels): inference runs the generic MLP kernels "
                                "+ separate compositing")
        latent = self.mip_latent(rays, ts)  # lazy: generated in the prologues of `first` and of the View MLP
        pre = self._train_forward_ls(rays, ts, pts, r_d) if latent is None and refl_latent is None else None
        first_out = self.first(pts, latent, pre=None if pre is None else (list(pre[0][:5]), None, pre[3]))
        if (ag.needs_grad(first_out) and latent is None and refl_latent is None and type(self.refl) is refl.View and pts.is_cuda
                and not self.refl.mlp.last_layer_act and self.refl.mlp.latent_size == first_out.shape[-1] - 1
                and r_d.shape == pts.shape[1:] and os.environ.get("NA_TRAIN_ROWS") != "0"):
            # training: density | the View MLP's init rows [x, elev, azim | intermediate] by ONE kernel (autograd.PlainHeadFn; slice
            # copies, elaz, expand and two cats before), the network from its rows (SkipConnMLP.forward_rows)
            C = first_out.shape[-1]
            # (with precomputed values the node reads neither positions nor directions: no copy of the direction slice)
            density, rows = ag.PlainHeadFn.apply(first_out.reshape(-1, C), pts.reshape(-1, 3),
                                                 r_d.reshape(-1, 3).contiguous() if pre is None else None, None if pre is None else pre[1])
            density = density.reshape(pts.shape[:-1])
            if self.training and self.noise_std > 0:
                density = density + utils.randn(density.shape, density.device) * self.noise_std
            rgb = self.refl.act(self.refl.mlp.forward_rows(rows, pre=None if pre is None else (list(pre[0][5:]), pre[2]))
                                .reshape(pts.shape[:-1] + (self.refl.out_features,)))
            return self._composite(density, rgb, ts, rays)
        assert pre is None, "the one-launch training forward implies the rows path above (its `first_out` is a placeholder)"
        if ag.needs_grad(first_out):
            density, intermediate = ag.SplitHeadFn.apply(first_out)  # (the slices' gradients written side by side: autograd.py)
        else:
            density, intermediate = first_out[..., 0].contiguous(), first_out[..., 1:]
        if self.training and self.noise_std > 0:
            density = density + utils.randn(density.shape, density.device) * self.noise_std
        view = r_d.unsqueeze(0).expand_as(pts)
        rl = cat_not_none(latent, cat_not_none(intermediate, refl_latent))
        rgb = self.refl(x=pts, view=view, latent=rl)  # `intermediate` is a column slice of first_out: passed by pitch
        return self._composite(density, rgb, ts, rays)


# ------------------------------------------------------------------------------------------------- VolSDF
class VolSDF(CommonNeRF):
    """src/nerf.py:861-1018, volume path only (uniform samples, Laplace density, relu, no sky term)."""

    def __init__(self, sdf, out_features: int = 3, occ_kind=None, integrator_kind="direct", w_transmission: bool = False,
                 scale_softplus: bool = False, **kwargs):
        super().__init__(**kwargs)
        assert occ_kind is None and not w_transmission, "occlusion / transmission are outside the volume path"
        self.sdf = sdf
        self.scale = nn.Parameter(torch.tensor(0.1))
        self.secondary = None
        self.out_features = out_features
        self.scale_softplus = scale_softplus

    @_f16x_policy
    def forward(self, rays):
        pts, self.ts, r_o, r_d, _ = compute_pts_ts(rays, self.t_near, self.t_far, self.steps, perturb=self._perturb())
        return self.from_pts(pts, self.ts, r_o, r_d, rays=rays)

    @property
    def intermediate_size(self): return self.sdf.intermediate_size

    def set_refl(self, r): self.sdf.refl = r

    @property
    def refl(self): return self.sdf.refl

    def _fusable_view(self, refl_latent=None):
        """View head + compositing as one kernel on the layer-synchronous engine (csrc/render_ls.hip, MODEL 2)"""
        r = self.sdf.refl
        wants_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        return (config.engine == "ls" and type(r) is refl.View and r.latent_size == 64 and r.out_features == 3
                and self.sdf.intermediate_size == 64 and getattr(r, "act_kind", None) in ops.SIGMOID and refl_latent is None
                and not self.training and not wants_grad)

    def _fusable_fourier_sdf(self):
        from . import sdf as _sdf
        from .neural_blocks import FourierEncoder
        u = self.sdf.underlying
        if type(u) is not _sdf.MLP or config.precision != "f16x" or config.engine != "ls":
            return False
        m = u.mlp
        return (type(m.enc) is FourierEncoder and m.enc.freqs == 128 and m.enc.input_dims == 3 and len(m.layers) == 6 and m.skip == 3
                and m.init.out_features == 256 and m.out.out_features == 65 and m.latent_size == 0 and m.act_name == "leaky_relu")

    def packed_fourier_sdf_ls(self, precision: str):
        lin = self.sdf.underlying.mlp._linears()
        stamp = utils.pack_stamp(lin)
        cache = self.__dict__.setdefault("_packed_fourier_ls", {})
        hit = cache.get(precision)
        if hit is None or stamp is None or hit[0] != stamp:
            cache[precision] = (stamp, ops.mlp_fourier_ls_pack(precision, [l.weight.data for l in lin], [l.bias.data for l in lin]))
        return cache[precision][1]

    def packed_view_ls(self, precision: str):
        lin = self.sdf.refl.mlp._linears()
        stamp = utils.pack_stamp(lin)
        cache = self.__dict__.setdefault("_packed_view_ls", {})
        hit = cache.get(precision)
        if hit is None or stamp is None or hit[0] != stamp:
            cache[precision] = (stamp, ops.render_view_ls_pack(precision, [l.weight.data for l in lin], [l.bias.data for l in lin]))
        return cache[precision][1]

    def packed_siren_ls(self, precision: str):
        lin = self.sdf.underlying.siren._linears() + self.sdf.refl.mlp._linears()
        stamp = utils.pack_stamp(lin)
        cache = self.__dict__.setdefault("_packed_siren_ls", {})
        hit = cache.get(precision)
        if hit is None or stamp is None or hit[0] != stamp:
            wb = lambda m: ([l.weight.data for l in m._linears()], [l.bias.data for l in m._linears()])
            cache[precision] = (stamp, ops.render_volsdf_siren_ls_pack(precision, wb(self.sdf.underlying.siren), wb(self.sdf.refl.mlp)))
        return cache[precision][1]

    def from_pts(self, pts, ts, r_o, r_d, refl_latent=None, rays=None):
        if rays is None: rays = torch.cat([r_o, r_d], dim=-1).contiguous()
        from . import sdf as _sdf
        if (self._fusable_view(refl_latent) and not ag.needs_grad(pts) and type(self.sdf.underlying) is _sdf.SIREN):
            # the SIREN SDF network fits the layer-synchronous engine too: the whole model is ONE kernel (MODEL 3)
            scale = torch.nn.functional.softplus(self.scale.data) if self.scale_softplus else self.scale.data
            object.__setattr__(self, "scale_post_act", scale)
            prec = config.kernel_precision(has_f16x=True)
            out, self.alpha, self.weights = ops.render_volsdf_siren_ls(rays.contiguous(), ts, scale, self.packed_siren_ls(prec), prec,
                                                                        self.sdf.refl.act_kind, "black", True, pts=pts.contiguous())
            return out
        if self._fusable_view(refl_latent) and not ag.needs_grad(pts):
            # SDF network (fused MLP kernel) -> one kernel for Laplace density, View head and compositing: the colour,
            # density and [x | elev azim] tensors of the operator chain are never materialised
            if self._fusable_fourier_sdf():
                # f16x: the Fourier-MLP SDF network as ONE launch of the layer-synchronous engine (MODEL 5) instead of the
                # 3-product generic kernel; its 256 Fourier features are generated in the kernel
                enc = self.sdf.underlying.mlp.enc
                basis = (enc.basis.data * float(enc.extra_scale)).contiguous() if float(enc.extra_scale) != 1.0 else enc.basis.data
                raw = ops.mlp_fourier_ls(rays.contiguous(), ts, basis, self.packed_fourier_sdf_ls("f16x"), "f16x", pts=pts.contiguous())
            else:
                raw = self.sdf.underlying(pts)
            scale = torch.nn.functional.softplus(self.scale.data) if self.scale_softplus else self.scale.data
            object.__setattr__(self, "scale_post_act", scale)
            prec = config.kernel_precision(has_f16x=True)
            out, self.alpha, self.weights = ops.render_view_ls(rays.contiguous(), ts, raw, scale, self.packed_view_ls(prec), prec,
                                                                self.sdf.refl.act_kind, "black", True, pts=pts.contiguous())
            return out
        sdf_vals, latent = self.sdf.from_pts(pts)
        if ag.needs_grad(sdf_vals, self.scale):
            scale = torch.nn.functional.softplus(self.scale) if self.scale_softplus else self.scale
            density = ag.LaplaceDensityFn.apply(sdf_vals.contiguous(), scale)
        else:
            scale = torch.nn.functional.softplus(self.scale.data) if self.scale_softplus else self.scale.data
            density = ops.laplace_density(sdf_vals.contiguous(), scale)
        object.__setattr__(self, "scale_post_act", scale)  # plain attribute: never a second registration of the Parameter
        if self.sdf.refl.can_use_normal:
            raise NotImplementedError("normal-dependent reflectance needs autograd normals (row N1)")
        view = r_d.unsqueeze(0).expand_as(pts)
        rgb = self.sdf.refl(x=pts, view=view, normal=None, latent=latent)  # column slice of the SDF output: by pitch
        return self._composite(density, rgb, ts, rays, softplus=False, with_sky=False)

    def set_sigmoid(self, kind="thin"):
        if not hasattr(self, "sdf"): return
        self.sigmoid_kind = kind
        self.refl.act = load_sigmoid(kind)
        self.refl.act_kind = kind


# ------------------------------------------------------------------------------------------------- DynamicNeRF
class DynamicNeRF(utils.PackedCacheMixin, nn.Module):
    """src/nerf.py:1209-1303, Bezier-spline deformation (spline > 1), with `refl_latent` columns riding through the spline into the
    canonical model's reflectance (`make dnerf`: --dyn-refl-latent 3).  The delta path (spline = 0) raises at HEAD in the reference
    (SURVEY header table) and raises here too."""

    def __init__(self, canonical: CommonNeRF, spline: int = 0, refl_latent: int = 0):
        super().__init__()
        self.canonical = canonical
        self.spline = spline
        self.refl_latent = max(refl_latent, 0)
        if spline <= 1:
            raise NotImplementedError("DynamicNeRF without a spline is broken in the reference (self.dp unset); use --spline N>1")
        if self.refl_latent > 16:
            raise NotImplementedError("--dyn-refl-latent > 16 (na_bezier_warp_latent carries 1..16 columns)")
        self.spline_n = spline
        # src/nerf.py:1243-1249: with a reflectance latent the network also emits enc_rigidity | spline * refl_latent control rows
        out_dims, enc_layout = spline * 3 + 1, [0, 0]
        if self.refl_latent > 0:
            out_dims += spline * self.refl_latent + 1
            enc_layout = [1, self.refl_latent * spline]
        self.mlp_out_layout = [1, 3 * spline] + enc_layout
        self.delta_estim = SkipConnMLP(in_size=3, out=out_dims, num_layers=5, hidden_size=256, init="xavier",
                                       enc=HashEncoder())
        self.delta_estim.zero_last_layer()
        self._init_packed_hooks()

    @property
    def nerf(self): return self.canonical
    @property
    def refl(self): return self.canonical.refl
    @property
    def sdf(self): return getattr(self.canonical, "sdf", None)
    @property
    def intermediate_size(self): return self.canonical.intermediate_size + self.refl_latent
    def total_latent_size(self): return self.canonical.total_latent_size()
    def set_refl(self, r): self.canonical.set_refl(r)
    def set_bg(self, bg): self.canonical.set_bg(bg)

    @property
    def rigid_dp(self):
        """dp * rigidity (src/nerf.py:1278), only read by the flow visualisation and regularisers: formed on demand so
        the render path carries no extra elementwise pass over the samples."""
        return self.dp * self.rigidity

    def ffjord_div(self, e):
        """utils.div_approx(model.pts, model.rigid_dp) of runner.py:697-699 (src/utils.py:467-478) at the samples of the
        last forward: <e, d(rigid_dp)/d(pts) . e> per sample, e [T,B,H,W,3] the caller's randn draw.  The reference
        contracts a vector-Jacobian product with e; here the same number comes from one forward-mode sweep (hash_jvp ->
        tangent MLP -> ffjord_div kernels).  Like the reference's, the estimate has no graph."""
        est, dest = self.delta_estim.forward_with_direction_tangent(self.pts, e)
        return ops.ffjord_div(est, dest, self._tt, e.contiguous(), self.spline_n).reshape(self.pts.shape[:-1])

    def sum_jacobian_div(self):
        """`utils.divergence(model.pts, model.dp)` of runner.py:694-696 at the samples of the last forward, with its graph.  What
        the reference computes: `autograd(x, field)` differentiates the SUM of the field's three components
        (grad_outputs = ones, src/utils.py:266-277, 461-464) and the result is summed over the coordinates, i.e.
        sum_ij d dp_j / d x_i = sum_j (J . (1,1,1))_j -- the sum of all Jacobian entries, not its trace.  One direction
        tangent with e = (1,1,1) through the hash encoder and the deformation MLP (first-order graph nodes), then the
        spline, which is linear in its control points (the Bezier combination of the control points' tangents, by the warp
        kernel's own differentiable dp output).  The sweep runs in exact fp32 like the FFJORD tangent: the hash features'
        derivatives scale with the grid resolutions and largely cancel in the sum (split-bf16 GEMMs: 3 % of the result).
        [T,B,H,W,1]"""
        from . import config
        e = torch.ones_like(self.pts)
        with config.train_precision_as("fp32"):
            _, dest = self.delta_estim.forward_with_direction_tangent_graph(self.pts, e)
        dest = dest.reshape(*self.pts.shape[:-1], -1)
        _, ddp, _ = ag.BezierWarpFn.apply(dest.contiguous(), self.pts, self._tt, self.spline_n)
        return ddp.sum(dim=-1, keepdim=True)

    def _deformation_ls_mode(self):
        """the deformation network as ONE launch of the layer-synchronous engine (csrc/render_ls.hip MODEL 4) in inference: "bf16x3"
        (the three-product split, the parity default under precisions f16x / bf16x3: config.deformation_engine "ls-bf16x3"), "f16x"
        (the 1.5-product mode, opt-in: config.py says why) or None (the generic fused MLP / the training path)"""
        wants_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        d = self.delta_estim
        if not (config.engine == "ls" and config.deformation_engine != "generic" and not self.training and not wants_grad
                and isinstance(d.enc, HashEncoder) and len(d.layers) == 5 and d.init.out_features == 256 and d.skip == 3
                and d.latent_size == 0 and d.act_name == "leaky_relu"):
            return None
        n_out = d.out.out_features
        if config.precision == "f16x" and config.deformation_engine == "ls" and n_out <= 32:
            return "f16x"
        if config.precision in ("f16x", "bf16x3") and n_out <= 64:
            return "bf16x3"
        return None

    def _fusable_deformation(self):
        """the f16x instance of the above (opt-in)"""
        return self._deformation_ls_mode() == "f16x"

    def packed_deformation_ls(self, precision: str):
        lin = self.delta_estim._linears()
        stamp = utils.pack_stamp(lin)
        cache = self.__dict__.setdefault("_packed_ls", {})
        hit = cache.get(precision)
        if hit is None or stamp is None or hit[0] != stamp:
            cache[precision] = (stamp, ops.mlp_hash_ls_pack(precision, [l.weight.data for l in lin], [l.bias.data for l in lin]))
        return cache[precision][1]

    @_f16x_policy
    def forward(self, rays_t):
        rays, t = rays_t
        c = self.canonical
        self.pts, self.ts, r_o, r_d, _ = compute_pts_ts(rays, c.t_near, c.t_far, c.steps,
                                                        perturb=1 if self.training else 0)
        c.ts = self.ts
        tt = self._tt = t[None, :, None, None].expand(*self.pts.shape[:-1]).contiguous()
        mode = self._deformation_ls_mode()
        if mode is not None:
            est = ops.mlp_hash_ls(rays, self.ts, self.delta_estim.enc.tables(), self.packed_deformation_ls(mode), mode,
                                  self.delta_estim.out.out_features)
        else:
            est = self.delta_estim(self.pts)
        # (the reference hands from_pts a zero-width `enc` when refl_latent == 0, src/nerf.py:1272-1278, 1303: None here)
        enc = None
        if ag.needs_grad(est):
            res = ag.BezierWarpFn.apply(est.contiguous(), self.pts, tt, self.spline_n, self.refl_latent)
        else:
            res = ops.bezier_warp(est, self.pts, tt, self.spline_n, self.refl_latent)
        if self.refl_latent > 0:
            warped, self.dp, self.rigidity, enc = res
        else:
            warped, self.dp, self.rigidity = res
        return c.from_pts(warped, self.ts, r_o, r_d, refl_latent=enc, rays=rays)


    @_f16x_policy
    def render_keyframes(self, rays):
        """src/nerf.py:1305-1319 (runner.py:1019-1039 writes them as keyframe_NN.png): the scene rendered at each Bezier
        control point, canonical.from_pts(pts + p_k * rigidity).  The reference splits the 3*spline_n control
        coordinates into spline_n - 1 chunks, which torch.split rejects (sizes must sum to the dimension); the intended
        one frame per control point is produced here.  Keyframe k goes through the warp kernel with every control point
        set to p_k (the Bernstein weights sum to one, so dp = p_k)."""
        assert self.spline > 0
        c = self.canonical
        self.pts, self.ts, r_o, r_d, _ = compute_pts_ts(rays, c.t_near, c.t_far, c.steps, perturb=0)
        c.ts = self.ts
        est = self.delta_estim(self.pts)
        tt = torch.zeros(self.pts.shape[:-1], device=rays.device)
        frames = []
        for k in range(self.spline_n):
            est_k = torch.cat([est[..., :1]] + [est[..., 1 + 3 * k:4 + 3 * k]] * self.spline_n, dim=-1).contiguous()
            warped, _, self.rigidity = ops.bezier_warp(est_k, self.pts, tt, self.spline_n)
            frames.append(c.from_pts(warped, self.ts, r_o, r_d, rays=rays))
        return frames


# ------------------------------------------------------------------------------------------------- registries
def _experimental(name):
    def cons(*a, **k):
        raise NotImplementedError(f"model kind '{name}' is an experimental variant outside the hot path (SURVEY 2 row 7)")
    return cons


# src/nerf.py:1706-1720 / 1698-1704: same keys
model_kinds = {"tiny": TinyNeRF, "plain": PlainNeRF, "volsdf": VolSDF,
               **{k: _experimental(k) for k in ["ae", "coarse_fine", "mpi", "voxel", "rig", "hist"]}}
dyn_model_kinds = {"plain": DynamicNeRF, **{k: _experimental(k) for k in ["ae", "rig", "long", "voxel"]}}


def load_nerf(args):
    """src/nerf.py:111-145."""
    from .sdf import load as load_sdf
    kwargs = {"mip": load_mip(args), "out_features": args.feature_space, "steps": args.steps, "t_near": args.near,
              "t_far": args.far, "intermediate_size": args.shape_to_refl_size, "sigmoid_kind": args.sigmoid_kind,
              "bg": args.bg}
    cons = model_kinds.get(args.model, None)
    if cons is None: raise NotImplementedError(args.model)
    if args.model == "volsdf":
        kwargs["sdf"] = load_sdf(args, with_integrator=False)
        kwargs["occ_kind"] = getattr(args, "occ_kind", None)
    return cons(**kwargs)


def load_dyn(args, model, device=None):
    """src/nerf.py:1680-1696."""
    cons = dyn_model_kinds.get(args.dyn_model, None)
    if cons is None: raise NotImplementedError(f"Unknown dyn kind: {args.dyn_model}")
    return cons(canonical=model, spline=args.spline, refl_latent=args.dyn_refl_latent)
