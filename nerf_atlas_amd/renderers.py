"""Occlusion (shadow) models of SURVEY 8(f) N4: src/renderers.py:29-163.  Each kind maps (points, a light, the SDF's
intersect_mask) to (direction to the light, attenuated spectrum).  The light, the visibility marching, the attenuation
MLPs and the final attenuation are HIP kernels (csrc/march.hip, the fused MLP engine); the surface integrators that call
these models (`direct` / `path`, src/renderers.py:187-226) are relighting and stay out of scope."""
import torch
import torch.nn as nn

from . import ops
from .neural_blocks import SkipConnMLP, FourierEncoder


def _masked(pts, mask): return pts if mask is None else pts[mask]


def _no_grad_only(mod):
    if torch.is_grad_enabled() and any(p.requires_grad for p in mod.parameters()):
        raise NotImplementedError(f"{type(mod).__name__}: training the occlusion model has no HIP backward yet; "
                                  "call it under torch.no_grad()")


def lighting_wo_isect(pts, lights, isect_fn, latent=None, mask=None):
    """src/renderers.py:29-31: no shadows."""
    dir, _, spectrum = lights(_masked(pts, mask), mask=mask)
    return dir, spectrum


class _Occ(nn.Module):
    last_throughput = None

    def _visible(self, isect_fn, pts, dir, **kw):
        visible, tput, _ = isect_fn(r_o=pts, r_d=dir, **kw)
        self.last_throughput = tput
        return visible


class LightingWIsect(_Occ):
    """src/renderers.py:34-46: hard shadows."""

    def __init__(self, latent_size: int = 0): super().__init__()

    def forward(self, pts, lights, isect_fn, latent=None, mask=None):
        pts = _masked(pts, mask)
        dir, dist, spectrum = lights(pts, mask=mask)
        far = dist.max().item() if mask.any() else 6  # like the reference, a hit mask is required here
        visible = self._visible(isect_fn, pts, dir, near=0.1, far=far)
        return dir, ops.occlusion_apply(spectrum, visible, None, 0, 0.0)


class LearnedLighting(_Occ):
    """src/renderers.py:48-68: hidden points keep sigmoid(MLP(x, elaz)) of the light."""

    def __init__(self, latent_size: int = 0):
        super().__init__()
        in_size = 5
        self.attenuation = SkipConnMLP(in_size=in_size, out=1, latent_size=latent_size, num_layers=5, hidden_size=128,
                                       enc=FourierEncoder(input_dims=in_size), init="xavier")

    def forward(self, pts, lights, isect_fn, latent=None, mask=None):
        _no_grad_only(self)
        pts = _masked(pts, mask)
        dir, dist, spectrum = lights(pts, mask=mask)
        far = dist.max().item() if mask.any() else 6
        visible = self._visible(isect_fn, pts, dir, near=2e-3, far=far, eps=1e-3)
        raw = self.attenuation(torch.cat([pts, ops.view_elaz(dir)], dim=-1), latent)
        return dir, ops.occlusion_apply(spectrum, visible, raw, 1, 1.0)


class LearnedConstantSoftLighting(_Occ):
    """src/renderers.py:70-84: hidden points keep sigmoid(alpha) of the light.  The reference's `if mask and mask.any()`
    cannot evaluate a multi-element mask; the intended reading (far = the farthest light when there is a hit, else 6) is
    implemented, and without a mask `far` is 6 exactly like the reference."""

    def __init__(self, latent_size: int = 0):
        super().__init__()
        self.alpha = nn.Parameter(torch.tensor(0.0))

    def hidden_value(self): return float(self.alpha.detach().sigmoid())

    def forward(self, pts, lights, isect_fn, latent=None, mask=None):
        _no_grad_only(self)
        pts = _masked(pts, mask)
        dir, dist, spectrum = lights(pts, mask=mask)
        far = dist.max().item() if (mask is not None and mask.any()) else 6
        visible = self._visible(isect_fn, pts, dir, near=1e-2, far=far, eps=1e-3)
        return dir, ops.occlusion_apply(spectrum, visible, None, 0, self.hidden_value())


def just_pos(pos, dir): return pos
def pos_elaz(pos, dir): return torch.cat([pos, ops.view_elaz(dir)], dim=-1)


all_learned_occ_kinds = {"pos": (just_pos, 3), "pos-elaz": (pos_elaz, 5)}


class AllLearnedOcc(_Occ):
    """src/renderers.py:96-121: a learned ambient-occlusion-like attenuation upshifted_sigmoid(MLP) everywhere."""

    def __init__(self, latent_size: int = 0, kind="pos"):
        super().__init__()
        self.component_fn, in_size = all_learned_occ_kinds[kind]
        self.attenuation = SkipConnMLP(in_size=in_size, out=1, latent_size=latent_size,
                                       enc=FourierEncoder(input_dims=in_size), num_layers=6, hidden_size=256, init="xavier")

    @property
    def all_learned_occ(self): return self

    def encode_raw(self, pts, dir, latent):
        self.raw_att = self.attenuation(self.component_fn(pts, dir), latent)
        return self.raw_att

    def encode(self, pts, dir, latent):
        return ops.sigmoid(self.encode_raw(pts, dir, latent), "upshifted")

    def forward(self, pts, lights, isect_fn, latent=None, mask=None):
        _no_grad_only(self)
        pts = _masked(pts, mask)
        dir, _, spectrum = lights(pts, mask=mask)
        return dir, ops.occlusion_apply(spectrum, None, self.encode_raw(pts, dir, latent), 2, 1.0)


class JointLearnedConstOcc(_Occ):
    """src/renderers.py:123-147: AllLearnedOcc times the constant soft shadow."""

    def __init__(self, latent_size: int = 0, alo: AllLearnedOcc = None, lcsl: LearnedConstantSoftLighting = None):
        if alo is None: alo = AllLearnedOcc(latent_size=latent_size)
        assert isinstance(alo, AllLearnedOcc), "Must pass an instance of AllLearnedOcc"
        if lcsl is None: lcsl = LearnedConstantSoftLighting(latent_size=latent_size)
        assert isinstance(lcsl, LearnedConstantSoftLighting), "Must pass an instance of LearnedConstantSoftLighting"
        super().__init__()
        self.alo, self.lcsl = alo, lcsl

    @property
    def all_learned_occ(self): return self.alo

    def forward(self, pts, lights, isect_fn, latent=None, mask=None):
        if mask is not None: raise NotImplementedError("TODO did not implement handling mask")
        _no_grad_only(self)
        dir, dist, spectrum = lights(pts, mask=mask)
        far = dist.max().item() if isinstance(dist, torch.Tensor) else dist
        raw = self.alo.encode_raw(pts, dir, latent)
        visible = self._visible(isect_fn, pts, dir, near=1e-1, far=far, eps=1e-3)
        return dir, ops.occlusion_apply(spectrum, visible, raw, 2, self.lcsl.hidden_value())


# src/renderers.py:149-156
occ_kinds = {None: lambda **kwargs: lighting_wo_isect, "hard": LightingWIsect, "learned": LearnedLighting,
             "learned-const": LearnedConstantSoftLighting, "all-learned": AllLearnedOcc,
             "joint-all-const": JointLearnedConstOcc}


def load_occlusion_kind(args, kind=None, latent_size: int = 0):
    """src/renderers.py:158-167."""
    con = occ_kinds.get(kind, -1)
    if con == -1: raise NotImplementedError(f"load occlusion: {args.occ_kind}")
    kwargs = {"latent_size": latent_size}
    if kind == "all-learned":
        k = getattr(args, "all_learned_occ_kind", None)
        kwargs["kind"] = k if k is not None else "pos-elaz"
    return con(**kwargs)


def load(args, shape, light_and_refl):
    """src/renderers.py:11-26: the surface integrators themselves are relighting."""
    if args.integrator_kind is None: return None
    raise NotImplementedError(f"integrator '{args.integrator_kind}' is the relighting path (out of scope); the occlusion "
                              "models it uses are available through load_occlusion_kind")
