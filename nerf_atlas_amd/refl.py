"""Colour heads on the hot path (src/refl.py:17-49,100-122,190-290,733-751): View, Positional, PosLinearView.
The relighting heads of the reference (Basic, Diffuse, CookTorrance, Rusin*, ...) keep their registry keys and
raise NotImplementedError (out of scope)."""
import torch
import torch.nn as nn

from . import autograd as ag
from . import ops
from .neural_blocks import HashEncoder, SkipConnMLP
from .utils import load_sigmoid


class IdentitySpace(nn.Module):
    def forward(self, x): return x
    @property
    def dims(self): return 3


class NoSpace(nn.Module):
    @property
    def dims(self): return 0


class Reflectance(nn.Module):
    """src/refl.py:100-122."""

    def __init__(self, act="thin", latent_size: int = 0, out_features: int = 3, bidirectional: bool = True, normal=None,
                 light=None, space=None):
        super().__init__()
        self.latent_size = latent_size
        self.out_features = out_features
        self.bidirectional = bidirectional
        self.act = load_sigmoid(act)
        self.act_kind = act if isinstance(act, str) else None  # (the fused renderers apply it in-kernel by name)

    @property
    def can_use_normal(self): return False

    @property
    def can_use_light(self): return False


def _normalize(v):
    """F.normalize(view, dim=-1) on the GPU; one normalisation per RAY when the directions are broadcast along the
    sample axis."""
    if v.dim() > 1 and v.stride(0) == 0:
        return ops.normalize3(v[0].contiguous()).unsqueeze(0).expand(v.shape)
    return ops.normalize3(v.contiguous())


class View(Reflectance):
    """src/refl.py:190-207: act(mlp([x | elaz(view)], latent)); 4x256, sin activations, siren init."""

    def __init__(self, space=None, view="elaz", **kwargs):
        super().__init__(**kwargs)
        assert view == "elaz"
        self.mlp = SkipConnMLP(in_size=5, out=self.out_features, latent_size=self.latent_size, num_layers=4,
                               hidden_size=256, init="siren", activation=torch.sin)

    def forward(self, x, view, normal=None, light=None, latent=None):
        if (view.dim() == 3 and view.stride(0) == 0 and x.shape == view.shape and x.is_cuda and x.dtype == torch.float32
                and x.is_contiguous() and not view.requires_grad):
            # [T, R, 3] points, directions broadcast along the sample axis: the rows [x | elev, azim] by ONE kernel (round 5)
            from . import autograd as ag
            return self.act(self.mlp(ag.ViewInputFn.apply(x, view[0].contiguous()), latent))
        if view.dim() > 1 and view.stride(0) == 0:
            # directions broadcast along the sample axis (r_d.unsqueeze(0).expand_as(pts)): one elev/azim per RAY
            v = ops.view_elaz(view[0].contiguous()).unsqueeze(0).expand(view.shape[:-1] + (2,))
        else:
            v = ops.view_elaz(view.contiguous())
        return self.act(self.mlp(torch.cat([x, v], dim=-1), latent))


class Positional(Reflectance):
    """src/refl.py:230-245."""

    def __init__(self, space=None, **kwargs):
        super().__init__(**kwargs)
        self.mlp = SkipConnMLP(in_size=3, out=self.out_features, latent_size=self.latent_size, enc=HashEncoder(),
                               num_layers=5, hidden_size=256)

    def forward(self, x, view, normal=None, light=None, latent=None):
        return self.act(self.mlp(x, latent))


class PosLinearView(Reflectance):
    """src/refl.py:248-290 (view='raw')."""

    def __init__(self, space=None, view="raw", intermediate_size=64, **kwargs):
        super().__init__(**kwargs)
        assert view == "raw"
        self.im = intermediate_size
        self.pos = SkipConnMLP(in_size=3, out=self.out_features + self.im, latent_size=self.latent_size,
                               enc=HashEncoder(input_dims=3), num_layers=2, hidden_size=256)
        self.view = SkipConnMLP(in_size=6, out=1, latent_size=self.latent_size + self.im, num_layers=2,
                                hidden_size=128, init="siren", activation=torch.sin)

    def forward(self, x, view, normal=None, light=None, latent=None):
        if hasattr(latent, "tensor") and not torch.is_tensor(latent):
            latent = latent.tensor()  # lazy IPE latent (utils.MipLatent): this head concatenates it, so materialise
        pos_all = self.act(self.pos(x, latent))  # [..., out_features + im]
        intermediate = pos_all[..., self.out_features:]
        view_latent = intermediate if latent is None else torch.cat([latent, intermediate], dim=-1)
        raw = self.view(torch.cat([x, _normalize(view)], dim=-1), view_latent)
        # (sigmoid(raw)/2 + 0.5) * pos_all[..., :out]
        if ag.needs_grad(raw, pos_all):
            return ag.PosLinearCombineFn.apply(raw, pos_all, self.out_features)
        return ops.pos_linear_combine(raw, pos_all, self.out_features)


def _out_of_scope(name):
    def cons(*a, **k):
        raise NotImplementedError(f"refl kind '{name}' is a relighting head outside the volume-rendering hot path")
    return cons


# src/refl.py:733-751: same keys
refl_kinds = {
    "pos": Positional, "view": View, "pos-linear-view": PosLinearView,
    **{k: _out_of_scope(k) for k in ["view-light", "basic", "diffuse", "cook-torrance", "rusin", "rusin-helmholtz",
                                     "sph-har", "fourier", "weighted"]},
}


def load(args, refl_kind: str, space_kind: str, latent_size: int):
    """src/refl.py:17-49 (the light / weighted branches are out of scope)."""
    if space_kind not in ("identity", "surface", "none"):
        raise NotImplementedError()
    cons = refl_kinds.get(refl_kind, None)
    if cons is None:
        raise NotImplementedError(f"refl kind: {refl_kind}")
    if getattr(args, "light_kind", None) is not None:
        raise NotImplementedError("lights are outside the volume-rendering hot path")
    return cons(latent_size=latent_size, act=args.sigmoid_kind, out_features=args.feature_space,
                normal=getattr(args, "normal_kind", None), bidirectional=getattr(args, "refl_bidirectional", True))
