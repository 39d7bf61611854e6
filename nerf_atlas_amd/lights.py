"""Lights of the surface path that the occlusion models of SURVEY 8(f) N4 sit on (src/lights.py).  Only the point light
is part of the row; the learned `Field` light belongs to the relighting integrators (out of scope)."""
import torch
import torch.nn as nn

from . import ops


class Light(nn.Module):
    """src/lights.py:23-29."""

    def __getitem__(self, _v): return self

    def forward(self, x): raise NotImplementedError()


class Point(Light):
    """src/lights.py:69-132: a batch of point lights `center` [n, L, 3] / `intensity` [n, L, 3]; `iter()` yields the L
    lights one by one (center [n, 3]) the way src/renderers.py:198 consumes them; forward returns
    (unit direction to the light, distance, spectrum = intensity / (4 pi dist^2)) from one HIP kernel."""

    def __init__(self, center=[0, 0, 0], train_center=False, intensity=[1], train_intensity=False, distance_decay=True):
        super().__init__()
        if not torch.is_tensor(center):
            center = torch.tensor([[center]], dtype=torch.float)
        self.center = nn.Parameter(center.detach().clone().float(), requires_grad=train_center)
        self.train_center = train_center
        if not torch.is_tensor(intensity):
            if len(intensity) == 1: intensity = intensity * 3
            intensity = torch.tensor(intensity, dtype=torch.float).expand_as(self.center)
        self.intensity = nn.Parameter(intensity.detach().clone().float().contiguous(), requires_grad=train_intensity)
        self.train_intensity = train_intensity
        self.distance_decay = distance_decay
        self.curr_idx = 0

    def set_idx(self, v): self.curr_idx = v

    def expand(self, n: int):
        return Point(center=self.center.repeat(n, 1, 1), intensity=self.intensity.repeat(n, 1, 1),
                     train_center=self.train_center, train_intensity=self.train_intensity,
                     distance_decay=self.distance_decay).to(self.center.device)

    def iter(self):
        for i in range(self.center.shape[1]):
            yield Point(center=self.center[:, i, :], intensity=self.intensity[:, i, :], train_center=self.train_center,
                        train_intensity=self.train_intensity, distance_decay=self.distance_decay).to(self.center.device)

    @property
    def supports_idx(self): return self.center.shape[0] > 1

    def _select(self, t, x, mask):
        """src/lights.py:119-121,127-129: t[curr_idx, None, None, :] broadcast over the batch, masked like the points"""
        loc = t[self.curr_idx, None, None, :]
        if loc.dim() < 4: loc = loc.unsqueeze(0)
        if mask is not None:
            if loc.shape[0] == 1 and loc.numel() == 3:
                return loc.reshape(3)        # one light for every point: no expansion needed
            return loc.expand(tuple(mask.shape) + (3,))[mask]
        if loc.numel() == 3:
            return loc.reshape(3)
        return loc.expand(x.shape).contiguous()

    def forward(self, x, mask=None):
        if self.train_center or self.train_intensity:
            if torch.is_grad_enabled():
                raise NotImplementedError("trainable point lights have no HIP backward (relighting path, out of scope)")
        loc = self._select(self.center.data, x, mask)
        intn = self._select(self.intensity.data, x, mask)
        return ops.point_light(x, loc, intn, self.distance_decay)


def _field(**kwargs):
    raise NotImplementedError("light kind 'field' belongs to the relighting integrators (SURVEY 2 row 11, out of scope)")


# src/lights.py:134-139
light_kinds = {"field": _field, "point": Point, "dataset": lambda **kwargs: None, None: None}


def load(args):
    """src/lights.py:10-21."""
    cons = light_kinds.get(args.light_kind, None)
    if cons is None: raise NotImplementedError(f"light kind: {args.light_kind}")
    kwargs = {}
    if args.light_kind == "point":
        kwargs["center"] = args.point_light_position[:3]
        kwargs["intensity"] = [args.light_intensity]
    return cons(**kwargs)
