"""Cameras with the reference's interface (src/cameras.py:10-66,177-223); rays come from the HIP kernel."""
import torch
import torch.nn as nn

from . import ops, utils


class Camera(nn.Module):
    def __init__(self):
        super().__init__()

    def sample_positions(self, positions):
        raise NotImplementedError()


def _crop_of(position_samples, size):
    """The reference passes a slice of the integer pixel grid (runner.py:495-503, positions[r,c]=(c,r)).
    Recover (top, left, h, w) from it so the kernel can regenerate the exact same coordinates."""
    h, w, _ = position_samples.shape
    if h == 0 or w == 0:
        return 0, 0, h, w
    first = position_samples[0, 0].tolist()
    l, t = int(round(first[0])), int(round(first[1]))
    return t, l, h, w


class NeRFCamera(Camera):
    def __init__(self, cam_to_world: torch.Tensor = None, focal: float = None, near: float = None, far: float = None):
        super().__init__()
        self.cam_to_world = nn.Parameter(cam_to_world, requires_grad=False)
        self.focal = focal
        self.near = near
        self.far = far

    def __len__(self):
        return self.cam_to_world.shape[0]

    @classmethod
    def identity(cls, batch_size: int):
        c2w = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0.]]).unsqueeze(0).expand(batch_size, 3, 4)
        return cls(cam_to_world=c2w.contiguous(), focal=0.5)

    def __getitem__(self, v):
        return NeRFCamera(cam_to_world=self.cam_to_world[v], focal=self.focal)

    def sample_positions(self, position_samples, size: int, with_noise=False, noise: torch.Tensor = None):
        """position_samples: [h,w,2] slice of the pixel grid (or a (t,l,h,w) crop tuple).  `noise` [h,w,2]
        uniform draws replace the reference's global-RNG rand_like; drawn here if jitter is on and none given."""
        crop = position_samples if isinstance(position_samples, tuple) else _crop_of(position_samples, size)
        if with_noise and noise is None:
            dev = self.cam_to_world.device  # u first, then v: the reference's draw order (src/cameras.py:55-58)
            noise = torch.cat([utils.rand((crop[2], crop[3], 1), dev), utils.rand((crop[2], crop[3], 1), dev)], dim=-1)
        return ops.raygen(self.cam_to_world.data, self.focal, size, crop, noise, float(with_noise or 0.0))


class DTUCamera(Camera):
    def __init__(self, pose: torch.Tensor = None, intrinsic: torch.Tensor = None):
        super().__init__()
        self.pose = nn.Parameter(pose, requires_grad=False)
        self.intrinsic = nn.Parameter(intrinsic, requires_grad=False)

    def __len__(self):
        return self.pose.shape[0]

    def __getitem__(self, v):
        return DTUCamera(pose=self.pose[v], intrinsic=self.intrinsic[v])

    def sample_positions(self, position_samples, size: int = 512, with_noise: bool = False):
        crop = position_samples if isinstance(position_samples, tuple) else _crop_of(position_samples, size)
        return ops.raygen_dtu(self.pose.data, self.intrinsic.data, size, crop)
