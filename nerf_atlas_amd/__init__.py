"""nerf_atlas_amd -- MI355X (gfx950) native NeRF volume-rendering hot path.

Host code mirrors the operator/plugin interface of JulianKnodt/nerf_atlas for the path
ray-gen -> stratified sampling -> encode -> SkipConnMLP -> alpha-composite; all
arithmetic runs in hand-written HIP kernels behind the C ABI of include/nerf_atlas_amd.h.
"""
from . import config  # noqa: F401

__version__ = "0.1.0"
