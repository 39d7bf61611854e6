"""Procedural, RNG-free tensors shared by the golden generator and the tests.

Values come from a splitmix64 integer hash of (element index, seed) so the same
bytes are produced on any machine without touching a library RNG state
(SURVEY.md §8(c) "Weights for goldens").  TEST INFRASTRUCTURE ONLY.
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def proc_uniform(shape, seed: int, amp: float = 1.0) -> np.ndarray:
    """float32 array of `shape`, uniform in [-amp, amp), deterministic in (shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
        h = _splitmix64(_splitmix64(idx))
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return ((u * 2.0 - 1.0) * amp).astype(np.float32).reshape(shape)


def name_seed(name: str) -> int:
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def proc_param(name: str, shape, salt: int = 0) -> np.ndarray:
    """Procedural value for a parameter/buffer called `name` (reference state_dict key).

    weights [out,in]: U(+-sqrt(6/in)) (keeps activations O(1) through the MLP);
    biases: U(+-0.1); hash tables `embs.*`: U(+-1.7) (unit variance like N(0,1));
    Fourier `basis` [D,F]: U(+-sigma*1.7) with sigma read from salt-free default 8
    (callers that need another sigma scale the result).
    """
    seed = (name_seed(name) + 7919 * salt) & 0x7FFFFFFF
    shape = tuple(shape)
    if "embs" in name:
        amp = 1.7
    elif name.endswith("basis"):
        amp = 1.7
    elif name.endswith("bias"):
        amp = 0.1
    elif len(shape) == 2:
        amp = float(np.sqrt(6.0 / shape[1]))
    else:
        amp = 1.0
    return proc_uniform(shape, seed, amp)
