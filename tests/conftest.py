import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        if v.dtype.kind in "fiub":
            out[k] = torch.from_numpy(v)
        else:
            out[k] = v
    return out


def golden_params(g, sigma=None):
    """Regenerate the procedural weights a fixture was made with (oracle/procedural.py)."""
    from oracle.procedural import proc_param
    params = {}
    for name, shp in zip(g["param_names"].tolist(), g["param_shapes"].tolist()):
        shape = tuple(int(s) for s in shp.split(",")) if shp else ()
        v = torch.from_numpy(proc_param(name, shape))
        if name.endswith("basis"):
            s = sigma if sigma is not None else (16.0 if ("sdf" in name or "underlying" in name) else 32.0)
            v = v * s
        params[name] = v
    return params


@pytest.fixture(scope="session")
def golden():
    return load_golden
