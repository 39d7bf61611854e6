"""N2 loaders (src/loaders.py:74-195) against golden outputs of the reference's own loaders on the analytic scene
(tests/golden/g14_loaders.npz, made by tools/gen_golden.py g14) -- bit-exact: it is json + PIL + /255 arithmetic.
CPU only."""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import load_golden

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools.make_scene import make_scene  # noqa: E402
from nerf_atlas_amd import loaders  # noqa: E402


@pytest.fixture(scope="module")
def scenes(tmp_path_factory):
    td = tmp_path_factory.mktemp("scenes")
    d = make_scene(str(td / "static"), size=24, n_train=4, n_test=2) + "/"
    dd = make_scene(str(td / "dyn"), size=24, n_train=5, n_test=2, dynamic=True) + "/"
    tf = json.load(open(dd + "transforms_train.json"))
    tf["frames"] = [tf["frames"][i] for i in (3, 0, 4, 1, 2)]
    for fr in tf["frames"]:
        fr["time"] = fr["time"] * 3.0 - 0.5
    json.dump(tf, open(dd + "transforms_train.json", "w"))
    return d, dd


def test_original_matches_reference_loader(scenes):
    g = load_golden("g14_loaders")
    d, _ = scenes
    for training in (True, False):
        for size in (24, 16):
            for white in (False, True):
                labels, cam, light = loaders.original(d, normalize=False, training=training, size=size, white_bg=white)
                key = f"orig_{'train' if training else 'test'}_{size}_{'w' if white else 'b'}"
                assert light is None
                assert torch.equal(labels, g[key + "_labels"]), key
                assert torch.equal(cam.cam_to_world.data, g[key + "_c2w"]), key
                assert float(cam.focal) == float(g[key + "_focal"])
                assert len(cam) == labels.shape[0]
    labels, cam, _ = loaders.original(d, normalize=True, training=True, size=24, with_mask=True)
    assert labels.shape[-1] == 4 and torch.equal(labels, g["orig_norm_mask_labels"])
    assert torch.equal(cam.cam_to_world.data, g["orig_norm_mask_c2w"])


def test_dnerf_matches_reference_loader(scenes):
    g = load_golden("g14_loaders")
    _, dd = scenes
    for gamma in (False, True):
        (labels, times), cam, _ = loaders.dnerf(dd, training=True, size=24, time_gamma=gamma, white_bg=False)
        assert torch.equal(times, g[f"dnerf_g{int(gamma)}_times"])
        assert torch.equal(labels, g[f"dnerf_g{int(gamma)}_labels"])
        assert torch.equal(cam.cam_to_world.data, g[f"dnerf_g{int(gamma)}_c2w"])
        assert float(times.min()) == 0.0 and float(times.max()) == 1.0 and bool((times[1:] >= times[:-1]).all())
    (labels, times), cam, _ = loaders.dnerf(dd, training=False, size=16, time_gamma=False, white_bg=True)
    assert torch.equal(labels, g["dnerf_test_labels"]) and torch.equal(times, g["dnerf_test_times"])
    assert float(cam.focal) == float(g["dnerf_test_focal"])


def test_load_dispatch_and_errors(scenes):
    d, dd = scenes
    args = types.SimpleNamespace(data=d, data_kind="original", derive_kind=True, model="plain", volsdf_alternate=False,
                                 size=24, bg="black", time_gamma=False)
    labels, cam, _ = loaders.load(args, training=False)
    assert labels.shape == (2, 24, 24, 3)
    args.data, args.data_kind = dd, "dnerf"
    (labels, times), cam, _ = loaders.load(args)
    assert labels.shape == (5, 24, 24, 3) and times.shape == (5,)
    for k in ("nerv_point", "single-video", "pixel-single", "nonsense"):
        args.data_kind = k
        with pytest.raises(NotImplementedError):
            loaders.load(args)
    with pytest.raises(FileNotFoundError):
        loaders.original(d + "missing/", training=True, size=8)


def _rot(a, b, c):
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    return (np.array([[cc, -sc, 0], [sc, cc, 0], [0, 0, 1]]) @ np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
            @ np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]]))


def test_decompose_projection_recovers_K_R_c():
    """No cv2 offline: the restated decomposition is checked by construction, P = s * K [R | -R c]."""
    rng = np.random.default_rng(5)
    for i in range(20):
        K = np.array([[2892.3 + 10 * i, 0.7 * (i % 3), 800.0 + i], [0, 2880.1 - 3 * i, 600.0 - i], [0, 0, 1.0]])
        R = _rot(*rng.uniform(-3, 3, 3))
        c = rng.uniform(-2, 2, 3)
        scale = [1.0, 3.7, -2.2][i % 3]
        P = scale * K @ np.concatenate([R, (-R @ c)[:, None]], axis=1)
        K2, R2, c2 = loaders.decompose_projection(P)
        K2 = K2 / K2[2, 2]
        assert np.allclose(K2, K, rtol=1e-9, atol=1e-6)
        assert np.allclose(R2, R, atol=1e-9) and abs(np.linalg.det(R2) - 1) < 1e-9
        assert np.allclose(c2[:3] / c2[3], c, atol=1e-9)


def test_dtu_loader_on_synthetic_directory(tmp_path):
    from PIL import Image
    n, size = 3, 20
    os.makedirs(tmp_path / "image")
    os.makedirs(tmp_path / "mask")
    rng = np.random.default_rng(9)
    cams = {}
    Ks, cs, Rs = [], [], []
    for i in range(n):
        Image.fromarray(rng.integers(0, 255, (size, size, 3), dtype=np.uint8), "RGB").save(tmp_path / "image" / f"{i:06}.png")
        Image.fromarray((rng.integers(0, 2, (size, size, 3)) * 255).astype(np.uint8), "RGB").save(tmp_path / "mask" / f"{i:03}.png")
        K = np.array([[2892.33, 0.0, 823.2], [0, 2883.18, 619.07], [0, 0, 1.0]])
        R = _rot(0.1 * i, 0.4 - 0.2 * i, 0.3)
        c = np.array([150.0 * (i - 1), 80.0, -600.0 - 20 * i])
        W = np.eye(4)
        W[:3, :4] = K @ np.concatenate([R, (-R @ c)[:, None]], axis=1)
        S = np.diag([300.0, 300.0, 300.0, 1.0])
        S[:3, 3] = [10.0, -20.0, 630.0]
        cams[f"world_mat_{i}"], cams[f"scale_mat_{i}"] = W, S
        Ks.append(K); Rs.append(R); cs.append((c - S[:3, 3]) / 300.0)
    (tmp_path / "image" / "._junk").write_bytes(b"x")
    np.savez(tmp_path / "cameras.npz", **cams)
    labels, cam, _ = loaders.dtu(str(tmp_path), size=16)
    assert labels.shape == (n, 16, 16, 3) and len(cam) == n
    cs = np.stack(cs)
    maxd = np.linalg.norm(cs, axis=-1).max()
    for i in range(n):
        assert np.allclose(cam.pose.data[i, :3, :3].numpy(), Rs[i].T, atol=1e-5)
        assert np.allclose(cam.pose.data[i, :3, 3].numpy(), cs[i] / maxd, atol=1e-5)
        Ki = cam.intrinsic.data[i, :3, :3].numpy()
        assert np.allclose(Ki / Ki[0, 0] * Ks[i][0, 0], Ks[i], rtol=1e-4, atol=2e-2)
    labels4, _, _ = loaders.dtu(str(tmp_path), size=16, with_mask=True)
    assert labels4.shape == (n, 16, 16, 4) and set(labels4[..., 3].unique().tolist()) <= {0.0, 1.0}
