#!/usr/bin/env python3
"""Write a small ANALYTIC scene in the on-disk formats the reference's loaders read (src/loaders.py:74-150):

    <dir>/transforms_train.json, transforms_test.json   (Blender: camera_angle_x + frames[file_path, transform_matrix])
    <dir>/train/r_000.png ...                            (RGBA, 8 bit)
    with --dynamic: frames also carry "time" and the small sphere moves (D-NeRF format)

The scene is ray-cast in closed form (two shaded spheres in front of a transparent background) with integer-only
randomness (none), so the files are bit-identical wherever this script runs: the training-parity fixture
(tools/ref_train_fixture.py, reference on CPU in the build container) and the GPU test (tests/test_gpu_train.py) see
the same dataset without shipping it.  No datasets exist offline; this is the stand-in.
"""
import argparse
import json
import math
import os

import numpy as np
from PIL import Image

FOV_X = 0.6911112070083618  # the Blender lego value


def look_at_origin(theta, phi, radius=4.0):
    """Blender-convention camera-to-world [4,4]: camera looks down its -z, +y up, placed on a sphere."""
    pos = radius * np.array([math.cos(phi) * math.cos(theta), math.cos(phi) * math.sin(theta), math.sin(phi)])
    back = pos / np.linalg.norm(pos)
    right = np.cross(np.array([0.0, 0.0, 1.0]), back)
    right /= np.linalg.norm(right)
    up = np.cross(back, right)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, back, pos
    return m


def ray_cast(c2w, size, t=0.0):
    """RGBA float image [size,size,4] of the analytic scene seen through the reference's pinhole model
    (src/cameras.py:45-66: d = ((u-S/2)/f, -(v-S/2)/f, -1), u = column, v = row)."""
    f = 0.5 * size / math.tan(0.5 * FOV_X)
    v, u = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64), indexing="ij")
    d = np.stack([(u - size * 0.5) / f, -(v - size * 0.5) / f, -np.ones_like(u)], -1)
    rd = d @ c2w[:3, :3].T
    rd /= np.linalg.norm(rd, axis=-1, keepdims=True)
    ro = c2w[:3, 3]
    spheres = [(np.array([0.0, 0.0, 0.0]), 0.9, np.array([0.9, 0.35, 0.2])),
               (np.array([1.1 * math.cos(2.5 * t), 1.1 * math.sin(2.5 * t), 0.45]), 0.4, np.array([0.2, 0.5, 0.95]))]
    best = np.full(u.shape, np.inf)
    rgb = np.zeros(u.shape + (3,))
    light = np.array([0.4, -0.5, 0.77])
    light = light / np.linalg.norm(light)
    for c, r, col in spheres:
        oc = ro - c
        b = (rd * oc).sum(-1)
        disc = b * b - (oc @ oc - r * r)
        hit = disc > 0
        tt = -b - np.sqrt(np.where(hit, disc, 0.0))
        hit &= (tt > 0) & (tt < best)
        p = ro + tt[..., None] * rd
        n = (p - c) / r
        stripes = 0.75 + 0.25 * np.sign(np.sin(9.0 * n[..., 2]) * np.sin(9.0 * np.arctan2(n[..., 1], n[..., 0])))
        shade = 0.25 + 0.75 * np.clip((n * light).sum(-1), 0, 1)
        colr = col[None, None, :] * (stripes * shade)[..., None]
        rgb = np.where(hit[..., None], colr, rgb)
        best = np.where(hit, tt, best)
    alpha = np.isfinite(best).astype(np.float64)
    return np.concatenate([rgb * alpha[..., None], alpha[..., None]], -1)


def write_split(out, name, poses, size, times=None):
    os.makedirs(os.path.join(out, name), exist_ok=True)
    frames = []
    for i, m in enumerate(poses):
        t = 0.0 if times is None else times[i]
        img = np.round(np.clip(ray_cast(m, size, t), 0, 1) * 255).astype(np.uint8)
        Image.fromarray(img, "RGBA").save(os.path.join(out, name, f"r_{i:03}.png"))
        fr = {"file_path": f"./{name}/r_{i:03}", "transform_matrix": [[float(x) for x in row] for row in m]}
        if times is not None:
            fr["time"] = float(t)
        frames.append(fr)
    with open(os.path.join(out, f"transforms_{name}.json"), "w") as f:
        json.dump({"camera_angle_x": FOV_X, "frames": frames}, f, indent=1)


def make_scene(out, size=48, n_train=12, n_test=3, dynamic=False):
    golden = math.pi * (3 - math.sqrt(5))
    train = [look_at_origin(i * golden, 0.25 + 0.5 * ((i * 7) % n_train) / n_train) for i in range(n_train)]
    test = [look_at_origin(0.4 + i * 2.1, 0.45) for i in range(n_test)]
    tt = te = None
    if dynamic:
        tt = [i / max(n_train - 1, 1) for i in range(n_train)]
        te = [(i + 0.5) / n_test for i in range(n_test)]
    write_split(out, "train", train, size, tt)
    write_split(out, "test", test, size, te)
    return out


if __name__ == "__main__":
    a = argparse.ArgumentParser()
    a.add_argument("out")
    a.add_argument("--size", type=int, default=48)
    a.add_argument("--n-train", type=int, default=12)
    a.add_argument("--n-test", type=int, default=3)
    a.add_argument("--dynamic", action="store_true")
    args = a.parse_args()
    print(make_scene(args.out, args.size, args.n_train, args.n_test, args.dynamic))
