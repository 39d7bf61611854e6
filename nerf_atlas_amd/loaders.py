"""Dataset loaders for the on-disk formats either side of the hot path (src/loaders.py:19-195).

Loader(...) -> (labels | (labels, times), Camera, None), with the reference's function names, keyword names and
return protocol, so `loaders.load(args, training)` can replace the reference's call (runner.py:1226, 1323).  Host
I/O only (json + PIL + numpy); the tensors returned live on the CPU like the reference's and the caller moves the
camera to the device.

  original  Blender `transforms_{train,test}.json` + RGBA PNGs        (src/loaders.py:74-101)
  dnerf     the same with a per-frame `time` (sorted, normalised)     (src/loaders.py:103-150)
  dtu       `image/*.png`, optional `mask/`, `cameras.npz`            (src/loaders.py:152-195)

The reference's DTU loader calls cv2.decomposeProjectionMatrix; cv2 is absent offline, so `decompose_projection`
restates the published algorithm (RQ decomposition of P[:, :3] with a positive diagonal, camera centre = null vector
of P).  Parity for that function is UNPINNED (no cv2 to run against); it is checked by recomposition instead.
"""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import cameras

# same keys as src/loaders.py:19-26; the kinds outside the hot path raise
kinds = {"original", "nerv_point", "dtu", "dnerf", "single-video", "pixel-single"}

_logged_size = False


def load_image(src, resize=None):
    """src/utils.py:209-216: PIL open, PIL default-filter resize, /255 -> float32 [H,W,C]."""
    from PIL import Image
    global _logged_size
    img = Image.open(src)
    if not _logged_size:
        print(f"[info]: Original image size is {img.width}w x {img.height}h, resized to {resize}")
        _logged_size = True
    if resize is not None:
        img = img.resize(resize)
    return torch.from_numpy(np.array(img, dtype=float) / 255).float()


def load(args, training=True):
    """src/loaders.py:29-72."""
    assert args.data is not None
    kind = args.data_kind
    if getattr(args, "derive_kind", False):
        if args.data.endswith(".mp4"): kind = "single-video"
        elif args.data.endswith(".jpg"): kind = "pixel-single"
    with_mask = (args.model == "sdf" or getattr(args, "volsdf_alternate", False)) and training
    size = args.size
    if kind == "original":
        return original(args.data, training=training, normalize=False, size=size, white_bg=args.bg == "white",
                        with_mask=with_mask)
    if kind == "dtu":
        return dtu(args.data, training=training, size=size, with_mask=with_mask)
    if kind == "dnerf":
        return dnerf(args.data, training=training, size=size, time_gamma=getattr(args, "time_gamma", False),
                     white_bg=args.bg == "white")
    raise NotImplementedError(f"load data: {kind}" + (" (outside the hot path, SURVEY 2 row 15)" if kind in kinds else ""))


def _focal(size, angle_x):
    return 0.5 * size / np.tan(0.5 * float(angle_x))


def original(dir=".", normalize=True, training=True, size=256, white_bg=False, with_mask=False):
    kind = "train" if training else "test"
    with open(dir + f"transforms_{kind}.json") as f:
        tfs = json.load(f)
    channels = 3 + with_mask
    focal = _focal(size, tfs["camera_angle_x"])
    imgs, c2ws = [], []
    for i, frame in enumerate(tfs["frames"]):
        fp = frame["file_path"] or f"test_{i:03}/nn"  # nerfactor leaves blanks (src/loaders.py:86-88)
        img = load_image(os.path.join(dir, fp + ".png"), resize=(size, size))
        if white_bg:
            img = img[..., :3] * img[..., -1:] + (1 - img[..., -1:])
        imgs.append(img[..., :channels])
        m = torch.tensor(frame["transform_matrix"], dtype=torch.float)[:3, :4]
        if normalize:
            m[:3, 3] = F.normalize(m[:3, 3], dim=-1)
        c2ws.append(m)
    imgs = torch.stack(imgs, dim=0)
    if with_mask:
        imgs[..., -1] = (imgs[..., -1] - 1e-5).ceil()
    return imgs, cameras.NeRFCamera(torch.stack(c2ws, dim=0), focal), None


def dnerf(dir=".", normalize=False, training=True, size=256, time_gamma=True, white_bg=False):
    kind = "train" if training else "test"
    with open(dir + f"transforms_{kind}.json") as f:
        tfs = json.load(f)
    is_gibson = "gibson" in dir
    angle = float(tfs["camera_angle_x"])
    if is_gibson:
        angle *= np.pi / 180
    focal = _focal(size, angle)
    rows = []
    for frame in tfs["frames"]:
        img = load_image(os.path.join(dir, frame["file_path"].rstrip(".png") + ".png"), resize=(size, size))
        if white_bg:
            img = img[..., :3] * img[..., -1:] + (1 - img[..., -1:])
        m = torch.tensor(frame["transform_matrix"], dtype=torch.float)
        if is_gibson:
            m = m.inverse()
        time = frame.get("time", frame.get("timestep"))
        assert time is not None, f"Missing time in frame {frame}"
        rows.append((time, m[:3, :4], img[..., :3]))
    times = [r[0] for r in rows]
    if sorted(times) != times:
        rows = sorted(rows, key=lambda r: r[0])  # stable, keyed on time only (src/loaders.py:128-132)
    times = torch.tensor([r[0] for r in rows])
    c2ws = torch.stack([r[1] for r in rows], dim=0)
    imgs = torch.stack([r[2] for r in rows], dim=0)
    lo, hi = times.min(), times.max()
    if lo < 0 or hi > 1:
        times = ((times - lo) / (hi - lo)).clamp_(min=0, max=1)
    if time_gamma:  # DNeRFAE experiment knob (src/loaders.py:147-148)
        imgs = imgs.pow((2 * times[:, None, None, None] - 1).exp())
    return (imgs, times), cameras.NeRFCamera(c2ws, focal), None


def decompose_projection(P):
    """K [3,3] (K[2,2] = 1 after the caller's normalisation), R [3,3] world->camera, c [4] homogeneous camera
    centre, for P = K [R | -R c]: the published decomposition behind cv2.decomposeProjectionMatrix (RQ of the left
    3x3 block with positive diagonal; centre = right null vector of P).  float64 like OpenCV."""
    P = np.asarray(P, dtype=np.float64)
    M = P[:, :3]
    # RQ via QR of the row-reversed transpose: M = K R, K upper triangular, R orthonormal
    rev = np.flipud(np.eye(3))
    q, r = np.linalg.qr((rev @ M).T)
    K = rev @ r.T @ rev
    R = rev @ q.T
    s = np.sign(np.diag(K))
    s[s == 0] = 1
    K = K * s[None, :]
    R = s[:, None] * R
    if np.linalg.det(R) < 0:  # P is defined up to sign; keep R a rotation
        R = -R
        K = -K
    _, _, vt = np.linalg.svd(P)
    c = vt[-1]
    return K, R, c


def dtu(path=".", training=True, size=256, with_mask=False):
    image_dir = os.path.join(path, "image")
    names = [f for f in sorted(os.listdir(image_dir)) if not f.startswith("._")]
    imgs = torch.stack([load_image(os.path.join(image_dir, f), resize=(size, size)) for f in names], dim=0)
    if with_mask:
        mask_dir = os.path.join(path, "mask")
        masks = [load_image(os.path.join(mask_dir, f), resize=(size, size)).max(dim=-1)[0].ceil()
                 for f in sorted(os.listdir(mask_dir)) if not f.startswith("._")]
        # the reference concatenates [N,H,W] onto [N,H,W,3] here and raises (src/loaders.py:171); intended: a 4th channel
        imgs = torch.cat([imgs, torch.stack(masks, dim=0)[..., None]], dim=-1)
    tfs = np.load(os.path.join(path, "cameras.npz"))
    intrinsics, poses = [], []
    for i in range(len(names)):
        P = (tfs[f"world_mat_{i}"] @ tfs[f"scale_mat_{i}"])[:3, :4]
        K, R, c = decompose_projection(P)
        K = K / K[2, 2]
        intr = np.eye(4)
        intr[:3, :3] = K
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = R.transpose()
        pose[:3, 3] = c[:3] / c[3]
        intrinsics.append(torch.from_numpy(intr).float())
        poses.append(torch.from_numpy(pose).float())
    poses = torch.stack(poses, dim=0)
    poses[:, :3, 3] /= torch.linalg.norm(poses[:, :3, 3], dim=-1).max()  # distances normalised to <= 1
    return imgs, cameras.DTUCamera(pose=poses, intrinsic=torch.stack(intrinsics, dim=0)), None
