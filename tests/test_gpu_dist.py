"""Two ranks on ONE GPU (gloo rendezvous, NA_DIST_BACKEND=gloo debug mode of nerf_atlas_amd/dist.py) through the REAL
fused HIP renderer: the frame gathered from the row bands of the two ranks is bit-identical to the single-rank frame,
and so is the tile-sharded frame of the mip model.  On the 8-GPU node the only
difference is the backend of the one collective ("nccl" = RCCL)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE, T, CROP = 96, 48, 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(mip):
    import math
    import nerf_atlas_amd.nerf as nerf
    import nerf_atlas_amd.cameras as cameras
    from nerf_atlas_amd.utils import CylinderGaussian
    torch.manual_seed(3)
    m = nerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted",
                       mip=CylinderGaussian() if mip else None).cuda().eval()
    focal = 0.5 * SIZE / math.tan(0.5 * 0.6911)
    cam = cameras.NeRFCamera(cam_to_world=torch.tensor([[[1.0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]]), focal=focal).cuda()
    return m, cam


def _worker(rank, world, port, engine, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), NA_DIST_BACKEND="gloo")
    from nerf_atlas_amd import config, dist as nd, render
    torch.cuda.set_device(0)
    config.set_engine(engine)
    r, w, _ = nd.init_from_env(backend="gloo")
    ok = True
    with torch.no_grad():
        # ---- row bands through the fused renderer + one gather
        m, cam = _build(False)
        def rows(r0, n):
            out, _ = render.render(m, cam, (r0, 0, n, SIZE), SIZE, with_noise=False)
            return out.squeeze(0)
        frame = nd.render_frame_sharded(rows, SIZE, r, w)
        if r == 0:
            ok &= bool(torch.equal(frame, rows(0, SIZE)))
        else:
            ok &= frame is None
        # ---- whole tiles (mip), round-robin ownership + one sum-reduce
        mm, cam = _build(True)
        tiles = render.tile_list(SIZE, CROP)
        mine = render.render_frame(mm, cam, SIZE, CROP, tiles=nd.shard_tiles(tiles, r, w))
        merged = nd.merge_tile_frames(mine, r, w)
        if r == 0:
            ok &= bool(torch.equal(merged, render.render_frame(mm, cam, SIZE, CROP)))
    q.put(ok)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("engine", ["ls", "reg"])
def test_two_ranks_one_gpu_bit_identical_frame(engine):
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, engine, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs: p.join(timeout=120)
    assert all(res), res
