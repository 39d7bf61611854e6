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


# ---------------------------------------------------------------------------- 4 and 8 ranks, ragged bands, frame shards
def _worker_many(rank, world, port, size, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), NA_DIST_BACKEND="gloo")
    import math
    import nerf_atlas_amd.nerf as nerf
    import nerf_atlas_amd.cameras as cameras
    from nerf_atlas_amd import dist as nd, render
    torch.cuda.set_device(0)
    r, w, _ = nd.init_from_env(backend="gloo")
    ok = []
    with torch.no_grad():
        n, backend = nd.first_collective(torch.device("cuda", 0))
        ok.append(n == world and backend == "gloo")
        torch.manual_seed(3)
        m = nerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").cuda().eval()
        focal = 0.5 * size / math.tan(0.5 * 0.6911)
        c2w = torch.tensor([[[1.0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]])
        cam = cameras.NeRFCamera(cam_to_world=c2w, focal=focal).cuda()
        bands = nd.row_bands(size, w)
        ok.append(sum(nr for _, nr in bands) == size and max(nr for _, nr in bands) - min(nr for _, nr in bands) <= 1)
        def rows(r0, nr):
            out, _ = render.render(m, cam, (r0, 0, nr, size), size, with_noise=False)
            return out.squeeze(0)
        # two frames in a row through the convenience wrapper: the second call must not overwrite the first result
        f1 = nd.render_frame_sharded(rows, size, r, w)
        m.refl.mlp.out.bias.add_(0.25)
        f2 = nd.render_frame_sharded(rows, size, r, w)
        if r == 0:
            ok.append(bool(torch.equal(f2, rows(0, size))))
            m.refl.mlp.out.bias.sub_(0.25)
            ok.append(bool(torch.equal(f1, rows(0, size))) and not torch.equal(f1, f2))
            ok.append(f1.data_ptr() != f2.data_ptr())
        else:
            ok.append(f1 is None and f2 is None)
        # D-NeRF time sweep sharded by FRAME (render.render_over_time, runner.py:998-1017): rank r renders frames r, r + w, ...
        torch.manual_seed(5)
        canon = nerf.PlainNeRF(steps=16, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted")
        dyn = nerf.DynamicNeRF(canonical=canon, spline=4).cuda().eval()
        for p_ in dyn.delta_estim.out.parameters(): p_.normal_(0, 0.05)  # (zero-initialised in the reference: make the scene move)
        cam2 = cameras.NeRFCamera(cam_to_world=c2w, focal=0.5 * 24 / math.tan(0.5 * 0.6911)).cuda()
        times = torch.linspace(0.05, 0.95, 11)
        mine = render.render_over_time(dyn, cam2, 24, 12, times, rank=r, world=w)
        ok.append([i for i, _ in mine] == list(range(r, len(times), w)))
        parts = [None] * w
        torch.distributed.all_gather_object(parts, [(i, f.cpu()) for i, f in mine])
        if r == 0:
            got = dict(kv for part in parts for kv in part)
            full = render.render_over_time(dyn, cam2, 24, 12, times)
            ok.append(sorted(got) == list(range(len(times))) and all(torch.equal(got[i], f.cpu()) for i, f in full))
            ok.append(not torch.equal(full[0][1], full[-1][1]))
    q.put(all(ok) if ok else False)
    torch.distributed.destroy_process_group()
    nd.reset_plans()


@pytest.mark.parametrize("world,size", [(4, 801), (8, 801)])
def test_many_ranks_ragged_bands_and_frame_shards(world, size):
    """4 and 8 ranks on one GPU (gloo): 801 rows do not divide (bands of 201 / 200 and 101 / 100 rows), the frame gathered
    through the padded send buffers is bit-identical to the single-rank frame, consecutive frames do not alias, and a D-NeRF
    time sweep sharded by frame reassembles exactly."""
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_many, args=(r, world, port, size, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs: p.join(timeout=120)
    assert all(res), res
