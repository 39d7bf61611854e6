"""world_size-2 gloo test of the multi-GPU sharding + gather path (nerf_atlas_amd/dist.py) on CPU.
The per-rank renderer is stubbed with a deterministic function of the pixel coordinates: what is under test is
the partition (every row exactly once) and the one collective that reassembles the frame."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _frame_fn(r0, n, size):
    rows = torch.arange(r0, r0 + n, dtype=torch.float32)[:, None, None]
    cols = torch.arange(size, dtype=torch.float32)[None, :, None]
    ch = torch.arange(3, dtype=torch.float32)[None, None, :]
    return rows * 1000 + cols + ch * 0.25


def _worker(rank, world, port, size, q, impl="gather"):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), NA_DIST_GATHER=impl)
    from nerf_atlas_amd import dist as nd
    r, w, _ = nd.init_from_env(backend="gloo")
    n, backend = nd.first_collective("cpu")
    ok = n == world and backend == "gloo"
    frame = nd.render_frame_sharded(lambda r0, n: _frame_fn(r0, n, size), size, r, w)
    # a second frame through the convenience wrapper must not overwrite the first (the plan's receive buffer)
    frame2 = nd.render_frame_sharded(lambda r0, n: _frame_fn(r0, n, size) + 1.0, size, r, w)
    if r == 0:
        ok &= bool(torch.equal(frame, _frame_fn(0, size, size))) and bool(torch.equal(frame2, _frame_fn(0, size, size) + 1.0))
        raw = nd.render_frame_sharded(lambda r0, n: _frame_fn(r0, n, size), size, r, w, copy=False)
        raw2 = nd.render_frame_sharded(lambda r0, n: _frame_fn(r0, n, size) + 2.0, size, r, w, copy=False)
        ok &= raw.data_ptr() == raw2.data_ptr()  # (copy=False: the persistent buffer itself, documented to alias)
    else:
        ok &= frame is None and frame2 is None
        nd.render_frame_sharded(lambda r0, n: _frame_fn(r0, n, size), size, r, w, copy=False)
        nd.render_frame_sharded(lambda r0, n: _frame_fn(r0, n, size) + 2.0, size, r, w, copy=False)
    q.put(bool(ok))
    torch.distributed.destroy_process_group()
    nd.reset_plans()


def _probe_worker(rank, world, port, size, q, mode):
    """mode "nogather": dist.gather raises NotImplementedError on every rank (a backend build without it) -> the group agrees on
    all_gather at construction, prints why, and the frames are right.  mode "broken": the probe meets a communicator-style
    RuntimeError -> it is re-raised, not downgraded to a fallback."""
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("NA_DIST_GATHER", None)
    import torch.distributed as dist
    from nerf_atlas_amd import dist as nd
    r, w, _ = nd.init_from_env(backend="gloo")
    real = dist.gather
    calls = []

    def fake(*a, **k):
        calls.append(1)
        if mode == "nogather":
            raise NotImplementedError("gather is not implemented by this backend build")
        raise RuntimeError("NCCL communicator was aborted on rank %d" % r)
    dist.gather = fake
    ok = True
    try:
        if mode == "nogather":
            plan = nd.BandGather(size, 3, r, w, "cpu")
            ok &= plan.impl == "all_gather" and len(calls) == 1
            for k in range(3):   # every frame goes through all_gather; dist.gather is never tried again
                f = plan(_frame_fn(*nd.row_bands(size, w)[r], size) + k)
                ok &= (f is None) if r else bool(torch.equal(f, _frame_fn(0, size, size) + k))
            ok &= len(calls) == 1
        else:
            try:
                nd.BandGather(size, 3, r, w, "cpu")
                ok = False
            except RuntimeError as e:
                ok &= "aborted" in str(e)
    finally:
        dist.gather = real
    q.put(bool(ok))
    torch.distributed.destroy_process_group()


def test_gather_implementation_is_chosen_once_by_the_group():
    ctx = mp.get_context("spawn")
    for mode in ("nogather", "broken"):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_probe_worker, args=(r, 2, port, 17, q, mode)) for r in range(2)]
        for p in procs: p.start()
        results = [q.get(timeout=120) for _ in procs]
        for p in procs: p.join(timeout=60)
        assert all(results), (mode, results)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_row_bands_partition():
    sys.path.insert(0, REPO)
    from nerf_atlas_amd import dist as nd
    for size in (800, 801, 7, 13):
        for world in (1, 2, 3, 4, 8):
            bands = nd.row_bands(size, world)
            rows = [r for r0, n in bands for r in range(r0, r0 + n)]
            assert rows == list(range(size))
            assert max(n for _, n in bands) - min(n for _, n in bands) <= 1
    tiles = list(range(10))
    assert sorted(sum((nd.shard_tiles(tiles, r, 4) for r in range(4)), [])) == tiles


def test_sharded_frame_gather_gloo_world2():
    ctx = mp.get_context("spawn")
    for size, impl in ((16, "gather"), (17, "gather"), (17, "all_gather")):  # even and ragged split; the fallback collective
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, size, q, impl)) for r in range(2)]
        for p in procs: p.start()
        results = [q.get(timeout=120) for _ in procs]
        for p in procs: p.join(timeout=60)
        assert all(results), results


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from nerf_atlas_amd import dist as nd
    r, w, _ = nd.init_from_env(backend="gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)),
              torch.nn.Parameter(torch.zeros(2, 2), requires_grad=False), torch.nn.Parameter(torch.zeros(4))]
    params[0].grad = torch.full((5, 3), float(r + 1))
    params[1].grad = torch.arange(7, dtype=torch.float32) * (r + 1)
    if r == 0:
        params[3].grad = torch.ones(4)  # rank 1 has no gradient for this one: treated as zeros
    n = nd.allreduce_gradients(params, w)
    ok = n == 15 + 7 + 4
    ok &= bool(torch.allclose(params[0].grad, torch.full((5, 3), 1.5)))
    ok &= bool(torch.allclose(params[1].grad, torch.arange(7, dtype=torch.float32) * 1.5))
    ok &= params[2].grad is None and bool(torch.allclose(params[3].grad, torch.full((4,), 0.5)))
    ok &= nd.shard_batch([4, 9, 2, 7], r, w) == ([4, 2] if r == 0 else [9, 7])
    q.put(ok)
    torch.distributed.destroy_process_group()


def test_world2_gradient_allreduce_and_batch_shards():
    """Data-parallel training replicas: one flat all-reduce averages the gradients; views shard round-robin."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(res), res


def test_a_visibility_subset_shorter_than_the_job_ends_the_rank_before_the_rendezvous(monkeypatch):
    """a driver-set HIP_VISIBLE_DEVICES shorter than N must give the "needs GPU k but only n are visible"
    exit, not a hang at the rendezvous: init_from_env checks before it touches the device or the process group."""
    import pytest
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    from nerf_atlas_amd import dist as nd
    for k, v in dict(RANK="3", WORLD_SIZE="4", LOCAL_RANK="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="1", HIP_VISIBLE_DEVICES="0,1").items():
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    with pytest.raises(SystemExit) as e:
        nd.init_from_env(backend="nccl")
    assert "rank 3 needs GPU 3 but only 2 are visible" in str(e.value) and "HIP_VISIBLE_DEVICES" in str(e.value)
    assert not dist.is_initialized()
