"""Multi-GPU: rays shard embarrassingly.  One process per GPU; a frame is split into contiguous
row bands (or whole tiles), every rank renders its share with replicated weights, and ONE collective gathers
the finished RGB to rank 0 (RCCL over xGMI on the GPU box: backend "nccl"; "gloo" in CPU tests)."""
import os
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; initialises the process group if world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("NA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def row_bands(size: int, world: int) -> List[Tuple[int, int]]:
    """[(row0, nrows)] per rank: contiguous bands, sizes differ by at most one row, every row exactly once."""
    base, extra = divmod(size, world)
    bands, r0 = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        bands.append((r0, n))
        r0 += n
    return bands


def shard_tiles(tiles: list, rank: int, world: int) -> list:
    """Round-robin tile ownership g, g+G, ... (whole tiles keep mip's radii_x row differences intact)."""
    return tiles[rank::world]


def gather_bands(local: torch.Tensor, size: int, rank: int, world: int, dst: int = 0) -> Optional[torch.Tensor]:
    """local: [nrows_rank, size, C] band of this rank.  Returns the [size,size,C] frame on `dst`, None elsewhere.
    Bands may differ by one row, so they are padded to the tallest band for the collective."""
    if world == 1:
        return local
    bands = row_bands(size, world)
    tall = max(n for _, n in bands)
    dev = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:
        local = local.cpu()  # debug mode (several ranks on one GPU): gloo gathers through host memory
    pad = torch.zeros(tall, size, local.shape[-1], device=local.device, dtype=local.dtype)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, outs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:n] for o, (_, n) in zip(outs, bands)], dim=0).to(dev)


def merge_tile_frames(frame: torch.Tensor, rank: int, world: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Tile sharding (mip: radii_x differences rows inside a crop, so whole tiles are the unit): every rank holds a full
    [size,size,C] frame with its own tiles filled and zeros elsewhere; every pixel has exactly one owner, so ONE
    sum-reduce to `dst` reassembles the frame exactly (x + 0 is exact).  Returns the frame on `dst`, None elsewhere."""
    if world == 1:
        return frame
    dev = frame.device
    buf = frame.cpu() if (dist.get_backend() == "gloo" and frame.is_cuda) else frame.clone()
    dist.reduce(buf, dst=dst, op=dist.ReduceOp.SUM)
    return buf.to(dev) if rank == dst else None


def render_frame_sharded(render_rows: Callable[[int, int], torch.Tensor], size: int, rank: int, world: int):
    """render_rows(row0, nrows) -> [nrows,size,3] on this rank's device; returns the frame on rank 0."""
    r0, n = row_bands(size, world)[rank]
    return gather_bands(render_rows(r0, n), size, rank, world)


# ------------------------------------------------------------------------------------------------- training replicas
def shard_batch(idxs: list, rank: int, world: int) -> list:
    """Views of one training batch owned by this rank (idxs[rank::world]).  With batch_size % world == 0 the mean of
    the per-rank mean losses equals the single-process loss over the whole batch."""
    return idxs[rank::world]


def allreduce_gradients(params, world: Optional[int] = None) -> int:
    """Average .grad over the replicas with ONE flat all-reduce (replicas + all-reduce of grads).
    PlainNeRF's parameters are ~11 MB of fp32, 8 MiB of it hash tables: a single bucket, which on the xGMI ring is
    latency- rather than bandwidth-bound, so there is nothing to overlap with backward.  Parameters without a gradient
    on this rank contribute zeros (every rank must flatten the same set).  Returns the number of elements reduced."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    params = [p for p in params if p.requires_grad]
    if world == 1 or not params:
        return 0
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    on_cuda = flat.is_cuda
    if dist.get_backend() == "gloo" and on_cuda:
        flat = flat.cpu()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat = flat / world
    if on_cuda and not flat.is_cuda:
        flat = flat.to(params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return off
