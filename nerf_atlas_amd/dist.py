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
            # one rank per GPU: a visibility subset shorter than the job (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES set by a driver)
            # ends THIS rank here, before the rendezvous, with the reason -- the launcher (torchrun, bench.py's own spawner) then
            # takes the other ranks down instead of leaving them in a rendezvous that can never complete
            ndev = torch.cuda.device_count()
            if local >= ndev:
                vis = {k: os.environ[k] for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if k in os.environ}
                raise SystemExit(f"[nerf_atlas_amd.dist] rank {rank} needs GPU {local} but only {ndev} are visible ({vis or 'no visibility variables set'}): "
                                 f"one rank per GPU with backend nccl (= RCCL); NA_DIST_BACKEND=gloo shares GPUs between ranks for flow tests")
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


class BandGather:
    """The frame-reassembly collective with its buffers allocated ONCE: a persistent `[world, tall, size, C]` receive
    tensor on `dst` (the frame itself when all bands have the same height: no concatenation) and, only when bands differ
    by a row, a padded send buffer.  One `dist.gather` per frame (RCCL over xGMI with backend "nccl"; with "gloo" and CUDA
    tensors -- several ranks on one GPU, a debug mode -- the band goes through a persistent host buffer).  Whether the frame
    collective is `gather` or `all_gather` is agreed by the whole group at construction (`impl`); constructing a BandGather is
    therefore itself a collective call: every rank of the group must do it."""

    def __init__(self, size: int, channels: int, rank: int, world: int, device, dtype=torch.float32, dst: int = 0):
        self.size, self.rank, self.world, self.dst = size, rank, world, dst
        self._channels = channels
        self.bands = row_bands(size, world)
        self.tall = max(n for _, n in self.bands)
        self.even = all(n == self.tall for _, n in self.bands)
        self.device = torch.device(device)
        self.via_host = world > 1 and dist.get_backend() == "gloo" and self.device.type == "cuda"
        cdev = torch.device("cpu") if self.via_host else self.device
        self.nrows = self.bands[rank][1]
        self.send = None if (self.even and not self.via_host) else torch.zeros(self.tall, size, channels, device=cdev, dtype=dtype)
        self.recv = torch.empty(world, self.tall, size, channels, device=cdev, dtype=dtype) if rank == dst else None
        self.recv_list = list(self.recv.unbind(0)) if rank == dst else None
        self.frame = torch.empty(size, size, channels, device=self.device, dtype=dtype) if (rank == dst and (self.via_host or not self.even)) else None
        # dist.gather or all_gather: decided ONCE, by the whole group, here (round 5): NA_DIST_GATHER=all_gather skips the
        # probe; otherwise every rank tries a one-element dist.gather on the real backend and the group agrees on the outcome with
        # an all_reduce(MIN) -- a backend build without gather raises on every rank alike, and if it ever raised on one rank only
        # the others would still learn it here, in a tiny collective at construction, instead of diverging into mismatched
        # collectives in the middle of a frame.  __call__ never falls back: an error there is an error.
        self.impl = "none" if world == 1 else self._choose_impl(cdev, dtype)

    def _choose_impl(self, cdev, dtype) -> str:
        want = os.environ.get("NA_DIST_GATHER", "gather")
        if want not in ("gather", "all_gather"):
            raise ValueError(f"NA_DIST_GATHER={want}: 'gather' or 'all_gather'")
        # EVERY rank reaches the all_reduce below whatever its probe did: 1 = gather works, 0 = this backend has no
        # gather, -1 = a hard error (a communicator error, a timeout, a shape problem), re-raised AFTER the group has agreed -- the
        # peers then raise too instead of waiting for a collective that never comes.  (What this cannot cover: a probe that blocks
        # inside dist.gather because a peer died -- that is the process group's timeout, as for any collective.)
        ok, why, hard = 1, "", None
        if want == "gather":
            try:
                probe = torch.zeros(1, device=cdev, dtype=dtype)
                dist.gather(probe, [torch.zeros_like(probe) for _ in range(self.world)] if self.rank == self.dst else None, dst=self.dst)
            except Exception as e:  # noqa: BLE001
                msg = str(e).lower()
                why = f"{type(e).__name__}: {e}"
                if isinstance(e, NotImplementedError) or (isinstance(e, RuntimeError) and any(
                        k in msg for k in ("not supported", "not implemented", "does not support", "unsupported"))):
                    ok = 0
                else:
                    ok, hard = -1, e
        else:
            ok = 0
        flag = torch.tensor([ok], device=cdev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        agreed = int(flag.item())
        if hard is not None:
            raise hard
        if agreed < 0:
            raise RuntimeError(f"[nerf_atlas_amd.dist] rank {self.rank}: another rank's dist.gather probe failed with a hard error "
                               f"(backend {dist.get_backend()}); see that rank's traceback")
        if agreed == 1:
            return "gather"
        if want == "gather":
            import sys
            print(f"[nerf_atlas_amd.dist] rank {self.rank}: dist.gather unavailable on backend {dist.get_backend()} "
                  f"({why or 'another rank reported it'}); the group uses all_gather for the frame", file=sys.stderr, flush=True)
        if self.recv is None:  # all_gather delivers every band to every rank
            self.recv = torch.empty(self.world, self.tall, self.size, self.send.shape[-1] if self.send is not None else self._channels,
                                    device=cdev, dtype=dtype)
            self.recv_list = list(self.recv.unbind(0))
        return "all_gather"

    def __call__(self, local: torch.Tensor) -> Optional[torch.Tensor]:
        if self.world == 1:
            return local
        assert local.shape[0] == self.nrows, (local.shape, self.nrows)
        if self.send is None:
            send = local.contiguous()
        else:
            self.send[: self.nrows].copy_(local)
            send = self.send
        if self.impl == "gather":
            dist.gather(send, self.recv_list, dst=self.dst)   # (no fallback here: the group chose at construction)
        else:
            dist.all_gather(self.recv_list, send)
        if self.rank != self.dst:
            return None
        if self.even:
            full = self.recv.view(self.world * self.tall, self.size, -1)
            if self.frame is None:
                return full
            self.frame.copy_(full)
            return self.frame
        r0 = 0
        for o, (_, n) in zip(self.recv_list, self.bands):
            self.frame[r0:r0 + n].copy_(o[:n])
            r0 += n
        return self.frame


_band_plans = {}


def reset_plans():
    """Drop the cached gather plans (their buffers belong to a process group: call after destroy_process_group)."""
    _band_plans.clear()


def gather_bands(local: torch.Tensor, size: int, rank: int, world: int, dst: int = 0, copy: bool = True) -> Optional[torch.Tensor]:
    """local: [nrows_rank, size, C] band of this rank.  Returns the [size,size,C] frame on `dst`, None elsewhere.
    Bands may differ by one row; the buffers of the collective are allocated on the first call and reused (`BandGather`,
    one plan per process group and shape).  The frame is a COPY by default: the plan's receive buffer is overwritten by the
    next call, so frames collected in a list would alias each other; `copy=False` (or a `BandGather` of your own, as
    bench.py keeps) returns the persistent buffer itself."""
    if world == 1:
        return local
    key = (size, local.shape[-1], rank, world, str(local.device), local.dtype, dst, dist.get_backend(), id(dist.group.WORLD))
    plan = _band_plans.get(key)
    if plan is None:
        if len(_band_plans) >= 16:  # (plans of destroyed groups and of shapes no longer rendered)
            _band_plans.clear()
        plan = _band_plans[key] = BandGather(size, local.shape[-1], rank, world, local.device, local.dtype, dst)
    frame = plan(local)
    return frame.clone() if (copy and frame is not None) else frame


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


def render_frame_sharded(render_rows: Callable[[int, int], torch.Tensor], size: int, rank: int, world: int, copy: bool = True):
    """render_rows(row0, nrows) -> [nrows,size,3] on this rank's device; returns the frame on rank 0 (a copy: see
    `gather_bands`; `copy=False` hands out the gather's persistent receive buffer, valid until the next call)."""
    r0, n = row_bands(size, world)[rank]
    return gather_bands(render_rows(r0, n), size, rank, world, copy=copy)


def first_collective(device) -> Tuple[int, str]:
    """The process group's first collective, made diagnosable: an all-reduce of ones on `device` (RCCL over xGMI with backend
    "nccl": communicator creation, IPC handle exchange and the first kernel all happen here).  Returns (ranks counted, backend).
    A failure is re-raised after this rank has printed who and where it is -- on an 8-GPU node eight interleaved tracebacks
    without that line say nothing about WHICH rank could not see its GPU or peer."""
    if not dist.is_initialized():
        return 1, "none"
    backend = dist.get_backend()
    try:
        t = torch.ones(1, device=device if backend == "nccl" else "cpu", dtype=torch.float32)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(round(float(t))), backend
    except Exception as e:  # noqa: BLE001 -- annotate and re-raise
        import sys
        vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", "<unset>"))
        print(f"[nerf_atlas_amd.dist] first collective FAILED on rank {dist.get_rank()}/{dist.get_world_size()} "
              f"(local {os.environ.get('LOCAL_RANK')}), device {device}, backend {backend}, visible devices {vis}, "
              f"cuda devices {torch.cuda.device_count()}, MASTER {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}, "
              f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}: {type(e).__name__}: {e}",
              file=sys.stderr, flush=True)
        raise


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
