"""Multi-GPU plumbing: one process per GPU over torch.distributed.

The path shards by independent frame pairs: no data-path
collective.  The only exchange is one broadcast of the shared parameter block
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests) and the
final reduction of (elapsed, pixel*iters) for the report.
"""
import ctypes as C
import os

import numpy as np

from . import capi


def env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_pairs(n_pairs, world, rank):
    """Static block distribution: pair k goes to rank floor(k*world/n_pairs)."""
    return [k for k in range(n_pairs) if (k * world) // n_pairs == rank]


def pack_block(blk, constraints=None):
    """ParamBlock (+ constraints) -> uint8 array."""
    cons = np.asarray(constraints if constraints is not None else [], dtype=np.float32).reshape(-1, 5)
    blk.n_constraints = len(cons)
    return np.concatenate([np.frombuffer(bytes(blk), dtype=np.uint8), cons.view(np.uint8).ravel()]).copy()


def unpack_block(raw):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    blk = capi.ParamBlock()
    n = C.sizeof(blk)
    C.memmove(C.addressof(blk), raw[:n].tobytes(), n)
    cons = raw[n:n + blk.n_constraints * 20].view(np.float32).reshape(-1, 5).copy()
    return blk, cons


MAX_BLOCK_BYTES = 1 << 24


def broadcast_block(raw, device):
    """Rank 0's packed block to every rank: a fixed 8-byte header (the payload size) first,
    then a payload of exactly that size, so every rank posts the same two collectives whatever
    rank 0 holds.  A block that is too large is refused on ALL ranks (the header is
    broadcast before anybody can raise), never truncated.  Returns the uint8 array every rank
    now agrees on."""
    import torch
    import torch.distributed as dist
    n = int(np.asarray(raw).size) if dist.get_rank() == 0 else 0
    head = torch.tensor([n], dtype=torch.int64, device=device)
    dist.broadcast(head, src=0)
    n = int(head.item())
    if n < C.sizeof(capi.ParamBlock) or n > MAX_BLOCK_BYTES:
        raise ValueError("parameter block of %d bytes (expected %d..%d)" % (n, C.sizeof(capi.ParamBlock), MAX_BLOCK_BYTES))
    buf = np.zeros(n, dtype=np.uint8)
    if dist.get_rank() == 0:
        buf[:] = np.ascontiguousarray(raw, dtype=np.uint8).ravel()
    t = torch.from_numpy(buf).to(device)
    dist.broadcast(t, src=0)
    return t.cpu().numpy()


def reduce_report(elapsed, units, device, *more):
    """(max over ranks of elapsed, sum over ranks of units[, sums of `more`]): one all-gather of
    a few scalars per rank."""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([elapsed, units] + [float(x) for x in more], dtype=torch.float64, device=device)
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    vals = torch.stack(every).cpu().numpy()
    out = (float(vals[:, 0].max()), float(vals[:, 1].sum()))
    return out + tuple(float(vals[:, 2 + k].sum()) for k in range(len(more)))
