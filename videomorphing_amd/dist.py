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


def broadcast_block(raw, device, max_constraints=256):
    """Rank 0's packed block to every rank (fixed-size buffer so that all ranks post
    the same collective).  Returns the uint8 array every rank now agrees on."""
    import torch
    import torch.distributed as dist
    cap = C.sizeof(capi.ParamBlock) + 20 * max_constraints
    buf = np.zeros(cap, dtype=np.uint8)
    if dist.get_rank() == 0:
        assert raw.size <= cap
        buf[:raw.size] = raw
    t = torch.from_numpy(buf).to(device)
    dist.broadcast(t, src=0)
    return t.cpu().numpy()


def reduce_report(elapsed, units, device):
    """(max over ranks of elapsed, sum over ranks of units)."""
    import torch
    import torch.distributed as dist
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    tsum = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    return tmax.item(), tsum.item()
