"""Multi-GPU launch helpers: one process per GPU, independent video streams sharded round-robin
(stream s -> rank s mod world), ONE collective: the broadcast of the packed weight blob from rank 0
(RCCL over xGMI with backend "nccl"; gloo in the CPU tests).  The frame path itself has no exchange
step: frame i of a stream consumes frame i-1's output of the same stream (fast_artistic_video.lua:153-169)."""
from __future__ import annotations

from typing import List, Optional


def streams_for_rank(n_streams: int, rank: int, world: int) -> List[int]:
    return [s for s in range(n_streams) if s % world == rank]


def broadcast_blob(blob: Optional[bytes], device, force: bool = False) -> bytes:
    """rank 0 passes the packed checkpoint (fav_net_pack_host), every rank gets the same bytes back.
    force: run the two broadcasts even in a one-rank group (exercises the RCCL path on a 1-GPU box)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        assert blob is not None
        return blob
    rank = dist.get_rank()
    n = torch.tensor([len(blob) if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, 0)
    if rank == 0:
        buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, 0)
    return buf.cpu().numpy().tobytes()


def max_over_ranks(seconds: float, device) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
