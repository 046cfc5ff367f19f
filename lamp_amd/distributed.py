"""Host-side logic of the data-parallel loop (one process per GPU).

Reference: lamp-data/src/main/scala/lamp/data/distributed/package.scala:171-445 (drive/follow: rank 0 creates the NCCL
unique id and hands it to the followers over the control plane, everybody calls ncclCommInitRank), :690-719
(averageGradients) and lamp-data/.../BatchStream.scala:378-402 (everyNth sharding).  The control plane here is a gloo
process group (the reference uses cats-effect queues or Akka TCP); the data plane is RCCL through the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence, Tuple


def env_rank() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_control_plane():
    """gloo process group from the torchrun environment (MASTER_ADDR/PORT, RANK, WORLD_SIZE)."""
    import torch.distributed as dist
    rank, _, world = env_rank()
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def exchange_unique_id(dist, make_id) -> bytes:
    """rank 0 calls make_id() (128 bytes, lamp_comm_get_unique_id) and broadcasts it - DistributedCommunicationRoot.onUniqueIdReady /
    NonRoot.join in the reference (DistributedCommunication.scala:15-62)."""
    import torch
    t = torch.zeros(128, dtype=torch.uint8)
    if dist.get_rank() == 0:
        raw = bytes(make_id())
        assert len(raw) == 128
        t = torch.tensor(list(raw), dtype=torch.uint8)
    dist.broadcast(t, 0)
    return bytes(t.tolist())


def rccl_communicator(dist):
    """RCCL communicator for this rank (blocking until the clique is complete, like ncclInitComm - STen.scala:629-641)."""
    from ._capi import lib

    def make():
        buf = (C.c_uint8 * 128)()
        lib.lamp_comm_get_unique_id(buf)
        return bytes(buf)
    uid = exchange_unique_id(dist, make)
    h = C.c_void_p()
    lib.lamp_comm_init_rank(C.byref(h), dist.get_world_size(), (C.c_uint8 * 128)(*uid), dist.get_rank())
    return h


def bucket_layout(numels: Sequence[int]) -> Tuple[List[int], int]:
    """offsets of each gradient inside the flat fp32 bucket; the bucket has one extra trailing element for numExamples."""
    offs, o = [], 0
    for n in numels:
        offs.append(o)
        o += int(n)
    return offs, o + 1


def every_nth(num_batches: int, n: int, offset: int) -> List[int]:
    """BatchStream.everyNth(n, offset): rank `offset` of `n` takes minibatches offset, offset+n, ... (BatchStream.scala:378-402).
    All ranks must see the same number of batches or the clique deadlocks (distributed/package.scala:613-616): the tail that does
    not fill a full round is dropped."""
    full = (num_batches // n) * n
    return list(range(offset, full, n))


def row_shard(n: int, world: int, rank: int) -> Tuple[int, int]:
    """rows [lo, hi) of rank `rank` when n rows are split into `world` equal blocks (the last ones padded: every rank owns
    ceil(n / world) rows so that the all-gather has equal contributions)."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def knn_search_sharded(features, k: int, comm, world: int, rank: int):
    """kNN graph of `features` against itself with the QUERY ROWS split over the ranks (independent row blocks, the data
    set replicated in every GPU's HBM, one all-gather of the [n / world, k] index blocks at the end).  Returns the full [n, k] i64
    index tensor on every rank.  lamp-knn itself is single-device (knn/package.scala:60-121)."""
    from . import sten as S
    from . import umap as U
    from ._capi import lib
    n = features.shape[0]
    per = (n + world - 1) // world
    lo, hi = row_shard(n, world, rank)
    # every rank contributes exactly `per` rows: short or empty blocks query the last rows again (dropped after the gather)
    qlo = max(min(lo, n - per), 0) if per <= n else 0
    mine = U.knn_search(features, features.slice(0, qlo, min(qlo + per, n)), k)
    if mine.shape[0] < per:                                            # n < per * 1: tiny inputs
        mine = S.STen.cat([mine] + [mine.slice(0, 0, 1)] * (per - mine.shape[0]), 0)
    if world == 1 or comm is None:
        return mine.slice(0, 0, n)
    out = S.STen.zeros([world * per, k], S.I64, features.device)
    lib.lamp_comm_all_gather(out, mine.contiguous(), comm)
    # block r holds the rows [qlo_r, qlo_r + per); only the last block can be shifted: put the rows back in order
    parts = []
    for r in range(world):
        rlo, rhi = row_shard(n, world, r)
        if rhi <= rlo:
            continue
        rq = max(min(rlo, n - per), 0)
        parts.append(out.slice(0, r * per + (rlo - rq), r * per + (rlo - rq) + (rhi - rlo)))
    return parts[0] if len(parts) == 1 else S.STen.cat(parts, 0)
