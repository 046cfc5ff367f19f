"""Host-side logic of the data-parallel loop (one process per GPU).

Reference: lamp-data/src/main/scala/lamp/data/distributed/package.scala:171-445 (drive/follow: rank 0 creates the NCCL
unique id and hands it to the followers over the control plane, everybody calls ncclCommInitRank), :690-719
(averageGradients) and lamp-data/.../BatchStream.scala:378-402 (everyNth sharding).  The control plane here is a small TCP
rendezvous (the reference uses cats-effect queues or Akka TCP); the data plane is RCCL through the C ABI.  No torch import.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import socket
import struct
import tempfile
import time
from typing import List, Optional, Sequence, Tuple


def env_rank() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


# ---- control plane -----------------------------------------------------------------------------------------------------------
# The reference hands the NCCL unique id from the root to the followers over its own control plane (cats-effect queues inside one
# JVM, or Akka TCP between hosts: DistributedCommunication.scala:15-62, AkkaDistributedCommunication.scala:59-68).  Here that is a
# few lines of TCP on the loop-back / cluster interface: rank 0 listens, every other rank keeps one connection to it, and each
# collective of the control plane is one message up and one reply down.  No tensor library is involved; the data plane is RCCL.

def _send_msg(sock: socket.socket, obj) -> None:
    raw = json.dumps(obj).encode()
    sock.sendall(struct.pack("<I", len(raw)) + raw)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        buf += chunk
    return bytes(buf)


MAX_MSG_BYTES = 1 << 20          # control-plane messages are ids, digests and a few floats per rank


def _recv_msg(sock: socket.socket):
    (n,) = struct.unpack("<I", _recv_exact(sock, 4))
    if n > MAX_MSG_BYTES:
        raise ConnectionError(f"control plane: a message of {n} bytes is not one of ours (limit {MAX_MSG_BYTES})")
    return json.loads(_recv_exact(sock, n).decode())


def rendezvous_file() -> str:
    """Where rank 0 publishes the port it listens on.  All ranks of one launch share MASTER_ADDR / MASTER_PORT and - being children
    of one launcher (torch.distributed.run's agent or bench.py's own spawner) - the parent pid, which keeps two launches apart even
    when they are given the same MASTER_PORT.  LAMP_RDZV_FILE overrides (multi-node: a path on a shared file system)."""
    if os.environ.get("LAMP_RDZV_FILE"):
        return os.environ["LAMP_RDZV_FILE"]
    key = f"{os.environ.get('MASTER_ADDR', '127.0.0.1')}_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
    return os.path.join(tempfile.gettempdir(), f"lamp_rdzv_{key}.json")


class ControlPlane:
    """rank 0 = server, ranks 1.. = clients; collectives: barrier, broadcast_bytes, all_reduce_max, all_gather."""

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", port: int = 0, rdzv: Optional[str] = None, timeout: float = 600.0):
        self.rank, self.world, self.timeout = rank, world, timeout
        self.peers: List[Optional[socket.socket]] = [None] * world       # rank 0 only
        self.up: Optional[socket.socket] = None                        # ranks > 0: connection to rank 0
        self._rdzv_written = None
        if world == 1:
            return
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", port))
            srv.listen(world)
            srv.settimeout(timeout)
            nonce = os.urandom(8).hex()
            if rdzv:
                # owner-only (mkstemp: 0600, O_EXCL, an unpredictable name - nobody can plant a symlink where this process is about to
                # write): the nonce keeps strangers on a shared host out of the clique, so they must not be able to read it
                fd, tmp = tempfile.mkstemp(prefix=os.path.basename(rdzv) + ".", suffix=".tmp", dir=os.path.dirname(rdzv) or ".")
                with os.fdopen(fd, "w") as f:
                    json.dump({"port": srv.getsockname()[1], "nonce": nonce, "pid": os.getpid()}, f)
                os.replace(tmp, rdzv)                                  # atomic: readers see nothing or the whole record (a stale one is replaced)
                self._rdzv_written = rdzv
            try:
                joined = 0
                while joined < world - 1:
                    c, _ = srv.accept()
                    # the handshake of ONE connection must not take rank 0 down or block the others: a stray or silent client (port
                    # scanner, a rank of another launch reading a stale record) is dropped after a few seconds and the loop goes on
                    try:
                        c.settimeout(5.0)
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        hello = _recv_msg(c)
                        r = int(hello["rank"])
                        if hello.get("world") != world or not (0 < r < world) or self.peers[r] is not None or (rdzv and hello.get("nonce") != nonce):
                            _send_msg(c, {"ok": False, "why": f"rank {r} of world {hello.get('world')} does not belong to this launch (world {world})"})
                            c.close()
                            continue
                        _send_msg(c, {"ok": True})
                        # the client confirms that it is still there: one that gave up waiting (rank 0 was busy with a silent stray)
                        # has closed this connection and is retrying on a new one - counting the dead one would end the accept loop
                        if not _recv_msg(c).get("ack"):
                            raise ConnectionError("no acknowledgement")
                        # ... and is told that it now counts: a client whose acknowledgement came too late for the 5 s above sees this
                        # connection closed instead and joins again on a new one (it used to keep the dead socket)
                        _send_msg(c, {"joined": True})
                        c.settimeout(timeout)
                    except (OSError, ValueError, KeyError, TypeError, AttributeError, ConnectionError, struct.error):
                        try:
                            c.close()
                        except OSError:
                            pass
                        continue
                    self.peers[r] = c
                    joined += 1
            finally:
                srv.close()
        else:
            deadline = time.monotonic() + timeout
            last = None
            while True:
                try:
                    if rdzv:
                        with open(rdzv) as f:
                            rec = json.load(f)
                        cport, nonce = int(rec["port"]), rec["nonce"]
                    else:
                        cport, nonce = port, None
                    s = socket.create_connection((addr, cport), timeout=15.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _send_msg(s, {"rank": rank, "world": world, "nonce": nonce})
                    ans = _recv_msg(s)            # still under the 15 s timeout: a stale record's port answers nothing, try again
                    if not ans.get("ok"):
                        s.close()
                        raise ConnectionError(ans.get("why", "refused"))
                    _send_msg(s, {"ack": True})
                    if not _recv_msg(s).get("joined"):    # rank 0 dropped this connection (acknowledgement too late): closed socket -> retry
                        s.close()
                        raise ConnectionError("rank 0 did not confirm the join")
                    s.settimeout(timeout)
                    self.up = s
                    break
                except (OSError, ValueError, KeyError, ConnectionError) as e:   # not published yet / stale record / not listening yet
                    last = e
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"control plane: rank {rank} could not reach rank 0 within {timeout:.0f} s ({last})")
                    time.sleep(0.05)

    # torch.distributed-flavoured accessors so call sites read the same
    def get_rank(self) -> int: return self.rank
    def get_world_size(self) -> int: return self.world

    def _round(self, mine, combine):
        """one message up from every rank, `combine(list_by_rank)` on rank 0, the result down to every rank"""
        if self.world == 1:
            return combine([mine])
        if self.rank == 0:
            vals = [mine] + [_recv_msg(self.peers[r]) for r in range(1, self.world)]
            out = combine(vals)
            for r in range(1, self.world):
                _send_msg(self.peers[r], out)
            return out
        _send_msg(self.up, mine)
        return _recv_msg(self.up)

    def barrier(self) -> None:
        self._round(None, lambda v: None)

    def broadcast_bytes(self, data: Optional[bytes], root: int = 0) -> bytes:
        return bytes.fromhex(self._round(data.hex() if (self.rank == root and data is not None) else None, lambda v: v[root]))

    def all_reduce_max(self, x: float) -> float:
        return float(self._round(float(x), max))

    def all_reduce_sum(self, x: float) -> float:
        return float(self._round(float(x), lambda v: math.fsum(v)))

    def all_gather(self, obj) -> list:
        return self._round(obj, list)

    def close(self) -> None:
        for s in self.peers + [self.up]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self.peers, self.up = [None] * self.world, None
        if self._rdzv_written:
            try:
                os.unlink(self._rdzv_written)
            except OSError:
                pass
            self._rdzv_written = None

    destroy_process_group = close


def init_control_plane(timeout: float = 600.0) -> ControlPlane:
    """Control plane from the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT - torch.distributed.run and
    bench.py's own spawner both set them).  MASTER_PORT itself belongs to the launcher (torchrun's store listens there), so rank 0
    binds an ephemeral port and publishes it in `rendezvous_file()`; LAMP_CONTROL_PORT pins the port instead."""
    rank, _, world = env_rank()
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("LAMP_CONTROL_PORT"):
        return ControlPlane(rank, world, addr, int(os.environ["LAMP_CONTROL_PORT"]), None, timeout)
    return ControlPlane(rank, world, addr, 0, rendezvous_file(), timeout)


def exchange_unique_id(cp, make_id) -> bytes:
    """rank 0 calls make_id() (128 bytes, lamp_comm_get_unique_id) and broadcasts it - DistributedCommunicationRoot.onUniqueIdReady /
    NonRoot.join in the reference (DistributedCommunication.scala:15-62)."""
    raw = None
    if cp.get_rank() == 0:
        raw = bytes(make_id())
        assert len(raw) == 128
    return cp.broadcast_bytes(raw, 0)


def rccl_communicator(cp):
    """RCCL communicator for this rank (blocking until the clique is complete, like ncclInitComm - STen.scala:629-641)."""
    from ._capi import lib

    def make():
        buf = (C.c_uint8 * 128)()
        lib.lamp_comm_get_unique_id(buf)
        return bytes(buf)
    uid = exchange_unique_id(cp, make)
    h = C.c_void_p()
    lib.lamp_comm_init_rank(C.byref(h), cp.get_world_size(), (C.c_uint8 * 128)(*uid), cp.get_rank())
    return h


def comm_count(comm) -> int:
    """ncclCommCount of a communicator: the number of ranks RCCL itself sees (bench.py refuses to print a line when it differs
    from --gpus)."""
    from ._capi import lib
    n = C.c_int(0)
    lib.lamp_comm_count(comm, C.byref(n))
    return n.value


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
