"""Multi-GPU front end: frames shard one per GPU, records come back with one all-gather.

The reference is single process ("Parallelism strategies: none"); extraction has
no cross-frame state, so a batch of frames is embarrassingly parallel: frame i goes to rank
i mod R, every rank runs its own xfh_ctx, and one RCCL all-gather over xGMI of the fixed-size
records (n_valid, mono_index, keypoints[nfeatures], desc[nfeatures x 64]; include/xfeat_hip.h)
hands everything to rank 0, which feeds the strictly sequential SLAM state machine in
timestamp order.  There is no other data-path collective.

One process per GPU (`python -m torch.distributed.run --nproc-per-node N ...` only LAUNCHES the ranks
and sets RANK / WORLD_SIZE / MASTER_*).  On the GPU the collective is the C ABI's (`xfh_comm_create`,
`xfh_allgather_records`, `xfh_gather_records_root`, `xfh_gather_compact_root`: librccl called directly by
libxfeat_hip.so, class `Comm` below); the 128-byte RCCL unique id travels from rank 0 to the others over a
plain TCP socket (`exchange_unique_id`), so no torch is needed in the data path.  `ShardPlan` is the index arithmetic of the
partitioning (frame i -> rank i mod R and back), shared by the RCCL path and the CPU tests (which move the bytes over gloo).
"""
from __future__ import annotations

import os


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1-process default)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_indices(n_frames: int, rank: int, world: int):
    """global frame indices handled by `rank`: frame i -> rank i mod world"""
    return list(range(rank, n_frames, world))


def frames_per_rank(n_frames: int, world: int) -> int:
    """every rank runs the same number of slots (the all-gather is fixed size); missing
    frames of the last round are padded and dropped again in `unshard`"""
    return (n_frames + world - 1) // world


def unshard(gathered, n_frames: int, world: int):
    """gathered[r][j] is rank r's j-th record -> list ordered by global frame index"""
    out = []
    for i in range(n_frames):
        out.append(gathered[i % world][i // world])
    return out


def exchange_unique_id(rank: int, world: int, addr: str, port: int, make_id, nbytes: int = 128, timeout: float = 120.0) -> bytes:
    """rank 0 calls make_id() -> bytes and serves it to the world-1 other ranks on addr:port; the others connect
    (retrying until rank 0 listens) and read it.  Plain TCP: this is the only out-of-band step RCCL needs."""
    import socket
    import time
    if world <= 1:
        return make_id()
    if rank == 0:
        uid = make_id()
        assert len(uid) == nbytes
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(uid)
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as cl:
                buf = b""
                while len(buf) < nbytes:
                    chunk = cl.recv(nbytes - len(buf))
                    if not chunk:
                        raise ConnectionError("short read")
                    buf += chunk
                return buf
        except (ConnectionRefusedError, ConnectionError, OSError):
            if time.time() > deadline:
                raise
            time.sleep(0.05)


class Comm:
    """RCCL communicator of one ctx through the C ABI (include/xfeat_hip.h, multi-GPU exchange)."""

    def __init__(self, ctx, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29533):
        import ctypes as C
        from . import capi
        self.ctx, self.rank, self.world, self.capi, self.C = ctx, rank, world, capi, C
        L = capi.lib()

        def make_id():
            buf = C.create_string_buffer(128)
            capi.check(L.xfh_comm_unique_id(buf), ctx.h)
            return buf.raw
        uid = exchange_unique_id(rank, world, addr, port, make_id)
        capi.check(L.xfh_comm_create(ctx.h, uid, rank, world), ctx.h)

    def allgather_records(self, d_records: int, B: int, d_all: int, gen: int = 0):
        self.capi.check(self.capi.lib().xfh_allgather_records(self.ctx.h, d_records, B, d_all, gen), self.ctx.h)

    def gather_records_root(self, d_records: int, B: int, d_all: int, root: int = 0, gen: int = 0):
        self.capi.check(self.capi.lib().xfh_gather_records_root(self.ctx.h, d_records, B, d_all or None, root, gen), self.ctx.h)

    def gather_compact_root(self, d_records: int, B: int, d_all: int, root: int = 0, gen: int = 0):
        sizes = (self.C.c_size_t * self.world)()
        self.capi.check(self.capi.lib().xfh_gather_compact_root(self.ctx.h, d_records, B, d_all or None, sizes, root, gen), self.ctx.h)
        return list(sizes)

    def allgather_bytes(self, d_send: int, nbytes: int, d_recv: int, gen: int = 0):
        self.capi.check(self.capi.lib().xfh_allgather_bytes(self.ctx.h, d_send, nbytes, d_recv, gen), self.ctx.h)

    def fence(self, gen: int):
        self.capi.check(self.capi.lib().xfh_comm_fence(self.ctx.h, gen), self.ctx.h)

    def wait_ctx(self, other):
        """the next collective also waits for what is queued on `other` (a second ctx of this GPU) so far"""
        self.capi.check(self.capi.lib().xfh_comm_wait_ctx(self.ctx.h, other.h), self.ctx.h)

    def fence_ctx(self, other, gen: int):
        self.capi.check(self.capi.lib().xfh_comm_fence_ctx(self.ctx.h, other.h, gen), self.ctx.h)

    def synchronize(self):
        self.capi.check(self.capi.lib().xfh_comm_synchronize(self.ctx.h), self.ctx.h)

    def barrier_max(self, value: float) -> float:
        """all-gather of one double per rank through RCCL: doubles as the barrier, returns the maximum over ranks"""
        import numpy as np
        send = self.capi.DeviceBuffer(8).upload(np.array([value], np.float64))
        recv = self.capi.DeviceBuffer(8 * self.world)
        self.allgather_bytes(send.ptr, 8, recv.ptr, 0)
        self.synchronize()
        out = recv.download(np.float64, self.world)
        send.free(); recv.free()
        return float(out.max())

    def close(self):
        self.capi.lib().xfh_comm_destroy(self.ctx.h)


class ShardPlan:
    """Who extracts which frame of a global batch, and where its record lands after the gather (BASELINE.json
    configs[3]: "Batch-8 1280x720 frames sharded 1/GPU across 8 MI355X, RCCL all-gather of kpts+desc").

    frame i -> rank i mod world, slot i // world of that rank; every rank runs `slots` = ceil(n / world) slots so that the
    exchange is fixed size, and a rank whose last slot has no frame repeats its last one (dropped again by `global_order`).
    Pure index arithmetic: the same object drives the RCCL path (`Comm`, bench.py, tests/test_gpu_comm_world2.py) and the CPU
    world-2 test over gloo (tests/test_dist_gloo.py)."""

    def __init__(self, n_frames: int, rank: int, world: int):
        assert n_frames >= 1 and 0 <= rank < world
        self.n, self.rank, self.world = n_frames, rank, world
        self.slots = frames_per_rank(n_frames, world)
        mine = shard_indices(n_frames, rank, world)
        # a rank without any frame of its own (n < world) still runs its slot: on frame 0, dropped afterwards
        self.local = mine + [mine[-1] if mine else 0] * (self.slots - len(mine))

    def global_order(self):
        """[(rank, slot)] of frame 0, 1, ... n-1 inside a gathered [world][slots] array of records"""
        return [(i % self.world, i // self.world) for i in range(self.n)]

    def unshard_bytes(self, gathered, record_bytes: int):
        """gathered: u8 array of world * slots * record_bytes (rank-major, the layout of xfh_allgather_records /
        xfh_gather_records_root) -> list of n record views in global frame order"""
        return [gathered[(r * self.slots + j) * record_bytes:(r * self.slots + j + 1) * record_bytes] for r, j in self.global_order()]
