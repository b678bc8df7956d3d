"""Multi-GPU front end: frames shard one per GPU, records come back with one all-gather.

The reference is single process ("Parallelism strategies: none"); extraction has
no cross-frame state, so a batch of frames is embarrassingly parallel: frame i goes to rank
i mod R, every rank runs its own xfh_ctx, and one RCCL all-gather over xGMI of the fixed-size
records (n_valid, mono_index, keypoints[nfeatures], desc[nfeatures x 64]; include/xfeat_hip.h)
hands everything to rank 0, which feeds the strictly sequential SLAM state machine in
timestamp order.  There is no other data-path collective.

One process per GPU (`python -m torch.distributed.run --nproc-per-node N ...`), backend
"nccl" (= RCCL on ROCm) for device tensors and "gloo" for the CPU tests.
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


def init_process_group(device_is_gpu: bool):
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    rank, _, world = env_world()
    dist.init_process_group("nccl" if device_is_gpu else "gloo", rank=rank, world_size=world)
    return dist


def all_gather_records(local):
    """local: uint8 tensor [slots * record_bytes] (device tensor -> RCCL, CPU tensor -> gloo).
    Returns a [world, slots * record_bytes] tensor on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    out = torch.empty((world, local.numel()), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(-1), local.contiguous().view(-1))
    return out


class ShardedFrontEnd:
    """extract_fn(frames_u8[B,H,W]) -> uint8 tensor [B * record_bytes] on this rank's device."""

    def __init__(self, extract_fn, record_bytes: int):
        self.extract_fn = extract_fn
        self.record_bytes = record_bytes

    def run(self, frames):
        """frames: the full batch [N,H,W] (every rank sees it, as every rank could read the
        sequence from disk).  Returns N record byte-blobs in frame order (valid on every rank)."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        n = len(frames)
        slots = frames_per_rank(n, world)
        mine = shard_indices(n, rank, world)
        idx = mine + [mine[-1] if mine else 0] * (slots - len(mine))      # pad the last round
        local = self.extract_fn(frames[idx])
        gathered = all_gather_records(local)
        per_rank = [[gathered[r, j * self.record_bytes:(j + 1) * self.record_bytes] for j in range(slots)] for r in range(world)]
        return unshard(per_rank, n, world)
