"""CPU test of the multi-GPU partitioning (xfeatslam_amd/dist.py: ShardPlan, the TCP bootstrap; xfh_unpack_compact) with
world sizes 2 and 3 over gloo: frame i -> rank i mod R, one all-gather of fixed-size records in the rank-major layout of
xfh_allgather_records, global frame order restored.  The extractor is replaced by the CPU oracle (test infrastructure) packed
into the record layout the HIP library writes.  The RCCL calls themselves (csrc/comm.cpp) run with 2 and 3 ranks in
tests/test_gpu_comm_world2.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT
from xfeatslam_amd import dist as xd, synth, weights as WT

NF = 64


def _rec_layout():
    from xfeatslam_amd import capi
    L = capi.lib()
    return int(L.xfh_record_bytes(NF)), int(L.xfh_record_kps_offset()), int(L.xfh_record_desc_offset(NF))


def _pack(kps, desc, nv, mono, rec, koff, doff):
    r = np.zeros(rec, np.uint8)
    r[:16].view(np.int32)[:] = (nv, mono, 0, 0)
    r[koff:koff + 28 * NF] = kps.view(np.uint8)
    r[doff:doff + 256 * NF] = desc.reshape(-1).view(np.uint8)
    return r


def _worker(rank, world, port, nframes, out_dir):
    """one rank: ShardPlan (the product's index arithmetic, xfeatslam_amd/dist.py) decides which frames this rank extracts
    and where each record lands; the bytes move with a gloo all-gather here (RCCL through xfh_comm_* on the GPU,
    tests/test_gpu_comm_world2.py) in the same rank-major [world][slots] layout xfh_allgather_records produces"""
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import oracle as O
    O.set_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rec, koff, doff = _rec_layout()
    orc = O.Oracle(WT.pack_blob(WT.make_synthetic(1234, 3.0)))
    frames = synth.frames(nframes, 64, 96, seed=5)
    plan = xd.ShardPlan(nframes, rank, world)
    local = torch.from_numpy(np.concatenate([_pack(*orc.extract(frames[i], NF, (0, 0)), rec, koff, doff) for i in plan.local]))
    assert local.numel() == plan.slots * rec
    gathered = torch.empty(world * local.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered, local)
    recs = plan.unshard_bytes(gathered.numpy(), rec)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.stack(recs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nframes", [(2, 4), (2, 5), (3, 2)])
def test_shard_and_all_gather_world_n(tmp_path, oracle_mod, world, nframes):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, nframes, str(tmp_path)), nprocs=world, join=True)
    views = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    a = views[0]
    assert a.shape[0] == nframes and all(np.array_equal(a, v) for v in views[1:])     # every rank holds all records, frame order
    rec, koff, doff = _rec_layout()
    orc = oracle_mod.Oracle(WT.pack_blob(WT.make_synthetic(1234, 3.0)))
    frames = synth.frames(nframes, 64, 96, seed=5)
    for i in range(nframes):
        assert np.array_equal(a[i], _pack(*orc.extract(frames[i], NF, (0, 0)), rec, koff, doff)), i


def test_shard_arithmetic():
    assert xd.shard_indices(8, 3, 8) == [3] and xd.shard_indices(10, 1, 4) == [1, 5, 9]
    assert xd.frames_per_rank(10, 4) == 3 and xd.frames_per_rank(8, 8) == 1
    g = [[f"r{r}j{j}" for j in range(3)] for r in range(4)]
    assert xd.unshard(g, 10, 4) == [f"r{i % 4}j{i // 4}" for i in range(10)]
    p = xd.ShardPlan(10, 1, 4)
    assert p.slots == 3 and p.local == [1, 5, 9] and xd.ShardPlan(10, 3, 4).local == [3, 7, 7]           # the last round is padded
    assert xd.ShardPlan(2, 2, 3).local == [0] and xd.ShardPlan(2, 2, 3).slots == 1                       # a rank without a frame of its own
    assert p.global_order() == [(i % 4, i // 4) for i in range(10)]
    flat = np.arange(4 * 3 * 2, dtype=np.uint8)                                                          # [world][slots] records of 2 bytes
    assert [v.tolist() for v in p.unshard_bytes(flat, 2)] == [[(i % 4 * 3 + i // 4) * 2, (i % 4 * 3 + i // 4) * 2 + 1] for i in range(10)]


def _uid_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    uid = xd.exchange_unique_id(rank, world, "127.0.0.1", port, lambda: bytes(range(128)))
    open(os.path.join(out_dir, f"uid{rank}.bin"), "wb").write(uid)


def test_unique_id_bootstrap_world3(tmp_path):
    """the out-of-band step of xfh_comm_create: rank 0 serves the 128-byte id over TCP, late and early joiners both get it"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_uid_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        assert open(tmp_path / f"uid{r}.bin", "rb").read() == bytes(range(128))


def test_unpack_compact_host_side():
    """xfh_unpack_compact (host code of the compact gather): a hand-packed shard of two frames restores the padded records"""
    import ctypes as C
    from xfeatslam_amd import capi
    L = capi.lib()
    nf, B = 8, 2
    nv, mono = [5, 0], [3, 0]                      # frame 0: 3 front + 2 back rows, frame 1: empty
    total = sum(nv)
    hdr_b = (B * 16 + 255) & ~255; kps_b = (total * 28 + 255) & ~255
    shard = np.zeros(256 + hdr_b + kps_b + total * 256, np.uint8)
    shard[:12].view(np.int32)[:] = (B, nf, total)
    for b in range(B):
        shard[256 + 16 * b:256 + 16 * b + 16].view(np.int32)[:] = (nv[b], mono[b], 77, 0)
    kps = np.zeros(total, capi.KP_DTYPE); kps["x"] = np.arange(total) + 1; kps["size"] = 1; kps["angle"] = -1; kps["class_id"] = -1
    desc = np.arange(total * 64, dtype=np.float32).reshape(total, 64)
    shard[256 + hdr_b:256 + hdr_b + total * 28] = kps.view(np.uint8)
    shard[256 + hdr_b + kps_b:] = desc.reshape(-1).view(np.uint8)
    for b in range(B):
        ok = np.zeros(nf, capi.KP_DTYPE); od = np.full((nf, 64), 9, np.float32); a, m = C.c_int(), C.c_int()
        assert L.xfh_unpack_compact(shard.ctypes.data, shard.nbytes, b, nf, ok.ctypes.data, od.ctypes.data, C.byref(a), C.byref(m)) == 0
        assert (a.value, m.value) == (nv[b], mono[b])
        if b == 0:
            assert ok["x"].tolist() == [1, 2, 3, 0, 0, 0, 4, 5] and ok["class_id"].tolist() == [-1] * 8 and ok["angle"][3] == -1 and ok["size"][3] == 0
            assert np.array_equal(od[:3], desc[:3]) and np.array_equal(od[6:], desc[3:5]) and not od[3:6].any()
        else:
            assert not ok["x"].any() and not od.any() and (ok["angle"] == -1).all()
    assert L.xfh_unpack_compact(shard.ctypes.data, 300, 0, nf, ok.ctypes.data, od.ctypes.data, None, None) == 1      # truncated shard
