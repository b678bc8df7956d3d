"""CPU test of the multi-GPU path (xfeatslam_amd/dist.py) with world_size 2 over gloo: frames
shard i -> rank i mod R, one all-gather of fixed-size records, rank order restored.  The
extractor is replaced by the CPU oracle (test infrastructure) packed into the same record
layout the HIP library writes, so the gather/unshard logic is exercised with real records."""
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
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import oracle as O
    O.set_threads(1)
    d = xd.init_process_group(device_is_gpu=False)
    rec, koff, doff = _rec_layout()
    orc = O.Oracle(WT.pack_blob(WT.make_synthetic(1234, 3.0)))

    def extract_fn(fr):
        return torch.from_numpy(np.concatenate([_pack(*orc.extract(f, NF, (0, 0)), rec, koff, doff) for f in fr]))
    frames = synth.frames(nframes, 64, 96, seed=5)
    recs = xd.ShardedFrontEnd(extract_fn, rec).run(frames)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), torch.stack(recs).numpy())
    d.barrier()
    d.destroy_process_group()


@pytest.mark.parametrize("nframes", [4, 5])
def test_shard_and_all_gather_world2(tmp_path, oracle_mod, nframes):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, nframes, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "r0.npy"); b = np.load(tmp_path / "r1.npy")
    assert a.shape[0] == nframes and np.array_equal(a, b)               # every rank holds all records, frame order
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
