"""GPU tests of the multi-GPU exchange behind the C ABI (xfh_comm_*, xfh_allgather_records, xfh_gather_records_root,
xfh_gather_compact_root) with a world of ONE rank: librccl is really called (communicator, all-gather, send/recv group,
stream ordering against the extraction), only the peers are missing.  World sizes > 1 run in the driver's scaling bench;
the shard / unshard logic and the TCP bootstrap are covered on CPU (tests/test_dist_gloo.py)."""
import ctypes as C
import socket

import numpy as np
import pytest

from xfeatslam_amd import capi, dist as xd, synth, weights as WT

pytestmark = pytest.mark.gpu


def test_rccl_world1_gathers_match_records(gpu_lib):
    from xfeatslam_amd.extractor import Context
    L = capi.lib()
    nf, H, W, B = 512, 96, 128, 3
    blob = WT.pack_blob(WT.make_synthetic(1234, 4.0))
    ctx = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=B)
    ctx.load_weights(blob)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    comm = xd.Comm(ctx, 0, 1, "127.0.0.1", port)
    assert L.xfh_comm_rank(ctx.h) == 0 and L.xfh_comm_world(ctx.h) == 1
    assert L.xfh_comm_create(ctx.h, bytes(128), 0, 1) == 1                      # already has a communicator
    frames = synth.frames(B, H, W, seed=9)
    frames[1] = 0                                                               # a frame without keypoints: empty compact segment
    rec = ctx.rec_bytes
    d_in = capi.DeviceBuffer(frames.nbytes).upload(frames)
    d_rec = [capi.DeviceBuffer(B * rec), capi.DeviceBuffer(B * rec)]
    d_all = capi.DeviceBuffer(B * rec)
    want = None
    for gen in (0, 1, 0):                                                       # ping-pong generations, fence before reuse
        comm.fence(gen)
        capi.check(L.xfh_extract_batch_device(ctx.h, d_in.ptr, B, H, W, 0, 0, d_rec[gen].ptr), ctx.h)
        comm.allgather_records(d_rec[gen].ptr, B, d_all.ptr, gen)
        comm.synchronize()
        got = d_all.download(np.uint8, B * rec)
        ctx.synchronize()
        local = d_rec[gen].download(np.uint8, B * rec)
        assert np.array_equal(got, local)
        want = local
    # gather-to-root (send / recv group + local copy) gives the same bytes
    d_all2 = capi.DeviceBuffer(B * rec)
    comm.gather_records_root(d_rec[0].ptr, B, d_all2.ptr, 0, 0)
    comm.synchronize()
    assert np.array_equal(d_all2.download(np.uint8, B * rec), want)
    # compact gather: header + valid rows only; unpack restores the padded record of every frame
    cap = int(L.xfh_compact_bytes_max(nf, B))
    d_c = capi.DeviceBuffer(cap)
    sizes = comm.gather_compact_root(d_rec[0].ptr, B, d_c.ptr, 0, 0)
    comm.synchronize()
    recs = ctx.parse_records(want, B)
    n_rows = sum(r[2] for r in recs)
    assert sizes[0] <= cap and sizes[0] == 256 + ((B * 16 + 255) & ~255) + ((n_rows * 28 + 255) & ~255) + n_rows * 256
    assert recs[1][2] == 0 and n_rows > 0 and sizes[0] < B * rec
    shard = d_c.download(np.uint8, sizes[0])
    for b in range(B):
        k = np.zeros(nf, capi.KP_DTYPE); d = np.zeros((nf, 64), np.float32); nv, mono = C.c_int(), C.c_int()
        assert L.xfh_unpack_compact(shard.ctypes.data, shard.nbytes, b, nf, k.ctypes.data, d.ctypes.data, C.byref(nv), C.byref(mono)) == 0
        assert (nv.value, mono.value) == (recs[b][2], recs[b][3])
        assert np.array_equal(k, recs[b][0]) and np.array_equal(d, recs[b][1])
    # the timing / barrier helper of bench.py
    assert comm.barrier_max(1.5) == 1.5
    comm.close()
    assert L.xfh_allgather_records(ctx.h, d_rec[0].ptr, B, d_all.ptr, 0) == 1   # no communicator any more
    ctx.close()
