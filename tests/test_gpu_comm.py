"""GPU tests of the multi-GPU exchange behind the C ABI (xfh_comm_*, xfh_allgather_records, xfh_gather_records_root,
xfh_gather_compact_root) with a world of ONE rank: librccl is really called (communicator, all-gather, send/recv group,
stream ordering against the extraction), only the peers are missing.  Worlds of 2 and 3 ranks run in tests/test_gpu_comm_world2.py
(over a test-only librccl stand-in: the box has one GPU) and in the driver's scaling bench (real RCCL); the partitioning
arithmetic and the TCP bootstrap are covered on CPU (tests/test_dist_gloo.py)."""
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
    # which RCCL this is (round 6): the file ncclAllGather lives in, its ncclGetVersion, and the HIP runtime on both sides -- a REAL librccl here
    which = L.xfh_comm_library().decode()
    print(f"\n  real RCCL, world 1: {which}", flush=True)
    assert "librccl" in which and "stubs" not in which and "RCCL 0.0.0" not in which, which
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


def test_several_ctx_feed_one_communicator(gpu_lib):
    """bench.py's layout: sub-batches of a step on separate ctx write one record buffer, ctx 0 owns the communicator.
    xfh_comm_wait_ctx orders the collective after the other ctx' streams, xfh_comm_fence_ctx keeps them from overwriting a
    generation the collective is still reading -- no host synchronisation between the steps."""
    from xfeatslam_amd.extractor import Context
    L = capi.lib()
    nf, H, W, B, S = 256, 96, 128, 6, 3
    blob = WT.pack_blob(WT.make_synthetic(1234, 6.0))
    ctxs = [Context(nfeatures=nf, max_height=H, max_width=W, max_batch=B) for _ in range(S)]
    for c in ctxs:
        c.load_weights(blob)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    comm = xd.Comm(ctxs[0], 0, 1, "127.0.0.1", port)
    steps = 5
    frames = synth.frames(steps * S * B, H, W, seed=21)
    rec = ctxs[0].rec_bytes
    d_in = capi.DeviceBuffer(frames.nbytes).upload(frames)
    d_rec = [capi.DeviceBuffer(S * B * rec) for _ in range(2)]
    d_all = [capi.DeviceBuffer(S * B * rec) for _ in range(steps)]          # one gather target per step, compared at the end
    for step in range(steps):
        g = step & 1
        comm.fence(g)
        for c in ctxs[1:]:
            comm.fence_ctx(c, g)
        for k, c in enumerate(ctxs):
            off = (step * S + k) * B
            capi.check(L.xfh_extract_batch_device(c.h, d_in.ptr + off * H * W, B, H, W, 0, 64, d_rec[g].ptr + k * B * rec), c.h)
        for c in ctxs[1:]:
            comm.wait_ctx(c)
        comm.allgather_records(d_rec[g].ptr, S * B, d_all[step].ptr, g)
    comm.synchronize()
    for c in ctxs:
        c.synchronize()
    ref = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=B, flags=capi.FLAG_SERIAL_BRANCH); ref.load_weights(blob)
    for step in range(steps):
        got = ctxs[0].parse_records(d_all[step].download(np.uint8, S * B * rec), S * B)
        want = []
        for k in range(S):
            off = (step * S + k) * B
            want += ref.extract_batch(frames[off:off + B], (0, 64))
        for i, (a, b) in enumerate(zip(got, want)):
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (step, i)
    # the compact gather of ALL sub-batches through ctx 0's communicator: S * B = 18 frames although ctx 0 was created for 6
    # (bench.py --gather compact with several ctx: the scratch of the exchange follows the call's frame count, not cfg.max_batch)
    last = (steps - 1) & 1
    cap = int(L.xfh_compact_bytes_max(nf, S * B))
    d_c = capi.DeviceBuffer(cap)
    sizes = comm.gather_compact_root(d_rec[last].ptr, S * B, d_c.ptr, 0, last)
    comm.synchronize()
    shard = d_c.download(np.uint8, sizes[0])
    want_last = ctxs[0].parse_records(d_all[steps - 1].download(np.uint8, S * B * rec), S * B)
    for i in (0, B, S * B - 1):
        k = np.zeros(nf, capi.KP_DTYPE); d = np.zeros((nf, 64), np.float32); nv, mono = C.c_int(), C.c_int()
        assert L.xfh_unpack_compact(shard.ctypes.data, shard.nbytes, i, nf, k.ctypes.data, d.ctypes.data, C.byref(nv), C.byref(mono)) == 0
        assert (nv.value, mono.value) == (want_last[i][2], want_last[i][3]) and np.array_equal(k, want_last[i][0]) and np.array_equal(d, want_last[i][1]), i
    assert L.xfh_comm_wait_ctx(ctxs[1].h, ctxs[0].h) == 1                      # ctx 1 has no communicator
    comm.close(); ref.close()
    for c in ctxs:
        c.close()
