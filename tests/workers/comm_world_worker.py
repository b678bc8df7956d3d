"""One rank of the multi-rank exchange test (tests/test_gpu_comm_world2.py): BASELINE.json configs[3] in small -- a global
batch of frames sharded frame i -> rank i mod R, every rank extracts its shard on ITS ctx, and the records travel through
xfh_comm_* (csrc/comm.cpp) in all three forms.  All ranks run on GPU 0 of the test box; librccl is the TEST-ONLY stand-in of
tests/stubs (real RCCL refuses two ranks on one device), found through LD_LIBRARY_PATH by comm.cpp's dlopen.

usage: comm_world_worker.py RANK WORLD PORT N_FRAMES OUT_DIR
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

NF, H, W, STEPS = 384, 96, 128, 3


def step_frames(n, step):
    from xfeatslam_amd import synth
    fr = synth.frames(n, H, W, seed=100 + 17 * step)
    if step == 1:
        fr[n // 2] = 0                       # a frame without keypoints: an empty segment in the compact shard
    return fr


def main():
    rank, world, port, n, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import ctypes as C
    from xfeatslam_amd import capi, dist as xd, weights as WT
    from xfeatslam_amd.extractor import Context
    L = capi.lib()
    plan = xd.ShardPlan(n, rank, world)
    S = plan.slots
    ctx = Context(nfeatures=NF, max_height=H, max_width=W, max_batch=S)
    ctx.load_weights(WT.pack_blob(WT.make_synthetic(1234, 6.0)))
    comm = xd.Comm(ctx, rank, world, "127.0.0.1", port)
    assert "tests/stubs/librccl.so.1" in open("/proc/self/maps").read(), "the worker must run on the test stand-in, not on a real librccl"
    assert L.xfh_comm_rank(ctx.h) == rank and L.xfh_comm_world(ctx.h) == world
    rec = ctx.rec_bytes
    d_in = capi.DeviceBuffer(S * H * W)
    d_rec = [capi.DeviceBuffer(S * rec) for _ in range(2)]
    d_ag = [capi.DeviceBuffer(world * S * rec) for _ in range(STEPS)]
    # ---- all-gather, ping-pong generations with fences, no host synchronisation between the steps -------------------
    ins = []
    for step in range(STEPS):
        g = step & 1
        comm.fence(g)
        local = np.ascontiguousarray(step_frames(n, step)[plan.local])
        buf = capi.DeviceBuffer(local.nbytes).upload(local); ins.append(buf)
        capi.check(L.xfh_extract_batch_device(ctx.h, buf.ptr, S, H, W, 0, 64, d_rec[g].ptr), ctx.h)
        comm.allgather_records(d_rec[g].ptr, S, d_ag[step].ptr, g)
    comm.synchronize(); ctx.synchronize()
    for step in range(STEPS):
        np.save(os.path.join(out_dir, f"allgather_s{step}_r{rank}.npy"), d_ag[step].download(np.uint8, world * S * rec))
    g_last = (STEPS - 1) & 1                  # d_rec[g_last] holds the records of the last step
    # ---- gather to a root (send / recv group, r * nb offsets), both roots ---------------------------------------------
    for root in range(world):
        d_all = capi.DeviceBuffer(world * S * rec if rank == root else 16)
        comm.gather_records_root(d_rec[g_last].ptr, S, d_all.ptr if rank == root else 0, root, g_last)
        comm.synchronize()
        if rank == root:
            np.save(os.path.join(out_dir, f"root{root}.npy"), d_all.download(np.uint8, world * S * rec))
        d_all.free()
    # ---- compact gather (sizes exchanged first, exact counts on the wire) on the step with the empty frame ----------
    local = np.ascontiguousarray(step_frames(n, 1)[plan.local])
    d_in.upload(local)
    comm.fence(0)
    capi.check(L.xfh_extract_batch_device(ctx.h, d_in.ptr, S, H, W, 0, 64, d_rec[0].ptr), ctx.h)
    cap = int(L.xfh_compact_bytes_max(NF, S))
    d_c = capi.DeviceBuffer(world * cap if rank == 0 else 16)
    sizes = comm.gather_compact_root(d_rec[0].ptr, S, d_c.ptr if rank == 0 else 0, 0, 0)
    comm.synchronize()
    if rank == 0:
        out = np.zeros((world, S, rec), np.uint8)
        koff, doff = ctx.kps_off, ctx.desc_off
        for r in range(world):
            assert 0 < sizes[r] <= cap
            shard = d_c.download(np.uint8, sizes[r], r * cap)
            for j in range(S):
                k = np.zeros(NF, capi.KP_DTYPE); d = np.zeros((NF, 64), np.float32); nv, mono = C.c_int(), C.c_int()
                assert L.xfh_unpack_compact(shard.ctypes.data, shard.nbytes, j, NF, k.ctypes.data, d.ctypes.data, C.byref(nv), C.byref(mono)) == 0
                out[r, j, :8].view(np.int32)[:] = (nv.value, mono.value)
                out[r, j, koff:koff + 28 * NF] = k.view(np.uint8)
                out[r, j, doff:doff + 256 * NF] = d.reshape(-1).view(np.uint8)
        np.save(os.path.join(out_dir, "compact.npy"), out.reshape(-1))
        np.save(os.path.join(out_dir, "compact_sizes.npy"), np.array(sizes, np.int64))
    # ---- the barrier / max helper of bench.py ------------------------------------------------------------------------
    assert comm.barrier_max(rank + 1.5) == world + 0.5
    comm.close(); ctx.close()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
