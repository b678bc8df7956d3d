"""Host-visible batched extraction (xfh_extract_batch_submit / _wait, csrc/pipeline.cpp): frames/s host to host for a few
(sub-batch, lanes) shapes, blocking calls and double-buffered asynchronous steps.  usage: python tools/host_batch_probe.py [frames_per_step]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from xfeatslam_amd import capi, synth, weights as WT  # noqa: E402
from xfeatslam_amd.extractor import Context  # noqa: E402

L = capi.lib()
H, W, NF = 480, 640, 4096
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
blob = WT.pack_blob(WT.make_synthetic(1234, 6.0))
frames = synth.frames(N, H, W, seed=42)
hin = capi.HostBuffer(frames.nbytes); hin.array[:] = frames.reshape(-1)
SHAPES = [tuple(int(v) for v in t.split("x")) for t in (sys.argv[2] if len(sys.argv) > 2 else "64x4,32x4,32x8,64x2,128x2,16x8").split(",")]
print("frames per step", N, "serial-branch", os.environ.get("XFH_PROBE_SERIAL", "0"), "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES", "default"), flush=True)
for S, lanes in SHAPES:
    ctx = Context(nfeatures=NF, max_height=H, max_width=W, max_batch=S, flags=capi.FLAG_SERIAL_BRANCH if os.environ.get("XFH_PROBE_SERIAL") else 0)
    ctx.load_weights(blob)
    assert L.xfh_pipeline_lanes(ctx.h, lanes) == 0
    rb = ctx.rec_bytes
    houts = [capi.HostBuffer(N * rb) for _ in range(2)]
    for _ in range(3):
        capi.check(L.xfh_extract_batch(ctx.h, hin.ptr, N, H, W, 0, 0, houts[0].ptr), ctx.h)
    K = 10
    t0 = time.perf_counter()
    for _ in range(K):
        capi.check(L.xfh_extract_batch(ctx.h, hin.ptr, N, H, W, 0, 0, houts[0].ptr), ctx.h)
    blocking = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    capi.check(L.xfh_extract_batch_submit(ctx.h, hin.ptr, N, H, W, 0, 0, houts[0].ptr), ctx.h)
    for t in range(1, K):
        capi.check(L.xfh_extract_batch_submit(ctx.h, hin.ptr, N, H, W, 0, 0, houts[t & 1].ptr), ctx.h)
        capi.check(L.xfh_extract_batch_wait(ctx.h), ctx.h)
    capi.check(L.xfh_extract_batch_wait(ctx.h), ctx.h)
    piped = (time.perf_counter() - t0) / K
    # three record buffers, two steps submitted ahead
    h3 = houts + [capi.HostBuffer(N * rb)]
    K3 = 24
    t0 = time.perf_counter()
    capi.check(L.xfh_extract_batch_submit(ctx.h, hin.ptr, N, H, W, 0, 0, h3[0].ptr), ctx.h)
    capi.check(L.xfh_extract_batch_submit(ctx.h, hin.ptr, N, H, W, 0, 0, h3[1].ptr), ctx.h)
    for t in range(2, K3):
        capi.check(L.xfh_extract_batch_submit(ctx.h, hin.ptr, N, H, W, 0, 0, h3[t % 3].ptr), ctx.h)
        capi.check(L.xfh_extract_batch_wait(ctx.h), ctx.h)
    capi.check(L.xfh_extract_batch_wait(ctx.h), ctx.h); capi.check(L.xfh_extract_batch_wait(ctx.h), ctx.h)
    piped3 = (time.perf_counter() - t0) / K3
    h3[2].free()
    nv = int(houts[0].array[:4].view(np.int32)[0])
    print(f"sub-batch {S:4d} lanes {lanes}: blocking {N / blocking:8.0f} frames/s ({blocking * 1e3:6.2f} ms / {N}), double-buffered {N / piped:8.0f} frames/s "
          f"({piped * 1e3:6.2f} ms), two ahead {N / piped3:8.0f}, PCIe out {N * rb / piped / 1e9:5.1f} GB/s in {N * H * W / piped / 1e9:4.1f} GB/s, n_valid[0] {nv}", flush=True)
    for h in houts:
        h.free()
    ctx.close()
