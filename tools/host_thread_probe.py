"""Would host-driven lanes beat the in-order lanes of csrc/pipeline.cpp?  N threads, each with its OWN ctx of `S` frames and its own pinned buffers, loop
{ blocking H2D copy; xfh_extract_batch_device; synchronize; blocking D2H copy } -- no copy command ever sits in a stream in front of a kernel, no GPU-side
event wait anywhere; ctypes releases the GIL inside every call.  usage: python tools/host_thread_probe.py [frames_total] ["SxN,SxN"]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from xfeatslam_amd import capi, synth, weights as WT  # noqa: E402
from xfeatslam_amd.extractor import Context  # noqa: E402

L = capi.lib()
H, W, NF = 480, 640, 4096
TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SHAPES = [tuple(int(v) for v in t.split("x")) for t in (sys.argv[2] if len(sys.argv) > 2 else "64x4,64x6,64x8,32x8,32x12").split(",")]
blob = WT.pack_blob(WT.make_synthetic(1234, 6.0))
for S, NT in SHAPES:
    frames = synth.frames(S, H, W, seed=42)
    ctxs, bufs = [], []
    for t in range(NT):
        c = Context(nfeatures=NF, max_height=H, max_width=W, max_batch=S)
        c.load_weights(blob)
        hin = capi.HostBuffer(frames.nbytes); hin.array[:] = frames.reshape(-1)
        hout = capi.HostBuffer(S * c.rec_bytes)
        din = capi.DeviceBuffer(frames.nbytes); drec = capi.DeviceBuffer(S * c.rec_bytes)
        ctxs.append(c); bufs.append((hin, hout, din, drec))
    per = TOTAL // (S * NT)

    def work(t, n):
        c = ctxs[t]; hin, hout, din, drec = bufs[t]
        for _ in range(n):
            capi.check(L.xfh_memcpy_h2d(din.ptr, hin.ptr, S * H * W))
            capi.check(L.xfh_extract_batch_device(c.h, din.ptr, S, H, W, 0, 0, drec.ptr), c.h)
            c.synchronize()
            capi.check(L.xfh_memcpy_d2h(hout.ptr, drec.ptr, S * c.rec_bytes))
    for n in (2, per):
        th = [threading.Thread(target=work, args=(t, n)) for t in range(NT)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
    nfr = per * S * NT
    print(f"sub-batch {S:3d} x {NT:2d} host threads: {nfr / dt:8.0f} frames/s host to host ({nfr} frames in {dt * 1e3:.1f} ms), PCIe out {nfr * ctxs[0].rec_bytes / dt / 1e9:.1f} GB/s", flush=True)
    for c in ctxs:
        c.close()
    for b in bufs:
        for x in b:
            x.free()
