"""k_dist_mfma back to back on device-resident rows (settled clock): python tools/dist_probe.py [n] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from xfeatslam_amd import capi, synth
from xfeatslam_amd.extractor import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
L = capi.lib()
ctx = Context(nfeatures=n, max_height=64, max_width=64)
d1, d2 = synth.descriptor_sets(n, n, noise=0.3)
b1 = capi.DeviceBuffer(d1.nbytes).upload(d1); b2 = capi.DeviceBuffer(d2.nbytes).upload(d2)
out = capi.DeviceBuffer(4 * n * n)
for rep in range(3):
    for _ in range(20):
        capi.check(L.xfh_distance_i32_device(ctx.h, b1.ptr, n, b2.ptr, n, out.ptr), ctx.h)
    ctx.synchronize()
    ctx.timing_enable(capi.K["DIST_I32"])
    t0 = time.perf_counter()
    for _ in range(iters):
        capi.check(L.xfh_distance_i32_device(ctx.h, b1.ptr, n, b2.ptr, n, out.ptr), ctx.h)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / iters * 1e6
    nl, ms = ctx.timing_read(); ctx.timing_enable(0)
    print(f"n={n}: wall {wall:.1f} us per launch, dispatch events {ms / max(nl, 1) * 1e3:.1f} us ({nl} launches), {4.0 * n * n / wall / 1e3:.0f} GB/s written, "
          f"{2.0 * n * n * 64 / wall / 1e6:.1f} TFLOP/s")
