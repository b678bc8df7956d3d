#!/usr/bin/env python
"""Matcher micro-benchmark on the GPU box: correctness vs oracle + per-kernel times for each
XFH_GEMM_VARIANT (one subprocess per variant).  Development tool."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def child():
    from xfeatslam_amd import capi, synth
    from xfeatslam_amd.extractor import Context
    from oracle import oracle as O
    lib = capi.lib()
    ctx = Context(nfeatures=64, max_height=32, max_width=32)
    ok = True
    for (n1, n2, z) in [(300, 200, 7), (129, 127, 0), (4096, 4096, 100), (1000, 4096, 0)]:
        d1, d2 = synth.descriptor_sets(n1, n2, zero_rows=z, noise=0.3)
        d2[5] = d2[2]
        a = O.match_mnn(d1, d2); b = ctx.match_mnn(d1, d2)
        ok &= np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2], equal_nan=True)
    n = 4096
    d1, d2 = synth.descriptor_sets(n, n, noise=0.3)
    b1 = capi.DeviceBuffer(d1.nbytes).upload(d1); b2 = capi.DeviceBuffer(d2.nbytes).upload(d2)
    o = capi.DeviceBuffer(n * 12 + 64)
    def run():
        capi.check(lib.xfh_match_mnn_device(ctx.h, b1.ptr, n, b2.ptr, n, -1.0, o.ptr, o.ptr + 4 * n, o.ptr + 8 * n, o.ptr + 12 * n), ctx.h)
    for _ in range(20): run()
    ctx.synchronize()
    res = []
    for rep in range(3):
        ctx.timing_enable(capi.K["MNN_GEMM"])
        for _ in range(200): run()
        nl, ms = ctx.timing_read()
        ctx.timing_enable(0)
        t = time.perf_counter()
        for _ in range(200): run()
        ctx.synchronize()
        dt = (time.perf_counter() - t) / 200
        res.append((ms / nl * 1e3, dt * 1e6))
    g = min(r[0] for r in res); w = min(r[1] for r in res)
    print(f"variant {os.environ.get('XFH_GEMM_VARIANT','0')}: parity={ok} gemm {g:.2f} us ({2.0*n*n*64/(g*1e-6)/1e12:.1f} TF, {2.0*n*n*64/(g*1e-6)/157.3e12*100:.1f}%)  whole call {w:.1f} us ({n*n/(w*1e-6):.3e} pairs/s)  all: {res}", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for v in sys.argv[1:] or ["0", "1", "2"]:
            env = dict(os.environ, XFH_GEMM_VARIANT=v)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
