#!/usr/bin/env python
"""Where does the time go in the pipelined host path (xfh_extract_submit / _collect, 2 frames in flight)?
Splits the wall time per frame into time inside submit (host enqueue) and inside collect (wait + copy-out).  GPU box tool."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xfeatslam_amd import capi, synth, weights as WT
from xfeatslam_amd.extractor import Context
lib = capi.lib()
H, W = 480, 640
blob = WT.pack_blob(WT.make_synthetic(1234, 6.0))
fr = [np.ascontiguousarray(f) for f in synth.frames(8, H, W, 42)]
for nf in (4096, 1000):
    hc = Context(nfeatures=nf, max_height=H, max_width=W); hc.load_weights(blob)
    k = np.zeros(nf, capi.KP_DTYPE); d = np.zeros((nf, 64), np.float32); nv, mono = C.c_int(), C.c_int()
    for i in range(20):
        capi.check(lib.xfh_extract(hc.h, fr[i % 8].ctypes.data, H, W, W, 0, 0, k.ctypes.data, d.ctypes.data, C.byref(nv), C.byref(mono)), hc.h)
    capi.check(lib.xfh_extract_submit(hc.h, fr[0].ctypes.data, H, W, W, 0, 0), hc.h)
    ts = tc = 0.0; n = 300
    t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter()
        capi.check(lib.xfh_extract_submit(hc.h, fr[(i + 1) % 8].ctypes.data, H, W, W, 0, 0), hc.h)
        b = time.perf_counter()
        capi.check(lib.xfh_extract_collect(hc.h, k.ctypes.data, d.ctypes.data, C.byref(nv), C.byref(mono)), hc.h)
        c = time.perf_counter()
        ts += b - a; tc += c - b
    tot = time.perf_counter() - t0
    capi.check(lib.xfh_extract_collect(hc.h, k.ctypes.data, d.ctypes.data, C.byref(nv), C.byref(mono)), hc.h)
    print(f"nfeatures {nf}: {tot / n * 1e3:.3f} ms/frame = submit {ts / n * 1e3:.3f} + collect {tc / n * 1e3:.3f}  (n_valid {nv.value})")
    hc.close()
