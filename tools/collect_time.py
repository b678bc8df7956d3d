#!/usr/bin/env python
"""where the blocking single-frame host call spends its time beyond the kernels: xfh_extract_submit (host copy in + launches), xfh_extract_collect with the
GPU already idle (host copy out of the pinned record + padding), and the blocking xfh_extract -- nfeatures 4096 and 1000"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from xfeatslam_amd import capi, synth, weights as WT
from xfeatslam_amd.extractor import Context
lib = capi.lib()
for nf in (4096, 1000):
    H, W = 480, 640
    ctx = Context(nfeatures=nf, max_height=H, max_width=W); ctx.load_weights(WT.pack_blob(WT.make_synthetic(1234, 6.0)))
    img = synth.image(H, W, 3)
    kps = np.zeros(nf, capi.KP_DTYPE); desc = np.zeros((nf, 64), np.float32)
    nv = C.c_int(0); mono = C.c_int(0)
    ts, tc, tt = [], [], []
    for it in range(200):
        t0 = time.perf_counter()
        capi.check(lib.xfh_extract_submit(ctx.h, img.ctypes.data, H, W, W, 0, 0), ctx.h)
        t1 = time.perf_counter()
        time.sleep(0.002)                     # the GPU is done long before
        t2 = time.perf_counter()
        capi.check(lib.xfh_extract_collect(ctx.h, kps.ctypes.data, desc.ctypes.data, C.byref(nv), C.byref(mono)), ctx.h)
        t3 = time.perf_counter()
        ts.append(t1 - t0); tc.append(t3 - t2)
        t4 = time.perf_counter()
        capi.check(lib.xfh_extract(ctx.h, img.ctypes.data, H, W, W, 0, 0, kps.ctypes.data, desc.ctypes.data, C.byref(nv), C.byref(mono)), ctx.h)
        tt.append(time.perf_counter() - t4)
    med = lambda v: sorted(v)[len(v) // 2] * 1e6
    print(f"nfeatures {nf}: submit (host copy in + {23} launches) {med(ts):.1f} us, collect with the GPU idle (host copy out + padding) {med(tc):.1f} us, blocking call {med(tt):.1f} us, n_valid {nv.value}")
    ctx.close()
