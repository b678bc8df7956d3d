#!/usr/bin/env python
"""single-frame latency of the host-pointer API (what XFextractor::operator() costs a SLAM thread)"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from xfeatslam_amd import capi, synth, weights as WT
from xfeatslam_amd.extractor import XFextractor
blob = WT.pack_blob(WT.make_synthetic(1234, 6.0))
for (H, W, nf) in [(480, 640, 1000), (480, 640, 4096), (720, 1280, 2000)]:
    ex = XFextractor(nf, 1.2, 8, 20, 7, weights=blob, max_height=H, max_width=W)
    img = synth.image(H, W, 42)
    for _ in range(5): ex(img)
    ts = []
    for _ in range(50):
        t = time.perf_counter(); ex(img); ts.append(time.perf_counter() - t)
    if "--paced" in sys.argv:                     # one call every 33 ms, as a 30-Hz SLAM loop would: the GPU idles (and clocks down) between frames
        tp = []
        for _ in range(40):
            time.sleep(0.033); t = time.perf_counter(); ex(img); tp.append(time.perf_counter() - t)
        tp = np.array(tp) * 1e3
        print(f"{H}x{W} nfeatures {nf}: one call every 33 ms: median {np.median(tp):.3f} ms  min {tp.min():.3f}  p90 {np.percentile(tp, 90):.3f}", flush=True)
    ts = np.array(ts) * 1e3
    print(f"{H}x{W} nfeatures {nf}: xfh_extract median {np.median(ts):.3f} ms  min {ts.min():.3f}  p90 {np.percentile(ts, 90):.3f}  (n_valid {ex.n_valid})", flush=True)
    d1 = np.zeros((nf, 64), np.float32)
    ret, k, d = ex(img)
    ts = []
    for _ in range(30):
        t = time.perf_counter(); ex.ctx.match_mnn(d, d); ts.append(time.perf_counter() - t)
    print(f"   xfh_match_mnn {nf}x{nf} host API median {np.median(np.array(ts)) * 1e3:.3f} ms", flush=True)
    ex.ctx.close()
