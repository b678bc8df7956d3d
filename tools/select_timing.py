#!/usr/bin/env python
"""k_select / k_nms_score / k_desc kernel times at B = 1 for sparse and dense candidate sets (development tool)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from xfeatslam_amd import capi, synth, weights as WT
from xfeatslam_amd.extractor import Context
lib = capi.lib()
for gain in (1.0, 3.0, 6.0):
    ctx = Context(nfeatures=4096, max_height=480, max_width=640, max_batch=1); ctx.load_weights(WT.pack_blob(WT.make_synthetic(1234, gain)))
    fr = synth.frames(1, 480, 640, seed=42)
    din = capi.DeviceBuffer(fr.nbytes).upload(fr); rec = capi.DeviceBuffer(ctx.rec_bytes)
    def call(): capi.check(lib.xfh_extract_batch_device(ctx.h, din.ptr, 1, 480, 640, 0, 0, rec.ptr), ctx.h)
    for _ in range(5): call()
    ctx.synchronize()
    hdr = rec.download(np.int32, 4)
    out = []
    for name in ("NMS", "SELECT", "DESC", "HEADS"):
        ctx.timing_enable(capi.K[name])
        for _ in range(50): call()
        n, ms = ctx.timing_read(); ctx.timing_enable(0)
        out.append(f"{name} {ms / n * 1e3:.1f} us")
    print(f"gain {gain}: n_valid {hdr[0]} candidates {hdr[2]}: " + ", ".join(out), flush=True)
    ctx.close()
