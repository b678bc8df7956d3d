#!/usr/bin/env python
"""N single-frame extractions back to back, device resident, no timers of any kind: the workload tools/b1_timeline.sh traces"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from xfeatslam_amd import capi, synth, weights as WT
from xfeatslam_amd.extractor import Context
lib = capi.lib()
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
MODE = int(os.environ.get("B1_MODE", "0"))      # BatchNorm mode: 0 batch statistics, 1 eval() statistics, 2 folded
ctx = Context(nfeatures=4096, max_height=H, max_width=W, max_batch=1, bn_mode=MODE); ctx.load_weights(WT.pack_blob(WT.make_synthetic(1234, 3.0, with_bn=MODE != 0)))
fr = synth.frames(1, H, W, seed=42)
din = capi.DeviceBuffer(fr.nbytes).upload(fr); rec = capi.DeviceBuffer(ctx.rec_bytes)
for _ in range(20): capi.check(lib.xfh_extract_batch_device(ctx.h, din.ptr, 1, H, W, 0, 0, rec.ptr), ctx.h)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(n): capi.check(lib.xfh_extract_batch_device(ctx.h, din.ptr, 1, H, W, 0, 0, rec.ptr), ctx.h)
ctx.synchronize()
print(f"{H}x{W}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per frame, back to back ({n} frames)", flush=True)
ctx.close()
