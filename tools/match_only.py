#!/usr/bin/env python
"""runs the 4096x4096 MNN match N times (for rocprofv3 passes). Development tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xfeatslam_amd import capi, synth
from xfeatslam_amd.extractor import Context
lib = capi.lib(); ctx = Context(nfeatures=64, max_height=32, max_width=32)
n = 4096
d1, d2 = synth.descriptor_sets(n, n, noise=0.3)
b1 = capi.DeviceBuffer(d1.nbytes).upload(d1); b2 = capi.DeviceBuffer(d2.nbytes).upload(d2)
o = capi.DeviceBuffer(n * 12 + 64)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    capi.check(lib.xfh_match_mnn_device(ctx.h, b1.ptr, n, b2.ptr, n, -1.0, o.ptr, o.ptr + 4 * n, o.ptr + 8 * n, o.ptr + 12 * n), ctx.h)
ctx.synchronize()
