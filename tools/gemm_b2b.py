#!/usr/bin/env python
"""k_mnn_gemm_img back to back (xfh_bench_mnn_gemm: wall time per launch) in a process with ONE small ctx, on synthetic descriptor sets --
the library's instance of the kernel under the conditions of tools/probes/mnn_probe (which links its own instance).  Optional: N extra idle ctx."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from xfeatslam_amd import capi, synth
from xfeatslam_amd.extractor import Context
lib = capi.lib(); ctx = Context(nfeatures=64, max_height=32, max_width=32)
extra = [Context(nfeatures=4096, max_height=480, max_width=640, max_batch=64) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0)]
n = 4096
d1, d2 = synth.descriptor_sets(n, n, noise=0.3)
p1, p2 = ctx.match_prepare(d1), ctx.match_prepare(d2)
for rep in range(int(os.environ.get('REPS', '3'))):
    us = C.c_double(0.0)
    capi.check(lib.xfh_bench_mnn_gemm(ctx.h, p1[0].ptr, n, p2[0].ptr, n, 300, C.byref(us)), ctx.h)
    print(f"k_mnn_gemm_img back to back, {len(extra)} idle ctx beside: {us.value:.2f} us per launch = {2.0 * n * n * 64 / (us.value * 1e-6) / 1e12 / 157.3:.3f} of 157.3 TFLOP/s", flush=True)
