#!/usr/bin/env python
"""Stress of the host-visible batch pipeline (xfh_extract_batch_submit / _wait / _drain): the section of
tests/test_gpu_extract.py::test_host_visible_batch_pipeline that keeps three submissions outstanding, repeated; on a mismatch it says
which frames of which submission differ and in what.  usage: pipeline_stress.py [iterations] [lanes] [S]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from xfeatslam_amd import capi, synth, weights as WT
from xfeatslam_amd.extractor import Context
L = capi.lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 6
H, W, nf = 96, 128, 256
n = 5 * S + 2
blob = WT.pack_blob(WT.make_synthetic(1234, 6.0))
fr = synth.frames(n, H, W, seed=123); fr[7] = 0
ref = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=n, flags=capi.FLAG_SERIAL_BRANCH); ref.load_weights(blob)
rb = ref.rec_bytes
d_in = capi.DeviceBuffer(fr.nbytes).upload(fr); d_rec = capi.DeviceBuffer(n * rb)
capi.check(L.xfh_extract_batch_device(ref.h, d_in.ptr, n, H, W, 0, 64, d_rec.ptr), ref.h); ref.synchronize()
want = ref.parse_records(d_rec.download(np.uint8, n * rb), n)
ctx = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=S); ctx.load_weights(blob)
hin = capi.HostBuffer(fr.nbytes); hin.array[:] = fr.reshape(-1)
houts = [capi.HostBuffer(n * rb) for _ in range(3)]
if os.environ.get("STRESS_LIKE_TEST"):                # the test's order: a blocking call from pageable memory first
    out = np.zeros(n * rb, np.uint8)
    capi.check(L.xfh_extract_batch(ctx.h, fr.ctypes.data, n, H, W, 0, 64, out.ctypes.data), ctx.h)
    got = ctx.parse_records(out, n)
    for i, (x, y) in enumerate(zip(got, want)):
        if not (np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2:] == y[2:]): print("blocking call: frame", i, "differs", flush=True)
assert L.xfh_pipeline_lanes(ctx.h, lanes) == 0
bad = 0
for it in range(iters):
    for h in houts:
        h.array[:] = 0
        capi.check(L.xfh_extract_batch_submit(ctx.h, hin.ptr, n, H, W, 0, 64, h.ptr), ctx.h)
    capi.check(L.xfh_extract_batch_drain(ctx.h), ctx.h)
    for j, h in enumerate(houts):
        got = ctx.parse_records(np.ascontiguousarray(h.array), n)
        for i, (x, y) in enumerate(zip(got, want)):
            kp_ok, d_ok, hdr_ok = np.array_equal(x[0], y[0]), np.array_equal(x[1], y[1]), x[2:] == y[2:]
            if not (kp_ok and d_ok and hdr_ok):
                bad += 1
                nd = int((x[1] != y[1]).any(axis=1).sum()) if x[1].shape == y[1].shape else -1
                nk = int((x[0] != y[0]).sum()) if x[0].shape == y[0].shape else -1
                print(f"iter {it} submission {j} frame {i} (sub-batch {i // S}, slot {i % S}, lane {(j * ((n + S - 1) // S) + i // S) % lanes}?): header {x[2:]} vs {y[2:]}, "
                      f"{nk} keypoint rows differ, {nd} descriptor rows differ", flush=True)
print(f"{iters} iterations x 3 submissions x {n} frames: {bad} bad records; lanes {lanes} S {S} "
      f"NO_RIDE={os.environ.get('XFH_NO_RIDE')} LEGACY={os.environ.get('XFH_SELECT_LEGACY')}", flush=True)
os._exit(0)
