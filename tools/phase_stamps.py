#!/usr/bin/env python
"""Where a single-frame convolution launch spends its time (development tool).

Needs the stamps build (make -C xfeatslam_amd/csrc stamps -> tools/ab/stamps.so; this script swaps it in for its own run):
one workgroup of every k_conv_mfma / k_conv_mfma16 launch writes the 100 MHz wall clock at
  0 entry | 1 weight + input loads issued | 2 statistics folded | 3 tile + first weights in LDS | 4 K loop done | 5 partials written
and the table shows the phases of every layer plus the gap to the next layer's entry (launch boundary + the other workgroups)."""
import ctypes as C, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
lib_path = os.path.join(ROOT, "xfeatslam_amd", "libxfeat_hip.so"); keep = "/tmp/keep_stamps.so"
shutil.copy(lib_path, keep); shutil.copy(os.path.join(ROOT, "tools", "ab", "stamps.so"), lib_path)
try:
    from xfeatslam_amd import capi, synth, weights as WT
    from xfeatslam_amd.extractor import Context
    lib = capi.lib()
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1            # frames per call (the stamped workgroup is the middle tile of frame 0)
    ctx = Context(nfeatures=4096, max_height=H, max_width=W, max_batch=B); ctx.load_weights(WT.pack_blob(WT.make_synthetic(1234, 3.0)))
    fr = synth.frames(B, H, W, seed=42)
    din = capi.DeviceBuffer(fr.nbytes).upload(fr); rec = capi.DeviceBuffer(B * ctx.rec_bytes)
    acc = np.zeros((32, 8)); nrun = 10
    order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 23, 18, 19]
    gaps = np.zeros(len(order))
    for it in range(5 + nrun):
        for _ in range(80 if B == 1 else 2):                                  # back to back: the stamps are those of the last frame, on a GPU that is awake
            capi.check(lib.xfh_extract_batch_device(ctx.h, din.ptr, B, H, W, 0, 0, rec.ptr), ctx.h)
        ctx.synchronize()
        st = np.zeros(32 * 8, np.uint64)
        assert lib.xfh_debug_stamps(st.ctypes.data_as(C.c_void_p)) == 0
        st = st.reshape(32, 8).astype(np.int64)
        if it < 5: continue
        acc += (st - st[:, :1]) * 0.01                       # us since the layer's entry
        for i in range(len(order) - 1):
            gaps[i] += (st[order[i + 1], 0] - st[order[i], 5]) * 0.01
    acc /= nrun; gaps /= nrun
    print(f"{H}x{W}, {B} frame(s) per call; us since the workgroup's entry          (gap = entry of the next layer - this one's last stamp)")
    print("layer   loads   fold  staged   kloop   done |  gap")
    for i, l in enumerate(order):
        print(f"{l:5d} " + " ".join(f"{acc[l, p]:7.2f}" for p in range(1, 6)) + f" | {gaps[i]:5.2f}")
    ctx.close()
finally:
    shutil.copy(keep, lib_path)
