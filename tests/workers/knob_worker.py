"""Worker of tests/test_gpu_select.py::test_select_forms_agree_on_frames: one extraction of three VGA frames through the DEBUG build of the
library (libxfeat_hip_knobs.so, `make -C xfeatslam_amd/csrc knobs`) with one test knob set in the environment; the records go to a file.
The knobs exist in that build only -- the shipped libxfeat_hip.so has one path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    import numpy as np
    from xfeatslam_amd import capi, synth, weights as WT
    from xfeatslam_amd.extractor import Context
    assert capi.LIB_PATH.endswith("libxfeat_hip_knobs.so"), capi.LIB_PATH
    lib = capi.lib()
    frames = synth.frames(3, 480, 640, seed=11)
    blob = WT.pack_blob(WT.make_synthetic(1234, 3.0))
    ctx = Context(nfeatures=4096, max_height=480, max_width=640, max_batch=3)
    ctx.load_weights(blob)
    din = capi.DeviceBuffer(frames.nbytes).upload(frames); rec = capi.DeviceBuffer(3 * ctx.rec_bytes)
    capi.check(lib.xfh_extract_batch_device(ctx.h, din.ptr, 3, 480, 640, 150, 400, rec.ptr), ctx.h)
    ctx.synchronize()
    np.save(out_path, rec.download(np.uint8, 3 * ctx.rec_bytes))
    ctx.close()


if __name__ == "__main__":
    main()
