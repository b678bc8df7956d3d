"""The top-k stage (k_select) alone, on crafted candidate sets, against numpy's sort of the same keys.

Reference semantics: torch::argsort(scores, descending) + the first nfeatures (src/XFextractor.cc:285-295), validity score > 0
(:313) and the lapping-area placement (:310-343).  The keys are unique, so the expected output is simply the sorted prefix; the
cases are the distributions that steer the kernel's bucket ranking and its fallbacks (xfeatslam_amd/csrc/kernels_misc.hip)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
W = 640


def make_keys(scores, idx):
    u = np.asarray(scores, np.float32).view(np.uint32).astype(np.uint64)
    ordv = np.where(u >> 31 == 1, u ^ 0xFFFFFFFF, u ^ 0x80000000)
    return ((~ordv & 0xFFFFFFFF) << np.uint64(32)) | np.asarray(idx, np.uint64)


def cases():
    r = np.random.default_rng(7)

    def idx(n):
        return r.choice(480 * 640 - 1, n, replace=False) + 1
    out = {}
    out["log_uniform_9000"] = (np.exp(r.uniform(np.log(1e-3), 0.0, 9000)), idx(9000))
    out["fewer_than_nfeatures"] = (r.uniform(0.01, 1.0, 300), idx(300))
    out["empty"] = (np.zeros(0), np.zeros(0, np.int64))
    out["one"] = (np.array([0.5]), np.array([77]))
    out["more_than_the_registers_hold_20000"] = (np.exp(r.uniform(np.log(1e-4), 0.0, 20000)), idx(20000))
    s = np.exp(r.uniform(np.log(1e-2), 0.0, 7000)); s[:6000] = 0.25                       # 6000 equal scores: one bucket, the fallback
    out["six_thousand_ties"] = (s, idx(7000))
    s = np.exp(r.uniform(np.log(1e-3), 0.0, 8000)); s[0] = -1.0; s[1] = 1e-30; s[2] = 0.0; s[3] = -0.0   # the (0,0) candidate's -1, outliers
    i = idx(8000); i[0] = 0
    out["negative_zero_and_tiny_outliers"] = (s, i)
    out["narrow_range"] = (np.float32(0.5) + np.arange(6000, dtype=np.float32) * np.float32(2.0 ** -24), idx(6000))
    out["all_equal_scores"] = (np.full(5000, 0.125), idx(5000))
    out["mostly_invalid"] = (np.concatenate([r.uniform(0.1, 1.0, 50), -r.uniform(0.1, 1.0, 5000)]), idx(5050))
    return out


CASES = cases()


@pytest.mark.parametrize("nfeatures", [4096, 1000])
@pytest.mark.parametrize("name", sorted(CASES))
def test_select_stage_on_crafted_candidates(name, nfeatures):
    from xfeatslam_amd import capi
    from xfeatslam_amd.extractor import Context
    lib = capi.lib()
    scores, idx = CASES[name]
    keys = make_keys(scores, idx)
    assert len(np.unique(keys)) == len(keys)
    order = np.sort(keys)
    N = min(len(keys), nfeatures)
    want = order[:N]
    sc = (~(want >> np.uint64(32)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    sc = np.where(sc >> 31 == 1, sc ^ 0x80000000, sc ^ 0xFFFFFFFF).astype(np.uint32).view(np.float32)
    lap0, lap1 = 100, 300
    x = (want & np.uint64(0xFFFFFFFF)).astype(np.int64) % W
    valid = sc > 0
    back = valid & (x >= lap0) & (x <= lap1)
    ctx = Context(nfeatures=nfeatures, max_height=480, max_width=640, max_batch=1)
    try:
        for form in (0, 1, 2):
            sel = np.zeros(nfeatures, np.uint64); n_out = C.c_int(-1); hdr = np.zeros(4, np.int32)
            kbuf = np.ascontiguousarray(keys)
            capi.check(lib.xfh_debug_select(ctx.h, kbuf.ctypes.data_as(C.c_void_p), len(keys), W, lap0, lap1, form,
                                            sel.ctypes.data_as(C.c_void_p), C.byref(n_out), hdr.ctypes.data_as(C.c_void_p)), ctx.h)
            assert n_out.value == N, (name, form)
            assert np.array_equal(sel[:N], want), (name, form, int(np.argmax(sel[:N] != want)))
            assert hdr.tolist() == [int(valid.sum()), int((valid & ~back).sum()), len(keys), 0], (name, form)
    finally:
        ctx.close()


def test_select_forms_agree_on_frames(tmp_path):
    """whole extraction with k_select forced to its radix + bitonic form, with the keypoint branch on a second stream instead of riding on
    the backbone's launches, and with the heatmap head as a launch of its own == the records of the shipped library.  The knobs that force
    those forms exist in the debug build only (libxfeat_hip_knobs.so, -DXFH_TEST_KNOBS): each variant runs in a worker process that
    loads that build; the shipped library ignores the variables (checked first)."""
    import os
    import subprocess
    import sys
    from conftest import records_equal
    from xfeatslam_amd import capi, synth, weights as WT
    from xfeatslam_amd.extractor import Context
    lib = capi.lib()
    assert os.path.exists(capi.KNOBS_LIB_PATH), "run `make -C xfeatslam_amd/csrc knobs` (or __graft_entry__.build())"
    frames = synth.frames(3, 480, 640, seed=11)
    blob = WT.pack_blob(WT.make_synthetic(1234, 3.0))
    os.environ["XFH_SELECT_LEGACY"] = "1"; os.environ["XFH_NO_RIDE"] = "1"        # the shipped build must not even look at these
    try:
        ctx = Context(nfeatures=4096, max_height=480, max_width=640, max_batch=3)
    finally:
        del os.environ["XFH_SELECT_LEGACY"]; del os.environ["XFH_NO_RIDE"]
    try:
        ctx.load_weights(blob)
        din = capi.DeviceBuffer(frames.nbytes).upload(frames); rec = capi.DeviceBuffer(3 * ctx.rec_bytes)
        capi.check(lib.xfh_extract_batch_device(ctx.h, din.ptr, 3, 480, 640, 150, 400, rec.ptr), ctx.h)
        ctx.synchronize()
        base = rec.download(np.uint8, 3 * ctx.rec_bytes)
        worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "workers", "knob_worker.py")
        for knob in (None, "XFH_SELECT_LEGACY", "XFH_NO_RIDE", "XFH_NO_NMS_HEAT"):
            env = dict(os.environ, XFEAT_HIP_LIB=capi.KNOBS_LIB_PATH)
            if knob:
                env[knob] = "1"
            out = str(tmp_path / f"rec_{knob}.npy")
            r = subprocess.run([sys.executable, worker, out], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (knob, r.stderr[-2000:])
            assert records_equal(ctx, base, np.load(out), 3), knob
    finally:
        ctx.close()


def test_select_stage_is_stable_over_many_launches():
    """the same candidate set 400 times through the bucket ranking (a set that takes its second level): every launch must give the
    sorted prefix.  Round 3 had a barrier missing between a read and a reset of an LDS flag there: one launch in some hundred, on some
    boxes only, left the level loop with part of its waves and returned a wrong selection for that frame."""
    from xfeatslam_amd import capi
    from xfeatslam_amd.extractor import Context
    lib = capi.lib()
    r = np.random.default_rng(3)
    for n, nfeatures in ((440, 256), (9000, 4096)):
        scores = np.exp(r.uniform(np.log(2e-2), np.log(0.2), n))            # a narrow spread: crowded buckets at the first level
        idx = r.choice(96 * 128 - 1, n, replace=False) + 1 if n < 96 * 128 else r.choice(480 * 640 - 1, n, replace=False) + 1
        keys = np.ascontiguousarray(make_keys(scores, idx))
        want = np.sort(keys)[:min(n, nfeatures)]
        ctx = Context(nfeatures=nfeatures, max_height=480, max_width=640, max_batch=1)
        try:
            sel = np.zeros(nfeatures, np.uint64); n_out = C.c_int(-1); hdr = np.zeros(4, np.int32)
            for it in range(400):
                capi.check(lib.xfh_debug_select(ctx.h, keys.ctypes.data_as(C.c_void_p), n, W, 0, 0, 1,
                                                sel.ctypes.data_as(C.c_void_p), C.byref(n_out), hdr.ctypes.data_as(C.c_void_p)), ctx.h)
                assert n_out.value == len(want) and np.array_equal(sel[:len(want)], want), (n, it)
        finally:
            ctx.close()
