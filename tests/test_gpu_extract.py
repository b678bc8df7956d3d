"""GPU parity tests of XFextractor::operator() through the C ABI against the oracle and the
goldens: identical keypoint sets and placement, descriptors within 1e-4 (north_star), plus
stage-by-stage tensors, batching, edge cases and size-independent properties."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import CAMPAIGN_GOLDENS, EXTRACT_GOLDENS, GOLDEN, check_extract_golden, golden_inputs, joined_desc_diff, kp_set, records_equal
from xfeatslam_amd import capi, synth, weights as WT

pytestmark = pytest.mark.gpu
DESC_TOL = 1e-4


def _ctx(nf, H, W, B=1):
    from xfeatslam_amd.extractor import Context
    return Context(nfeatures=nf, max_height=H, max_width=W, max_batch=B)


@pytest.mark.parametrize("H,W,gain,nf,lap", [(96, 128, 1.0, 256, (0, 0)), (480, 640, 1.0, 4096, (0, 0)),
                                              (480, 640, 6.0, 4096, (0, 1000)), (480, 640, 6.0, 1000, (200, 400)),
                                              (720, 1280, 1.0, 4096, (0, 1000)), (170, 230, 2.0, 300, (100, 150)),
                                              (1080, 1920, 6.0, 4096, (0, 1000)),      # > 16k candidates: radix-select path
                                              (480, 640, 6.0, 8000, (0, 0)),           # nfeatures > 4096: generic sort path
                                              (64, 32, 6.0, 64, (0, 0))])              # smallest supported image
def test_extract_matches_oracle(gpu_lib, oracle_mod, H, W, gain, nf, lap):
    blob = WT.pack_blob(WT.make_synthetic(1234, gain))
    img = synth.image(H, W, 42)
    ok, od, onv, omono = oracle_mod.Oracle(blob).extract(img, nf, lap)
    ctx = _ctx(nf, H, W)
    ctx.load_weights(blob)
    (hk, hd, hnv, hmono, hnc), = ctx.extract_batch(img[None], lap)
    ctx.close()
    assert (hnv, hmono) == (onv, omono)
    assert kp_set(hk) == kp_set(ok)                                   # identical keypoint index sets
    dd, ds, n = joined_desc_diff(hk, hd, ok, od)
    assert n == onv and dd < DESC_TOL and ds < 1e-6
    pad = hk["size"] == 0
    assert np.array_equal(pad, ok["size"] == 0)                        # same slots filled (front / back)
    assert np.all(hd[pad] == 0) and np.array_equal(hk[pad], ok[pad])   # padding = default KeyPoint, zero rows
    nrm = np.linalg.norm(hd[~pad].astype(np.float64), axis=1)
    assert np.all(np.abs(nrm - 1.0) < 1e-5)                            # unit descriptors


@pytest.mark.parametrize("name", EXTRACT_GOLDENS)
def test_extract_matches_golden(gpu_lib, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    H, W, nf = int(g["H"]), int(g["W"]), int(g["nfeatures"])
    w, img = golden_inputs(g)
    ctx = _ctx(nf, H, W)
    ctx.load_weights(WT.pack_blob(w))
    (kps, desc, nv, mono, nc), = ctx.extract_batch(img[None], tuple(int(v) for v in g["lap"]))
    ctx.close()
    # candidate / valid counts, identical keypoint set, scores and sampled descriptors joined by position; differences only inside a
    # near-tie at the top-k cut (conftest.check_extract_golden prints how many)
    check_extract_golden(g, kps, desc, nv, mono, nc, DESC_TOL)


@pytest.mark.parametrize("B", [12, 40])
@pytest.mark.parametrize("name", [n for n in CAMPAIGN_GOLDENS if n.startswith("c6_vga_") and "nf1000" not in n][::2] + [n for n in CAMPAIGN_GOLDENS if "nf1000" in n][:2])
def test_batch_regimes_match_golden(gpu_lib, name, B):
    """round 6: the LARGE-BATCH kernels meet libtorch-operator output first-hand.  test_extract_matches_golden runs one frame per call, i.e. the single-frame forms
    (16x16x4 MFMA tiles, consumers folding the statistics, riders); a sub-batch of 12 runs the finalize kernels and the second stream, one of 40 the 32x32x2
    persistent / streamed forms the bench runs.  The golden's frame sits at positions 0 and B - 1 of a batch of otherwise different frames (statistics are per
    frame: the neighbours must not matter) and both records are checked against the ATen fixture with the checker of the single-frame test."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    H, W, nf = int(g["H"]), int(g["W"]), int(g["nfeatures"])
    w, img = golden_inputs(g)
    def other(i):          # a different frame, cheap to make: the golden's frame shifted and re-scaled in brightness
        return np.clip(np.roll(img, 13 * i, axis=1).astype(np.int32) * (3 + i % 5) // 4 + (i % 3) * 9, 0, 255).astype(np.uint8)
    fr = np.stack([img if i in (0, B - 1) else other(i) for i in range(B)])
    ctx = _ctx(nf, H, W, B=B)
    ctx.load_weights(WT.pack_blob(w))
    recs = ctx.extract_batch(fr, tuple(int(v) for v in g["lap"]))
    ctx.close()
    for pos in (0, B - 1):
        kps, desc, nv, mono, nc = recs[pos]
        check_extract_golden(g, kps, desc, nv, mono, nc, DESC_TOL)
    for a, b in zip(recs[0], recs[B - 1]):
        assert np.array_equal(a, b)                                   # same frame, same record, wherever it sits in the batch


def test_config3_batch8_720p_device_resident(gpu_lib, oracle_mod):
    """BASELINE.json configs[3], the per-GPU side: a batch of 8 frames of 1280x720 (resized to 1280x704 inside)
    through xfh_extract_batch_device with frames and records resident in HBM.  Frame 0 is the committed golden
    (extract_720p.npz); every frame is compared with the oracle; statistics are per frame (the batch holds the golden frame
    twice, at positions 0 and 5, and a constant frame)."""
    g = np.load(os.path.join(GOLDEN, "extract_720p.npz"))
    H, W, nf = int(g["H"]), int(g["W"]), int(g["nfeatures"])
    lap = tuple(int(v) for v in g["lap"])
    blob = WT.pack_blob(WT.make_synthetic(1234, float(g["gain"])))
    fr = np.stack([synth.image(H, W, int(g["seed"]) + (0 if i in (0, 5) else i)) for i in range(8)])
    fr[6] = 77
    L = capi.lib()
    ctx = _ctx(nf, H, W, B=8); ctx.load_weights(blob)
    d_in = capi.DeviceBuffer(fr.nbytes).upload(fr)
    d_rec = capi.DeviceBuffer(8 * ctx.rec_bytes)
    capi.check(L.xfh_extract_batch_device(ctx.h, d_in.ptr, 8, H, W, lap[0], lap[1], d_rec.ptr), ctx.h)
    ctx.synchronize()
    recs = ctx.parse_records(d_rec.download(np.uint8, 8 * ctx.rec_bytes), 8)
    ctx.close()
    kps, desc, nv, mono, nc = recs[0]
    assert (nv, mono, nc) == (int(g["n_valid"]), int(g["mono_index"]), int(g["n_candidates"]))
    assert kp_set(kps) == set(map(tuple, g["xy"].tolist()))
    pos = {(int(k["x"]), int(k["y"])): i for i, k in enumerate(kps) if k["size"] > 0}
    ridx = np.array([pos[tuple(p)] for p in g["desc_rows_xy"].tolist()])
    assert np.abs(desc[ridx] - g["desc_rows"]).max() < DESC_TOL
    for a, b in zip(recs[0], recs[5]):
        assert np.array_equal(a, b)                                   # same frame, same record, wherever it sits in the batch
    assert recs[6][2] == 0 and not recs[6][1].any()                   # constant frame: no keypoints, zero descriptors
    orc = oracle_mod.Oracle(blob)
    for b in (1, 2, 3, 4, 7):
        ok, od, onv, omono = orc.extract(fr[b], nf, lap)
        hk, hd, hnv, hmono, _ = recs[b]
        assert (hnv, hmono) == (onv, omono) and kp_set(hk) == kp_set(ok), b
        dd, ds, n = joined_desc_diff(hk, hd, ok, od)
        assert n == onv and dd < DESC_TOL and ds < 1e-6, b


def test_concurrent_contexts_share_the_gpu(gpu_lib):
    """four ctx on one GPU with work in flight at the same time (how bench.py fills the device: the HBM-bound kernels of one
    sub-batch run beside the MFMA-bound convolutions of another): back-to-back device-resident calls, no synchronisation in
    between, every record equal to the one the frame gets from a lone serial ctx."""
    from xfeatslam_amd.extractor import Context
    H, W, nf, S, B = 96, 128, 256, 4, 12
    blob = WT.pack_blob(WT.make_synthetic(1234, 6.0))
    fr = synth.frames(2 * S * B, H, W, seed=77)
    L = capi.lib()
    ref_ctx = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=B, flags=capi.FLAG_SERIAL_BRANCH); ref_ctx.load_weights(blob)
    ref = []
    for i in range(0, len(fr), B):
        ref += ref_ctx.extract_batch(fr[i:i + B], (0, 64))
    ref_ctx.close()
    ctxs = [_ctx(nf, H, W, B=B) for _ in range(S)]
    for c in ctxs:
        c.load_weights(blob)
    d_in = capi.DeviceBuffer(fr.nbytes).upload(fr)
    d_rec = capi.DeviceBuffer(len(fr) * ctxs[0].rec_bytes)
    rb = ctxs[0].rec_bytes
    for rnd in range(2):
        for k, c in enumerate(ctxs):
            off = (rnd * S + k) * B
            capi.check(L.xfh_extract_batch_device(c.h, d_in.ptr + off * H * W, B, H, W, 0, 64, d_rec.ptr + off * rb), c.h)
    for c in ctxs:
        c.synchronize()
    recs = ctxs[0].parse_records(d_rec.download(np.uint8, len(fr) * rb), len(fr))
    for c in ctxs:
        c.close()
    for i, (a, b) in enumerate(zip(recs, ref)):
        for x, y in zip(a, b):
            assert np.array_equal(x, y), i


def test_stage_tensors_match_oracle(gpu_lib, oracle_mod, weights_std):
    """every intermediate of the forward pass; convolutions and statistics are the same fp32/fp64
    expression on both sides (one fma chain in (ky,kx,ci) order) -> expected bit exact"""
    _, blob = weights_std
    img = synth.image(160, 224, 9)
    orc = oracle_mod.Oracle(blob); orc.extract(img, 512, (0, 0))
    ctx = _ctx(512, 160, 224, B=40); ctx.load_weights(blob)
    T, OT = capi.T, oracle_mod.T
    # block1.0's map is never written (block1.1 recomputes what it consumes, k_block1_stats makes the statistics pass): its
    # statistics and the map of block1.1 cover it
    raw = lambda i, fr=0: ctx.debug_tensor(T["RAW0"] + i, fr) if i else np.zeros(0, np.float32)
    with pytest.raises(Exception):
        ctx.debug_tensor(T["RAW0"])
    # three regimes must agree bit for bit: B = 40 (statistics finalised by k_bn_finalize, persistent short-K kernels, 32x32x2 tiles
    # for every layer), B = 12 (the same, but k_conv_mfma16 = 16x16x4 tiles for the 3x3 layers with >= 64 input channels) and
    # B = 1 (k_conv_mfma16, everything folded by the consumers)
    snaps = []
    for nb in (40, 12):
        ctx.extract_batch(np.stack([img] * nb))
        snaps.append({i: (raw(i, nb - 1), ctx.debug_tensor(T["STAT0"] + i, nb - 1)) for i in range(23)})
    ctx.extract_batch(img[None])
    for big in snaps:
        for i in range(23):
            assert np.array_equal(big[i][0], raw(i)) and np.array_equal(big[i][1], ctx.debug_tensor(T["STAT0"] + i)), f"regimes differ at layer {i}"
    # x1 + skip1(x), the fusion input and the normalised features are never materialised on the GPU (they are computed while the
    # consuming kernels stage their inputs); raw maps 4 (block2.0) and 16 (block_fusion.0) and the descriptors cover them
    for nm in ["X", "XSTAT", "SKIP_POOL", "FEATS"]:
        assert np.array_equal(ctx.debug_tensor(T[nm]), orc.tensor(OT[nm])), nm
    for i in range(23):
        if i:
            a, b = raw(i), orc.tensor(OT["RAW0"] + i)
            assert a.shape == b.shape and np.abs(a - b).max() <= 1e-5, f"raw {i}"
        assert np.abs(ctx.debug_tensor(T["STAT0"] + i) - orc.tensor(OT["STAT0"] + i)).max() <= 1e-5, f"stat {i}"
    # round 5: both sides evaluate exp() as libtorch's vector kernels do (xfh_expf / xfo_expf, every step one IEEE fp32 operation): the
    # sigmoid and softmax maps are the same bits (rounds 1-4: device expf against glibc expf, <= 1 ulp apart)
    assert np.array_equal(ctx.debug_tensor(T["H1"]), orc.tensor(OT["H1"]))
    assert np.array_equal(ctx.debug_tensor(T["K1H"]), orc.tensor(OT["K1H"]))
    hs, os_ = ctx.debug_tensor(T["SEL"]).reshape(-1, 3), orc.tensor(OT["SEL"]).reshape(-1, 3)
    assert set(map(tuple, hs[:, :2].astype(int))) == set(map(tuple, os_[:, :2].astype(int)))
    ctx.close()


@pytest.mark.parametrize("nb", [5, 12, 40])
def test_batch_is_per_frame(gpu_lib, oracle_mod, weights_dense, nb):
    """frames in one batch are normalised with their OWN statistics (the reference is always B=1):
    a batched call equals single-frame calls bit for bit, and is deterministic run to run.  The batch regimes use
    different kernels / tilings (B <= 8: statistics folded by the consuming convolution; B > 8: k_bn_finalize and
    persistent short-K kernels; B <= 32: 16x16x4 MFMA tiles for the wide 3x3 layers, above: 32x32x2) and must agree bit
    for bit."""
    _, blob = weights_dense
    fr = synth.frames(nb, 96, 160, seed=11)
    fr[3] = 200                                              # a constant frame in the middle of the batch
    ctx = _ctx(200, 96, 160, B=nb); ctx.load_weights(blob)
    batch = ctx.extract_batch(fr, (0, 50))
    again = ctx.extract_batch(fr, (0, 50))
    for b in range(nb):
        single, = ctx.extract_batch(fr[b:b + 1], (0, 50))
        for x, y, z in zip(batch[b], single, again[b]):
            assert np.array_equal(x, y) and np.array_equal(x, z)
        if b < 6:
            ok, od, onv, omono = oracle_mod.Oracle(blob).extract(fr[b], 200, (0, 50))
            assert (batch[b][2], batch[b][3]) == (onv, omono) and kp_set(batch[b][0]) == kp_set(ok)
    assert batch[3][2] == 0 and np.all(batch[3][1] == 0)     # constant frame: no keypoints, all padding
    ctx.close()


def test_status_codes_and_api_surface(gpu_lib, weights_std):
    L = gpu_lib
    _, blob = weights_std
    from xfeatslam_amd.extractor import XFextractor
    ex = XFextractor.__new__(XFextractor)
    ctx = _ctx(100, 64, 96)
    img = synth.image(64, 96, 1)
    kps = np.zeros(100, capi.KP_DTYPE); desc = np.zeros((100, 64), np.float32)
    nv, mono = C.c_int(), C.c_int()
    args = (img.ctypes.data, 64, 96, 96, 0, 0, kps.ctypes.data, desc.ctypes.data, C.byref(nv), C.byref(mono))
    assert L.xfh_extract(ctx.h, *args) == 4                                   # no weights yet
    assert L.xfh_load_weights(ctx.h, b"garbage" * 10, 70) == 5
    ctx.load_weights(blob)
    assert L.xfh_extract(ctx.h, None, 64, 96, 96, 0, 0, kps.ctypes.data, desc.ctypes.data, C.byref(nv), C.byref(mono)) == 2   # empty image
    assert L.xfh_extract(ctx.h, img.ctypes.data, 16, 16, 16, 0, 0, kps.ctypes.data, desc.ctypes.data, C.byref(nv), C.byref(mono)) == 3
    assert L.xfh_extract(ctx.h, img.ctypes.data, 640, 640, 640, 0, 0, kps.ctypes.data, desc.ctypes.data, C.byref(nv), C.byref(mono)) == 3
    assert L.xfh_extract(ctx.h, *args) == 0 and L.xfh_detect_and_compute(ctx.h, *args) == 0
    # strided input (cv::Mat with padding) gives the same result as the dense one
    k0, d0 = kps.copy(), desc.copy()
    padded = np.zeros((64, 128), np.uint8); padded[:, :96] = img
    assert L.xfh_extract(ctx.h, padded.ctypes.data, 64, 96, 128, 0, 0, kps.ctypes.data, desc.ctypes.data, C.byref(nv), C.byref(mono)) == 0
    assert np.array_equal(k0, kps) and np.array_equal(d0, desc)
    ctx.close()
    # python mirror of the reference class: -1 on empty image, nfeatures rows, getters
    ex = XFextractor(100, 1.2, 8, 20, 7, weights=blob, max_height=64, max_width=96)
    assert ex(np.zeros((0, 0), np.uint8))[0] == -1
    ret, k, d = ex(img, None, (0, 0))
    assert ret == nv.value and len(k) == 100 and d.shape == (100, 64) and ex.GetLevels() == 8
    assert abs(ex.GetScaleFactors()[7] - 1.2 ** 7) < 1e-4 and len(ex.mvImagePyramid) == 8
    ret, k, d = ex(np.full((64, 96), 7, np.uint8))
    assert ret == 0 and d is None                                              # _descriptors.release()


def test_extract_to_match_device_resident(gpu_lib, oracle_mod, weights_dense):
    """records stay in HBM between xfh_extract_batch_device and xfh_match_mnn_device"""
    L = gpu_lib
    _, blob = weights_dense
    nf = 512
    fr = np.stack([synth.image(128, 160, 3), np.roll(synth.image(128, 160, 3), 2, axis=1)])
    ctx = _ctx(nf, 128, 160, B=2); ctx.load_weights(blob)
    din = capi.DeviceBuffer(fr.nbytes).upload(fr)
    rec = capi.DeviceBuffer(ctx.rec_bytes * 2)
    capi.check(L.xfh_extract_batch_device(ctx.h, din.ptr, 2, 128, 160, 0, 0, rec.ptr), ctx.h)
    out = capi.DeviceBuffer(12 * nf + 64)
    d1p, d2p = rec.ptr + ctx.desc_off, rec.ptr + ctx.rec_bytes + ctx.desc_off
    capi.check(L.xfh_match_mnn_device(ctx.h, d1p, nf, d2p, nf, -1.0, out.ptr, out.ptr + 4 * nf, out.ptr + 8 * nf, out.ptr + 12 * nf), ctx.h)
    ctx.synchronize()
    k = int(out.download(np.int32, 1, 12 * nf)[0])
    i1, i2 = out.download(np.int32, k), out.download(np.int32, k, 4 * nf)
    recs = ctx.parse_records(rec.download(np.uint8, ctx.rec_bytes * 2), 2)
    a = oracle_mod.match_mnn(recs[0][1], recs[1][1])
    assert np.array_equal(a[0], i1) and np.array_equal(a[1], i2)
    # a frame against itself: every valid keypoint is its own mutual nearest neighbour at distance ~0
    capi.check(L.xfh_match_mnn_device(ctx.h, d1p, nf, d1p, nf, -1.0, out.ptr, out.ptr + 4 * nf, out.ptr + 8 * nf, out.ptr + 12 * nf), ctx.h)
    ctx.synchronize()
    k = int(out.download(np.int32, 1, 12 * nf)[0])
    i1, i2, dd = out.download(np.int32, k), out.download(np.int32, k, 4 * nf), out.download(np.float32, k, 8 * nf)
    valid = np.where(recs[0][0]["size"] > 0)[0]
    got = dict(zip(i1.tolist(), i2.tolist()))
    assert all(got.get(int(v)) == int(v) for v in valid)
    sel = np.isin(i1, valid)
    assert np.all(np.nan_to_num(dd[sel], nan=0.0) < 1e-3)
    ctx.close()


def test_running_stats_mode(gpu_lib, oracle_mod):
    """XFH_BN_RUNNING_STATS (upstream-XFeat eval() semantics) against the oracle in the same mode"""
    from xfeatslam_amd.extractor import Context
    w = WT.make_synthetic(1234, 6.0, with_bn=True)
    blob = WT.pack_blob(w)
    fr = synth.frames(2, 160, 224, seed=5)
    ctx = Context(nfeatures=512, max_height=160, max_width=224, max_batch=2, bn_mode=1)
    ctx.load_weights(blob)
    recs = ctx.extract_batch(fr, (0, 100))
    orc = oracle_mod.Oracle(blob, bn_mode=1)
    for b in range(2):
        ok, od, onv, omono = orc.extract(fr[b], 512, (0, 100))
        hk, hd, hnv, hmono, _ = recs[b]
        assert (hnv, hmono) == (onv, omono) and kp_set(hk) == kp_set(ok)
        dd, ds, n = joined_desc_diff(hk, hd, ok, od)
        assert n == onv and dd < DESC_TOL
    # the two modes really differ, and a blob without running statistics is refused in this mode
    ok0, _, _, _ = oracle_mod.Oracle(blob, bn_mode=0).extract(fr[0], 512, (0, 100))
    assert kp_set(ok0) != kp_set(recs[0][0])
    assert gpu_lib.xfh_load_weights(ctx.h, WT.pack_blob(WT.make_synthetic(1234, 6.0)), len(WT.pack_blob(WT.make_synthetic(1234, 6.0)))) == 5
    ctx.close()


def test_running_folded_mode(gpu_lib, oracle_mod):
    """XFH_BN_RUNNING_FOLDED: BatchNorm folded into the convolutions at load time.  Folding re-rounds the
    weights, so the comparison with the oracle's eval() mode is a tolerance one: descriptors of the common keypoints within
    1e-4, and the keypoint sets equal up to threshold/NMS near-ties (<= 1 % symmetric difference)."""
    from xfeatslam_amd.extractor import Context
    w = WT.make_synthetic(1234, 6.0, with_bn=True)
    blob = WT.pack_blob(w)
    for B, H, W in ((2, 160, 224), (12, 96, 128), (1, 480, 640)):      # batches <= 8 (riders, single-frame tiles, three-stage head chain), persistent-kernel regime, one VGA frame
        fr = synth.frames(B, H, W, seed=5)
        ctx = Context(nfeatures=512, max_height=H, max_width=W, max_batch=B, bn_mode=2)
        ctx.load_weights(blob)
        recs = ctx.extract_batch(fr, (0, 100))
        if B <= 8:
            # the same frames as the head of a 12-frame batch: other kernels and tilings (two streams instead of riders, k_conv_mfma_p / k_chain1x1<2>
            # instead of the single-frame forms), the same accumulation order per output -- identical records
            big = Context(nfeatures=512, max_height=H, max_width=W, max_batch=12, bn_mode=2)
            big.load_weights(blob)
            recs12 = big.extract_batch(np.concatenate([fr, synth.frames(12 - B, H, W, seed=77)]), (0, 100))
            big.close()
            for b in range(B):
                for f in recs[b][0].dtype.names:
                    assert np.array_equal(recs[b][0][f], recs12[b][0][f]), (B, b, f)
                assert np.array_equal(recs[b][1], recs12[b][1]) and recs[b][2:4] == recs12[b][2:4], (B, b)
        ctx1 = Context(nfeatures=512, max_height=H, max_width=W, max_batch=B, bn_mode=1)
        ctx1.load_weights(blob)
        recs1 = ctx1.extract_batch(fr, (0, 100))
        ctx1.close(); ctx.close()
        orc = oracle_mod.Oracle(blob, bn_mode=1)
        for b in range(B):
            hk, hd, hnv, hmono, _ = recs[b]
            if b < 2:
                ok, od, onv, omono = orc.extract(fr[b], 512, (0, 100))
            else:                                                       # the exact (unfolded) HIP mode stands in for the oracle, itself checked above
                ok, od, onv, omono, _ = recs1[b]
            a, o = kp_set(hk), kp_set(ok)
            assert len(a ^ o) <= max(2, 0.01 * len(o)), (B, b, len(a ^ o), len(o))
            dd, ds, n = joined_desc_diff(hk, hd, ok, od)
            assert n >= 0.99 * onv - 2 and dd < 1e-4, (B, b, dd, n, onv)


def test_keypoint_rescale_flag(gpu_lib, oracle_mod):
    """XFH_FLAG_RESCALE_KEYPOINTS (upstream-XFeat coordinates) against the oracle in the same mode; the default stays the
    reference's no-op rescale.  The lapping-area split follows the reported x."""
    from xfeatslam_amd.extractor import Context
    blob = WT.pack_blob(WT.make_synthetic(1234, 3.0))
    img = synth.image(170, 230, 3)
    for flags in (0, capi.FLAG_RESCALE_KEYPOINTS):
        ok, od, onv, omono = oracle_mod.Oracle(blob, rescale=bool(flags)).extract(img, 300, (100, 150))
        ctx = Context(nfeatures=300, max_height=170, max_width=230, flags=flags); ctx.load_weights(blob)
        (hk, hd, hnv, hmono, _), = ctx.extract_batch(img[None], (100, 150))
        ctx.close()
        assert (hnv, hmono) == (onv, omono)
        for f in ("x", "y", "size", "angle", "octave", "class_id"):
            assert np.array_equal(hk[f], ok[f]), f                                # identical slots and coordinates
        assert np.abs(hk["response"] - ok["response"]).max() < 1e-6 and np.abs(hd - od).max() < DESC_TOL
        if flags:
            v = hk["size"] > 0
            assert hk["y"][v].max() > 160 - 8 and np.any(hk["x"][v] != np.round(hk["x"][v]))


def test_submit_collect_stereo_pair(gpu_lib, weights_dense):
    """xfh_extract_submit / xfh_extract_collect: two ctx (the reference's mpXFextractorLeft / Right, Tracking.cc:597-600)
    with both frames in flight give the same records as two blocking xfh_extract calls."""
    import time
    from xfeatslam_amd.capi import KP_DTYPE
    L = capi.lib()
    _, blob = weights_dense
    nf, H, W = 1000, 480, 640
    imgs = [synth.image(H, W, 5), synth.image(H, W, 6)]
    ctxs = []
    for _ in range(2):
        c = _ctx(nf, H, W); c.load_weights(blob); ctxs.append(c)
    def blocking(c, im, lap):
        k = np.zeros(nf, KP_DTYPE); d = np.zeros((nf, 64), np.float32); nv, mono = C.c_int(), C.c_int()
        capi.check(L.xfh_extract(c.h, im.ctypes.data, H, W, W, lap[0], lap[1], k.ctypes.data, d.ctypes.data, C.byref(nv), C.byref(mono)), c.h)
        return k, d, nv.value, mono.value
    laps = [(0, 0), (0, 1000)]                                # Frame.cc:311 / :495
    ref = [blocking(c, im, lap) for c, im, lap in zip(ctxs, imgs, laps)]
    t0 = time.perf_counter(); [blocking(c, im, lap) for c, im, lap in zip(ctxs, imgs, laps)]; t_seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    for c, im, lap in zip(ctxs, imgs, laps):
        capi.check(L.xfh_extract_submit(c.h, im.ctypes.data, H, W, W, lap[0], lap[1]), c.h)
    out = []
    for c in ctxs:
        k = np.zeros(nf, KP_DTYPE); d = np.zeros((nf, 64), np.float32); nv, mono = C.c_int(), C.c_int()
        capi.check(L.xfh_extract_collect(c.h, k.ctypes.data, d.ctypes.data, C.byref(nv), C.byref(mono)), c.h)
        out.append((k, d, nv.value, mono.value))
    t_par = time.perf_counter() - t0
    for a, b in zip(ref, out):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    assert L.xfh_extract_collect(ctxs[0].h, out[0][0].ctypes.data, out[0][1].ctypes.data, None, None) == 1   # nothing pending
    print(f"stereo pair: sequential {t_seq * 1e3:.2f} ms, both in flight {t_par * 1e3:.2f} ms")
    # one ctx, XFH_MAX_INFLIGHT = 2 frames in flight (frame t+1 uploaded and computed while frame t is collected), in order
    c = ctxs[0]
    seq = [imgs[0], imgs[1], imgs[0][::-1].copy(), imgs[1]]
    want = [blocking(c, im, laps[0]) for im in seq]
    capi.check(L.xfh_extract_submit(c.h, seq[0].ctypes.data, H, W, W, 0, 0), c.h)
    got = []
    for t in range(len(seq)):
        if t + 1 < len(seq):
            capi.check(L.xfh_extract_submit(c.h, seq[t + 1].ctypes.data, H, W, W, 0, 0), c.h)
            if t == 0:
                assert L.xfh_extract_submit(c.h, seq[0].ctypes.data, H, W, W, 0, 0) == 1      # ring full: collect first
        k = np.full(nf, 7, KP_DTYPE); d = np.full((nf, 64), 7, np.float32); nv, mono = C.c_int(), C.c_int()
        capi.check(L.xfh_extract_collect(c.h, k.ctypes.data, d.ctypes.data, C.byref(nv), C.byref(mono)), c.h)
        got.append((k, d, nv.value, mono.value))
    for a, b in zip(want, got):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    for c in ctxs:
        c.close()


def test_random_sizes_all_batch_regimes(gpu_lib, oracle_mod):
    """seeded fuzz over image sizes that are not multiples of the tile sizes, feature counts, lapping areas and batch sizes
    (B = 1: single-frame tiles, B <= 8: consumer-side statistics, B > 8: k_bn_finalize + persistent short-K kernels):
    every frame of every batch must reproduce the oracle's keypoints, slots and descriptors."""
    from xfeatslam_amd.extractor import Context
    rng = np.random.RandomState(2024)
    blob = WT.pack_blob(WT.make_synthetic(1234, 4.0))
    orc = oracle_mod.Oracle(blob)
    for trial in range(6):
        H, W = int(rng.randint(64, 330)), int(rng.randint(64, 420))
        nf = int(rng.choice([64, 300, 1000, 5000]))
        B = int(rng.choice([1, 3, 9, 11]))
        x0 = int(rng.randint(0, W)); lap = (x0, int(x0 + rng.randint(0, W)))
        fr = synth.frames(B, H, W, seed=100 + trial)
        ctx = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=B); ctx.load_weights(blob)
        recs = ctx.extract_batch(fr, lap)
        ctx.close()
        for b in range(B):
            ok, od, onv, omono = orc.extract(fr[b], nf, lap)
            hk, hd, hnv, hmono, _ = recs[b]
            assert (hnv, hmono) == (onv, omono), (trial, H, W, nf, B, b)
            # slot ORDER among keypoints whose scores differ by one ulp may swap (device expf vs glibc expf):
            # compare the sets, the front / back padding layout and the descriptors joined by position
            assert kp_set(hk) == kp_set(ok), (trial, H, W, nf, B, b)
            assert np.array_equal(hk["size"] == 0, ok["size"] == 0)
            dd, ds, n = joined_desc_diff(hk, hd, ok, od)
            assert n == onv and dd < DESC_TOL and ds < 1e-6


@pytest.mark.parametrize("nf", [512, 300])
def test_extraction_emits_prepared_match_images(gpu_lib, oracle_mod, nf):
    """xfh_extract_batch_device_images: k_desc also writes each frame's descriptors as the matcher's prepared image.  The image
    is bit for bit what xfh_match_prepare_device makes of the record's descriptor block (sparse frames: padding slots are rows of
    zeros; nf = 300: the rows up to the panel boundary too), the record is unchanged, and the frame-to-frame match on the
    images gives the oracle's pairs."""
    from xfeatslam_amd.extractor import Context
    L = capi.lib()
    H, W, B = 96, 128, 3
    blob = WT.pack_blob(WT.make_synthetic(1234, 2.0))                   # low gain: fewer candidates than slots -> padding rows
    fr = synth.frames(B, H, W, seed=31)
    fr[2] = np.roll(fr[1], 3, axis=1)                                   # a shifted copy: many mutual matches
    ctx = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=B); ctx.load_weights(blob)
    rb, ib = ctx.rec_bytes, int(L.xfh_match_image_bytes(nf))
    d_in = capi.DeviceBuffer(fr.nbytes).upload(fr)
    d_rec, d_rec2 = capi.DeviceBuffer(B * rb), capi.DeviceBuffer(B * rb)
    d_img = capi.DeviceBuffer(B * ib)
    capi.check(L.xfh_extract_batch_device_images(ctx.h, d_in.ptr, B, H, W, 0, 64, d_rec.ptr, d_img.ptr), ctx.h)
    capi.check(L.xfh_extract_batch_device(ctx.h, d_in.ptr, B, H, W, 0, 64, d_rec2.ptr), ctx.h)
    ctx.synchronize()
    raw = d_rec.download(np.uint8, B * rb)
    assert np.array_equal(raw, d_rec2.download(np.uint8, B * rb))
    recs = ctx.parse_records(raw, B)
    assert any(r[2] < nf for r in recs)                                 # padding slots exist
    imgs = d_img.download(np.uint8, B * ib).reshape(B, ib)
    d_one = capi.DeviceBuffer(ib)
    for b in range(B):
        capi.check(L.xfh_match_prepare_device(ctx.h, d_rec.ptr + b * rb + ctx.desc_off, nf, d_one.ptr), ctx.h)
        ctx.synchronize()
        assert np.array_equal(imgs[b], d_one.download(np.uint8, ib)), b
    class Img:                                                          # match_mnn_prepared takes (buffer-with-ptr, n)
        def __init__(self, ptr): self.ptr = ptr
    i1, i2, dist = ctx.match_mnn_prepared((Img(d_img.ptr + 1 * ib), nf), (Img(d_img.ptr + 2 * ib), nf))
    a = oracle_mod.match_mnn(recs[1][1], recs[2][1])
    assert np.array_equal(a[0], i1) and np.array_equal(a[1], i2) and len(i1) > 10
    # the C-side timing loop runs the same call
    us = C.c_double(0)
    out = capi.DeviceBuffer(12 * nf + 64)
    assert L.xfh_bench_match_prepared(ctx.h, d_img.ptr + ib, nf, d_img.ptr + 2 * ib, nf, -1.0, out.ptr, out.ptr + 4 * nf, out.ptr + 8 * nf, out.ptr + 12 * nf, 5, C.byref(us)) == 0
    assert us.value > 0
    ctx.close()


def test_match_records_drops_padding_pairs(gpu_lib, oracle_mod):
    """xfh_match_records_device (n_valid-aware option): the reference's match() lets the zero rows that pad a record
    take part; this form reports the same pairs minus those that touch a padding slot.  Frames with few keypoints, one lapping
    area that puts some of them into the back segment."""
    from xfeatslam_amd.extractor import Context
    L = capi.lib()
    H, W, nf = 96, 128, 512
    blob = WT.pack_blob(WT.make_synthetic(1234, 2.0))
    fr = synth.frames(2, H, W, seed=31)
    fr[1] = np.roll(fr[0], 2, axis=1)
    ctx = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=2); ctx.load_weights(blob)
    rb, ib = ctx.rec_bytes, int(L.xfh_match_image_bytes(nf))
    d_in = capi.DeviceBuffer(fr.nbytes).upload(fr)
    d_rec, d_img = capi.DeviceBuffer(2 * rb), capi.DeviceBuffer(2 * ib)
    capi.check(L.xfh_extract_batch_device_images(ctx.h, d_in.ptr, 2, H, W, 40, 90, d_rec.ptr, d_img.ptr), ctx.h)
    out = capi.DeviceBuffer(12 * nf + 64)
    capi.check(L.xfh_match_records_device(ctx.h, d_rec.ptr, d_img.ptr, d_rec.ptr + rb, d_img.ptr + ib, -1.0, out.ptr, out.ptr + 4 * nf, out.ptr + 8 * nf, out.ptr + 12 * nf), ctx.h)
    ctx.synchronize()
    n = int(out.download(np.int32, 1, 12 * nf)[0])
    got = set(zip(out.download(np.int32, n).tolist(), out.download(np.int32, n, 4 * nf).tolist()))
    recs = ctx.parse_records(d_rec.download(np.uint8, 2 * rb), 2)
    (k1, d1, nv1, mo1, _), (k2, d2, nv2, mo2, _) = recs
    assert 0 < nv1 < nf and 0 < mo1 < nv1                                 # padding exists, and a back segment
    a = oracle_mod.match_mnn(d1, d2)                                      # the reference's semantics: padding rows take part
    valid1 = lambda i: i < mo1 or i >= nf - (nv1 - mo1)
    valid2 = lambda j: j < mo2 or j >= nf - (nv2 - mo2)
    want = {(int(i), int(j)) for i, j in zip(a[0], a[1]) if valid1(i) and valid2(j)}
    assert got == want and len(want) > 10
    assert all(k1["size"][i] > 0 and k2["size"][j] > 0 for i, j in got)
    ctx.close()


def test_reload_weights_while_batches_in_flight(gpu_lib):
    """xfh_load_weights replaces the packed buffers, and the pipeline lanes are host threads that hold copies of the pointers.
    A reload issued while multi-sub-batch submits are outstanding must first wait for the lanes: the outstanding submits complete with the OLD weights
    (records equal to a blocking call before the reload), later calls use the NEW weights, nothing reads freed memory.  Repeated, alternating two sets."""
    L = gpu_lib
    H, W, nf, S = 96, 128, 256, 4
    n = 6 * S
    fr = synth.frames(n, H, W, seed=321)
    blobs = [WT.pack_blob(WT.make_synthetic(1234, 6.0)), WT.pack_blob(WT.make_family("heavy", 9))]
    ctx = _ctx(nf, H, W, B=S)
    rb = ctx.rec_bytes
    hin = capi.HostBuffer(fr.nbytes); hin.array[:] = fr.reshape(-1)
    want = []
    for b in blobs:
        ctx.load_weights(b)
        o = np.zeros(n * rb, np.uint8)
        capi.check(L.xfh_extract_batch(ctx.h, hin.ptr, n, H, W, 0, 64, o.ctypes.data), ctx.h)
        want.append(o)
    houts = [capi.HostBuffer(n * rb) for _ in range(2)]
    cur = 1
    for it in range(12):
        for h in houts:
            h.array[:] = 0
            capi.check(L.xfh_extract_batch_submit(ctx.h, hin.ptr, n, H, W, 0, 64, h.ptr), ctx.h)
        nxt = 1 - cur
        ctx.load_weights(blobs[nxt])                         # while two submits (12 sub-batches) are queued on the lanes
        capi.check(L.xfh_extract_batch_drain(ctx.h), ctx.h)
        for h in houts:
            assert records_equal(ctx, h.array, want[cur], n), it        # the outstanding work saw the old weights, complete and intact
        cur = nxt
        o = np.zeros(n * rb, np.uint8)
        capi.check(L.xfh_extract_batch(ctx.h, hin.ptr, n, H, W, 0, 64, o.ctypes.data), ctx.h)
        assert records_equal(ctx, o, want[cur], n), it
    for h in houts + [hin]:
        h.free()
    ctx.close()


def test_host_visible_batch_pipeline(gpu_lib, oracle_mod, weights_dense):
    """xfh_extract_batch / _submit / _wait (host frames in, host records out): a call of B frames with B far above
    cfg.max_batch is cut into sub-batches that go into one queue drained by the lanes (own activations and streams, shared weights, a worker
    thread each that drives copy in / kernels / copy out).  Pinned and pageable caller buffers, blocking and asynchronous form, ragged last
    sub-batch, several submits outstanding: every record equals the one a lone serial ctx produces for the frame, bit for bit;
    frame 0 is checked against the oracle."""
    from xfeatslam_amd.extractor import Context
    L = gpu_lib
    _, blob = weights_dense
    H, W, nf, S = 96, 128, 256, 6
    n = 5 * S + 2                                            # 6 sub-batches on 4 lanes (two lanes run twice: both buffer generations), ragged tail
    fr = synth.frames(n, H, W, seed=123)
    fr[7] = 0                                                # a frame without keypoints
    ref_ctx = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=n, flags=capi.FLAG_SERIAL_BRANCH); ref_ctx.load_weights(blob)
    rb = ref_ctx.rec_bytes
    d_in = capi.DeviceBuffer(fr.nbytes).upload(fr); d_rec = capi.DeviceBuffer(n * rb)
    capi.check(L.xfh_extract_batch_device(ref_ctx.h, d_in.ptr, n, H, W, 0, 64, d_rec.ptr), ref_ctx.h)
    ref_ctx.synchronize()
    want = d_rec.download(np.uint8, n * rb)
    ref_ctx.close()
    ctx = _ctx(nf, H, W, B=S); ctx.load_weights(blob)
    same = lambda got: records_equal(ctx, got, want, n)
    # pageable caller memory, blocking call
    out = np.zeros(n * rb, np.uint8)
    capi.check(L.xfh_extract_batch(ctx.h, fr.ctypes.data, n, H, W, 0, 64, out.ctypes.data), ctx.h)
    assert same(out)
    # pinned caller memory (xfh_host_alloc), three submits outstanding, one wait; different lane counts
    hin = capi.HostBuffer(fr.nbytes); hin.array[:] = fr.reshape(-1)
    houts = [capi.HostBuffer(n * rb) for _ in range(3)]
    for lanes in (4, 2, 1, 8):
        assert L.xfh_pipeline_lanes(ctx.h, lanes) == 0
        for h in houts:
            h.array[:] = 0
            capi.check(L.xfh_extract_batch_submit(ctx.h, hin.ptr, n, H, W, 0, 64, h.ptr), ctx.h)
        capi.check(L.xfh_extract_batch_wait(ctx.h), ctx.h)       # the oldest submit
        assert same(houts[0].array), lanes
        capi.check(L.xfh_extract_batch_drain(ctx.h), ctx.h)
        assert L.xfh_extract_batch_wait(ctx.h) == 1                # nothing outstanding any more
        for h in houts:
            assert same(h.array), lanes
    assert L.xfh_pipeline_lanes(ctx.h, 0) == 1 and L.xfh_pipeline_lanes(ctx.h, 9) == 1
    # a registered caller buffer
    reg = np.zeros(n * rb + 4096, np.uint8)
    assert L.xfh_host_register(reg.ctypes.data, reg.nbytes) == 0
    capi.check(L.xfh_extract_batch(ctx.h, hin.ptr, n, H, W, 0, 64, reg.ctypes.data), ctx.h)
    assert L.xfh_host_unregister(reg.ctypes.data) == 0
    assert same(reg[:n * rb])
    # the single-frame ring and the batch path share the ctx' first frame buffer: refused while a submission is outstanding
    img = np.ascontiguousarray(fr[0])
    capi.check(L.xfh_extract_submit(ctx.h, img.ctypes.data, H, W, W, 0, 64), ctx.h)
    k = np.zeros(nf, capi.KP_DTYPE); d = np.zeros((nf, 64), np.float32); nv, mono = C.c_int(), C.c_int()
    assert L.xfh_extract_batch(ctx.h, hin.ptr, n, H, W, 0, 64, houts[0].ptr) == 1
    assert L.xfh_extract(ctx.h, img.ctypes.data, H, W, W, 0, 64, k.ctypes.data, d.ctypes.data, C.byref(nv), C.byref(mono)) == 1
    capi.check(L.xfh_extract_collect(ctx.h, k.ctypes.data, d.ctypes.data, C.byref(nv), C.byref(mono)), ctx.h)
    r0 = ctx.parse_records(want, 1)[0]
    assert (nv.value, mono.value) == (r0[2], r0[3]) and np.array_equal(k, r0[0]) and np.array_equal(d, r0[1])
    # reloading the weights reaches the lanes as well
    ctx.load_weights(WT.pack_blob(WT.make_synthetic(77, 6.0)))
    capi.check(L.xfh_extract_batch(ctx.h, hin.ptr, n, H, W, 0, 64, houts[0].ptr), ctx.h)
    a = houts[0].array.copy()
    assert not same(a)
    one = _ctx(nf, H, W, B=n); one.load_weights(WT.pack_blob(WT.make_synthetic(77, 6.0)))
    capi.check(L.xfh_extract_batch(one.h, hin.ptr, n, H, W, 0, 64, houts[1].ptr), one.h)
    assert records_equal(ctx, a, houts[1].array, n)
    one.close()
    ok, od, onv, omono = oracle_mod.Oracle(blob).extract(fr[0], nf, (0, 64))
    assert (r0[2], r0[3]) == (onv, omono) and kp_set(r0[0]) == kp_set(ok)
    assert L.xfh_extract_batch_wait(None) == 1 and L.xfh_host_alloc(None, 16) == 1
    for h in houts + [hin]:
        h.free()
    ctx.close()


def test_batch_pipeline_queue_limits_and_teardown(gpu_lib, weights_dense):
    """the ring of outstanding submits (XFH_MAX_BATCHES_INFLIGHT = 8: the ninth is refused until one is waited for), submits of one sub-batch (they
    run on the ctx itself) between submits of many (worker lanes), in order; and xfh_destroy with submits still outstanding: the lanes finish
    what is queued, then the ctx goes away -- no hang, no write after the call returns"""
    from xfeatslam_amd.extractor import Context
    L = gpu_lib
    _, blob = weights_dense
    H, W, nf, S = 96, 128, 256, 4
    n = 3 * S + 1
    fr = synth.frames(n, H, W, seed=321)
    ref = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=n); ref.load_weights(blob)
    rb = ref.rec_bytes
    d_in = capi.DeviceBuffer(fr.nbytes).upload(fr); d_rec = capi.DeviceBuffer(n * rb)
    capi.check(L.xfh_extract_batch_device(ref.h, d_in.ptr, n, H, W, 0, 0, d_rec.ptr), ref.h)
    ref.synchronize()
    want = d_rec.download(np.uint8, n * rb)
    ref.close()
    ctx = _ctx(nf, H, W, B=S); ctx.load_weights(blob)
    hin = capi.HostBuffer(fr.nbytes); hin.array[:] = fr.reshape(-1)
    houts = [capi.HostBuffer(n * rb) for _ in range(8)]
    sizes = [n, S, n, 2, n, S - 1, n, n]                    # many sub-batches (lanes) and single ones (the ctx itself), alternating
    for h, b in zip(houts, sizes):
        capi.check(L.xfh_extract_batch_submit(ctx.h, hin.ptr, b, H, W, 0, 0, h.ptr), ctx.h)
    assert L.xfh_extract_batch_submit(ctx.h, hin.ptr, n, H, W, 0, 0, houts[0].ptr) == 1       # the ring is full
    for h, b in zip(houts, sizes):
        capi.check(L.xfh_extract_batch_wait(ctx.h), ctx.h)                                    # oldest first
        assert records_equal(ctx, h.array[:b * rb], want[:b * rb], b), b
    assert L.xfh_extract_batch_wait(ctx.h) == 1
    # teardown with work in flight
    for h in houts[:4]:
        h.array[:] = 0
        capi.check(L.xfh_extract_batch_submit(ctx.h, hin.ptr, n, H, W, 0, 0, h.ptr), ctx.h)
    ctx.close()                                              # xfh_destroy: returns when the lanes are done
    for h in houts[:4]:
        assert records_equal(_RecView(nf, rb), h.array, want, n)      # (the ctx is gone: a layout-only view parses the records)
    for h in houts + [hin]:
        h.free()
    d_in.free(); d_rec.free()


class _RecView:
    """parse_records without a live ctx (the layout only depends on nfeatures)"""
    def __init__(self, nf, rb):
        self.nfeatures, self.rec_bytes = nf, rb
        self.kps_off = int(capi.lib().xfh_record_kps_offset()); self.desc_off = int(capi.lib().xfh_record_desc_offset(nf))

    def parse_records(self, raw, B):
        from xfeatslam_amd.extractor import Context
        return Context.parse_records(self, raw, B)


def test_eval_bn_modes_through_the_pipeline_lanes(gpu_lib, weights_dense):
    """the lanes of a ctx in an eval()-BatchNorm mode carry the parent's statistics slots (ctx_share_weights)"""
    L = gpu_lib
    from xfeatslam_amd.extractor import Context
    blob = WT.pack_blob(WT.make_synthetic(1234, 6.0, with_bn=True))
    H, W, nf, S, n = 64, 96, 128, 3, 11
    fr = synth.frames(n, H, W, seed=5)
    for mode in (1, 2):
        one = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=n, bn_mode=mode); one.load_weights(blob)
        want = one.extract_batch(fr, (0, 0)); one.close()
        ctx = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=S, bn_mode=mode); ctx.load_weights(blob)
        got = ctx.extract_batch(fr, (0, 0)); ctx.close()
        for i, (a, b) in enumerate(zip(got, want)):
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (mode, i)


def test_exact_score_tie_straddles_the_cut(gpu_lib, oracle_mod, weights_dense):
    """two NMS candidates with bit-equal scores on either side of rank nfeatures (tests/test_oracle.py::test_exact_score_tie_at_the_cut
    holds the frames): the 64-bit selection key is (~ordered(score), y * W + x), so the candidate with the lower pixel index wins the
    cut -- the reference's stable order.  The tie is located in the DEVICE's own score list; since round 5 device and oracle evaluate the
    same exp(), so the oracle ties at the same pair in every frame and the two selected sets must be identical."""
    _, blob = weights_dense
    H, W = 256, 320
    orc = oracle_mod.Oracle(blob)
    checked = same_pair = 0
    for seed in (16, 19, 27, 50):
        img = synth.image(H, W, seed)
        full = _ctx(8192, H, W); full.load_weights(blob)
        full.extract_batch(img[None])
        sel = full.debug_tensor(capi.T["SEL"]).reshape(-1, 3)            # (x, y, score) in selection order, all candidates
        full.close()
        s = sel[:, 2]
        ties = np.where((s[1:] == s[:-1]) & (s[1:] > 0))[0]
        assert np.all(np.diff(s) <= 0)
        for r in ties[:2]:
            a, b = sel[r], sel[r + 1]
            assert a[1] * W + a[0] < b[1] * W + b[0]                     # ties are ordered by pixel index
            ctx = _ctx(int(r) + 1, H, W); ctx.load_weights(blob)
            (kps, desc, nv, mono, nc), = ctx.extract_batch(img[None])
            ctx.close()
            got = kp_set(kps)
            assert nv == r + 1 and (int(a[0]), int(a[1])) in got and (int(b[0]), int(b[1])) not in got
            assert got == {(int(x), int(y)) for x, y, _ in sel[:r + 1]}
            checked += 1
            ok, od, onv, omono = orc.extract(img, int(r) + 1, (0, 0))
            ocand = orc.tensor(oracle_mod.T["CAND"]).reshape(-1, 3)
            osc = {(int(x), int(y)): v for x, y, v in ocand}
            if osc[(int(a[0]), int(a[1]))] == osc[(int(b[0]), int(b[1]))]:            # the oracle ties at the same pair
                assert kp_set(ok) == got
                same_pair += 1
    assert checked >= 4 and same_pair == checked


def test_border_and_origin_candidates(gpu_lib, oracle_mod, weights_dense):
    """NMS candidates at (0,0) (masked to -1, XFextractor.cc:283) and on the last row / column (the nearest
    sample with align_corners=false reads outside the map: score 0) are never reported -- frame found by search (64x96, seed 53)"""
    _, blob = weights_dense
    H, W = 64, 96
    img = synth.image(H, W, 53)
    orc = oracle_mod.Oracle(blob)
    ok, od, onv, omono = orc.extract(img, 512, (0, 40))
    cand = orc.tensor(oracle_mod.T["CAND"]).reshape(-1, 3)
    xs, ys = cand[:, 0].astype(int), cand[:, 1].astype(int)
    assert np.any((xs == 0) & (ys == 0)) and np.any(xs == W - 1) and np.any(ys == H - 1)
    ctx = _ctx(512, H, W); ctx.load_weights(blob)
    (kps, desc, nv, mono, nc), = ctx.extract_batch(img[None], (0, 40))
    assert nc == len(cand) and (nv, mono) == (onv, omono) and kp_set(kps) == kp_set(ok)
    got = kp_set(kps)
    assert (0, 0) not in got and not any(x == W - 1 or y == H - 1 for x, y in got)
    dd, ds, n = joined_desc_diff(kps, desc, ok, od)
    assert n == onv and dd < DESC_TOL and ds < 1e-6
    ctx.close()
