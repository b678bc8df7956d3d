"""Soak of the hand-rolled synchronisation on the GPU (round 4).  The kernels of this library synchronise by hand in many places: LDS
flags and level loops in k_select, riders that share a launch with a backbone layer, counted s_waitcnt / raw s_barrier pipelines in the
match GEMMs, agent-scope publish / poll collectors in k_mnn_post, phase-skewed wave groups in k_mnn_gemm_seg.  A race there shows as ONE
wrong record in hundreds of launches, on some boxes only (round 3: a missing barrier in k_select, one wrong selection in ~40 runs).  These
tests repeat every batch regime (B = 1, <= 8, <= 32, > 32: different kernels and tilings) and the match paths some thousand times (about 20 s
on an MI355X), with a
second ctx keeping the GPU busy in between, and require every result to be bit-identical to the first one -- which is checked against the
oracle once."""
import ctypes as C

import numpy as np
import pytest

from xfeatslam_amd import capi, synth, weights as WT

pytestmark = pytest.mark.gpu


def _busy_ctx(lib, blob, frames):
    """a second ctx whose extractions are queued (never waited for) between the iterations of the soaked one"""
    from xfeatslam_amd.extractor import Context
    c = Context(nfeatures=512, max_height=frames.shape[1], max_width=frames.shape[2], max_batch=len(frames))
    c.load_weights(blob)
    din = capi.DeviceBuffer(frames.nbytes).upload(frames); rec = capi.DeviceBuffer(len(frames) * c.rec_bytes)

    def kick():
        capi.check(lib.xfh_extract_batch_device(c.h, din.ptr, len(frames), frames.shape[1], frames.shape[2], 0, 0, rec.ptr), c.h)
    return c, kick, (din, rec)


@pytest.mark.parametrize("H,W,B,iters", [(96, 160, 1, 2000), (96, 160, 8, 2000), (96, 160, 12, 1500), (96, 160, 40, 1000), (96, 160, 64, 1000),
                                         (480, 640, 1, 1500), (480, 640, 8, 400), (480, 640, 12, 200)])
def test_extraction_soak(gpu_lib, oracle_mod, H, W, B, iters):
    from xfeatslam_amd.extractor import Context
    lib = capi.lib()
    nf = 512 if H < 480 else 4096
    blob = WT.pack_blob(WT.make_synthetic(1234, 6.0))
    # two different frame sets, alternating from call to call (round 6): every intermediate buffer of the ctx -- statistic partials, folded statistics, raw maps,
    # candidate lists -- then holds ANOTHER call's values when a call starts, so a consumer that read a left-over instead of its producer's output would show
    fsets = [synth.frames(B, H, W, seed=21), synth.frames(B, H, W, seed=22)[::-1].copy()]
    ctx = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=B)
    busy, kick, bufs = _busy_ctx(lib, blob, synth.frames(4, 96, 160, seed=5))
    try:
        ctx.load_weights(blob)
        dins = [capi.DeviceBuffer(f.nbytes).upload(f) for f in fsets]
        rec = capi.DeviceBuffer(B * ctx.rec_bytes).upload(np.zeros(B * ctx.rec_bytes, np.uint8))
        first = [None, None]
        for it in range(iters):
            if it % 2:
                kick()                                   # the other ctx' kernels share the CUs with this iteration
            k2 = (it // 3 + it) % 2                      # which frame set: AB BA AB ... (both orders of succession occur)
            frames = fsets[k2]
            capi.check(lib.xfh_extract_batch_device(ctx.h, dins[k2].ptr, B, H, W, 0, 0, rec.ptr), ctx.h)
            ctx.synchronize()
            raw = rec.download(np.uint8, B * ctx.rec_bytes)
            if first[k2] is None:
                first[k2] = raw
                # the first result of each set is the oracle's result (frame 0 and the last frame)
                recs = ctx.parse_records(raw, B)
                orc = oracle_mod.Oracle(blob)
                for b in sorted({0, B - 1}):
                    ok, od, onv, omono = orc.extract(frames[b], nf, (0, 0))
                    kps, desc, nv, mono, _ = recs[b]
                    v1, v2 = kps["size"] > 0, ok["size"] > 0
                    assert (nv, mono) == (onv, omono)
                    assert set(zip(kps["x"][v1].astype(int), kps["y"][v1].astype(int))) == set(zip(ok["x"][v2].astype(int), ok["y"][v2].astype(int)))
            else:
                assert np.array_equal(raw, first[k2]), f"iteration {it} (set {k2}): records differ from the set's first result (first byte {int(np.argmax(raw != first[k2]))})"
        assert not np.array_equal(first[0], first[1])
        busy.synchronize()
    finally:
        ctx.close(); busy.close()


@pytest.mark.parametrize("H,W,B,iters,mode", [(96, 160, 2, 800, 2), (480, 640, 1, 500, 2), (96, 160, 8, 600, 2), (96, 160, 2, 600, 1), (96, 160, 40, 300, 2)])
def test_extraction_soak_eval_modes(gpu_lib, H, W, B, iters, mode):
    """the eval()-BatchNorm modes have kernels of their own in the small-batch regime (round 4: riders with the bias + ReLU epilogue, the three-stage
    1x1 chain of the heads, the two-stage chain that folds its producer's partials): every record of every iteration bit-identical to the first
    (which tests/test_gpu_extract.py::test_running_stats_mode / test_running_folded_mode compare with the oracle)"""
    from xfeatslam_amd.extractor import Context
    lib = capi.lib()
    nf = 512 if H < 480 else 4096
    blob = WT.pack_blob(WT.make_synthetic(1234, 6.0, with_bn=True))
    frames = synth.frames(B, H, W, seed=23)
    ctx = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=B, bn_mode=mode)
    busy, kick, bufs = _busy_ctx(lib, WT.pack_blob(WT.make_synthetic(1234, 6.0)), synth.frames(4, 96, 160, seed=5))
    try:
        ctx.load_weights(blob)
        din = capi.DeviceBuffer(frames.nbytes).upload(frames)
        rec = capi.DeviceBuffer(B * ctx.rec_bytes).upload(np.zeros(B * ctx.rec_bytes, np.uint8))
        first = None
        for it in range(iters):
            if it % 2:
                kick()
            capi.check(lib.xfh_extract_batch_device(ctx.h, din.ptr, B, H, W, 0, 0, rec.ptr), ctx.h)
            ctx.synchronize()
            raw = rec.download(np.uint8, B * ctx.rec_bytes)
            if first is None:
                first = raw
                assert all(r[2] > 0 for r in ctx.parse_records(raw, B))          # keypoints were found
            else:
                assert np.array_equal(raw, first), f"iteration {it}: records differ from iteration 0 (first byte {int(np.argmax(raw != first))})"
        busy.synchronize()
    finally:
        ctx.close(); busy.close()


@pytest.mark.parametrize("n1,n2,iters", [(4096, 4096, 6000), (1000, 777, 4000), (257, 4097, 4000)])
def test_prepared_match_soak(gpu_lib, oracle_mod, n1, n2, iters):
    """back-to-back prepared-image matches (k_mnn_gemm_img + k_mnn_post: LDS-DMA pipeline, plane keys, collectors that poll the pairs the writers
    publish and that the next call's GEMM re-arms): the list of every call equals the oracle's.  Round 6: consecutive calls ALTERNATE between two different
    descriptor pairs (and the two roles of one of them), so a key plane or a (column, value) pair left over from the previous call -- the scratch is the same
    memory every call, and an agent-scope load is not by itself proof against a stale line in another XCD's L2 (tools/probes/mnn_tail_probe.hip) -- would give
    a wrong list instead of silently the right one."""
    from xfeatslam_amd.extractor import Context
    lib = capi.lib()
    sets = []
    for seed, noise, swap in ((9, 0.3, False), (10, 0.45, False), (9, 0.3, True)):
        d1, d2 = synth.descriptor_sets(n1, n2, zero_rows=3, noise=noise, seed=seed)
        d2[5] = d2[2]
        if swap and n1 == n2:
            d1, d2 = d2, d1
        elif swap:
            d1 = d1[::-1].copy()
        sets.append((d1, d2, oracle_mod.match_mnn(d1, d2)))
    assert not np.array_equal(sets[0][2][1], sets[1][2][1])
    ctx = Context(nfeatures=64, max_height=32, max_width=32)
    busy, kick, bufs = _busy_ctx(lib, WT.pack_blob(WT.make_synthetic(1234, 6.0)), synth.frames(4, 96, 160, seed=5))
    try:
        prep = [(ctx.match_prepare(d1), ctx.match_prepare(d2)) for d1, d2, _ in sets]
        nm = min(n1, n2)
        out = capi.DeviceBuffer(12 * nm + 64)
        for it in range(iters):
            if it % 4 == 1:
                kick()
            k3 = (it * 7) % 3 if it % 5 else it % 3                 # which pair this call matches: changes from call to call, in no fixed rhythm with the reads below
            p1, p2 = prep[k3]
            capi.check(lib.xfh_match_mnn_prepared_device(ctx.h, p1[0].ptr, n1, p2[0].ptr, n2, -1.0, out.ptr + 64, out.ptr + 64 + 4 * nm, out.ptr + 64 + 8 * nm, out.ptr), ctx.h)
            if it % 3 == 0:                              # two of three calls run back to back on the stream, the third is read
                want = sets[k3][2]
                ctx.synchronize()
                k = int(out.download(np.int32, 1)[0])
                assert k == len(want[0]), (it, k3, k)
                assert np.array_equal(out.download(np.int32, k, 64), want[0]) and np.array_equal(out.download(np.int32, k, 64 + 4 * nm), want[1]), (it, k3)
                assert np.array_equal(out.download(np.float32, k, 64 + 8 * nm), want[2], equal_nan=True), (it, k3)
        busy.synchronize()
        for a, b in prep:
            a[0].free(); b[0].free()
        out.free()
    finally:
        ctx.close(); busy.close()


def test_batched_match_soak(gpu_lib, oracle_mod):
    """the many-pairs call 900 times (phase-skewed wave groups, double-buffered LDS-DMA panels, running row keys, k_mnn_post_batch's per-pair
    collectors), mixed shapes, a busy second ctx: every pair list of every call equals the oracle's"""
    from xfeatslam_amd.extractor import Context
    lib = capi.lib()
    shapes = [(4096, 4096), (4096, 4096), (1000, 777), (257, 4097), (4096, 2500), (129, 127), (4096, 4096), (300, 200)]
    ctx = Context(nfeatures=64, max_height=32, max_width=32)
    busy, kick, bufs = _busy_ctx(lib, WT.pack_blob(WT.make_synthetic(1234, 6.0)), synth.frames(4, 96, 160, seed=5))
    try:
        data, prepared, want = [], [], []
        for k, (n1, n2) in enumerate(shapes):
            d1, d2 = synth.descriptor_sets(n1, n2, zero_rows=(4 if k % 2 else 0), noise=0.3, seed=40 + k)
            if n2 > 40:
                d2[7] = d2[1]
            data.append((d1, d2)); prepared.append((ctx.match_prepare(d1), ctx.match_prepare(d2))); want.append(oracle_mod.match_mnn(d1, d2))
        P = len(shapes)
        NM = 4096                                          # every output slot holds the largest pair: the pairs ROTATE through the slots from call to call (round 6),
        off = np.arange(P + 1, dtype=np.int64) * (12 * NM + 64)      # so every piece of the call's scratch meets other data than in the call before (stale planes / pairs would show)
        out = capi.DeviceBuffer(int(off[-1])); cnt = capi.DeviceBuffer(4 * P + 64)
        i1 = (C.c_void_p * P)(*[out.ptr + int(off[p]) for p in range(P)]); i2 = (C.c_void_p * P)(*[out.ptr + int(off[p]) + 4 * NM for p in range(P)])
        ds = (C.c_void_p * P)(*[out.ptr + int(off[p]) + 8 * NM for p in range(P)])
        tables = [ctx.pair_tables(prepared[r:] + prepared[:r]) for r in range(P)]
        for it in range(900):
            if it % 4 == 1:
                kick()
            r = (it * 3) % P if it % 2 else (it // 2) % P
            p1, n1, p2, n2 = tables[r]
            capi.check(lib.xfh_match_mnn_prepared_batch_device(ctx.h, P, p1, n1, p2, n2, -1.0, i1, i2, ds, cnt.ptr), ctx.h)
            if it % 3 == 0:
                ctx.synchronize()
                ks = cnt.download(np.int32, P)
                for p in range(P):
                    w = want[(p + r) % P]
                    k = int(ks[p])
                    assert k == len(w[0]), (it, r, p, k)
                    assert np.array_equal(out.download(np.int32, k, int(off[p])), w[0]) and np.array_equal(out.download(np.int32, k, int(off[p]) + 4 * NM), w[1]), (it, r, p)
        busy.synchronize()
        for a, b in prepared:
            a[0].free(); b[0].free()
        out.free(); cnt.free()
    finally:
        ctx.close(); busy.close()
