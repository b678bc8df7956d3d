"""GPU parity tests of the matching half through the C ABI (xfh_match_mnn*, xfh_distance_i32*)
against the oracle and the committed goldens.  Pair lists must be identical; distances are the
same fp32 expression on both sides (bit exact)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, MATCH_X_GOLDENS, match_x_inputs
from xfeatslam_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mctx(gpu_lib):
    from xfeatslam_amd.extractor import Context
    c = Context(nfeatures=64, max_height=32, max_width=32)
    yield c
    c.close()


@pytest.mark.parametrize("n1,n2,zero,noise", [(256, 256, 0, 0.3), (300, 200, 7, 0.3), (1, 5, 0, 0.3), (5, 1, 0, 0.3),
                                               (129, 127, 0, 0.5), (128, 128, 0, 0.3), (1000, 4096, 0, 0.4),
                                               (4096, 4096, 0, 0.3), (4096, 4096, 100, 0.3),
                                               (4097, 300, 0, 0.3), (5000, 4500, 9, 0.3)])          # 17 / 20 d1 panels: the column-key planes no longer divide by the four lanes of a candidate
def test_mnn_matches_oracle(mctx, oracle_mod, n1, n2, zero, noise):
    d1, d2 = synth.descriptor_sets(n1, n2, zero_rows=zero, noise=noise)
    a = oracle_mod.match_mnn(d1, d2)
    b = mctx.match_mnn(d1, d2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2], equal_nan=True)            # same fp32 expression: bit exact
    assert np.all(np.diff(b[0]) > 0)                             # ascending queryIdx, one match per row


@pytest.mark.parametrize("n1,n2,zero", [(256, 256, 0), (300, 200, 7), (4096, 4096, 100), (1000, 4097, 0), (257, 255, 3)])
def test_mnn_prepared_images_match_oracle(mctx, oracle_mod, n1, n2, zero):
    """xfh_match_prepare_device + xfh_match_mnn_prepared_device (two launches) = xfh_match_mnn on the same rows; prepared and raw
    calls may alternate on one ctx (the arg-max keys are re-zeroed by whichever style ran last)."""
    d1, d2 = synth.descriptor_sets(n1, n2, zero_rows=zero, noise=0.3)
    a = oracle_mod.match_mnn(d1, d2)
    p1, p2 = mctx.match_prepare(d1), mctx.match_prepare(d2)
    for rep in range(3):
        b = mctx.match_mnn_prepared(p1, p2)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2], equal_nan=True)
        if rep == 1:
            c = mctx.match_mnn(d2, d1)           # a raw call in between leaves other keys behind
            r = oracle_mod.match_mnn(d2, d1)
            assert np.array_equal(c[0], r[0]) and np.array_equal(c[1], r[1])
    for thr in (0.5, 0.95):
        a2 = oracle_mod.match_mnn(d1, d2, thr); b2 = mctx.match_mnn_prepared(p1, p2, thr)
        assert np.array_equal(a2[0], b2[0]) and np.array_equal(a2[1], b2[1])
    p1[0].free(); p2[0].free()


@pytest.mark.parametrize("name", ["match_256", "match_300x200_zero7", "match_4096", "match_4096_zero100"])
def test_mnn_matches_golden(mctx, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    d1, d2 = synth.descriptor_sets(int(g["n1"]), int(g["n2"]), zero_rows=int(g["zero_rows"]), noise=float(g["noise"]))
    i1, i2, dist = mctx.match_mnn(d1, d2)
    assert np.array_equal(i1, g["idx1"]) and np.array_equal(i2, g["idx2"])      # identical match pairs
    assert np.allclose(dist, g["dist"], atol=2e-6, equal_nan=True)
    assert np.array_equal(mctx.distance_i32(d1[:48], d2[:40]), g["dist_i32_corner"])


@pytest.mark.parametrize("name", MATCH_X_GOLDENS)
def test_mnn_matches_golden_extracted(mctx, name):
    """round 6: ORBmatcher::match on descriptor blocks that came out of the extractor -- zero padding rows where a frame has fewer than
    nfeatures keypoints, the lapping split's row order, exact duplicate rows, a ragged 4096 x 1000 pair -- the fixture carries the
    blocks (int16 / 2^14) and the lists libtorch's matmul / max give on them (tests/golden/make_golden.py: match_extracted_case)"""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    d1, d2 = match_x_inputs(g)
    i1, i2, dist = mctx.match_mnn(d1, d2)
    assert np.array_equal(i1, g["idx1"]) and np.array_equal(i2, g["idx2"])      # identical match pairs
    assert np.allclose(dist, g["dist"], atol=2e-6, equal_nan=True)
    p1, p2 = mctx.match_prepare(d1), mctx.match_prepare(d2)                     # the device-resident form the tracker would use
    j1, j2, _ = mctx.match_mnn_prepared(p1, p2)
    p1[0].free(); p2[0].free()
    assert np.array_equal(j1, g["idx1"]) and np.array_equal(j2, g["idx2"])
    assert np.array_equal(mctx.distance_i32(d1[:48], d2[:40]), g["dist_i32_corner"])
    assert np.array_equal(mctx.distance_i32(d1[-24:], d2[-24:]), g["dist_i32_tail"])


def test_mnn_batched_call_matches_every_golden(mctx):
    """round 6: the many-pairs path (k_mnn_gemm_seg + k_mnn_post_batch, xfh_match_mnn_prepared_batch_device) meets libtorch's lists first-hand: all nine match
    goldens -- Gaussian rows and extracted descriptor blocks, square, ragged, with zero and duplicate rows -- as the pairs of ONE call (and each pair twice, so
    that a workgroup walks tiles of different pairs)."""
    names = ["match_256", "match_300x200_zero7", "match_4096", "match_4096_zero100"] + list(MATCH_X_GOLDENS)
    gs, preps = [], []
    for name in names:
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        if name in MATCH_X_GOLDENS:
            d1, d2 = match_x_inputs(g)
        else:
            d1, d2 = synth.descriptor_sets(int(g["n1"]), int(g["n2"]), zero_rows=int(g["zero_rows"]), noise=float(g["noise"]))
        gs.append(g); preps.append((mctx.match_prepare(d1), mctx.match_prepare(d2)))
    res = mctx.match_mnn_prepared_batch(preps + preps[::-1])
    assert len(res) == 2 * len(names)
    for k, (name, g) in enumerate(zip(names, gs)):
        for r in (res[k], res[2 * len(names) - 1 - k]):
            assert np.array_equal(r[0], g["idx1"]) and np.array_equal(r[1], g["idx2"]), name
            assert np.allclose(r[2], g["dist"], atol=2e-6, equal_nan=True), name
    for a, b in preps:
        a[0].free(); b[0].free()


def test_mnn_edge_cases(mctx, oracle_mod):
    d1, d2 = synth.descriptor_sets(200, 180, noise=0.2)
    # duplicate rows => exact ties; first maximum (lowest index) wins on both axes
    d2[10] = d2[3]; d2[77] = d2[3]; d1[50] = d1[20]
    a = oracle_mod.match_mnn(d1, d2); b = mctx.match_mnn(d1, d2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert 10 not in b[1].tolist() and 77 not in b[1].tolist() and 50 not in b[0].tolist()
    # zero-padded rows take part: (0,0) is reported with dist sqrt(2)
    z1 = d1.copy(); z2 = d2.copy(); z1[0] = 0; z2[0] = 0
    z1[1:] = np.abs(z1[1:]); z2[1:] = -np.abs(z2[1:])
    a = oracle_mod.match_mnn(z1, z2); b = mctx.match_mnn(z1, z2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and (0, 0) in list(zip(b[0].tolist(), b[1].tolist()))
    # min_cossim gate
    for thr in (0.5, 0.9, 0.99):
        a = oracle_mod.match_mnn(d1, d2, thr); b = mctx.match_mnn(d1, d2, thr)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # empty sides, unnormalised inputs (normalisation is part of match())
    assert len(mctx.match_mnn(d1[:0], d2)[0]) == 0 and len(mctx.match_mnn(d1, d2[:0])[0]) == 0
    s = (np.arange(200, dtype=np.float32) + 1)[:, None]
    a = oracle_mod.match_mnn(d1 * s, d2 * 3.0); b = mctx.match_mnn(d1 * s, d2 * 3.0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2], equal_nan=True)
    # permutation property: matching against a permuted copy of itself recovers the permutation
    perm = np.random.RandomState(0).permutation(200)
    i1, i2, dist = mctx.match_mnn(d1[:200], d1[:200][perm])
    inv = np.argsort(perm)
    ok = [k for k in range(200) if k not in (20, 50)]            # rows 20 and 50 are duplicates
    got = dict(zip(i1.tolist(), i2.tolist()))
    assert all(got.get(k) == inv[k] for k in ok)


@pytest.mark.parametrize("n1,n2", [(700, 900), (4096, 4096), (130, 260)])
def test_mnn_ties_across_candidate_groups(mctx, oracle_mod, n1, n2):
    """The kernel keeps value maxima per group of 16 d2 rows / 16 d1 rows and names the member afterwards
    (k_mnn_post).  Duplicated descriptors give exact ties inside a group, across neighbouring groups,
    across 128/256-row tiles and across workgroups; the first index must win on both axes."""
    rs = np.random.RandomState(n1 + n2)
    d1, d2 = synth.descriptor_sets(n1, n2, noise=0.25)
    for src, dsts in [(3, (1, 2)), (5, (7, 9, 300)), (64, (65, 80, 127)), (17, (n2 - 1,)), (255, (256, 257, 511 % n2))]:
        for d in dsts:
            d2[d % n2] = d2[src % n2]
    for src, dsts in [(2, (0,)), (20, (21, 35)), (15, (16, 31)), (100, (n1 - 1,)), (127, (128,))]:
        for d in dsts:
            d1[d % n1] = d1[src % n1]
    k = min(n1, n2) // 3
    d2[rs.randint(0, n2, k)] = d2[rs.randint(0, n2, k)]      # many random duplicates on top
    d1[rs.randint(0, n1, k)] = d1[rs.randint(0, n1, k)]
    a = oracle_mod.match_mnn(d1, d2); b = mctx.match_mnn(d1, d2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2], equal_nan=True)
    # all rows identical: every dot product ties, (0, 0) is the only mutual pair
    e = np.tile(d1[:1], (300, 1))
    i1, i2, _ = mctx.match_mnn(e, e[:200])
    assert i1.tolist() == [0] and i2.tolist() == [0]


def test_distance_i32_exact(mctx, oracle_mod):
    """k_dist_mfma: MFMA bulk (|a|^2 + |b|^2 - 2ab) + exact recomputation wherever the value is within the error bound of an
    integer; the checker is the oracle's sequential fp32-difference / fp64-accumulate expression.  Zero rows (distance exactly
    512 against unit rows), duplicates (distance 0), unnormalised rows, and both sets scaled down to norms of 1e-4 .. 3e-2 with duplicates across them, all stress the fix-up path."""
    for n1, n2, z in [(512, 384, 0), (70, 33, 3), (1, 1, 0), (64, 65, 0), (4096, 4096, 100), (129, 1000, 5)]:
        d1, d2 = synth.descriptor_sets(n1, n2, noise=0.3, zero_rows=z)
        if n2 > 10:
            d2[3] = d1[0]; d2[7] = d2[2]
        assert np.array_equal(mctx.distance_i32(d1, d2), oracle_mod.distance_i32(d1, d2))       # integer: bit exact
    d1, d2 = synth.descriptor_sets(300, 260, noise=0.5)
    s1 = (np.arange(300, dtype=np.float32) % 7 + 0.25)[:, None]; s2 = (np.arange(260, dtype=np.float32) % 5 * 0.5 + 1e-3)[:, None]
    assert np.array_equal(mctx.distance_i32(d1 * s1, d2 * s2), oracle_mod.distance_i32(d1 * s1, d2 * s2))
    # BOTH sets tiny, with exact duplicates across them: the safety test's own fp32 roundings decide here -- a relative error bound
    # below 2^-26 let a v32 of -1e-10 between identical rows through as floor(v32) = -1 (dist_mfma.hip.h: the absolute 2^-23 terms)
    for sc in (1e-4, 1e-3, 2e-3, 1e-2, 3e-2):
        a = (d1 * np.float32(sc)).astype(np.float32); b = (d2 * np.float32(sc)).astype(np.float32)
        b[::3] = a[:260:3]; b[5] = 0; a[9] = 0
        got, want = mctx.distance_i32(a, b), oracle_mod.distance_i32(a, b)
        assert np.array_equal(got, want), (sc, int((got != want).sum()), got[got != want][:4], want[got != want][:4])
    q = (np.round(d1 * 64) / 64).astype(np.float32); r = (np.round(d2 * 64) / 64).astype(np.float32)       # many exactly representable distances
    assert np.array_equal(mctx.distance_i32(q, r), oracle_mod.distance_i32(q, r))
    d1, d2 = synth.descriptor_sets(n1, n2, noise=0.3, zero_rows=z)
    # consistency with the scalar host metric that ORBmatcher::DescriptorDistance replaces
    L = capi.lib()
    t = mctx.distance_i32(d1, d2)
    for (i, j) in [(0, 0), (5, 7), (63, 64)]:
        assert t[i, j] == L.xfh_descriptor_distance(d1[i].ctypes.data, d2[j].ctypes.data)


def test_mnn_device_resident_and_deterministic(mctx):
    L = capi.lib()
    n = 4096
    d1, d2 = synth.descriptor_sets(n, n, noise=0.3)
    b1 = capi.DeviceBuffer(d1.nbytes).upload(d1); b2 = capi.DeviceBuffer(d2.nbytes).upload(d2)
    out = capi.DeviceBuffer(12 * n + 64)
    res = []
    for _ in range(3):
        capi.check(L.xfh_match_mnn_device(mctx.h, b1.ptr, n, b2.ptr, n, -1.0, out.ptr, out.ptr + 4 * n, out.ptr + 8 * n, out.ptr + 12 * n), mctx.h)
        mctx.synchronize()
        k = int(out.download(np.int32, 1, 12 * n)[0])
        res.append((out.download(np.int32, k), out.download(np.int32, k, 4 * n), out.download(np.float32, k, 8 * n)))
    host = mctx.match_mnn(d1, d2)
    for r in res:
        assert np.array_equal(r[0], host[0]) and np.array_equal(r[1], host[1]) and np.array_equal(r[2], host[2], equal_nan=True)
    # misaligned device pointer is rejected, not read
    assert L.xfh_match_mnn_device(mctx.h, b1.ptr + 4, n - 1, b2.ptr, n, -1.0, out.ptr, out.ptr, out.ptr, out.ptr) == 1


def test_best2_csr_matches_oracle(mctx, oracle_mod):
    """guided matching primitive (SearchByProjection inner loop) -- integer work: bit exact"""
    rng = np.random.RandomState(3)
    for nq, nt, mx, noise, init in [(500, 4096, 80, 0.25, 256), (64, 100, 300, 0.2, 256), (200, 1000, 20, 0.5, 1 << 30), (1, 1, 1, 0.1, 256)]:
        q, tg = synth.descriptor_sets(nq, nt, noise=noise)
        if nt > 10:
            tg[7] = tg[3]
        counts = rng.randint(0, mx + 1, nq)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        ind = rng.randint(0, nt, off[-1]).astype(np.int32)
        a = oracle_mod.best2_csr(q, tg, off, ind, init)
        b = mctx.best2_csr(q, tg, off, ind, init)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    # invalid lists are rejected on the host, nothing is read out of bounds
    L = capi.lib()
    q, tg = synth.descriptor_sets(4, 8)
    off = np.array([0, 2, 2, 3, 4], np.int32); ind = np.array([0, 9, 1, 2], np.int32)
    o = [np.zeros(4, np.int32) for _ in range(4)]
    assert L.xfh_best2_csr(mctx.h, q.ctypes.data, 4, tg.ctypes.data, 8, off.ctypes.data, ind.ctypes.data, 256, *[x.ctypes.data for x in o]) == 1


def test_distinctive_csr_matches_oracle(mctx, oracle_mod):
    """MapPoint::ComputeDistinctiveDescriptors batched over map points -- integer work: bit exact"""
    rng = np.random.RandomState(8)
    tb, _ = synth.descriptor_sets(2000, 1, noise=0.3)
    tb[11] = tb[4]; tb[12] = tb[4]; tb[100:110] = 0            # duplicates and zero-padded rows
    counts = np.concatenate([[0, 1, 2, 3, 63, 64, 65, 128, 200, 256], rng.randint(0, 40, 500)]).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    ind = rng.randint(0, 2000, off[-1]).astype(np.int32)
    ind[off[3]:off[3] + 3] = [4, 11, 12]
    a = oracle_mod.distinctive_csr(tb, off, ind)
    b = mctx.distinctive_csr(tb, off, ind)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert b[0][0] == -1 and b[1][0] == 0x7fffffff and b[0][1] == 0 and b[1][1] == 0
    # groups above XFH_MAX_GROUP and out-of-range rows are rejected on the host
    L = capi.lib()
    off2 = np.array([0, 257], np.int32); ind2 = np.zeros(257, np.int32); o = [np.zeros(1, np.int32) for _ in range(2)]
    assert L.xfh_distinctive_csr(mctx.h, tb.ctypes.data, 2000, off2.ctypes.data, ind2.ctypes.data, 1, o[0].ctypes.data, o[1].ctypes.data) == 1
    off3 = np.array([0, 2], np.int32); ind3 = np.array([0, 2000], np.int32)
    assert L.xfh_distinctive_csr(mctx.h, tb.ctypes.data, 2000, off3.ctypes.data, ind3.ctypes.data, 1, o[0].ctypes.data, o[1].ctypes.data) == 1


def test_mnn_random_shapes(mctx, oracle_mod):
    """seeded fuzz over ragged sizes around the 64 / 128 / 256 tile edges, with duplicated and all-zero rows"""
    rng = np.random.RandomState(77)
    for trial in range(12):
        n1, n2 = int(rng.randint(1, 700)), int(rng.randint(1, 700))
        d1, d2 = synth.descriptor_sets(n1, n2, noise=float(rng.uniform(0.05, 0.6)), zero_rows=int(rng.randint(0, 4)))
        for _ in range(int(rng.randint(0, 6))):
            d2[rng.randint(0, n2)] = d2[rng.randint(0, n2)]
            d1[rng.randint(0, n1)] = d1[rng.randint(0, n1)]
        thr = float(rng.choice([-1.0, 0.5, 0.9]))
        a = oracle_mod.match_mnn(d1, d2, thr); b = mctx.match_mnn(d1, d2, thr)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (trial, n1, n2, thr)
        assert np.array_equal(a[2], b[2], equal_nan=True)


def test_match_under_four_busy_extraction_ctx(gpu_lib, oracle_mod):
    """k_mnn_post's collector blocks poll (column, value) pairs that writer blocks of the same grid publish; forward progress must
    not depend on the GPU being otherwise idle.  Four extraction ctx keep 4 x 24 frames in flight on their own streams while a
    fifth ctx runs 4096 x 4096 matches back to back (raw rows and prepared images): every call returns the oracle's pairs."""
    from xfeatslam_amd import weights as WT
    from xfeatslam_amd.extractor import Context
    L = gpu_lib
    H, W, nf, B = 256, 320, 1024, 24
    blob = WT.pack_blob(WT.make_synthetic(1234, 6.0))
    busy = [Context(nfeatures=nf, max_height=H, max_width=W, max_batch=B) for _ in range(4)]
    for c in busy:
        c.load_weights(blob)
    fr = synth.frames(B, H, W, seed=3)
    d_in = capi.DeviceBuffer(fr.nbytes).upload(fr)
    d_rec = [capi.DeviceBuffer(B * busy[0].rec_bytes) for _ in busy]
    m = Context(nfeatures=4096, max_height=32, max_width=32)
    d1, d2 = synth.descriptor_sets(4096, 4096, noise=0.3, zero_rows=50)
    want = oracle_mod.match_mnn(d1, d2)
    p1, p2 = m.match_prepare(d1), m.match_prepare(d2)
    for rnd in range(6):
        for k, c in enumerate(busy):                       # ~40 ms of queued extraction work per round, never waited for here
            for _ in range(3):
                capi.check(L.xfh_extract_batch_device(c.h, d_in.ptr, B, H, W, 0, 0, d_rec[k].ptr), c.h)
        for _ in range(4):
            a = m.match_mnn(d1, d2)
            b = m.match_mnn_prepared(p1, p2)
            for got in (a, b):
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), rnd
    for c in busy:
        c.synchronize(); c.close()
    m.close()


def test_nan_and_inf_descriptors_return(mctx):
    """NaN / Inf descriptor rows are outside the contract (include/xfeat_hip.h: results for such rows are unspecified), but the
    calls must come back: no hang in the collectors, a status, and sane pairs for the finite rows that do not touch them"""
    d1, d2 = synth.descriptor_sets(300, 260, noise=0.2)
    want = mctx.match_mnn(d1, d2)
    bad1, bad2 = d1.copy(), d2.copy()
    bad1[7] = np.nan; bad1[100, 3] = np.inf; bad2[11] = -np.inf; bad2[200, 63] = np.nan
    try:
        got = mctx.match_mnn(bad1, bad2)
    except capi.XfhError as e:                             # a refused call is fine too, a hang is not
        assert e.status in (1, 6)
        return
    n = len(got[0])
    assert 0 <= n <= 260 and np.all((got[0] >= 0) & (got[0] < 300)) and np.all((got[1] >= 0) & (got[1] < 260))
    clean = {(int(a), int(b)) for a, b in zip(want[0], want[1]) if a not in (7, 100) and b not in (11, 200)}
    have = {(int(a), int(b)) for a, b in zip(got[0], got[1])}
    assert len(clean - have) <= 8                           # pairs of finite rows survive unless a poisoned row displaced their partner
    assert mctx.match_mnn(d1, d2)[0].tolist() == want[0].tolist()       # and the ctx is still usable


# ---- many pairs in one call (xfh_match_mnn_prepared_batch_device: k_mnn_gemm_seg + k_mnn_post_batch) ---------------------------------
def test_mnn_batch_one_frame_against_partners(mctx, oracle_mod):
    """the tracker's case (ORBmatcher.cc:358-372 called once per frame pair): one frame against several partners, one call.  Every pair list is the
    oracle's; the batched call equals the pair-by-pair calls bit for bit."""
    q, _ = synth.descriptor_sets(4096, 16, noise=0.3, seed=3)
    partners = []
    for k, (n, noise, zero) in enumerate([(4096, 0.3, 0), (4096, 0.5, 100), (2500, 0.3, 3), (4096, 0.2, 0), (1000, 0.4, 0)]):
        _, d2 = synth.descriptor_sets(4096, n, zero_rows=zero, noise=noise, seed=3)        # noisy permuted copies of q's rows (same seed -> same d1)
        partners.append(d2)
    pq = mctx.match_prepare(q)
    pp = [mctx.match_prepare(d) for d in partners]
    res = mctx.match_mnn_prepared_batch([(pq, p) for p in pp])
    for d2, p, r in zip(partners, pp, res):
        a = oracle_mod.match_mnn(q, d2)
        assert np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1]) and np.array_equal(a[2], r[2], equal_nan=True)
        s = mctx.match_mnn_prepared(pq, p)
        assert np.array_equal(s[0], r[0]) and np.array_equal(s[1], r[1]) and np.array_equal(s[2], r[2], equal_nan=True)
    for thr in (0.5, 0.9):
        res = mctx.match_mnn_prepared_batch([(pq, p) for p in pp], thr)
        for d2, r in zip(partners, res):
            a = oracle_mod.match_mnn(q, d2, thr)
            assert np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1])
    pq[0].free()
    for p in pp:
        p[0].free()


@pytest.mark.parametrize("shapes", [
    [(4096, 4096)] * 8,                                                          # BENCH's batched leg: every workgroup walks 8 tiles
    [(300, 200), (1, 5), (5, 1), (129, 127), (257, 4097), (1000, 777)],          # ragged: tiles with masked rows and columns, tiny pairs
    [(256, 256)],                                                                # one tile, one workgroup
    [(4096, 4096), (1000, 777), (257, 4097), (1, 5), (129, 127), (300, 200), (2500, 4096), (4096, 2500), (512, 512),
     (700, 300), (64, 4096), (4096, 64), (1024, 1024), (333, 334), (256, 257), (255, 256), (2048, 1024), (77, 99)],   # 18 pairs: two launches of <= 16
])
def test_mnn_batch_matches_oracle(mctx, oracle_mod, shapes):
    """mixed shapes, duplicates (ties inside and across candidate groups, tiles and workgroups) and zero rows: every pair list equals the oracle's"""
    data, prepared = [], []
    for k, (n1, n2) in enumerate(shapes):
        d1, d2 = synth.descriptor_sets(n1, n2, zero_rows=(5 if min(n1, n2) > 100 and k % 2 else 0), noise=0.3, seed=100 + k)
        if n2 > 40:
            d2[min(10, n2 - 1)] = d2[3]; d2[n2 - 1] = d2[3]
        if n1 > 60:
            d1[50] = d1[20]; d1[n1 - 1] = d1[0]
        data.append((d1, d2)); prepared.append((mctx.match_prepare(d1), mctx.match_prepare(d2)))
    for rep in range(2):                                                          # twice: the key planes and pairs of the first call are left behind
        res = mctx.match_mnn_prepared_batch(prepared)
        for (d1, d2), r in zip(data, res):
            a = oracle_mod.match_mnn(d1, d2)
            assert np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1]) and np.array_equal(a[2], r[2], equal_nan=True)
            assert np.all(np.diff(r[0]) > 0)
    for p1, p2 in prepared:
        p1[0].free(); p2[0].free()


def test_mnn_batch_empty_sides_and_arguments(mctx):
    d1, d2 = synth.descriptor_sets(300, 200, noise=0.3)
    p1, p2 = mctx.match_prepare(d1), mctx.match_prepare(d2)
    assert mctx.match_mnn_prepared_batch([]) == []
    L = capi.lib()
    P = 3
    out = capi.DeviceBuffer(3 * 4096 + 64); cnt = capi.DeviceBuffer(64).upload(np.full(16, -7, np.int32))
    img1 = (C.c_void_p * P)(p1[0].ptr, None, p1[0].ptr); n1 = (C.c_int * P)(300, 0, 300)
    img2 = (C.c_void_p * P)(p2[0].ptr, p2[0].ptr, None); n2 = (C.c_int * P)(200, 200, 0)
    i1 = (C.c_void_p * P)(out.ptr, None, None); i2 = (C.c_void_p * P)(out.ptr + 1024, None, None); ds = (C.c_void_p * P)(out.ptr + 2048, None, None)
    assert L.xfh_match_mnn_prepared_batch_device(mctx.h, P, img1, n1, img2, n2, -1.0, i1, i2, ds, cnt.ptr) == 0
    mctx.synchronize()
    k = cnt.download(np.int32, 3)
    assert k[0] > 0 and k[1] == 0 and k[2] == 0                                   # an empty side: no matches (ORBmatcher.cc:358: nothing to compare)
    n1b = (C.c_int * P)(300, -1, 300)
    assert L.xfh_match_mnn_prepared_batch_device(mctx.h, P, img1, n1b, img2, n2, -1.0, i1, i2, ds, cnt.ptr) == 1
    assert L.xfh_match_mnn_prepared_batch_device(mctx.h, P, None, n1, img2, n2, -1.0, i1, i2, ds, cnt.ptr) == 1
    assert L.xfh_match_mnn_prepared_batch_device(mctx.h, 1, img1, n1, img2, n2, -1.0, i1, i2, ds, None) == 1
    out.free(); cnt.free(); p1[0].free(); p2[0].free()
