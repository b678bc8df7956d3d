import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """the CPU oracle (test infrastructure only), built on demand"""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def weights_std():
    from xfeatslam_amd import weights as WT
    w = WT.make_synthetic(1234, 1.0)
    return w, WT.pack_blob(w)


@pytest.fixture(scope="session")
def weights_dense():
    from xfeatslam_amd import weights as WT
    w = WT.make_synthetic(1234, 6.0)
    return w, WT.pack_blob(w)


EXTRACT_GOLDENS = ["extract_96x128", "extract_vga", "extract_odd_170x230", "extract_720p", "extract_vga_dense_mono",
                   # round 5, the parity campaign: weight family x image family (tests/golden/make_golden.py)
                   "extract_vga_pruned_blobs", "extract_vga_heavy_saturated", "extract_720p_denormal_lowcontrast", "extract_vga_dc_steps"]


# round 6: ATen fixtures for every weight family at VGA (x 2 image families) and 720p, the TUM1.yaml nfeatures = 1000 cases, and
# ORBmatcher::match on extracted descriptor blocks -- names from the generator's own tables (tests/golden/make_golden.py)
def _golden_tables():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


_MG = _golden_tables()
CAMPAIGN_GOLDENS = list(_MG.CAMPAIGN_NAMES)
MATCH_X_GOLDENS = list(_MG.MATCH_X_NAMES)
EXTRACT_GOLDENS = EXTRACT_GOLDENS + CAMPAIGN_GOLDENS


def match_x_inputs(g):
    """the two descriptor blocks a match_x_* fixture carries (int16 / 2^14, exact in fp32)"""
    q = np.float32(_MG.DESC_Q)
    return np.ascontiguousarray(g["q1"].astype(np.float32) / q), np.ascontiguousarray(g["q2"].astype(np.float32) / q)


def check_extract_golden(g, kps, desc, nv, mono, nc, desc_tol=1e-4):
    """One extraction output (the C oracle's on CPU, the HIP path's on the GPU) against an ATen fixture: candidate count, valid count,
    IDENTICAL keypoint set, scores and sampled descriptors joined by position.  Where the fixture's top-k cut fell inside a group of
    scores closer than 2e-6 (`cut_gap`, stored by make_golden.py), libtorch's own order inside that group is summation noise (SURVEY.md
    Q10) and the two sets may differ ONLY in keypoints whose score is within 2e-6 of the cut score; the count is returned and printed.
    An exact tie in the fixture (cut_gap == 0: periodic frames, a saturated heatmap) counts as such a group too: the other side's values
    for the same cells differ in the last bit, so its order inside the group is its own (the exact-tie order of ONE implementation is
    pinned elsewhere: tests/test_gpu_select.py against the oracle)."""
    fam = str(g["family"]) if "family" in g.files else ""
    assert (nv, nc) == (int(g["n_valid"]), int(g["n_candidates"])), ((nv, nc), (int(g["n_valid"]), int(g["n_candidates"])))
    have, want = kp_set(kps), set(map(tuple, g["xy"].tolist()))
    pos = {(int(k["x"]), int(k["y"])): i for i, k in enumerate(kps) if k["size"] > 0}
    gsc = {tuple(p): float(s) for p, s in zip(g["xy"].tolist(), g["score"])}
    swapped = have ^ want
    if swapped:
        cut, gap = (float(g["cut_score"]), float(g["cut_gap"])) if "cut_gap" in g.files else (float("nan"), float("nan"))
        assert 0.0 <= gap < 2e-6, f"keypoint sets differ ({len(swapped)}) away from a near-tie at the cut (gap {gap})"
        for xy in swapped:
            sc = gsc[xy] if xy in gsc else float(kps["response"][pos[xy]])
            assert abs(sc - cut) < 2e-6, (xy, sc, cut)
        print(f"  [{len(swapped) // 2} keypoints swapped inside the near-tie at the cut, gap {gap:.1e}]", flush=True)
    else:
        assert mono == int(g["mono_index"])
    common = [tuple(p) for p in g["xy"].tolist() if tuple(p) in pos]
    if common:
        idx = np.array([pos[p] for p in common])
        ref = np.array([gsc[p] for p in common], np.float32)
        assert np.abs(kps["response"][idx] - ref).max() < (2e-4 if fam == "peaky" else 5e-5 if fam else 1e-5)
    rows = [(tuple(p), r) for p, r in zip(g["desc_rows_xy"].tolist(), g["desc_rows"]) if tuple(p) in pos]
    if rows:
        ridx = np.array([pos[p] for p, _ in rows])
        assert np.abs(desc[ridx] - np.stack([r for _, r in rows])).max() < desc_tol
    return len(swapped) // 2


def golden_inputs(g):
    """(weights dict, image) a golden extraction case was generated from: the round-1 fixtures carry (gain, seed) of make_synthetic(1234) /
    synth.image, the round-5 ones also a weight family, its seed and an image family"""
    from xfeatslam_amd import synth, weights as WT
    fam = str(g["family"]) if "family" in g.files else ""
    if fam:
        w = WT.make_family(fam, int(g["wseed"]), float(g["gain"]))
        img = synth.image_family(str(g["image_family"]), int(g["H"]), int(g["W"]), int(g["seed"]))
    else:
        w = WT.make_synthetic(1234, float(g["gain"]))
        img = synth.image(int(g["H"]), int(g["W"]), int(g["seed"]))
    return w, img


def kp_set(kps):
    v = kps["size"] > 0
    return set(zip(kps["x"][v].astype(int).tolist(), kps["y"][v].astype(int).tolist()))


def joined_desc_diff(k1, d1, k2, d2):
    """max |desc| / |score| difference over keypoints present in both outputs (position join,
    SURVEY.md Q10: keypoint order among near-tied scores is not defined by the reference)"""
    a = {(int(k["x"]), int(k["y"])): i for i, k in enumerate(k1) if k["size"] > 0}
    b = {(int(k["x"]), int(k["y"])): i for i, k in enumerate(k2) if k["size"] > 0}
    common = [k for k in a if k in b]
    if not common:
        return 0.0, 0.0, 0
    ia = np.array([a[k] for k in common]); ib = np.array([b[k] for k in common])
    return (float(np.abs(d1[ia] - d2[ib]).max()), float(np.abs(k1["response"][ia] - k2["response"][ib]).max()), len(common))


@pytest.fixture(scope="session")
def gpu_lib():
    """libxfeat_hip.so on a box with a GPU; fails loudly (no fallback) if it is not usable"""
    from xfeatslam_amd import capi
    L = capi.lib()
    assert L.xfh_device_count() > 0, "no HIP device visible: gpu tests need an MI355X"
    return L


def records_equal(ctx, raw_a, raw_b, n):
    """two byte blobs of n records hold the same records: header fields, keypoints and descriptors (the alignment gaps between
    the sections of a record are never written by the kernels, so whole-blob comparison would compare uninitialised bytes)"""
    a = ctx.parse_records(np.ascontiguousarray(raw_a).reshape(-1), n); b = ctx.parse_records(np.ascontiguousarray(raw_b).reshape(-1), n)
    ok = True
    for i, (x, y) in enumerate(zip(a, b)):
        if x[2:] != y[2:] or not np.array_equal(x[0], y[0]) or not np.array_equal(x[1], y[1]):
            nk = int((x[0] != y[0]).sum()) if x[0].shape == y[0].shape else -1
            nd = int((x[1] != y[1]).any(axis=1).sum()) if x[1].shape == y[1].shape else -1
            print(f"records_equal: record {i} of {n} differs: header {x[2:]} vs {y[2:]}, {nk} keypoint rows, {nd} descriptor rows", flush=True)
            ok = False
    return ok
