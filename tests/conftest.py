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
