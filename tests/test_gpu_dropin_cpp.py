"""The C++ drop-in classes (include/xfeat/XFextractor.h, ORBmatcher_xfeat.h) driven the way
Frame::ExtractXF drives the reference (src/Frame.cc:611-618), compared with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, joined_desc_diff, kp_set
from xfeatslam_amd import capi, synth, weights as WT

pytestmark = pytest.mark.gpu


def test_cpp_dropin_end_to_end(gpu_lib, oracle_mod, tmp_path):
    exe = str(tmp_path / "dropin_test")
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "dropin_test.cpp"),
           "-L" + os.path.join(ROOT, "xfeatslam_amd"), "-lxfeat_hip", "-Wl,-rpath," + os.path.join(ROOT, "xfeatslam_amd"),
           "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    blob = WT.pack_blob(WT.make_synthetic(1234, 6.0))
    H, W, nf, lap = 192, 256, 600, (0, 100)
    img = synth.image(H, W, 8)
    (tmp_path / "w.xfhw").write_bytes(blob)
    (tmp_path / "img.raw").write_bytes(img.tobytes())
    r = subprocess.run([exe, str(tmp_path / "w.xfhw"), str(tmp_path / "img.raw"), str(H), str(W), str(nf), str(lap[0]), str(lap[1]),
                        str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = (tmp_path / "out.bin").read_bytes()
    ret, nkeys, drows, nm = struct.unpack_from("<4i", raw, 0)
    kps = np.frombuffer(raw, capi.KP_DTYPE, nkeys, 16)
    desc = np.frombuffer(raw, np.float32, drows * 64, 16 + 28 * nkeys).reshape(drows, 64)
    m = np.frombuffer(raw, np.dtype([("q", "<i4"), ("t", "<i4"), ("d", "<f4")]), nm, 16 + 28 * nkeys + 256 * drows)
    ok, od, onv, omono = oracle_mod.Oracle(blob).extract(img, nf, lap)
    assert ret == omono and nkeys == nf and drows == nf
    assert kp_set(kps) == kp_set(ok)
    dd, ds, n = joined_desc_diff(kps, desc, ok, od)
    assert n == onv and dd < 1e-4
    a = oracle_mod.match_mnn(desc, desc)
    assert np.array_equal(a[0], m["q"]) and np.array_equal(a[1], m["t"])


def test_frontend_replay_example(gpu_lib, tmp_path):
    """examples/frontend_replay.cpp (shaped after rgbd_tum.cc): extract + match against the previous frame"""
    exe = str(tmp_path / "frontend_replay")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "frontend_replay.cpp"),
                           "-L" + os.path.join(ROOT, "xfeatslam_amd"), "-lxfeat_hip", "-Wl,-rpath," + os.path.join(ROOT, "xfeatslam_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    (tmp_path / "w.xfhw").write_bytes(WT.pack_blob(WT.make_synthetic(1234, 6.0)))
    # PGM sequence + TUM-style association list
    os.makedirs(tmp_path / "rgb")
    lines = []
    for i in range(4):
        img = synth.image(96, 128, 20 + i)
        with open(tmp_path / "rgb" / f"{i}.pgm", "wb") as f:
            f.write(b"P5\n# frame\n128 96\n255\n" + img.tobytes())
        lines.append(f"{i}.0 rgb/{i}.pgm {i}.0 depth/{i}.png")
    (tmp_path / "assoc.txt").write_text("\n".join(lines) + "\n")
    for args in ([str(tmp_path / "assoc.txt"), str(tmp_path)], ["--synthetic", "5", "128", "160"]):
        r = subprocess.run([exe, str(tmp_path / "w.xfhw")] + args, capture_output=True, text=True, env=dict(os.environ, XFH_NFEATURES="300"))
        assert r.returncode == 0, r.stderr
        assert "median front-end time" in r.stdout and "keypoints/frame" in r.stdout
