"""CPU tests: the C-ABI library loads and exports every symbol include/xfeat_hip.h declares
(no compute without a GPU), host-side logic, C++ drop-in wrapper compiles."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from xfeatslam_amd import capi, synth, weights as WT


@pytest.fixture(scope="module", autouse=True)
def _built_library():
    """the C-ABI library is a build product (git-ignored): on a checkout where __graft_entry__.build() has not run yet, build it
    here (hipcc cross-compiles for gfx950 without a GPU) -- the product itself never falls back, it just fails to load"""
    if not os.path.exists(capi.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "xfeatslam_amd", "csrc"), "-s", "-j8"])


def test_header_symbols_exported_and_bound():
    declared = set()
    for name in ("xfeat_hip.h", "xfeat_hip_bench.h"):              # the drop-in surface, and the measurement / debugging entry points
        hdr = open(os.path.join(ROOT, "include", name)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        found = set(re.findall(r"\b(xfh_[a-z0-9_]+)\s*\(", hdr))
        assert (name == "xfeat_hip_bench.h") == any(f.startswith(("xfh_bench_", "xfh_timing_", "xfh_debug_")) for f in found), name
        declared |= found
    bound = {s[0] for s in capi.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    L = capi.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_abi_struct_layout_and_helpers():
    L = capi.lib()
    assert capi.KP_DTYPE.itemsize == 28 and C.sizeof(capi.Config) == 7 * 4 + 8 * 4
    cfg = capi.Config(); L.xfh_config_default(C.byref(cfg))
    assert (cfg.max_height, cfg.max_width, cfg.nfeatures, cfg.max_batch) == (480, 640, 4096, 1)
    assert abs(cfg.nms_threshold - 0.05) < 1e-7
    assert L.xfh_record_kps_offset() == 16
    assert L.xfh_record_desc_offset(4096) % 256 == 0 and L.xfh_record_desc_offset(4096) >= 16 + 28 * 4096
    assert L.xfh_record_bytes(4096) >= L.xfh_record_desc_offset(4096) + 4096 * 256
    assert L.xfh_strerror(2) == b"empty image" and b"gfx950" in L.xfh_version()
    for k in range(10):
        assert L.xfh_kernel_name(k)


def test_no_device_fails_loudly():
    L = capi.lib()
    if L.xfh_device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = capi.Config(); L.xfh_config_default(C.byref(cfg))
    h = C.c_void_p()
    assert L.xfh_create(C.byref(cfg), C.byref(h)) == capi.ERR_NO_DEVICE      # never a CPU fallback
    from xfeatslam_amd.extractor import Context
    with pytest.raises(capi.XfhError):
        Context()


def test_rccl_selection_is_explicit_and_reported():
    """comm.cpp: $XFH_RCCL_LIB names the librccl and nothing is tried behind it; xfh_comm_library() says which file, version and HIP runtime (no GPU
    needed: dlopen only).  Each case in its own process -- the choice is made once per process."""
    import subprocess
    import sys
    code = ("import sys; from xfeatslam_amd import capi; import ctypes as C; L = capi.lib(); print(L.xfh_comm_library().decode()); "
            "print(L.xfh_comm_unique_id(C.create_string_buffer(128)) if sys.argv[1] == '1' else 0)")
    def run(env_lib, want_id=True):          # (a real librccl on a box without a GPU fails ncclGetUniqueId noisily: only its identity is asked for)
        env = dict(os.environ)
        env.pop("XFH_RCCL_LIB", None)
        if env_lib is not None:
            env["XFH_RCCL_LIB"] = env_lib
        r = subprocess.run([sys.executable, "-c", code, "1" if want_id else "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out = r.stdout.strip().splitlines()
        return out[-2] if len(out) > 1 else "", int(out[-1])
    which, rc = run("/nonexistent/librccl.so.1")
    assert which == "" and rc == capi.ERR_COMM                    # explicit and absent: an error, not a silent fallback to another copy
    stub = os.path.join(ROOT, "tests", "stubs", "librccl.so.1")
    if not os.path.exists(stub):
        subprocess.check_call(["make", "-C", os.path.dirname(stub), "-s"])
    which, rc = run(stub)
    assert "tests/stubs/librccl.so.1" in which and "RCCL 0.0.0" in which and "HIP runtime" in which and rc == 0
    which, _ = run(None, want_id=False)
    if os.path.exists("/opt/rocm/lib/librccl.so.1"):
        assert os.path.realpath("/opt/rocm/lib/librccl.so.1") in which and "RCCL 0.0.0" not in which, which


def test_descriptor_distance_host(oracle_mod):
    from xfeatslam_amd.extractor import ORBmatcher
    d1, d2 = synth.descriptor_sets(64, 64, noise=0.4, zero_rows=2)
    for i in range(64):
        assert ORBmatcher.DescriptorDistance(d1[i], d2[i]) == oracle_mod.descriptor_distance(d1[i], d2[i])
    assert ORBmatcher.TH_LOW == 100 and ORBmatcher.TH_HIGH == 1000


def test_weight_blob_roundtrip_and_determinism():
    w = WT.make_synthetic(1234)
    assert sum(v.size for v in w.values()) == WT.N_PARAMS == 657910
    blob = WT.pack_blob(w)
    w2 = WT.unpack_blob(blob)
    assert list(w2) == [n for n, _ in WT.TENSORS]
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    assert WT.pack_blob(WT.make_synthetic(1234)) == blob
    assert float(np.abs(w["block3.1.layer.0.weight"]).max()) <= 1.0 / np.sqrt(576) + 1e-7
    sd = {"net." + k: v for k, v in w.items()}
    assert all(np.array_equal(WT.from_state_dict(sd)[k], w[k]) for k in w)
    img = synth.image(64, 96, 5)
    assert img.dtype == np.uint8 and img.min() == 0 and img.max() == 255 and np.array_equal(img, synth.image(64, 96, 5))


def test_scale_tables_match_reference_formula():
    # XFextractor.cc:80-96 with the TUM1.yaml values (scaleFactor 1.2, 8 levels), fp32 arithmetic
    from xfeatslam_amd.extractor import scale_tables
    sf, isf, s2, is2 = scale_tables(8, 1.2)
    ref = [np.float32(1.0)]
    for _ in range(7):
        ref.append(np.float32(ref[-1] * np.float32(1.2)))
    assert np.array_equal(sf, np.array(ref, np.float32)) and np.array_equal(s2, sf * sf)
    assert np.array_equal(isf, np.float32(1.0) / sf) and np.array_equal(is2, np.float32(1.0) / s2)
    assert sf.dtype == np.float32 and len(sf) == 8


def test_cpp_dropin_compiles_and_links():
    out = "/tmp/xfh_dropin_test"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "dropin_test.cpp"), "-L" + os.path.join(ROOT, "xfeatslam_amd"), "-lxfeat_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "xfeatslam_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_cpp_dropin_opencv_branch_compiles():
    """the wrappers' `#if XFEAT_HAVE_OPENCV` branch (the reference's cv::InputArray / cv::OutputArray signature, include/XFextractor.h:41-43) is
    type-checked by a compiler: tests/cpp/cv_branch_test.cpp against tests/stubs/opencv_api, an API-shaped stand-in for the few cv:: members
    the branch uses (NOT OpenCV: the image has none; tests/test_gpu_dropin_cpp.py runs the program)"""
    out = "/tmp/xfh_cv_branch_test"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "stubs", "opencv_api"),
           os.path.join(ROOT, "tests", "cpp", "cv_branch_test.cpp"), "-L" + os.path.join(ROOT, "xfeatslam_amd"), "-lxfeat_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "xfeatslam_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # and the product headers never see the stand-in by themselves: without it on the include path they take the cvlite branch
    probe = '#include "xfeat/XFextractor.h"\nstatic_assert(XFEAT_HAVE_OPENCV == 0, "no OpenCV in this image");\nint main() { return 0; }\n'
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-x", "c++", "-"], input=probe, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_host_code_under_sanitizers(tmp_path):
    """libxfeat_hip's HOST code built with AddressSanitizer + UBSan (make -C xfeatslam_amd/csrc asan; device code is not
    instrumented): tests/cpp/asan_host_test.cpp drives every entry point that needs no GPU, hostile compact shards included"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "xfeatslam_amd", "csrc"), "asan", "-s", "-j8"])
    exe = str(tmp_path / "asan_host_test")
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "asan_host_test.cpp"), "-L" + os.path.join(ROOT, "xfeatslam_amd"), "-lxfeat_hip_asan",
                           "-Wl,-rpath," + os.path.join(ROOT, "xfeatslam_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "asan_host_test ok" in r.stdout, r.stderr[-3000:]


def test_convert_weights_tool(tmp_path):
    import torch
    w = WT.make_synthetic(99)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    sd["fine_matcher.0.weight"] = torch.zeros(512, 128)            # extra tensors are ignored
    sd["block1.0.layer.1.running_mean"] = torch.zeros(4)
    torch.save(sd, tmp_path / "xfeat.pt")
    r = subprocess.run(["python", os.path.join(ROOT, "tools", "convert_weights.py"), str(tmp_path / "xfeat.pt"), str(tmp_path / "o.xfhw")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "o.xfhw").read_bytes() == WT.pack_blob(w)


def test_convert_weights_tool_reads_a_libtorch_archive(tmp_path):
    """the form the reference actually loads (XFextractor.cc:133-137: torch::serialize::InputArchive + model->load): a TorchScript
    zip whose parameters carry libtorch's module-tree names -- Sequential children "0", "1", ..., BasicLayerImpl's Sequential
    registered as `layer` (XFeat.cc:22).  A scripted module with that tree, saved with torch.jit.save, converts key for key
    (SURVEY.md Appendix B), BatchNorm buffers and the unused fine_matcher included."""
    import torch
    from torch import nn
    w = WT.make_synthetic(321, with_bn=True)

    class BasicLayer(nn.Module):
        def __init__(self, cin, cout, k, stride):
            super().__init__()
            self.layer = nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False), nn.BatchNorm2d(cout, affine=False), nn.ReLU())

        def forward(self, x):
            return self.layer(x)

    class XFeatModel(nn.Module):
        def __init__(self):
            super().__init__()
            B = BasicLayer
            self.norm = nn.InstanceNorm2d(1)
            self.skip1 = nn.Sequential(nn.AvgPool2d(4, 4), nn.Conv2d(1, 24, 1))
            self.block1 = nn.Sequential(B(1, 4, 3, 1), B(4, 8, 3, 2), B(8, 8, 3, 1), B(8, 24, 3, 2))
            self.block2 = nn.Sequential(B(24, 24, 3, 1), B(24, 24, 3, 1))
            self.block3 = nn.Sequential(B(24, 64, 3, 2), B(64, 64, 3, 1), B(64, 64, 1, 1))
            self.block4 = nn.Sequential(B(64, 64, 3, 2), B(64, 64, 3, 1), B(64, 64, 3, 1))
            self.block5 = nn.Sequential(B(64, 128, 3, 2), B(128, 128, 3, 1), B(128, 128, 3, 1), B(128, 64, 1, 1))
            self.block_fusion = nn.Sequential(B(64, 64, 3, 1), B(64, 64, 3, 1), nn.Conv2d(64, 64, 1))
            self.heatmap_head = nn.Sequential(B(64, 64, 1, 1), B(64, 64, 1, 1), nn.Conv2d(64, 1, 1), nn.Sigmoid())
            self.keypoint_head = nn.Sequential(B(64, 64, 1, 1), B(64, 64, 1, 1), B(64, 64, 1, 1), nn.Conv2d(64, 65, 1))
            self.fine_matcher = nn.Sequential(nn.Linear(128, 512), nn.BatchNorm1d(512, affine=False), nn.ReLU(), nn.Linear(512, 64))

        def forward(self, x):
            return self.block1(self.norm(x))

    m = XFeatModel()
    sd = m.state_dict()
    # key for key: every tensor the path consumes exists under the Appendix-B name with the Appendix-B shape
    for name, shape in WT.TENSORS + WT.BN_TENSORS:
        assert name in sd and tuple(sd[name].shape) == tuple(shape), name
    with torch.no_grad():
        for name, _ in WT.TENSORS + WT.BN_TENSORS:
            sd[name].copy_(torch.from_numpy(w[name]))
    torch.jit.save(torch.jit.script(m), str(tmp_path / "xfeat.pt"))
    with pytest.raises(Exception):
        torch.load(str(tmp_path / "xfeat.pt"), map_location="cpu", weights_only=True)          # not a torch.save pickle: only the jit branch can read it
    r = subprocess.run(["python", os.path.join(ROOT, "tools", "convert_weights.py"), str(tmp_path / "xfeat.pt"), str(tmp_path / "o.xfhw")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "o.xfhw").read_bytes() == WT.pack_blob(w)


def test_png_reader_and_gray_conversion(tmp_path):
    """include/xfeat/image_io.h (replay harness input): every PNG filter type, gray / gray+alpha / RGB / RGBA, several IDAT chunks,
    and the reference's colour conversion (imread BGR order + Camera.RGB flag, OpenCV fixed point); cross-checked with PIL's
    decoder when it is installed"""
    import subprocess
    from pngutil import opencv_gray, write_png
    exe = str(tmp_path / "image_io_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "image_io_test.cpp"), "-lz", "-o", exe])
    rng = np.random.RandomState(4)
    smooth = (np.add.outer(np.arange(37), np.arange(53)) * 3 % 256).astype(np.uint8)
    cases = [("g", smooth), ("ga", np.stack([smooth, 255 - smooth], -1)), ("rgb", rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)),
             ("rgba", np.stack([smooth, smooth[::-1], rng.randint(0, 256, smooth.shape).astype(np.uint8), smooth], -1))]
    for name, img in cases:
        for filt in ([0], [1], [2], [3], [4], [0, 1, 2, 3, 4]):
            path = str(tmp_path / f"{name}_{len(filt)}_{filt[0]}.png")
            write_png(path, img, filt)
            try:
                from PIL import Image
                assert np.array_equal(np.asarray(Image.open(path)).reshape(img.shape), img)          # the writer itself is a valid encoder
            except ImportError:
                pass
            for flag in (0, 1):
                out = str(tmp_path / "o.bin")
                subprocess.check_call([exe, path, str(flag), out])
                raw = open(out, "rb").read()
                hdr, data = raw.split(b"\n", 1)
                assert tuple(map(int, hdr.split())) == (37, 53, 1 if img.ndim == 2 else img.shape[2])
                assert np.array_equal(np.frombuffer(data, np.uint8).reshape(37, 53), opencv_gray(img, flag)), (name, filt, flag)
    # PGM path and a truncated PNG
    with open(tmp_path / "a.pgm", "wb") as f:
        f.write(b"P5\n# c\n53 37\n255\n" + smooth.tobytes())
    subprocess.check_call([exe, str(tmp_path / "a.pgm"), "1", str(tmp_path / "o.bin")])
    assert open(tmp_path / "o.bin", "rb").read().split(b"\n", 1)[1] == smooth.tobytes()
    bad = open(tmp_path / "rgb_1_0.png", "rb").read()[:200]
    open(tmp_path / "bad.png", "wb").write(bad)
    assert subprocess.call([exe, str(tmp_path / "bad.png"), "1", str(tmp_path / "o.bin")]) == 1
    # crafted headers: a PNG whose IHDR claims 2^31 x 2^31 pixels and a PGM claiming 10^6 x 10^6 are refused before anything is
    # allocated; a second IHDR does not override the first
    import struct
    import zlib
    good = open(tmp_path / "g_1_0.png", "rb").read()

    def chunk(ty, data):
        return struct.pack(">I", len(data)) + ty + data + struct.pack(">I", zlib.crc32(ty + data) & 0xFFFFFFFF)
    huge = good[:8] + chunk(b"IHDR", struct.pack(">IIBBBBB", 0x7FFFFFFF, 0x7FFFFFFF, 8, 0, 0, 0, 0)) + good[33:]
    open(tmp_path / "huge.png", "wb").write(huge)
    assert subprocess.call([exe, str(tmp_path / "huge.png"), "1", str(tmp_path / "o.bin")]) == 1
    twice = good[:33] + chunk(b"IHDR", struct.pack(">IIBBBBB", 5, 5, 8, 0, 0, 0, 0)) + good[33:]
    open(tmp_path / "twice.png", "wb").write(twice)
    subprocess.check_call([exe, str(tmp_path / "twice.png"), "1", str(tmp_path / "o.bin")])
    assert open(tmp_path / "o.bin", "rb").read().split(b"\n", 1)[0] == b"37 53 1"
    open(tmp_path / "huge.pgm", "wb").write(b"P5\n1000000 1000000\n255\n")
    assert subprocess.call([exe, str(tmp_path / "huge.pgm"), "1", str(tmp_path / "o.bin")]) == 1


def test_panel_layout_is_conflict_free():
    """xfeatslam_amd/csrc/mnn_layout.h: the panel image of the match GEMM.  Restates mnn_pos / mnn_swz / mnn_piece and checks,
    for every wave position, tile, k group and lane half, that the 16-lane groups in which a ds_read_b128 is serviced
    ({0-3,12-15,20-27}, {4-11,16-19,28-31} per lane half; MI355X_MICROARCH.md, LDS) touch 16 different 16-byte slots of the
    256-byte bank row -- for BOTH read patterns (as d1: MFMA rows; as d2: MFMA columns) -- and that the position map is a
    permutation."""
    def pos(row): return (row & 128) | ((row & 3) << 5) | ((row & 127) >> 2)
    def swz(p): return ((p >> 2) ^ (p >> 5)) & 3
    def addr(p, g, half): return (g >> 1) * 4096 + p * 16 + ((((g & 1) << 1 | half) ^ swz(p)) << 2)
    assert sorted(pos(r) for r in range(256)) == list(range(256))
    hdr = open(os.path.join(ROOT, "xfeatslam_amd", "csrc", "mnn_layout.h")).read()
    assert "(r256 & 128) | ((r256 & 3) << 5) | ((r256 & 127) >> 2)" in hdr and "((pos >> 2) ^ (pos >> 5)) & 3" in hdr
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    patterns = [lambda i, wr=wr, rt=rt: wr * 64 + ((i >> 2) & 1) * 32 + rt * 16 + (i & 3) + 4 * ((i >> 3) & 3) for wr in range(4) for rt in range(2)]
    patterns += [lambda i, wc=wc, ct=ct: wc * 128 + 4 * i + ct for wc in range(2) for ct in range(4)]
    for row_of_lane in patterns:
        for g in range(8):
            for half in range(2):
                for grp in groups:
                    slots = {(addr(pos(row_of_lane(i)), g, half) // 4) % 16 for i in grp}
                    assert len(slots) == 16


def test_many_pairs_work_plan_invariants():
    """xfeatslam_amd/csrc/mnn_seg_plan.h (the plan of xfh_match_mnn_prepared_batch_device, shared by the host, k_mnn_gemm_seg and k_mnn_post): for random mixes of
    pair shapes and workgroup counts the workgroups' tile ranges partition the tile sequence into G non-empty, contiguous, balanced pieces; every d1 panel
    (a run of P2 consecutive tiles) is touched by a run of consecutive workgroups whose count never exceeds what the key buffer reserves for it, and the
    row-key plane index 2 * (w - w_first) + group the GEMM writes stays below the plane count k_mnn_post reads for that panel."""
    L = capi.lib()
    rng = np.random.RandomState(7)
    shapes = [[(4096, 4096)] * 8, [(4096, 4096)], [(1, 5)], [(300, 200), (1, 5), (5, 1), (129, 127), (257, 4097), (1000, 777)], [(4096, 4096)] * 16]
    for _ in range(40):
        P = int(rng.randint(1, 17))
        shapes.append([(int(rng.randint(1, 6000)), int(rng.randint(1, 6000))) for _ in range(P)])
    for sh in shapes:
        for num_cu in (256, 1, 7, 100, 304):
            P = len(sh)
            n1 = (C.c_int * P)(*[a for a, _ in sh]); n2 = (C.c_int * P)(*[b for _, b in sh])
            T, G, keys = C.c_int(), C.c_int(), C.c_ulonglong()
            tile0 = (C.c_int * P)(); pmax = (C.c_int * P)(); lo = (C.c_int * (num_cu + 1))()
            assert L.xfh_debug_match_plan(P, n1, n2, num_cu, C.byref(T), C.byref(G), tile0, pmax, lo, C.byref(keys)) == 0
            P1 = [(a + 255) // 256 for a, _ in sh]; P2 = [(b + 255) // 256 for _, b in sh]
            assert T.value == sum(a * b for a, b in zip(P1, P2)) and G.value == min(T.value, num_cu)
            lo = list(lo)[:G.value + 1]
            assert lo[0] == 0 and lo[-1] == T.value and all(b > a for a, b in zip(lo, lo[1:]))          # a partition into non-empty ranges
            sizes = [b - a for a, b in zip(lo, lo[1:])]
            assert max(sizes) - min(sizes) <= 1                                                           # balanced
            wg_of = np.repeat(np.arange(G.value), sizes)
            assert np.array_equal(wg_of, (np.arange(T.value, dtype=np.int64) * G.value) // T.value)     # = tile * G / T, the form the kernels use
            need = 0
            for p in range(P):
                assert tile0[p] == sum(a * b for a, b in zip(P1[:p], P2[:p]))
                for by in range(P1[p]):
                    t0 = tile0[p] + by * P2[p]
                    ws = wg_of[t0:t0 + P2[p]]
                    assert np.all(np.diff(ws) >= 0) and np.all(np.diff(ws) <= 1)                          # consecutive workgroups
                    planes = 2 * (int(ws[-1]) - int(ws[0]) + 1)
                    assert planes <= pmax[p], (sh, num_cu, p, by, planes, pmax[p])
                need += (pmax[p] * P1[p] + P1[p] * P2[p] + P1[p]) * 256
            assert keys.value == need
    assert L.xfh_debug_match_plan(0, None, None, 256, None, None, None, None, None, None) == 1


def test_match_gemm_register_contract():
    """k_mnn_gemm_img and k_mnn_gemm_seg run two waves per SIMD on a budget of 256 VGPRs each: accumulators 128 + (seg: the d1 strip 64 + operands 32).  A spill
    inside the K loop would put scratch traffic in front of every MFMA group; the one spill that exists (a 64-bit constant of the rare new-d1-panel path of
    k_mnn_gemm_seg) must stay the only one."""
    import re
    import shutil
    import subprocess
    if not shutil.which("c++filt") or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs hipcc and c++filt")
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "kres.sh"), "kernels_mnn_gemm.hip", "-fno-honor-nans"], capture_output=True, text=True, timeout=600).stdout
    rows = {}
    for line in out.splitlines():
        m = re.match(r"(?:void )?(\S.*?)\s+vgpr\s+(\d+)\s+agpr\s+(\d+)\s+scratch\s+(\d+)\s+occ\s+(\d+)", line)
        if m:
            rows[m.group(1)] = tuple(int(x) for x in m.groups()[1:])
    img = [k for k in rows if k.startswith("k_mnn_gemm_img")]; seg = [k for k in rows if k.startswith("k_mnn_gemm_seg")]
    assert len(img) == 1 and len(seg) == 1, out[-1500:]
    assert rows[img[0]][2] == 0 and rows[img[0]][0] + rows[img[0]][1] <= 256 and rows[img[0]][3] >= 2, rows[img[0]]
    assert rows[seg[0]][2] <= 16 and rows[seg[0]][0] + rows[seg[0]][1] <= 256 and rows[seg[0]][3] >= 2, rows[seg[0]]


def test_kernel_occupancy_contract():
    """The throughput kernels sit at register-count edges: the dominant 3x3 64->64 convolution and its block_fusion.0 instance run
    two 8-wave workgroups per CU (<= 128 VGPRs), and a refactor that nudges the allocator over the edge halves their occupancy
    without any test failing (it happened twice in round 3: 118 -> 130 and 127 -> 166 VGPRs, +15 % / +30 % on those kernels).
    tools/kres.sh reads clang's kernel-resource-usage remarks (no GPU needed); no kernel may spill to scratch."""
    import re
    import shutil
    import subprocess
    if not shutil.which("c++filt") or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs hipcc and c++filt")
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "kres.sh"), "kernels_conv.hip"], capture_output=True, text=True, timeout=600).stdout
    rows = {}
    for line in out.splitlines():
        m = re.match(r"(?:void )?(\S.*?)\s+vgpr\s+(\d+)\s+agpr\s+(\d+)\s+scratch\s+(\d+)\s+occ\s+(\d+)", line)
        if m:
            rows[m.group(1)] = tuple(int(x) for x in m.groups()[1:])
    assert len(rows) > 40, out[-2000:]
    spills = {k: v for k, v in rows.items() if v[2] != 0}
    assert not spills, spills
    edge = [k for k in rows if re.match(r"k_conv_mfma<64, 64, 3, 1, 4, 2, 1, 16, [17], [02], 32, 1", k)]      # PRO_BN and PRO_FUSEA (block_fusion.0 on the activated pyramid maps)
    assert len(edge) == 4, sorted(rows)
    for k in edge:
        vgpr, agpr, _, occ = rows[k]
        assert vgpr + agpr <= 128 and occ >= 4, (k, rows[k])
