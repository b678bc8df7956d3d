#!/usr/bin/env python
"""Stage-by-stage parity report of libxfeat_hip.so against the CPU oracle, plus first
timings.  Run on a GPU box:  python tools/gpu_stage_check.py [--quick]
Writes gpurun_out/stage_check.log.  Development tool (uses the oracle as the checker)."""
from __future__ import annotations

import ctypes as C
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xfeatslam_amd import capi, synth, weights as WT          # noqa: E402
from xfeatslam_amd.extractor import Context                   # noqa: E402
from oracle import oracle as O                                # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "stage_check.log"), "w")


def P(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n"); LOG.flush()


def section(name, fn):
    P(f"\n=== {name}")
    t = time.time()
    try:
        fn()
    except Exception:
        P("EXCEPTION:\n" + traceback.format_exc())
    P(f"--- {name}: {time.time() - t:.1f}s")


LAYER_NAMES = ["block1.0", "block1.1", "block1.2", "block1.3", "block2.0", "block2.1", "block3.0", "block3.1", "block3.2",
               "block4.0", "block4.1", "block4.2", "block5.0", "block5.1", "block5.2", "block5.3", "fusion.0", "fusion.1",
               "heat.0", "heat.1", "kp.0", "kp.1", "kp.2"]


def cmp(name, a, b):
    if a.shape != b.shape:
        P(f"  {name:12s} SHAPE MISMATCH hip {a.shape} oracle {b.shape}")
        return
    if a.size == 0:
        P(f"  {name:12s} empty"); return
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    bad = int((~np.isfinite(a)).sum())
    P(f"  {name:12s} n={a.size:8d} max|d|={d.max():.3e} mean|d|={d.mean():.3e} max|ref|={np.abs(b).max():.3e} "
      f"exact={100.0 * (a == b).mean():6.2f}% nonfinite={bad} argmax={int(d.argmax())}")


def match_checks():
    ctx = Context(nfeatures=64, max_height=32, max_width=32)
    for (n1, n2, z, noise) in [(256, 256, 0, 0.3), (300, 200, 7, 0.3), (1, 5, 0, 0.3), (129, 127, 0, 0.5),
                               (4096, 4096, 0, 0.3), (4096, 4096, 100, 0.3), (1000, 4096, 0, 0.4)]:
        d1, d2 = synth.descriptor_sets(n1, n2, zero_rows=z, noise=noise)
        a = O.match_mnn(d1, d2)
        b = ctx.match_mnn(d1, d2)
        same = len(a[0]) == len(b[0]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        dd = float(np.nanmax(np.abs(a[2] - b[2]))) if same and len(a[2]) else float("nan")
        P(f"  mnn {n1}x{n2} zero={z}: oracle {len(a[0])} hip {len(b[0])} same_pairs={same} max|ddist|={dd:.3e} "
          f"nan_equal={np.array_equal(np.isnan(a[2]), np.isnan(b[2])) if same else None}")
        if not same and len(b[0]):
            P("    first hip pairs", b[0][:8], b[1][:8], "oracle", a[0][:8], a[1][:8])
    d1, d2 = synth.descriptor_sets(512, 384, noise=0.3)
    x = O.distance_i32(d1, d2); y = ctx.distance_i32(d1, d2)
    P(f"  dist_i32 512x384 equal={np.array_equal(x, y)} ndiff={(x != y).sum()} range {y.min()}..{y.max()}")
    d1, d2 = synth.descriptor_sets(70, 33, noise=0.3, zero_rows=3)
    x = O.distance_i32(d1, d2); y = ctx.distance_i32(d1, d2)
    P(f"  dist_i32 70x33  equal={np.array_equal(x, y)} ndiff={(x != y).sum()}")
    ctx.close()


def match_timing():
    lib = capi.lib()
    ctx = Context(nfeatures=64, max_height=32, max_width=32)
    n = 4096
    d1, d2 = synth.descriptor_sets(n, n, noise=0.3)
    b1 = capi.DeviceBuffer(d1.nbytes).upload(d1); b2 = capi.DeviceBuffer(d2.nbytes).upload(d2)
    o = capi.DeviceBuffer(n * 12 + 64)
    def run():
        capi.check(lib.xfh_match_mnn_device(ctx.h, b1.ptr, n, b2.ptr, n, -1.0, o.ptr, o.ptr + 4 * n, o.ptr + 8 * n, o.ptr + 12 * n), ctx.h)
    for _ in range(5): run()
    ctx.synchronize()
    ctx.timing_enable(capi.K["MNN_GEMM"])
    t = time.perf_counter()
    K = 200
    for _ in range(K): run()
    ctx.synchronize()
    dt = (time.perf_counter() - t) / K
    nl, ms = ctx.timing_read()
    P(f"  mnn 4096x4096 whole call {dt * 1e6:.1f} us  -> {n * n / dt:.3e} pairs/s ; gemm kernel avg {ms / max(nl, 1) * 1e3:.2f} us over {nl} launches"
      f" -> {2.0 * n * n * 64 / (ms / max(nl, 1) * 1e-3) / 1e12:.1f} TFLOP/s")
    ctx.timing_enable(0)
    t = time.perf_counter()
    for _ in range(K): run()
    ctx.synchronize()
    dt = (time.perf_counter() - t) / K
    P(f"  mnn 4096x4096 whole call (no events) {dt * 1e6:.1f} us -> {n * n / dt:.3e} pairs/s")
    nm = o.download(np.int32, 1, 12 * n)[0]
    P(f"  n_matches {nm}")
    ctx.close()


def extract_checks(H, W, gain, nf=4096, lap=(0, 0), seed=42):
    w = WT.make_synthetic(1234, gain)
    blob = WT.pack_blob(w)
    img = synth.image(H, W, seed)
    orc = O.Oracle(blob)
    ok, od, onv, omono = orc.extract(img, nf, lap)
    ctx = Context(nfeatures=nf, max_height=H, max_width=W, max_batch=2)
    ctx.load_weights(blob)
    recs = ctx.extract_batch(np.stack([img, img[::-1].copy()]), lap)
    hk, hd, hnv, hmono, hnc = recs[0]
    P(f"  {H}x{W} gain={gain} lap={lap}: oracle n_valid={onv} mono={omono} cand={len(orc.tensor(O.T['CAND'])) // 3} | hip n_valid={hnv} mono={hmono} cand={hnc}")
    for nm in ["X", "XSTAT", "SKIP_POOL"]:
        cmp(nm, ctx.debug_tensor(capi.T[nm]), orc.tensor(O.T[nm]))
    for i in range(23):
        if i:                                                            # block1.0 is never materialised on the GPU
            cmp("raw " + LAYER_NAMES[i], ctx.debug_tensor(capi.T["RAW0"] + i), orc.tensor(O.T["RAW0"] + i))
        cmp("st  " + LAYER_NAMES[i], ctx.debug_tensor(capi.T["STAT0"] + i), orc.tensor(O.T["STAT0"] + i))
        if i == 17:
            cmp("FEATS", ctx.debug_tensor(capi.T["FEATS"]), orc.tensor(O.T["FEATS"]))
    cmp("H1", ctx.debug_tensor(capi.T["H1"]), orc.tensor(O.T["H1"]))
    cmp("K1H", ctx.debug_tensor(capi.T["K1H"]), orc.tensor(O.T["K1H"]))
    hs = ctx.debug_tensor(capi.T["SEL"]).reshape(-1, 3); os_ = orc.tensor(O.T["SEL"]).reshape(-1, 3)
    P(f"  SEL rows hip {len(hs)} oracle {len(os_)}")
    s1 = set(map(tuple, hs[:, :2].astype(int))); s2 = set(map(tuple, os_[:, :2].astype(int)))
    P(f"  SEL xy-set equal={s1 == s2} symdiff={len(s1 ^ s2)}; same order={len(hs) == len(os_) and np.array_equal(hs[:, :2], os_[:, :2])}")
    # final outputs
    v1 = ok["size"] > 0; v2 = hk["size"] > 0
    k1 = set(zip(ok["x"][v1].astype(int), ok["y"][v1].astype(int))); k2 = set(zip(hk["x"][v2].astype(int), hk["y"][v2].astype(int)))
    P(f"  final keypoint sets equal={k1 == k2} |oracle|={len(k1)} |hip|={len(k2)} symdiff={len(k1 ^ k2)}")
    slots_same = np.array_equal(ok["x"], hk["x"]) and np.array_equal(ok["y"], hk["y"])
    P(f"  same slot layout={slots_same}; padding identical={np.array_equal(ok[~v1], hk[~v2]) if (~v1).sum() == (~v2).sum() else False}")
    d1 = {(int(k['x']), int(k['y'])): i for i, k in enumerate(ok) if k['size'] > 0}
    d2 = {(int(k['x']), int(k['y'])): i for i, k in enumerate(hk) if k['size'] > 0}
    common = [k for k in d1 if k in d2]
    if common:
        dm = max(float(np.abs(od[d1[k]] - hd[d2[k]]).max()) for k in common)
        sm = max(abs(float(ok[d1[k]]['response']) - float(hk[d2[k]]['response'])) for k in common)
        P(f"  joined: max|ddesc|={dm:.3e} max|dscore|={sm:.3e} over {len(common)} keypoints")
    # frame 1 (flipped image) sanity: batch index plumbing
    ok2, od2, onv2, _ = orc.extract(img[::-1].copy(), nf, lap)
    hk2 = recs[1][0]
    v1 = ok2["size"] > 0; v2 = hk2["size"] > 0
    k1 = set(zip(ok2["x"][v1].astype(int), ok2["y"][v1].astype(int))); k2 = set(zip(hk2["x"][v2].astype(int), hk2["y"][v2].astype(int)))
    P(f"  frame1 (flipped) sets equal={k1 == k2} |oracle|={len(k1)} |hip|={len(k2)}")
    ctx.close()


def extract_timing(B, H=480, W=640, gain=6.0, iters=30):
    lib = capi.lib()
    blob = WT.pack_blob(WT.make_synthetic(1234, gain))
    ctx = Context(nfeatures=4096, max_height=H, max_width=W, max_batch=B)
    ctx.load_weights(blob)
    fr = synth.frames(min(B, 4), H, W)
    fr = np.concatenate([fr] * ((B + len(fr) - 1) // len(fr)))[:B]
    din = capi.DeviceBuffer(fr.nbytes).upload(fr)
    dout = capi.DeviceBuffer(ctx.rec_bytes * B)
    def run():
        capi.check(lib.xfh_extract_batch_device(ctx.h, din.ptr, B, H, W, 0, 0, dout.ptr), ctx.h)
    for _ in range(3): run()
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(iters): run()
    ctx.synchronize()
    dt = (time.perf_counter() - t) / iters
    hdr = dout.download(np.int32, 4)
    P(f"  extract B={B} {H}x{W}: {dt * 1e3:.3f} ms/batch -> {B / dt:.1f} frames/s (n_valid {hdr[0]}, cand {hdr[2]})")
    # per-kernel-family breakdown
    for kname in ["PREPROC", "CONV_DIRECT", "CONV_MFMA", "HEADS", "NMS", "SELECT", "DESC"]:
        ctx.timing_enable(capi.K[kname])
        for _ in range(5): run()
        nl, ms = ctx.timing_read()
        P(f"    {kname:12s} {ms / 5:.3f} ms/batch over {nl // 5} launches")
    if B <= 8:
        for li in range(24):
            ctx.timing_enable(capi.K["CONV_MFMA"] if li >= 4 else capi.K["CONV_DIRECT"], 1 << li)
            for _ in range(5): run()
            nl, ms = ctx.timing_read()
            P(f"      conv layer {li:2d} {ms / 5 * 1e3:.1f} us/batch")
    ctx.timing_enable(0)
    ctx.close()


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    tonly = "--timing-only" in sys.argv
    P("lib:", capi.lib().xfh_version().decode(), "devices:", capi.lib().xfh_device_count())
    if not tonly:
        section("match parity", match_checks)
    section("match timing", match_timing)
    if not tonly:
        section("extract 96x128", lambda: extract_checks(96, 128, 1.0, nf=256))
        section("extract VGA gain1", lambda: extract_checks(480, 640, 1.0))
    if not quick and not tonly:
        section("extract VGA dense", lambda: extract_checks(480, 640, 6.0, lap=(0, 1000)))
        section("extract 720p", lambda: extract_checks(720, 1280, 1.0, lap=(0, 1000)))
    section("extract timing B=1", lambda: extract_timing(1))
    section("extract timing B=8", lambda: extract_timing(8))
    if not quick:
        section("extract timing B=32", lambda: extract_timing(32, iters=10))
    P("done")
