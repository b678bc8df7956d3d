#!/usr/bin/env python
"""Generates the committed golden vectors from the ATen-operator restatement
(oracle/torch_restatement.py), i.e. from the same libtorch CPU kernels the reference
executes.  Run in the development container:  python tests/golden/make_golden.py
Inputs are regenerated from seeds (xfeatslam_amd/synth.py, weights.py); only expected
outputs are stored.  Fixtures are data, not reference source."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import torch_restatement as TR          # noqa: E402
from xfeatslam_amd import synth, weights as WT      # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def extract_case(name, H, W, gain, nf, lap, seed=42, keep_desc=64, family="", wseed=1234, image_family="noise"):
    """family "" = the round-1 weight set make_synthetic(1234, gain); otherwise a member of weights.FAMILIES (round 5: the parity campaign)"""
    w = WT.make_family(family, wseed, gain) if family else WT.make_synthetic(wseed, gain)
    img = synth.image_family(image_family, H, W, seed)
    taps = {}
    kps, desc, nv, mono = TR.extract(img, w, nf, lap, taps)
    v = np.where(kps["size"] > 0)[0]
    # canonical order (y, x) so that the fixture does not depend on near-tie ordering
    order = v[np.lexsort((kps["x"][v], kps["y"][v]))]
    rows = order[:: max(1, len(order) // keep_desc)][:keep_desc]
    # the score the top-k cut fell on and its distance to the first score left out (NaN: every candidate was kept).  Near-tied scores
    # around the cut are ordered by summation noise on either side, conftest.check_extract_golden uses these two
    cs = np.sort(taps["cand_scores"][0].numpy())[::-1]
    cut_score, cut_gap = (float(cs[nf - 1]), float(cs[nf - 1] - cs[nf])) if len(cs) > nf else (float("nan"), float("nan"))
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        H=H, W=W, gain=gain, nfeatures=nf, lap=np.array(lap), seed=seed, n_valid=nv, mono_index=mono,
        family=family, wseed=wseed, image_family=image_family,
        n_candidates=int(taps["cand"].shape[1]), cut_score=np.float32(cut_score), cut_gap=np.float32(cut_gap),
        xy=np.stack([kps["x"][order], kps["y"][order]], 1).astype(np.int32),
        score=kps["response"][order].astype(np.float32),
        desc_rows_xy=np.stack([kps["x"][rows], kps["y"][rows]], 1).astype(np.int32),
        desc_rows=desc[rows].astype(np.float32),
        desc_checksum=np.float64(np.abs(desc[order].astype(np.float64)).sum()),
        h1_sample=taps["H1"][0, 0].numpy()[::7, ::7].astype(np.float32),
        k1h_sample=taps["K1h"][0, 0].numpy()[::37, ::41].astype(np.float32),
        front_xy=np.stack([kps["x"][:mono], kps["y"][:mono]], 1).astype(np.int32)[:32],
    )
    print(f"{name:44s} n_valid {nv:5d} mono {mono:5d} cand {taps['cand'].shape[1]:6d} cut_gap {cut_gap:.2e}", flush=True)


def match_case(name, n1, n2, zero_rows, noise):
    d1, d2 = synth.descriptor_sets(n1, n2, zero_rows=zero_rows, noise=noise)
    i0, i1, dist, cos = TR.match_mnn(d1, d2)
    # margin of every row/column arg-max (for the near-tie audit)
    top2 = np.partition(cos, -2, axis=1)[:, -2:]
    gap = float((top2[:, 1] - top2[:, 0]).min()) if n2 > 1 else 1.0
    np.savez_compressed(os.path.join(OUT, name + ".npz"), n1=n1, n2=n2, zero_rows=zero_rows, noise=noise,
                        idx1=i0, idx2=i1, dist=dist.astype(np.float32), min_row_gap=gap,
                        dist_i32_corner=TR.distance_i32(d1[:48], d2[:40]))
    print(name, "matches", len(i0), "min top1-top2 gap", gap)


def campaign_cases():
    """Round 6: ATen fixtures for EVERY weight family of weights.FAMILIES at full size, so that the GPU path meets libtorch-operator
    output first-hand and not only through the C oracle.  Deterministic table:
      22 x VGA    family i with image families (3i) % 8 and (3i + 4) % 8 (every image family is met >= 2 times), nfeatures 4096
      11 x 720p   family i with image family (5i + 1) % 8 (resized to 1280x704 inside)
       4 x VGA    nfeatures = 1000 with lapping {0,0} / {0,1000}: /root/reference/examples/RGB-D/TUM1.yaml:43-55 and Frame.cc:495"""
    IMF, FAM = synth.IMAGE_FAMILIES, WT.FAMILIES
    laps_vga = [(0, 0), (0, 213), (100, 400), (0, 1000)]
    out = []
    for i, fam in enumerate(FAM):
        for j, k in enumerate(((3 * i) % 8, (3 * i + 4) % 8)):
            out.append(dict(name=f"c6_vga_{fam}_{IMF[k]}", H=480, W=640, gain=3.0, nf=4096, lap=laps_vga[(2 * i + j) % 4],
                            seed=200 + 2 * i + j, family=fam, wseed=11 + i, image_family=IMF[k]))
    for i, fam in enumerate(FAM):
        k = (5 * i + 1) % 8
        out.append(dict(name=f"c6_720p_{fam}_{IMF[k]}", H=720, W=1280, gain=3.0, nf=4096, lap=((0, 1000), (0, 426), (0, 0))[i % 3],
                        seed=300 + i, family=fam, wseed=31 + i, image_family=IMF[k]))
    for i, (fam, imf) in enumerate((("uniform", "noise"), ("normal", "gradient"), ("scaled", "steps"), ("peaky", "blobs"))):
        out.append(dict(name=f"c6_vga_nf1000_{fam}_{imf}", H=480, W=640, gain=3.0, nf=1000, lap=((0, 0), (0, 1000))[i % 2],
                        seed=400 + i, family=fam, wseed=51 + i, image_family=imf))
    return out


CAMPAIGN_NAMES = [c["name"] for c in campaign_cases()]
DESC_Q = 16384.0          # match_x_* fixtures: descriptor blocks stored as int16 = rint(desc * 2^14), exactly representable in fp32


def second_view(img, dy, dx, seed):
    """a second frame of the same scene: the first one shifted by (dy, dx) pixels (edge replicated) plus +-2 grey levels of noise"""
    H, W = img.shape
    p = np.pad(img, ((abs(dy),) * 2, (abs(dx),) * 2), mode="edge")
    b = p[abs(dy) + dy:abs(dy) + dy + H, abs(dx) + dx:abs(dx) + dx + W].astype(np.int32)
    n = np.rint((WT.uniform01(seed, 77, H * W).reshape(H, W) - 0.5) * 4.0).astype(np.int32)
    return np.clip(b + n, 0, 255).astype(np.uint8)


def match_extracted_case(name, family, imf, H, W, nf1, nf2, lap, wseed, seed, dup=0):
    """ORBmatcher::match on EXTRACTED descriptor blocks: two views of one scene through TR.extract (zero padding rows where the frame has
    fewer than nfeatures valid keypoints, the lapping split's back-to-front rows), `dup` duplicated rows on each side, quantised to
    int16 / 2^14 and stored -- the fixture carries its inputs because extracted descriptors differ by ~1e-5 between implementations,
    enough to flip an arg-max -- then through TR.match_mnn."""
    w = WT.make_family(family, wseed, 3.0)
    a = synth.image_family(imf, H, W, seed)
    b = second_view(a, 3, -5, seed)
    _, da, nva, _ = TR.extract(a, w, nf1, lap)
    _, db, nvb, _ = TR.extract(b, w, nf2, lap)
    q1 = np.rint(da * DESC_Q).astype(np.int16); q2 = np.rint(db * DESC_Q).astype(np.int16)
    for t in range(dup):                              # exact duplicates: every arg-max must resolve to the lowest index
        q2[(37 * t + 11) % nf2] = q2[(53 * t + 3) % nf2]
        q1[(41 * t + 7) % nf1] = q1[(59 * t + 5) % nf1]
    d1 = q1.astype(np.float32) / np.float32(DESC_Q); d2 = q2.astype(np.float32) / np.float32(DESC_Q)
    i0, i1, dist, cos = TR.match_mnn(d1, d2)
    top2 = np.partition(cos, -2, axis=1)[:, -2:]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), q1=q1, q2=q2, n_valid=np.array([nva, nvb]), idx1=i0, idx2=i1,
                        dist=dist.astype(np.float32), min_row_gap=float((top2[:, 1] - top2[:, 0]).min()),
                        dist_i32_corner=TR.distance_i32(d1[:48], d2[:40]), dist_i32_tail=TR.distance_i32(d1[-24:], d2[-24:]))
    print(f"{name:44s} {nf1}x{nf2} valid {nva}/{nvb} zero rows {int((~q1.any(1)).sum())}/{int((~q2.any(1)).sum())} matches {len(i0)}", flush=True)


MATCH_X = [
    dict(name="match_x_uniform_noise_1000", family="uniform", imf="noise", H=480, W=640, nf1=1000, nf2=1000, lap=(0, 0), wseed=61, seed=500),
    dict(name="match_x_pruned_blobs_pad", family="pruned", imf="blobs", H=480, W=640, nf1=2048, nf2=2048, lap=(0, 1000), wseed=16, seed=210),
    dict(name="match_x_tiny_saturated_pad_dup", family="tiny", imf="saturated", H=480, W=640, nf1=1000, nf2=1000, lap=(0, 0), wseed=15, seed=208, dup=6),
    dict(name="match_x_pruned_checker8_dup", family="pruned", imf="checker8", H=480, W=640, nf1=1000, nf2=1000, lap=(0, 213), wseed=63, seed=502, dup=24),
    dict(name="match_x_heavy_steps_720p_ragged", family="heavy", imf="steps", H=720, W=1280, nf1=4096, nf2=1000, lap=(0, 426), wseed=64, seed=503, dup=5),
]
MATCH_X_NAMES = [c["name"] for c in MATCH_X]


if __name__ == "__main__":
    # python tests/golden/make_golden.py [r1] [r5] [campaign] [match_x]      (no argument: everything)
    which = set(sys.argv[1:]) or {"r1", "r5", "campaign", "match_x"}
    import torch
    if "r1" in which:
        # the five round-1 fixtures are kept byte for byte as committed in round 1 (`git checkout` them after a run: np.savez
        # re-compresses); they were generated by this code before the `family` / `cut_*` keys existed and load without them
        extract_case("extract_96x128", 96, 128, 1.0, 256, (0, 0))
        extract_case("extract_vga", 480, 640, 1.0, 4096, (0, 0))
        extract_case("extract_vga_dense_mono", 480, 640, 6.0, 4096, (0, 1000))
        extract_case("extract_720p", 720, 1280, 1.0, 4096, (0, 1000))
        extract_case("extract_odd_170x230", 170, 230, 2.0, 300, (100, 150))
        match_case("match_256", 256, 256, 0, 0.3)
        match_case("match_300x200_zero7", 300, 200, 7, 0.3)
        match_case("match_4096", 4096, 4096, 0, 0.3)
        match_case("match_4096_zero100", 4096, 4096, 100, 0.3)
    # everything below at one thread (libtorch's vector kernels on every element, see oracle/xfeat_oracle.c: xfo_expf)
    torch.set_num_threads(1)
    if "r5" in which:
        # round 5: four cases of the parity campaign (weight family x image family)
        extract_case("extract_vga_pruned_blobs", 480, 640, 3.0, 4096, (0, 213), seed=100, family="pruned", wseed=5, image_family="blobs")
        extract_case("extract_vga_heavy_saturated", 480, 640, 3.0, 2000, (100, 400), seed=101, family="heavy", wseed=5, image_family="saturated")
        extract_case("extract_720p_denormal_lowcontrast", 720, 1280, 3.0, 4096, (0, 426), seed=100, family="heat_denormal", wseed=5, image_family="lowcontrast")
        extract_case("extract_vga_dc_steps", 480, 640, 3.0, 4096, (0, 0), seed=101, family="dc", wseed=5, image_family="steps")
    if "campaign" in which:
        # round 6: every weight family at VGA (x2 image families) and at 720p, the TUM1.yaml nfeatures = 1000 cases
        for c in campaign_cases():
            extract_case(**c)
    if "match_x" in which:
        # round 6: ORBmatcher::match on extracted descriptor blocks
        for c in MATCH_X:
            match_extracted_case(**c)
