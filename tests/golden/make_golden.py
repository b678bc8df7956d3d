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


def extract_case(name, H, W, gain, nf, lap, seed=42, keep_desc=64):
    w = WT.make_synthetic(1234, gain)
    img = synth.image(H, W, seed)
    taps = {}
    kps, desc, nv, mono = TR.extract(img, w, nf, lap, taps)
    v = np.where(kps["size"] > 0)[0]
    # canonical order (y, x) so that the fixture does not depend on near-tie ordering
    order = v[np.lexsort((kps["x"][v], kps["y"][v]))]
    rows = order[:: max(1, len(order) // keep_desc)][:keep_desc]
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        H=H, W=W, gain=gain, nfeatures=nf, lap=np.array(lap), seed=seed, n_valid=nv, mono_index=mono,
        n_candidates=int(taps["cand"].shape[1]),
        xy=np.stack([kps["x"][order], kps["y"][order]], 1).astype(np.int32),
        score=kps["response"][order].astype(np.float32),
        desc_rows_xy=np.stack([kps["x"][rows], kps["y"][rows]], 1).astype(np.int32),
        desc_rows=desc[rows].astype(np.float32),
        desc_checksum=np.float64(np.abs(desc[order].astype(np.float64)).sum()),
        h1_sample=taps["H1"][0, 0].numpy()[::7, ::7].astype(np.float32),
        k1h_sample=taps["K1h"][0, 0].numpy()[::37, ::41].astype(np.float32),
        front_xy=np.stack([kps["x"][:mono], kps["y"][:mono]], 1).astype(np.int32)[:32],
    )
    print(name, "n_valid", nv, "mono", mono, "cand", taps["cand"].shape[1])


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


if __name__ == "__main__":
    extract_case("extract_96x128", 96, 128, 1.0, 256, (0, 0))
    extract_case("extract_vga", 480, 640, 1.0, 4096, (0, 0))
    extract_case("extract_vga_dense_mono", 480, 640, 6.0, 4096, (0, 1000))
    extract_case("extract_720p", 720, 1280, 1.0, 4096, (0, 1000))
    extract_case("extract_odd_170x230", 170, 230, 2.0, 300, (100, 150))
    match_case("match_256", 256, 256, 0, 0.3)
    match_case("match_300x200_zero7", 300, 200, 7, 0.3)
    match_case("match_4096", 4096, 4096, 0, 0.3)
    match_case("match_4096_zero100", 4096, 4096, 100, 0.3)
