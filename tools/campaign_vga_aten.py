#!/usr/bin/env python
"""The body of tests/test_oracle.py::test_campaign_oracle_vs_aten at FULL size for all 88 (weight family, image family) pairs: the C
oracle against the ATen-operator restatement (libtorch's CPU kernels, one thread) at 480x640, nfeatures 4096.  The CPU suite runs the 88
pairs at 160x224 and one image family per weight family at VGA; this is the whole grid once, kept as a log:

    python tools/campaign_vga_aten.py > profiles/r06_campaign_vga_aten.log

Development container only (it imports torch on the CPU; nothing here runs on the GPU box).  Exit status 1 if any pair breaks the rule of
the test: identical candidate / valid counts, identical keypoint sets except inside a group of scores within 2e-6 at the top-k cut,
descriptors within 1e-4 (position-joined)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                                             # noqa: E402
from conftest import joined_desc_diff, kp_set            # noqa: E402
from oracle import oracle as O                           # noqa: E402
from oracle import torch_restatement as TR               # noqa: E402
from xfeatslam_amd import synth, weights as WT           # noqa: E402


def main():
    torch.set_num_threads(1)
    O.build()
    H, W, nf = 480, 640, 4096
    bad = 0
    worst = dict(desc=0.0, score=0.0, k1h=0.0)
    print(f"# oracle/xfeat_oracle.c vs oracle/torch_restatement.py (torch {torch.__version__}, 1 thread), {H}x{W}, nfeatures {nf}, lap (0, {W // 3})")
    for fi, family in enumerate(WT.FAMILIES):
        w = WT.make_family(family, 3)
        orc = O.Oracle(WT.pack_blob(w))
        for imf in synth.IMAGE_FAMILIES:
            img = synth.image_family(imf, H, W, 5)
            lap = (0, W // 3)
            kps, desc, nv, mono = orc.extract(img, nf, lap)
            taps = {}
            k2, d2, nv2, mono2 = TR.extract(img, w, nf, lap, taps)
            cand = orc.tensor(O.T["CAND"]).reshape(-1, 3)
            sc = np.sort(cand[:, 2])[::-1] if len(cand) else np.zeros(0)
            cut, gap = (float(sc[nf - 1]), float(sc[nf - 1] - sc[nf])) if len(sc) > nf else (None, float("nan"))
            s1, s2 = kp_set(kps), kp_set(k2)
            dd, ds, n = joined_desc_diff(kps, desc, k2, d2)
            k1h = float(np.abs(orc.tensor(O.T["K1H"]) - taps["K1h"][0].permute(1, 2, 0).numpy().ravel()).max())
            ok = nv == nv2 and len(cand) == int(taps["cand"].shape[1]) and dd < 1e-4
            if s1 != s2:
                ok = ok and cut is not None and gap < 2e-6
                sc1 = {(int(k["x"]), int(k["y"])): float(k["response"]) for k in kps if k["size"] > 0}
                sc2 = {(int(k["x"]), int(k["y"])): float(k["response"]) for k in k2 if k["size"] > 0}
                ok = ok and all(abs(sc1.get(xy, sc2.get(xy)) - cut) < 2e-6 for xy in s1 ^ s2)
            else:
                ok = ok and mono == mono2
            worst = dict(desc=max(worst["desc"], dd), score=max(worst["score"], ds), k1h=max(worst["k1h"], k1h))
            bad += not ok
            print(f"{family:13s} {imf:11s} C={len(cand):6d}/{int(taps['cand'].shape[1]):6d} nv={nv:4d}/{nv2:4d} mono={mono:4d}/{mono2:4d} set_diff={len(s1 ^ s2):4d} "
                  f"desc={dd:.1e} score={ds:.1e} k1h={k1h:.1e} cut_gap={gap:.1e} {'ok' if ok else 'FAIL'}", flush=True)
    print(f"# {len(WT.FAMILIES) * len(synth.IMAGE_FAMILIES)} pairs, {bad} outside the rule; worst descriptor {worst['desc']:.1e}, score {worst['score']:.1e}, K1h {worst['k1h']:.1e}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
