"""Development checker: the parity campaign on the CPU -- C oracle against the ATen-operator restatement over weight families x image
families (the distilled version is tests/test_oracle.py::test_campaign_oracle_vs_aten).  python tests/tools/campaign_cpu.py [H W]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import joined_desc_diff, kp_set          # noqa: E402
from oracle import oracle as O, torch_restatement as TR  # noqa: E402
from xfeatslam_amd import synth, weights as WT          # noqa: E402


def audit(orc, nf):
    """near-tie audit of the oracle's own run: gap at the top-k cut, |mean|/sigma of the worst channel, dead channels"""
    cand = orc.tensor(O.T["CAND"]).reshape(-1, 3)
    sc = np.sort(cand[:, 2])[::-1] if len(cand) else np.zeros(0)
    gap = float(sc[nf - 1] - sc[nf]) if len(sc) > nf else float("nan")
    ties = int((np.diff(sc[sc > 0]) == 0).sum()) if len(sc) else 0
    worst, dead = 0.0, 0
    for i in range(O.NUM_LAYERS):
        st = orc.tensor(O.T["STAT0"] + i)
        C = len(st) // 2
        beta, alpha = st[:C], st[C:]
        dead += int((alpha > 316.0).sum())
        worst = max(worst, float(np.abs(beta).max()))
    return dict(C=len(cand), gap=gap, ties=ties, mean_over_sigma=worst, dead=dead)


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (160, 224)
    nf = 512
    import torch
    torch.set_num_threads(1)
    for wf in WT.FAMILIES:
        for imf in synth.IMAGE_FAMILIES:
            w = WT.make_family(wf, 3)
            img = synth.image_family(imf, H, W, 5)
            orc = O.Oracle(WT.pack_blob(w))
            t0 = time.time()
            kps, desc, nv, mono = orc.extract(img, nf, (0, W // 3))
            taps = {}
            k2, d2, nv2, mono2 = TR.extract(img, w, nf, (0, W // 3), taps)
            a = audit(orc, nf)
            s1, s2 = kp_set(kps), kp_set(k2)
            dd, ds, n = joined_desc_diff(kps, desc, k2, d2)
            k1h = np.abs(orc.tensor(O.T["K1H"]) - taps["K1h"][0].permute(1, 2, 0).numpy().ravel()).max()
            print(f"{wf:14s} {imf:12s} C={a['C']:6d} nv={nv:4d}/{nv2:4d} set_diff={len(s1 ^ s2):3d} desc={dd:.1e} score={ds:.1e} K1h={k1h:.1e} "
                  f"cut_gap={a['gap']:.1e} ties={a['ties']:4d} |mu|/sig={a['mean_over_sigma']:.1f} dead={a['dead']:3d}  {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    main()
