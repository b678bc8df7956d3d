#!/usr/bin/env python
"""weights/xfeat.pt (libtorch archive = TorchScript zip, or an upstream-XFeat state_dict saved
with torch.save) -> flat blob for xfh_load_weights_file (format: xfeatslam_amd/weights.py).
Only the 31 tensors XFeatModel::forward touches are kept; BatchNorm buffers
are irrelevant because the reference normalises with batch statistics."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xfeatslam_amd import weights as WT


def load_any(path):
    import torch
    try:
        m = torch.jit.load(path, map_location="cpu")
        return {k: v for k, v in m.state_dict().items()}
    except Exception:
        sd = torch.load(path, map_location="cpu")
        return sd.get("state_dict", sd) if isinstance(sd, dict) else sd.state_dict()


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit("usage: convert_weights.py xfeat.pt out.xfhw")
    blob = WT.pack_blob(WT.from_state_dict(load_any(sys.argv[1])))
    open(sys.argv[2], "wb").write(blob)
    print(f"wrote {sys.argv[2]}: {len(blob)} bytes, {WT.N_PARAMS} parameters")
