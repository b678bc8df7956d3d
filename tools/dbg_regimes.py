import sys; sys.path.insert(0, ".")
import numpy as np
from xfeatslam_amd import capi, synth, weights as WT
from xfeatslam_amd.extractor import Context
from oracle import oracle as O
blob = WT.pack_blob(WT.make_synthetic(1234, 1.0))
img = synth.image(160, 224, 9)
ctx = Context(nfeatures=512, max_height=160, max_width=224, max_batch=12); ctx.load_weights(blob)
T = capi.T
orc = O.Oracle(blob); orc.extract(img, 512, (0, 0))
ctx.extract_batch(np.stack([img] * 12))
big = {i: (ctx.debug_tensor(T["RAW0"] + i, 11), ctx.debug_tensor(T["STAT0"] + i, 11)) for i in range(23)}
ctx.extract_batch(img[None])
for i in range(8):
    r1, s1 = ctx.debug_tensor(T["RAW0"] + i), ctx.debug_tensor(T["STAT0"] + i)
    ro, so = orc.tensor(O.T["RAW0"] + i), orc.tensor(O.T["STAT0"] + i)
    print(i, "raw B12 vs B1 ndiff", int((big[i][0] != r1).sum()), "stat ndiff", int((big[i][1] != s1).sum()), "| B1 vs oracle raw", int((r1 != ro).sum()), "stat", int((s1 != so).sum()), "| B12 vs oracle raw", int((big[i][0] != ro).sum()), "stat", int((big[i][1] != so).sum()))
    if (big[i][1] != s1).any():
        j = np.nonzero(big[i][1] != s1)[0][:4]; print("   stat idx", j, big[i][1][j], s1[j], so[j])
ctx.extract_batch(np.stack([img] * 12))
for fr in (0, 5, 11):
    r = ctx.debug_tensor(T["RAW0"] + 3, fr).reshape(40, 56, 24); ro = orc.tensor(O.T["RAW0"] + 3).reshape(40, 56, 24)
    d = (r != ro)
    ys, xs, cs = np.nonzero(d)
    print("frame", fr, "ndiff", d.sum(), "rows", sorted(set(ys.tolist()))[:12], "cols", sorted(set(xs.tolist()))[:20], "chans", sorted(set(cs.tolist()))[:8], "example", r[ys[0], xs[0], cs[0]] if len(ys) else None, ro[ys[0], xs[0], cs[0]] if len(ys) else None)
