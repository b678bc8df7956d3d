"""ATen-operator restatement of the reference's extraction and matching path.

TEST INFRASTRUCTURE ONLY (see oracle/xfeat_oracle.h).  This file calls, statement by
statement, the same libtorch CPU operators the reference calls through the C++ API
(`torch::nn::functional::*` == `torch.nn.functional.*`, same ATen kernels), so it is the
closest thing to "running the reference" this environment allows: the reference's own
sources cannot be compiled here (they include OpenCV headers the image lacks) and it
ships no tests or golden vectors.  It pins oracle/xfeat_oracle.c (tests/
test_oracle_vs_torch.py) and generates tests/golden/* (tests/golden/make_golden.py).

Line references: /root/reference/src/XFeat.cc, src/XFextractor.cc, src/ORBmatcher.cc.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

_BLOCKS = {
    "block1": [(3, 1), (3, 2), (3, 1), (3, 2)],
    "block2": [(3, 1), (3, 1)],
    "block3": [(3, 2), (3, 1), (1, 1)],
    "block4": [(3, 2), (3, 1), (3, 1)],
    "block5": [(3, 2), (3, 1), (3, 1), (1, 1)],
}


def _t(w):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}


def _basic(x, w, ks, stride, taps, name, bn=None):
    # BasicLayerImpl (XFeat.cc:7-28): Conv2d(bias=False) -> BatchNorm2d(affine=False) -> ReLU.
    # The module is never put in eval() => training=True, batch statistics.
    x = F.conv2d(x, w, None, stride=stride, padding=ks // 2, dilation=1)
    if taps is not None:
        taps[name] = x
    if bn is None:
        x = F.batch_norm(x, None, None, None, None, True, 0.1, 1e-5)
    else:       # eval() semantics of upstream XFeat (not what the reference does): running statistics
        x = F.batch_norm(x, bn[name + ".layer.1.running_mean"], bn[name + ".layer.1.running_var"], None, None, False, 0.1, 1e-5)
    return F.relu(x)


def _seq(x, w, block, taps, bn=None):
    for i, (ks, st) in enumerate(_BLOCKS[block]):
        x = _basic(x, w[f"{block}.{i}.layer.0.weight"], ks, st, taps, f"{block}.{i}", bn)
    return x


def model_forward(x, w, taps=None, bn=None):
    """XFeatModel::forward (XFeat.cc:135-173). x: [1,1,H,W] float."""
    with torch.no_grad():
        x = x.mean(1, True)                                                      # :148
        x = F.instance_norm(x, None, None, None, None, True, 0.1, 1e-5)          # :149
        if taps is not None:
            taps["xhat"] = x
        x1 = _seq(x, w, "block1", taps, bn)                                          # :152
        skip = F.conv2d(F.avg_pool2d(x, 4, 4), w["skip1.1.weight"], w["skip1.1.bias"])   # :36-39
        x2 = _seq(x1 + skip, w, "block2", taps, bn)                                  # :153
        x3 = _seq(x2, w, "block3", taps, bn)                                         # :154
        x4 = _seq(x3, w, "block4", taps, bn)                                         # :155
        x5 = _seq(x4, w, "block5", taps, bn)                                         # :156
        size = [x3.size(2), x3.size(3)]
        x4 = F.interpolate(x4, size=size, mode="bilinear", align_corners=False)  # :159-161
        x5 = F.interpolate(x5, size=size, mode="bilinear", align_corners=False)  # :162-164
        f = x3 + x4 + x5
        if taps is not None:
            taps["fuse_in"] = f
        f = _basic(f, w["block_fusion.0.layer.0.weight"], 3, 1, taps, "block_fusion.0", bn)
        f = _basic(f, w["block_fusion.1.layer.0.weight"], 3, 1, taps, "block_fusion.1", bn)
        feats = F.conv2d(f, w["block_fusion.2.weight"], w["block_fusion.2.bias"])  # :166
        h = _basic(feats, w["heatmap_head.0.layer.0.weight"], 1, 1, taps, "heatmap_head.0", bn)
        h = _basic(h, w["heatmap_head.1.layer.0.weight"], 1, 1, taps, "heatmap_head.1", bn)
        heatmap = torch.sigmoid(F.conv2d(h, w["heatmap_head.2.weight"], w["heatmap_head.2.bias"]))  # :169
        # unfold2d (:124-133)
        B, Cc, H, W = x.shape
        ws = 8
        u = x.unfold(2, ws, ws).unfold(3, ws, ws).reshape(B, Cc, H // ws, W // ws, ws * ws)
        u = u.permute(0, 1, 4, 2, 3).reshape(B, -1, H // ws, W // ws)
        k = _basic(u, w["keypoint_head.0.layer.0.weight"], 1, 1, taps, "keypoint_head.0", bn)
        k = _basic(k, w["keypoint_head.1.layer.0.weight"], 1, 1, taps, "keypoint_head.1", bn)
        k = _basic(k, w["keypoint_head.2.layer.0.weight"], 1, 1, taps, "keypoint_head.2", bn)
        keypoints = F.conv2d(k, w["keypoint_head.3.weight"], w["keypoint_head.3.bias"])  # :170
    return feats, keypoints, heatmap


def _normgrid(pos, H, W):
    # InterpolateSparse2d::normgrid (XFeat.cc:181-186); pos is Long -> true division
    size = torch.tensor([W - 1, H - 1], dtype=pos.dtype)
    return 2.0 * (pos / size) - 1.0


def _interp(x, pos, H, W, mode):
    # InterpolateSparse2d::forward (XFeat.cc:188-210)
    grid = _normgrid(pos, H, W).unsqueeze(-2).to(x.dtype)
    x = F.grid_sample(x, grid, mode=mode, align_corners=False)
    return x.permute(0, 2, 3, 1).squeeze(-2)


def extract(gray: np.ndarray, weights, nfeatures: int = 4096, lapping=(0, 0), taps=None, eval_mode: bool = False, rescale: bool = False):
    """XFextractor::operator() (XFextractor.cc:250-356).  Returns
    (kps[nfeatures] structured, desc[nfeatures,64], n_valid, mono_index)."""
    from .oracle import KP_DTYPE
    w = _t(weights)
    with torch.no_grad():
        img = torch.from_numpy(np.ascontiguousarray(gray, np.uint8))
        Hh, Ww = img.shape
        x = img.reshape(1, Hh, Ww, 1).permute(0, 3, 1, 2).to(torch.float) / 255.0   # parseInput :166-167
        x = x.to(torch.float)                                                        # :185
        _H, _W = (Hh // 32) * 32, (Ww // 32) * 32
        rh, rw = Hh / _H, Ww / _W
        x = F.interpolate(x, size=[_H, _W], mode="bilinear", align_corners=False)   # :198-200
        if taps is not None:
            taps["x"] = x
        M1, K1, H1 = model_forward(x, w, taps, w if eval_mode else None)             # :268
        M1 = F.normalize(M1, dim=1)                                                  # :273
        # getKptsHeatmap (:204-217)
        scores = F.softmax(K1 * 1.0, 1)[:, :64]
        B, _, h, ww = scores.shape
        heat = scores.permute(0, 2, 3, 1).reshape(B, h, ww, 8, 8)
        K1h = heat.permute(0, 1, 3, 2, 4).reshape(B, 1, h * 8, ww * 8)
        # NMS (:219-248)
        local_max = F.max_pool2d(K1h, 5, stride=1, padding=2)
        pos = (K1h == local_max) & (K1h > 0.05)
        mk = pos[0].nonzero()[..., 1:].flip(-1)
        mkpts = torch.zeros(1, mk.size(0), 2, dtype=torch.long)
        if mk.size(0) > 0:
            mkpts[0, :mk.size(0)] = mk
        # scores (:280-282)
        sc = (_interp(K1h, mkpts, _H, _W, "nearest") * _interp(H1, mkpts, _H, _W, "bilinear")).squeeze(-1)
        mask = torch.all(mkpts == 0, -1)
        sc.masked_fill_(mask, -1)
        if taps is not None:
            taps.update(M1n=M1, K1=K1, H1=H1, K1h=K1h, cand=mkpts.clone(), cand_scores=sc.clone())
        # top-k (:285-295)
        idxs = sc.neg().argsort(-1, False)
        mx = mkpts[..., 0].gather(-1, idxs)[:, :nfeatures]
        my = mkpts[..., 1].gather(-1, idxs)[:, :nfeatures]
        mkpts = torch.cat([mx.unsqueeze(-1), my.unsqueeze(-1)], -1)
        sc = sc.gather(-1, idxs)[:, :nfeatures]
        feats = _interp(M1, mkpts, _H, _W, "bilinear")                               # :298
        feats = F.normalize(feats, dim=-1)                                           # :301
        if rescale:     # not the reference: upstream XFeat multiplies in floating point
            mkpts = mkpts * torch.tensor([rw, rh]).view(1, 1, -1)
        else:
            sf = torch.tensor([rw, rh], dtype=mkpts.dtype).view(1, 1, -1)            # :304 (Long!)
            mkpts = mkpts * sf
        if taps is not None:
            taps.update(sel=mkpts.clone(), sel_scores=sc.clone())
        # pack (:310-356)
        kps = np.zeros(nfeatures, KP_DTYPE)
        kps["angle"] = -1.0
        kps["class_id"] = -1
        desc = np.zeros((nfeatures, 64), np.float32)
        valid = sc[0] > 0
        vk = mkpts[0][valid]; vs = sc[0][valid]; vd = feats[0][valid]
        mono, stereo = 0, nfeatures - 1
        for i in range(vk.size(0)):
            xx = float(vk[i][0]); yy = float(vk[i][1]); s = float(vs[i])
            if xx >= lapping[0] and xx <= lapping[1]:
                slot = stereo; stereo -= 1
            else:
                slot = mono; mono += 1
            kps[slot] = (xx, yy, 1.0, -1.0, s, 0, -1)
            desc[slot] = vd[i].numpy()
        return kps, desc, int(vk.size(0)), mono


def match_mnn(d1: np.ndarray, d2: np.ndarray, min_cossim: float = -1.0):
    """ORBmatcher::match, the commented-out definition (ORBmatcher.cc:340-405) with the
    descriptors read as float."""
    with torch.no_grad():
        f1 = F.normalize(torch.from_numpy(np.ascontiguousarray(d1, np.float32)), dim=-1)   # :358
        f2 = F.normalize(torch.from_numpy(np.ascontiguousarray(d2, np.float32)), dim=-1)   # :359
        cossim = torch.matmul(f1, f2.t())                                                   # :363
        cossim_t = torch.matmul(f2, f1.t())                                                 # :364
        _, m12 = cossim.max(1)                                                              # :367
        _, m21 = cossim_t.max(1)                                                            # :368
        idx0 = torch.arange(m12.size(0))
        mutual = m21[m12] == idx0                                                           # :372
        if min_cossim > 0:
            best, _ = cossim.max(1)
            good = best > min_cossim
            keep = mutual & good
        else:
            keep = mutual
        i0 = idx0[keep]; i1 = m12[keep]
        cd = 1.0 - cossim[i0, i1]
        dist = torch.sqrt(2 * cd)                                                           # :398-399
    return i0.numpy().astype(np.int32), i1.numpy().astype(np.int32), dist.numpy(), cossim.numpy()


def distance_i32(d1: np.ndarray, d2: np.ndarray) -> np.ndarray:
    """ORBmatcher::DescriptorDistance (ORBmatcher.cc:2246-2247), dense: fp32 difference,
    fp64 sum of squares (OpenCV 4.5.4 normDiffL2Sqr_<float,double>), fp32 * 512, trunc."""
    a = np.ascontiguousarray(d1, np.float32)[:, None, :]
    b = np.ascontiguousarray(d2, np.float32)[None, :, :]
    diff = (a - b).astype(np.float32)
    s = (diff.astype(np.float64) ** 2).sum(-1)
    return (s.astype(np.float32) * np.float32(512)).astype(np.int32)
