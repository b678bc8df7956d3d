"""Canonical XFeat weight set for the hot path: tensor table, deterministic synthetic
generator, flat blob (de)serialiser and a state_dict converter.

The 31 tensors are the ones `XFeatModel::forward` touches (reference
`src/XFeat.cc:30-122`, names as libtorch registers them).  The
886 848-float `fine_matcher` MLP (`XFeat.cc:94-108`) and every BatchNorm buffer are
not part of the path (BatchNorm runs on batch statistics) and are not
stored.

Blob layout (little endian), consumed by both `oracle/xfeat_oracle.c` and
`xfeatslam_amd/csrc/weights.cpp`:

    char     magic[8]   = "XFHWGT01"
    uint32   n_tensors
    uint32   reserved
    entry[n_tensors]:  char name[48]; uint32 ndim; uint32 dims[4]; uint64 offset_floats
    float32  data[...]            (tensors in PyTorch OIHW order, C-contiguous)
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

MAGIC = b"XFHWGT01"

# (name, shape) in canonical order -- OIHW for conv weights.
TENSORS = [
    ("skip1.1.weight", (24, 1, 1, 1)),
    ("skip1.1.bias", (24,)),
    ("block1.0.layer.0.weight", (4, 1, 3, 3)),
    ("block1.1.layer.0.weight", (8, 4, 3, 3)),
    ("block1.2.layer.0.weight", (8, 8, 3, 3)),
    ("block1.3.layer.0.weight", (24, 8, 3, 3)),
    ("block2.0.layer.0.weight", (24, 24, 3, 3)),
    ("block2.1.layer.0.weight", (24, 24, 3, 3)),
    ("block3.0.layer.0.weight", (64, 24, 3, 3)),
    ("block3.1.layer.0.weight", (64, 64, 3, 3)),
    ("block3.2.layer.0.weight", (64, 64, 1, 1)),
    ("block4.0.layer.0.weight", (64, 64, 3, 3)),
    ("block4.1.layer.0.weight", (64, 64, 3, 3)),
    ("block4.2.layer.0.weight", (64, 64, 3, 3)),
    ("block5.0.layer.0.weight", (128, 64, 3, 3)),
    ("block5.1.layer.0.weight", (128, 128, 3, 3)),
    ("block5.2.layer.0.weight", (128, 128, 3, 3)),
    ("block5.3.layer.0.weight", (64, 128, 1, 1)),
    ("block_fusion.0.layer.0.weight", (64, 64, 3, 3)),
    ("block_fusion.1.layer.0.weight", (64, 64, 3, 3)),
    ("block_fusion.2.weight", (64, 64, 1, 1)),
    ("block_fusion.2.bias", (64,)),
    ("heatmap_head.0.layer.0.weight", (64, 64, 1, 1)),
    ("heatmap_head.1.layer.0.weight", (64, 64, 1, 1)),
    ("heatmap_head.2.weight", (1, 64, 1, 1)),
    ("heatmap_head.2.bias", (1,)),
    ("keypoint_head.0.layer.0.weight", (64, 64, 1, 1)),
    ("keypoint_head.1.layer.0.weight", (64, 64, 1, 1)),
    ("keypoint_head.2.layer.0.weight", (64, 64, 1, 1)),
    ("keypoint_head.3.weight", (65, 64, 1, 1)),
    ("keypoint_head.3.bias", (65,)),
]
N_PARAMS = sum(int(np.prod(s)) for _, s in TENSORS)  # 657 910

# Optional tensors: BatchNorm running statistics of the 23 BasicLayers.  They do not influence the
# reference's output (it normalises with batch statistics); they are only needed for
# the XFH_BN_RUNNING_STATS mode (upstream-XFeat eval() semantics).
BN_LAYERS = [n[:-len(".layer.0.weight")] for n, _ in TENSORS if n.endswith(".layer.0.weight")]
BN_TENSORS = [(f"{l}.layer.1.{k}", (dict(TENSORS)[f"{l}.layer.0.weight"][0],)) for l in BN_LAYERS for k in ("running_mean", "running_var")]

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Counter-based 64-bit mixer (Steele/Lea/Flood splitmix64 finaliser), vectorised."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, stream: int, n: int) -> np.ndarray:
    """n float64 values in [0,1), reproducible from (seed, stream) alone."""
    base = splitmix64(np.array([seed], dtype=np.uint64))[0] ^ splitmix64(
        np.array([stream + 0x51ED], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) + base) & _M64
    bits = splitmix64(ctr) >> np.uint64(11)
    return bits.astype(np.float64) * (1.0 / 9007199254740992.0)


def make_synthetic(seed: int = 1234, kp_logit_gain: float = 1.0, with_bn: bool = False) -> "OrderedDict[str, np.ndarray]":
    """Deterministic stand-in for the absent `weights/xfeat.pt`.

    conv weights/biases ~ U(-1/sqrt(fan_in), +1/sqrt(fan_in)) (the libtorch Conv2d default
    bound).  `kp_logit_gain` scales `keypoint_head.3.weight`; random-weight nets give a
    nearly flat 65-way softmax (few NMS candidates), gain≈6 produces a "dense" frame
    with more than 4096 candidates at VGA so the top-k cut is exercised.
    """
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for i, (name, shape) in enumerate(TENSORS):
        if name.endswith("bias"):
            wshape = dict(TENSORS)[name[:-4] + "weight"]
            fan_in = int(np.prod(wshape[1:]))
        else:
            fan_in = int(np.prod(shape[1:]))
        bound = 1.0 / np.sqrt(fan_in)
        u = uniform01(seed, i, int(np.prod(shape)))
        w = ((2.0 * u - 1.0) * bound).astype(np.float32).reshape(shape)
        if name == "keypoint_head.3.weight":
            w = (w * np.float32(kp_logit_gain)).astype(np.float32)
        out[name] = w
    if with_bn:
        # plausible running statistics of a post-conv activation: mean ~ U(-0.5, 0.5), var ~ U(0.5, 1.5)
        for j, (name, shape) in enumerate(BN_TENSORS):
            u = uniform01(seed, 1000 + j, shape[0])
            out[name] = ((u - 0.5) if name.endswith("mean") else (0.5 + u)).astype(np.float32)
    return out


# ---- weight families of the parity campaign (tests/test_oracle.py, tests/test_gpu_campaign.py) -------------------------------
# The trained `weights/xfeat.pt` is absent, so breadth has to stand in for it: every family below is a deterministic function of
# (name, seed) and stresses one thing a trained net can do and U(-b, b) draws never do.
FAMILIES = ("uniform", "normal", "heavy", "scaled", "tiny", "pruned", "dc", "heat_on", "heat_off", "heat_denormal", "peaky")


def _normal01(seed: int, stream: int, n: int) -> np.ndarray:
    u1 = uniform01(seed, 2 * stream + 5000, n)
    u2 = uniform01(seed, 2 * stream + 5001, n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def make_family(family: str, seed: int = 1, kp_logit_gain: float = 3.0) -> "OrderedDict[str, np.ndarray]":
    """One member of a weight family (same tensor table as `make_synthetic`).

    uniform        the `make_synthetic` draw with another seed
    normal         N(0, 1/fan_in)
    heavy          Student-t, 3 degrees of freedom, scaled to the same variance (a few weights 10-30 x the rest)
    scaled         BasicLayer i multiplied by 10 (i even) or 0.1 (i odd): raw maps of very different magnitude feed BatchNorm
    tiny           every third BasicLayer x 3e-3: its channel variances (~3e-6) fall below eps = 1e-5, rstd saturates towards 316
    pruned         a quarter of the output filters of every BasicLayer are all-zero (dead channels: raw map = 0, variance 0), and so
                   is one input channel of every layer that follows
    dc             a quarter of the output filters of every BasicLayer behind block1.0 are `|w| + 4 b`: all-positive filters on
                   non-negative inputs, the largest |mean| / sigma this architecture can produce (BasicLayers carry no bias)
    heat_on/off    heatmap_head.2.bias = +30 / -110: sigmoid saturates to exactly 1 / exactly 0 (no valid keypoint at all)
    heat_denormal  heatmap_head.2.bias = -87: reliabilities around 1e-38, i.e. on both sides of the smallest normal fp32 number; the
                   scores (x K1h <= 1) are fp32 denormals but > 0, so every candidate stays valid
    peaky          keypoint_head.3 x 12: a near one-hot 65-way softmax, exp() arguments down to -100
    """
    if family not in FAMILIES:
        raise ValueError(family)
    shapes = dict(TENSORS)
    w = make_synthetic(seed, kp_logit_gain)
    if family in ("normal", "heavy"):
        for i, (name, shape) in enumerate(TENSORS):
            fan_in = int(np.prod((shapes[name[:-4] + "weight"] if name.endswith("bias") else shape)[1:]))
            n = int(np.prod(shape))
            z = _normal01(seed, i, n)
            if family == "heavy":
                chi = sum(_normal01(seed, 100 + 3 * i + k, n) ** 2 for k in range(3)) / 3.0
                z = z / np.sqrt(chi) / np.sqrt(3.0)                       # t3 has variance 3
            a = (z / np.sqrt(fan_in)).astype(np.float32).reshape(shape)
            if name == "keypoint_head.3.weight":
                a = (a * np.float32(kp_logit_gain)).astype(np.float32)
            w[name] = a
    basic = [n for n, _ in TENSORS if n.endswith(".layer.0.weight")]
    if family == "scaled":
        for i, n in enumerate(basic):
            w[n] = (w[n] * np.float32(10.0 if i % 2 == 0 else 0.1)).astype(np.float32)
    elif family == "tiny":
        for n in basic[1::3]:
            w[n] = (w[n] * np.float32(3e-3)).astype(np.float32)
    elif family == "pruned":
        for i, n in enumerate(basic):
            co, ci = shapes[n][0], shapes[n][1]
            dead = uniform01(seed, 900 + i, co) < 0.25
            dead[0] = True
            w[n][dead] = 0.0
            if ci > 1:
                w[n][:, int(uniform01(seed, 950 + i, 1)[0] * ci)] = 0.0
    elif family == "dc":
        for i, n in enumerate(basic):
            if n == "block1.0.layer.0.weight" or n.startswith("keypoint_head.0"):       # inputs of mean 0: a DC filter gives no mean there
                continue
            co = shapes[n][0]
            b = 1.0 / np.sqrt(float(np.prod(shapes[n][1:])))
            pick = uniform01(seed, 800 + i, co) < 0.25
            pick[co - 1] = True
            w[n][pick] = (np.abs(w[n][pick]) + np.float32(4.0 * b)).astype(np.float32)
    elif family == "heat_on":
        w["heatmap_head.2.bias"][:] = 30.0
    elif family == "heat_off":
        w["heatmap_head.2.bias"][:] = -110.0
    elif family == "heat_denormal":
        w["heatmap_head.2.bias"][:] = -87.0
    elif family == "peaky":
        w["keypoint_head.3.weight"] = (w["keypoint_head.3.weight"] * np.float32(12.0 / kp_logit_gain)).astype(np.float32)
    return w


def pack_blob(weights: "dict[str, np.ndarray]") -> bytes:
    entries = []
    data = []
    off = 0
    table = list(TENSORS)
    if all(n in weights for n, _ in BN_TENSORS):
        table += BN_TENSORS
    for name, shape in table:
        a = np.ascontiguousarray(weights[name], dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {a.shape} != {shape}")
        dims = list(shape) + [1] * (4 - len(shape))
        entries.append(struct.pack("<48sI4IQ", name.encode(), len(shape), *dims, off))
        data.append(a.tobytes())
        off += a.size
    head = MAGIC + struct.pack("<II", len(table), 0)
    return head + b"".join(entries) + b"".join(data)


def unpack_blob(blob: bytes) -> "OrderedDict[str, np.ndarray]":
    if blob[:8] != MAGIC:
        raise ValueError("bad magic")
    n, _ = struct.unpack_from("<II", blob, 8)
    esz = struct.calcsize("<48sI4IQ")
    base = 16 + n * esz
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for i in range(n):
        name, ndim, d0, d1, d2, d3, off = struct.unpack_from("<48sI4IQ", blob, 16 + i * esz)
        shape = (d0, d1, d2, d3)[:ndim]
        cnt = int(np.prod(shape))
        out[name.rstrip(b"\0").decode()] = np.frombuffer(
            blob, dtype="<f4", count=cnt, offset=base + 4 * off).reshape(shape).copy()
    return out


def from_state_dict(sd) -> "OrderedDict[str, np.ndarray]":
    """Pick the path's tensors out of an upstream-XFeat / libtorch-archive state_dict
    (any mapping name -> array-like; keys may carry a `net.` prefix).  Used by
    tools/convert_weights.py once a real xfeat.pt is available."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in TENSORS:
        src = None
        for k in (name, "net." + name):
            if k in sd:
                src = sd[k]
                break
        if src is None:
            raise KeyError(name)
        a = np.asarray(src.detach().cpu().numpy() if hasattr(src, "detach") else src, dtype=np.float32)
        out[name] = a.reshape(shape)
    for name, shape in BN_TENSORS:                      # optional: only if every one is present
        for k in (name, "net." + name):
            if k in sd:
                src = sd[k]
                out[name] = np.asarray(src.detach().cpu().numpy() if hasattr(src, "detach") else src, dtype=np.float32).reshape(shape)
                break
    if not all(n in out for n, _ in BN_TENSORS):
        for n, _ in BN_TENSORS:
            out.pop(n, None)
    return out
