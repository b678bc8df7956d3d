"""Python mirror of the reference's `ORB_SLAM3::XFextractor` (include/XFextractor.h:32-67,
src/XFextractor.cc:75-356) on top of the C ABI in include/xfeat_hip.h.

Same constructor arguments, same `operator()` behaviour (exactly `nfeatures` output rows,
default keypoints / zero descriptor rows as padding, lapping-area placement, return value =
monoIndex, -1 for an empty image) and the same scale getters.  All compute happens in
libxfeat_hip.so on the GPU; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import KP_DTYPE, Config, check, lib


class Context:
    """Owns one xfh_ctx (one per GPU)."""

    def __init__(self, nfeatures=4096, max_height=480, max_width=640, max_batch=1, device=0, nms_threshold=0.05, bn_mode=0, flags=0):
        cfg = Config()
        lib().xfh_config_default(C.byref(cfg))
        cfg.device, cfg.max_height, cfg.max_width = device, max_height, max_width
        cfg.nfeatures, cfg.max_batch, cfg.nms_threshold = nfeatures, max_batch, nms_threshold
        cfg.bn_mode = bn_mode          # 0 = batch statistics (the reference), 1 = running statistics (upstream eval()), 2 = the same folded into the weights
        cfg.flags = flags              # capi.FLAG_*; 0 = the reference's behaviour
        h = C.c_void_p()
        check(lib().xfh_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.device = device
        self.nfeatures = nfeatures
        self.max_batch = max_batch
        self.rec_bytes = int(lib().xfh_record_bytes(nfeatures))
        self.kps_off = int(lib().xfh_record_kps_offset())
        self.desc_off = int(lib().xfh_record_desc_offset(nfeatures))

    def close(self):
        if getattr(self, "h", None):
            lib().xfh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, blob: bytes):
        check(lib().xfh_load_weights(self.h, blob, len(blob)), self.h)

    def synchronize(self):
        check(lib().xfh_synchronize(self.h), self.h)

    # -- records --------------------------------------------------------------------------
    def parse_records(self, raw: np.ndarray, B: int):
        """raw u8 [B*rec_bytes] -> list of (kps, desc, n_valid, mono_index, n_candidates)"""
        out = []
        nf = self.nfeatures
        for b in range(B):
            r = raw[b * self.rec_bytes:(b + 1) * self.rec_bytes]
            hdr = r[:16].view(np.int32)
            kps = r[self.kps_off:self.kps_off + 28 * nf].view(KP_DTYPE).copy()
            desc = r[self.desc_off:self.desc_off + 256 * nf].view(np.float32).reshape(nf, 64).copy()
            out.append((kps, desc, int(hdr[0]), int(hdr[1]), int(hdr[2])))
        return out

    def extract_batch(self, frames: np.ndarray, lapping=(0, 0)):
        frames = np.ascontiguousarray(frames, np.uint8)
        B, H, W = frames.shape
        raw = np.empty(B * self.rec_bytes, np.uint8)
        check(lib().xfh_extract_batch(self.h, frames.ctypes.data, B, H, W, int(lapping[0]), int(lapping[1]), raw.ctypes.data), self.h)
        return self.parse_records(raw, B)

    def debug_tensor(self, tid: int, frame: int = 0) -> np.ndarray:
        n = C.c_size_t(0)
        check(lib().xfh_debug_tensor(self.h, tid, frame, None, 0, C.byref(n)), self.h)
        out = np.empty(n.value, np.float32)
        if n.value:
            check(lib().xfh_debug_tensor(self.h, tid, frame, out.ctypes.data, n.value, C.byref(n)), self.h)
        return out

    # -- matching -------------------------------------------------------------------------
    def match_mnn(self, d1: np.ndarray, d2: np.ndarray, min_cossim: float = -1.0):
        d1 = np.ascontiguousarray(d1, np.float32); d2 = np.ascontiguousarray(d2, np.float32)
        n = max(1, min(len(d1), len(d2)))
        i1 = np.zeros(n, np.int32); i2 = np.zeros(n, np.int32); dist = np.zeros(n, np.float32)
        nm = C.c_int(0)
        check(lib().xfh_match_mnn(self.h, d1.ctypes.data, len(d1), d2.ctypes.data, len(d2), float(min_cossim),
                                  i1.ctypes.data, i2.ctypes.data, dist.ctypes.data, C.byref(nm)), self.h)
        k = nm.value
        return i1[:k].copy(), i2[:k].copy(), dist[:k].copy()

    def match_prepare(self, desc: np.ndarray):
        """xfh_match_prepare_device: upload n x 64 rows and build their panel image once -> (DeviceBuffer image, n)"""
        d = np.ascontiguousarray(desc, np.float32)
        n = len(d)
        raw = capi.DeviceBuffer(max(d.nbytes, 16)).upload(d)
        img = capi.DeviceBuffer(max(lib().xfh_match_image_bytes(n), 16))
        check(lib().xfh_match_prepare_device(self.h, raw.ptr, n, img.ptr), self.h)
        self.synchronize()
        raw.free()
        return img, n

    def match_mnn_prepared(self, p1, p2, min_cossim: float = -1.0):
        """xfh_match_mnn_prepared_device on two prepared sets (results identical to match_mnn on their rows)"""
        (img1, n1), (img2, n2) = p1, p2
        nm = max(1, min(n1, n2))
        out = capi.DeviceBuffer(nm * 12 + 64)
        check(lib().xfh_match_mnn_prepared_device(self.h, img1.ptr, n1, img2.ptr, n2, float(min_cossim),
                                                  out.ptr + 64, out.ptr + 64 + 4 * nm, out.ptr + 64 + 8 * nm, out.ptr), self.h)
        self.synchronize()
        k = int(out.download(np.int32, 1)[0])
        if k < 0 or k > nm:
            out.free()
            raise capi.XfhError(6, "k_mnn_post: collector timed out (n_matches < 0)")
        i1 = out.download(np.int32, nm, 64)[:k]; i2 = out.download(np.int32, nm, 64 + 4 * nm)[:k]; dist = out.download(np.float32, nm, 64 + 8 * nm)[:k]
        out.free()
        return i1, i2, dist

    @staticmethod
    def pair_tables(pairs):
        """host-side pointer / size arrays of a many-pairs call: pairs = [((img1, n1), (img2, n2)), ...] -> (p1, n1, p2, n2) ctypes arrays"""
        P = len(pairs)
        p1 = (C.c_void_p * P)(*[a[0].ptr for a, _ in pairs]); p2 = (C.c_void_p * P)(*[b[0].ptr for _, b in pairs])
        n1 = (C.c_int * P)(*[a[1] for a, _ in pairs]); n2 = (C.c_int * P)(*[b[1] for _, b in pairs])
        return p1, n1, p2, n2

    def match_mnn_prepared_batch(self, pairs, min_cossim: float = -1.0):
        """xfh_match_mnn_prepared_batch_device: ORBmatcher::match of every (prepared set, prepared set) pair in one persistent GEMM launch +
        one post launch -> [(idx1, idx2, dist), ...], pair by pair what match_mnn_prepared returns"""
        P = len(pairs)
        if P == 0:
            return []
        nm = [max(1, min(a[1], b[1])) for a, b in pairs]
        off = np.concatenate([[0], np.cumsum([(12 * k + 63) // 64 * 64 for k in nm])]).astype(np.int64)
        out = capi.DeviceBuffer(int(off[-1]) + 64); cnt = capi.DeviceBuffer(4 * P + 64)
        p1, n1, p2, n2 = self.pair_tables(pairs)
        i1 = (C.c_void_p * P)(*[out.ptr + int(off[p]) for p in range(P)])
        i2 = (C.c_void_p * P)(*[out.ptr + int(off[p]) + 4 * nm[p] for p in range(P)])
        ds = (C.c_void_p * P)(*[out.ptr + int(off[p]) + 8 * nm[p] for p in range(P)])
        check(lib().xfh_match_mnn_prepared_batch_device(self.h, P, p1, n1, p2, n2, float(min_cossim), i1, i2, ds, cnt.ptr), self.h)
        self.synchronize()
        ks = cnt.download(np.int32, P)
        res = []
        for p in range(P):
            k = int(ks[p])
            if k < 0 or k > nm[p]:
                out.free(); cnt.free()
                raise capi.XfhError(6, "k_mnn_post_batch: collector timed out (n_matches < 0)")
            res.append((out.download(np.int32, nm[p], int(off[p]))[:k], out.download(np.int32, nm[p], int(off[p]) + 4 * nm[p])[:k],
                        out.download(np.float32, nm[p], int(off[p]) + 8 * nm[p])[:k]))
        out.free(); cnt.free()
        return res

    def distance_i32(self, d1: np.ndarray, d2: np.ndarray) -> np.ndarray:
        d1 = np.ascontiguousarray(d1, np.float32); d2 = np.ascontiguousarray(d2, np.float32)
        out = np.zeros((len(d1), len(d2)), np.int32)
        check(lib().xfh_distance_i32(self.h, d1.ctypes.data, len(d1), d2.ctypes.data, len(d2), out.ctypes.data), self.h)
        return out

    def best2_csr(self, queries, targets, offsets, indices, init_dist: int = 256):
        """guided matching primitive: best / second-best (int)(512*d^2) over per-query candidate lists"""
        q = np.ascontiguousarray(queries, np.float32); tg = np.ascontiguousarray(targets, np.float32)
        off = np.ascontiguousarray(offsets, np.int32); ind = np.ascontiguousarray(indices, np.int32)
        nq = len(q)
        out = [np.zeros(max(nq, 1), np.int32) for _ in range(4)]
        check(lib().xfh_best2_csr(self.h, q.ctypes.data, nq, tg.ctypes.data, len(tg), off.ctypes.data, ind.ctypes.data, int(init_dist),
                                  *[o.ctypes.data for o in out]), self.h)
        return tuple(o[:nq] for o in out)

    def distinctive_csr(self, table, offsets, indices):
        """MapPoint::ComputeDistinctiveDescriptors over CSR groups of descriptor rows ->
        (position inside the group of the descriptor with the least median distance, that median)"""
        tb = np.ascontiguousarray(table, np.float32)
        off = np.ascontiguousarray(offsets, np.int32); ind = np.ascontiguousarray(indices, np.int32)
        ng = len(off) - 1
        pos = np.zeros(max(ng, 1), np.int32); med = np.zeros(max(ng, 1), np.int32)
        check(lib().xfh_distinctive_csr(self.h, tb.ctypes.data, len(tb), off.ctypes.data, ind.ctypes.data, ng, pos.ctypes.data, med.ctypes.data), self.h)
        return pos[:ng], med[:ng]

    # -- timing ---------------------------------------------------------------------------
    def timing_enable(self, kernel_id: int, layer_mask: int = 0):
        check(lib().xfh_timing_enable(self.h, kernel_id, layer_mask), self.h)

    def timing_read(self):
        n = C.c_int(0); ms = C.c_double(0.0)
        check(lib().xfh_timing_read(self.h, C.byref(n), C.byref(ms)), self.h)
        return n.value, ms.value


def scale_tables(nlevels: int, scaleFactor: float):
    """mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2 (XFextractor.cc:80-96),
    fp32 arithmetic as in the reference"""
    sf = np.ones(nlevels, np.float32); s2 = np.ones(nlevels, np.float32)
    for i in range(1, nlevels):
        sf[i] = np.float32(sf[i - 1] * np.float32(scaleFactor))
        s2[i] = np.float32(sf[i] * sf[i])
    return sf, (np.float32(1.0) / sf).astype(np.float32), s2, (np.float32(1.0) / s2).astype(np.float32)


class XFextractor:
    """Drop-in for `ORB_SLAM3::XFextractor` (reference include/XFextractor.h:32-67)."""

    def __init__(self, nfeatures: int, scaleFactor: float, nlevels: int, iniThFAST: int, minThFAST: int,
                 weights: bytes | None = None, max_height: int = 480, max_width: int = 640, device: int = 0):
        self.nfeatures, self.scaleFactor, self.nlevels = nfeatures, float(np.float32(scaleFactor)), nlevels
        self.iniThFAST, self.minThFAST = iniThFAST, minThFAST
        self.mvScaleFactor, self.mvInvScaleFactor, self.mvLevelSigma2, self.mvInvLevelSigma2 = scale_tables(nlevels, scaleFactor)
        self.mvImagePyramid = [None] * nlevels          # sized, never filled (XFextractor.cc:98)
        self.ctx = Context(nfeatures, max_height, max_width, 1, device)
        if weights is not None:
            self.ctx.load_weights(weights)

    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        """returns (ret, keypoints[nfeatures], descriptors[nfeatures,64] or None); ret = -1 for
        an empty image, else monoIndex (XFextractor.cc:250-356)."""
        if image is None or getattr(image, "size", 0) == 0:
            return -1, None, None                        # :253-254
        image = np.asarray(image)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise ValueError("image must be CV_8UC1")     # assert at :257
        H, W = image.shape
        img = np.ascontiguousarray(image)
        kps = np.zeros(self.nfeatures, KP_DTYPE)
        desc = np.zeros((self.nfeatures, 64), np.float32)
        nv, mono = C.c_int(0), C.c_int(0)
        check(lib().xfh_extract(self.ctx.h, img.ctypes.data, H, W, W, int(vLappingArea[0]), int(vLappingArea[1]),
                                kps.ctypes.data, desc.ctypes.data, C.byref(nv), C.byref(mono)), self.ctx.h)
        self.n_valid = nv.value
        if nv.value == 0:
            return mono.value, kps, None                 # _descriptors.release() (:350-353)
        return mono.value, kps, desc

    detectAndCompute = __call__

    def GetLevels(self): return self.nlevels
    def GetScaleFactor(self): return self.scaleFactor
    def GetScaleFactors(self): return self.mvScaleFactor
    def GetInverseScaleFactors(self): return self.mvInvScaleFactor
    def GetScaleSigmaSquares(self): return self.mvLevelSigma2
    def GetInverseScaleSigmaSquares(self): return self.mvInvLevelSigma2


class ORBmatcher:
    """The XFeat half of `ORB_SLAM3::ORBmatcher` (reference include/ORBmatcher.h:43,77)."""
    TH_HIGH = 1000      # ORBmatcher.cc:34 (USE_ORB unset)
    TH_LOW = 100        # ORBmatcher.cc:35

    def __init__(self, nnratio: float = 0.6, checkOri: bool = True, ctx: Context | None = None):
        self.mfNNratio, self.mbCheckOrientation = nnratio, checkOri
        self.ctx = ctx or Context(nfeatures=1, max_height=32, max_width=32)

    @staticmethod
    def DescriptorDistance(a: np.ndarray, b: np.ndarray) -> int:
        a = np.ascontiguousarray(a, np.float32).ravel(); b = np.ascontiguousarray(b, np.float32).ravel()
        return int(lib().xfh_descriptor_distance(a.ctypes.data, b.ctypes.data))

    def match(self, desc1: np.ndarray, desc2: np.ndarray, min_cossim: float = -1.0):
        """-> list of (queryIdx, trainIdx, distance) like std::vector<cv::DMatch>"""
        i1, i2, d = self.ctx.match_mnn(desc1, desc2, min_cossim)
        return list(zip(i1.tolist(), i2.tolist(), d.tolist()))
