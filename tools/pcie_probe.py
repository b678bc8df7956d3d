"""Raw PCIe copy bandwidth of the box (hipMemcpyAsync between pinned host memory and HBM), 1-4 streams per direction, one and both
directions at once: the ceiling of the host-visible batch pipeline (csrc/pipeline.cpp).  usage: python tools/pcie_probe.py"""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p


def ck(e):
    assert e == 0, e


MB = 1 << 20
N = 74 * MB            # one 64-frame sub-batch of records
host = [vp() for _ in range(8)]; dev = [vp() for _ in range(8)]; st = [vp() for _ in range(8)]
for i in range(8):
    ck(hip.hipHostMalloc(C.byref(host[i]), C.c_size_t(N), 0)); ck(hip.hipMalloc(C.byref(dev[i]), C.c_size_t(N)))
    ck(hip.hipStreamCreateWithFlags(C.byref(st[i]), 1))
    C.memset(host[i], 1, N)


def run(n_d2h, n_h2d, iters=12):
    def go():
        for _ in range(iters):
            for i in range(n_d2h):
                ck(hip.hipMemcpyAsync(host[i], dev[i], C.c_size_t(N), 2, st[i]))
            for i in range(n_h2d):
                ck(hip.hipMemcpyAsync(dev[4 + i], host[4 + i], C.c_size_t(N), 1, st[4 + i]))
        for s in st:
            ck(hip.hipStreamSynchronize(s))
    go()
    t0 = time.perf_counter(); go(); dt = time.perf_counter() - t0
    return n_d2h * iters * N / dt / 1e9, n_h2d * iters * N / dt / 1e9


for a, b in ((1, 0), (2, 0), (4, 0), (0, 1), (0, 2), (0, 4), (1, 1), (2, 2), (4, 4)):
    d, h = run(a, b)
    print(f"{a} D2H stream(s) + {b} H2D stream(s), 74 MB copies: D2H {d:6.1f} GB/s, H2D {h:6.1f} GB/s", flush=True)
