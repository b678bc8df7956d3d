// calib.hip -- xfh_bench_calib (include/xfeat_hip_bench.h): kernels that move an exactly known number of bytes per launch, one per
// access width the extraction kernels use, so that the FETCH_SIZE / WRITE_SIZE counters of rocprofv3 can be calibrated on this
// box before any measured-traffic figure is quoted (MI355X_MICROARCH.md, HBM section).  Measurement tooling, not product path.
#include "ctx.h"

template <int MODE>
__global__ __launch_bounds__(256)
void k_calib(float* __restrict__ buf, size_t n_floats, float* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (MODE == 0) {                                         // read, 4 B per lane
        float a = 0.f;
        for (size_t i = t; i < n_floats; i += stride) a += buf[i];
        if (a == 1.2345e-30f) *sink = a;
    } else if (MODE == 1) {                                  // read, 16 B per lane
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (size_t i = t; i < n_floats / 4; i += stride) a += ((const f32x4*)buf)[i];
        if (a[0] + a[1] + a[2] + a[3] == 1.2345e-30f) *sink = a[0];
    } else if (MODE == 2) {                                  // write, 4 B per lane
        for (size_t i = t; i < n_floats; i += stride) buf[i] = (float)i;
    } else if (MODE == 3) {                                  // write, 16 B per lane
        for (size_t i = t; i < n_floats / 4; i += stride) ((f32x4*)buf)[i] = f32x4{(float)i, 0.f, 1.f, 2.f};
    } else if (MODE == 4) {                                  // read, 32 B per lane (an 8-channel NHWC pixel: block1.1 / block1.2 inputs)
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (size_t i = t; i < n_floats / 8; i += stride) { a += ((const f32x4*)buf)[2 * i]; a += ((const f32x4*)buf)[2 * i + 1]; }
        if (a[0] + a[1] + a[2] + a[3] == 1.2345e-30f) *sink = a[0];
    } else {                                                 // write, 32 B per lane
        for (size_t i = t; i < n_floats / 8; i += stride) { ((f32x4*)buf)[2 * i] = f32x4{(float)i, 0.f, 1.f, 2.f}; ((f32x4*)buf)[2 * i + 1] = f32x4{3.f, 4.f, 5.f, 6.f}; }
    }
}

extern "C" int xfh_bench_calib(xfh_ctx* c, int mode, size_t nbytes, int iters) {
    if (!c || mode < 0 || mode > 5 || nbytes < 4096 || (nbytes & 31) || iters < 1) return XFH_ERR_INVALID_ARG;
    if (hipSetDevice(c->cfg.device) != hipSuccess) return XFH_ERR_HIP;
    float* buf = nullptr; float* sink = nullptr;
    if (hipMalloc((void**)&buf, nbytes) != hipSuccess) return XFH_ERR_OUT_OF_MEMORY;
    if (hipMalloc((void**)&sink, 256) != hipSuccess) { hipFree(buf); return XFH_ERR_OUT_OF_MEMORY; }
    hipMemsetAsync(buf, 0, nbytes, c->stream);
    const size_t nf = nbytes / 4;
    const dim3 grid(256 * 8), block(256);
    for (int i = 0; i < iters; ++i) {
        switch (mode) {
            case 0: hipLaunchKernelGGL(k_calib<0>, grid, block, 0, c->stream, buf, nf, sink); break;
            case 1: hipLaunchKernelGGL(k_calib<1>, grid, block, 0, c->stream, buf, nf, sink); break;
            case 2: hipLaunchKernelGGL(k_calib<2>, grid, block, 0, c->stream, buf, nf, sink); break;
            case 3: hipLaunchKernelGGL(k_calib<3>, grid, block, 0, c->stream, buf, nf, sink); break;
            case 4: hipLaunchKernelGGL(k_calib<4>, grid, block, 0, c->stream, buf, nf, sink); break;
            default: hipLaunchKernelGGL(k_calib<5>, grid, block, 0, c->stream, buf, nf, sink); break;
        }
    }
    const hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(buf); hipFree(sink);
    if (e != hipSuccess) { c->hip_err = std::string("xfh_bench_calib: ") + hipGetErrorString(e); return XFH_ERR_HIP; }
    return XFH_OK;
}
