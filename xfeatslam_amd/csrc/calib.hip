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

// xfh_bench_sclk: the shader clock the GPU holds while every SIMD issues v_mfma_f32_32x32x2_f32 back to back.  One workgroup of 8 waves per
// CU runs `mfmas` MFMAs per wave over 8 independent accumulators; wave 0 of every workgroup reports its span in shader clocks (clock64 =
// s_memtime) and in the constant 100 MHz counter (wall_clock64 = s_memrealtime): sclk = 100 MHz * clocks / ticks, and clocks / MFMAs = the
// issue period (64 cycles alone, 128 with two waves per SIMD).  The peak the match GEMM is priced against (157.3 TFLOP/s) assumes 2.4 GHz; boxes
// of this pool hold 2.0 - 2.2 GHz under this load (profiles/r04_pipe_probe.log), which caps ANY f32 MFMA kernel at sclk / 2.4 of that peak.
__global__ __launch_bounds__(512, 2)
void k_sclk(int mfmas, long long* __restrict__ out, float* __restrict__ sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[8];
    for (int q = 0; q < 8; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    const float a = (float)lane * 1e-3f, b = 1.0f;
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < mfmas; it += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    }
    float r = 0.f;
    for (int q = 0; q < 8; ++q) r += acc[q][0] + acc[q][15];
    const long long c1 = clock64(), w1 = wall_clock64();
    if (wave == 0 && lane == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = w1 - w0; }
    if (r == 123.456f) *sink = r;
}
extern "C" int xfh_bench_sclk(xfh_ctx* c, int mfmas, double* sclk_mhz, double* cycles_per_mfma) {
    if (!c || mfmas < 8 || !sclk_mhz || !cycles_per_mfma) return XFH_ERR_INVALID_ARG;
    if (hipSetDevice(c->cfg.device) != hipSuccess) return XFH_ERR_HIP;
    const int G = c->num_cu;
    long long* out = nullptr; float* sink = nullptr;
    if (hipMalloc((void**)&out, (size_t)G * 16) != hipSuccess) return XFH_ERR_OUT_OF_MEMORY;
    if (hipMalloc((void**)&sink, 256) != hipSuccess) { hipFree(out); return XFH_ERR_OUT_OF_MEMORY; }
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_sclk, dim3(G), dim3(512), 0, c->stream, mfmas, out, sink);     // the last launch counts: clocks settled
    hipError_t e = hipStreamSynchronize(c->stream);
    long long* h = (long long*)malloc((size_t)G * 16);
    if (e == hipSuccess) e = hipMemcpy(h, out, (size_t)G * 16, hipMemcpyDeviceToHost);
    double clk = 0, tick = 0;
    for (int b = 0; b < G; ++b) { clk += (double)h[2 * b]; tick += (double)h[2 * b + 1]; }
    free(h); hipFree(out); hipFree(sink);
    if (e != hipSuccess) { c->hip_err = std::string("xfh_bench_sclk: ") + hipGetErrorString(e); return XFH_ERR_HIP; }
    *sclk_mhz = tick > 0 ? 100.0 * clk / tick : 0.0;
    *cycles_per_mfma = clk / G / mfmas;
    return XFH_OK;
}
