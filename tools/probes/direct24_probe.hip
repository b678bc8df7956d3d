// direct24_probe.hip -- feasibility probe (round 5): the 3x3 24->24 convolutions (block2.0 / block2.1) as a DIRECT convolution on the packed-fp32 VALU
// (v_pk_fma_f32, weights as scalar-register pairs) instead of v_mfma_f32_32x32x2_f32 with N = 24 padded to 32.  On gfx950 both have the same peak, and the
// direct form wastes no lanes on padding.  Raw staging, no prologue / statistics: timing only.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CONSTAS __attribute__((address_space(4)))
#define CI 24
#define CO 24
#ifndef CHUNK
#define CHUNK 2
#endif
#ifndef HALF
#define HALF 12
#endif
template <int PX>       // output pixels per thread along x (1 or 2)
__global__ __launch_bounds__(256)
void k_direct24(const float* __restrict__ in, const float* __restrict__ w, float* __restrict__ out, int H, int W) {
    constexpr int TW = 16 * PX, TH = 16, IW = TW + 2, IH = TH + 2;
    __shared__ __attribute__((aligned(16))) float s_in[IH * IW * CI];
    const int t = threadIdx.x, b = blockIdx.z;
    const int tiles_x = (W + TW - 1) / TW;
    const int tx0 = (blockIdx.x % tiles_x) * TW, ty0 = (blockIdx.x / tiles_x) * TH;
    const float* ib = in + (size_t)b * H * W * CI;
    for (int e = t; e < IH * IW * (CI / 4); e += 256) {
        const int pix = e / (CI / 4), g = e % (CI / 4);
        const int gy = ty0 - 1 + pix / IW, gx = tx0 - 1 + pix % IW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *(const f32x4*)(ib + ((size_t)gy * W + gx) * CI + g * 4);
        *(f32x4*)(s_in + pix * CI + g * 4) = v;
    }
    __syncthreads();
    const int tx = (t & 15) * PX, ty = t >> 4;
    f32x2 acc[PX][CO / 2];
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int c = 0; c < CO / 2; ++c) acc[p][c] = f32x2{0.f, 0.f};
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const CONSTAS float* wt = (const CONSTAS float*)w + tap * CI * CO;
#pragma unroll
        for (int hf = 0; hf < CI / HALF; ++hf) {           // HALF input channels at a time: their values for both pixels in registers
            float v[PX][HALF];
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                const float* sp = s_in + ((ty + ky) * IW + tx + p + kx) * CI + hf * HALF;
#pragma unroll
                for (int g = 0; g < HALF / 4; ++g) { const f32x4 q = *(const f32x4*)(sp + g * 4); v[p][g * 4] = q.x; v[p][g * 4 + 1] = q.y; v[p][g * 4 + 2] = q.z; v[p][g * 4 + 3] = q.w; }
            }
#pragma unroll
            for (int c2 = 0; c2 < HALF; c2 += CHUNK) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int ci = c2; ci < c2 + CHUNK; ++ci) {
                    const CONSTAS f32x2* wr = (const CONSTAS f32x2*)(wt + (hf * HALF + ci) * CO);
#pragma unroll
                    for (int c = 0; c < CO / 2; ++c) {
                        const f32x2 ww = wr[c];
#pragma unroll
                        for (int p = 0; p < PX; ++p) acc[p][c] = __builtin_elementwise_fma(f32x2{v[p][ci], v[p][ci]}, ww, acc[p][c]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const int oy = ty0 + ty, ox = tx0 + tx + p;
        if (oy < H && ox < W) {
            float* o = out + (size_t)b * H * W * CO + ((size_t)oy * W + ox) * CO;
#pragma unroll
            for (int g = 0; g < CO / 4; ++g) *(f32x4*)(o + g * 4) = f32x4{acc[p][2 * g][0], acc[p][2 * g][1], acc[p][2 * g + 1][0], acc[p][2 * g + 1][1]};
        }
    }
}
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, H = 120, W = 160;
    float *in, *w, *out;
    hipMalloc((void**)&in, (size_t)B * H * W * CI * 4); hipMalloc((void**)&out, (size_t)B * H * W * CO * 4); hipMalloc((void**)&w, 9 * CI * CO * 4);
    std::vector<float> hw(9 * CI * CO); for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) >> 20 & 1023) / 1024.f - 0.5f;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); hipMemset(in, 0x3c, (size_t)B * H * W * CI * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int px = 1; px <= 2; ++px) {
        const int tw = 16 * px, tiles = ((W + tw - 1) / tw) * ((H + 15) / 16);
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 5; ++i) { if (px == 1) hipLaunchKernelGGL(k_direct24<1>, dim3(tiles, 1, B), dim3(256), 0, 0, in, w, out, H, W); else hipLaunchKernelGGL(k_direct24<2>, dim3(tiles, 1, B), dim3(256), 0, 0, in, w, out, H, W); }
            hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i) { if (px == 1) hipLaunchKernelGGL(k_direct24<1>, dim3(tiles, 1, B), dim3(256), 0, 0, in, w, out, H, W); else hipLaunchKernelGGL(k_direct24<2>, dim3(tiles, 1, B), dim3(256), 0, 0, in, w, out, H, W); }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("direct 24->24 3x3, %d px per thread, B = %d: %.1f us per launch = %.1f TFLOP/s (k_conv_mfma_p: 640-670 us = 76-80)\n", px, B, ms * 100, 2.0 * B * H * W * 9 * CI * CO / (ms * 1e-4) / 1e12);
        }
    }
    return 0;
}
