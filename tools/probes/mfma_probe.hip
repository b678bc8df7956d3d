// mfma_probe.hip -- floor measurements for the 4096x4096x64 match GEMM on gfx950 (development
// probe, not part of the library): how long do 512 workgroups x 4 waves x 256 f32 MFMAs take
// (a) bare, (b) with the LDS operand reads, (c) with a global->LDS prologue of 64 KB per WG.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDK 68

template <int MODE, int NCHAIN>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ d1, const float* __restrict__ d2, float* out, int ntile) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 128 * LDK];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 31, h = lane >> 5;
    if (MODE >= 2) {
        const int sub = t & 15, r0 = t >> 4;
        f32x4 va[8], vb[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            va[p] = *(const f32x4*)(d1 + (size_t)(blockIdx.y * 128 + p * 16 + r0) * 64 + sub * 4);
            vb[p] = *(const f32x4*)(d2 + (size_t)(blockIdx.x * 128 + p * 16 + r0) * 64 + sub * 4);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            *(f32x4*)(smem + (p * 16 + r0) * LDK + sub * 4) = va[p];
            *(f32x4*)(smem + 128 * LDK + (p * 16 + r0) * LDK + sub * 4) = vb[p];
        }
        __syncthreads();
    }
    const float* pa = smem + ((wave >> 1) * 64 + i) * LDK + 4 * h;
    const float* pb = smem + 128 * LDK + ((wave & 1) * 64 + i) * LDK + 4 * h;
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float fa = (float)lane, fb = (float)t;
    for (int tile = 0; tile < ntile; ++tile) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4 a0, a1, b0, b1;
            if (MODE >= 1) {
                a0 = *(const f32x4*)(pa + g * 8); a1 = *(const f32x4*)(pa + 32 * LDK + g * 8);
                b0 = *(const f32x4*)(pb + g * 8); b1 = *(const f32x4*)(pb + 32 * LDK + g * 8);
            } else { a0 = f32x4{fa, fa, fa, fa}; a1 = a0; b0 = f32x4{fb, fb, fb, fb}; b1 = b0; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (NCHAIN == 4) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[3], 0, 0, 0);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[0], 0, 0, 0);
                }
            }
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 123.456f) out[t] = s;
}

template <int MODE, int NCHAIN>
static void run(const char* name, const float* d1, const float* d2, float* out, dim3 grid, int ntile) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<MODE, NCHAIN>), grid, dim3(256), 0, 0, d1, d2, out, ntile);
    hipDeviceSynchronize();
    float best = 1e9f, tot = 0.f;
    const int N = 50;
    for (int i = 0; i < N; ++i) {
        hipExtLaunchKernelGGL((probe<MODE, NCHAIN>), grid, dim3(256), 0, 0, e0, e1, 0, d1, d2, out, ntile);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; tot += ms;
    }
    // back-to-back launches timed as a group
    hipEventRecord(e0, 0);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((probe<MODE, NCHAIN>), grid, dim3(256), 0, 0, d1, d2, out, ntile);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float msg; hipEventElapsedTime(&msg, e0, e1);
    const double mfma = (double)grid.x * grid.y * 4 * 128.0 * ntile;
    const double flops = mfma * 2 * 32 * 32 * 2;
    printf("%-34s grid %4dx%-3d ntile %d: kernel avg %.2f us min %.2f us | back-to-back %.2f us/launch -> %.1f TF (%.1f%% of 157.3)\n",
           name, grid.x, grid.y, ntile, tot / N * 1e3, best * 1e3, msg / 100 * 1e3, flops / (msg / 100 * 1e-3) / 1e12, flops / (msg / 100 * 1e-3) / 157.3e12 * 100);
}

int main() {
    float *d1, *d2, *out;
    hipMalloc(&d1, 4096 * 64 * 4); hipMalloc(&d2, 4096 * 64 * 4); hipMalloc(&out, 4096);
    std::vector<float> h(4096 * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(d1, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d2, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 4>("bare MFMA, 4 chains", d1, d2, out, dim3(32, 32), 1);
    run<0, 1>("bare MFMA, 1 chain", d1, d2, out, dim3(32, 32), 1);
    run<0, 4>("bare MFMA, 4 chains, 2 tiles/WG", d1, d2, out, dim3(16, 32), 2);
    run<0, 4>("bare MFMA, 4 chains, 8 tiles/WG", d1, d2, out, dim3(4, 32), 8);
    run<0, 4>("bare MFMA, 4 chains, 64 tiles/WG", d1, d2, out, dim3(16, 32), 64);
    run<1, 4>("LDS reads + MFMA", d1, d2, out, dim3(32, 32), 1);
    run<1, 4>("LDS reads + MFMA, 2 tiles/WG", d1, d2, out, dim3(16, 32), 2);
    run<2, 4>("global->LDS + MFMA", d1, d2, out, dim3(32, 32), 1);
    run<2, 4>("global->LDS + MFMA, 2 tiles/WG", d1, d2, out, dim3(16, 32), 2);
    run<2, 4>("global->LDS + MFMA, 64 tiles/WG", d1, d2, out, dim3(16, 32), 64);
    return 0;
}
