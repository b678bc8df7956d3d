// mfma4_probe.hip -- v_mfma_f32_4x4x1_16b_f32 on gfx950: (1) operand / result layout and numerics (one fp32 fma per output and k: a chain
// of them is the sequential fma chain in k order, like 32x32x2 and 16x16x4), (2) issue rate with twelve independent accumulators
// (two pixel blocks x six channel groups: the shape a 24-channel convolution needs), registers only and with the planned LDS operand
// reads (A: one ds_read_b128 per lane = 4 consecutive k of the lane's pixel; B: a broadcast ds_read_b128, 4 distinct addresses).
// hipcc --offload-arch=gfx950 -O3 -o mfma4_probe mfma4_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// one wave, 16 blocks: D[b][i][j] = sum_k A[b][i][k] * B[b][k][j]
__global__ void k_layout(const float* A, const float* B, float* D, int K) {
    const int l = threadIdx.x, b = l >> 2, q = l & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        const float a = A[(b * 4 + q) * K + k];          // assumed: lane (b, i = q) holds A[b][i][k]
        const float bb = B[(b * K + k) * 4 + q];         // assumed: lane (b, j = q) holds B[b][k][j]
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bb, acc, 0, 0, 0);
    }
    for (int v = 0; v < 4; ++v) D[(b * 4 + v) * 4 + q] = acc[v];      // assumed: lane (b, j = q), register v = row i
}

template <int MODE>     // 0: register operands, 1: LDS operands (A per lane, B broadcast), 2: 32x32x2 reference loop (registers)
__global__ __launch_bounds__(256)
void k_rate(float* out, int iters, float a0, float b0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // >= 64 KB; the launch asks for more to limit the workgroups per CU
    const int t = threadIdx.x, l = t & 63;
    for (int i = t; i < 10240; i += 256) lds[i] = 1e-3f * (float)(i & 7);
    __syncthreads();
    if (MODE == 2) {
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        f32x16 acc[4];
        for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 6; ++u)                       // 24 MFMAs of 64 cycles = the cycles of 192 4x4x1
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[n], 0, 0, 0);
        float s = 0.f;
        for (int n = 0; n < 4; ++n) s += acc[n][0] + acc[n][15];
        if (s == 123.f) out[t] = s;
        return;
    }
    f32x4 acc[2][6];
    for (int nb = 0; nb < 2; ++nb) for (int g = 0; g < 6; ++g) acc[nb][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* pa0 = lds + (t & 63) * 28 + (t >> 6) * 64;                 // pixel stride 28 floats (24 channels + 4)
    const float* pa1 = pa0 + 64 * 28;
    const float* pb = lds + 4096 + (l & 3) * 220;                             // weight row stride 220 floats
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {                                       // 4 x (4 k x 12 accumulators) = 192 MFMAs of 8 cycles per iteration
            f32x4 av0, av1, bv[6];
            if (MODE == 1) {
                av0 = *(const f32x4*)(pa0 + kk * 4); av1 = *(const f32x4*)(pa1 + kk * 4);
#pragma unroll
                for (int g = 0; g < 6; ++g) bv[g] = *(const f32x4*)(pb + g * 4 * 220 + kk * 4);
            } else {
                av0 = f32x4{a0, a0, a0, a0}; av1 = av0;
#pragma unroll
                for (int g = 0; g < 6; ++g) bv[g] = f32x4{b0, b0, b0, b0};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    acc[0][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av0[j], bv[g][j], acc[0][g], 0, 0, 0);
                    acc[1][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av1[j], bv[g][j], acc[1][g], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
    for (int nb = 0; nb < 2; ++nb) for (int g = 0; g < 6; ++g) s += acc[nb][g][0] + acc[nb][g][3];
    if (s == 123.f) out[t] = s;
}

int main() {
    const int K = 72;
    std::vector<float> A(64 * K), B(16 * K * 4), D(256);
    srand(11);
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 4096 * 4));
    int bad = 0, bad_pair = 0;
    for (int tr = 0; tr < 100; ++tr) {
        for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * (tr % 3 == 0 ? 1e-3f : 1.f);
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
        CK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < 16; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            float s = 0.f, p = 0.f;
            for (int k = 0; k < K; ++k) s = fmaf(A[(b * 4 + i) * K + k], B[(b * K + k) * 4 + j], s);
            for (int k = 0; k < K; k += 2) p = p + (A[(b * 4 + i) * K + k] * B[(b * K + k) * 4 + j] + A[(b * 4 + i) * K + k + 1] * B[(b * K + k + 1) * 4 + j]);
            if (s != D[(b * 4 + i) * 4 + j]) ++bad;
            if (p != D[(b * 4 + i) * 4 + j]) ++bad_pair;
        }
    }
    printf("4x4x1_16b: 100 trials x 256 outputs, K = %d: mismatches vs sequential fma chain in k order (assumed layout): %d; vs pairwise sums: %d\n", K, bad, bad_pair);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    CK(hipFuncSetAttribute((const void*)k_rate<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_rate<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_rate<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int wps = 4; wps >= 1; wps >>= 1)                   // waves per SIMD: 4 (four workgroups of 4 waves per CU), 2, 1
    for (int mode = 0; mode < 3; ++mode) {
        const int grid = 256 * wps;
        const size_t lds = wps == 4 ? 64 * 1024 / 1 / 2 + 32 * 1024 : wps == 2 ? 80 * 1024 : 120 * 1024;      // 64 KB x 4 does not fit: 40 KB each; 80 KB x 2; 120 KB x 1
        if (mode == 0) printf("-- %d wave(s) per SIMD (%zu KB of LDS per workgroup)\n", wps, (wps == 4 ? (size_t)40 * 1024 : lds) / 1024);
        for (int rep = 0; rep < 2; ++rep) {
            const size_t L = wps == 4 ? (size_t)40 * 1024 : lds;
            CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(grid), dim3(256), L, 0, dD, iters, 1.0f, 1e-9f);
            else if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(grid), dim3(256), L, 0, dD, iters, 1.0f, 1e-9f);
            else hipLaunchKernelGGL(k_rate<2>, dim3(grid), dim3(256), L, 0, dD, iters, 1.0f, 1e-9f);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = mode == 2 ? (double)grid * 4 * iters * 24 * 4096.0 : (double)grid * 4 * iters * 192 * 512.0;
            if (rep) printf("%s: %.3f ms, %.1f TFLOP/s\n", mode == 0 ? "4x4x1, register operands" : mode == 1 ? "4x4x1, LDS operands (2 A reads + 6 broadcast B reads per 48 MFMAs)" : "32x32x2, register operands", ms, flops / ms / 1e9);
        }
    }
    return bad ? 1 : 0;
}
