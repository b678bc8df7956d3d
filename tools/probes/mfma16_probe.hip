// mfma16_probe.hip -- is v_mfma_f32_16x16x4_f32 an exact fp32 fma chain in k order (like 32x32x2, which the convolutions rely on for
// bit-exactness with the oracle), and what is its dependent-issue latency?  hipcc --offload-arch=gfx950 -O3 -o mfma16_probe mfma16_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// one wave: D[16][16] = sum over K of A[16][K] * B[K][16], K = 4 * nstep, accumulated by nstep chained MFMAs
__global__ void k_probe(const float* A, const float* B, float* D, int K) {
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const float a = A[i * K + k0 + q], b = B[(k0 + q) * 16 + i];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = acc[r];
}
__global__ void k_lat(float* out, int n, float a, float b) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = wall_clock64();
    for (int k = 0; k < n; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    const long long t1 = wall_clock64();
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0);
}
int main() {
    const int K = 64;
    std::vector<float> A(16 * K), B(K * 16), D(256), R(256);
    srand(7);
    int bad_seq = 0, bad_pair = 0, trials = 200;
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 512 * 4));
    for (int t = 0; t < trials; ++t) {
        for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * (t % 3 == 0 ? 1e-3f : 1.f);
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
        CK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            float s = 0.f;                                            // sequential chain in k order
            for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[k * 16 + j], s);
            if (s != D[i * 16 + j]) ++bad_seq;
            float p = 0.f;                                            // pairs first: (k0*.. + k1*..) style would differ
            for (int k = 0; k < K; k += 2) p = p + (A[i * K + k] * B[k * 16 + j] + A[i * K + k + 1] * B[(k + 1) * 16 + j]);
            if (p != D[i * 16 + j]) ++bad_pair;
        }
    }
    printf("16x16x4: %d trials x 256 outputs: mismatches vs sequential fma chain in k order: %d; vs pairwise sums: %d\n", trials, bad_seq, bad_pair);
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, dD, 4096, 1.0f, 1e-9f);
    CK(hipMemcpy(D.data(), dD, 65 * 4, hipMemcpyDeviceToHost));
    printf("dependent 16x16x4 chain: %.1f wall-clock ticks (100 MHz) per 4096 -> %.2f ns each\n", D[64], D[64] * 10.0 / 4096);
    return bad_seq ? 1 : 0;
}
