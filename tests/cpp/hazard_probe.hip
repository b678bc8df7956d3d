// hazard_probe.hip -- compiled ON THE GPU BOX by tests/test_gpu_hazard.py with that box's own hipcc (not the container's): does the library's
// hand-counted MFMA -> VALU-read padding (common.h: XFH_MFMA_SETTLE) still separate the last MFMA of a K loop from an epilogue that starts
// behind a TAKEN branch, with this compiler and on this GPU?  The kernels reproduce the shape every MFMA kernel of the library has --
// K loop, XFH_MFMA_SETTLE(), a scalar branch, VALU reads of the accumulators -- for both MFMA forms the library issues
// (v_mfma_f32_32x32x2_f32: 16 passes, v_mfma_f32_16x16x4_f32: 8 passes) and for both directions of the branch; the result is compared
// with a plain fp32 fma chain computed on the host.  A third pair of launches runs WITHOUT the padding: informational (shows whether this
// compiler pads the taken path on its own).  Exit code 0 = every padded launch exact; prints one line per launch.
#include "../../xfeatslam_amd/csrc/common.h"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

template <bool SETTLE>
__global__ void k_probe32(const float* __restrict__ A, const float* __restrict__ B, const int* __restrict__ flag, float* __restrict__ out, int K) {
    // C[32x32] = A[32xK] . B[Kx32]; lane i = row (A) / column (B), lane-half h = k parity
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + h], B[(k + h) * 32 + i], acc, 0, 0, 0);
    if (SETTLE) XFH_MFMA_SETTLE();
    const int f = __builtin_amdgcn_readfirstlane(flag[0]);
    float o[16];
    if (f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = acc[r] + 1.0f;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = acc[r] * 2.0f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = o[r];
}
template <bool SETTLE>
__global__ void k_probe16(const float* __restrict__ A, const float* __restrict__ B, const int* __restrict__ flag, float* __restrict__ out, int K) {
    // C[16x16] = A[16xK] . B[Kx16]; lane i = row / column, q = lane >> 4 = k mod 4
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k + q], B[(k + q) * 16 + i], acc, 0, 0, 0);
    if (SETTLE) XFH_MFMA_SETTLE();
    const int f = __builtin_amdgcn_readfirstlane(flag[0]);
    float o[4];
    if (f) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = acc[r] + 1.0f;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = acc[r] * 2.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(4 * q + r) * 16 + i] = o[r];
}

int main() {
    const int K = 64;
    std::vector<float> A(32 * K), B(K * 32), ref32(32 * 32), ref16(16 * 16), got(32 * 32);
    unsigned x = 777u;
    for (auto& v : A) { x = x * 1664525u + 1013904223u; v = (float)((x >> 9) & 0x3fff) / 8192.f - 1.f; }
    for (auto& v : B) { x = x * 1664525u + 1013904223u; v = (float)((x >> 9) & 0x3fff) / 8192.f - 1.f; }
    float *dA, *dB, *dO; int* dF;
    if (hipMalloc((void**)&dA, A.size() * 4) != hipSuccess) { printf("no device\n"); return 3; }
    hipMalloc((void**)&dB, B.size() * 4); hipMalloc((void**)&dO, got.size() * 4); hipMalloc((void**)&dF, 4);
    int rt = 0; hipRuntimeGetVersion(&rt);
    printf("probe built with clang %d.%d.%d, HIP %d.%d.%d; runtime HIP %d\n", __clang_major__, __clang_minor__, __clang_patchlevel__, HIP_VERSION_MAJOR, HIP_VERSION_MINOR, HIP_VERSION_PATCH, rt);
    int bad_padded = 0;
    for (int form = 0; form < 2; ++form) {
        const int N = form ? 16 : 32;
        // host reference: one fp32 fma chain in k order per output = what the MFMA computes; A is N x K (row stride K), B is K x N
        std::vector<float> a2(N * K), b2(K * N);
        for (int r = 0; r < N; ++r) for (int k = 0; k < K; ++k) a2[r * K + k] = A[r * K + k];
        for (int k = 0; k < K; ++k) for (int c = 0; c < N; ++c) b2[k * N + c] = B[k * 32 + c];
        hipMemcpy(dA, a2.data(), a2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, b2.data(), b2.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> ref(N * N);
        for (int r = 0; r < N; ++r) for (int c = 0; c < N; ++c) { float s = 0.f; for (int k = 0; k < K; ++k) s = fmaf(a2[r * K + k], b2[k * N + c], s); ref[r * N + c] = s; }
        for (int settle = 1; settle >= 0; --settle)
            for (int f = 0; f < 2; ++f) {
                int bad = 0;
                for (int rep = 0; rep < 50; ++rep) {
                    hipMemcpy(dF, &f, 4, hipMemcpyHostToDevice);
                    hipMemset(dO, 0xff, got.size() * 4);
                    if (form == 0) { if (settle) hipLaunchKernelGGL(k_probe32<true>, dim3(1), dim3(64), 0, 0, dA, dB, dF, dO, K); else hipLaunchKernelGGL(k_probe32<false>, dim3(1), dim3(64), 0, 0, dA, dB, dF, dO, K); }
                    else { if (settle) hipLaunchKernelGGL(k_probe16<true>, dim3(1), dim3(64), 0, 0, dA, dB, dF, dO, K); else hipLaunchKernelGGL(k_probe16<false>, dim3(1), dim3(64), 0, 0, dA, dB, dF, dO, K); }
                    hipMemcpy(got.data(), dO, (size_t)N * N * 4, hipMemcpyDeviceToHost);
                    for (int e = 0; e < N * N; ++e) { const float want = f ? ref[e] + 1.0f : ref[e] * 2.0f; if (!(got[e] == want)) ++bad; }
                }
                printf("%s  %-22s branch flag %d: %d wrong values in 50 launches\n", form ? "v_mfma_f32_16x16x4_f32" : "v_mfma_f32_32x32x2_f32", settle ? "with XFH_MFMA_SETTLE" : "WITHOUT the padding", f, bad);
                if (settle) bad_padded += bad;
            }
    }
    printf(bad_padded ? "HAZARD: the padded sequence returned stale accumulators\n" : "hazard probe ok\n");
    return bad_padded ? 1 : 0;
}
