// hazard_probe.hip -- compiled ON THE GPU BOX by tests/test_gpu_hazard.py with that box's own hipcc (not the container's).  Two parts:
//  (1) the shape every MFMA kernel of the library has -- K loop, XFH_MFMA_SETTLE() (common.h), a scalar branch, VALU reads of the accumulators -- for both MFMA
//      forms the library issues (v_mfma_f32_32x32x2_f32: 16 passes, v_mfma_f32_16x16x4_f32: 8 passes), both directions of the branch, with and without the
//      macro, compared with a plain fp32 fma chain computed on the host;
//  (2) the hazard itself with the compiler out of the picture (round 6, k_sweep32 / k_sweep16 below): last MFMA + N wait states + optional taken branch + first
//      read (VALU / global_store / ds_write) inside one inline-asm statement, N = 0 .. 20.  On gfx950 every configuration is exact from N = 0 on: the
//      hardware interlocks the dependency, which is why the hand-counted pad of rounds 2-5 was removed from XFH_MFMA_SETTLE.  Should a future GPU or
//      runtime return stale accumulators at small N, the sweep shows it ("red configurations"), and the test fails if N = 20 -- the old pad, -DXFH_SETTLE_NOPS --
//      is not clean either.
// Exit code 0 = part (1) exact with the macro and the sweep clean at N = 20; prints one line per launch group.
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

// ---- the hazard itself, compiler out of the picture (round 6: "a hazard test that can fail").  The LAST MFMA of the chain, N wait states and -- optionally --
// a taken scalar branch sit in ONE inline-asm statement: the compiler's hazard recogniser cannot see an MFMA in there, so it pads nothing, and the first VALU
// read of the accumulators follows the statement directly (scheduling barrier).  Swept over N this shows (a) that the hardware does NOT interlock the
// MFMA -> VALU-read dependency (small N returns the accumulators of the step before: the stale read common.h describes), (b) how many wait states the
// two MFMA forms need with and without a taken branch in between, (c) that XFH_MFMA_SETTLE's 20 (s_nop 15; s_nop 3) cover it.
// CONS: who reads the accumulators first -- 0: a VALU instruction, 1: global_store_dword (VMEM reads the registers), 2: ds_write_b32 (LDS reads them)
template <int N, bool BR, int CONS>
__global__ void k_sweep32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int K) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K - 2; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + h], B[(k + h) * 32 + i], acc, 0, 0, 0);
    const float a = A[i * K + K - 2 + h], b = B[(K - 2 + h) * 32 + i];
    float* o = out + (4 * h) * 32 + i;
    __builtin_amdgcn_sched_barrier(0);
    if (BR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t.rept %c3\n\ts_nop 0\n\t.endr\n\ts_cmp_eq_u32 %4, 1\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\ts_nop 0\n1:" : "+v"(acc) : "v"(a), "v"(b), "n"(N), "s"(1) : "scc");
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t.rept %c3\n\ts_nop 0\n\t.endr" : "+v"(acc) : "v"(a), "v"(b), "n"(N));
    __builtin_amdgcn_sched_barrier(0);
    if (CONS == 0) {
        float r0 = acc[0] + 1.0f, r1 = acc[1] + 1.0f, r2 = acc[2] + 1.0f, r3 = acc[3] + 1.0f;          // the first reads: N, N+1, ... wait states behind the MFMA
        __builtin_amdgcn_sched_barrier(0);
        o[0] = r0; o[32] = r1; o[64] = r2; o[96] = r3;
    } else if (CONS == 1) {
        o[0] = acc[0]; o[32] = acc[1]; o[64] = acc[2]; o[96] = acc[3];
    } else {
        __shared__ float sh[4 * 64];
        sh[lane] = acc[0]; sh[64 + lane] = acc[1]; sh[128 + lane] = acc[2]; sh[192 + lane] = acc[3];
        __builtin_amdgcn_sched_barrier(0);
        o[0] = sh[lane]; o[32] = sh[64 + lane]; o[64] = sh[128 + lane]; o[96] = sh[192 + lane];
    }
}
template <int N, bool BR, int CONS>
__global__ void k_sweep16(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int K) {
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K - 4; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k + q], B[(k + q) * 16 + i], acc, 0, 0, 0);
    const float a = A[i * K + K - 4 + q], b = B[(K - 4 + q) * 16 + i];
    float* o = out + (4 * q) * 16 + i;
    __builtin_amdgcn_sched_barrier(0);
    if (BR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\t.rept %c3\n\ts_nop 0\n\t.endr\n\ts_cmp_eq_u32 %4, 1\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\ts_nop 0\n1:" : "+v"(acc) : "v"(a), "v"(b), "n"(N), "s"(1) : "scc");
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\t.rept %c3\n\ts_nop 0\n\t.endr" : "+v"(acc) : "v"(a), "v"(b), "n"(N));
    __builtin_amdgcn_sched_barrier(0);
    if (CONS == 0) {
        float r0 = acc[0] + 1.0f, r1 = acc[1] + 1.0f, r2 = acc[2] + 1.0f, r3 = acc[3] + 1.0f;
        __builtin_amdgcn_sched_barrier(0);
        o[0] = r0; o[16] = r1; o[32] = r2; o[48] = r3;
    } else if (CONS == 1) {
        o[0] = acc[0]; o[16] = acc[1]; o[32] = acc[2]; o[48] = acc[3];
    } else {
        __shared__ float sh[4 * 64];
        sh[lane] = acc[0]; sh[64 + lane] = acc[1]; sh[128 + lane] = acc[2]; sh[192 + lane] = acc[3];
        __builtin_amdgcn_sched_barrier(0);
        o[0] = sh[lane]; o[16] = sh[64 + lane]; o[32] = sh[128 + lane]; o[48] = sh[192 + lane];
    }
}
template <int N, bool BR, int CONS>
static void launch_sweep(int form, const float* dA, const float* dB, float* dO, int K) {
    if (form == 0) hipLaunchKernelGGL((k_sweep32<N, BR, CONS>), dim3(1), dim3(64), 0, 0, dA, dB, dO, K);
    else hipLaunchKernelGGL((k_sweep16<N, BR, CONS>), dim3(1), dim3(64), 0, 0, dA, dB, dO, K);
}
template <bool BR, int CONS>
static void launch_sweep_n(int n, int form, const float* dA, const float* dB, float* dO, int K) {
    switch (n) {
#define CASE(v) case v: launch_sweep<v, BR, CONS>(form, dA, dB, dO, K); break;
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(6) CASE(8) CASE(10) CASE(12) CASE(14) CASE(16) CASE(18) CASE(20)
#undef CASE
    }
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
    // ---- sweep: last MFMA + N wait states (+ taken branch) + VALU read, no compiler padding (k_sweep32 / k_sweep16)
    int red = 0, bad_at_pad = 0;
    for (int form = 0; form < 2; ++form) {
        const int N = form ? 16 : 32;
        std::vector<float> a2(N * K), b2(K * N), ref(N * N);
        for (int r = 0; r < N; ++r) for (int k = 0; k < K; ++k) a2[r * K + k] = A[r * K + k];
        for (int k = 0; k < K; ++k) for (int c = 0; c < N; ++c) b2[k * N + c] = B[k * 32 + c];
        hipMemcpy(dA, a2.data(), a2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, b2.data(), b2.size() * 4, hipMemcpyHostToDevice);
        for (int r = 0; r < N; ++r) for (int c = 0; c < N; ++c) { float s2 = 0.f; for (int k = 0; k < K; ++k) s2 = fmaf(a2[r * K + k], b2[k * N + c], s2); ref[r * N + c] = s2; }
        static const int ns[13] = {0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20};
        static const char* cons_name[3] = {"VALU read", "global_store reads the registers", "ds_write reads the registers"};
        for (int cons = 0; cons < 3; ++cons)
        for (int br = 0; br < 2; ++br) {
            printf("sweep %s, %s, %s: wrong values of the first four accumulator reads (of %d, 20 launches each) at N wait states =", form ? "v_mfma_f32_16x16x4_f32" : "v_mfma_f32_32x32x2_f32",
                   cons_name[cons], br ? "taken s_cbranch in between" : "no branch", 20 * 4 * 64);
            int first_clean = -1;
            for (int ni = 0; ni < 13; ++ni) {
                const int n = ns[ni];
                int bad = 0;
                const float plus = cons == 0 ? 1.0f : 0.0f;
                for (int rep = 0; rep < 20; ++rep) {
                    hipMemset(dO, 0xff, got.size() * 4);
                    if (cons == 0) { if (br) launch_sweep_n<true, 0>(n, form, dA, dB, dO, K); else launch_sweep_n<false, 0>(n, form, dA, dB, dO, K); }
                    else if (cons == 1) { if (br) launch_sweep_n<true, 1>(n, form, dA, dB, dO, K); else launch_sweep_n<false, 1>(n, form, dA, dB, dO, K); }
                    else { if (br) launch_sweep_n<true, 2>(n, form, dA, dB, dO, K); else launch_sweep_n<false, 2>(n, form, dA, dB, dO, K); }
                    hipMemcpy(got.data(), dO, (size_t)N * N * 4, hipMemcpyDeviceToHost);
                    // the rows the four reads cover: 32x32: rows (r & 3) + 4 h for r = 0..3 -> rows 4h + r; 16x16: rows 4 q + r
                    for (int g = 0; g < (form ? 4 : 2); ++g) for (int r = 0; r < 4; ++r) for (int c = 0; c < N; ++c) {
                        const int row = 4 * g + r;
                        if (!(got[row * N + c] == ref[row * N + c] + plus)) ++bad;
                    }
                }
                printf(" %d:%d", n, bad);
                if (bad) { ++red; first_clean = -1; } else if (first_clean < 0) first_clean = n;
                if (n == 20) bad_at_pad += bad;
            }
            printf("  -> clean from N = %d on\n", first_clean);
        }
    }
    printf("%s; XFH_MFMA_SETTLE = 20 wait states: %s\n", red ? "hazard reproduced: the hardware does not interlock MFMA -> VALU read (red configurations above)" : "NO configuration failed: this GPU / probe shows no MFMA -> VALU read hazard",
           bad_at_pad ? "NOT ENOUGH" : "covers every form");
    printf("sweep red configurations: %d\n", red);
    printf((bad_padded || bad_at_pad) ? "HAZARD: the padded sequence returned stale accumulators\n" : "hazard probe ok\n");
    return (bad_padded || bad_at_pad) ? 1 : 0;
}
