// pipe_probe.hip -- do v_mfma_f32_32x32x2_f32 and plain VALU instructions of two waves on one SIMD overlap, take turns, or starve each
// other -- and what does s_setprio change?  (development probe; decides the phase structure of k_mnn_gemm_seg.)
// One workgroup of 8 waves per CU (waves w and w + 4 share a SIMD).  Role M: `nm` MFMAs over 8 independent accumulators.  Role V: `nv`
// independent v_max3_f32.  Every wave reports its own clock64() span.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// mode: bit0 = waves 4..7 run V (else idle), bit1 = waves 4..7 run M too (all-MFMA), prio_v / prio_m = s_setprio levels
template <int PV, int PM>
__global__ __launch_bounds__(512, 2) void k_pipe(int mode, int nm, int nv, long long* out, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool second = wave >= 4;
    const bool doM = !second || (mode & 2), doV = second && (mode & 1);
    __syncthreads();
    const long long t0 = clock64();
    float r = 0.f;
    if (doM) {
        if (PM) __builtin_amdgcn_s_setprio(PM);
        f32x16 acc[8];
        for (int q = 0; q < 8; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
        const float a = (float)lane * 1e-3f, b = 1.0f;
        for (int it = 0; it < nm; it += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q) r += acc[q][0] + acc[q][15];
    } else if (doV) {
        if (PV) __builtin_amdgcn_s_setprio(PV);
        float v[8];
        for (int q = 0; q < 8; ++q) v[q] = (float)(lane + q);
        for (int it = 0; it < nv; it += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(v[(q + 1) & 7]), "v"(v[(q + 3) & 7]));
        }
        for (int q = 0; q < 8; ++q) r += v[q];
    }
    const long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 123.456f) *sink = r;
}

// same wave: K plain VALU instructions behind every MFMA -- do they run in the MFMA's shadow (64 cycles per MFMA stay 64)?
template <int K, int TWO>
__global__ __launch_bounds__(512, 2) void k_shadow(int nm, long long* out, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (!TWO && wave >= 4) return;
    const long long t0 = clock64();
    f32x16 acc[8];
    for (int q = 0; q < 8; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    float v[16];
    for (int q = 0; q < 16; ++q) v[q] = (float)(lane + q);
    const float a = (float)lane * 1e-3f, b = 1.0f;
    for (int it = 0; it < nm; it += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[k & 15]) : "v"(v[(k + 5) & 15]), "v"(v[(k + 9) & 15]));
        }
    }
    float r = 0.f;
    for (int q = 0; q < 8; ++q) r += acc[q][0] + acc[q][15];
    for (int q = 0; q < 16; ++q) r += v[q];
    const long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 123.456f) *sink = r;
}

// "yielding": the M waves put NOPS * 16 cycles of s_nop behind every MFMA (nothing of the vector type at the head of the wave while the
// matrix pipe is busy anyway): do the V waves' instructions then stream through, and does the MFMA rate survive?
template <int NOPS, int SLEEP>
__global__ __launch_bounds__(512, 2) void k_yield(int nm, int nv, long long* out, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    const long long t0 = clock64();
    float r = 0.f;
    if (wave < 4) {
        f32x16 acc[8];
        for (int q = 0; q < 8; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
        const float a = (float)lane * 1e-3f, b = 1.0f;
        for (int it = 0; it < nm; it += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
                if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
#pragma unroll
                for (int k = 0; k < NOPS; ++k) asm volatile("s_nop 15");
            }
        }
        for (int q = 0; q < 8; ++q) r += acc[q][0] + acc[q][15];
    } else {
        float v[8];
        for (int q = 0; q < 8; ++q) v[q] = (float)(lane + q);
        for (int it = 0; it < nv; it += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(v[(q + 1) & 7]), "v"(v[(q + 3) & 7]));
        }
        for (int q = 0; q < 8; ++q) r += v[q];
    }
    const long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 123.456f) *sink = r;
}

// dependent accumulation chains: NCH independent accumulators per wave, WPS waves per SIMD (1 or 2): cycles per MFMA
template <int NCH, int WPS>
__global__ __launch_bounds__(512, 2) void k_chain(int nm, long long* out, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (WPS == 1 && wave >= 4) return;
    f32x16 acc[NCH];
    for (int q = 0; q < NCH; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    const float a = (float)lane * 1e-3f, b = 1.0f;
    const long long t0 = clock64();
    for (int it = 0; it < nm; it += NCH) {
#pragma unroll
        for (int q = 0; q < NCH; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    }
    float r = 0.f;
    for (int q = 0; q < NCH; ++q) r += acc[q][0] + acc[q][15];
    const long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 123.456f) *sink = r;
}

int main() {
    long long* out; float* sink; CK(hipMalloc(&out, 256 * 8 * 8)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int pv, int pm, int mode, int nm, int nv) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (pv == 0 && pm == 0) hipLaunchKernelGGL((k_pipe<0, 0>), dim3(256), dim3(512), 0, 0, mode, nm, nv, out, sink);
            else if (pv == 3 && pm == 0) hipLaunchKernelGGL((k_pipe<3, 0>), dim3(256), dim3(512), 0, 0, mode, nm, nv, out, sink);
            else hipLaunchKernelGGL((k_pipe<0, 3>), dim3(256), dim3(512), 0, 0, mode, nm, nv, out, sink);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        std::vector<long long> h(256 * 8); CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
        m /= 1024; v /= 1024;
        printf("%-58s kernel %8.2f us | M waves %9.0f ticks (%.2f per MFMA) | second waves %9.0f ticks (%.3f per VALU)\n", name, best * 1e3, m, m / nm, v, nv ? v / nv : 0.0);
    };
    const int NM = 4096, NV = 4096;
    run("M alone (one wave per SIMD)", 0, 0, 0, NM, 0);
    run("M on both waves of every SIMD", 0, 0, 2, NM, 0);
    run("V alone: second waves only (M waves: 8 MFMAs)", 0, 0, 1, 8, NV);
    run("M + V, no priorities", 0, 0, 1, NM, NV);
    run("M + V, V waves at priority 3", 3, 0, 1, NM, NV);
    run("M + V, M waves at priority 3", 0, 3, 1, NM, NV);
    run("M + V (V short: 512), no priorities", 0, 0, 1, NM, 512);
    run("M + V (V short: 512), V at priority 3", 3, 0, 1, NM, 512);
    run("M + V (V long: 32768), no priorities", 0, 0, 1, NM, 32768);
    run("M + V (V long: 32768), V at priority 3", 3, 0, 1, NM, 32768);
    auto shadow = [&](const char* name, auto kern, int two) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, NM, out, sink); CK(hipDeviceSynchronize());
        std::vector<long long> h(256 * 8); CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        double m = 0; int n = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < (two ? 8 : 4); ++w) { m += (double)h[b * 8 + w]; ++n; }
        printf("%-58s %.2f ticks per MFMA\n", name, m / n / NM);
    };
    shadow("one wave per SIMD: MFMA + 0 VALU", k_shadow<0, 0>, 0);
    shadow("one wave per SIMD: MFMA + 2 VALU", k_shadow<2, 0>, 0);
    shadow("one wave per SIMD: MFMA + 4 VALU", k_shadow<4, 0>, 0);
    shadow("one wave per SIMD: MFMA + 8 VALU", k_shadow<8, 0>, 0);
    shadow("one wave per SIMD: MFMA + 12 VALU", k_shadow<12, 0>, 0);
    shadow("one wave per SIMD: MFMA + 16 VALU", k_shadow<16, 0>, 0);
    shadow("two waves per SIMD: MFMA + 0 VALU", k_shadow<0, 1>, 1);
    shadow("two waves per SIMD: MFMA + 2 VALU", k_shadow<2, 1>, 1);
    shadow("two waves per SIMD: MFMA + 4 VALU", k_shadow<4, 1>, 1);
    shadow("two waves per SIMD: MFMA + 8 VALU", k_shadow<8, 1>, 1);
    shadow("two waves per SIMD: MFMA + 16 VALU", k_shadow<16, 1>, 1);
    auto chain = [&](const char* name, auto kern, int wps) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, NM, out, sink); CK(hipDeviceSynchronize());
        std::vector<long long> h(256 * 8); CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        double m = 0; int n = 0, mx = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 4 * wps; ++w) { m += (double)h[b * 8 + w]; ++n; }
        printf("%-58s %.2f ticks per MFMA of a wave\n", name, m / n / NM);
    };
    chain("1 wave per SIMD, 1 accumulator chain", k_chain<1, 1>, 1);
    chain("1 wave per SIMD, 2 accumulator chains", k_chain<2, 1>, 1);
    chain("1 wave per SIMD, 4 accumulator chains", k_chain<4, 1>, 1);
    chain("2 waves per SIMD, 1 accumulator chain each", k_chain<1, 2>, 2);
    chain("2 waves per SIMD, 2 accumulator chains each", k_chain<2, 2>, 2);
    auto yield = [&](const char* name, auto kern, int nv) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, NM, nv, out, sink); CK(hipDeviceSynchronize());
        std::vector<long long> h(256 * 8); CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
        m /= 1024; v /= 1024;
        printf("%-58s M %.2f ticks per MFMA | V: %d VALU in %.0f ticks = %.2f VALU per MFMA time of M\n", name, m / NM, nv, v, nv / (v / (m / NM)));
    };
    yield("M + V, M yields 0 nops", k_yield<0, 0>, 16384);
    yield("M + V, M yields 1 x s_nop 15", k_yield<1, 0>, 16384);
    yield("M + V, M yields 2 x s_nop 15", k_yield<2, 0>, 16384);
    yield("M + V, M yields 3 x s_nop 15", k_yield<3, 0>, 32768);
    yield("M + V, M yields 4 x s_nop 15", k_yield<4, 0>, 32768);
    yield("M + V, M yields s_sleep 1", k_yield<0, 1>, 32768);
    yield("M alone, 3 x s_nop 15", k_yield<3, 0>, 8);
    yield("M alone, 4 x s_nop 15", k_yield<4, 0>, 8);
    return 0;
}
