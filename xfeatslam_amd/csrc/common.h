// common.h -- shared declarations of libxfeat_hip.so (gfx950 only, no CPU fallback).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/xfeat_hip.h"
#include "../../include/xfeat_hip_bench.h"

#define XFH_NUM_LAYERS 23
#define XFH_DESC_DIM 64
#define CAND_CNT_STRIDE 64   // ints between per-frame candidate counters: one 256-B line each (atomics on one line serialise)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

// XFeatModel::XFeatModel (reference src/XFeat.cc:41-90): BasicLayer(cin, cout, k, stride)
struct LayerSpec { int cin, cout, ks, stride; const char* name; };
static const LayerSpec XFH_LAYERS[XFH_NUM_LAYERS] = {
    {1, 4, 3, 1, "block1.0"},   {4, 8, 3, 2, "block1.1"},   {8, 8, 3, 1, "block1.2"},
    {8, 24, 3, 2, "block1.3"},  {24, 24, 3, 1, "block2.0"}, {24, 24, 3, 1, "block2.1"},
    {24, 64, 3, 2, "block3.0"}, {64, 64, 3, 1, "block3.1"}, {64, 64, 1, 1, "block3.2"},
    {64, 64, 3, 2, "block4.0"}, {64, 64, 3, 1, "block4.1"}, {64, 64, 3, 1, "block4.2"},
    {64, 128, 3, 2, "block5.0"}, {128, 128, 3, 1, "block5.1"}, {128, 128, 3, 1, "block5.2"},
    {128, 64, 1, 1, "block5.3"},
    {64, 64, 3, 1, "block_fusion.0"}, {64, 64, 3, 1, "block_fusion.1"},
    {64, 64, 1, 1, "heatmap_head.0"}, {64, 64, 1, 1, "heatmap_head.1"},
    {64, 64, 1, 1, "keypoint_head.0"}, {64, 64, 1, 1, "keypoint_head.1"},
    {64, 64, 1, 1, "keypoint_head.2"},
};

// monotone float -> uint map (larger float => larger uint) and its inverse
__host__ __device__ inline unsigned f2ord(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(unsigned u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __builtin_bit_cast(float, u);
}

// exp() as libtorch's CPU softmax / sigmoid kernels evaluate it (Vectorized<float>::exp = Sleef expf u10, FMA form; reached by the
// reference through torch::sigmoid, src/XFeat.cc:82, and F::softmax, src/XFextractor.cc:207): q = rint(d * log2 e), two-fma
// Cody-Waite reduction, degree-6 Horner chain in fma, 1 + (s*s*u + s), scaling by 2^q as two exact multiplications, 0 below -104,
// inf above 100.  Every step is one IEEE fp32 operation, so the device result equals the oracle's xfo_expf -- and libtorch's own
// vector kernels -- bit for bit (rounds 1-4 called the device library's expf here: 1 ulp away from glibc's in 1 % of the elements,
// and both away from libtorch's).  Denormal results stay denormal (the kernels are compiled with fp32 denormals on, the gfx9 default).
__device__ __forceinline__ float xfh_expf(float d) {
    const float qf = __builtin_rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    const int q = (int)__builtin_fminf(__builtin_fmaxf(qf, -300.f), 300.f);
    float s = __builtin_fmaf(qf, -0.693145751953125f, d);
    s = __builtin_fmaf(qf, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
    u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
    u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
    u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
    u = __builtin_fmaf(u, s, 0.5f);
    u = 1.0f + __builtin_fmaf(s * s, u, s);
    u = u * __builtin_bit_cast(float, (unsigned)((q >> 1) + 127) << 23) * __builtin_bit_cast(float, (unsigned)((q - (q >> 1)) + 127) << 23);
    if (d < -104.f) u = 0.f;
    if (d > 100.f) u = __builtin_inff();
    return u;
}

// XFH_MFMA_SETTLE: the boundary between a K loop and its epilogue.  Rounds 2-5 padded it by hand (s_nop 15; s_nop 3 = 20 wait states) after a stale accumulator
// read had been seen in k_conv_mfma_p<8,24,...> behind a taken branch, where hipcc's hazard recogniser had left 10 wait states.  Round 6 measured the hardware
// instead of the compiler (tests/cpp/hazard_probe.hip: the last MFMA, N = 0..20 wait states, an optional taken s_cbranch and the first read of the result in ONE
// inline-asm statement the recogniser cannot see into): gfx950 interlocks the dependency -- VALU, global_store and ds_write reads of the result of both MFMA forms the
// library issues are correct with ZERO wait states, with and without the branch (profiles/r06_hazard_probe.log) -- and the whole GPU suite, campaign and goldens
// included, is green without the pad (profiles/r06_nosettle.md: 203 tests; one serial 256-frame step 121.0 -> 120.0 ms of kernel time).  What had been seen was
// therefore not this hazard; the pad is gone.  The scheduling barriers stay: they cost nothing and keep epilogue instructions out of the K loop's tail.
// -DXFH_SETTLE_NOPS restores the old pad (A/B builds: make variant_all DEFS=-DXFH_SETTLE_NOPS).
#ifdef XFH_SETTLE_NOPS
#define XFH_MFMA_SETTLE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define XFH_MFMA_SETTLE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

// ---- statistics finalisation ------------------------------------------------------------
// Fold the npart (sum, sum of squares) fp64 partials of every channel in a FIXED order into
// (mean, rstd).  Called by k_bn_finalize (one workgroup per frame) and, for small batches, by
// every workgroup of the CONSUMING convolution (which saves the dependent finalize launch); the
// order does not depend on the number of threads, so all callers produce the same bits.
// red: >= 512 doubles of LDS; stat: 2*C floats (LDS or global); C <= 256.  Ends with a barrier.
// Weights that a whole wave reads at the same address go through the CONSTANT address space: the compiler then fetches them with
// scalar loads (SGPR operands of the FMAs).  Through a plain global pointer it cannot prove the memory read-only and issues one
// global_load_dwordx4 per weight quad with 64 identical lane addresses -- in k_conv_direct 144 of them per wave, which kept the
// texture address unit busy for longer than the 576 FMAs took (5 -> 2 us of the kernel's single-frame time).  Only for memory that no
// kernel writes (the weight blob, uploaded at xfh_load_weights).
#define XFH_CONST __attribute__((address_space(4)))
// BN_FOLD_BATCH: 8 costs the registers of the loop it replaces (kernels that also serve large batches sit at an occupancy edge:
// k_conv_mfma's PRO_FUSE instance went from 118 to 130 VGPRs with 16, i.e. from two workgroups per CU to one); the single-frame
// kernels take 16
template <int BN_FOLD_BATCH = 8>
__device__ __forceinline__ void bn_fold(const double* __restrict__ p, int npart, int C, double count, float* stat, double* red, int t, int nthr) {
    const int SL = 256 / C;
    for (int idx = t; idx < SL * C; idx += nthr) {
        const int c = idx % C, j = idx / C;
        double s = 0.0, ss = 0.0;
        // BN_FOLD_BATCH pairs of loads in flight, summed in index order.  The last batch is padded with clamped addresses whose values are
        // replaced by +0.0 (x + 0.0 == x bit for bit): a remainder loop would walk its loads one memory round trip after the other, and at
        // small batches this fold is on the critical path of every kernel (20 partials = 5 dependent trips before, 1 now)
        for (int q = j; q < npart; q += BN_FOLD_BATCH * SL) {
            double v[BN_FOLD_BATCH], w[BN_FOLD_BATCH];
#pragma unroll
            for (int u = 0; u < BN_FOLD_BATCH; ++u) {
                const int qq = min(q + u * SL, npart - 1);
                v[u] = p[((size_t)qq * C + c) * 2 + 0];
                w[u] = p[((size_t)qq * C + c) * 2 + 1];
            }
#pragma unroll
            for (int u = 0; u < BN_FOLD_BATCH; ++u) {
                const bool in = q + u * SL < npart;
                s += in ? v[u] : 0.0; ss += in ? w[u] : 0.0;
            }
        }
        red[idx * 2 + 0] = s;
        red[idx * 2 + 1] = ss;
    }
    __syncthreads();
    for (int c = t; c < C; c += nthr) {
        double S = 0.0, SS = 0.0;
        for (int q = 0; q < SL; ++q) { S += red[(q * C + c) * 2 + 0]; SS += red[(q * C + c) * 2 + 1]; }
        const double mean = S / count;
        double var = SS / count - mean * mean;
        if (var < 0.0) var = 0.0;
        // stored as (beta, alpha) = (-mean * rstd, rstd), both fp32: the layer is applied as ONE fma, x * alpha + beta -- the arithmetic of ATen's
        // batch_norm / instance_norm on the CPU (alpha = invstd, beta = bias - mean * alpha, out = fma(x, alpha, beta): bit for bit on every element,
        // tests/test_oracle.py::test_norm_apply_is_atens_fma), and one instruction instead of two on the pipe the MFMAs use
        const float rstd = (float)(1.0 / sqrt(var + 1e-5));
        stat[c] = -((float)mean * rstd);
        stat[C + c] = rstd;
    }
    __syncthreads();
}

// statistics of a producer layer as a consumer sees them: either finalised (stat: [B][2*C]: beta = -mean * rstd, alpha = rstd) or, for small
// batches, still as fp64 partials that every workgroup of the consumer folds itself (bn_fold; the same order, hence the
// same bits as k_bn_finalize) -- workgroup 0 of each frame publishes the result to stat_out for the debug API and for
// consumers that do not fold
struct StatSrc {
    const float* stat;
    const double* part; size_t part_stride; int npart; double count; float* stat_out;
};

// statistics of frame b into LDS (2*C floats): folded here from the partials, or copied.  `red`: >= 512 doubles of LDS
// scratch (only touched when folding).  Ends with a barrier.
template <int BATCH = 8>
__device__ __forceinline__ void stage_stat(const StatSrc& st, int b, int C, bool publish, float* s_stat, double* red, int t, int nthr) {
    if (st.part) {
        bn_fold<BATCH>(st.part + (size_t)b * st.part_stride, st.npart, C, st.count, s_stat, red, t, nthr);
        if (publish) for (int q = t; q < 2 * C; q += nthr) st.stat_out[(size_t)b * 2 * C + q] = s_stat[q];
    } else {
        const float* g = st.stat + (size_t)b * 2 * C;
        for (int q = t; q < 2 * C; q += nthr) s_stat[q] = g[q];
        __syncthreads();
    }
}

// per-frame output record header (include/xfeat_hip.h, xfh_record_bytes)
struct RecordHeader { int32_t n_valid, mono_index, n_candidates, reserved; };

// ---- kernel timing hook ---------------------------------------------------------------
struct KTimer {
    int kernel_id = 0;
    unsigned layer_mask = 0;
    int launches = 0;
    static const int MAXEV = 4096;   // launches recorded between xfh_timing_enable and xfh_timing_read; later ones are counted in `dropped`
    hipEvent_t* ev = nullptr;   // 2*MAXEV events, lazily created
    int dropped = 0;
    int nev = 0;
};

// ---- matcher workspace ----------------------------------------------------------------
struct MatchWs {
    float* img1 = nullptr; float* img2 = nullptr; size_t cap_p1 = 0, cap_p2 = 0;   // panel images of the two sets (capacity in panels of 256 rows)
    u64* keys = nullptr; u64* partR = nullptr; u64* partC = nullptr; u64* pairs = nullptr;   // one allocation: arg-max key planes (one per GEMM block column / row), (column, value) pairs
    float* h_d1 = nullptr; float* h_d2 = nullptr; size_t cap_in = 0;   // device staging of host inputs
    int* o_buf = nullptr; size_t cap_out = 0;                          // device outputs of the host call: n, idx1[nm], idx2[nm], dist[nm]
    int* h_out = nullptr; size_t cap_hout = 0;                         // pinned mirror of o_buf
    int32_t* o_tab = nullptr; size_t cap_tab = 0;
    void* b2_buf = nullptr; size_t cap_b2 = 0;                             // staging for the host-pointer best2 call
    u64* bkeys = nullptr; size_t cap_bkeys = 0;                            // key planes + pairs of the many-pairs call (xfh_match_mnn_prepared_batch_device), grown on demand
};

// ---- launchers implemented in the .hip files ------------------------------------------
struct xfh_ctx;
hipError_t match_ws_reserve(xfh_ctx* c, int n1, int n2);
hipError_t launch_mnn(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, float min_cossim,
                      int* idx1, int* idx2, float* dist, int* n_matches);
hipError_t launch_match_prepare(xfh_ctx* c, const float* d, int n, float* img);
hipError_t bench_mnn_gemm(xfh_ctx* c, const float* img1, int n1, const float* img2, int n2, int iters, double* us_per_launch);
hipError_t launch_mnn_prepared(xfh_ctx* c, const float* img1, int n1, const float* img2, int n2, float min_cossim,
                               int* idx1, int* idx2, float* dist, int* n_matches, const int* hdr1 = nullptr, const int* hdr2 = nullptr);
struct MnnBatch;
hipError_t launch_mnn_gemm_seg(xfh_ctx* c, const MnnBatch& jb);           // kernels_mnn_gemm.hip: the persistent many-pairs GEMM (mnn_gemm_seg.hip.h)
struct XfhMatchPair {                                                     // one pair of a many-pairs call: prepared images in, match list out (device pointers)
    const float* img1; int n1; const float* img2; int n2;
    int* idx1; int* idx2; float* dist; int* n_matches;
};
hipError_t launch_mnn_batch(xfh_ctx* c, const XfhMatchPair* pairs, int n_pairs, float min_cossim);
hipError_t bench_mnn_gemm_batch(xfh_ctx* c, const XfhMatchPair* pairs, int n_pairs, int iters, double* us_per_launch, double* sclk_mhz);
hipError_t launch_mnn_gemm(xfh_ctx* c, const float* img1, int n1, const float* img2, int n2, u64* partR, size_t ldr, u64* partC, size_t ldc, u64* pairs);   // kernels_mnn_gemm.hip
hipError_t launch_dist_i32(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, int32_t* out);
hipError_t launch_distinctive(xfh_ctx* c, const float* table, const int* offsets, const int* indices, int n_groups, int max_group,
                              int* best_pos, int* best_median);
hipError_t launch_best2(xfh_ctx* c, const float* q, int nq, const float* tg, const int* offsets, const int* indices, int init_dist,
                        int* best_idx, int* best_dist, int* second_idx, int* second_dist);
