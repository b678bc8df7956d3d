// heads_kp4.hip.h -- keypoint_head.3 + softmax + depth-to-space for small batches, as a device function: it is a kernel of its own
// (k_heads_kp4, kernels_misc.hip) and a RIDER on a convolution launch of the backbone (k_conv_mfma_ride, kernels_conv.hip).
#pragma once
#include "common.h"

// IEEE a / d for several a and one d: the refined reciprocal and the quotient correction of the compiler's own division
// expansion (v_rcp + Newton step, q = a*r, two residual corrections), with the reciprocal shared.  Bit-identical to a / d
// whenever the expansion's scaling stage is the identity: d in [1e-12, 1e6], |a| <= 1e6 and either zero or above 1e-20
// (k_desc: always; k_heads_kp: softmax terms below 1e-20 may differ in their last denormal bits).
struct Recip { float d, r; };
__device__ __forceinline__ Recip recip_of(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
    return Recip{d, r};
}
__device__ __forceinline__ float div_by(float a, const Recip& k) {
    float q = a * k.r;
    q = fmaf(fmaf(-k.d, q, a), k.r, q);
    return fmaf(fmaf(-k.d, q, a), k.r, q);
}


struct Kp4Args {
    const float* rawK; StatSrc sK; size_t raw_stride;      // keypoint_head.2 (raw map + its statistics)
    const float* wk /* [64][68] */; const float* bk /* [65] */;
    int Hh, Wh; float* K1h; size_t k1h_stride;
};

// k_heads_kp4: the small-batch form (B <= 8) of k_heads_kp.  One frame gives k_heads_kp 38 workgroups of two waves, and every lane
// walks 64 x 65 dependent-free but sequentially issued fmas plus 65 expf and 64 quotients: 35 us of a 0.37-ms frame on a mostly
// idle GPU.  Here FOUR lanes share a pixel (lane j takes outputs 16 j .. 16 j + 15, lane 3 also the dustbin), a workgroup is 32
// pixels, so four times as many waves run chains a quarter as long.  Every output is the same fma chain over k, the maximum is exact
// in any order, and the softmax sum keeps the reference order n = 0 .. 64: the running sum travels from lane j to lane j + 1 by shuffle
// before lane j + 1 adds its terms -- the result is bit for bit k_heads_kp's (tests/test_gpu_extract.py::test_batch_is_per_frame
// compares the two).  The weights are lane-dependent now, so they come from LDS ([k][68], broadcast over the pixels) instead of the
// scalar cache.
#define HK4_PX 32
#define HK4_LD 33
#define HK4_LDS_FLOATS (64 * 68 + 64 * HK4_LD + 128 + 1024)
// the first 4 * HK4_PX threads of the workgroup (any others must have left the kernel); lds: HK4_LDS_FLOATS floats, 16-byte aligned
__device__ __forceinline__ void heads_kp4_body(const Kp4Args ka, int block, int b, float* lds) {      // (by value: through a reference the kernel arguments stop being scalar loads)
    const float* __restrict__ rawK = ka.rawK; const StatSrc& sK = ka.sK; const size_t raw_stride = ka.raw_stride;
    const float* __restrict__ wk = ka.wk; const float* __restrict__ bk = ka.bk;
    const int Hh = ka.Hh, Wh = ka.Wh; float* __restrict__ K1h = ka.K1h; const size_t k1h_stride = ka.k1h_stride;
    float* sW = lds;                                   // [64][68]
    float* sA = sW + 64 * 68;                          // [64][HK4_LD]
    float* st = sA + 64 * HK4_LD;                      // 128
    double* red = (double*)(st + 128);                 // 512 doubles (64 * 68 + 64 * 33 + 128 floats = 6592: 8-byte aligned)
    const int t = threadIdx.x;
    const int npix = Hh * Wh, p0 = block * HK4_PX;
    const int lp = t >> 2, j = t & 3, pix = p0 + lp;
    // raw values of the 32 pixels (4 float4 per thread) and the weights, all loads in flight before the statistics are staged
    f32x4 rv[4], wv[9];
    {
        const float* rp = rawK + (size_t)b * raw_stride;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int item = t + k * 4 * HK4_PX, ip = item >> 4, g = item & 15;
            rv[k] = *(const f32x4*)(rp + (size_t)min(p0 + ip, npix - 1) * 64 + g * 4);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) { const int f = t + k * 4 * HK4_PX; wv[k] = *(const f32x4*)(wk + (size_t)min(f, 64 * 17 - 1) * 4); }
    }
    stage_stat(sK, b, 64, block == 0, st, red, t, 4 * HK4_PX);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int item = t + k * 4 * HK4_PX, ip = item >> 4, g = item & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q) sA[(g * 4 + q) * HK4_LD + ip] = fmaxf(fmaf(rv[k][q], st[64 + g * 4 + q], st[g * 4 + q]), 0.f);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) { const int f = t + k * 4 * HK4_PX; if (f < 64 * 17) *(f32x4*)(sW + f * 4) = wv[k]; }
    __syncthreads();
    float acc[17];
#pragma unroll
    for (int n = 0; n < 17; ++n) acc[n] = 0.f;
    const float* wj = sW + 16 * j;
#pragma unroll 2
    for (int k = 0; k < 64; ++k) {
        const float a = sA[k * HK4_LD + lp];
        const f32x4 w0 = *(const f32x4*)(wj + k * 68), w1 = *(const f32x4*)(wj + k * 68 + 4), w2 = *(const f32x4*)(wj + k * 68 + 8), w3 = *(const f32x4*)(wj + k * 68 + 12);
        const float wd = sW[k * 68 + 64];                // the dustbin column (used by lane 3)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[q] = fmaf(a, w0[q], acc[q]); acc[4 + q] = fmaf(a, w1[q], acc[4 + q]);
            acc[8 + q] = fmaf(a, w2[q], acc[8 + q]); acc[12 + q] = fmaf(a, w3[q], acc[12 + q]);
        }
        acc[16] = fmaf(a, wd, acc[16]);
    }
    const int nown = j == 3 ? 17 : 16;                   // lane 3: outputs 48 .. 64
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int n = 0; n < 17; ++n) {
        acc[n] += bk[min(16 * j + n, 64)];
        if (n < nown) mx = fmaxf(mx, acc[n]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2));
#pragma unroll
    for (int n = 0; n < 17; ++n) acc[n] = xfh_expf(acc[n] - mx);
    // sum over n = 0 .. 64 in that order: lane r continues the sum lane r - 1 has reached
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float sr = __shfl(sum, (t & ~3) | (r > 0 ? r - 1 : 0));       // (round 0 starts from 0)
        if (r == 0) sr = 0.f;
#pragma unroll
        for (int n = 0; n < 17; ++n) if (n < 16 || r == 3) sr += acc[n];
        if (j == r) sum = sr;
    }
    sum = __shfl(sum, t | 3);                            // the total sits in lane 3
    const Recip ks = recip_of(sum);                      // 64 softmax quotients share the divisor
    if (pix < npix) {
        const int y = pix / Wh, x = pix % Wh;
        // outputs 16 j .. 16 j + 15 = rows 2 j, 2 j + 1 of the pixel's 8 x 8 cell (depth-to-space)
        float* o = K1h + (size_t)b * k1h_stride + (size_t)(8 * y + 2 * j) * (8 * Wh) + 8 * x;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(f32x4*)(o + (size_t)i * 8 * Wh) = f32x4{div_by(acc[i * 8], ks), div_by(acc[i * 8 + 1], ks), div_by(acc[i * 8 + 2], ks), div_by(acc[i * 8 + 3], ks)};
            *(f32x4*)(o + (size_t)i * 8 * Wh + 4) = f32x4{div_by(acc[i * 8 + 4], ks), div_by(acc[i * 8 + 5], ks), div_by(acc[i * 8 + 6], ks), div_by(acc[i * 8 + 7], ks)};
        }
    }
}

