// kernels_misc.hip -- everything around the convolutions of XFextractor::operator()
// (reference src/XFextractor.cc:250-356) on gfx950:
//   k_preproc      parseInput + preprocessTensor (:161-202) and InstanceNorm partial sums
//   k_norm_aux     InstanceNorm apply, unfold2d(x,8) (XFeat.cc:124-133), AvgPool4 of skip1 (:36-39)
//   (x1 + skip1(x), XFeat.cc:153, and x3 + up(x4) + up(x5), :159-166, are computed by the consuming convolutions while
//    they stage their input: kernels_conv.hip PRO_B2IN / PRO_FUSE; F::normalize(M1, dim=1), XFextractor.cc:273, is
//    applied per bilinear tap inside k_desc)
//   k_heads_heat   heatmap_head.2 + sigmoid; k_heads_kp: keypoint_head.3 + softmax + depth-to-space
//                  (XFeat.cc:81-82,89; XFextractor.cc:204-217)
//   k_nms_score    5x5 NMS, threshold, nearest*bilinear score (XFextractor.cc:219-248, 280-282)
//   k_select       top-k by (score desc, index asc), validity, lapping placement (:285-295, 310-343)
//   k_desc         F::normalize of the sampled feature pixels (:273), bilinear descriptor sampling + L2 normalise + record packing (:298-301, 323-343)
// Compiled with -ffp-contract=off: every fused multiply-add below is written as fmaf().
#include "ctx.h"
#include "mnn_layout.h"
#include "heads_kp4.hip.h"
#include <stdlib.h>

// ---- small helpers -----------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
    v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}

// ATen upsample_bilinear2d (align_corners=false) source index / weights; see oracle lin_coeff
__device__ __forceinline__ void lin_coeff(int in, int out, int d, int& i0, int& i1, float& l0, float& l1) {
    const float scale = (float)in / (float)out;
    float src = fmaf(scale, (float)d + 0.5f, -0.5f);
    if (src < 0.f) src = 0.f;
    int a = (int)src;
    if (a > in - 1) a = in - 1;
    float lam = src - (float)a;
    lam = fminf(fmaxf(lam, 0.f), 1.f);
    i0 = a; i1 = a + ((a < in - 1) ? 1 : 0);
    l1 = lam; l0 = 1.f - lam;
}

// normgrid (XFeat.cc:181-186) + grid_sample unnormalise (align_corners=false, ATen CPU form)
__device__ __forceinline__ float grid_coord(int pos, int full, int size) {
    const float g = 2.0f * ((float)pos / (float)(full - 1)) - 1.0f;
    return (g + 1.0f) * ((float)size / 2.0f) - 0.5f;
}

// ---- k_preproc ---------------------------------------------------------------------------
// grid (ceil(H*W/1024), 1, B), 256 threads, 4 consecutive pixels per thread.
__global__ __launch_bounds__(256)
void k_preproc(const uint8_t* __restrict__ gray, size_t gray_stride, int H0, int W0, int H, int W,
               float* __restrict__ X, size_t x_stride, double* __restrict__ part, int npart, int* __restrict__ cand_count) {
    __shared__ double red[8];
    const int t = threadIdx.x, b = blockIdx.z;
    if (blockIdx.x == 0 && t == 0) cand_count[b * CAND_CNT_STRIDE] = 0;     // consumed by k_nms_score much later in the stream
    const uint8_t* g = gray + (size_t)b * gray_stride;
    const int p0 = (blockIdx.x * 256 + t) * 4;
    double s = 0.0, ss = 0.0;
    if (p0 < H * W) {
        const int y = p0 / W, x = p0 % W;
        f32x4 v;
        if (H == H0 && W == W0) {
            const uchar4 u = *(const uchar4*)(g + (size_t)y * W0 + x);
            v = f32x4{(float)u.x / 255.0f, (float)u.y / 255.0f, (float)u.z / 255.0f, (float)u.w / 255.0f};
        } else {
            int y0, y1; float hy0, hy1;
            lin_coeff(H0, H, y, y0, y1, hy0, hy1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int x0, x1; float wx0, wx1;
                lin_coeff(W0, W, x + q, x0, x1, wx0, wx1);
                const float p00 = (float)g[(size_t)y0 * W0 + x0] / 255.0f, p01 = (float)g[(size_t)y0 * W0 + x1] / 255.0f;
                const float p10 = (float)g[(size_t)y1 * W0 + x0] / 255.0f, p11 = (float)g[(size_t)y1 * W0 + x1] / 255.0f;
                const float top = fmaf(wx0, p00, wx1 * p01), bot = fmaf(wx0, p10, wx1 * p11);
                v[q] = fmaf(hy0, top, hy1 * bot);
            }
        }
        *(f32x4*)(X + (size_t)b * x_stride + p0) = v;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const double d = (double)v[q]; s += d; ss = fma(d, d, ss); }
    }
    s = wave_sum(s); ss = wave_sum(ss);
    if ((t & 63) == 0) { red[(t >> 6) * 2] = s; red[(t >> 6) * 2 + 1] = ss; }
    __syncthreads();
    if (t == 0) {
        double* p = part + ((size_t)b * npart + blockIdx.x) * 2;
        p[0] = (red[0] + red[2]) + (red[4] + red[6]);
        p[1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}

// ---- k_norm_aux: one thread per 4x4 pixel block -------------------------------------------
// xs: InstanceNorm statistics of the image; for small batches every workgroup folds k_preproc's partials itself
// (workgroup 0 publishes them for block1.0), which saves the finalize launch
__global__ __launch_bounds__(256)
void k_norm_aux(const float* __restrict__ X, size_t x_stride, StatSrc xs, int H, int W,
                float* __restrict__ pool, size_t pool_stride) {
    __shared__ double red[512];
    __shared__ float s_stat[2];
    const int b = blockIdx.z;
    stage_stat(xs, b, 1, blockIdx.x == 0, s_stat, red, threadIdx.x, 256);
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int w4 = W / 4, h4 = H / 4;
    if (q >= w4 * h4) return;
    const int by = q / w4, bx = q % w4;
    const float m = s_stat[0], r = s_stat[1];
    const float* x = X + (size_t)b * x_stride;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = by * 4 + i, xx = bx * 4;
        f32x4 v = *(const f32x4*)(x + (size_t)y * W + xx);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = fmaf(v[j], r, m); s += v[j]; }
    }
    pool[(size_t)b * pool_stride + q] = s / 16.0f;
}

// ---- k_heads_heat / k_heads_kp: 128 pixels per workgroup, one pixel per lane -----------------------------
#define HF_PX 128
#define HF_LD 129
// heatmap head: 64 -> 1, sigmoid (XFeat.cc:81-82)
__global__ __launch_bounds__(HF_PX)
void k_heads_heat(const float* __restrict__ rawH, StatSrc sH,     // heatmap_head.1
                  size_t raw_stride, const float* __restrict__ wh, const float* __restrict__ bh,
                  int npix, float* __restrict__ H1, size_t h1_stride) {
    __shared__ float sA[64 * HF_LD];
    __shared__ float st[128];
    const int t = threadIdx.x, b = blockIdx.z;
    const int p0 = blockIdx.x * HF_PX;
    const int pix = p0 + t;
    // the workgroup's 128 x 64 raw values: all sixteen loads of a thread are issued up front (clamped addresses, no branch around
    // them -- behind a branch they ran as sixteen dependent round trips: 36 us of this kernel's single-frame time), the statistics
    // are staged while they fly
    f32x4 rv[16];
    {
        const float* rp = rawH + (size_t)b * raw_stride;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int item = t + k * HF_PX, lp = item >> 4, g = item & 15;
            rv[k] = *(const f32x4*)(rp + (size_t)min(p0 + lp, npix - 1) * 64 + g * 4);
        }
    }
    stage_stat(sH, b, 64, blockIdx.x == 0, st, (double*)sA, t, HF_PX);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int item = t + k * HF_PX, lp = item >> 4, g = item & 15;
#pragma unroll
        for (int j = 0; j < 4; ++j) sA[(g * 4 + j) * HF_LD + lp] = fmaxf(fmaf(rv[k][j], st[64 + g * 4 + j], st[g * 4 + j]), 0.f);
    }
    __syncthreads();
    float acc = 0.f;
    for (int k = 0; k < 64; ++k) acc = fmaf(sA[k * HF_LD + t], wh[k], acc);
    acc += bh[0];
    if (pix < npix) H1[(size_t)b * h1_stride + pix] = 1.0f / (1.0f + xfh_expf(0.f - acc));
}

// keypoint head: 64 -> 65, softmax, drop dustbin, depth-to-space (XFeat.cc:89, XFextractor.cc:204-217).
// Runs on the ctx's second stream together with keypoint_head.0-2: the whole branch only depends on the
// normalised image, not on the backbone.
__global__ __launch_bounds__(HF_PX)
void k_heads_kp(const float* __restrict__ rawK, StatSrc sK,      // keypoint_head.2
                size_t raw_stride, const float* __restrict__ wk /* [64][68] */, const float* __restrict__ bk /* [65] */,
                int Hh, int Wh, float* __restrict__ K1h, size_t k1h_stride) {
    __shared__ float sA[64 * HF_LD];
    __shared__ float st[128];
    const int t = threadIdx.x, b = blockIdx.z;
    const int npix = Hh * Wh, p0 = blockIdx.x * HF_PX;
    const int pix = p0 + t;
    // the workgroup's 128 x 64 raw values: all sixteen loads of a thread are issued up front (clamped addresses, no branch around
    // them -- behind a branch they ran as sixteen dependent round trips: 36 us of this kernel's single-frame time), the statistics
    // are staged while they fly
    f32x4 rv[16];
    {
        const float* rp = rawK + (size_t)b * raw_stride;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int item = t + k * HF_PX, lp = item >> 4, g = item & 15;
            rv[k] = *(const f32x4*)(rp + (size_t)min(p0 + lp, npix - 1) * 64 + g * 4);
        }
    }
    stage_stat(sK, b, 64, blockIdx.x == 0, st, (double*)sA, t, HF_PX);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int item = t + k * HF_PX, lp = item >> 4, g = item & 15;
#pragma unroll
        for (int j = 0; j < 4; ++j) sA[(g * 4 + j) * HF_LD + lp] = fmaxf(fmaf(rv[k][j], st[64 + g * 4 + j], st[g * 4 + j]), 0.f);
    }
    __syncthreads();
    float acc[65];
#pragma unroll
    for (int n = 0; n < 65; ++n) acc[n] = 0.f;
#pragma unroll 2
    for (int k = 0; k < 64; ++k) {
        const float a = sA[k * HF_LD + t];
        const XFH_CONST float* wr = (const XFH_CONST float*)wk + k * 68;                 // wave-uniform row, constant address space: scalar loads
#pragma unroll
        for (int n = 0; n < 65; ++n) acc[n] = fmaf(a, wr[n], acc[n]);
    }
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int n = 0; n < 65; ++n) { acc[n] += bk[n]; mx = fmaxf(mx, acc[n]); }
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < 65; ++n) { acc[n] = xfh_expf(acc[n] - mx); sum += acc[n]; }
    const Recip ks = recip_of(sum);                       // 64 softmax quotients share the divisor
    if (pix < npix) {
        const int y = pix / Wh, x = pix % Wh;
        float* o = K1h + (size_t)b * k1h_stride + (size_t)(8 * y) * (8 * Wh) + 8 * x;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *(f32x4*)(o + (size_t)i * 8 * Wh) = f32x4{div_by(acc[i * 8], ks), div_by(acc[i * 8 + 1], ks), div_by(acc[i * 8 + 2], ks), div_by(acc[i * 8 + 3], ks)};
            *(f32x4*)(o + (size_t)i * 8 * Wh + 4) = f32x4{div_by(acc[i * 8 + 4], ks), div_by(acc[i * 8 + 5], ks), div_by(acc[i * 8 + 6], ks), div_by(acc[i * 8 + 7], ks)};
        }
    }
}

// k_heads_kp4 (heads_kp4.hip.h): the small-batch form of k_heads_kp as a kernel of its own (eval()-BatchNorm modes; in the reference's
// batch-statistics mode the same body rides on block3.0's launch, kernels_conv.hip)
__global__ __launch_bounds__(4 * HK4_PX)
void k_heads_kp4(Kp4Args k) {
    __shared__ __attribute__((aligned(16))) float smem_kp4[HK4_LDS_FLOATS];
    heads_kp4_body(k, blockIdx.x, blockIdx.z, smem_kp4);
}

// ---- k_feat_norm: ||feats(p)||_2 of every feature pixel, one thread per pixel -------------------
// F::normalize(M1, dim=1) (XFextractor.cc:273) divides every pixel of the feature map by max(||.||_2, 1e-12): the norm is the
// oracle's expression (fp64 sum of squares over the channels in order, fp32 sqrt, max) and is all that is stored -- the
// normalised map itself is never written, k_desc divides the four pixels a sample touches.  A thread walks the 64 channels
// of its pixel: 128 fp64 operations per pixel, where a cross-lane reduction per sampled pixel cost ten times that.
// For batches <= 8 the same blocks ride on the k_nms_score launch instead (its grid is extended by fn_blocks workgroups that take
// this path): the norms depend on feats only, and a launch of their own was 5 us on the critical path of a single frame.
__device__ __forceinline__ void feat_norm_px(const float* __restrict__ feats, size_t m_stride, int npix, float* __restrict__ nrm, size_t n_stride, int b, int p) {
    if (p >= npix) return;
    const float* m = feats + (size_t)b * m_stride + (size_t)p * 64;
    double ss = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const f32x4 v = *(const f32x4*)(m + g * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) ss = fma((double)v[q], (double)v[q], ss);
    }
    nrm[(size_t)b * n_stride + p] = fmaxf((float)sqrt(ss), 1e-12f);
}
__global__ __launch_bounds__(256)
void k_feat_norm(const float* __restrict__ feats, size_t m_stride, int npix, float* __restrict__ nrm, size_t n_stride) {
    feat_norm_px(feats, m_stride, npix, nrm, n_stride, blockIdx.z, blockIdx.x * 256 + threadIdx.x);
}

// ---- k_nms_score: 64x16 pixels per workgroup ---------------------------------------------------
// phase 1: the tile (+2 halo, fetched as aligned float4 with -inf outside the image, as max_pool2d
// pads) goes to LDS; every thread owns 4 consecutive pixels of one row and evaluates the 5x5 maximum
// from registers (15 ds_read_b128, ~60 VALU) and only records the (few) candidates;
// phase 2: the candidates are scored densely, one per lane (the score arithmetic -- divides,
// nearbyint, four taps -- would otherwise run mostly masked for every wave).
#define NMS_TW 64
#define NMS_TH 16
#define NMS_LD 72
#define NMS_HCW (NMS_TW / 8 + 2)
#define NMS_HCH (NMS_TH / 8 + 2)
#define NMS_HC (NMS_HCW * NMS_HCH)
template <bool HEAT>          // HEAT: the heatmap head is computed here (batches <= 8); otherwise k_heads_heat wrote H1 and none of its LDS is carried
__global__ __launch_bounds__(256)
void k_nms_score(const float* __restrict__ K1h, size_t k_stride, const float* __restrict__ H1, size_t h_stride,
                 int H, int W, float thr, u64* __restrict__ cand, size_t cand_cap, int* __restrict__ cand_count,
                 int nms_blocks, const float* __restrict__ fn_feats, size_t fn_m_stride, int fn_npix, float* __restrict__ fn_nrm, size_t fn_n_stride,
                 const float* __restrict__ rawH, StatSrc sH, size_t raw_stride, const float* __restrict__ wh, const float* __restrict__ bh, float* __restrict__ H1out) {
    __shared__ __attribute__((aligned(16))) float s[(NMS_TH + 4) * NMS_LD];
    __shared__ u64 keys[1024];
    __shared__ unsigned short cpx[1024];
    __shared__ int cnt, base;
    // rawH != null (batches <= 8): heatmap_head.2 + sigmoid (k_heads_heat) computed HERE for the NMS_HC cells whose reliability this
    // tile's scores can touch (the tile's 8 x 2 cells and a ring of one) -- 64 channels x 40 cells per workgroup, the same fma chain
    // per cell, instead of a launch of its own in front of this one (8 us on the critical path of a single frame)
    __shared__ float s_act[HEAT ? NMS_HC * 65 : 1];
    __shared__ float s_hst[HEAT ? 128 : 1];
    __shared__ float s_h1[HEAT ? NMS_HC : 1];
    const int t = threadIdx.x, b = blockIdx.z;
    if ((int)blockIdx.x >= nms_blocks) {          // riding feat-norm blocks (batches <= 8, see k_feat_norm)
        feat_norm_px(fn_feats, fn_m_stride, fn_npix, fn_nrm, fn_n_stride, b, ((int)blockIdx.x - nms_blocks) * 256 + t);
        return;
    }
    const int tiles_x = (W + NMS_TW - 1) / NMS_TW;
    const int tx0 = (blockIdx.x % tiles_x) * NMS_TW, ty0 = (blockIdx.x / tiles_x) * NMS_TH;
    const float* k = K1h + (size_t)b * k_stride;
    const float NEG = -__builtin_huge_valf();
    const int Wh = W >> 3, Hh = H >> 3;
    const int cxb = (tx0 >> 3) - 1, cyb = (ty0 >> 3) - 1;         // first cell column / row of the NMS_HCW x NMS_HCH cell window
    constexpr int NHL = HEAT ? (NMS_HC * 16 + 255) / 256 : 1;
    f32x4 hv[NHL];
    if constexpr (HEAT) {
        const float* rp = rawH + (size_t)b * raw_stride;
#pragma unroll
        for (int q = 0; q < NHL; ++q) {
            const int item = min(t + q * 256, NMS_HC * 16 - 1), cell = item >> 4, g = item & 15;
            const int cy = min(max(cyb + cell / NMS_HCW, 0), Hh - 1), cx = min(max(cxb + cell % NMS_HCW, 0), Wh - 1);
            hv[q] = *(const f32x4*)(rp + ((size_t)cy * Wh + cx) * 64 + g * 4);
        }
    }
    if (t == 0) cnt = 0;
    for (int item = t; item < (NMS_TH + 4) * (NMS_LD / 4); item += 256) {
        const int iy = item / (NMS_LD / 4), c4 = item % (NMS_LD / 4);
        const int gy = ty0 - 2 + iy, gx = tx0 - 4 + 4 * c4;              // W % 4 == 0: a float4 is all in or all out
        f32x4 v = {NEG, NEG, NEG, NEG};
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *(const f32x4*)(k + (size_t)gy * W + gx);
        *(f32x4*)(s + iy * NMS_LD + 4 * c4) = v;
    }
    if constexpr (HEAT) {
        stage_stat(sH, b, 64, blockIdx.x == 0, s_hst, (double*)keys, t, 256);      // (keys: free until the scoring phase; ends with a barrier)
#pragma unroll
        for (int q = 0; q < NHL; ++q) {
            const int item = t + q * 256, cell = item >> 4, g = item & 15;
            if (item < NMS_HC * 16)
#pragma unroll
                for (int j = 0; j < 4; ++j) s_act[cell * 65 + g * 4 + j] = fmaxf(fmaf(hv[q][j], s_hst[64 + g * 4 + j], s_hst[g * 4 + j]), 0.f);
        }
    }
    __syncthreads();
    if (HEAT && t < NMS_HC) {
        float acc = 0.f;
        for (int c = 0; c < 64; ++c) acc = fmaf(s_act[t * 65 + c], wh[c], acc);
        acc += bh[0];
        const float hval = 1.0f / (1.0f + xfh_expf(0.f - acc));
        s_h1[t] = hval;
        const int r = t / NMS_HCW, c = t % NMS_HCW, cy = cyb + r, cx = cxb + c;
        if (r >= 1 && r <= NMS_TH / 8 && c >= 1 && c <= NMS_TW / 8 && cy < Hh && cx < Wh) H1out[(size_t)b * h_stride + (size_t)cy * Wh + cx] = hval;   // the tile's own cells
    }
    {
        const int tx = t & 15, ty = t >> 4;
        float m[4] = {NEG, NEG, NEG, NEG}, ctr[4] = {NEG, NEG, NEG, NEG};
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const float* row = s + (ty + r) * NMS_LD + 4 * tx;
            const f32x4 f0 = *(const f32x4*)row, f1 = *(const f32x4*)(row + 4), f2 = *(const f32x4*)(row + 8);
            const float f[12] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float h = fmaxf(fmaxf(fmaxf(f[j + 2], f[j + 3]), fmaxf(f[j + 4], f[j + 5])), f[j + 6]);
                m[j] = fmaxf(m[j], h);
                if (r == 2) ctr[j] = f[j + 4];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (ty0 + ty < H && tx0 + 4 * tx + j < W && ctr[j] == m[j] && ctr[j] > thr)
                cpx[atomicAdd(&cnt, 1)] = (unsigned short)(ty * NMS_TW + 4 * tx + j);
    }
    __syncthreads();
    const int n = cnt;
    const float* h1 = H1 + (size_t)b * h_stride;
    auto h1_at = [&](int y, int x) -> float {        // valid cell (y, x): from the window computed above, or from the map k_heads_heat wrote
        if constexpr (HEAT) return s_h1[min(max(y - cyb, 0), NMS_HCH - 1) * NMS_HCW + min(max(x - cxb, 0), NMS_HCW - 1)];
        return h1[y * Wh + x];
    };
    for (int e = t; e < n; e += 256) {
        const int ly = cpx[e] / NMS_TW, lx2 = cpx[e] % NMS_TW;
        const int gy = ty0 + ly, gx = tx0 + lx2;
        // score = nearest(K1h) * bilinear(H1) at (x,y)   (XFextractor.cc:280)
        const float fx = nearbyintf(grid_coord(gx, W, W)), fy = nearbyintf(grid_coord(gy, H, H));
        float nv = 0.f;
        if (fx >= 0.f && fx <= (float)(W - 1) && fy >= 0.f && fy <= (float)(H - 1))
            nv = k[(size_t)(int)fy * W + (int)fx];
        const float ix = grid_coord(gx, W, Wh), iy = grid_coord(gy, H, Hh);
        const float xw = floorf(ix), yn = floorf(iy);
        const float w = ix - xw, ee = 1.0f - w, nn = iy - yn, so = 1.0f - nn;
        const float nw = ee * so, ne = w * so, sw = ee * nn, se = w * nn;
        const int x0 = (int)xw, y0 = (int)yn, x1 = x0 + 1, y1 = y0 + 1;
        const bool vx0 = x0 >= 0 && x0 < Wh, vx1 = x1 >= 0 && x1 < Wh, vy0 = y0 >= 0 && y0 < Hh, vy1 = y1 >= 0 && y1 < Hh;
        const float a = (vx0 && vy0) ? h1_at(y0, x0) : 0.f, bb = (vx1 && vy0) ? h1_at(y0, x1) : 0.f;
        const float d = (vx0 && vy1) ? h1_at(y1, x0) : 0.f, g = (vx1 && vy1) ? h1_at(y1, x1) : 0.f;
        const float hb = ((a * nw + bb * ne) + d * sw) + g * se;
        float score = nv * hb;
        if (gx == 0 && gy == 0) score = -1.0f;                       // :281-282
        // ascending key order == descending score, then ascending linear index (stable argsort)
        keys[e] = ((u64)(~f2ord(score)) << 32) | (u64)(unsigned)(gy * W + gx);
    }
    if (t == 0) base = atomicAdd(&cand_count[b * CAND_CNT_STRIDE], n);
    __syncthreads();
    for (int e = t; e < n; e += 256)
        if ((size_t)(base + e) < cand_cap) cand[(size_t)b * cand_cap + base + e] = keys[e];
}

// ---- k_select_generic: any nfeatures (fallback when nfeatures > 4096): full bitonic sort -----------
#define SEL_LDS_KEYS 16384
template <bool LDSMEM>
__device__ __forceinline__ void bitonic_sort(u64* a, int n, int t) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int idx = t; idx < (n >> 1); idx += 1024) {
                const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                const int l = i | j;
                const u64 x = a[i], y = a[l];
                const bool asc = (i & k) == 0;
                if ((x > y) == asc) { a[i] = y; a[l] = x; }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(1024)
void k_select_generic(u64* __restrict__ cand, size_t cand_cap, const int* __restrict__ cand_count, int W, int nfeatures,
              int lap0, int lap1, float rw, int* __restrict__ slot_src, u64* __restrict__ sel_key, int* __restrict__ sel_n,
              uint8_t* __restrict__ records, size_t rec_bytes) {
    extern __shared__ __attribute__((aligned(16))) u64 skeys[];
    __shared__ int wsumF[16], wsumB[16], wsumV[16];
    __shared__ int baseF, baseB, baseV;
    const int t = threadIdx.x, b = blockIdx.x, lane = t & 63, wave = t >> 6;
    u64* gk = cand + (size_t)b * cand_cap;
    int C = cand_count[b * CAND_CNT_STRIDE];
    if ((size_t)C > cand_cap) C = (int)cand_cap;
    int n = 1024;
    while (n < C) n <<= 1;
    const u64* sorted;
    if (n <= SEL_LDS_KEYS) {
        for (int e = t; e < n; e += 1024) skeys[e] = (e < C) ? gk[e] : ~0ull;
        __syncthreads();
        bitonic_sort<true>(skeys, n, t);
        sorted = skeys;
    } else {
        for (int e = C + t; e < n; e += 1024) gk[e] = ~0ull;
        __syncthreads();
        bitonic_sort<false>(gk, n, t);
        sorted = gk;
    }
    const int N = C < nfeatures ? C : nfeatures;
    for (int e = t; e < nfeatures; e += 1024) slot_src[(size_t)b * nfeatures + e] = -1;
    if (t == 0) { baseF = 0; baseB = 0; baseV = 0; }
    __syncthreads();
    for (int i0 = 0; i0 < N; i0 += 1024) {
        const int i = i0 + t;
        bool valid = false, back = false;
        u64 key = 0;
        if (i < N) {
            key = sorted[i];
            const float score = ord2f(~(unsigned)(key >> 32));
            const int x = (int)((unsigned)(key & 0xFFFFFFFFull) % (unsigned)W);
            valid = score > 0.f;                                          // XFextractor.cc:313
            { const float xf = (float)x * rw; back = valid && (xf >= (float)lap0 && xf <= (float)lap1); }   // :332 (rw = 1 unless XFH_FLAG_RESCALE_KEYPOINTS)
            sel_key[(size_t)b * nfeatures + i] = key;
        }
        const bool front = valid && !back;
        const u64 mF = __ballot(front), mB = __ballot(back);
        const u64 lt = (1ull << lane) - 1ull;
        if (lane == 0) { wsumF[wave] = __popcll(mF); wsumB[wave] = __popcll(mB); }
        __syncthreads();
        int oF = baseF, oB = baseB;
        for (int w = 0; w < wave; ++w) { oF += wsumF[w]; oB += wsumB[w]; }
        if (front) slot_src[(size_t)b * nfeatures + oF + __popcll(mF & lt)] = i;
        if (back) slot_src[(size_t)b * nfeatures + (nfeatures - 1 - (oB + __popcll(mB & lt)))] = i;
        __syncthreads();
        if (t == 0) {
            int sF = 0, sB = 0;
            for (int w = 0; w < 16; ++w) { sF += wsumF[w]; sB += wsumB[w]; }
            baseF += sF; baseB += sB;
        }
        __syncthreads();
    }
    if (t == 0) {
        sel_n[b] = N;
        RecordHeader* hdr = (RecordHeader*)(records + (size_t)b * rec_bytes);
        hdr->n_valid = baseF + baseB; hdr->mono_index = baseF; hdr->n_candidates = cand_count[b * CAND_CNT_STRIDE]; hdr->reserved = 0;
    }
    (void)wsumV; (void)baseV;
}

// ---- k_select: top-k for nfeatures <= 4096, one workgroup of 1024 threads per frame ---------------
//   * C > 4096 candidates: an 8-bit MSB radix select (LDS histogram) first finds the key prefix below
//     which exactly nfeatures keys lie (keys are unique: score bits + linear pixel index) and
//     compacts those keys;
//   * the <= 4096 keys are sorted with a bitonic network that keeps 4 keys per thread in registers:
//     strides 1-2 are register compare-exchanges, strides up to 32 threads are wave shuffles, and only
//     the 10 widest of the 78 passes go through LDS with a barrier.  (Sorting all candidates with 16
//     keys per thread was measured slower, 140 us vs 50 us: the ds_bpermute volume dominates.)
//   * validity (score > 0) and front/back slots come from one ballot scan (thread t owns ranks
//     KPT*t .. KPT*t+KPT-1).
#define SEL_FAST_MAX 4096
__device__ __forceinline__ void cmpx(u64& a, u64& b, bool asc) {          // a has the lower index
    const bool sw = (a > b) == asc;
    const u64 x = sw ? b : a, y = sw ? a : b;
    a = x; b = y;
}
template <int KPT, int J>
__device__ __forceinline__ void reg_stage(u64 (&key)[KPT], int k, int t) {
#pragma unroll
    for (int q = 0; q < KPT; ++q)
        if ((q & J) == 0) cmpx(key[q], key[q + J], ((KPT * t + q) & k) == 0);
}
// sorts KPT*NT keys ascending with NT threads (the first NT of the workgroup; the others must have left: a terminated wave does not
// take part in s_barrier); thread t ends with ranks KPT*t .. KPT*t+KPT-1
template <int KPT, int NT = 1024>
__device__ __forceinline__ void hybrid_bitonic(u64 (&key)[KPT], u64* sk, int t) {
    constexpr int NTOT = KPT * NT;
    for (int k = 2; k <= NTOT; k <<= 1) {
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j >= 64 * KPT) {
                const int m = j / KPT;                        // partner thread distance (>= 64: other wave)
#pragma unroll
                for (int q = 0; q < KPT; ++q) sk[q * NT + t] = key[q];
                __syncthreads();
                const bool lower = (t & m) == 0;
#pragma unroll
                for (int q = 0; q < KPT; ++q) {
                    const u64 o = sk[q * NT + (t ^ m)];
                    const bool keep_min = (lower == (((KPT * t + q) & k) == 0));
                    key[q] = keep_min ? (key[q] < o ? key[q] : o) : (key[q] > o ? key[q] : o);
                }
                __syncthreads();
            } else if (j >= KPT) {
                const int m = j / KPT;                        // partner lane distance (< 64: same wave)
                const bool lower = (t & m) == 0;
#pragma unroll
                for (int q = 0; q < KPT; ++q) {
                    const u64 o = __shfl_xor(key[q], m);
                    const bool keep_min = (lower == (((KPT * t + q) & k) == 0));
                    key[q] = keep_min ? (key[q] < o ? key[q] : o) : (key[q] > o ? key[q] : o);
                }
            } else {
                switch (j) {
                    case 1: reg_stage<KPT, 1>(key, k, t); break;
                    case 2: reg_stage<KPT, 2>(key, k, t); break;
                    case 4: if constexpr (KPT > 4) reg_stage<KPT, 4>(key, k, t); break;
                    case 8: if constexpr (KPT > 8) reg_stage<KPT, 8>(key, k, t); break;
                }
            }
        }
    }
}
// validity, slots and bookkeeping for the sorted keys held KPT per thread
template <int KPT, int NT = 1024>
__device__ __forceinline__ void place_sorted(const u64 (&key)[KPT], int t, int b, int N, int W, int nfeatures, int lap0, int lap1, float rw,
                                             int* __restrict__ slot_src, u64* __restrict__ sel_key, int* __restrict__ sel_n,
                                             uint8_t* __restrict__ records, size_t rec_bytes, int n_cand, int* wsumF, int* wsumB) {
    const int lane = t & 63, wave = t >> 6;
    for (int e = t; e < nfeatures; e += NT) slot_src[(size_t)b * nfeatures + e] = -1;
    __syncthreads();
    bool front[KPT], back[KPT];
    int cF = 0, cB = 0;
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
        const int i = KPT * t + q;
        front[q] = false; back[q] = false;
        if (i < N) {
            const float score = ord2f(~(unsigned)(key[q] >> 32));
            const int x = (int)((unsigned)(key[q] & 0xFFFFFFFFull) % (unsigned)W);
            const bool valid = score > 0.f;                               // XFextractor.cc:313
            { const float xf = (float)x * rw; back[q] = valid && (xf >= (float)lap0 && xf <= (float)lap1); }   // :332 (rw = 1 unless XFH_FLAG_RESCALE_KEYPOINTS)
            front[q] = valid && !back[q];
            sel_key[(size_t)b * nfeatures + i] = key[q];
        }
        cF += front[q] ? 1 : 0; cB += back[q] ? 1 : 0;
    }
    int iF = cF, iB = cB;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int oF = __shfl_up(iF, d), oB = __shfl_up(iB, d);
        if (lane >= d) { iF += oF; iB += oB; }
    }
    if (lane == 63) { wsumF[wave] = iF; wsumB[wave] = iB; }
    __syncthreads();
    int oF = iF - cF, oB = iB - cB, totF = 0, totB = 0;
    for (int w = 0; w < NT / 64; ++w) {
        if (w < wave) { oF += wsumF[w]; oB += wsumB[w]; }
        totF += wsumF[w]; totB += wsumB[w];
    }
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
        if (front[q]) slot_src[(size_t)b * nfeatures + oF++] = KPT * t + q;
        if (back[q]) slot_src[(size_t)b * nfeatures + (nfeatures - 1 - oB++)] = KPT * t + q;
    }
    if (t == 0) {
        sel_n[b] = N;
        RecordHeader* hdr = (RecordHeader*)(records + (size_t)b * rec_bytes);
        hdr->n_valid = totF + totB; hdr->mono_index = totF; hdr->n_candidates = n_cand; hdr->reserved = 0;
    }
}

// the radix-select + bitonic form: any candidate distribution (the fallback of k_select, and what it was before the bucket ranking)
__device__ __forceinline__ void select_radix_bitonic(const u64* __restrict__ gk, int C, int N, int n_cand, int W, int nfeatures,
              int lap0, int lap1, float rw, int* __restrict__ slot_src, u64* __restrict__ sel_key, int* __restrict__ sel_n,
              uint8_t* __restrict__ records, size_t rec_bytes, u64* sk /* 16384 keys */, int* hist /* 256 */, int* wsumF, int* wsumB, int* s_misc /* 4 */) {
    int& s_digit = s_misc[0]; int& s_need = s_misc[1]; int& s_done = s_misc[2]; int& s_cnt = s_misc[3];
    const int t = threadIdx.x, b = blockIdx.x, lane = t & 63, wave = t >> 6;

    // Sort size: 4096 keys on 1024 threads, or -- when at most 1024 keypoints are wanted (TUM1.yaml asks for 1000) or the frame has at
    // most 1024 candidates -- 1024 keys on the first 256 threads: three LDS exchange stages between four waves instead of ten between
    // sixteen (the sort and the placement, not the radix passes, are most of this kernel: 35 of 49 us at 4096 keys).
    const bool small = N <= 1024;
    const int cap = small ? 1024 : SEL_FAST_MAX;
    auto sort_and_place = [&](bool from_lds) {
        if (small) {
            if (t >= 256) return;                            // (every barrier below is among the first four waves only)
            u64 key[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) key[q] = from_lds ? sk[4 * t + q] : ((4 * t + q < C) ? gk[4 * t + q] : ~0ull);
            if (from_lds) __syncthreads();
            hybrid_bitonic<4, 256>(key, sk, t);
            place_sorted<4, 256>(key, t, b, N, W, nfeatures, lap0, lap1, rw, slot_src, sel_key, sel_n, records, rec_bytes, n_cand, wsumF, wsumB);
        } else {
            u64 key[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) key[q] = from_lds ? sk[4 * t + q] : ((4 * t + q < C) ? gk[4 * t + q] : ~0ull);
            if (from_lds) __syncthreads();
            hybrid_bitonic<4>(key, sk, t);
            place_sorted<4>(key, t, b, N, W, nfeatures, lap0, lap1, rw, slot_src, sel_key, sel_n, records, rec_bytes, n_cand, wsumF, wsumB);
        }
    };
    if (C <= cap) { sort_and_place(false); return; }
    // ---- more candidates than the sort takes: radix select (8-bit digits from the top) until the keys at or below the current prefix
    // fit the sort (<= cap), which then orders them exactly and the first N are taken.  Typically two
    // passes: the keys that share the top 16 bits of the N-th score are few.
    int shift = 64;
    u64 prefix = 0;
    {
        int need = N;                                    // keys still to take among those that match the prefix
        for (int pass = 0; pass < 8; ++pass) {
            const int sh = 56 - 8 * pass;
            for (int e = t; e < 256; e += 1024) hist[e] = 0;
            __syncthreads();
            for (int e = t; e < C; e += 1024) {
                const u64 k = gk[e];
                if (pass == 0 || (k >> (sh + 8)) == prefix) atomicAdd(&hist[(int)((k >> sh) & 255)], 1);
            }
            __syncthreads();
            if (wave == 0) {
                const int h0 = hist[lane * 4], h1 = hist[lane * 4 + 1], h2 = hist[lane * 4 + 2], h3 = hist[lane * 4 + 3];
                const int s4 = h0 + h1 + h2 + h3;
                int incl = s4;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
                const int excl = incl - s4;
                if (excl < need && need <= incl) {
                    int cum = excl, dg = lane * 4, hv = h0;
                    if (cum + h0 < need) { cum += h0; dg = lane * 4 + 1; hv = h1;
                        if (cum + h1 < need) { cum += h1; dg = lane * 4 + 2; hv = h2;
                            if (cum + h2 < need) { cum += h2; dg = lane * 4 + 3; hv = h3; } } }
                    // keys strictly better than the new prefix: (N - need) + cum; with the hv keys that share it they must fit the sort
                    s_digit = dg; s_need = need - cum; s_done = ((N - need) + cum + hv <= cap) ? 1 : 0;
                }
            }
            __syncthreads();
            prefix = (prefix << 8) | (u64)s_digit;
            need = s_need;
            shift = sh;
            const int done = s_done;
            __syncthreads();
            if (done) break;
        }
    }
    if (t == 0) s_cnt = 0;
    for (int e = t; e < cap; e += 1024) sk[e] = ~0ull;
    __syncthreads();
    for (int e = t; e < C; e += 1024) {
        const u64 k = gk[e];
        if ((k >> shift) <= prefix) sk[atomicAdd(&s_cnt, 1)] = k;
    }
    __syncthreads();
    sort_and_place(true);
}


// ---- k_select: top-k for nfeatures <= 4096, one workgroup of 1024 threads per frame ---------------
// Bucket ranking.  The keys are unique, so a key's output slot is its RANK, and the rank is
//     (number of keys in lower buckets) + (number of smaller keys in its own bucket)
// for any order-preserving bucket function.  Here: 4096 buckets over the 12 bits below the highest bit in which the candidates differ
// (for scores spread over ten binades that is 256 buckets per binade, a few keys per bucket).
//   A  every thread takes up to 16 candidates into registers (more: re-read from memory); OR / AND of the keys give the window;
//   B  LDS histogram;  C  scan -> first key of every bucket, and the bucket T in which rank N falls;
//   D  the keys of buckets <= T go to their bucket's segment of an LDS array (order inside a segment: arbitrary);
//   E  every such key counts the smaller keys of its segment -> its final slot (slots >= N are dropped);  F  placement as before.
// One pass over the candidates and no sorting network: 47 -> 17 us for 9 000 candidates / 4096 keypoints on one CU (the bitonic network
// alone was 35 us).  The result is the same array whatever the method (a sort of unique keys); distributions the buckets cannot split
// (a bucket > 256 keys: thousands of equal scores) or that overflow the segment array take select_radix_bitonic instead
// (XFH_SELECT_LEGACY=1 forces it, tests/test_gpu_extract.py runs both).  Negative scores (the (0,0) candidate's -1, XFextractor.cc:281)
// would stretch the window by 30 bits: they share the last bucket and are left out of the window.
#define SEL_NB 4096
#define SEL_SEGCAP 8192
#define SEL_BUCKET_MAX 256
#define SEL_BUCKET_FINE 8           // no selected bucket above this: the second level would cost more than it saves
#define SEL_KREG 16
#define SEL_LDS_BYTES (SEL_LDS_KEYS * 8 + 64)
__device__ __forceinline__ u64 wave_or64(u64 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v |= __shfl_xor(v, d);
    return v;
}
__global__ __launch_bounds__(1024)
void k_select(const u64* __restrict__ cand, size_t cand_cap, const int* __restrict__ cand_count, int W, int nfeatures,
              int lap0, int lap1, float rw, int* __restrict__ slot_src, u64* __restrict__ sel_key, int* __restrict__ sel_n,
              uint8_t* __restrict__ records, size_t rec_bytes, int legacy) {
    extern __shared__ __attribute__((aligned(16))) u64 sk[];        // legacy: 16384 keys; bucket path: bykey[8192] | sorted[4096] | hist[4096] | cum[4097]
    __shared__ int hist256[256];
    __shared__ int wsumF[16], wsumB[16];
    __shared__ int s_misc[4];
    __shared__ unsigned s_or[2], s_nand[2];
    __shared__ int s_T, s_segtot, s_maxb, s_bmin, wtot[16];
    const int t = threadIdx.x, b = blockIdx.x, lane = t & 63, wave = t >> 6;
    const u64* gk = cand + (size_t)b * cand_cap;
    const int n_cand = cand_count[b * CAND_CNT_STRIDE];
    int C = n_cand;
    if ((size_t)C > cand_cap) C = (int)cand_cap;
    const int N = C < nfeatures ? C : nfeatures;
    if (legacy) {
        select_radix_bitonic(gk, C, N, n_cand, W, nfeatures, lap0, lap1, rw, slot_src, sel_key, sel_n, records, rec_bytes, sk, hist256, wsumF, wsumB, s_misc);
        return;
    }
    u64* bykey = sk;
    u64* sorted = sk + SEL_SEGCAP;
    int* hist = (int*)(sk + SEL_SEGCAP + SEL_FAST_MAX);
    int* cum = hist + SEL_NB;                                       // SEL_NB + 1 entries
    // ---- A: candidates -> registers, window
    u64 key[SEL_KREG];
    u64 acc_or = 0ull, acc_and = ~0ull;
#pragma unroll
    for (int q = 0; q < SEL_KREG; ++q) {
        const int e = q * 1024 + t;
        key[q] = e < C ? gk[e] : ~0ull;
    }
    for (int e = t; e < SEL_NB; e += 1024) hist[e] = 0;
    if (t < 2) { s_or[t] = 0u; s_nand[t] = 0u; }
    if (t == 0) { s_T = 0; s_segtot = 0; s_maxb = 0; s_bmin = 0; }
#pragma unroll
    for (int q = 0; q < SEL_KREG; ++q)
        if (q * 1024 + t < C && !(key[q] >> 63)) { acc_or |= key[q]; acc_and &= key[q]; }
    for (int e = SEL_KREG * 1024 + t; e < C; e += 1024) { const u64 k = gk[e]; if (!(k >> 63)) { acc_or |= k; acc_and &= k; } }
    acc_or = wave_or64(acc_or); acc_and = ~wave_or64(~acc_and);
    __syncthreads();
    if (lane == 0) {
        atomicOr(&s_or[0], (unsigned)acc_or); atomicOr(&s_or[1], (unsigned)(acc_or >> 32));
        atomicOr(&s_nand[0], (unsigned)~acc_and); atomicOr(&s_nand[1], (unsigned)(~acc_and >> 32));
    }
    __syncthreads();
    int shift = 0;
    {
        const u64 o = ((u64)s_or[1] << 32) | s_or[0], na = ((u64)s_nand[1] << 32) | s_nand[0];
        const u64 d = o & na;                                       // bits in which two non-negative keys differ
        if (d) { const int hb = 63 - __builtin_clzll(d); shift = hb > 11 ? hb - 11 : 0; }
    }
    // bucket of a key at the current level: level 0 = the 12 bits from the highest differing bit of ALL candidates; level 1 = the
    // selected buckets [bmin, T0] of level 0 split 2^r ways (the top nfeatures of 10 000 candidates usually span a fraction of the
    // window: one more histogram over them buys 2^r times finer buckets, and step E is quadratic in the bucket size)
    int sh = shift, base = 0, sh0 = shift, T0 = SEL_NB - 1;
    u64 msk = (u64)(SEL_NB - 1);
    auto bucket = [&](u64 k) -> int { return (k >> 63) ? SEL_NB - 1 : (int)((k >> sh) & msk) - base; };
    auto chosen = [&](u64 k) -> bool { return ((k >> 63) ? SEL_NB - 1 : (int)((k >> sh0) & (u64)(SEL_NB - 1))) <= T0; };
    int T = 0, segtot = 0;
    for (int level = 0; level < 2; ++level) {
        // ---- B: histogram
#pragma unroll
        for (int q = 0; q < SEL_KREG; ++q)
            if (q * 1024 + t < C && chosen(key[q])) atomicAdd(&hist[bucket(key[q])], 1);
        for (int e = SEL_KREG * 1024 + t; e < C; e += 1024) { const u64 k = gk[e]; if (chosen(k)) atomicAdd(&hist[bucket(k)], 1); }
        __syncthreads();
        // ---- C: exclusive scan (thread t owns buckets 4t .. 4t+3), the bucket of rank N, the first occupied bucket
        int h[4];
        int s4 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { h[i] = hist[4 * t + i]; s4 += h[i]; }
        int incl = s4;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int run = incl - s4;
        for (int w = 0; w < wave; ++w) run += wtot[w];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cum[4 * t + i] = run;
            if (run < N && N <= run + h[i]) { s_T = 4 * t + i; s_segtot = run + h[i]; }
            if (run == 0 && h[i] > 0) s_bmin = 4 * t + i;
            run += h[i];
        }
        if (t == 1023) cum[SEL_NB] = run;
        __syncthreads();
        T = s_T; segtot = s_segtot;
        const int bmin = s_bmin;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * t + i <= T && h[i] > SEL_BUCKET_FINE) atomicMax(&s_maxb, h[i]);
        int r = 0;
        if (level == 0 && N > 0) {
            const int span = T - bmin + 1;
            while ((span << (r + 1)) <= SEL_NB && r < sh) ++r;
        }
        __syncthreads();                                            // s_maxb complete; every thread has read s_T / s_segtot / s_bmin
        const int maxb0 = s_maxb;
        if (r == 0 || maxb0 <= SEL_BUCKET_FINE) break;              // (s_maxb is 0 unless a selected bucket holds more than SEL_BUCKET_FINE keys)
        __syncthreads();                                            // ... and s_maxb, BEFORE thread 0 resets it below: without this barrier a wave that is a few
                                                                    // instructions behind read the reset value, left the loop while the others went on to level 1,
                                                                    // and the workgroup's barriers no longer matched (a rare, box-dependent corruption of one frame's selection)
        // refine: level 1 over the keys of buckets <= T
        T0 = T; sh = sh0 - r; msk = ((u64)SEL_NB << r) - 1ull; base = bmin << r;
        for (int e = t; e < SEL_NB; e += 1024) hist[e] = 0;
        if (t == 0) { s_T = 0; s_segtot = 0; s_maxb = 0; s_bmin = 0; }
        __syncthreads();
    }
    if (segtot > SEL_SEGCAP || s_maxb > SEL_BUCKET_MAX) {            // (uniform) not a distribution for buckets
        __syncthreads();
        select_radix_bitonic(gk, C, N, n_cand, W, nfeatures, lap0, lap1, rw, slot_src, sel_key, sel_n, records, rec_bytes, sk, hist256, wsumF, wsumB, s_misc);
        return;
    }
    // ---- D: keys of the buckets <= T into their segments
    if (N > 0) {
#pragma unroll
        for (int q = 0; q < SEL_KREG; ++q)
            if (q * 1024 + t < C && chosen(key[q])) {
                const int bk = bucket(key[q]);
                if (bk <= T) bykey[cum[bk] + atomicSub(&hist[bk], 1) - 1] = key[q];
            }
        for (int e = SEL_KREG * 1024 + t; e < C; e += 1024) {
            const u64 k = gk[e];
            if (!chosen(k)) continue;
            const int bk = bucket(k);
            if (bk <= T) bykey[cum[bk] + atomicSub(&hist[bk], 1) - 1] = k;
        }
    }
    __syncthreads();
    // ---- E: rank inside the segment -> final slot
    for (int i = t; i < (N > 0 ? segtot : 0); i += 1024) {
        const u64 k = bykey[i];
        const int bk = bucket(k), s0 = cum[bk], s1 = cum[bk + 1];
        int r = 0;
        for (int j = s0; j < s1; ++j) r += bykey[j] < k ? 1 : 0;
        if (s0 + r < N) sorted[s0 + r] = k;
    }
    __syncthreads();
    // ---- F: validity, lapping placement, header
    u64 k4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) k4[q] = (4 * t + q < N) ? sorted[4 * t + q] : ~0ull;
    place_sorted<4>(k4, t, b, N, W, nfeatures, lap0, lap1, rw, slot_src, sel_key, sel_n, records, rec_bytes, n_cand, wsumF, wsumB);
}

// ---- k_desc: 16 lanes per output slot (4 slots per wave), a lane owns 4 descriptor channels -------
__global__ __launch_bounds__(256)
void k_desc(const float* __restrict__ feats, size_t m_stride, const float* __restrict__ fnorm, size_t n_stride,
            const int* __restrict__ slot_src, const u64* __restrict__ sel_key,
            int H, int W, int nfeatures, float rw, float rh, uint8_t* __restrict__ records, size_t rec_bytes, size_t kps_off, size_t desc_off, int write_padding,
            float* __restrict__ images /* or null: one prepared match image per frame (mnn_layout.h) */, size_t image_floats) {
    const int b = blockIdx.z;
    const int slot = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
    const bool in_range = slot < nfeatures;
    uint8_t* rec = records + (size_t)b * rec_bytes;
    float* kp = (float*)(rec + kps_off + (size_t)slot * 28);
    float* dd = (float*)(rec + desc_off + (size_t)slot * 256) + l * 4;
    const int src = in_range ? slot_src[(size_t)b * nfeatures + slot] : -1;
    const bool live = src >= 0;
    // the reduction below runs in every lane of the wave; slots without a keypoint carry zeros through it
    u64 key = 0;
    if (live) key = sel_key[(size_t)b * nfeatures + src];
    const float score = ord2f(~(unsigned)(key >> 32));
    const unsigned idx = (unsigned)(key & 0xFFFFFFFFull);
    const int x = (int)(idx % (unsigned)W), y = (int)(idx / (unsigned)W);
    const int Wh = W >> 3, Hh = H >> 3;
    const float ix = grid_coord(x, W, Wh), iy = grid_coord(y, H, Hh);
    const float xw = floorf(ix), yn = floorf(iy);
    const float w = ix - xw, e = 1.0f - w, n = iy - yn, so = 1.0f - n;
    const float nw = e * so, ne = w * so, sw = e * n, se = w * n;
    const int x0 = (int)xw, y0 = (int)yn, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < Wh, vx1 = x1 >= 0 && x1 < Wh, vy0 = y0 >= 0 && y0 < Hh, vy1 = y1 >= 0 && y1 < Hh;
    // the four feature pixels the sample touches (zero outside the map: grid_sample padding_mode zeros), each divided by its
    // norm (k_feat_norm): the values a normalised copy of the map would hold
    const float* m = feats + (size_t)b * m_stride + l * 4;
    const float* fn = fnorm + (size_t)b * n_stride;
    const int cx0 = min(max(x0, 0), Wh - 1), cx1 = min(max(x1, 0), Wh - 1), cy0 = min(max(y0, 0), Hh - 1), cy1 = min(max(y1, 0), Hh - 1);
    const size_t p00 = (size_t)cy0 * Wh + cx0, p01 = (size_t)cy0 * Wh + cx1, p10 = (size_t)cy1 * Wh + cx0, p11 = (size_t)cy1 * Wh + cx1;
    f32x4 r0 = *(const f32x4*)(m + p00 * 64), r1 = *(const f32x4*)(m + p01 * 64), r2 = *(const f32x4*)(m + p10 * 64), r3 = *(const f32x4*)(m + p11 * 64);
    const Recip k0 = recip_of(fn[p00]), k1 = recip_of(fn[p01]), k2 = recip_of(fn[p10]), k3 = recip_of(fn[p11]);
    const bool t0 = live && vx0 && vy0, t1 = live && vx1 && vy0, t2 = live && vx0 && vy1, t3 = live && vx1 && vy1;
    f32x4 v;
    double ss = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = t0 ? div_by(r0[q], k0) : 0.f, bb = t1 ? div_by(r1[q], k1) : 0.f, d = t2 ? div_by(r2[q], k2) : 0.f, g = t3 ? div_by(r3[q], k3) : 0.f;
        v[q] = ((a * nw + bb * ne) + d * sw) + g * se;
        ss = fma((double)v[q], (double)v[q], ss);
    }
    ss += __shfl_xor(ss, 8); ss += __shfl_xor(ss, 4); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 1);
    const Recip kn = recip_of(fmaxf((float)sqrt(ss), 1e-12f));
    const f32x4 dsc = live ? f32x4{div_by(v[0], kn), div_by(v[1], kn), div_by(v[2], kn), div_by(v[3], kn)} : f32x4{0.f, 0.f, 0.f, 0.f};
    // the matcher's prepared image of this frame: every slot up to the panel boundary, padding slots as rows of zeros
    if (images) mnn_emit_row(dsc, slot, l, images + (size_t)b * image_floats);
    if (!in_range) return;
    if (!live) {
        if (!write_padding) return;        // host-visible record (xfh_extract_submit): the host pads, nothing crosses PCIe
        // default cv::KeyPoint(): pt (0,0), size 0, angle -1, response 0, octave 0, class_id -1
        if (l < 5) kp[l] = (l == 3) ? -1.f : 0.f;
        else if (l < 7) ((int*)kp)[l] = (l == 5) ? 0 : -1;
        *(f32x4*)dd = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    *(f32x4*)dd = dsc;
    if (l < 7) {
        // KeyPoint(x, y, 1, -1, score): octave 0, class_id -1 (XFextractor.cc:329); the Long rescale at :304-305
        // multiplies by 1: rw = rh = 1 unless XFH_FLAG_RESCALE_KEYPOINTS asks for input coordinates
        const float val = l == 0 ? (float)x * rw : l == 1 ? (float)y * rh : l == 2 ? 1.f : l == 3 ? -1.f : score;
        if (l < 5) kp[l] = val;
        else ((int*)kp)[l] = (l == 5) ? 0 : -1;
    }
}

// ---------------------------------------------------------------------------------------------
// pipeline
hipError_t launch_basic_layer(xfh_ctx* c, int li, const float* in, size_t in_stride, int src, int pro, int Hin, int Win, int B);
hipError_t launch_block1_stats(xfh_ctx* c, const StatSrc& xs, int H, int W, int B);
hipError_t launch_fusion_chain(xfh_ctx* c, int Hh, int Wh, int B, int* done);
hipError_t launch_finalize_image(xfh_ctx* c, int B, int npart, double count);
StatSrc stat_src(xfh_ctx* c, int j, int B);
bool consumer_fold(int B);

#define CK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return _e; } while (0)
enum { PRO_PLAIN = 0, PRO_BN = 1, PRO_IN = 2, PRO_B2IN = 3, PRO_FUSE = 4, PRO_L0 = 5, PRO_UNFOLD = 6 };

hipError_t run_extract(xfh_ctx* c, const uint8_t* d_gray, int B, int H0, int W0, int lap0, int lap1, uint8_t* d_records, bool write_padding, float* d_images) {
    const int H = (H0 / 32) * 32, W = (W0 / 32) * 32;
    c->B = B; c->H0 = H0; c->W0 = W0; c->H = H; c->W = W;
    const int h4 = H / 4, w4 = W / 4, h8 = H / 8, w8 = W / 8;
    hipStream_t s = c->stream;
    const size_t xs = (size_t)c->Hmax * c->Wmax;
    const int nf = c->cfg.nfeatures;
    const size_t rec = xfh_record_bytes(nf);
    const bool resc = (c->cfg.flags & XFH_FLAG_RESCALE_KEYPOINTS) != 0;
    const float rw = resc ? (float)((double)W0 / (double)W) : 1.0f, rh = resc ? (float)((double)H0 / (double)H) : 1.0f;

    // image -> float, resize, InstanceNorm statistics
    const int npre = (H * W + 1023) / 1024;
    launch_k(c, XFH_K_PREPROC, -1, k_preproc, dim3(npre, 1, B), dim3(256), 0, d_gray, (size_t)H0 * W0, H0, W0, H, W, c->X, xs, c->pre_part, c->pre_npart, c->cand_count);
    CK(hipGetLastError());
    StatSrc xsrc{};
    xsrc.stat = c->xstat;
    if (consumer_fold(B)) { xsrc.part = c->pre_part; xsrc.part_stride = (size_t)c->pre_npart * 2; xsrc.npart = npre; xsrc.count = (double)H * (double)W; xsrc.stat_out = c->xstat; }
    else CK(launch_finalize_image(c, B, npre, (double)H * (double)W));
    // skip1's AvgPool4 of the normalised image (and, for B <= 8, the published image statistics): in the reference's batch-statistics
    // mode k_block1_stats produces both on its one pass over the image; with eval() statistics that kernel is not needed and
    // k_norm_aux does it
    if (c->cfg.bn_mode == XFH_BN_BATCH_STATS) CK(launch_block1_stats(c, xsrc, H, W, B));
    else {
        hipLaunchKernelGGL(k_norm_aux, dim3((h4 * w4 + 255) / 256, 1, B), dim3(256), 0, s, c->X, xs, xsrc, H, W, c->skip_pool, xs / 16);
        CK(hipGetLastError());
    }
    // keypoint branch (keypoint_head.0-3 on unfold2d(x), softmax, depth-to-space) on the second stream: it only needs
    // the normalised image, so it runs beside the backbone (memory-bound 1x1 layers next to MFMA-bound 3x3 layers)
    const bool ride = ride_mode(c, B);        // the branch rides on block1.3 .. block3.0 instead (k_conv_mfma_ride): no second stream, no fork, no join
    if (!ride) {
        const bool two = !(c->cfg.flags & XFH_FLAG_SERIAL_BRANCH);
        hipStream_t branch = two ? c->aux_stream : s;
        hipError_t e = hipEventRecord(c->ev_fork, s);
        if (e == hipSuccess) e = hipStreamWaitEvent(branch, c->ev_fork, 0);
        CK(e);
        c->stream = branch;
        e = launch_basic_layer(c, 20, c->X, xs, -1, PRO_UNFOLD, h8, w8, B);          // unfold2d(x-hat) is never materialised
        if (e == hipSuccess) e = launch_basic_layer(c, 21, c->raw[20], c->raw_stride[20], 20, PRO_BN, h8, w8, B);
        if (e == hipSuccess) e = launch_basic_layer(c, 22, c->raw[21], c->raw_stride[21], 21, PRO_BN, h8, w8, B);
        if (e == hipSuccess) {
            if (consumer_fold(B))        // small batches (at 256 frames it measured 287 vs 263 us): four lanes per pixel (k_heads_kp4), the same bits
                launch_k(c, XFH_K_HEADS, -1, k_heads_kp4, dim3((h8 * w8 + HK4_PX - 1) / HK4_PX, 1, B), dim3(4 * HK4_PX), 0,
                         Kp4Args{(const float*)c->raw[22], stat_src(c, 22, B), c->raw_stride[22], (const float*)c->w.kp3_w, (const float*)c->w.kp3_b, h8, w8, c->K1h, xs});
            else
            launch_k(c, XFH_K_HEADS, -1, k_heads_kp, dim3((h8 * w8 + HF_PX - 1) / HF_PX, 1, B), dim3(HF_PX), 0,
                     (const float*)c->raw[22], stat_src(c, 22, B), c->raw_stride[22], (const float*)c->w.kp3_w, (const float*)c->w.kp3_b, h8, w8, c->K1h, xs);
            e = hipGetLastError();
        }
        c->stream = s;
        // the fork is always joined, also on an error path: the main stream must not run ahead of (or be destroyed before) the branch
        const hipError_t e2 = hipEventRecord(c->ev_join, branch);
        if (e != hipSuccess || e2 != hipSuccess) { hipStreamSynchronize(branch); return e != hipSuccess ? e : e2; }
    }
    // backbone + heatmap head on the main stream; whatever happens, the branch is joined afterwards (K1h is needed by the NMS,
    // and the main stream must never run ahead of, or be destroyed before, the second stream)
    auto backbone = [&]() -> hipError_t {
    // block1
    CK(launch_basic_layer(c, 1, c->X, xs, 0, PRO_L0, H, W, B));               // block1.0 is recomputed from the image while staging
    CK(launch_basic_layer(c, 2, c->raw[1], c->raw_stride[1], 1, PRO_BN, c->lh[1], c->lw[1], B));
    if (ride) {                                 // + keypoint_head.0, .1, .2 and .3 / softmax as riders
        CK(launch_layer_with_rider(c, 3, c->raw[2], c->raw_stride[2], 2, PRO_BN, c->lh[2], c->lw[2], B, c->K1h, xs));
        CK(launch_layer_with_rider(c, 4, c->raw[3], c->raw_stride[3], 3, PRO_B2IN, h4, w4, B, c->K1h, xs));
        CK(launch_layer_with_rider(c, 5, c->raw[4], c->raw_stride[4], 4, PRO_BN, h4, w4, B, c->K1h, xs));
        CK(launch_layer_with_rider(c, 6, c->raw[5], c->raw_stride[5], 5, PRO_BN, h4, w4, B, c->K1h, xs));
    } else {
    CK(launch_basic_layer(c, 3, c->raw[2], c->raw_stride[2], 2, PRO_BN, c->lh[2], c->lw[2], B));
    // block2 (block2.0 adds skip1(x) to x1 while staging), block3
    CK(launch_basic_layer(c, 4, c->raw[3], c->raw_stride[3], 3, PRO_B2IN, h4, w4, B));
    CK(launch_basic_layer(c, 5, c->raw[4], c->raw_stride[4], 4, PRO_BN, h4, w4, B));
    CK(launch_basic_layer(c, 6, c->raw[5], c->raw_stride[5], 5, PRO_BN, h4, w4, B));
    }
    CK(launch_basic_layer(c, 7, c->raw[6], c->raw_stride[6], 6, PRO_BN, h8, w8, B));
    CK(launch_basic_layer(c, 8, c->raw[7], c->raw_stride[7], 7, PRO_BN, h8, w8, B));
    // block4, block5
    CK(launch_basic_layer(c, 9, c->raw[8], c->raw_stride[8], 8, PRO_BN, h8, w8, B));
    CK(launch_basic_layer(c, 10, c->raw[9], c->raw_stride[9], 9, PRO_BN, c->lh[9], c->lw[9], B));
    CK(launch_basic_layer(c, 11, c->raw[10], c->raw_stride[10], 10, PRO_BN, c->lh[10], c->lw[10], B));
    CK(launch_basic_layer(c, 12, c->raw[11], c->raw_stride[11], 11, PRO_BN, c->lh[11], c->lw[11], B));
    CK(launch_basic_layer(c, 13, c->raw[12], c->raw_stride[12], 12, PRO_BN, c->lh[12], c->lw[12], B));
    CK(launch_basic_layer(c, 14, c->raw[13], c->raw_stride[13], 13, PRO_BN, c->lh[13], c->lw[13], B));
    CK(launch_basic_layer(c, 15, c->raw[14], c->raw_stride[14], 14, PRO_BN, c->lh[14], c->lw[14], B));
    // pyramid fusion: block_fusion.0 builds x3 + up2(x4) + up4(x5) while staging
    CK(launch_basic_layer(c, 16, c->raw[8], c->raw_stride[8], 8, PRO_FUSE, h8, w8, B));
    CK(launch_basic_layer(c, 17, c->raw[16], c->raw_stride[16], 16, PRO_BN, h8, w8, B));
    // block_fusion.2 and the heatmap head: fusion.2 and heatmap_head.0 (no BatchNorm between them) are one kernel, with folded BatchNorms and B <= 8 heatmap_head.1 too
    int head_done = 0;
    CK(launch_fusion_chain(c, h8, w8, B, &head_done));
    if (head_done < 1) CK(launch_basic_layer(c, 18, c->feats, c->raw_stride[17], -1, PRO_PLAIN, h8, w8, B));
    if (head_done < 2) CK(launch_basic_layer(c, 19, c->raw[18], c->raw_stride[18], 18, PRO_BN, h8, w8, B));
    if (consumer_fold(B) && !c->no_nms_heat) return hipGetLastError();           // small batches: heatmap_head.2 + sigmoid are computed inside k_nms_score
    hipLaunchKernelGGL(k_heads_heat, dim3((h8 * w8 + HF_PX - 1) / HF_PX, 1, B), dim3(HF_PX), 0, s,
                       (const float*)c->raw[19], stat_src(c, 19, B), c->raw_stride[19], (const float*)c->w.heat2_w, (const float*)c->w.heat2_b, h8 * w8, c->H1, xs / 64);
    return hipGetLastError();
    };
    {
        const hipError_t eb = backbone();
        const hipError_t ej = ride ? hipSuccess : hipStreamWaitEvent(s, c->ev_join, 0);          // K1h of the keypoint branch
        CK(eb); CK(ej);
    }
    // NMS + score, top-k + placement, descriptors
    const int nms_blocks = ((W + NMS_TW - 1) / NMS_TW) * ((H + NMS_TH - 1) / NMS_TH);
    const int fn_blocks = consumer_fold(B) ? (h8 * w8 + 255) / 256 : 0;          // small batches: the feature norms ride on this launch
    if (fn_blocks && !c->no_nms_heat)
        launch_k(c, XFH_K_NMS, -1, k_nms_score<true>, dim3(nms_blocks + fn_blocks, 1, B), dim3(256), 0, c->K1h, xs, c->H1, xs / 64,
                       H, W, c->cfg.nms_threshold, c->cand, c->cand_cap, c->cand_count,
                       nms_blocks, (const float*)c->feats, c->raw_stride[17], h8 * w8, c->feat_nrm, xs / 64,
                       fn_blocks ? (const float*)c->raw[19] : (const float*)nullptr, stat_src(c, 19, B), c->raw_stride[19], (const float*)c->w.heat2_w, (const float*)c->w.heat2_b, c->H1);
    else
        launch_k(c, XFH_K_NMS, -1, k_nms_score<false>, dim3(nms_blocks + fn_blocks, 1, B), dim3(256), 0, c->K1h, xs, c->H1, xs / 64,
                       H, W, c->cfg.nms_threshold, c->cand, c->cand_cap, c->cand_count,
                       nms_blocks, (const float*)c->feats, c->raw_stride[17], h8 * w8, c->feat_nrm, xs / 64,
                       fn_blocks ? (const float*)c->raw[19] : (const float*)nullptr, stat_src(c, 19, B), c->raw_stride[19], (const float*)c->w.heat2_w, (const float*)c->w.heat2_b, c->H1);
    CK(hipGetLastError());
    if (nf <= SEL_FAST_MAX) {
        XFH_SET_LDS_ATTR_ONCE(c, k_select, SEL_LDS_BYTES);
        launch_k(c, XFH_K_SELECT, -1, k_select, dim3(B), dim3(1024), SEL_LDS_BYTES, (const u64*)c->cand, c->cand_cap, (const int*)c->cand_count, W, nf,
                 lap0, lap1, rw, c->slot_src, c->sel_key, c->sel_n, d_records, rec, c->select_legacy ? 1 : 0);
    } else {
        XFH_SET_LDS_ATTR_ONCE(c, k_select_generic, SEL_LDS_KEYS * 8);
        launch_k(c, XFH_K_SELECT, -1, k_select_generic, dim3(B), dim3(1024), SEL_LDS_KEYS * 8, c->cand, c->cand_cap, (const int*)c->cand_count, W, nf,
                 lap0, lap1, rw, c->slot_src, c->sel_key, c->sel_n, d_records, rec);
    }
    CK(hipGetLastError());
    if (!fn_blocks) {
        hipLaunchKernelGGL(k_feat_norm, dim3((h8 * w8 + 255) / 256, 1, B), dim3(256), 0, s, (const float*)c->feats, c->raw_stride[17], h8 * w8, c->feat_nrm, xs / 64);
        CK(hipGetLastError());
    }
    const int dslots = d_images ? (nf + MNN_PANEL - 1) / MNN_PANEL * MNN_PANEL : nf;      // image rows run to the panel boundary
    launch_k(c, XFH_K_DESC, -1, k_desc, dim3((dslots + 15) / 16, 1, B), dim3(256), 0, c->feats, c->raw_stride[17], (const float*)c->feat_nrm, xs / 64, c->slot_src, c->sel_key, H, W, nf, rw, rh,
                       d_records, rec, xfh_record_kps_offset(), xfh_record_desc_offset(nf), write_padding ? 1 : 0,
             d_images, (size_t)(dslots / MNN_PANEL) * MNN_PANEL_FLOATS);
    return hipGetLastError();
}

// ---- development / tests: k_select on a caller-made candidate set (frame 0 of the ctx) ---------------
#define HIPCK(c, x) do { hipError_t _e = (x); if (_e != hipSuccess) { (c)->hip_err = std::string(#x) + ": " + hipGetErrorString(_e); return XFH_ERR_HIP; } } while (0)
// keys: n unique (score, pixel) keys as k_nms_score writes them; sel_out: the selected keys in output order (capacity nfeatures);
// hdr_out: n_valid, mono_index, n_candidates, reserved.  form: 0 = as configured, 1 = bucket ranking (with its own fallback), 2 = radix + bitonic
extern "C" int xfh_debug_select(xfh_ctx* c, const unsigned long long* keys, int n, int width, int lap0, int lap1, int form, unsigned long long* sel_out, int* n_out, int* hdr_out) {
    if (!c || !keys || n < 0 || (size_t)n > c->cand_cap || width <= 0 || !sel_out || !n_out || !hdr_out || c->cfg.nfeatures > SEL_FAST_MAX) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    const int nf = c->cfg.nfeatures;
    const size_t rec = xfh_record_bytes(nf);
    HIPCK(c, hipMemcpyAsync(c->cand, keys, sizeof(u64) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(c->cand_count, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
    XFH_SET_LDS_ATTR_ONCE(c, k_select, SEL_LDS_BYTES);
    const int legacy = form == 0 ? (c->select_legacy ? 1 : 0) : (form == 2 ? 1 : 0);
    hipLaunchKernelGGL(k_select, dim3(1), dim3(1024), SEL_LDS_BYTES, c->stream, (const u64*)c->cand, c->cand_cap, (const int*)c->cand_count, width, nf,
                       lap0, lap1, 1.0f, c->slot_src, c->sel_key, c->sel_n, c->d_records, rec, legacy);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(n_out, c->sel_n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(hdr_out, c->d_records, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    const int N = *n_out;
    if (N < 0 || N > nf) return XFH_ERR_HIP;
    if (N) HIPCK(c, hipMemcpy(sel_out, c->sel_key, sizeof(u64) * (size_t)N, hipMemcpyDeviceToHost));
    return XFH_OK;
}
