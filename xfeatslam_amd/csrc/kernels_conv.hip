// kernels_conv.hip -- the XFeat backbone on gfx950 (reference XFeatModel::forward,
// src/XFeat.cc:135-173; BasicLayer = Conv2d(bias=false) -> BatchNorm2d(batch statistics,
// SURVEY.md Q1) -> ReLU, src/XFeat.cc:7-28).
//
// Data layout: activations are NHWC fp32, one slab per frame, and are stored RAW (conv
// output before BatchNorm).  Every conv kernel also emits per-workgroup fp64 (sum, sum^2)
// partials per output channel; k_bn_finalize folds them in a fixed order into (mean, rstd)
// and the CONSUMER applies relu((x-mean)*rstd) while it stages its input tile.  That keeps
// one pass over HBM per layer although the normalisation needs a frame-wide reduction.
//
//   k_conv_direct : block1 (C_in 1/4/8): one output pixel per lane, weights through the
//                   scalar cache, input tile (+halo) in LDS.  HBM/latency bound.
//   k_conv_mfma   : every other conv as an implicit GEMM on v_mfma_f32_32x32x2_f32:
//                   M = output pixels (32 per wave), N = C_out (32 per MFMA tile),
//                   K = (ky,kx,ci).  The activated input tile (+halo) sits in LDS for the
//                   whole K loop; weights stream through a double-buffered LDS chunk.
//   k_bn_finalize : partials -> mean / rstd.
//
// Numerics: each output is ONE fp32 fma chain in (ky,kx,ci) order starting at 0, which is
// exactly what the f32 MFMA computes when K is walked in order with a single accumulator,
// and what oracle/xfeat_oracle.c does; statistics are fp64.
#include "ctx.h"
#include "heads_kp4.hip.h"
#include <stdlib.h>

struct ConvArgs {
    const float* in;        // raw input slab (or plain input)
    StatSrc st;             // statistics of `in` (PRO_BN / PRO_B2IN), InstanceNorm statistics of the image (PRO_IN: st.stat = [B][2])
    size_t in_stride;       // floats per frame
    int Hin, Win;
    const float* w;         // packed weights
    const float* bias;      // EPI_BIAS only
    float* out;             // raw output slab
    size_t out_stride;
    int Hout, Wout;
    double* part;           // [B][npart][COUT][2]
    size_t part_stride;     // doubles per frame
    int tiles_x;
    // PRO_B2IN (block2.0: input = relu(bn(block1.3)) + skip1(x), XFeat.cc:153): pooled image and the 1x1 skip convolution
    const float* pool; size_t pool_stride; const float* skip_w; const float* skip_b;
    // PRO_FUSE (block_fusion.0: input = x3 + up2(x4) + up4(x5), XFeat.cc:159-166): `in` / st are block3.2 (x3)
    const float* r4; size_t s4; int H4, W4; StatSrc st4;      // block4.2
    const float* r5; size_t s5; int H5, W5; StatSrc st5;      // block5.3
    // PRO_L0 (block1.1: its input relu(bn(block1.0)) is recomputed from the image while staging): `in` = image X, st = statistics of
    // block1.0, xstat = InstanceNorm statistics [B][2], w0 = block1.0 weights [9][4], bias0 = its folded-BN bias (EPI_BIAS_RELU) or null
    const float* xstat; const float* w0; const float* bias0;
    int dbg;                                                   // layer index (XFH_STAMPS builds)
};

// PRO_PLAIN: input used as is; PRO_BN: relu((x - mean) * rstd) of the producer; PRO_IN: InstanceNorm of the image;
// PRO_B2IN / PRO_FUSE: the two element-wise glue steps of the backbone computed while staging (no intermediate tensor)
// PRO_UNFOLD (keypoint_head.0): the input is unfold2d(x-hat, 8) (XFeat.cc:124-133) read straight from the image: channel 8*dy + dx of
// cell (cy, cx) is pixel (8*cy + dy, 8*cx + dx), InstanceNorm applied while staging; `in` = image X, a.xstat = its statistics
enum { PRO_PLAIN = 0, PRO_BN = 1, PRO_IN = 2, PRO_B2IN = 3, PRO_FUSE = 4, PRO_L0 = 5, PRO_UNFOLD = 6, PRO_FUSEA = 7 /* PRO_FUSE on x4 / x5 maps that k_act_pyramid has already activated */ };
// EPI_STATS: raw map + fp64 statistic partials (BasicLayer in batch-statistics mode); EPI_BIAS: + bias, no statistics
// (block_fusion.2); EPI_BIAS_RELU: relu(. + bias) -- a BasicLayer whose BatchNorm was folded into weights and bias at load
// (XFH_BN_RUNNING_FOLDED): the stored map is already activated and its consumers see identity statistics
enum { EPI_STATS = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2 };

// development only (make STAMPS=1, tools/phase_stamps.py): wall-clock stamps (100 MHz) of the phases of one workgroup per layer
#ifdef XFH_STAMPS
__device__ unsigned long long g_stamps[32 * 8];
#define XFH_STAMP(a, ph) do { if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_stamps[((a).dbg & 31) * 8 + (ph)] = wall_clock64(); } while (0)
extern "C" int xfh_debug_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof g_stamps); }
#define XFH_STAMP_ROW(row, ph) do { if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_stamps[(row) * 8 + (ph)] = wall_clock64(); } while (0)
#else
#define XFH_STAMP(a, ph) do { } while (0)
#define XFH_STAMP_ROW(row, ph) do { } while (0)
#endif

// ATen upsample_bilinear2d (align_corners=false) source index / weights; see oracle lin_coeff
__device__ __forceinline__ void lin_coeff_c(int in, int out, int d, int& i0, int& i1, float& l0, float& l1) {
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
// 8 channels (group g) of relu(bn(raw)) at one pixel; st: LDS mean[64], rstd[64]
template <bool ACT = false>       // ACT: the map holds activated values already (k_act_pyramid)
__device__ __forceinline__ void ld_act8(const float* __restrict__ raw, const float* st, size_t pix, int g, f32x4& v0, f32x4& v1) {
    const float* p = raw + pix * 64 + g * 8;
    v0 = *(const f32x4*)p; v1 = *(const f32x4*)(p + 4);
    if constexpr (ACT) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v0[q] = fmaxf(fmaf(v0[q], st[64 + g * 8 + q], st[g * 8 + q]), 0.f);
        v1[q] = fmaxf(fmaf(v1[q], st[64 + g * 8 + 4 + q], st[g * 8 + 4 + q]), 0.f);
    }
}
// bilinear sample (ATen arithmetic: fma(w0, v0, w1*v1) per axis) of relu(bn(raw)) for 8 channels
template <bool ACT = false>
__device__ __forceinline__ void up_bilinear8(const float* __restrict__ raw, const float* st, int Hi, int Wi, int Ho, int Wo, int y, int x, int g, f32x4& o0, f32x4& o1) {
    int y0, y1, x0, x1; float hy0, hy1, wx0, wx1;
    lin_coeff_c(Hi, Ho, y, y0, y1, hy0, hy1);
    lin_coeff_c(Wi, Wo, x, x0, x1, wx0, wx1);
    f32x4 a0, a1, b0, b1, c0, c1, d0, d1;
    ld_act8<ACT>(raw, st, (size_t)y0 * Wi + x0, g, a0, a1); ld_act8<ACT>(raw, st, (size_t)y0 * Wi + x1, g, b0, b1);
    ld_act8<ACT>(raw, st, (size_t)y1 * Wi + x0, g, c0, c1); ld_act8<ACT>(raw, st, (size_t)y1 * Wi + x1, g, d0, d1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float t0 = fmaf(wx0, a0[j], wx1 * b0[j]), u0 = fmaf(wx0, c0[j], wx1 * d0[j]);
        o0[j] = fmaf(hy0, t0, hy1 * u0);
        const float t1 = fmaf(wx0, a1[j], wx1 * b1[j]), u1 = fmaf(wx0, c1[j], wx1 * d1[j]);
        o1[j] = fmaf(hy0, t1, hy1 * u1);
    }
}

// ------------------------------------------------------------------------------------
// k_act_pyramid (batches > 32): relu(bn(.)) of the two coarse maps of the pyramid, x4 (1/16) and x5 (1/32), written once.  block_fusion.0 builds its
// input x3 + up2(x4) + up4(x5) while staging (PRO_FUSE): every staged value takes eight bilinear taps, and activating each tap on the fly cost 3 VALU
// instructions per tap and channel -- half of that kernel's extra vector work over a plain 3x3 layer, on the pipe its MFMAs use -- for maps that are
// 1/4 and 1/16 of its own input.  Same fma + max per element, computed once instead of at every use: identical bits.  Needs finalised statistics
// (k_bn_finalize or the weight file's), hence not for batches <= 8.
__global__ __launch_bounds__(256)
void k_act_pyramid(const float* __restrict__ raw4, size_t s4, const float* __restrict__ st4, int n4 /* floats per frame */,
                   const float* __restrict__ raw5, size_t s5, const float* __restrict__ st5, int n5, float* __restrict__ act4, float* __restrict__ act5) {
    const int b = blockIdx.z;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < (n4 + n5) / 4; q += gridDim.x * 256) {
        const bool five = q >= n4 / 4;
        const int e = (five ? q - n4 / 4 : q) * 4, ch = e & 63;
        const float* st = (five ? st5 : st4) + (size_t)b * 128;
        const size_t off = (size_t)b * (five ? s5 : s4) + e;
        f32x4 v = *(const f32x4*)((five ? raw5 : raw4) + off);
        const f32x4 be = *(const f32x4*)(st + ch), al = *(const f32x4*)(st + 64 + ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], al[j], be[j]), 0.f);
        *(f32x4*)((five ? act5 : act4) + off) = v;
    }
}

// ------------------------------------------------------------------------------------
// statistics finalisation: grid = B, block = 256.  thread (c, j): channel c, slice j.
__global__ __launch_bounds__(256)
void k_bn_finalize(const double* __restrict__ part, size_t part_stride, int npart, int C, double count,
                   float* __restrict__ stat /* [B][2*C] */) {
    __shared__ double red[256 * 2];
    const int b = blockIdx.x;
    bn_fold<16>(part + (size_t)b * part_stride, npart, C, count, stat + (size_t)b * 2 * C, red, threadIdx.x, 256);
}

// ------------------------------------------------------------------------------------
// block1.0 (Conv 1->4, 3x3 on the InstanceNorm'ed image) is never written to HBM: at full resolution its 4-channel map is
// the largest tensor of the network (4.9 MB per VGA frame, written once and read once = 2.5 GB per 256-frame batch), while
// computing it costs 36 fma per pixel.  k_block1_stats makes the one pass its BatchNorm statistics need (image in, fp64
// partials out), and block1.1 recomputes the values it consumes while staging its input tile (PRO_L0 in k_conv_direct).
// Both evaluate l0_conv() on the same normalised pixels, so the statistics describe exactly the values that are used.
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}
// win: 3x3 window of the normalised image (zero outside: Conv2d padding); w: [9][4], wave-uniform.  Two channels per
// v_pk_fma_f32 (the same IEEE fma per channel as the scalar chain, in the same tap order).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void l0_conv(const float (&win)[9], const XFH_CONST float* __restrict__ w, float (&acc)[4]) {
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const f32x2 v = {win[k], win[k]};
        a0 = __builtin_elementwise_fma(v, f32x2{w[k * 4], w[k * 4 + 1]}, a0);
        a1 = __builtin_elementwise_fma(v, f32x2{w[k * 4 + 2], w[k * 4 + 3]}, a1);
    }
    acc[0] = a0.x; acc[1] = a0.y; acc[2] = a1.x; acc[3] = a1.y;
}

// grid (tiles of 128 x 32 pixels, 1, B), 256 threads = 32 x 8: a thread owns a strip of 4 pixels x 4 rows and slides a
// 3 x 6 register window down it; the image comes straight from global memory (one 16-byte load + the two halo pixels per
// row, the overlap between neighbours is served by the vector L1) -- no LDS staging, one barrier for the 4-wave fold.
constexpr int L0S_TW = 128, L0S_TH = 32, L0S_R = 4;
// The thread's 4 x 4 block is also one cell of skip1's AvgPool2d(4) (XFeat.cc:36-39): its mean of the normalised pixels goes to `pool`
// (same summation order as a row-major loop over the block), and for small batches the kernel folds and publishes the image
// statistics it normalises with (stage_stat) -- so no other kernel has to read the image between k_preproc and block1.
__global__ __launch_bounds__(256)
void k_block1_stats(const float* __restrict__ X, size_t x_stride, StatSrc xs, int H, int W, int tiles_x,
                    const float* __restrict__ w0, double* __restrict__ part, size_t part_stride, float* __restrict__ pool, size_t pool_stride) {
    __shared__ double s_red[512];
    __shared__ float s_xst[2];
    const int t = threadIdx.x, b = blockIdx.z, tile = blockIdx.x;
    XFH_STAMP_ROW(0, 0);
    const int gx = (tile % tiles_x) * L0S_TW + (t & 31) * 4, gy0 = (tile / tiles_x) * L0S_TH + (t >> 5) * L0S_R;
    const float* x = X + (size_t)b * x_stride;
    // the thread's L0S_R + 2 rows of the raw image, columns gx-1 .. gx+4: every load is issued here, before the statistics are folded
    // and without a branch around it (clamped addresses; what lies outside the image is zeroed below) -- behind the fold and a branch per
    // row they ran as seven dependent memory round trips, most of this kernel's single-frame time
    f32x4 rv[L0S_R + 2]; float rl[L0S_R + 2], rr[L0S_R + 2];
    {
        const int cx = min(gx, W - 4), cl = max(min(gx, W) - 1, 0), cr = min(gx + 4, W - 1);
#pragma unroll
        for (int i = 0; i < L0S_R + 2; ++i) {
            const float* p = x + (size_t)min(max(gy0 - 1 + i, 0), H - 1) * W;
            rv[i] = *(const f32x4*)(p + cx); rl[i] = p[cl]; rr[i] = p[cr];
        }
    }
    XFH_STAMP_ROW(0, 1);
    stage_stat<16>(xs, b, 1, tile == 0, s_xst, s_red, t, 256);
    XFH_STAMP_ROW(0, 2);
    const float m = s_xst[0], r = s_xst[1];
    // row i (image row gy0 - 1 + i) of the normalised, zero-padded image at columns gx-1 .. gx+4 (W % 4 == 0: the 16-byte load is all in or all out)
    auto load_row = [&](int i, float (&row)[6]) {
        const int gy = gy0 - 1 + i;
        const bool iny = gy >= 0 && gy < H;
        const f32x4 v = rv[i];
        const bool inx = iny && gx < W;
        row[1] = inx ? fmaf(v.x, r, m) : 0.f; row[2] = inx ? fmaf(v.y, r, m) : 0.f; row[3] = inx ? fmaf(v.z, r, m) : 0.f; row[4] = inx ? fmaf(v.w, r, m) : 0.f;
        row[0] = (iny && gx > 0 && gx - 1 < W) ? fmaf(rl[i], r, m) : 0.f;
        row[5] = (iny && gx + 4 < W) ? fmaf(rr[i], r, m) : 0.f;
    };
    float win[3][6];
    load_row(0, win[0]);
    load_row(1, win[1]);
    double sum[4] = {0.0, 0.0, 0.0, 0.0}, sq[4] = {0.0, 0.0, 0.0, 0.0};
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < L0S_R; ++i) {
        load_row(i + 2, win[(i + 2) % 3]);
        const float (&r0)[6] = win[i % 3], (&r1)[6] = win[(i + 1) % 3], (&r2)[6] = win[(i + 2) % 3];
#pragma unroll
        for (int q = 0; q < 4; ++q) ps += r1[q + 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float wq[9] = {r0[q], r0[q + 1], r0[q + 2], r1[q], r1[q + 1], r1[q + 2], r2[q], r2[q + 1], r2[q + 2]};
            float acc[4];
            l0_conv(wq, (const XFH_CONST float*)w0, acc);
            if (gy0 + i < H && gx + q < W) {
#pragma unroll
                for (int co = 0; co < 4; ++co) { const double d = (double)acc[co]; sum[co] += d; sq[co] = fma(d, d, sq[co]); }
            }
        }
    }
    XFH_STAMP_ROW(0, 4);
#pragma unroll
    for (int co = 0; co < 4; ++co) { sum[co] = wave_sum_f64(sum[co]); sq[co] = wave_sum_f64(sq[co]); }
    if (gy0 < H && gx < W) pool[(size_t)b * pool_stride + (size_t)(gy0 >> 2) * (W >> 2) + (gx >> 2)] = ps / 16.0f;
    if ((t & 63) == 0) {
#pragma unroll
        for (int co = 0; co < 4; ++co) { s_red[(t >> 6) * 8 + co * 2] = sum[co]; s_red[(t >> 6) * 8 + co * 2 + 1] = sq[co]; }
    }
    __syncthreads();
    if (t < 8)
        part[(size_t)b * part_stride + (size_t)tile * 8 + t] = (s_red[t] + s_red[8 + t]) + (s_red[16 + t] + s_red[24 + t]);
    XFH_STAMP_ROW(0, 5);
}

// ------------------------------------------------------------------------------------
// direct convolution 3x3 for block1.  16x16 output pixels per workgroup.
// FOLD (small batches, PRO_BN): the workgroup folds the producer's statistic partials itself (LDS); otherwise the
// finalised statistics come through the scalar cache -- staging them through LDS first would put a second dependent
// memory round trip into every workgroup's latency chain, which is what bounds these kernels (measured 3.5x slower).
// workgroup x of n -> tile: the workgroups x = k (mod 8) -- one XCD -- take the k-th contiguous eighth of the tile sequence (a bijection)
__device__ __forceinline__ int xcd_tile(int x, int n) {
    const int k = x & 7, q = n >> 3, r = n & 7;
    return k * q + (k < r ? k : r) + (x >> 3);
}

template <int CIN, int COUT, int ST, int PRO, bool FOLD, int EPI>
__global__ __launch_bounds__(256)
void k_conv_direct(ConvArgs a) {
    constexpr int TI = 15 * ST + 3;
    constexpr int SL = 256 / COUT;
    static_assert(PRO != PRO_L0 || (CIN == 4 && ST == 2 && COUT == 8), "PRO_L0 is block1.1");
    constexpr int XW = TI + 2, XS = XW + 1;      // PRO_L0: image window of the tile's block1.0 inputs, row stride
    // LDS: the input tile; the statistics transposition of the epilogue reuses it (one more barrier) -- at 22 KB (block1.1) / 11 KB
    // (block1.2) a workgroup fits beside two workgroups of the dominant convolution (134 of 160 KB) when several ctx share the GPU
    constexpr int SIN = TI * TI * CIN > 256 * (COUT + 1) ? TI * TI * CIN : 256 * (COUT + 1);
    __shared__ __attribute__((aligned(16))) float s_in[SIN];
    float* s_out = s_in;
    __shared__ __attribute__((aligned(16))) float s_xwin[PRO == PRO_L0 ? XW * XS : 1];
    __shared__ double s_red[FOLD ? 512 : 8 * COUT];
    // FOLD (batches <= 8, latency bound): the 9 x CIN x COUT weights in LDS -- every lane reads the same address (a broadcast), operands
    // arrive as registers.  Through a plain global pointer each weight quad is a 64-lane global_load with identical addresses, 144 per
    // wave: 5 us of address-unit time in a single frame (as scalar loads the SGPR file holds an eighth of them at a time and the FMAs wait
    // for every batch: 1 us slower than LDS).  Large batches keep the global loads: there the kernel is bound by the image traffic, the
    // weight quads hit the vector L1, and the LDS reads measured 340 -> 540 us at 256 frames.
    constexpr bool WLDS = FOLD;
    constexpr int NWQ = 9 * CIN * COUT / 4;
    __shared__ __attribute__((aligned(16))) float s_wt[WLDS ? NWQ * 4 : 4];
    const int t = threadIdx.x, b = blockIdx.z;
    XFH_STAMP(a, 0);
    // XCD-aware tile order.  Workgroups go round robin over the eight XCDs (each with its own L2), so with the plain order two
    // neighbouring tiles never share an L2 and every halo line is fetched twice from the fabric: calibrated counters
    // (profiles/pmc_traffic.json) read 2.55x the algorithmic fetch bytes for block1.2, and at that traffic the kernel sits at 84 % of the
    // memory rate.  Workgroup x therefore takes tile  (x & 7) * n / 8 + (x >> 3): XCD k walks a contiguous eighth of the tile sequence.
    const int tile = xcd_tile(blockIdx.x, gridDim.x), tx0 = (tile % a.tiles_x) * 16, ty0 = (tile / a.tiles_x) * 16;
    const float* in = a.in + (size_t)b * a.in_stride;
    __shared__ float s_stat[FOLD ? 2 * CIN : 1];
    static_assert(NWQ <= 256, "one weight quad per thread");
    f32x4 wq4 = {0.f, 0.f, 0.f, 0.f};
    if (WLDS && t < NWQ) wq4 = *(const f32x4*)(a.w + t * 4);
    // (FOLD: the producer's statistics are folded below, AFTER this workgroup's input loads are in flight -- one memory round trip, not two)
    const float* st = a.st.stat + (size_t)b * 2 * (PRO == PRO_IN ? 1 : CIN);                      // wave-uniform: scalar loads

    // stage the activated input tile, zero outside the image (Conv2d zero padding)
    constexpr int VEC = (CIN >= 4) ? 4 : 1;
    constexpr int G = CIN / VEC;
    if constexpr (PRO == PRO_L0) {
        // block1.0 recomputed: normalised image window -> conv 1->4 -> BatchNorm + ReLU -> s_in (see k_block1_stats)
        float* s_x = s_xwin;
        const float xm = a.xstat[b * 2], xr = a.xstat[b * 2 + 1];
        // all loads of a thread are issued before the first use (clamped addresses, no branches around them)
        constexpr int NX = (XW * XW + 255) / 256;
        float xv[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int item = t + k * 256, iy = item / XW, ix = item % XW;
            const int gy = ty0 * ST - 2 + iy, gx = tx0 * ST - 2 + ix;
            xv[k] = in[(size_t)min(max(gy, 0), a.Hin - 1) * a.Win + min(max(gx, 0), a.Win - 1)];
        }
        if constexpr (FOLD) stage_stat<16>(a.st, b, CIN, tile == 0, s_stat, s_red, t, 256);            // s_red is free until the epilogue
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int item = t + k * 256, iy = item / XW, ix = item % XW;
            const int gy = ty0 * ST - 2 + iy, gx = tx0 * ST - 2 + ix;
            const bool ok = gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
            if (item < XW * XW) s_x[iy * XS + ix] = ok ? fmaf(xv[k], xr, xm) : 0.f;
        }
        __syncthreads();
        for (int pix = t; pix < TI * TI; pix += 256) {
            const int iy = pix / TI, ix = pix % TI;
            const int gy = ty0 * ST - 1 + iy, gx = tx0 * ST - 1 + ix;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) {
                const float* sp = s_x + iy * XS + ix;
                const float win[9] = {sp[0], sp[1], sp[2], sp[XS], sp[XS + 1], sp[XS + 2], sp[2 * XS], sp[2 * XS + 1], sp[2 * XS + 2]};
                float acc[4];
                l0_conv(win, (const XFH_CONST float*)a.w0, acc);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if constexpr (EPI == EPI_BIAS_RELU) v[q] = fmaxf(acc[q] + a.bias0[q], 0.f);
                    else {
                        const float m = FOLD ? s_stat[q] : st[q], r = FOLD ? s_stat[CIN + q] : st[CIN + q];
                        v[q] = fmaxf(fmaf(acc[q], r, m), 0.f);
                    }
                }
            }
            *(f32x4*)(s_in + pix * CIN) = v;
        }
    } else
    {
        // a thread stages whole pixels (the channel group index is then a compile-time constant and the statistics stay in
        // scalar registers); all loads are issued before the first use: clamped addresses, no branches around them
        constexpr int NI = (TI * TI + 255) / 256;
        f32x4 iv[NI][G];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int pix = t + k * 256;
            const int gy = min(max(ty0 * ST - 1 + pix / TI, 0), a.Hin - 1), gx = min(max(tx0 * ST - 1 + pix % TI, 0), a.Win - 1);
            const float* p = in + ((size_t)gy * a.Win + gx) * CIN;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if constexpr (VEC == 4) iv[k][g] = *(const f32x4*)(p + g * 4);
                else iv[k][g].x = p[0];
            }
        }
        if constexpr (FOLD) stage_stat<16>(a.st, b, CIN, tile == 0, s_stat, s_red, t, 256);            // s_red is free until the epilogue
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int pix = t + k * 256;
            const int gy = ty0 * ST - 1 + pix / TI, gx = tx0 * ST - 1 + pix % TI;
            const bool ok = gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
            if (pix >= TI * TI) continue;
            if constexpr (VEC == 4) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    f32x4 v = iv[k][g];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float m = FOLD ? s_stat[g * 4 + q] : st[g * 4 + q], r = FOLD ? s_stat[CIN + g * 4 + q] : st[CIN + g * 4 + q];
                        v[q] = ok ? fmaxf(fmaf(v[q], r, m), 0.f) : 0.f;
                    }
                    *(f32x4*)(s_in + pix * CIN + g * 4) = v;
                }
            } else {
                s_in[pix] = ok ? fmaf(iv[k][0].x, st[1], st[0]) : 0.f;        // InstanceNorm, no ReLU
            }
        }
    }
    if (WLDS && t < NWQ) *(f32x4*)(s_wt + t * 4) = wq4;
    __syncthreads();
    XFH_STAMP(a, 3);

    const int tx = t & 15, ty = t >> 4;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float* sp = s_in + ((ty * ST + ky) * TI + tx * ST + kx) * CIN;
            float v[CIN];
            if constexpr (VEC == 4) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const f32x4 q = *(const f32x4*)(sp + g * 4);
                    v[g * 4 + 0] = q.x; v[g * 4 + 1] = q.y; v[g * 4 + 2] = q.z; v[g * 4 + 3] = q.w;
                }
            } else {
                v[0] = sp[0];
            }
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                if constexpr (WLDS) {
                    const float* wr = s_wt + ((ky * 3 + kx) * CIN + ci) * COUT;
#pragma unroll
                    for (int g4 = 0; g4 < COUT / 4; ++g4) {
                        const f32x4 w4 = *(const f32x4*)(wr + g4 * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[g4 * 4 + e] = fmaf(v[ci], w4[e], acc[g4 * 4 + e]);
                    }
                } else {
                    const float* wr = a.w + ((ky * 3 + kx) * CIN + ci) * COUT;     // wave-uniform (scalar loads through XFH_CONST measured the same at 256 frames)
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[co] = fmaf(v[ci], wr[co], acc[co]);
                }
            }
        }

    XFH_STAMP(a, 4);
    const int oy = ty0 + ty, ox = tx0 + tx;
    const bool valid = oy < a.Hout && ox < a.Wout;
    if constexpr (EPI == EPI_BIAS_RELU) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = fmaxf(acc[co] + a.bias[co], 0.f);      // folded BatchNorm + ReLU
    }
    if (valid) {
        float* o = a.out + (size_t)b * a.out_stride + ((size_t)oy * a.Wout + ox) * COUT;
#pragma unroll
        for (int g = 0; g < COUT / 4; ++g)
            *(f32x4*)(o + g * 4) = f32x4{acc[g * 4], acc[g * 4 + 1], acc[g * 4 + 2], acc[g * 4 + 3]};
    }
    if constexpr (EPI != EPI_STATS) return;
    // per-channel fp64 partial sums of this tile: thread (c, j) adds the pixels j, j + SL, ... of channel c, the slices
    // of a wave are folded with shuffles and the four waves through LDS -- a fixed order, and no serial chain
    __syncthreads();                                      // every thread is done with the input tile: its memory becomes s_out
#pragma unroll
    for (int co = 0; co < COUT; ++co) s_out[t * (COUT + 1) + co] = valid ? acc[co] : 0.f;
    __syncthreads();
    {
        const int c = t % COUT, j = t / COUT;
        double s = 0.0, ss = 0.0;
#pragma unroll
        for (int p = 0; p < 256 / SL; ++p) {
            const double v = (double)s_out[(j + p * SL) * (COUT + 1) + c];
            s += v; ss = fma(v, v, ss);
        }
#pragma unroll
        for (int off = COUT; off < 64; off <<= 1) { s += __shfl_xor(s, off); ss += __shfl_xor(ss, off); }
        if ((t & 63) < COUT) { s_red[((t >> 6) * COUT + c) * 2] = s; s_red[((t >> 6) * COUT + c) * 2 + 1] = ss; }
    }
    __syncthreads();
    if (t < 2 * COUT)
        a.part[(size_t)b * a.part_stride + (size_t)tile * COUT * 2 + t] = (s_red[t] + s_red[2 * COUT + t]) + (s_red[4 * COUT + t] + s_red[6 * COUT + t]);
    XFH_STAMP(a, 5);
}

// Epilogue shared by k_conv_mfma and k_conv_mfma_p.  C/D layout of v_mfma_f32_32x32x2_f32: a lane holds channel (lane&31) of
// tile n and the pixels (r&3) + 8*(r>>2) + 4*h of the wave's WH x WW pixel block.  The pixel row / column of register r
// splits into a compile-time part and 4*h (WW >= 8), so the store address is  wave-uniform base + scalar offset(r) +
// one per-lane 32-bit offset -- no 64-bit vector arithmetic and, for tiles that lie inside the map, no predicates.
// (The conv kernels issue their VALU instructions on the same pipe as the f32 MFMA: PMC showed 2.5-9 VALU
// instructions per MFMA in these kernels, most of them address arithmetic.)  Statistics: fp64 (sum, sum^2) per channel
// over the lane's valid pixels in register order -- the order is part of the numerics contract.
// Weight chunks through BUFFER loads: resource = the layer's weight matrix (wave-uniform, four SGPRs), per-lane 32-bit byte offset, the chunk as the
// scalar offset -- `buffer_load_dwordx4 v, v_off, s[rsrc], s_chunk offen`, no vector arithmetic per load.  As `base + (size_t)chunk * WCH + (size_t)f * 4`
// every chunk's loads carried a 64-bit vector add of their own (v_lshl_add_u64 / v_add_co + v_addc: ~50 VALU instructions per wave and tile in the
// 3x3 64 -> 64 instance, on the pipe the MFMAs use).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wbuf_make(const float* w) { return __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ f32x4 wbuf_ld(__amdgpu_buffer_rsrc_t r, int f /* quad index of the lane */, int chunk_float_off /* wave-uniform */) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)f * 16u, chunk_float_off * 4, 0));
}
// the lane's NT bias values; issued before the K loop so that the epilogue never waits for a global load
template <int COUT, int NT, int EPI>
__device__ __forceinline__ void conv_bias(const float* __restrict__ bias, int co0, float (&bv)[NT]) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        bv[n] = 0.f;
        if constexpr (EPI != EPI_STATS) bv[n] = (COUT % 32 == 0 || co0 + n * 32 < COUT) ? bias[co0 + n * 32] : 0.f;
    }
}

template <int COUT, int NT, int WW, int WH, int EPI>
__device__ __forceinline__ void conv_epilogue(const f32x16 (&acc)[NT], float* __restrict__ wave_out /* pixel (0,0) of the wave's block */, int Wout,
                                              int rows_left, int cols_left, int co0, int h, const float (&bias)[NT] /* conv_bias(): loaded before the K loop */,
                                              double (&sum)[NT], double (&sq)[NT]) {
    static_assert(WW >= 8 && WW * WH == 32, "pixel block of a wave");
    XFH_MFMA_SETTLE();                                              // common.h: the epilogue branches
    const bool full = rows_left >= WH && cols_left >= WW;          // wave-uniform
    const int lane_off = 4 * h * COUT + co0;
    const int rowstride = Wout * COUT;
    auto value = [&](int n, int r, float* dst, float bv) {
        float v = acc[n][r];
        if constexpr (EPI != EPI_STATS) v += bv;
        if constexpr (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
        dst[lane_off] = v;
        if constexpr (EPI == EPI_STATS) { const double dv = (double)v; sum[n] += dv; sq[n] = fma(dv, dv, sq[n]); }
    };
    // Two copies of the loop under ONE wave-uniform branch: written as `ok = full ? cok : (...)` inside a single loop the compiler kept a
    // predicate (v_cmp / v_cndmask / s_and_saveexec / s_cbranch_execz) around every one of the 16 * NT values also for the tiles inside the map
    // -- 50 VALU instructions and 80 scalar ones per wave and tile on the pipe the MFMAs use (a seventh of the 1x1 layers' vector work).
    if (full) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            sum[n] = 0.0; sq[n] = 0.0;
            if (COUT % 32 == 0 || co0 + n * 32 < COUT) {          // (one predicate per channel tile where the channels are padded)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pb = (r & 3) + 8 * (r >> 2);         // + 4*h
                    value(n, r, wave_out + (pb / WW) * rowstride + (pb % WW) * COUT + n * 32, bias[n]);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = co0 + n * 32;
        sum[n] = 0.0; sq[n] = 0.0;
        const float bv = bias[n];
        const bool cok = COUT % 32 == 0 || co < COUT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pb = (r & 3) + 8 * (r >> 2);                 // + 4*h
            const int prow = pb / WW, pcol = pb % WW;              // compile-time; pcol + 4*h < WW
            float* dst = wave_out + prow * rowstride + pcol * COUT + n * 32;      // uniform
            if (cok && prow < rows_left && pcol + 4 * h < cols_left) value(n, r, dst, bv);
        }
    }
}

// ------------------------------------------------------------------------------------
// implicit-GEMM convolution on the f32 matrix cores.
//   WM x WN waves per workgroup; each wave owns 32 output pixels (WH x WW) and NT tiles of 32
//   output channels.  COUTP = WN*NT*32 >= COUT (padded weight rows are zero).
#ifndef XFH_PD
#define XFH_PD 3
#endif

// The body is a device function of (tile, frame) so that it can also run as a RIDER in another layer's launch (k_conv_mfma_ride below).
// `a` is taken BY VALUE: through a `const ConvArgs&` the compiler no longer reads the kernel arguments with scalar loads -- the PRO_FUSE
// instance went from 127 to 166 VGPRs (two workgroups per CU to one: 950 -> 1230 us at 256 frames).
template <int CIN, int COUT, int KS, int ST, int WM, int WN, int NT, int WW, int PRO, int EPI, int CBMAX = 64, int TPC = 1, int PD = XFH_PD>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs a, const int tile, const int b, float* smem) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int WH = 32 / WW, TH = WM * WH, TW = WW;
    constexpr int COUTP = WN * NT * 32;
    constexpr int PAD = KS / 2;
    constexpr int TIH = (TH - 1) * ST + KS, TIW = (TW - 1) * ST + KS;
    constexpr int CP = CIN + 4;                  // LDS pixel stride (floats)
    constexpr int CB = CIN > CBMAX ? CBMAX : CIN;   // channels per weight chunk and tap
    constexpr int NCB = CIN / CB;
    static_assert(TPC == 1 || NCB == 1, "several taps per chunk only with all channels in the chunk");
    static_assert((KS * KS) % TPC == 0, "taps per chunk");
    constexpr int NCHUNK = KS * KS * NCB / TPC;
    constexpr int NWBUF = NCHUNK > 1 ? 2 : 1;
    constexpr int KC = TPC * CB;                 // K elements per chunk: TPC taps x CB channels
    constexpr int WS = KC + 4;                   // LDS weight row stride (WS/4 odd: conflict-free b128 reads)
    constexpr int WCH = COUTP * KC;              // floats per chunk in global memory
    constexpr int NWLD = (WCH / 4 + NTHR - 1) / NTHR;
    constexpr int G = CIN / 8;
    constexpr int IN_FLOATS = TIH * TIW * CP;
    constexpr int W_FLOATS = COUTP * WS;

    float* s_in = smem;
    float* s_w = smem + IN_FLOATS;               // two buffers of W_FLOATS
    float* s_stat = s_w + NWBUF * W_FLOATS;      // 2*CIN floats (PRO_BN), 4*CIN (PRO_B2IN: + skip weights / bias), 3*128 (PRO_FUSE)

    const int t = threadIdx.x;
    const int tx0 = (tile % a.tiles_x) * TW, ty0 = (tile / a.tiles_x) * TH;
    const float* in = a.in + (size_t)b * a.in_stride;

    XFH_STAMP(a, 0);
    // ---- issue the first weight chunk, stage producer statistics ------------------------
    // PD weight chunks are in flight in a ring of register sets (chunk c in set c % PD).  PD = 1 suffices when two or more workgroups
    // share a CU; the kernels that run one workgroup of four waves per CU waited at every chunk for the next one's weights (1.6 us per
    // chunk of 0.85 us of MFMAs in the stride-2 64 -> 64 layer at 256 frames)
    f32x4 wreg[PD][NWLD];
    const __amdgpu_buffer_rsrc_t wrs = wbuf_make(a.w);
#pragma unroll
    for (int c = 0; c < PD; ++c)
#pragma unroll
        for (int q = 0; q < NWLD; ++q) {
            const int f = t + q * NTHR;
            if (c < NCHUNK && f < WCH / 4) wreg[c][q] = wbuf_ld(wrs, f, c * WCH);
        }
    // ---- the raw input tile: every load of a thread is issued here, before the statistics are staged and before the first use
    // (clamped addresses, no branches around the loads: behind a per-item branch they run one memory round trip after the other,
    // and with one or two workgroups per CU nothing else covers that).  PRO_FUSE gathers its taps per item below.
    constexpr int NITEM = TIH * TIW * G;
    constexpr int NIT = (NITEM + NTHR - 1) / NTHR;
    constexpr bool FUSE = PRO == PRO_FUSE || PRO == PRO_FUSEA;
    constexpr bool BATCHED = !FUSE;
    f32x4 r0[BATCHED ? NIT : 1], r1[BATCHED ? NIT : 1];
    float rp[PRO == PRO_B2IN ? NIT : 1];
    if constexpr (BATCHED) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int item = t + k * NTHR, pix = item / G, g = item % G;
            const int gy = min(max(ty0 * ST - PAD + pix / TIW, 0), a.Hin - 1), gx = min(max(tx0 * ST - PAD + pix % TIW, 0), a.Win - 1);
            const float* p = PRO == PRO_UNFOLD ? in + ((size_t)(gy * 8 + g) * (a.Win * 8) + gx * 8) : in + ((size_t)gy * a.Win + gx) * CIN + g * 8;
            r0[k] = *(const f32x4*)p;
            r1[k] = *(const f32x4*)(p + 4);
            if constexpr (PRO == PRO_B2IN) rp[k] = a.pool[(size_t)b * a.pool_stride + (size_t)gy * a.Win + gx];
        }
    }
    XFH_STAMP(a, 1);
    // producer statistics -> LDS (folded here for small batches; s_in is still free and serves as fp64 scratch)
    if constexpr (PRO == PRO_BN || PRO == PRO_B2IN || FUSE) stage_stat(a.st, b, CIN, tile == 0, s_stat, (double*)s_in, t, NTHR);
    if constexpr (PRO == PRO_B2IN) {
        for (int q = t; q < CIN; q += NTHR) { s_stat[2 * CIN + q] = a.skip_w[q]; s_stat[3 * CIN + q] = a.skip_b[q]; }
        __syncthreads();
    }
    if constexpr (PRO == PRO_FUSE) {
        stage_stat(a.st4, b, 64, tile == 0, s_stat + 128, (double*)s_in, t, NTHR);
        stage_stat(a.st5, b, 64, tile == 0, s_stat + 256, (double*)s_in, t, NTHR);
    }
    XFH_STAMP(a, 2);
    // ---- stage the activated input tile (zero padding outside the image) ----------------
    if constexpr (BATCHED) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int item = t + k * NTHR, pix = item / G, g = item % G;
            const int gy = ty0 * ST - PAD + pix / TIW, gx = tx0 * ST - PAD + pix % TIW;
            const bool ok = gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
            if (NITEM % NTHR != 0 && item >= NITEM) continue;
            f32x4 v0 = r0[k], v1 = r1[k];
            if constexpr (PRO == PRO_UNFOLD) {
                const float xm = a.xstat[b * 2], xr = a.xstat[b * 2 + 1];
#pragma unroll
                for (int q = 0; q < 4; ++q) { v0[q] = fmaf(v0[q], xr, xm); v1[q] = fmaf(v1[q], xr, xm); }
            }
            if constexpr (PRO == PRO_BN || PRO == PRO_B2IN) {
                const f32x4 m0 = *(const f32x4*)(s_stat + g * 8), m1 = *(const f32x4*)(s_stat + g * 8 + 4);
                const f32x4 q0 = *(const f32x4*)(s_stat + CIN + g * 8), q1 = *(const f32x4*)(s_stat + CIN + g * 8 + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v0[q] = fmaxf(fmaf(v0[q], q0[q], m0[q]), 0.f);
                    v1[q] = fmaxf(fmaf(v1[q], q1[q], m1[q]), 0.f);
                }
            }
            if constexpr (PRO == PRO_B2IN) {            // x1 + skip1(x): AvgPool4 of the normalised image, 1x1 conv with bias (XFeat.cc:36-39,153)
                const float pl = rp[k];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v0[q] = v0[q] + (pl * s_stat[2 * CIN + g * 8 + q] + s_stat[3 * CIN + g * 8 + q]);
                    v1[q] = v1[q] + (pl * s_stat[2 * CIN + g * 8 + 4 + q] + s_stat[3 * CIN + g * 8 + 4 + q]);
                }
            }
            if constexpr (KS > 1) { if (!ok) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; } }       // Conv2d zero padding; a 1x1 layer's overhang pixels feed only outputs that are never stored
            // k permutation inside each group of 8: position 4*(k&1) + (k>>1)
            float* d = s_in + pix * CP + g * 8;
            *(f32x4*)d = f32x4{v0.x, v0.z, v1.x, v1.z};
            *(f32x4*)(d + 4) = f32x4{v0.y, v0.w, v1.y, v1.w};
        }
    } else
    for (int item = t; item < TIH * TIW * G; item += NTHR) {
        const int pix = item / G, g = item % G;
        const int iy = pix / TIW, ix = pix % TIW;
        const int gy = ty0 * ST - PAD + iy, gx = tx0 * ST - PAD + ix;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) {
            const float* p = in + ((size_t)gy * a.Win + gx) * CIN + g * 8;
            v0 = *(const f32x4*)p;
            v1 = *(const f32x4*)(p + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v0[q] = fmaxf(fmaf(v0[q], s_stat[CIN + g * 8 + q], s_stat[g * 8 + q]), 0.f);
                v1[q] = fmaxf(fmaf(v1[q], s_stat[CIN + g * 8 + 4 + q], s_stat[g * 8 + 4 + q]), 0.f);
            }
            // x3 + up2(x4) + up4(x5) with ATen's bilinear arithmetic (XFeat.cc:159-166)
            f32x4 u0, u1, w0, w1;
            up_bilinear8<PRO == PRO_FUSEA>(a.r4 + (size_t)b * a.s4, s_stat + 128, a.H4, a.W4, a.Hin, a.Win, gy, gx, g, u0, u1);
            up_bilinear8<PRO == PRO_FUSEA>(a.r5 + (size_t)b * a.s5, s_stat + 256, a.H5, a.W5, a.Hin, a.Win, gy, gx, g, w0, w1);
#pragma unroll
            for (int q = 0; q < 4; ++q) { v0[q] = (v0[q] + u0[q]) + w0[q]; v1[q] = (v1[q] + u1[q]) + w1[q]; }
        }
        // k permutation inside each group of 8: position 4*(k&1) + (k>>1)
        float* d = s_in + pix * CP + g * 8;
        *(f32x4*)d = f32x4{v0.x, v0.z, v1.x, v1.z};
        *(f32x4*)(d + 4) = f32x4{v0.y, v0.w, v1.y, v1.w};
    }
    // first weight chunk -> LDS buffer 0
#pragma unroll
    for (int q = 0; q < NWLD; ++q) {
        const int f = t + q * NTHR;
        if (f < WCH / 4) {
            const int n = f / (KC / 4), c4 = f % (KC / 4);
            *(f32x4*)(s_w + n * WS + c4 * 4) = wreg[0][q];
            if (PD < NCHUNK) wreg[0][q] = wbuf_ld(wrs, f, PD * WCH);
        }
    }
    __syncthreads();
    XFH_STAMP(a, 3);

    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int pr = i / WW, pc = i % WW;                     // pixel of this lane inside the wave tile
    const int ly = (wm * WH + pr) * ST, lx = pc * ST;        // its top-left input position in the tile

    float biasv[NT];
    conv_bias<COUT, NT, EPI>(a.bias, wn * NT * 32 + i, biasv);
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

#pragma unroll
    for (int ch = 0; ch < NCHUNK; ++ch) {
#pragma unroll
        for (int tt = 0; tt < TPC; ++tt) {
            const int tap = TPC == 1 ? ch / NCB : ch * TPC + tt, cb = TPC == 1 ? ch % NCB : 0;
            const int ky = tap / KS, kx = tap % KS;
            const float* pa = s_in + ((ly + ky) * TIW + lx + kx) * CP + cb * CB + 4 * h;
            const float* pw = s_w + (ch & 1) * W_FLOATS + (wn * NT * 32 + i) * WS + tt * CB + 4 * h;
#pragma unroll
            for (int kk = 0; kk < CB / 8; ++kk) {
                const f32x4 av = *(const f32x4*)(pa + kk * 8);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f32x4 bv = *(const f32x4*)(pw + n * 32 * WS + kk * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[n], 0, 0, 0);
                }
            }
        }
        if (ch + 1 < NCHUNK) {                              // chunk ch + 1 (issued PD chunks ago) -> the other LDS buffer; its register set takes chunk ch + 1 + PD
            float* wd = s_w + ((ch + 1) & 1) * W_FLOATS;
#pragma unroll
            for (int q = 0; q < NWLD; ++q) {
                const int f = t + q * NTHR;
                if (f < WCH / 4) {
                    const int n = f / (KC / 4), c4 = f % (KC / 4);
                    *(f32x4*)(wd + n * WS + c4 * 4) = wreg[(ch + 1) % PD][q];
                    if (ch + 1 + PD < NCHUNK) wreg[(ch + 1) % PD][q] = wbuf_ld(wrs, f, (ch + 1 + PD) * WCH);
                }
            }
        }
        __syncthreads();
    }

    XFH_STAMP(a, 4);
    // ---- epilogue ---------------------------------------------------------------------
    double sum[NT], sq[NT];
    {
        const int oy0 = ty0 + wm * WH;
        float* wave_out = a.out + (size_t)b * a.out_stride + ((size_t)oy0 * a.Wout + tx0) * COUT;
        conv_epilogue<COUT, NT, WW, WH, EPI>(acc, wave_out, a.Wout, a.Hout - oy0, a.Wout - tx0, wn * NT * 32 + i, h, biasv, sum, sq);
    }
    if constexpr (EPI == EPI_STATS) {
        double* s_red = (double*)smem;        // [WM][COUTP][2], the tiles are no longer needed
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            sum[n] += __shfl_xor(sum[n], 32);
            sq[n] += __shfl_xor(sq[n], 32);
            if (h == 0) {
                const int co = (wn * NT + n) * 32 + i;
                s_red[(wm * COUTP + co) * 2 + 0] = sum[n];
                s_red[(wm * COUTP + co) * 2 + 1] = sq[n];
            }
        }
        __syncthreads();
        for (int co = t; co < COUT; co += NTHR) {
            double S = 0.0, SS = 0.0;
#pragma unroll
            for (int m = 0; m < WM; ++m) { S += s_red[(m * COUTP + co) * 2 + 0]; SS += s_red[(m * COUTP + co) * 2 + 1]; }
            double* p = a.part + (size_t)b * a.part_stride + ((size_t)tile * COUT + co) * 2;
            p[0] = S; p[1] = SS;
        }
    }
    XFH_STAMP(a, 5);
}

template <int CIN, int COUT, int KS, int ST, int WM, int WN, int NT, int WW, int PRO, int EPI, int CBMAX = 64, int TPC = 1, int PD = XFH_PD>
__global__ __launch_bounds__(64 * WM * WN)
void k_conv_mfma(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_mfma_body<CIN, COUT, KS, ST, WM, WN, NT, WW, PRO, EPI, CBMAX, TPC, PD>(a, blockIdx.x, blockIdx.z, smem);
}

// k_conv_mfma_ride: a backbone layer (the HOST: its tiles are the workgroups below n_host) and, in the same launch, one step of the
// keypoint branch (the RIDER: the workgroups from n_host on).  For batches <= 8 (every BatchNorm mode but the folded one, whose layers
// have another epilogue) the branch
// -- keypoint_head.0-2 (1x1 convolutions on unfold2d(x-hat)) and keypoint_head.3 + softmax + depth-to-space -- does not get a stream of
// its own: its four dependent steps ride on block1.3, block2.0, block2.1 and block3.0, which follow each other on the one stream
// anyway.  Forking a second stream costs the main queue 7-11 us (a hipEventRecord there is a barrier packet with a signal; attaching
// the event to a kernel's completion signal measured the same) and any wait on another queue as much; a rider costs nothing but CUs
// that the host's 150 tiles leave idle.  Host and rider never touch each other's tensors.  256 threads: the kp4 rider uses the first 128
// (the other two waves leave at once; a terminated wave does not take part in s_barrier).
enum { RIDE_UNFOLD = 0 /* keypoint_head.0 */, RIDE_BN = 1 /* keypoint_head.1 / .2 */, RIDE_KP4 = 2 /* keypoint_head.3 + softmax */ };
// EPI: EPI_STATS (batch statistics / eval() statistics from the weight file) or EPI_BIAS_RELU (folded BatchNorms) for host and rider alike.
template <int CIN, int COUT, int KS, int ST, int WM, int WN, int NT, int WW, int PRO, int CBMAX, int TPC, int RIDER, int EPI = EPI_STATS>
__global__ __launch_bounds__(256)
void k_conv_mfma_ride(ConvArgs a, int n_host, ConvArgs r, Kp4Args k) {
    static_assert(WM * WN == 4, "256 threads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int blk = blockIdx.x;
    if (blk < n_host) conv_mfma_body<CIN, COUT, KS, ST, WM, WN, NT, WW, PRO, EPI, CBMAX, TPC>(a, blk, blockIdx.z, smem);
    else if constexpr (RIDER == RIDE_KP4) { if (threadIdx.x < 4 * HK4_PX) heads_kp4_body(k, blk - n_host, blockIdx.z, smem); }
    else conv_mfma_body<64, 64, 1, 1, 4, 1, 2, 16, RIDER == RIDE_UNFOLD ? PRO_UNFOLD : PRO_BN, EPI>(r, blk - n_host, blockIdx.z, smem);
}

// ------------------------------------------------------------------------------------
// k_conv_mfma_p: persistent form of k_conv_mfma for the layers whose whole weight matrix fits in LDS next to one
// input tile (the 1x1 layers, the 24->24 and 8->24 3x3 layers).  These layers have short K loops (32-108 MFMAs per
// wave and tile), so k_conv_mfma spends as long staging a tile as computing it.  Here a workgroup
//   - loads the weights ONCE and walks over tiles  blockIdx.x, blockIdx.x + gridDim.x, ...  (frames x tiles),
//   - issues the global loads of the NEXT tile before the MFMAs of the current one (registers hold them), and applies
//     BN + ReLU when it moves them to LDS afterwards,
//   - has no barrier inside the K loop (all taps resident).
// Same arithmetic, same tiles and the same statistics partials as k_conv_mfma: the two are interchangeable bit for bit.
template <int CIN, int COUT, int KS, int ST, int WM, int WN, int NT, int WW, int PRO, int EPI>
__global__ __launch_bounds__(64 * WM * WN)
void k_conv_mfma_p(ConvArgs a, int ntile, int total) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int WH = 32 / WW, TH = WM * WH, TW = WW;
    constexpr int COUTP = WN * NT * 32;
    constexpr int PAD = KS / 2;
    constexpr int TIH = (TH - 1) * ST + KS, TIW = (TW - 1) * ST + KS;
    constexpr int CP = CIN + 4;
    constexpr int KTOT = KS * KS * CIN;
    constexpr int WS = KTOT + 4;                 // WS/4 odd: conflict-free ds_read_b128
    constexpr int G = CIN / 8;
    constexpr int NITEM = TIH * TIW * G;
    constexpr int NE = NTHR / G * G;             // staging threads: a multiple of G, so that a thread keeps one channel group
    constexpr int NIT = (NITEM + NE - 1) / NE;
    constexpr int IN_FLOATS = TIH * TIW * CP;
    constexpr int W_FLOATS = COUTP * WS;
    static_assert(NIT <= 32, "inside mask");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + IN_FLOATS;
    double* s_red = (double*)(s_w + W_FLOATS);   // [WM][COUTP][2]

    const int t = threadIdx.x;
    // ---- all weights, once: global [n][KTOT] (k-permuted per group of 8) -> LDS rows of WS floats
    for (int f = t; f < COUTP * (KTOT / 4); f += NTHR) {
        const int n = f / (KTOT / 4), c4 = f % (KTOT / 4);
        *(f32x4*)(s_w + n * WS + c4 * 4) = *(const f32x4*)(a.w + (size_t)f * 4);
    }
    const int g = t % G;                         // this thread's channel group in every item it stages
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int pr = i / WW, pc = i % WW;
    const int ly = (wm * WH + pr) * ST, lx = pc * ST;

    f32x4 v0[NIT], v1[NIT];
    float pl[NIT];                                // PRO_B2IN: pooled image value of the item's pixel
    unsigned inside = 0u;
    auto load_tile = [&](int tile) {              // global -> registers (raw values), remembers which items lie inside the image
        // branch-free: clamped addresses, every load issued; `inside` is what decides later.  (With a branch per item the compiler
        // zero-fills the registers and nests two exec-mask regions around each pair of loads: ~30 instructions per item on the
        // vector pipe the MFMAs of the current tile are waiting for.)
        const int b = tile / ntile, tl = tile - b * ntile;
        const int tx0 = (tl % a.tiles_x) * TW, ty0 = (tl / a.tiles_x) * TH;
        const float* in = a.in + (size_t)b * a.in_stride + (PRO == PRO_UNFOLD ? 0 : g * 8);
        inside = 0u;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int item = t + k * NE;
            const int pix = item / G;
            const int iy = pix / TIW, ix = pix % TIW;
            const int gy = ty0 * ST - PAD + iy, gx = tx0 * ST - PAD + ix;
            if constexpr (KS == 1) {
                const int cy = min(max(gy, 0), a.Hin - 1), cx = min(max(gx, 0), a.Win - 1);
                const float* p = PRO == PRO_UNFOLD ? in + ((cy * 8 + g) * (a.Win * 8) + cx * 8) : in + (cy * a.Win + cx) * CIN;
                v0[k] = *(const f32x4*)p; v1[k] = *(const f32x4*)(p + 4);
                // (a pixel of the tile's overhang keeps the clamped pixel's values: in a 1x1 convolution it feeds nothing but its own output
                // pixel, which the epilogue neither stores nor counts -- zeroing it cost 8 v_cndmask per item on the pipe the MFMAs use)
            } else {                              // 3x3 layers (24 -> 24, 8 -> 24): the branchy form measured 3-5 % faster there
                v0[k] = f32x4{0.f, 0.f, 0.f, 0.f}; v1[k] = v0[k];
                if (t < NE && item < NITEM && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) {
                    const float* p = in + (gy * a.Win + gx) * CIN;
                    v0[k] = *(const f32x4*)p; v1[k] = *(const f32x4*)(p + 4);
                    if constexpr (PRO == PRO_B2IN) pl[k] = a.pool[(size_t)b * a.pool_stride + (size_t)gy * a.Win + gx];
                    inside |= 1u << k;
                }
            }
        }
    };
    auto store_tile = [&](int tile) {             // registers -> activated, k-permuted LDS tile
        f32x4 m0, m1, r0, r1, w0, w1, c0, c1;
        if constexpr (PRO == PRO_BN || PRO == PRO_B2IN) {
            const int b = tile / ntile;
            const float* st = a.st.stat + (size_t)b * 2 * CIN + g * 8;      // large batches: always finalised statistics
            m0 = *(const f32x4*)st; m1 = *(const f32x4*)(st + 4); r0 = *(const f32x4*)(st + CIN); r1 = *(const f32x4*)(st + CIN + 4);
        }
        if constexpr (PRO == PRO_B2IN) {
            w0 = *(const f32x4*)(a.skip_w + g * 8); w1 = *(const f32x4*)(a.skip_w + g * 8 + 4);
            c0 = *(const f32x4*)(a.skip_b + g * 8); c1 = *(const f32x4*)(a.skip_b + g * 8 + 4);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int item = t + k * NE;
            if (t < NE && item < NITEM) {
                f32x4 x0 = v0[k], x1 = v1[k];
                [[maybe_unused]] const bool in_img = inside & (1u << k);
                if constexpr (KS == 1) {
                    float xm = 0.f, xr = 1.f;
                    if constexpr (PRO == PRO_UNFOLD) { const int fb = tile / ntile; xm = a.xstat[fb * 2]; xr = a.xstat[fb * 2 + 1]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if constexpr (PRO == PRO_UNFOLD) { x0[q] = fmaf(x0[q], xr, xm); x1[q] = fmaf(x1[q], xr, xm); }
                        if constexpr (PRO == PRO_BN) {
                            x0[q] = fmaxf(fmaf(x0[q], r0[q], m0[q]), 0.f);
                            x1[q] = fmaxf(fmaf(x1[q], r1[q], m1[q]), 0.f);
                        }
                    }
                } else if constexpr (PRO == PRO_BN || PRO == PRO_B2IN) {
                    if (in_img) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            x0[q] = fmaxf(fmaf(x0[q], r0[q], m0[q]), 0.f);
                            x1[q] = fmaxf(fmaf(x1[q], r1[q], m1[q]), 0.f);
                            if constexpr (PRO == PRO_B2IN) { x0[q] = x0[q] + (pl[k] * w0[q] + c0[q]); x1[q] = x1[q] + (pl[k] * w1[q] + c1[q]); }
                        }
                    }
                }
                float* d = s_in + (item / G) * CP + g * 8;
                *(f32x4*)d = f32x4{x0.x, x0.z, x1.x, x1.z};
                *(f32x4*)(d + 4) = f32x4{x0.y, x0.w, x1.y, x1.w};
            }
        }
    };

    int tile = blockIdx.x;
    if (tile >= total) return;
    float biasv[NT];
    conv_bias<COUT, NT, EPI>(a.bias, wn * NT * 32 + i, biasv);
    load_tile(tile);
    store_tile(tile);
    __syncthreads();                              // weights and the first tile are in LDS
    while (true) {
        const int next = tile + gridDim.x;
        const bool has_next = next < total;
        if (has_next) load_tile(next);            // in flight during the MFMAs below

        f32x16 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            const float* pa = s_in + ((ly + ky) * TIW + lx + kx) * CP + 4 * h;
            const float* pw = s_w + (wn * NT * 32 + i) * WS + tap * CIN + 4 * h;
#pragma unroll
            for (int kk = 0; kk < CIN / 8; ++kk) {
                const f32x4 av = *(const f32x4*)(pa + kk * 8);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f32x4 bv = *(const f32x4*)(pw + n * 32 * WS + kk * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[n], 0, 0, 0);
                }
            }
        }

        // ---- epilogue of `tile` (C/D layout: channel (lane&31) of tile n, pixels (r&3) + 8*(r>>2) + 4*h)
        const int b = tile / ntile, tl = tile - b * ntile;
        const int tx0 = (tl % a.tiles_x) * TW, ty0 = (tl / a.tiles_x) * TH;
        double sum[NT], sq[NT];
        {
            const int oy0 = ty0 + wm * WH;
            float* wave_out = a.out + (size_t)b * a.out_stride + ((size_t)oy0 * a.Wout + tx0) * COUT;
            conv_epilogue<COUT, NT, WW, WH, EPI>(acc, wave_out, a.Wout, a.Hout - oy0, a.Wout - tx0, wn * NT * 32 + i, h, biasv, sum, sq);
        }
        if constexpr (EPI == EPI_STATS) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                sum[n] += __shfl_xor(sum[n], 32);
                sq[n] += __shfl_xor(sq[n], 32);
                if (h == 0) {
                    const int co = (wn * NT + n) * 32 + i;
                    s_red[(wm * COUTP + co) * 2 + 0] = sum[n];
                    s_red[(wm * COUTP + co) * 2 + 1] = sq[n];
                }
            }
        }
        __syncthreads();                          // every wave is done with s_in; s_red is complete
        if constexpr (EPI == EPI_STATS) {
            for (int co = t; co < COUT; co += NTHR) {
                double S = 0.0, SS = 0.0;
#pragma unroll
                for (int m = 0; m < WM; ++m) { S += s_red[(m * COUTP + co) * 2 + 0]; SS += s_red[(m * COUTP + co) * 2 + 1]; }
                double* p = a.part + (size_t)b * a.part_stride + ((size_t)tl * COUT + co) * 2;
                p[0] = S; p[1] = SS;
            }
        }
        if (!has_next) break;
        store_tile(next);
        __syncthreads();
        tile = next;
    }
}

// ------------------------------------------------------------------------------------
// k_conv_mfma_t: persistent form of k_conv_mfma for the 3x3 layers whose weight matrix does NOT fit in LDS (the stride-2 layers and the >= 64-channel
// 3x3 layers at large batches).  k_conv_mfma runs these with one or two workgroups per CU, and a workgroup is a strictly serial chain -- raw tile
// from global memory (a round trip), statistics (another), activation + LDS, the first weight chunk (a third), K loop, epilogue: phase stamps
// of the 64 -> 64 stride-2 layer show 7 of 23 us before the first MFMA, with nothing else resident on the CU to cover them.  Here a workgroup
//   - walks over tiles  blockIdx.x, blockIdx.x + gridDim.x, ...  (frames x tiles) like k_conv_mfma_p,
//   - issues the raw global loads of the NEXT tile (and its frame's statistics) before the K loop of the current one: registers hold them,
//     BatchNorm + ReLU are applied when they move to LDS after the epilogue,
//   - streams the weights as an ENDLESS cycle of chunks through the same two LDS buffers and PD register sets: every tile uses the same
//     weights, so the last chunk iterations of a tile already bring in the first chunks of the next one and no tile starts with a weight wait.
// Same tiles, same MFMAs in the same order, same statistics partials as k_conv_mfma: interchangeable bit for bit (test_batch_is_per_frame).
template <int CIN, int COUT, int KS, int ST, int WM, int WN, int NT, int WW, int PRO, int EPI, int CBMAX = 64>
__global__ __launch_bounds__(64 * WM * WN)
void k_conv_mfma_t(ConvArgs a, int ntile, int total) {
    static_assert(PRO == PRO_BN, "layers behind a finalised BatchNorm (large batches)");
    constexpr int NTHR = 64 * WM * WN;
    constexpr int WH = 32 / WW, TH = WM * WH, TW = WW;
    constexpr int COUTP = WN * NT * 32;
    constexpr int PAD = KS / 2;
    constexpr int TIH = (TH - 1) * ST + KS, TIW = (TW - 1) * ST + KS;
    constexpr int CP = CIN + 4;
    constexpr int CB = CIN > CBMAX ? CBMAX : CIN;
    constexpr int NCB = CIN / CB;
    constexpr int NCHUNK = KS * KS * NCB;
    static_assert(NCHUNK >= 3 && NCHUNK % 3 == 0, "the chunk ring (three LDS buffers, three register sets) closes on itself");
    constexpr int WS = CB + 4;
    constexpr int WCH = COUTP * CB;
    constexpr int NWLD = (WCH / 4 + NTHR - 1) / NTHR;
    constexpr int G = CIN / 8;
    constexpr int NITEM = TIH * TIW * G;
    constexpr int NE = NTHR / G * G;
    constexpr int NIT = (NITEM + NE - 1) / NE;
    constexpr int IN_FLOATS = TIH * TIW * CP;
    constexpr int W_FLOATS = COUTP * WS;
    static_assert(NIT <= 32, "inside mask");
    constexpr int SPC = CB / 8;                                  // k steps of 8 per chunk
    constexpr int GS = (NT == 1 && SPC % 2 == 0) ? 2 : 1;        // steps per operand group: at least 8 MFMAs
    constexpr int GPC = SPC / GS, NGT = NCHUNK * GPC;            // groups per chunk / per tile

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + IN_FLOATS;               // three buffers of W_FLOATS
    double* s_red = (double*)(s_w + 3 * W_FLOATS);   // [WM][COUTP][2]

    const int t = threadIdx.x;
    const int g = t % G;                         // this thread's channel group in every item it stages
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int pr = i / WW, pc = i % WW;
    const int ly = (wm * WH + pr) * ST, lx = pc * ST;

    int tile = blockIdx.x;
    if (tile >= total) return;
    // ---- the weight ring.  Chunk c of the endless sequence c = 0, 1, 2, ... is chunk c % NCHUNK of the layer; it lives in LDS buffer c % 3 while it is
    // multiplied, and before that in register set c % 3.  While chunk c is multiplied, chunk c + 1 is already in LDS (written one chunk earlier, made
    // visible by the barrier in between), chunk c + 2 moves from its registers to the buffer chunk c - 1 has just left, and that register set is refilled
    // with chunk c + 5.  ONE barrier per chunk, at its start, and no wait behind it: the operands of a chunk's first group are read BEFORE the barrier.
    f32x4 wreg[3][NWLD];
    const __amdgpu_buffer_rsrc_t wrs = wbuf_make(a.w);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < NWLD; ++q) {
            const int f = t + q * NTHR;
            if (f < WCH / 4) wreg[c][q] = wbuf_ld(wrs, f, c * WCH);
        }

    f32x4 v0[NIT], v1[NIT], m0, m1, r0, r1;
    unsigned inside = 0u;
    auto load_tile = [&](int tl_) {               // global -> registers (raw values + the frame's statistics), clamped addresses, no branches
        const int b = tl_ / ntile, tl = tl_ - b * ntile;
        const int tx0 = (tl % a.tiles_x) * TW, ty0 = (tl / a.tiles_x) * TH;
        const float* in = a.in + (size_t)b * a.in_stride + g * 8;
        inside = 0u;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int item = t + k * NE;
            const int pix = (NITEM % NE != 0 && k == NIT - 1) ? min(item, NITEM - 1) / G : item / G;
            const int gy = ty0 * ST - PAD + pix / TIW, gx = tx0 * ST - PAD + pix % TIW;
            const int cy = min(max(gy, 0), a.Hin - 1), cx = min(max(gx, 0), a.Win - 1);
            const float* p = in + ((size_t)cy * a.Win + cx) * CIN;
            v0[k] = *(const f32x4*)p; v1[k] = *(const f32x4*)(p + 4);
            inside |= (gy == cy && gx == cx) ? (1u << k) : 0u;
        }
        const float* st = a.st.stat + (size_t)b * 2 * CIN + g * 8;          // large batches: always finalised statistics
        m0 = *(const f32x4*)st; m1 = *(const f32x4*)(st + 4); r0 = *(const f32x4*)(st + CIN); r1 = *(const f32x4*)(st + CIN + 4);
    };
    auto store_tile = [&]() {                     // registers -> activated, k-permuted LDS tile (zero padding outside the image)
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int item = t + k * NE;
            if (t < NE && item < NITEM) {
                f32x4 x0 = v0[k], x1 = v1[k];
                const bool in_img = inside & (1u << k);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x0[q] = fmaxf(fmaf(x0[q], r0[q], m0[q]), 0.f);
                    x1[q] = fmaxf(fmaf(x1[q], r1[q], m1[q]), 0.f);
                    x0[q] = in_img ? x0[q] : 0.f; x1[q] = in_img ? x1[q] : 0.f;
                }
                float* d = s_in + (item / G) * CP + g * 8;
                *(f32x4*)d = f32x4{x0.x, x0.z, x1.x, x1.z};
                *(f32x4*)(d + 4) = f32x4{x0.y, x0.w, x1.y, x1.w};
            }
        }
    };
    // ring position rc (compile time, any value with the right residues): its register set -> its LDS buffer; the set is refilled with position rc + 3
    auto push_chunk = [&](int rc) {
        float* wd = s_w + (rc % 3) * W_FLOATS;
        const int wsrc_off = ((rc + 3) % NCHUNK) * WCH;
#pragma unroll
        for (int q = 0; q < NWLD; ++q) {
            const int f = t + q * NTHR;
            if (f < WCH / 4) {
                const int n = f / (CB / 4), c4 = f % (CB / 4);
                *(f32x4*)(wd + n * WS + c4 * 4) = wreg[rc % 3][q];
                wreg[rc % 3][q] = wbuf_ld(wrs, f, wsrc_off);
            }
        }
    };

    float biasv[NT];
    conv_bias<COUT, NT, EPI>(a.bias, wn * NT * 32 + i, biasv);
    load_tile(tile);
    store_tile();
    push_chunk(0);
    push_chunk(1);
    __syncthreads();                              // the first tile and the first two weight chunks are in LDS

    f32x4 av[2][GS], bv[2][GS][NT];
    auto load_group = [&](int q, int buf) {       // operands of group q of the tile (chunk q / GPC): A from the input tile, B from the chunk's buffer
#pragma unroll
        for (int u = 0; u < GS; ++u) {
            const int ch = q / GPC, kk = (q % GPC) * GS + u;
            const int tap = ch / NCB, cb = ch % NCB;
            const int ky = tap / KS, kx = tap % KS;
            av[buf][u] = *(const f32x4*)(s_in + ((ly + ky) * TIW + lx + kx) * CP + cb * CB + 4 * h + kk * 8);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                bv[buf][u][n] = *(const f32x4*)(s_w + (ch % 3) * W_FLOATS + (wn * NT * 32 + i + n * 32) * WS + 4 * h + kk * 8);
        }
    };
    while (true) {
        const int next = tile + gridDim.x;
        const bool has_next = next < total;
        f32x16 acc[NT];
        auto mfma_part = [&](int buf, int part, bool first) {      // half of a group's MFMAs, k ascending per accumulator
            if constexpr (GS == 2) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][part][j], bv[buf][part][n][j], acc[n], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 2 * part; j < 2 * part + 2; ++j)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][0][j], bv[buf][0][n][j], acc[n], 0, 0, 0);
            }
        };
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        load_group(0, 0);                         // (the barrier in front of this tile made the tile and chunks 0, 1 visible)
        push_chunk(2);                            // into the buffer the previous tile's last chunk has left
        if (has_next) load_tile(next);            // in flight during the K loop below
#pragma unroll
        for (int q = 0; q < NGT; ++q) {
            const int cur = q & 1;
            if (q > 0 && q % GPC == 0) {          // a chunk begins: every wave has left the previous chunk -> its buffer takes the chunk after next
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
                push_chunk(q / GPC + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_part(cur, 0, q == 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < NGT) load_group(q + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_part(cur, 1, false);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue of `tile` (C/D layout: channel (lane&31) of tile n, pixels (r&3) + 8*(r>>2) + 4*h)
        const int b = tile / ntile, tl = tile - b * ntile;
        const int tx0 = (tl % a.tiles_x) * TW, ty0 = (tl / a.tiles_x) * TH;
        double sum[NT], sq[NT];
        {
            const int oy0 = ty0 + wm * WH;
            float* wave_out = a.out + (size_t)b * a.out_stride + ((size_t)oy0 * a.Wout + tx0) * COUT;
            conv_epilogue<COUT, NT, WW, WH, EPI>(acc, wave_out, a.Wout, a.Hout - oy0, a.Wout - tx0, wn * NT * 32 + i, h, biasv, sum, sq);
        }
        if constexpr (EPI == EPI_STATS) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                sum[n] += __shfl_xor(sum[n], 32);
                sq[n] += __shfl_xor(sq[n], 32);
                if (h == 0) {
                    const int co = (wn * NT + n) * 32 + i;
                    s_red[(wm * COUTP + co) * 2 + 0] = sum[n];
                    s_red[(wm * COUTP + co) * 2 + 1] = sq[n];
                }
            }
        }
        __syncthreads();                          // every wave is done with s_in and with the tile's last chunk; s_red is complete
        if constexpr (EPI == EPI_STATS) {
            for (int co = t; co < COUT; co += NTHR) {
                double S = 0.0, SS = 0.0;
#pragma unroll
                for (int m = 0; m < WM; ++m) { S += s_red[(m * COUTP + co) * 2 + 0]; SS += s_red[(m * COUTP + co) * 2 + 1]; }
                double* p = a.part + (size_t)b * a.part_stride + ((size_t)tl * COUT + co) * 2;
                p[0] = S; p[1] = SS;
            }
        }
        if (!has_next) break;
        store_tile();
        __syncthreads();                          // the next tile is in LDS (and s_red has been read)
        tile = next;
    }
}

// ------------------------------------------------------------------------------------
// k_conv_mfma16: the SINGLE-FRAME form of the 3x3 stride-1 layers with >= 64 channels, on v_mfma_f32_16x16x4_f32.
// One frame gives a layer 40-150 workgroups; with 32x32x2 tiles a wave then walks an accumulation chain of 288-576 dependent
// MFMAs of 64 cycles each (7.7-15 us of pure chain per layer) on a mostly idle GPU.  The 16x16x4 instruction is the same exact
// fp32 fma chain in k order (tools/probes/mfma16_probe.hip: 0 mismatches in 51 200 outputs) with a dependent latency of 20 ns per
// four k: a wave owns 16 pixels x 16 channels, four times as many waves share the work and the chain shrinks 2.6x -- same bits.
//   lane (p = l & 15, q = l >> 4): A = pixel p of the wave's pixel group, B = channel p of its channel group, k = k0 + q;
//   LDS rows (pixels / weight rows) keep their channels permuted inside each group of 16 (channel 4j + q at position 4q + j), so one
//   ds_read_b128 per operand feeds four consecutive MFMAs; D: register r = pixel 4q + r, lane column p = channel.
// Workgroup = PGY pixel groups (GW x 16/GW pixels each, stacked vertically) x COUT/16 channel groups; staging, weight streaming
// (one tap x 64 channels per chunk through two LDS buffers, PD chunks in flight in registers) and statistics partials as in k_conv_mfma.
template <int CIN, int COUT, int ST, int GW, int PGY, int PRO, int EPI, int PD = 3, int CGS = 1>
__global__ __launch_bounds__(64 * PGY * (COUT / 16 / CGS))
void k_conv_mfma16(ConvArgs a) {
    static_assert(PRO == PRO_BN || PRO == PRO_FUSE, "3x3 layers behind a BatchNorm (block_fusion.0: + the pyramid sum)");
    // CGS: the output channels are split over CGS workgroups (blockIdx.y) of CG = COUT / 16 / CGS channel groups each -- same tiles, same
    // partials, more SIMDs: a single frame's 20-150 tiles x 8 waves put two dependent MFMA chains on every SIMD they touch (the f32
    // MFMA pipe runs one at a time: 38 ns per step instead of 17) while most of the 1024 SIMDs have nothing to do
    constexpr int GH = 16 / GW, CG = COUT / 16 / CGS, COUTW = CG * 16, NW = PGY * CG, NTHR = 64 * NW;
    static_assert(COUT % (16 * CGS) == 0, "channel split");
    constexpr int TH = GH * PGY, TW = GW, TIH = (TH - 1) * ST + 3, TIW = (TW - 1) * ST + 3, CP = CIN + 4;
    constexpr int CB = 64, NCB = CIN / CB, NCHUNK = 9 * NCB, KC = CB, WS = KC + 4;
    constexpr int WCH = COUT * KC /* a chunk in memory */, WCW = COUTW * KC /* this workgroup's rows of it */, NWLD = (WCW / 4 + NTHR - 1) / NTHR;
    constexpr int G = CIN / 8, NITEM = TIH * TIW * G, NIT = (NITEM + NTHR - 1) / NTHR;
    constexpr int IN_FLOATS = TIH * TIW * CP, W_FLOATS = COUTW * WS;
    static_assert(sizeof(double) * 512 <= sizeof(float) * IN_FLOATS, "bn_fold scratch in the input tile");
    static_assert(sizeof(double) * PGY * COUTW * 2 <= sizeof(float) * IN_FLOATS, "statistics scratch in the input tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + IN_FLOATS;               // three buffers of W_FLOATS
    float* s_stat = s_w + 3 * W_FLOATS;          // 2 * CIN (PRO_FUSE: 3 * 128)

    constexpr int FB = NTHR > 512 ? 8 : 16;      // fold batch (bn_fold): the 16-wave form has 128 registers per thread
    const int t = threadIdx.x, b = blockIdx.z;
    const int tile = blockIdx.x, tx0 = (tile % a.tiles_x) * TW, ty0 = (tile / a.tiles_x) * TH;
    const float* in = a.in + (size_t)b * a.in_stride;
    const int co0 = blockIdx.y * COUTW;                    // first output channel of this workgroup
    const float* wg = a.w + (size_t)co0 * KC;              // its rows of chunk 0
    const __amdgpu_buffer_rsrc_t wrs = wbuf_make(wg);

    XFH_STAMP(a, 0);
    // weight chunks in flight: PD of them, in a ring of register sets (chunk c in set c % PD).  With one chunk ahead the K loop of a
    // single frame ran at one memory round trip per chunk (0.7 us against 0.3 us of MFMA chain): 18 of them in the 128-channel layers.
    f32x4 wreg[PD][NWLD];
#pragma unroll
    for (int c = 0; c < PD; ++c)
#pragma unroll
        for (int q = 0; q < NWLD; ++q) {
            const int f = t + q * NTHR;
            if (c < NCHUNK && f < WCW / 4) wreg[c][q] = wbuf_ld(wrs, f, c * WCH);
        }
    f32x4 r0[NIT], r1[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int item = t + k * NTHR, pix = item / G, g = item % G;
        const int gy = min(max(ty0 * ST - 1 + pix / TIW, 0), a.Hin - 1), gx = min(max(tx0 * ST - 1 + pix % TIW, 0), a.Win - 1);
        const float* p = in + ((size_t)gy * a.Win + gx) * CIN + g * 8;
        r0[k] = *(const f32x4*)p;
        r1[k] = *(const f32x4*)(p + 4);
    }
    XFH_STAMP(a, 1);
    stage_stat<FB>(a.st, b, CIN, tile == 0, s_stat, (double*)s_in, t, NTHR);
    if constexpr (PRO == PRO_FUSE) {
        stage_stat<FB>(a.st4, b, 64, tile == 0, s_stat + 128, (double*)s_in, t, NTHR);
        stage_stat<FB>(a.st5, b, 64, tile == 0, s_stat + 256, (double*)s_in, t, NTHR);
    }
    XFH_STAMP(a, 2);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int item = t + k * NTHR, pix = item / G, g = item % G;
        const int gy = ty0 * ST - 1 + pix / TIW, gx = tx0 * ST - 1 + pix % TIW;
        const bool ok = gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        if (NITEM % NTHR != 0 && item >= NITEM) continue;
        f32x4 v0 = r0[k], v1 = r1[k];
        const f32x4 m0 = *(const f32x4*)(s_stat + g * 8), m1 = *(const f32x4*)(s_stat + g * 8 + 4);
        const f32x4 q0 = *(const f32x4*)(s_stat + CIN + g * 8), q1 = *(const f32x4*)(s_stat + CIN + g * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v0[e] = fmaxf(fmaf(v0[e], q0[e], m0[e]), 0.f);
            v1[e] = fmaxf(fmaf(v1[e], q1[e], m1[e]), 0.f);
        }
        if constexpr (PRO == PRO_FUSE) {                // x3 + up2(x4) + up4(x5) with ATen's bilinear arithmetic (XFeat.cc:159-166)
            // no branch around the sixteen tap loads of an item (clamped coordinates; what lies outside the image is zeroed below): behind
            // `if (ok)` the items of a thread fetched their taps one memory round trip after the other
            const int cy = min(max(gy, 0), a.Hin - 1), cx = min(max(gx, 0), a.Win - 1);
            f32x4 u0, u1, w0, w1;
            up_bilinear8(a.r4 + (size_t)b * a.s4, s_stat + 128, a.H4, a.W4, a.Hin, a.Win, cy, cx, g, u0, u1);
            // the 16-wave form has 128 registers per thread: with the taps of both maps in flight at once (64 registers) it spilled nine of them;
            // there the second map's taps are issued after the first map is combined (one more round trip on 2-32-frame batches with > 256 tiles)
            if constexpr (NTHR > 512) __builtin_amdgcn_sched_barrier(0);
            up_bilinear8(a.r5 + (size_t)b * a.s5, s_stat + 256, a.H5, a.W5, a.Hin, a.Win, cy, cx, g, w0, w1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = (v0[e] + u0[e]) + w0[e]; v1[e] = (v1[e] + u1[e]) + w1[e]; }
        }
        if (!ok) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
        // channels 8g + e and 8g + 4 + e sit next to each other in the group-of-16 permutation: position 4e + 2(g & 1) (+ 1)
        float* d = s_in + pix * CP + (g >> 1) * 16 + 2 * (g & 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) *(f32x2*)(d + 4 * e) = f32x2{v0[e], v1[e]};
    }
    static_assert(PD >= 2 && NCHUNK >= 3, "two chunks go to LDS before the loop");
#pragma unroll
    for (int c = 0; c < 2; ++c)                                 // chunks 0 and 1 -> LDS buffers 0 and 1; their register sets take chunks PD and PD + 1
#pragma unroll
        for (int q = 0; q < NWLD; ++q) {
            const int f = t + q * NTHR;
            if (f < WCW / 4) {
                const int n = f / (KC / 4), c4 = f % (KC / 4);
                *(f32x4*)(s_w + c * W_FLOATS + n * WS + c4 * 4) = wreg[c][q];
                if (c + PD < NCHUNK) wreg[c][q] = wbuf_ld(wrs, f, (c + PD) * WCH);
            }
        }
    __syncthreads();
    XFH_STAMP(a, 3);

    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, p = lane & 15, q = lane >> 4;
    const int pg = wave / CG, cg = wave % CG;
    const int ly = pg * GH + p / GW, lx = p % GW;               // this lane's output pixel inside the tile (A operand row)
    float biasv = 0.f;
    if constexpr (EPI != EPI_STATS) biasv = a.bias[co0 + cg * 16 + p];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // K loop, software pipelined over the chunks: while the 16 MFMAs of chunk ch run (one dependent chain: the k order is the numerics
    // contract), the operands of chunk ch + 1 are read from LDS (its weights were written one iteration earlier, before the last
    // barrier) and the weights of chunk ch + 2 go from their registers to the third buffer.  With two buffers the chain stopped at
    // every barrier for the write -> barrier -> read round trip: 0.6 us per chunk against 0.32 us of MFMA.
    auto read_ops = [&](int ch, f32x4 (&av)[CB / 16], f32x4 (&bv)[CB / 16]) {
        const int tap = ch / NCB, cb = ch % NCB, ky = tap / 3, kx = tap % 3;
        const float* pa = s_in + ((ly * ST + ky) * TIW + lx * ST + kx) * CP + cb * CB + 4 * q;
        const float* pw = s_w + (ch % 3) * W_FLOATS + (cg * 16 + p) * WS + 4 * q;
#pragma unroll
        for (int kb = 0; kb < CB / 16; ++kb) { av[kb] = *(const f32x4*)(pa + kb * 16); bv[kb] = *(const f32x4*)(pw + kb * 16); }
    };
    f32x4 av[CB / 16], bv[CB / 16], avn[CB / 16], bvn[CB / 16];
    read_ops(0, av, bv);
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ++ch) {
        if (ch + 1 < NCHUNK) read_ops(ch + 1, avn, bvn);
#pragma unroll
        for (int kb = 0; kb < CB / 16; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][j], bv[kb][j], acc, 0, 0, 0);
        if (ch + 2 < NCHUNK) {                              // chunk ch + 2 (issued PD chunks ago) -> the buffer chunk ch - 1 was read from; its register set takes chunk ch + 2 + PD
            float* wd = s_w + ((ch + 2) % 3) * W_FLOATS;
            const int wsrc_off = (ch + 2 + PD) * WCH;
#pragma unroll
            for (int u = 0; u < NWLD; ++u) {
                const int f = t + u * NTHR;
                if (f < WCW / 4) {
                    const int n = f / (KC / 4), c4 = f % (KC / 4);
                    *(f32x4*)(wd + n * WS + c4 * 4) = wreg[(ch + 2) % PD][u];
                    if (ch + 2 + PD < NCHUNK) wreg[(ch + 2) % PD][u] = wbuf_ld(wrs, f, wsrc_off);
                }
            }
        }
        if (ch + 1 < NCHUNK) {
            __syncthreads();
#pragma unroll
            for (int kb = 0; kb < CB / 16; ++kb) { av[kb] = avn[kb]; bv[kb] = bvn[kb]; }
        }
    }
    __syncthreads();
    XFH_MFMA_SETTLE();
    XFH_STAMP(a, 4);
    // ---- epilogue: register r = pixel 4q + r of the group, lane column p = channel cg*16 + p
    double sum = 0.0, sq = 0.0;
    float* out = a.out + (size_t)b * a.out_stride + co0 + cg * 16 + p;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pi = 4 * q + r, oy = ty0 + pg * GH + pi / GW, ox = tx0 + pi % GW;
        if (oy < a.Hout && ox < a.Wout) {
            float v = acc[r];
            if constexpr (EPI != EPI_STATS) v += biasv;
            if constexpr (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
            out[((size_t)oy * a.Wout + ox) * COUT] = v;
            if constexpr (EPI == EPI_STATS) { const double dv = (double)v; sum += dv; sq = fma(dv, dv, sq); }
        }
    }
    if constexpr (EPI == EPI_STATS) {
        sum += __shfl_xor(sum, 16); sq += __shfl_xor(sq, 16);
        sum += __shfl_xor(sum, 32); sq += __shfl_xor(sq, 32);
        double* s_red = (double*)smem;                 // the tiles are no longer needed (all waves passed the last barrier)
        if (q == 0) { s_red[(pg * COUTW + cg * 16 + p) * 2] = sum; s_red[(pg * COUTW + cg * 16 + p) * 2 + 1] = sq; }
        __syncthreads();
        for (int co = t; co < COUTW; co += NTHR) {
            double S = 0.0, SS = 0.0;
#pragma unroll
            for (int m = 0; m < PGY; ++m) { S += s_red[(m * COUTW + co) * 2]; SS += s_red[(m * COUTW + co) * 2 + 1]; }
            double* pp = a.part + (size_t)b * a.part_stride + ((size_t)tile * COUT + co0 + co) * 2;
            pp[0] = S; pp[1] = SS;
        }
    }
    XFH_STAMP(a, 5);
}

// ------------------------------------------------------------------------------------
// k_chain1x1: consecutive 1x1 convolutions 64 -> 64 at 1/8 resolution in ONE pass over the pixels, wherever nothing but a bias
// (and a ReLU) separates them.  In use: block_fusion.2 -> heatmap_head.0 (fusion.2 is a bare Conv2d, XFeat.cc:75) in
// every BatchNorm mode and batch size, and block_fusion.2 -> heatmap_head.0 -> heatmap_head.1 (bias+ReLU hand-over) for batches <= 8 with
// folded BatchNorms, where a launch is mostly fixed cost; the three-stage chains measured slower at 256 frames (launch_fusion_chain).  Persistent like k_conv_mfma_p (same tiles, same K order, hence
// the same bits per layer); the output of a stage goes from the MFMA's C/D layout (lane = channel, registers = pixels) straight
// back into the wave's own 32 pixel rows of the LDS tile in the A-operand layout (k-permuted channels) -- a 1x1 convolution needs
// no other wave's pixels, so there is no barrier between the stages and the handed-on map is not read back from HBM.
enum { MID_BIAS_STORE = 0 /* y = acc + bias, stored (fusion.2 -> feats) and handed on */, MID_BIAS_RELU = 1 /* folded BatchNorm + ReLU, handed on only */ };
struct ChainArgs {
    ConvArgs a;                               // input / prologue / geometry; a.w, a.bias: stage 0; a.out, a.part: the LAST stage
    const float* w1; const float* bias1;      // stage 1
    const float* w2; const float* bias2;      // stage 2 (NL == 3)
    float* mid_out; size_t mid_stride;        // MID_BIAS_STORE: where stage 0's map is stored
};
// SSTAT (batches <= 8): the producer's statistics are staged in LDS per frame -- folded from its partials by this workgroup when no
// k_bn_finalize runs (stage_stat) -- instead of being read from the finalized slots.
template <int NL, int WM, int PRO, int MID0, int EPI, bool SSTAT = false>
__global__ __launch_bounds__(64 * WM)
void k_chain1x1(ChainArgs ca, int ntile, int total) {
    constexpr int CIN = 64, COUT = 64, NT = 2, WW = 16, WH = 2, TH = WM * WH, TW = WW;
    constexpr int NTHR = 64 * WM, COUTP = 64, CP = CIN + 4, WS = CIN + 4, G = CIN / 8;
    constexpr int NITEM = TH * TW * G, NIT = (NITEM + NTHR - 1) / NTHR;
    constexpr int IN_FLOATS = TH * TW * CP, W_FLOATS = COUTP * WS;
    static_assert(NTHR % G == 0 && NIT <= 32, "staging");
    const ConvArgs& a = ca.a;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + IN_FLOATS;               // NL weight matrices
    double* s_red = (double*)(s_w + NL * W_FLOATS);
    float* s_stat = (float*)(s_red + WM * COUTP * 2);      // SSTAT: mean[64], rstd[64] of the current frame
    static_assert(!SSTAT || sizeof(double) * 512 <= sizeof(float) * IN_FLOATS, "bn_fold scratch in the input tile");
    int stat_b = -1;

    const int t = threadIdx.x;
    {
        const float* ws[3] = {a.w, ca.w1, ca.w2};
#pragma unroll
        for (int l = 0; l < NL; ++l)
            for (int f = t; f < COUTP * (CIN / 4); f += NTHR) {
                const int n = f / (CIN / 4), c4 = f % (CIN / 4);
                *(f32x4*)(s_w + l * W_FLOATS + n * WS + c4 * 4) = *(const f32x4*)(ws[l] + (size_t)f * 4);
            }
    }
    const int g = t % G;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, i = lane & 31, h = lane >> 5;
    const int wm = wave;

    f32x4 v0[NIT], v1[NIT];
    auto load_tile = [&](int tile) {
        const int b = tile / ntile, tl = tile - b * ntile;
        const int tx0 = (tl % a.tiles_x) * TW, ty0 = (tl / a.tiles_x) * TH;
        const float* in = a.in + (size_t)b * a.in_stride + g * 8;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int pix = (t + k * NTHR) / G;
            const int gy = ty0 + pix / TW, gx = tx0 + pix % TW;
            const int cy = min(gy, a.Hin - 1), cx = min(gx, a.Win - 1);
            const float* p = in + (cy * a.Win + cx) * CIN;
            v0[k] = *(const f32x4*)p; v1[k] = *(const f32x4*)(p + 4);       // (overhang pixels keep the clamped pixel's values: k_conv_mfma_p)
        }
    };
    auto store_tile = [&](int tile) {
        f32x4 m0, m1, r0, r1;
        if constexpr (PRO == PRO_BN) {
            if constexpr (SSTAT) {                          // (s_in is free here: the previous tile's last reader is behind a barrier)
                const int b = tile / ntile;
                if (b != stat_b) { stage_stat<16>(a.st, b, CIN, tile - b * ntile == 0, s_stat, (double*)s_in, t, NTHR); stat_b = b; }
            }
            const float* st = SSTAT ? s_stat + g * 8 : a.st.stat + (size_t)(tile / ntile) * 2 * CIN + g * 8;
            m0 = *(const f32x4*)st; m1 = *(const f32x4*)(st + 4); r0 = *(const f32x4*)(st + CIN); r1 = *(const f32x4*)(st + CIN + 4);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int item = t + k * NTHR;
            if (NITEM % NTHR != 0 && item >= NITEM) continue;
            f32x4 x0 = v0[k], x1 = v1[k];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (PRO == PRO_BN) {
                    x0[q] = fmaxf(fmaf(x0[q], r0[q], m0[q]), 0.f);
                    x1[q] = fmaxf(fmaf(x1[q], r1[q], m1[q]), 0.f);
                }
            }
            float* d = s_in + (item / G) * CP + g * 8;
            *(f32x4*)d = f32x4{x0.x, x0.z, x1.x, x1.z};
            *(f32x4*)(d + 4) = f32x4{x0.y, x0.w, x1.y, x1.w};
        }
    };

    int tile = blockIdx.x;
    if (tile >= total) return;
    float bias0[NT], bias1[NT], biasl[NT];
    conv_bias<COUT, NT, EPI_BIAS>(a.bias, i, bias0);
    if constexpr (NL == 3) conv_bias<COUT, NT, EPI_BIAS>(ca.bias1, i, bias1);
    conv_bias<COUT, NT, EPI>(NL == 3 ? ca.bias2 : ca.bias1, i, biasl);
    load_tile(tile);
    store_tile(tile);
    __syncthreads();
    const float* pa = s_in + (wm * 32 + i) * CP + 4 * h;         // this lane's pixel row as the A operand
    float* const mid_row = s_in + (wm * 32 + 4 * h) * CP;        // C/D layout: register r holds pixel (r&3) + 8*(r>>2) + 4*h of the wave's block
    while (true) {
        const int next = tile + gridDim.x;
        const bool has_next = next < total;
        if (has_next) load_tile(next);
        const int b = tile / ntile, tl = tile - b * ntile;
        const int tx0 = (tl % a.tiles_x) * TW, ty0 = (tl / a.tiles_x) * TH, oy0 = ty0 + wm * WH;

        f32x16 acc[NT];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
            const float* pw = s_w + l * W_FLOATS + i * WS + 4 * h;
#pragma unroll
            for (int kk = 0; kk < CIN / 8; ++kk) {
                const f32x4 av = *(const f32x4*)(pa + kk * 8);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f32x4 bv = *(const f32x4*)(pw + n * 32 * WS + kk * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[n], 0, 0, 0);
                }
            }
            if (l == NL - 1) break;
            // ---- hand the stage's output on as the next stage's A operand (this wave's own pixel rows)
            const bool store_mid = l == 0 && MID0 == MID_BIAS_STORE;
            if (store_mid) {
                double dsum[NT], dsq[NT];
                float* wave_out = ca.mid_out + (size_t)b * ca.mid_stride + ((size_t)oy0 * a.Wout + tx0) * COUT;
                conv_epilogue<COUT, NT, WW, WH, EPI_BIAS>(acc, wave_out, a.Wout, a.Hout - oy0, a.Wout - tx0, i, h, bias0, dsum, dsq);
            } else {
                XFH_MFMA_SETTLE();
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int c = n * 32 + i;
                const int pos = (c & ~7) + 4 * (c & 1) + ((c & 7) >> 1);      // k permutation inside each group of 8
                const float bv = l == 0 ? bias0[n] : bias1[n];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[n][r] + bv;
                    if (!store_mid) v = fmaxf(v, 0.f);
                    mid_row[((r & 3) + 8 * (r >> 2)) * CP + pos] = v;
                }
            }
        }

        // ---- epilogue of the last stage (as k_conv_mfma_p)
        double sum[NT], sq[NT];
        {
            float* wave_out = a.out + (size_t)b * a.out_stride + ((size_t)oy0 * a.Wout + tx0) * COUT;
            conv_epilogue<COUT, NT, WW, WH, EPI>(acc, wave_out, a.Wout, a.Hout - oy0, a.Wout - tx0, i, h, biasl, sum, sq);
        }
        if constexpr (EPI == EPI_STATS) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                sum[n] += __shfl_xor(sum[n], 32);
                sq[n] += __shfl_xor(sq[n], 32);
                if (h == 0) {
                    const int co = n * 32 + i;
                    s_red[(wm * COUTP + co) * 2 + 0] = sum[n];
                    s_red[(wm * COUTP + co) * 2 + 1] = sq[n];
                }
            }
        }
        __syncthreads();                          // every wave is done with s_in; s_red is complete
        if constexpr (EPI == EPI_STATS) {
            for (int co = t; co < COUT; co += NTHR) {
                double S = 0.0, SS = 0.0;
#pragma unroll
                for (int m = 0; m < WM; ++m) { S += s_red[(m * COUTP + co) * 2 + 0]; SS += s_red[(m * COUTP + co) * 2 + 1]; }
                double* p = a.part + (size_t)b * a.part_stride + ((size_t)tl * COUT + co) * 2;
                p[0] = S; p[1] = SS;
            }
        }
        if (!has_next) break;
        store_tile(next);
        __syncthreads();
        tile = next;
    }
}

// ------------------------------------------------------------------------------------
// host side: layer -> template instance
template <int CIN, int COUT, int KS, int ST, int WM, int WN, int NT, int WW, int PRO, int EPI, int CBMAX = 64, int TPC = 1>
static hipError_t conv_mfma_launch(xfh_ctx* c, const ConvArgs& a, int B, int* npart_out, int layer) {
    constexpr int WH = 32 / WW, TH = WM * WH, TW = WW;
    constexpr int COUTP = WN * NT * 32;
    constexpr int TIH = (TH - 1) * ST + KS, TIW = (TW - 1) * ST + KS;
    constexpr int CB = CIN > CBMAX ? CBMAX : CIN;
    constexpr int NWBUF = (KS * KS * (CIN / CB) / TPC) > 1 ? 2 : 1;
    constexpr int STATF = (PRO == PRO_FUSE || PRO == PRO_FUSEA) ? 384 : (PRO == PRO_B2IN ? 4 * CIN : 2 * CIN);
    constexpr size_t LDS = sizeof(float) * ((size_t)TIH * TIW * (CIN + 4) + NWBUF * (size_t)COUTP * (TPC * CB + 4) + STATF);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert(sizeof(double) * WM * COUTP * 2 <= LDS, "stat scratch");
    static_assert(sizeof(double) * 512 <= sizeof(float) * (size_t)TIH * TIW * (CIN + 4), "bn_fold scratch in the input tile");
    ConvArgs aa = a;
    aa.dbg = layer;
    aa.tiles_x = (a.Wout + TW - 1) / TW;
    const int tiles_y = (a.Hout + TH - 1) / TH;
    const int ntile = aa.tiles_x * tiles_y;
    if (npart_out) *npart_out = ntile;
    auto kern = k_conv_mfma<CIN, COUT, KS, ST, WM, WN, NT, WW, PRO, EPI, CBMAX, TPC>;
    XFH_SET_LDS_ATTR_ONCE(c, kern, LDS);
    launch_k(c, XFH_K_CONV_MFMA, layer, kern, dim3(ntile, 1, B), dim3(64 * WM * WN), LDS, aa);
    return hipGetLastError();
}

#ifndef XFH_CONV_T
#define XFH_CONV_T 1       // bit mask (A/B builds): which layers take the persistent streamed-weights form k_conv_mfma_t at batches > 32 -- 1: block4.0 (64 -> 64 s2),
                           // 2: block3.0 / block5.0 (24 -> 64 s2, 64 -> 128 s2), 4: block4.1/2, 8: block5.1/2, 16: block3.1 / block_fusion.1; 0: k_conv_mfma everywhere.
                           // Measured per 256-frame launch on one box (tools/ab_serial.sh): block4.0 317 -> 274 us (113 KB of LDS: ONE 4-wave workgroup per CU, nothing
                           // else covers its serial phases); every other layer runs 2 workgroups per CU, which already overlap each other's phases, and gets SLOWER
                           // (24 -> 64 s2 443 -> 469, 64 -> 128 s2 149 -> 158, 64 -> 64 at 1/16 205 -> 209, 128 -> 128 232 -> 266, dominant 64 -> 64 790 -> 810)
#endif
template <int CIN, int COUT, int KS, int ST, int WM, int WN, int NT, int WW, int PRO, int EPI, int CBMAX = 64>
static hipError_t conv_mfma_t_launch(xfh_ctx* c, const ConvArgs& a, int B, int* npart_out, int layer) {
    constexpr int WH = 32 / WW, TH = WM * WH, TW = WW;
    constexpr int COUTP = WN * NT * 32;
    constexpr int TIH = (TH - 1) * ST + KS, TIW = (TW - 1) * ST + KS;
    constexpr int CB = CIN > CBMAX ? CBMAX : CIN;
    constexpr size_t LDS = sizeof(float) * ((size_t)TIH * TIW * (CIN + 4) + 3 * (size_t)COUTP * (CB + 4)) + sizeof(double) * WM * COUTP * 2;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    ConvArgs aa = a;
    aa.dbg = layer;
    aa.tiles_x = (a.Wout + TW - 1) / TW;
    const int ntile = aa.tiles_x * ((a.Hout + TH - 1) / TH);
    if (npart_out) *npart_out = ntile;
    auto kern = k_conv_mfma_t<CIN, COUT, KS, ST, WM, WN, NT, WW, PRO, EPI, CBMAX>;
    XFH_SET_LDS_ATTR_ONCE(c, kern, LDS);
    const int total = ntile * B;
    const int per_cu = (int)((160 * 1024) / LDS) > 2 ? 2 : (int)((160 * 1024) / LDS);
    const int grid = total < c->num_cu * per_cu ? total : c->num_cu * per_cu;
    launch_k(c, XFH_K_CONV_MFMA, layer, kern, dim3(grid), dim3(64 * WM * WN), LDS, aa, ntile, total);
    return hipGetLastError();
}

template <int CIN, int COUT, int ST, int GW, int PGY, int PRO, int EPI, int CGS = 1, int PD = 3>
static hipError_t conv_mfma16_launch(xfh_ctx* c, const ConvArgs& a, int B, int* npart_out, int layer) {
    constexpr int GH = 16 / GW, TH = GH * PGY, TW = GW, NW = PGY * (COUT / 16 / CGS);
    constexpr size_t LDS = sizeof(float) * ((size_t)((TH - 1) * ST + 3) * ((TW - 1) * ST + 3) * (CIN + 4) + 3 * (size_t)(COUT / CGS) * 68 + (PRO == PRO_FUSE ? 384 : 2 * CIN));
    static_assert(LDS <= 160 * 1024, "LDS budget");
    ConvArgs aa = a;
    aa.dbg = layer;
    aa.tiles_x = (a.Wout + TW - 1) / TW;
    const int ntile = aa.tiles_x * ((a.Hout + TH - 1) / TH);
    if (npart_out) *npart_out = ntile;
    auto kern = k_conv_mfma16<CIN, COUT, ST, GW, PGY, PRO, EPI, PD, CGS>;
    XFH_SET_LDS_ATTR_ONCE(c, kern, LDS);
    launch_k(c, XFH_K_CONV_MFMA, layer, kern, dim3(ntile, CGS, B), dim3(64 * NW), LDS, aa);
    return hipGetLastError();
}

template <int CIN, int COUT, int KS, int ST, int WM, int WN, int NT, int WW, int PRO, int EPI>
static hipError_t conv_mfma_p_launch(xfh_ctx* c, const ConvArgs& a, int B, int* npart_out, int layer) {
    constexpr int WH = 32 / WW, TH = WM * WH, TW = WW;
    constexpr int COUTP = WN * NT * 32;
    constexpr int TIH = (TH - 1) * ST + KS, TIW = (TW - 1) * ST + KS;
    constexpr size_t LDS = sizeof(float) * ((size_t)TIH * TIW * (CIN + 4) + (size_t)COUTP * (KS * KS * CIN + 4)) + sizeof(double) * WM * COUTP * 2;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    ConvArgs aa = a;
    aa.dbg = layer;
    aa.tiles_x = (a.Wout + TW - 1) / TW;
    const int ntile = aa.tiles_x * ((a.Hout + TH - 1) / TH);
    if (npart_out) *npart_out = ntile;
    auto kern = k_conv_mfma_p<CIN, COUT, KS, ST, WM, WN, NT, WW, PRO, EPI>;
    XFH_SET_LDS_ATTR_ONCE(c, kern, LDS);
    const int total = ntile * B;
    const int per_cu = (int)((160 * 1024) / LDS) > 4 ? 4 : (int)((160 * 1024) / LDS);
    const int grid = total < 256 * per_cu ? total : 256 * per_cu;
    launch_k(c, XFH_K_CONV_MFMA, layer, kern, dim3(grid), dim3(64 * WM * WN), LDS, aa, ntile, total);
    return hipGetLastError();
}

// Batch regimes (tests/test_gpu_extract.py::test_batch_is_per_frame checks that they agree bit for bit):
//   B > 8 : persistent kernels for the short-K layers (k_conv_mfma_p), k_bn_finalize after every layer, the keypoint branch on the
//           ctx's second stream;
//   B <= 8: a dependent launch costs ~3 us on this GPU whatever it does (and a dependency on ANOTHER queue 7-11 us), so EVERY consumer
//           folds the statistics of its producer itself (bn_fold in each workgroup: same order, same bits) and k_bn_finalize is never
//           launched -- at large batches the per-workgroup re-read of the partials would cost more than the launch (A/B: the threshold
//           at 4 or at 16 is slower on either side); in the batch-statistics mode the keypoint branch RIDES on block1.3 .. block3.0
//           (k_conv_mfma_ride) and everything runs on one stream; the 24-channel layers take all nine / three taps per weight chunk;
//   B <= 32: the 3x3 layers with >= 64 input channels on 16x16x4 MFMAs (k_conv_mfma16); by tile count (not by batch size) their
//           output channels are split over two workgroups (split_channels) or, for the 1/8-resolution layers, the tiles grow from
//           2x16 to 4x16 pixels (XFH_M16_TALL).
#ifndef XFH_M16_TALL
#define XFH_M16_TALL 256
#endif
static bool persistent(int B) { return B > 8; }
bool consumer_fold(int B) { return B <= 8; }
// k_conv_mfma16 splits its output channels over two workgroups (one wave per SIMD, see the kernel) while the split grid still fits one
// round of the 256 CUs; beyond that the extra staging costs more than the idle SIMDs it fills (150 tiles -> 300 workgroups: measured slower)
static bool split_channels(int Hout, int Wout, int TH, int TW, int B) { return (long)((Hout + TH - 1) / TH) * ((Wout + TW - 1) / TW) * B <= 128; }
static bool small_batch(int B) { return B <= 32; }      // k_conv_mfma16 for the 3x3 layers with >= 64 input channels: +23 % at B = 9, +3 % at B = 32, even at 48, -3 % at 64 (A/B on one box)

template <int CIN, int COUT, int ST, int PRO>
static hipError_t conv_direct_launch(xfh_ctx* c, const ConvArgs& a, int B, int* npart_out, int layer) {
    ConvArgs aa = a;
    aa.dbg = layer;
    aa.tiles_x = (a.Wout + 15) / 16;
    const int ntile = aa.tiles_x * ((a.Hout + 15) / 16);
    if (npart_out) *npart_out = ntile;
    if (a.bias) launch_k(c, XFH_K_CONV_DIRECT, layer, k_conv_direct<CIN, COUT, ST, PRO, false, EPI_BIAS_RELU>, dim3(ntile, 1, B), dim3(256), 0, aa);
    else if ((PRO == PRO_BN || PRO == PRO_L0) && a.st.part) launch_k(c, XFH_K_CONV_DIRECT, layer, k_conv_direct<CIN, COUT, ST, PRO, PRO == PRO_BN || PRO == PRO_L0, EPI_STATS>, dim3(ntile, 1, B), dim3(256), 0, aa);
    else launch_k(c, XFH_K_CONV_DIRECT, layer, k_conv_direct<CIN, COUT, ST, PRO, false, EPI_STATS>, dim3(ntile, 1, B), dim3(256), 0, aa);
    return hipGetLastError();
}

// number of statistic partials a layer produces per frame (needed to size buffers up front)
int conv_layer_npart(int li, int Hout, int Wout) {
    auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
    if (li == 0) return cdiv(Wout, L0S_TW) * cdiv(Hout, L0S_TH);          // k_block1_stats
    if (li < 3) return cdiv(Wout, 16) * cdiv(Hout, 16);
    switch (li) {
        case 7: case 16: case 17: return cdiv(Wout, 16) * cdiv(Hout, 2);     // 8x16 pixels; 2x16 in the small-batch configuration (the larger count sizes the buffers)
        case 9: case 10: case 11: return cdiv(Wout, 8) * cdiv(Hout, 4);              // single frame: 4x8-pixel tiles (k_conv_mfma16); batches: 16x8
        case 12: case 13: case 14: return cdiv(Wout, 4) * cdiv(Hout, 4);     // single frame: 4x4-pixel tiles; batches: 16x8 / 4x8
        case 15: return cdiv(Wout, 8) * cdiv(Hout, 4);                       // WM=1, WW=8
        default: return cdiv(Wout, 16) * cdiv(Hout, 8);                       // WM=4, WW=16
    }
}

// statistics of layer j as its consumer gets them in a batch of B frames
StatSrc stat_src(xfh_ctx* c, int j, int B) {
    StatSrc s{};
    s.stat = c->stat[j];
    if (c->cfg.bn_mode == XFH_BN_BATCH_STATS && consumer_fold(B)) {
        s.part = c->part[j]; s.part_stride = c->part_stride[j]; s.npart = c->npart[j];
        s.count = (double)c->lh[j] * (double)c->lw[j]; s.stat_out = c->stat[j];
    }
    return s;
}

// BasicLayer li: conv + statistics partials (+ finalize for large batches).  `in`: producer tensor; src >= 0: the
// BasicLayer whose BatchNorm + ReLU is applied while staging (PRO_BN / PRO_B2IN / PRO_FUSE), src == -1: plain input,
// src == -2: InstanceNorm of the image (block1.0).
template <int EPI>
static hipError_t launch_basic_layer_t(xfh_ctx* c, int li, const float* in, size_t in_stride, int src, int pro, int Hin, int Win, int B) {
    const LayerSpec& L = XFH_LAYERS[li];
    const int pad = L.ks / 2;
    const int Hout = (Hin + 2 * pad - L.ks) / L.stride + 1, Wout = (Win + 2 * pad - L.ks) / L.stride + 1;
    c->lh[li] = Hout; c->lw[li] = Wout;
    ConvArgs a{};
    a.in = in; a.in_stride = in_stride; a.Hin = Hin; a.Win = Win;
    if (src >= 0) a.st = stat_src(c, src, B);
    else if (src == -2) a.st.stat = c->xstat;
    a.w = (li < 3) ? c->w.direct[li] : c->w.mfma[li];
    a.bias = (EPI == EPI_BIAS_RELU) ? c->w.bn_bias[li] : nullptr;
    a.out = c->raw[li]; a.out_stride = c->raw_stride[li]; a.Hout = Hout; a.Wout = Wout;
    a.part = c->part[li]; a.part_stride = c->part_stride[li];
    const bool running = c->cfg.bn_mode != XFH_BN_BATCH_STATS;
    if (pro == PRO_B2IN) {
        const size_t xs = (size_t)c->Hmax * c->Wmax;
        a.pool = c->skip_pool; a.pool_stride = xs / 16; a.skip_w = c->w.skip_w; a.skip_b = c->w.skip_b;
    }
    if (pro == PRO_FUSE) {
        a.r4 = c->raw[11]; a.s4 = c->raw_stride[11]; a.H4 = c->lh[11]; a.W4 = c->lw[11]; a.st4 = stat_src(c, 11, B);
        a.r5 = c->raw[15]; a.s5 = c->raw_stride[15]; a.H5 = c->lh[15]; a.W5 = c->lw[15]; a.st5 = stat_src(c, 15, B);
    }
    int np = 0;
    hipError_t e = hipSuccess;
    switch (li) {
        case 1:
            a.xstat = c->xstat; a.w0 = c->w.direct[0]; a.bias0 = (EPI == EPI_BIAS_RELU) ? c->w.bn_bias[0] : nullptr;
            e = conv_direct_launch<4, 8, 2, PRO_L0>(c, a, B, &np, li);
            break;
        case 2: e = conv_direct_launch<8, 8, 1, PRO_BN>(c, a, B, &np, li); break;
        case 3:
            // all nine taps of the weights in one LDS chunk (no barrier inside the K = 72 loop): 49.5 -> 46.2 us at B = 32;
            // the same form measured slower for the 24 -> 24 layers (95 vs 91 us)
            a.w = c->w.alt[li];
            if (persistent(B)) e = conv_mfma_p_launch<8, 24, 3, 2, 4, 1, 1, 16, PRO_BN, EPI>(c, a, B, &np, li);
            else e = conv_mfma_launch<8, 24, 3, 2, 4, 1, 1, 16, PRO_BN, EPI, 64, 9>(c, a, B, &np, li);
            break;
        case 4:                                                                                                 // input = relu(bn(block1.3)) + skip1(x), computed while staging
            if (persistent(B)) { a.w = c->w.alt[li]; e = conv_mfma_p_launch<24, 24, 3, 1, 4, 1, 1, 16, PRO_B2IN, EPI>(c, a, B, &np, li); }
            else { a.w = c->w.alt[li]; e = conv_mfma_launch<24, 24, 3, 1, 4, 1, 1, 16, PRO_B2IN, EPI, 64, 9>(c, a, B, &np, li); }      // B <= 8: all nine taps in one chunk -- one weight round trip, no barrier in the K loop
            break;
        case 5:
            if (persistent(B)) { a.w = c->w.alt[li]; e = conv_mfma_p_launch<24, 24, 3, 1, 4, 1, 1, 16, PRO_BN, EPI>(c, a, B, &np, li); }
            else { a.w = c->w.alt[li]; e = conv_mfma_launch<24, 24, 3, 1, 4, 1, 1, 16, PRO_BN, EPI, 64, 9>(c, a, B, &np, li); }
            break;
        case 6:
            if (consumer_fold(B)) { a.w = c->w.alt[li]; e = conv_mfma_launch<24, 64, 3, 2, 4, 1, 2, 16, PRO_BN, EPI, 64, 3>(c, a, B, &np, li); }   // three taps per chunk: 3 weight round trips instead of 9
            else if constexpr ((XFH_CONV_T & 2) != 0) { if (!small_batch(B)) e = conv_mfma_t_launch<24, 64, 3, 2, 4, 1, 2, 16, PRO_BN, EPI>(c, a, B, &np, li); else e = conv_mfma_launch<24, 64, 3, 2, 4, 1, 2, 16, PRO_BN, EPI>(c, a, B, &np, li); }
            else e = conv_mfma_launch<24, 64, 3, 2, 4, 1, 2, 16, PRO_BN, EPI>(c, a, B, &np, li);
            break;      // 8x16 pixels; 8x8 pixels (46 KB, three workgroups per CU) measured 526 -> 757 us at B = 256
        case 7: case 17: case 16:
            // the dominant 3x3 64->64 instance: 8x16 pixels x 64 channels per workgroup, 8 waves (4 x 2), 32-channel weight
            // chunks (67 KB LDS -> 2 workgroups per CU); measured 70 us vs 77 us for the 4-wave / 64-channel-chunk form at
            // B = 16 (profiles/r01_conv_cfg.log); 6x16 / 10x16 pixels on 6 / 10 waves (no wasted rows at VGA) 770 -> 959 / 888 us at B = 256 (round 4).  Single frame: 2x16 pixels per workgroup, 2 waves, three taps per chunk.
            // block_fusion.0 (16) builds its input x3 + up2(x4) + up4(x5) while staging.
            if (small_batch(B)) {
                a.w = c->w.m16[li];
                // 2x16 pixels, 8 waves of 16 x 16 (channels split over two workgroups while that fits one round of the CUs; 150 tiles at VGA:
                // not split); from XFH_M16_TALL tiles on (a 1280x720 frame: 440), 4x16 pixels on 16 waves: less halo per staged pixel
                const long t2 = (long)((Hout + 1) / 2) * ((Wout + 15) / 16) * B;
                if (li == 16) e = split_channels(Hout, Wout, 2, 16, B) ? conv_mfma16_launch<64, 64, 1, 16, 2, PRO_FUSE, EPI, 2>(c, a, B, &np, li)
                                : t2 > XFH_M16_TALL ? conv_mfma16_launch<64, 64, 1, 16, 4, PRO_FUSE, EPI>(c, a, B, &np, li) : conv_mfma16_launch<64, 64, 1, 16, 2, PRO_FUSE, EPI>(c, a, B, &np, li);
                else e = split_channels(Hout, Wout, 2, 16, B) ? conv_mfma16_launch<64, 64, 1, 16, 2, PRO_BN, EPI, 2>(c, a, B, &np, li)
                       : t2 > XFH_M16_TALL ? conv_mfma16_launch<64, 64, 1, 16, 4, PRO_BN, EPI>(c, a, B, &np, li) : conv_mfma16_launch<64, 64, 1, 16, 2, PRO_BN, EPI>(c, a, B, &np, li);
            } else {
                a.w = c->w.alt[li];
                if (li == 16) {
                    // the coarse maps activated once (k_act_pyramid) instead of at every one of block_fusion.0's eight taps per staged value
                    const int n4 = c->lh[11] * c->lw[11] * 64, n5 = c->lh[15] * c->lw[15] * 64;
                    hipLaunchKernelGGL(k_act_pyramid, dim3(((n4 + n5) / 4 + 255) / 256, 1, B), dim3(256), 0, c->stream, (const float*)c->raw[11], c->raw_stride[11],
                                       (const float*)c->stat[11], n4, (const float*)c->raw[15], c->raw_stride[15], (const float*)c->stat[15], n5, c->act4, c->act5);
                    a.r4 = c->act4; a.r5 = c->act5;
                    e = conv_mfma_launch<64, 64, 3, 1, 4, 2, 1, 16, PRO_FUSEA, EPI, 32>(c, a, B, &np, li);
                }
                else if constexpr ((XFH_CONV_T & 16) != 0) e = conv_mfma_t_launch<64, 64, 3, 1, 4, 2, 1, 16, PRO_BN, EPI, 32>(c, a, B, &np, li);
                else e = conv_mfma_launch<64, 64, 3, 1, 4, 2, 1, 16, PRO_BN, EPI, 32>(c, a, B, &np, li);
            }
            break;
        case 8:
            if (persistent(B)) e = conv_mfma_p_launch<64, 64, 1, 1, 4, 1, 2, 16, PRO_BN, EPI>(c, a, B, &np, li);
            else e = conv_mfma_launch<64, 64, 1, 1, 4, 1, 2, 16, PRO_BN, EPI>(c, a, B, &np, li);
            break;
        case 9:
            // stride 2 (4.5 input pixels per output pixel): 113 KB of LDS = one workgroup per CU.  Measured alternative: 4x8 pixels with
            // 32-channel chunks (60 KB, two workgroups per CU) 403 -> 558 us at B = 256 -- a workgroup streams the whole 147 KB weight
            // matrix from L2 for its tile, so halving the tile doubles that traffic; the small maps are bound by it
            if (small_batch(B)) { a.w = c->w.m16[li]; e = split_channels(Hout, Wout, 4, 8, B) ? conv_mfma16_launch<64, 64, 2, 8, 2, PRO_BN, EPI, 2>(c, a, B, &np, li) : conv_mfma16_launch<64, 64, 2, 8, 2, PRO_BN, EPI>(c, a, B, &np, li); }     // single frame: 4x8 pixels, 8 waves of 16 x 16
            else if constexpr ((XFH_CONV_T & 1) != 0) e = conv_mfma_t_launch<64, 64, 3, 2, 2, 2, 1, 8, PRO_BN, EPI>(c, a, B, &np, li);   // persistent, next tile prefetched
            else e = conv_mfma_launch<64, 64, 3, 2, 2, 2, 1, 8, PRO_BN, EPI>(c, a, B, &np, li);
            break;
        case 10: case 11:
            if (small_batch(B)) { a.w = c->w.m16[li]; e = split_channels(Hout, Wout, 4, 8, B) ? conv_mfma16_launch<64, 64, 1, 8, 2, PRO_BN, EPI, 2>(c, a, B, &np, li) : conv_mfma16_launch<64, 64, 1, 8, 2, PRO_BN, EPI>(c, a, B, &np, li); }                    // 4x8 pixels, 8 waves of 16 x 16
            else if constexpr ((XFH_CONV_T & 4) != 0) { a.w = c->w.alt[li]; e = conv_mfma_t_launch<64, 64, 3, 1, 4, 2, 1, 8, PRO_BN, EPI, 32>(c, a, B, &np, li); }
            else { a.w = c->w.alt[li]; e = conv_mfma_launch<64, 64, 3, 1, 4, 2, 1, 8, PRO_BN, EPI, 32>(c, a, B, &np, li); }   // 16x8 pixels, 8 waves, 32-channel chunks (67 KB): half the weight streaming per pixel of the 8x8 form
            break;
        case 12:
            if (small_batch(B)) { a.w = c->w.m16[li]; e = split_channels(Hout, Wout, 4, 4, B) ? conv_mfma16_launch<64, 128, 2, 4, 1, PRO_BN, EPI, 2>(c, a, B, &np, li) : conv_mfma16_launch<64, 128, 2, 4, 1, PRO_BN, EPI>(c, a, B, &np, li); break; }     // single frame: 4x4 pixels, 8 waves of 16 x 16
            a.w = c->w.alt[li];
            if constexpr ((XFH_CONV_T & 2) != 0) e = conv_mfma_t_launch<64, 128, 3, 2, 1, 4, 1, 8, PRO_BN, EPI, 32>(c, a, B, &np, li);
            else e = conv_mfma_launch<64, 128, 3, 2, 1, 4, 1, 8, PRO_BN, EPI, 32>(c, a, B, &np, li);
            break;      // 32-channel chunks: 111 -> 78 KB of LDS, two workgroups per CU (212 -> 168 us at B = 256)
        case 13: case 14:       // 32-channel weight chunks: 69 KB LDS -> 2 workgroups per CU, 58 -> 50 us at B = 32
            if constexpr ((XFH_CONV_T & 8) != 0) { if (!small_batch(B)) { a.w = c->w.alt[li]; e = conv_mfma_t_launch<128, 128, 3, 1, 4, 4, 1, 8, PRO_BN, EPI, 32>(c, a, B, &np, li); break; } }
            if (!small_batch(B)) { a.w = c->w.alt[li]; e = conv_mfma_launch<128, 128, 3, 1, 4, 4, 1, 8, PRO_BN, EPI, 32>(c, a, B, &np, li); }   // 16x8 pixels x 128 channels, 16 waves (132 KB): the 590 KB weight matrix is streamed once per 128 pixels
            else { a.w = c->w.m16[li]; e = split_channels(Hout, Wout, 4, 4, B) ? conv_mfma16_launch<128, 128, 1, 4, 1, PRO_BN, EPI, 2>(c, a, B, &np, li) : conv_mfma16_launch<128, 128, 1, 4, 1, PRO_BN, EPI>(c, a, B, &np, li); }                                   // 4x4 pixels, 8 waves of 16 x 16
            break;
        case 15: e = conv_mfma_launch<128, 64, 1, 1, 1, 2, 1, 8, PRO_BN, EPI>(c, a, B, &np, li); break;
        case 18:                                                                                               // input: feats
            if (persistent(B)) e = conv_mfma_p_launch<64, 64, 1, 1, 4, 1, 2, 16, PRO_PLAIN, EPI>(c, a, B, &np, li);
            else e = conv_mfma_launch<64, 64, 1, 1, 4, 1, 2, 16, PRO_PLAIN, EPI>(c, a, B, &np, li);
            break;
        case 20:                                                                                               // input: unfold2d(x-hat), read from the image while staging
            a.xstat = c->xstat;
            if (persistent(B)) e = conv_mfma_p_launch<64, 64, 1, 1, 4, 1, 2, 16, PRO_UNFOLD, EPI>(c, a, B, &np, li);
            else e = conv_mfma_launch<64, 64, 1, 1, 4, 1, 2, 16, PRO_UNFOLD, EPI>(c, a, B, &np, li);
            break;
        case 19: case 21: case 22:
            if (persistent(B)) e = conv_mfma_p_launch<64, 64, 1, 1, 4, 1, 2, 16, PRO_BN, EPI>(c, a, B, &np, li);
            else e = conv_mfma_launch<64, 64, 1, 1, 4, 1, 2, 16, PRO_BN, EPI>(c, a, B, &np, li);
            break;
        default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    c->npart[li] = np;
    if (running || consumer_fold(B)) return hipGetLastError();     // statistics from the weight file / folded by the consumers
    hipLaunchKernelGGL(k_bn_finalize, dim3(B), dim3(256), 0, c->stream, (const double*)c->part[li], c->part_stride[li], np,
                       L.cout, (double)Hout * (double)Wout, c->stat[li]);
    return hipGetLastError();
}

// ---- a backbone layer with a rider (k_conv_mfma_ride): batches <= 8, batch-statistics mode -------------------------------------
// ConvArgs of BasicLayer li in the batch-statistics mode (the part of launch_basic_layer_t that does not depend on the kernel form)
static ConvArgs stats_layer_args(xfh_ctx* c, int li, const float* in, size_t in_stride, int src, int pro, int Hin, int Win, int B, int TH, int TW, int* ntile) {
    const LayerSpec& L = XFH_LAYERS[li];
    const int pad = L.ks / 2;
    const int Hout = (Hin + 2 * pad - L.ks) / L.stride + 1, Wout = (Win + 2 * pad - L.ks) / L.stride + 1;
    c->lh[li] = Hout; c->lw[li] = Wout;
    ConvArgs a{};
    a.in = in; a.in_stride = in_stride; a.Hin = Hin; a.Win = Win;
    if (src >= 0) a.st = stat_src(c, src, B);
    a.w = c->w.mfma[li];
    a.out = c->raw[li]; a.out_stride = c->raw_stride[li]; a.Hout = Hout; a.Wout = Wout;
    a.part = c->part[li]; a.part_stride = c->part_stride[li];
    a.xstat = c->xstat;
    if (pro == PRO_B2IN) {
        const size_t xs = (size_t)c->Hmax * c->Wmax;
        a.pool = c->skip_pool; a.pool_stride = xs / 16; a.skip_w = c->w.skip_w; a.skip_b = c->w.skip_b;
    }
    a.dbg = li;
    a.tiles_x = (Wout + TW - 1) / TW;
    *ntile = a.tiles_x * ((Hout + TH - 1) / TH);
    c->npart[li] = *ntile;
    return a;
}
template <int CIN, int COUT, int KS, int ST, int WM, int WN, int NT, int WW, int PRO, int CBMAX, int TPC, int RIDER, int EPI = EPI_STATS>
static hipError_t ride_launch(xfh_ctx* c, const ConvArgs& a, int n_host, const ConvArgs& r, const Kp4Args& k, int n_rider, int B, int layer) {
    constexpr int WH = 32 / WW, TH = WM * WH, TW = WW, COUTP = WN * NT * 32;
    constexpr int TIH = (TH - 1) * ST + KS, TIW = (TW - 1) * ST + KS;
    constexpr int CB = CIN > CBMAX ? CBMAX : CIN;
    constexpr int NWBUF = (KS * KS * (CIN / CB) / TPC) > 1 ? 2 : 1;
    constexpr int STATF = PRO == PRO_B2IN ? 4 * CIN : 2 * CIN;
    constexpr size_t LDS_HOST = sizeof(float) * ((size_t)TIH * TIW * (CIN + 4) + NWBUF * (size_t)COUTP * (TPC * CB + 4) + STATF);
    constexpr size_t LDS_1X1 = sizeof(float) * ((size_t)8 * 16 * 68 + (size_t)64 * 68 + 128);      // k_conv_mfma<64, 64, 1, 1, 4, 1, 2, 16>
    constexpr size_t LDS_RIDER = RIDER == RIDE_KP4 ? sizeof(float) * HK4_LDS_FLOATS : LDS_1X1;
    constexpr size_t LDS = LDS_HOST > LDS_RIDER ? LDS_HOST : LDS_RIDER;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert(sizeof(double) * WM * COUTP * 2 <= LDS_HOST && sizeof(double) * 512 <= sizeof(float) * (size_t)TIH * TIW * (CIN + 4), "scratch");
    auto kern = k_conv_mfma_ride<CIN, COUT, KS, ST, WM, WN, NT, WW, PRO, CBMAX, TPC, RIDER, EPI>;
    XFH_SET_LDS_ATTR_ONCE(c, kern, LDS);
    launch_k(c, XFH_K_CONV_MFMA, layer, kern, dim3(n_host + n_rider, 1, B), dim3(256), LDS, a, n_host, r, k);
    return hipGetLastError();
}
bool ride_mode(const xfh_ctx* c, int B) { return consumer_fold(B) && !c->no_ride; }      // (A/B on one box: +9.5 % at 2 frames, +5.5 % at 4, even at 8)
// host: BasicLayer li in 3 .. 6 (block1.3, block2.0, block2.1, block3.0); rider: step li - 3 of the keypoint branch
hipError_t launch_layer_with_rider(xfh_ctx* c, int li, const float* in, size_t in_stride, int src, int pro, int Hin, int Win, int B,
                                   const float* K1h, size_t k1h_stride) {
    const int h8 = c->H / 8, w8 = c->W / 8;
    const size_t xs = (size_t)c->Hmax * c->Wmax;
    int nh = 0, nr = 0;
    ConvArgs r{}; Kp4Args k{};
    const bool folded = c->cfg.bn_mode == XFH_BN_RUNNING_FOLDED;       // folded BatchNorms: relu(. + bias) in the epilogue of host and rider, no statistics
    switch (li) {
        case 3: {
            ConvArgs a = stats_layer_args(c, li, in, in_stride, src, pro, Hin, Win, B, 8, 16, &nh); a.w = c->w.alt[li];
            r = stats_layer_args(c, 20, c->X, xs, -1, PRO_UNFOLD, h8, w8, B, 8, 16, &nr);
            if (folded) { a.bias = c->w.bn_bias[li]; r.bias = c->w.bn_bias[20]; return ride_launch<8, 24, 3, 2, 4, 1, 1, 16, PRO_BN, 64, 9, RIDE_UNFOLD, EPI_BIAS_RELU>(c, a, nh, r, k, nr, B, li); }
            return ride_launch<8, 24, 3, 2, 4, 1, 1, 16, PRO_BN, 64, 9, RIDE_UNFOLD>(c, a, nh, r, k, nr, B, li);
        }
        case 4: {
            ConvArgs a = stats_layer_args(c, li, in, in_stride, src, pro, Hin, Win, B, 8, 16, &nh); a.w = c->w.alt[li];
            r = stats_layer_args(c, 21, c->raw[20], c->raw_stride[20], 20, PRO_BN, h8, w8, B, 8, 16, &nr);
            if (folded) { a.bias = c->w.bn_bias[li]; r.bias = c->w.bn_bias[21]; return ride_launch<24, 24, 3, 1, 4, 1, 1, 16, PRO_B2IN, 64, 9, RIDE_BN, EPI_BIAS_RELU>(c, a, nh, r, k, nr, B, li); }
            return ride_launch<24, 24, 3, 1, 4, 1, 1, 16, PRO_B2IN, 64, 9, RIDE_BN>(c, a, nh, r, k, nr, B, li);
        }
        case 5: {
            ConvArgs a = stats_layer_args(c, li, in, in_stride, src, pro, Hin, Win, B, 8, 16, &nh); a.w = c->w.alt[li];
            r = stats_layer_args(c, 22, c->raw[21], c->raw_stride[21], 21, PRO_BN, h8, w8, B, 8, 16, &nr);
            if (folded) { a.bias = c->w.bn_bias[li]; r.bias = c->w.bn_bias[22]; return ride_launch<24, 24, 3, 1, 4, 1, 1, 16, PRO_BN, 64, 9, RIDE_BN, EPI_BIAS_RELU>(c, a, nh, r, k, nr, B, li); }
            return ride_launch<24, 24, 3, 1, 4, 1, 1, 16, PRO_BN, 64, 9, RIDE_BN>(c, a, nh, r, k, nr, B, li);
        }
        case 6: {
            ConvArgs a = stats_layer_args(c, li, in, in_stride, src, pro, Hin, Win, B, 8, 16, &nh); a.w = c->w.alt[li];
            k = Kp4Args{(const float*)c->raw[22], stat_src(c, 22, B), c->raw_stride[22], (const float*)c->w.kp3_w, (const float*)c->w.kp3_b, h8, w8, const_cast<float*>(K1h), k1h_stride};
            nr = (h8 * w8 + HK4_PX - 1) / HK4_PX;
            if (folded) { a.bias = c->w.bn_bias[li]; return ride_launch<24, 64, 3, 2, 4, 1, 2, 16, PRO_BN, 64, 3, RIDE_KP4, EPI_BIAS_RELU>(c, a, nh, r, k, nr, B, li); }
            return ride_launch<24, 64, 3, 2, 4, 1, 2, 16, PRO_BN, 64, 3, RIDE_KP4>(c, a, nh, r, k, nr, B, li);
        }
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_basic_layer(xfh_ctx* c, int li, const float* in, size_t in_stride, int src, int pro, int Hin, int Win, int B) {
    if (c->cfg.bn_mode == XFH_BN_RUNNING_FOLDED) return launch_basic_layer_t<EPI_BIAS_RELU>(c, li, in, in_stride, src, pro, Hin, Win, B);
    return launch_basic_layer_t<EPI_STATS>(c, li, in, in_stride, src, pro, Hin, Win, B);
}

// block1.0 in the batch-statistics mode: its BatchNorm statistics (the map itself is recomputed inside block1.1) + skip1's AvgPool4 +,
// for B <= 8, the fold of the image statistics (xs.part != null) that every later kernel reads from c->xstat
hipError_t launch_block1_stats(xfh_ctx* c, const StatSrc& xs, int H, int W, int B) {
    const int np = conv_layer_npart(0, H, W);
    c->lh[0] = H; c->lw[0] = W; c->npart[0] = np;
    const size_t xsz = (size_t)c->Hmax * c->Wmax;
    launch_k(c, XFH_K_CONV_DIRECT, 0, k_block1_stats, dim3(np, 1, B), dim3(256), 0, (const float*)c->X, xsz, xs, H, W,
             (W + L0S_TW - 1) / L0S_TW, (const float*)c->w.direct[0], c->part[0], c->part_stride[0], c->skip_pool, xsz / 16);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || consumer_fold(B)) return e;
    hipLaunchKernelGGL(k_bn_finalize, dim3(B), dim3(256), 0, c->stream, (const double*)c->part[0], c->part_stride[0], np, 4, (double)H * (double)W, c->stat[0]);
    return hipGetLastError();
}

// ---- chains of 1x1 layers (k_chain1x1) ----------------------------------------------------------
template <int NL, int WM, int PRO, int MID0, int EPI, bool SSTAT = false>
static hipError_t chain_launch(xfh_ctx* c, const ChainArgs& ca, int B, int* npart_out, int layer) {
    constexpr int TH = 2 * WM, TW = 16;
    constexpr size_t LDS = sizeof(float) * ((size_t)TH * TW * 68 + (size_t)NL * 64 * 68 + (SSTAT ? 128 : 0)) + sizeof(double) * WM * 64 * 2;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    ChainArgs aa = ca;
    aa.a.tiles_x = (ca.a.Wout + TW - 1) / TW;
    const int ntile = aa.a.tiles_x * ((ca.a.Hout + TH - 1) / TH);
    if (npart_out) *npart_out = ntile;
    auto kern = k_chain1x1<NL, WM, PRO, MID0, EPI, SSTAT>;
    XFH_SET_LDS_ATTR_ONCE(c, kern, LDS);
    const int total = ntile * B;
    const int per_cu = (int)((160 * 1024) / LDS);
    const int grid = total < 256 * per_cu ? total : 256 * per_cu;
    launch_k(c, XFH_K_CONV_MFMA, layer, kern, dim3(grid), dim3(64 * WM), LDS, aa, ntile, total);
    return hipGetLastError();
}

// block_fusion.2 (-> feats) and the heatmap-head layers that follow it without a BatchNorm in between: heatmap_head.0 always
// (fusion.2 is a bare Conv2d), heatmap_head.1 too when the BatchNorms are folded.  *done = number of head layers computed.
hipError_t launch_fusion_chain(xfh_ctx* c, int Hh, int Wh, int B, int* done) {
    *done = 0;
    const bool folded = c->cfg.bn_mode == XFH_BN_RUNNING_FOLDED;
    ChainArgs ca{};
    ConvArgs& a = ca.a;
    a.in = c->raw[17]; a.st = stat_src(c, 17, B); a.in_stride = c->raw_stride[17]; a.Hin = Hh; a.Win = Wh; a.Hout = Hh; a.Wout = Wh;
    a.w = c->w.fus2; a.bias = c->w.fus2_bias;
    ca.mid_out = c->feats; ca.mid_stride = c->raw_stride[17];
    ca.w1 = c->w.mfma[18];
    int np = 0;
    hipError_t e;
    if (!persistent(B) && folded) {
        // folded BatchNorms, batches <= 8: block_fusion.2 -> heatmap_head.0 -> heatmap_head.1 in ONE launch.  A launch of one frame's 40 tiles is
        // 7-10 us of mostly fixed time (boundary, weights, ramp); the three-stage form that loses 3 % at 256 frames (below) saves two of them here
        ca.bias1 = c->w.bn_bias[18]; ca.w2 = c->w.mfma[19]; ca.bias2 = c->w.bn_bias[19];
        a.out = c->raw[19]; a.out_stride = c->raw_stride[19];
        e = chain_launch<3, 4, PRO_BN, MID_BIAS_STORE, EPI_BIAS_RELU>(c, ca, B, &np, 19);
        if (e != hipSuccess) return e;
        c->lh[18] = Hh; c->lw[18] = Wh; c->npart[18] = np; c->lh[19] = Hh; c->lw[19] = Wh; c->npart[19] = np;
        *done = 2;
        return hipSuccess;
    }
    if (consumer_fold(B)) {
        // batches <= 8, BatchNorm statistics per frame or from the weight file: block_fusion.2 -> heatmap_head.0 in one launch as for B > 8,
        // the statistics of block_fusion.1 folded from its partials by each workgroup like every other consumer of this regime does
        a.out = c->raw[18]; a.out_stride = c->raw_stride[18]; a.part = c->part[18]; a.part_stride = c->part_stride[18];
        e = chain_launch<2, 4, PRO_BN, MID_BIAS_STORE, EPI_STATS, true>(c, ca, B, &np, 18);
        if (e != hipSuccess) return e;
        c->lh[18] = Hh; c->lw[18] = Wh; c->npart[18] = np;
        *done = 1;
        return hipSuccess;
    }
    if (!folded) {
        a.out = c->raw[18]; a.out_stride = c->raw_stride[18]; a.part = c->part[18]; a.part_stride = c->part_stride[18];
        e = chain_launch<2, 4, PRO_BN, MID_BIAS_STORE, EPI_STATS>(c, ca, B, &np, 18);
        if (e != hipSuccess) return e;
        c->lh[18] = Hh; c->lw[18] = Wh; c->npart[18] = np;
        *done = 1;
        if (c->cfg.bn_mode == XFH_BN_BATCH_STATS)
            hipLaunchKernelGGL(k_bn_finalize, dim3(B), dim3(256), 0, c->stream, (const double*)c->part[18], c->part_stride[18], np, 64, (double)Hh * (double)Wh, c->stat[18]);
        return hipGetLastError();
    }
    // folded BatchNorms: more layers could follow in the same pass (heatmap_head.1 here; keypoint_head.0 -> .1 -> .2).  Measured at
    // 256 frames: three stages (87 KB of LDS: 4 waves per CU, each stage waiting for the previous one's MFMAs and an LDS round trip)
    // -3 % on the whole step, two-stage chains for both heads -0.3 % -- these layers are bound by the vector pipe (MFMA + the
    // VALU work of staging / handing on), not by the 315 MB each intermediate map costs in HBM traffic, so only this one is kept
    ca.bias1 = c->w.bn_bias[18];
    a.out = c->raw[18]; a.out_stride = c->raw_stride[18];
    e = chain_launch<2, 4, PRO_BN, MID_BIAS_STORE, EPI_BIAS_RELU>(c, ca, B, &np, 18);
    if (e != hipSuccess) return e;
    c->lh[18] = Hh; c->lw[18] = Wh; c->npart[18] = np;
    *done = 1;
    return hipSuccess;
}

// InstanceNorm statistics of the image reuse the finalize kernel with C = 1
hipError_t launch_finalize_image(xfh_ctx* c, int B, int npart, double count) {
    hipLaunchKernelGGL(k_bn_finalize, dim3(B), dim3(256), 0, c->stream, (const double*)c->pre_part,
                       (size_t)c->pre_npart * 2, npart, 1, count, c->xstat);
    return hipGetLastError();
}
