/*
 * xfeat_oracle.c -- plain-C restatement of the reference's XFeat front end.
 * TEST INFRASTRUCTURE ONLY; see xfeat_oracle.h for the parity status ("parity unpinned"
 * by reference execution; pinned against the ATen-operator restatement).
 *
 * Every function cites the reference lines it follows (paths relative to the upstream
 * tree).  Numerics conventions (shared with the HIP kernels so that diffs stay tiny):
 *   - convolutions accumulate in fp32 as one fmaf chain per output, K ordered (ky,kx,ci),
 *     starting from 0; a bias, where the layer has one, is added after the chain;
 *   - BatchNorm/InstanceNorm statistics are fp64 (mean, biased variance), applied in
 *     fp32 as ONE fused multiply-add  x * alpha + beta  with alpha = rstd, beta = -(mean * rstd) (both fp32) -- the arithmetic of
 *     ATen's batch_norm / instance_norm CPU kernels (alpha = invstd * weight, beta = bias - mean * alpha, out = x * alpha + beta
 *     compiled to an fma: bit-identical on every element, tests/test_oracle.py::test_norm_apply_is_atens_fma)
 *     [reference: training-mode BN, SURVEY.md Q1];
 *   - L2 norms use an fp64 sum of squares, then fp32 sqrt/max/divide.
 * Build: see oracle/Makefile (-O2 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#include "xfeat_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ layer table */
/* XFeatModel::XFeatModel, src/XFeat.cc:30-90: the 23 BasicLayer(cin,cout,k,stride,pad) */
typedef struct { int cin, cout, ks, stride; const char* name; } layer_spec;
static const layer_spec LAYERS[XFO_NUM_LAYERS] = {
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

struct xfo_ctx {
    /* weights, repacked [ky][kx][ci][co] */
    float* w[XFO_NUM_LAYERS];
    float skip_w[24], skip_b[24];
    float* fus2_w; float fus2_b[64];    /* [ci][co] */
    float heat2_w[64]; float heat2_b;
    float* kp3_w; float kp3_b[65];      /* [ci][65] */
    float* bn_stat[XFO_NUM_LAYERS];     /* optional running statistics as (beta[C] = -mean * rstd, alpha[C] = rstd) */
    int bn_mode;                        /* 0 = batch statistics (the reference), 1 = running statistics */
    int rescale;                        /* 0 = the reference's Long-typed no-op (Q2), 1 = float rescale to input coordinates */
    /* intermediates of the last call */
    int H, W;                           /* after resize */
    float* t[128]; int64_t tn[128];
};

static int g_threads = 0;
void xfo_set_threads(int n) { g_threads = n; }
/* default: all cores, but at most 16 threads -- the parallel loops are rows of small maps, and a
 * 256-thread team (GPU-box host) spends its time in fork/join (measured 13 s per VGA frame) */
static int default_threads(void) {
#ifdef _OPENMP
    const int m = omp_get_max_threads();
    return m > 16 ? 16 : m;
#else
    return 1;
#endif
}
int xfo_get_threads(void) { return g_threads > 0 ? g_threads : default_threads(); }
#define NT() (g_threads > 0 ? g_threads : default_threads())

/* ------------------------------------------------------------------ weight blob */
typedef struct { char name[48]; uint32_t ndim; uint32_t dims[4]; uint64_t off; } blob_entry;

static const float* blob_find(const void* blob, size_t nbytes, const char* name, uint32_t* dims) {
    const unsigned char* p = (const unsigned char*)blob;
    if (nbytes < 16 || memcmp(p, "XFHWGT01", 8) != 0) return NULL;
    uint32_t n; memcpy(&n, p + 8, 4);
    const size_t esz = 48 + 4 + 16 + 8;
    const unsigned char* data = p + 16 + (size_t)n * esz;
    for (uint32_t i = 0; i < n; ++i) {
        const unsigned char* e = p + 16 + (size_t)i * esz;
        if (strncmp((const char*)e, name, 48) == 0) {
            uint32_t nd; memcpy(&nd, e + 48, 4);
            memcpy(dims, e + 52, 16);
            uint64_t off; memcpy(&off, e + 68, 8);
            size_t cnt = 1; for (uint32_t k = 0; k < nd; ++k) cnt *= dims[k];
            if ((size_t)(data - p) + 4 * (off + cnt) > nbytes) return NULL;
            return (const float*)(data + 4 * off);
        }
    }
    return NULL;
}

/* OIHW -> [ky][kx][ci][co] */
static float* repack(const float* w, int co, int ci, int ks) {
    float* o = (float*)malloc(sizeof(float) * (size_t)co * ci * ks * ks);
    for (int a = 0; a < co; ++a) for (int b = 0; b < ci; ++b)
        for (int y = 0; y < ks; ++y) for (int x = 0; x < ks; ++x)
            o[(((size_t)y * ks + x) * ci + b) * co + a] = w[(((size_t)a * ci + b) * ks + y) * ks + x];
    return o;
}

xfo_ctx* xfo_create(const void* blob, size_t nbytes) {
    xfo_ctx* c = (xfo_ctx*)calloc(1, sizeof(xfo_ctx));
    uint32_t d[4]; char nm[64];
    for (int i = 0; i < XFO_NUM_LAYERS; ++i) {
        snprintf(nm, sizeof nm, "%s.layer.0.weight", LAYERS[i].name);
        const float* w = blob_find(blob, nbytes, nm, d);
        if (!w || (int)d[0] != LAYERS[i].cout || (int)d[1] != LAYERS[i].cin) { xfo_destroy(c); return NULL; }
        c->w[i] = repack(w, LAYERS[i].cout, LAYERS[i].cin, LAYERS[i].ks);
        /* optional BatchNorm running statistics (eval() semantics of upstream XFeat) */
        char n2[80]; uint32_t dm[4], dv[4];
        snprintf(nm, sizeof nm, "%s.layer.1.running_mean", LAYERS[i].name);
        snprintf(n2, sizeof n2, "%s.layer.1.running_var", LAYERS[i].name);
        const float* rm = blob_find(blob, nbytes, nm, dm);
        const float* rv = blob_find(blob, nbytes, n2, dv);
        if (rm && rv && (int)dm[0] == LAYERS[i].cout && (int)dv[0] == LAYERS[i].cout) {
            c->bn_stat[i] = (float*)malloc(sizeof(float) * 2 * LAYERS[i].cout);
            for (int ch = 0; ch < LAYERS[i].cout; ++ch) {
                const float rs = (float)(1.0 / sqrt((double)rv[ch] + 1e-5));
                c->bn_stat[i][ch] = -(rm[ch] * rs);
                c->bn_stat[i][LAYERS[i].cout + ch] = rs;
            }
        }
    }
    const float* p;
    if (!(p = blob_find(blob, nbytes, "skip1.1.weight", d))) { xfo_destroy(c); return NULL; }
    memcpy(c->skip_w, p, 24 * 4);
    if (!(p = blob_find(blob, nbytes, "skip1.1.bias", d))) { xfo_destroy(c); return NULL; }
    memcpy(c->skip_b, p, 24 * 4);
    if (!(p = blob_find(blob, nbytes, "block_fusion.2.weight", d))) { xfo_destroy(c); return NULL; }
    c->fus2_w = repack(p, 64, 64, 1);
    if (!(p = blob_find(blob, nbytes, "block_fusion.2.bias", d))) { xfo_destroy(c); return NULL; }
    memcpy(c->fus2_b, p, 64 * 4);
    if (!(p = blob_find(blob, nbytes, "heatmap_head.2.weight", d))) { xfo_destroy(c); return NULL; }
    memcpy(c->heat2_w, p, 64 * 4);
    if (!(p = blob_find(blob, nbytes, "heatmap_head.2.bias", d))) { xfo_destroy(c); return NULL; }
    c->heat2_b = p[0];
    if (!(p = blob_find(blob, nbytes, "keypoint_head.3.weight", d))) { xfo_destroy(c); return NULL; }
    c->kp3_w = repack(p, 65, 64, 1);
    if (!(p = blob_find(blob, nbytes, "keypoint_head.3.bias", d))) { xfo_destroy(c); return NULL; }
    memcpy(c->kp3_b, p, 65 * 4);
    return c;
}

static void free_tensors(xfo_ctx* c) {
    for (int i = 0; i < 128; ++i) { free(c->t[i]); c->t[i] = NULL; c->tn[i] = 0; }
}
void xfo_destroy(xfo_ctx* c) {
    if (!c) return;
    for (int i = 0; i < XFO_NUM_LAYERS; ++i) { free(c->w[i]); free(c->bn_stat[i]); }
    free(c->fus2_w); free(c->kp3_w);
    free_tensors(c);
    free(c);
}
static float* talloc(xfo_ctx* c, int id, int64_t n) {
    free(c->t[id]);
    c->t[id] = (float*)calloc((size_t)(n > 0 ? n : 1), sizeof(float));
    c->tn[id] = n;
    return c->t[id];
}
int xfo_set_bn_mode(xfo_ctx* c, int mode) {
    if (mode == 1) for (int i = 0; i < XFO_NUM_LAYERS; ++i) if (!c->bn_stat[i]) return -1;
    c->bn_mode = mode;
    return 0;
}
int xfo_set_rescale(xfo_ctx* c, int on) { c->rescale = on ? 1 : 0; return 0; }
int xfo_get_tensor(xfo_ctx* c, int id, const float** ptr, int64_t* count) {
    if (id < 0 || id >= 128 || !c->t[id]) return -1;
    *ptr = c->t[id]; *count = c->tn[id];
    return 0;
}

/* ------------------------------------------------------------------ building blocks */

/* exp() as libtorch's CPU softmax and sigmoid kernels evaluate it: Vectorized<float>::exp() = Sleef_expf{8,16}_u10 in its
 * FMA form -- q = rint(d * log2(e)); Cody-Waite reduction with two fma; degree-6 Horner polynomial in fma; 1 + (s*s*u + s);
 * scaling by 2^q as two exact multiplications; 0 below -104, inf above 100.  The reference reaches it through
 * torch::sigmoid (src/XFeat.cc:82) and F::softmax (src/XFextractor.cc:207).  Probed bit for bit in this container
 * (tests/test_oracle.py::test_exp_is_atens_vector_exp): torch.sigmoid and F.softmax(dim=1) of a [1,65,h,w] tensor at one thread
 * reproduce this arithmetic in EVERY element (sequential channel sum, true division); glibc's expf does not (1 % of the elements
 * differ by an ulp).  With more threads libtorch's own chunk tails go through a scalar loop, i.e. the library is not bit-stable
 * over thread counts there; the vector form is what all but the last few pixels of a map see. */
float xfo_expf(float d) {
    const float qf = rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    const int q = (int)(qf < -300.f ? -300.f : (qf > 300.f ? 300.f : qf));      /* out-of-range arguments are overridden below */
    float s = fmaf(qf, -0.693145751953125f, d);
    s = fmaf(qf, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    union { float f; int32_t i; } a, b;
    a.i = (int32_t)((uint32_t)((q >> 1) + 127) << 23);
    b.i = (int32_t)((uint32_t)((q - (q >> 1)) + 127) << 23);
    u = u * a.f * b.f;
    if (d < -104.f) u = 0.f;
    if (d > 100.f) u = INFINITY;
    return u;
}
void xfo_expf_array(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = xfo_expf(x[i]); }

/* ATen upsample_bilinear2d, align_corners=false (used by F::interpolate at
 * src/XFextractor.cc:198-200 and src/XFeat.cc:159-165): per output index the source
 * coordinate is max(scale*(d+0.5)-0.5, 0) with scale = in/out in fp32.  The libtorch CPU
 * kernel evaluates it with one fused multiply-add and blends as fma(w0, v0, w1*v1)
 * (probed bit-exactly against F.interpolate, see tests/test_oracle_vs_torch.py). */
static void lin_coeff(int in, int out, int d, int* i0, int* i1, float* l0, float* l1) {
    const float scale = (float)in / (float)out;
    float src = fmaf(scale, (float)d + 0.5f, -0.5f);
    if (src < 0.f) src = 0.f;
    int a = (int)src;
    if (a > in - 1) a = in - 1;
    float lam = src - (float)a;
    if (lam < 0.f) lam = 0.f;
    if (lam > 1.f) lam = 1.f;
    *i0 = a; *i1 = a + ((a < in - 1) ? 1 : 0);
    *l1 = lam; *l0 = 1.f - lam;
}

/* bilinear resize of an NHWC map: row = fma(w0, p_0, w1*p_1), out = fma(h0, row0, h1*row1) */
static void resize_bilinear(const float* in, int Hi, int Wi, int C, float* out, int Ho, int Wo) {
#pragma omp parallel for num_threads(NT()) schedule(static)
    for (int y = 0; y < Ho; ++y) {
        int y0, y1; float hy0, hy1;
        lin_coeff(Hi, Ho, y, &y0, &y1, &hy0, &hy1);
        for (int x = 0; x < Wo; ++x) {
            int x0, x1; float wx0, wx1;
            lin_coeff(Wi, Wo, x, &x0, &x1, &wx0, &wx1);
            const float* p00 = in + ((size_t)y0 * Wi + x0) * C;
            const float* p01 = in + ((size_t)y0 * Wi + x1) * C;
            const float* p10 = in + ((size_t)y1 * Wi + x0) * C;
            const float* p11 = in + ((size_t)y1 * Wi + x1) * C;
            float* o = out + ((size_t)y * Wo + x) * C;
            for (int c = 0; c < C; ++c) {
                float top = fmaf(wx0, p00[c], wx1 * p01[c]);
                float bot = fmaf(wx0, p10[c], wx1 * p11[c]);
                o[c] = fmaf(hy0, top, hy1 * bot);
            }
        }
    }
}

/* batch statistics over n rows of C channels: fp64 mean and biased variance, eps 1e-5
 * (BatchNorm2d(affine=false) in training mode, src/XFeat.cc:19; InstanceNorm2d(1),
 * src/XFeat.cc:32,149).  stat[0..C) = beta = -(mean * rstd), stat[C..2C) = alpha = rstd (fp32; see the header). */
static void batch_stats(const float* x, int64_t n, int C, float* stat) {
    double* s = (double*)calloc((size_t)C * 2, sizeof(double));
    const int nt = NT();
    double* part = (double*)calloc((size_t)nt * C, sizeof(double));
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double* p = part + (size_t)tid * C;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i)
            for (int c = 0; c < C; ++c) p[c] += (double)x[i * C + c];
    }
    for (int t = 0; t < nt; ++t) for (int c = 0; c < C; ++c) s[c] += part[(size_t)t * C + c];
    for (int c = 0; c < C; ++c) s[c] /= (double)n;
    memset(part, 0, sizeof(double) * (size_t)nt * C);
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double* p = part + (size_t)tid * C;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i)
            for (int c = 0; c < C; ++c) { double d = (double)x[i * C + c] - s[c]; p[c] += d * d; }
    }
    for (int t = 0; t < nt; ++t) for (int c = 0; c < C; ++c) s[C + c] += part[(size_t)t * C + c];
    for (int c = 0; c < C; ++c) {
        double var = s[C + c] / (double)n;
        const float rs = (float)(1.0 / sqrt(var + 1e-5));
        stat[c] = -((float)s[c] * rs);
        stat[C + c] = rs;
    }
    free(part); free(s);
}

/* BN apply + ReLU: relu(fma(x, alpha, beta)), src/XFeat.cc:19-20 */
static void bn_relu(const float* raw, int64_t n, int C, const float* stat, float* act) {
#pragma omp parallel for num_threads(NT()) schedule(static)
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < C; ++c) {
            float v = fmaf(raw[i * C + c], stat[C + c], stat[c]);
            act[i * C + c] = v > 0.f ? v : 0.f;
        }
}

/* Conv2d(cin,cout,ks,stride,pad=ks/2,bias=false), src/XFeat.cc:14-18.  NHWC in/out,
 * weights [ky][kx][ci][co]; one fp32 fmaf chain per output in (ky,kx,ci) order.  Output
 * channels are processed in register blocks of 32 (4 AVX2 vectors) so the chain of every
 * output stays in a register; the arithmetic per output is unchanged. */
typedef float v8f __attribute__((vector_size(32), aligned(4)));
static void conv_nhwc(const float* in, int Hi, int Wi, int Ci, const float* w, int Co, int ks, int st,
                      float* out, int Ho, int Wo) {
    const int pad = ks / 2;
#pragma omp parallel for num_threads(NT()) schedule(static)
    for (int oy = 0; oy < Ho; ++oy) {
        for (int ox = 0; ox < Wo; ++ox) {
            float* op = out + ((size_t)oy * Wo + ox) * Co;
            int co0 = 0;
            for (; co0 + 32 <= Co; co0 += 32) {
                v8f a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
                for (int ky = 0; ky < ks; ++ky) {
                    const int iy = oy * st + ky - pad;
                    if (iy < 0 || iy >= Hi) continue;
                    for (int kx = 0; kx < ks; ++kx) {
                        const int ix = ox * st + kx - pad;
                        if (ix < 0 || ix >= Wi) continue;
                        const float* ip = in + ((size_t)iy * Wi + ix) * Ci;
                        const float* wp = w + ((size_t)(ky * ks + kx) * Ci) * Co + co0;
                        for (int ci = 0; ci < Ci; ++ci) {
                            const float v = ip[ci];
                            const v8f vv = {v, v, v, v, v, v, v, v};
                            const v8f* wr = (const v8f*)(wp + (size_t)ci * Co);
                            a0 = __builtin_ia32_vfmaddps256(vv, wr[0], a0);
                            a1 = __builtin_ia32_vfmaddps256(vv, wr[1], a1);
                            a2 = __builtin_ia32_vfmaddps256(vv, wr[2], a2);
                            a3 = __builtin_ia32_vfmaddps256(vv, wr[3], a3);
                        }
                    }
                }
                *(v8f*)(op + co0) = a0; *(v8f*)(op + co0 + 8) = a1; *(v8f*)(op + co0 + 16) = a2; *(v8f*)(op + co0 + 24) = a3;
            }
            if (co0 < Co) {                      /* tail (C_out 4, 8, 24, 65): scalar fmaf chains */
                float acc[32];
                const int nc = Co - co0;
                for (int c = 0; c < nc; ++c) acc[c] = 0.f;
                for (int ky = 0; ky < ks; ++ky) {
                    const int iy = oy * st + ky - pad;
                    if (iy < 0 || iy >= Hi) continue;
                    for (int kx = 0; kx < ks; ++kx) {
                        const int ix = ox * st + kx - pad;
                        if (ix < 0 || ix >= Wi) continue;
                        const float* ip = in + ((size_t)iy * Wi + ix) * Ci;
                        const float* wp = w + ((size_t)(ky * ks + kx) * Ci) * Co + co0;
                        for (int ci = 0; ci < Ci; ++ci) {
                            const float v = ip[ci];
                            const float* wr = wp + (size_t)ci * Co;
                            for (int c = 0; c < nc; ++c) acc[c] = fmaf(v, wr[c], acc[c]);
                        }
                    }
                }
                memcpy(op + co0, acc, sizeof(float) * nc);
            }
        }
    }
}

static int out_dim(int in, int ks, int st) { return (in + 2 * (ks / 2) - ks) / st + 1; }

/* one BasicLayer: conv -> BN(batch stats) -> ReLU; keeps the raw map and the stats */
static float* basic_layer(xfo_ctx* c, int li, const float* in, int Hi, int Wi, int* Ho, int* Wo) {
    const layer_spec* L = &LAYERS[li];
    *Ho = out_dim(Hi, L->ks, L->stride); *Wo = out_dim(Wi, L->ks, L->stride);
    const int64_t n = (int64_t)(*Ho) * (*Wo);
    float* raw = talloc(c, XFO_T_RAW0 + li, n * L->cout);
    conv_nhwc(in, Hi, Wi, L->cin, c->w[li], L->cout, L->ks, L->stride, raw, *Ho, *Wo);
    float* st = talloc(c, XFO_T_STAT0 + li, 2 * L->cout);
    if (c->bn_mode == 1 && c->bn_stat[li]) memcpy(st, c->bn_stat[li], sizeof(float) * 2 * L->cout);
    else batch_stats(raw, n, L->cout, st);
    float* act = (float*)malloc(sizeof(float) * (size_t)n * L->cout);
    bn_relu(raw, n, L->cout, st, act);
    return act;
}

/* InterpolateSparse2d::normgrid (src/XFeat.cc:181-186): Long positions divided by
 * (W-1, H-1) in fp32, then 2*g-1; followed by grid_sample(align_corners=false)'s
 * unnormalise as the ATen CPU kernel writes it: (g + 1) * (size/2) - 0.5. */
static float grid_coord(int pos, int full, int size) {
    float g = 2.0f * ((float)pos / (float)(full - 1)) - 1.0f;
    return (g + 1.0f) * ((float)size / 2.0f) - 0.5f;
}

/* grid_sample nearest, zeros padding (src/XFeat.cc:200): nearbyint (half to even) */
static float sample_nearest(const float* map, int Hm, int Wm, int x, int y, int H, int W) {
    float fx = nearbyintf(grid_coord(x, W, Wm));
    float fy = nearbyintf(grid_coord(y, H, Hm));
    if (!(fx >= 0.f && fx <= (float)(Wm - 1) && fy >= 0.f && fy <= (float)(Hm - 1))) return 0.f;
    return map[(size_t)(int)fy * Wm + (int)fx];
}

/* grid_sample bilinear, zeros padding (src/XFeat.cc:196): C channels at one position.
 * out = ((nw*v_nw + ne*v_ne) + sw*v_sw) + se*v_se, weights from floor() distances. */
static void sample_bilinear(const float* map, int Hm, int Wm, int C, int x, int y, int H, int W, float* out) {
    const float ix = grid_coord(x, W, Wm), iy = grid_coord(y, H, Hm);
    const float xw = floorf(ix), yn = floorf(iy);
    const float w = ix - xw, e = 1.0f - w, n = iy - yn, s = 1.0f - n;
    const float nw = e * s, ne = w * s, sw = e * n, se = w * n;
    const int x0 = (int)xw, y0 = (int)yn, x1 = x0 + 1, y1 = y0 + 1;
    const int vx0 = x0 >= 0 && x0 < Wm, vx1 = x1 >= 0 && x1 < Wm;
    const int vy0 = y0 >= 0 && y0 < Hm, vy1 = y1 >= 0 && y1 < Hm;
    for (int c = 0; c < C; ++c) {
        const float a = (vx0 && vy0) ? map[((size_t)y0 * Wm + x0) * C + c] : 0.f;
        const float b = (vx1 && vy0) ? map[((size_t)y0 * Wm + x1) * C + c] : 0.f;
        const float d = (vx0 && vy1) ? map[((size_t)y1 * Wm + x0) * C + c] : 0.f;
        const float g = (vx1 && vy1) ? map[((size_t)y1 * Wm + x1) * C + c] : 0.f;
        out[c] = ((a * nw + b * ne) + d * sw) + g * se;
    }
}

/* F::normalize(dim=C): x / max(||x||_2, 1e-12) */
static void l2_normalize(const float* in, int C, float* out) {
    double ss = 0.0;
    for (int c = 0; c < C; ++c) ss += (double)in[c] * (double)in[c];
    float nrm = (float)sqrt(ss);
    if (nrm < 1e-12f) nrm = 1e-12f;
    for (int c = 0; c < C; ++c) out[c] = in[c] / nrm;
}

typedef struct { float score; int idx; } cand_t;
static int cand_cmp(const void* a, const void* b) {
    const cand_t* p = (const cand_t*)a; const cand_t* q = (const cand_t*)b;
    if (p->score > q->score) return -1;
    if (p->score < q->score) return 1;
    return (p->idx > q->idx) - (p->idx < q->idx);   /* argsort(-score) is stable: index order */
}

/* ------------------------------------------------------------------ extraction */
int xfo_extract(xfo_ctx* c, const uint8_t* gray, int H0, int W0, int nfeatures, int lap0, int lap1,
                xfo_keypoint* kps, float* desc, int* n_valid, int* mono_index) {
    if (!gray || H0 <= 0 || W0 <= 0) return -1;           /* XFextractor.cc:253-254 */
    const int H = (H0 / 32) * 32, W = (W0 / 32) * 32;      /* preprocessTensor :188-191 */
    if (H < 32 || W < 32) return -2;
    free_tensors(c);
    c->H = H; c->W = W;
    const int h8 = H / 8, w8 = W / 8, h4 = H / 4, w4 = W / 4;

    /* parseInput (:161-168): u8 -> f32, true division by 255 */
    float* x0 = (float*)malloc(sizeof(float) * (size_t)H0 * W0);
    for (int64_t i = 0; i < (int64_t)H0 * W0; ++i) x0[i] = (float)gray[i] / 255.0f;
    /* preprocessTensor (:198-200): bilinear resize to multiples of 32 */
    float* x = talloc(c, XFO_T_X, (int64_t)H * W);
    if (H == H0 && W == W0) memcpy(x, x0, sizeof(float) * (size_t)H * W);
    else resize_bilinear(x0, H0, W0, 1, x, H, W);
    free(x0);

    /* XFeatModel::forward (src/XFeat.cc:135-173) */
    /* :148 mean over the single channel is the identity; :149 InstanceNorm2d(1) */
    float* xst = talloc(c, XFO_T_XSTAT, 2);
    batch_stats(x, (int64_t)H * W, 1, xst);
    float* xh = (float*)malloc(sizeof(float) * (size_t)H * W);
#pragma omp parallel for num_threads(NT()) schedule(static)
    for (int64_t i = 0; i < (int64_t)H * W; ++i) xh[i] = fmaf(x[i], xst[1], xst[0]);

    int Ho, Wo;
    /* :152 block1 */
    float* a = basic_layer(c, 0, xh, H, W, &Ho, &Wo);
    float* b = basic_layer(c, 1, a, Ho, Wo, &Ho, &Wo); free(a);
    a = basic_layer(c, 2, b, Ho, Wo, &Ho, &Wo); free(b);
    b = basic_layer(c, 3, a, Ho, Wo, &Ho, &Wo); free(a);          /* x1: [h4][w4][24] */
    /* :153 skip1 = AvgPool2d(4,4) -> Conv2d(1,24,1) with bias (:36-39); x1 + skip1(x) */
    float* pool = talloc(c, XFO_T_SKIP_POOL, (int64_t)h4 * w4);
    for (int y = 0; y < h4; ++y) for (int xx = 0; xx < w4; ++xx) {
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += xh[(size_t)(4 * y + i) * W + 4 * xx + j];
        pool[(size_t)y * w4 + xx] = s / 16.0f;
    }
    float* b2in = talloc(c, XFO_T_B2IN, (int64_t)h4 * w4 * 24);
    for (int64_t p = 0; p < (int64_t)h4 * w4; ++p)
        for (int ch = 0; ch < 24; ++ch)
            b2in[p * 24 + ch] = b[p * 24 + ch] + (fmaf(pool[p], c->skip_w[ch], 0.f) + c->skip_b[ch]);
    free(b);
    /* block2 */
    a = basic_layer(c, 4, b2in, h4, w4, &Ho, &Wo);
    b = basic_layer(c, 5, a, Ho, Wo, &Ho, &Wo); free(a);           /* x2 */
    /* :154 block3 */
    a = basic_layer(c, 6, b, Ho, Wo, &Ho, &Wo); free(b);
    b = basic_layer(c, 7, a, Ho, Wo, &Ho, &Wo); free(a);
    float* x3 = basic_layer(c, 8, b, Ho, Wo, &Ho, &Wo); free(b);   /* [h8][w8][64] */
    /* :155 block4 */
    int H4, W4, H5, W5;
    a = basic_layer(c, 9, x3, h8, w8, &H4, &W4);
    b = basic_layer(c, 10, a, H4, W4, &H4, &W4); free(a);
    float* x4 = basic_layer(c, 11, b, H4, W4, &H4, &W4); free(b);
    /* :156 block5 */
    a = basic_layer(c, 12, x4, H4, W4, &H5, &W5);
    b = basic_layer(c, 13, a, H5, W5, &H5, &W5); free(a);
    a = basic_layer(c, 14, b, H5, W5, &H5, &W5); free(b);
    float* x5 = basic_layer(c, 15, a, H5, W5, &H5, &W5); free(a);
    /* :159-166 pyramid fusion: bilinear upsample x4, x5 to x3's size, x3 + x4 + x5 */
    float* u4 = (float*)malloc(sizeof(float) * (size_t)h8 * w8 * 64);
    float* u5 = (float*)malloc(sizeof(float) * (size_t)h8 * w8 * 64);
    resize_bilinear(x4, H4, W4, 64, u4, h8, w8);
    resize_bilinear(x5, H5, W5, 64, u5, h8, w8);
    float* fin = talloc(c, XFO_T_FUSE_IN, (int64_t)h8 * w8 * 64);
    for (int64_t i = 0; i < (int64_t)h8 * w8 * 64; ++i) fin[i] = (x3[i] + u4[i]) + u5[i];
    free(u4); free(u5); free(x3); free(x4); free(x5);
    /* block_fusion (:72-76): two BasicLayers then Conv2d(64,64,1) with bias, no BN */
    a = basic_layer(c, 16, fin, h8, w8, &Ho, &Wo);
    b = basic_layer(c, 17, a, Ho, Wo, &Ho, &Wo); free(a);
    float* feats = talloc(c, XFO_T_FEATS, (int64_t)h8 * w8 * 64);
    conv_nhwc(b, h8, w8, 64, c->fus2_w, 64, 1, 1, feats, h8, w8); free(b);
    for (int64_t p = 0; p < (int64_t)h8 * w8; ++p) for (int ch = 0; ch < 64; ++ch) feats[p * 64 + ch] += c->fus2_b[ch];
    /* :169 heatmap_head (:78-83): 2 BasicLayers 1x1, Conv2d(64,1,1)+bias, Sigmoid */
    a = basic_layer(c, 18, feats, h8, w8, &Ho, &Wo);
    b = basic_layer(c, 19, a, Ho, Wo, &Ho, &Wo); free(a);
    float* H1 = talloc(c, XFO_T_H1, (int64_t)h8 * w8);
    for (int64_t p = 0; p < (int64_t)h8 * w8; ++p) {
        float acc = 0.f;
        for (int ch = 0; ch < 64; ++ch) acc = fmaf(b[p * 64 + ch], c->heat2_w[ch], acc);
        acc += c->heat2_b;
        H1[p] = 1.0f / (1.0f + xfo_expf(0.f - acc));
    }
    free(b);
    /* :170 keypoint_head(unfold2d(x, 8)); unfold2d (:124-133): channel = i*8 + j with i the
     * row inside the 8x8 cell */
    float* xu = talloc(c, XFO_T_XUNFOLD, (int64_t)h8 * w8 * 64);
    for (int y = 0; y < h8; ++y) for (int xx = 0; xx < w8; ++xx)
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j)
            xu[((size_t)y * w8 + xx) * 64 + i * 8 + j] = xh[(size_t)(8 * y + i) * W + 8 * xx + j];
    a = basic_layer(c, 20, xu, h8, w8, &Ho, &Wo);
    b = basic_layer(c, 21, a, Ho, Wo, &Ho, &Wo); free(a);
    a = basic_layer(c, 22, b, Ho, Wo, &Ho, &Wo); free(b);
    float* logits = talloc(c, XFO_T_LOGITS, (int64_t)h8 * w8 * 65);
    conv_nhwc(a, h8, w8, 64, c->kp3_w, 65, 1, 1, logits, h8, w8); free(a);
    for (int64_t p = 0; p < (int64_t)h8 * w8; ++p) for (int ch = 0; ch < 65; ++ch) logits[p * 65 + ch] += c->kp3_b[ch];
    free(xh);

    /* XFextractor::operator() continues (src/XFextractor.cc:273): M1 = normalize(M1, dim=1) */
    float* m1n = talloc(c, XFO_T_M1N, (int64_t)h8 * w8 * 64);
    for (int64_t p = 0; p < (int64_t)h8 * w8; ++p) l2_normalize(feats + p * 64, 64, m1n + p * 64);

    /* getKptsHeatmap (:204-217): softmax over 65 logits (temperature 1), drop the dustbin,
     * depth-to-space: K1h[8h+i][8w+j] = p[i*8+j] */
    float* K1h = talloc(c, XFO_T_K1H, (int64_t)H * W);
#pragma omp parallel for num_threads(NT()) schedule(static)
    for (int y = 0; y < h8; ++y) for (int xx = 0; xx < w8; ++xx) {
        const float* lg = logits + ((size_t)y * w8 + xx) * 65;
        float mx = lg[0];
        for (int k = 1; k < 65; ++k) if (lg[k] > mx) mx = lg[k];
        float e[65]; float sum = 0.f;
        for (int k = 0; k < 65; ++k) { e[k] = xfo_expf(lg[k] - mx); sum += e[k]; }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j)
            K1h[(size_t)(8 * y + i) * W + 8 * xx + j] = e[i * 8 + j] / sum;
    }

    /* NMS (:219-248): 5x5 max-pool (stride 1, pad 2, -inf), pos = (x == max) & (x > 0.05),
     * nonzero() in row-major order, flipped to (x, y) */
    const float thr = 0.05f;
    cand_t* cand = (cand_t*)malloc(sizeof(cand_t) * (size_t)H * W);
    int C = 0;
    for (int y = 0; y < H; ++y) for (int xx = 0; xx < W; ++xx) {
        const float v = K1h[(size_t)y * W + xx];
        if (!(v > thr)) continue;
        float m = -INFINITY;
        for (int dy = -2; dy <= 2; ++dy) { const int yy = y + dy; if (yy < 0 || yy >= H) continue;
            for (int dx = -2; dx <= 2; ++dx) { const int x2 = xx + dx; if (x2 < 0 || x2 >= W) continue;
                const float t = K1h[(size_t)yy * W + x2]; if (t > m) m = t; } }
        if (v == m) { cand[C].idx = y * W + xx; cand[C].score = 0.f; ++C; }
    }
    /* scores (:280-282): nearest(K1h) * bilinear(H1) at the keypoints; (0,0) -> -1 */
    float* candt = talloc(c, XFO_T_CAND, (int64_t)C * 3);
    for (int i = 0; i < C; ++i) {
        const int xx = cand[i].idx % W, y = cand[i].idx / W;
        float hb;
        sample_bilinear(H1, h8, w8, 1, xx, y, H, W, &hb);
        float s = sample_nearest(K1h, H, W, xx, y, H, W) * hb;
        if (xx == 0 && y == 0) s = -1.0f;
        cand[i].score = s;
        candt[i * 3 + 0] = (float)xx; candt[i * 3 + 1] = (float)y; candt[i * 3 + 2] = s;
    }
    /* top-k (:285-295): argsort(-scores) ascending (stable), first nfeatures */
    qsort(cand, (size_t)C, sizeof(cand_t), cand_cmp);
    const int N = C < nfeatures ? C : nfeatures;
    float* sel = talloc(c, XFO_T_SEL, (int64_t)N * 3);
    for (int i = 0; i < N; ++i) {
        sel[i * 3 + 0] = (float)(cand[i].idx % W); sel[i * 3 + 1] = (float)(cand[i].idx / W); sel[i * 3 + 2] = cand[i].score;
    }

    /* pack (:304-356).  The (Long) rescale at :304-305 multiplies by trunc(rw)=trunc(rh)=1
     *  and is a no-op.  Output vectors are nfeatures long, default
     * cv::KeyPoint() and zero descriptor rows where nothing is written. */
    for (int i = 0; i < nfeatures; ++i) {
        kps[i].x = 0.f; kps[i].y = 0.f; kps[i].size = 0.f; kps[i].angle = -1.f; kps[i].response = 0.f;
        kps[i].octave = 0; kps[i].class_id = -1;
    }
    memset(desc, 0, sizeof(float) * (size_t)nfeatures * 64);
    int mono = 0, stereo = nfeatures - 1, nv = 0;
    for (int i = 0; i < N; ++i) {
        if (!(cand[i].score > 0.f)) continue;                     /* valid = scores > 0 (:313) */
        const int xx = cand[i].idx % W, y = cand[i].idx / W;
        float d[64], dn[64];
        /* :298-301 bilinear sample of the normalised map, then L2 normalise */
        sample_bilinear(m1n, h8, w8, 64, xx, y, H, W, d);
        l2_normalize(d, 64, dn);
        int slot;
        /* rescale mode 1 (not the reference): mkpts.float() * (rw, rh) as upstream XFeat does, rw = W0 / W in fp32 */
        const float kx = c->rescale ? (float)xx * (float)((double)W0 / (double)W) : (float)xx;
        const float ky = c->rescale ? (float)y * (float)((double)H0 / (double)H) : (float)y;
        if (kx >= (float)lap0 && kx <= (float)lap1) slot = stereo--; else slot = mono++;   /* :332-343 */
        kps[slot].x = kx; kps[slot].y = ky; kps[slot].size = 1.f; kps[slot].angle = -1.f;
        kps[slot].response = cand[i].score; kps[slot].octave = 0; kps[slot].class_id = -1;
        memcpy(desc + (size_t)slot * 64, dn, sizeof dn);
        ++nv;
    }
    free(cand);
    *n_valid = nv; *mono_index = mono;
    return 0;
}

/* ------------------------------------------------------------------ matching */
int xfo_descriptor_distance(const float* a, const float* b) {
    /* ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:2246-2247:
     *   float normDist = cv::norm(a, b, cv::NORM_L2SQR); return (int)(normDist * 512);
     * cv::norm (OpenCV 4.5.4, modules/core/src/norm.cpp, normDiffL2Sqr_<float,double>)
     * takes the fp32 difference and accumulates its square in fp64. */
    double s = 0.0;
    for (int k = 0; k < 64; ++k) { const float d = a[k] - b[k]; s += (double)d * (double)d; }
    const float nd = (float)s;
    return (int)(nd * 512);
}

int xfo_distance_i32(const float* d1, int n1, const float* d2, int n2, int32_t* out) {
#pragma omp parallel for num_threads(NT()) schedule(static)
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n2; ++j)
            out[(size_t)i * n2 + j] = xfo_descriptor_distance(d1 + (size_t)i * 64, d2 + (size_t)j * 64);
    return 0;
}

/* inner loop of ORBmatcher::SearchByProjection (src/ORBmatcher.cc:75-119; same shape at :450-500,
 * :1928-1953, ...): candidates visited in list order, strict '<' against best and second, both
 * starting at init_dist (256 in the reference), best index starting at -1. */
int xfo_best2_csr(const float* q, int nq, const float* tg, const int* offsets, const int* indices, int init_dist,
                  int* best_idx, int* best_dist, int* second_idx, int* second_dist) {
    for (int i = 0; i < nq; ++i) {
        int bestDist = init_dist, bestDist2 = init_dist, bestIdx = -1, idx2 = -1;
        for (int p = offsets[i]; p < offsets[i + 1]; ++p) {
            const int idx = indices[p];
            const int dist = xfo_descriptor_distance(q + (size_t)i * 64, tg + (size_t)idx * 64);
            if (dist < bestDist) { bestDist2 = bestDist; idx2 = bestIdx; bestDist = dist; bestIdx = idx; }
            else if (dist < bestDist2) { bestDist2 = dist; idx2 = idx; }
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist; second_idx[i] = idx2; second_dist[i] = bestDist2;
    }
    return 0;
}

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:329-403), one call per group: group g observes the
 * descriptor rows indices[offsets[g] .. offsets[g+1]) of `table`.  :372-385 pairwise DescriptorDistance (diagonal 0),
 * :388-401 per row: sort, median = sorted[0.5*(N-1)] (truncated), strict '<' keeps the first row with the least
 * median.  Empty group: the reference returns without touching mDescriptor (:383-384) -> -1 / INT_MAX here. */
static int cmp_int(const void* a, const void* b) { const int x = *(const int*)a, y = *(const int*)b; return (x > y) - (x < y); }
int xfo_distinctive_csr(const float* table, const int* offsets, const int* indices, int n_groups, int* best_pos, int* best_median) {
    for (int g = 0; g < n_groups; ++g) {
        const int beg = offsets[g], N = offsets[g + 1] - beg;
        best_pos[g] = -1; best_median[g] = 0x7fffffff;
        if (N <= 0) continue;
        int* D = (int*)malloc(sizeof(int) * (size_t)N * N);
        int* v = (int*)malloc(sizeof(int) * (size_t)N);
        for (int i = 0; i < N; ++i) {
            D[(size_t)i * N + i] = 0;
            for (int j = i + 1; j < N; ++j) {
                const int d = xfo_descriptor_distance(table + (size_t)indices[beg + i] * 64, table + (size_t)indices[beg + j] * 64);
                D[(size_t)i * N + j] = d; D[(size_t)j * N + i] = d;
            }
        }
        int BestMedian = 0x7fffffff, BestIdx = 0;
        for (int i = 0; i < N; ++i) {
            memcpy(v, D + (size_t)i * N, sizeof(int) * (size_t)N);
            qsort(v, (size_t)N, sizeof(int), cmp_int);
            const int median = v[(size_t)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = i; }
        }
        best_pos[g] = BestIdx; best_median[g] = BestMedian;
        free(D); free(v);
    }
    return 0;
}

int xfo_match_mnn(const float* d1, int n1, const float* d2, int n2, float min_cossim,
                  int* idx1, int* idx2, float* dist, int* n_matches) {
    /* src/ORBmatcher.cc:358-359 normalise rows */
    float* f1 = (float*)malloc(sizeof(float) * (size_t)n1 * 64);
    float* f2t = (float*)malloc(sizeof(float) * (size_t)n2 * 64);   /* [k][j] */
    for (int i = 0; i < n1; ++i) l2_normalize(d1 + (size_t)i * 64, 64, f1 + (size_t)i * 64);
    for (int j = 0; j < n2; ++j) {
        float t[64]; l2_normalize(d2 + (size_t)j * 64, 64, t);
        for (int k = 0; k < 64; ++k) f2t[(size_t)k * n2 + j] = t[k];
    }
    /* :363-368 cossim = f1 f2^T; match12 = row arg-max, match21 = column arg-max (the
     * reference forms f2 f1^T separately; same numbers).  First maximum wins ties. */
    int* m12 = (int*)malloc(sizeof(int) * (size_t)n1);
    float* v12 = (float*)malloc(sizeof(float) * (size_t)n1);
    const int nt = NT();
    float* colv = (float*)malloc(sizeof(float) * (size_t)nt * n2);
    int* coli = (int*)malloc(sizeof(int) * (size_t)nt * n2);
    for (size_t q = 0; q < (size_t)nt * n2; ++q) { colv[q] = -INFINITY; coli[q] = 0x7fffffff; }
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        float* row = (float*)malloc(sizeof(float) * (size_t)n2);
        float* cv = colv + (size_t)tid * n2; int* ci = coli + (size_t)tid * n2;
#pragma omp for schedule(static)
        for (int i = 0; i < n1; ++i) {
            for (int j = 0; j < n2; ++j) row[j] = 0.f;
            for (int k = 0; k < 64; ++k) {
                const float av = f1[(size_t)i * 64 + k];
                const float* br = f2t + (size_t)k * n2;
                for (int j = 0; j < n2; ++j) row[j] = fmaf(av, br[j], row[j]);
            }
            int bj = 0; float bv = n2 > 0 ? row[0] : 0.f;
            for (int j = 1; j < n2; ++j) if (row[j] > bv) { bv = row[j]; bj = j; }
            m12[i] = bj; v12[i] = bv;
            for (int j = 0; j < n2; ++j) if (row[j] > cv[j]) { cv[j] = row[j]; ci[j] = i; }
        }
        free(row);
    }
    int* m21 = (int*)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    for (int j = 0; j < n2; ++j) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int t = 0; t < nt; ++t) {
            const float v = colv[(size_t)t * n2 + j]; const int i = coli[(size_t)t * n2 + j];
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
        m21[j] = bi;
    }
    /* :371-403 mutual check, optional min_cossim gate, DMatch(i, j, sqrt(2 (1 - cos))) */
    int n = 0;
    for (int i = 0; i < n1 && n2 > 0; ++i) {
        const int j = m12[i];
        if (m21[j] != i) continue;
        if (min_cossim > 0.f && !(v12[i] > min_cossim)) continue;
        idx1[n] = i; idx2[n] = j;
        const float cd = 1.0f - v12[i];
        dist[n] = sqrtf(2 * cd);
        ++n;
    }
    *n_matches = n;
    free(f1); free(f2t); free(m12); free(v12); free(colv); free(coli); free(m21);
    return 0;
}
