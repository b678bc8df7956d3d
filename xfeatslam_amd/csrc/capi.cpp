// capi.cpp -- implementation of include/xfeat_hip.h: context, weights, entry points.
// Built with hipcc for gfx950 only.  There is no CPU path: without a HIP device every
// compute entry point fails with XFH_ERR_NO_DEVICE / XFH_ERR_HIP.
#include "ctx.h"
#include "mnn_seg_plan.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

int conv_layer_npart(int li, int Hout, int Wout);

// ---- tracing / verbosity switches (ctx.h) ---------------------------------------------------------------------------------
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
Roctx* roctx() {
    static Roctx* R = []() -> Roctx* {
        const char* e = getenv("XFH_ROCTX");
        if (!e || !*e || *e == '0') return nullptr;
        static Roctx r;
        for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            *(void**)(&r.push) = dlsym(h, "roctxRangePushA");
            *(void**)(&r.pop) = dlsym(h, "roctxRangePop");
            if (r.push && r.pop) return &r;
        }
        return nullptr;
    }();
    return R;
}
}  // namespace
void xfh_trace_push(const char* name) { if (Roctx* r = roctx()) r->push(name); }
void xfh_trace_pop() { if (Roctx* r = roctx()) r->pop(); }
bool xfh_verbose() { static const bool v = []() { const char* e = getenv("XFH_VERBOSE"); return e && *e && *e != '0'; }(); return v; }

#define HIPCK(c, x) do { hipError_t _e = (x); if (_e != hipSuccess) { (c)->hip_err = std::string(#x) + ": " + hipGetErrorString(_e); return XFH_ERR_HIP; } } while (0)

extern "C" {

// "xfeat_hip 0.1 (gfx950); built with <clang version of the hipcc that compiled this file>, HIP <HIP_VERSION of the headers>; runtime HIP <hipRuntimeGetVersion>,
// driver <hipDriverGetVersion>": the library carries hand-counted MFMA hazard padding (common.h: XFH_MFMA_SETTLE), so the compiler it was built with and
// the runtime it meets are part of its identity (bench.py prints the string, tests/test_gpu_hazard.py checks the padding with the box's own compiler)
const char* xfh_version(void) {
    static const std::string v = [] {                // (a function-local static: initialised once, also under concurrent first calls)
        int rt = 0, drv = 0;
        if (hipRuntimeGetVersion(&rt) != hipSuccess) rt = 0;
        if (hipDriverGetVersion(&drv) != hipSuccess) drv = 0;
        char buf[320];
        snprintf(buf, sizeof buf, "xfeat_hip 0.1 (gfx950); built with clang %d.%d.%d, HIP %d.%d.%d; runtime HIP %d, driver %d",
                 __clang_major__, __clang_minor__, __clang_patchlevel__, HIP_VERSION_MAJOR, HIP_VERSION_MINOR, HIP_VERSION_PATCH, rt, drv);
        return std::string(buf);
    }();
    return v.c_str();
}

const char* xfh_strerror(int s) {
    switch (s) {
        case XFH_OK: return "ok";
        case XFH_ERR_INVALID_ARG: return "invalid argument";
        case XFH_ERR_EMPTY_IMAGE: return "empty image";
        case XFH_ERR_BAD_SIZE: return "image size unsupported (need >= 32x32 and <= ctx maximum)";
        case XFH_ERR_NO_WEIGHTS: return "weights not loaded";
        case XFH_ERR_BAD_WEIGHTS: return "malformed weight blob";
        case XFH_ERR_HIP: return "HIP runtime error";
        case XFH_ERR_NO_DEVICE: return "no HIP device";
        case XFH_ERR_OUT_OF_MEMORY: return "out of memory";
        case XFH_ERR_BATCH_TOO_LARGE: return "batch larger than ctx max_batch";
        case XFH_ERR_IO: return "file i/o error";
        case XFH_ERR_COMM: return "RCCL error";
        default: return "unknown status";
    }
}

const char* xfh_kernel_name(int id) {
    static const char* n[XFH_K_COUNT] = {"none", "k_mnn_gemm", "k_conv_mfma", "k_conv_direct", "k_nms_score", "k_select",
                                         "k_desc", "k_heads_kp", "k_dist_i32", "k_preproc", "k_best2_csr", "k_distinctive_csr", "k_mnn_gemm_seg"};
    return (id >= 0 && id < XFH_K_COUNT) ? n[id] : "?";
}

int xfh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void xfh_config_default(xfh_config* cfg) {
    memset(cfg, 0, sizeof(*cfg));
    cfg->device = 0; cfg->max_height = 480; cfg->max_width = 640; cfg->nfeatures = 4096;
    cfg->max_batch = 1; cfg->bn_mode = XFH_BN_BATCH_STATS; cfg->nms_threshold = 0.05f;
}

size_t xfh_record_kps_offset(void) { return sizeof(RecordHeader); }
size_t xfh_record_desc_offset(int nf) { return (sizeof(RecordHeader) + (size_t)nf * sizeof(xfh_keypoint) + 255) & ~(size_t)255; }
size_t xfh_record_bytes(int nf) { return (xfh_record_desc_offset(nf) + (size_t)nf * 64 * sizeof(float) + 255) & ~(size_t)255; }

static int out_dim(int in, int ks, int st) { return (in + 2 * (ks / 2) - ks) / st + 1; }

static void layer_dims(int H, int W, int* lh, int* lw) {
    // input of layer i: see run_extract
    static const int src[XFH_NUM_LAYERS] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 8, 16, 17, 18, -2, 20, 21};
    for (int i = 0; i < XFH_NUM_LAYERS; ++i) {
        int hi, wi;
        if (src[i] == -1) { hi = H; wi = W; }
        else if (src[i] == -2) { hi = H / 8; wi = W / 8; }
        else { hi = lh[src[i]]; wi = lw[src[i]]; }
        lh[i] = out_dim(hi, XFH_LAYERS[i].ks, XFH_LAYERS[i].stride);
        lw[i] = out_dim(wi, XFH_LAYERS[i].ks, XFH_LAYERS[i].stride);
    }
}

int xfh_create(const xfh_config* cfg, xfh_ctx** out) {
    if (!cfg || !out) return XFH_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->max_height < 32 || cfg->max_width < 32 || cfg->nfeatures < 1 || cfg->nfeatures > 65536 || cfg->max_batch < 1)
        return XFH_ERR_INVALID_ARG;
    if (cfg->bn_mode != XFH_BN_BATCH_STATS && cfg->bn_mode != XFH_BN_RUNNING_STATS && cfg->bn_mode != XFH_BN_RUNNING_FOLDED) return XFH_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return XFH_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return XFH_ERR_NO_DEVICE;
    int num_cu = 0;
    {   // the code objects in this library are gfx950 only
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) return XFH_ERR_NO_DEVICE;
        num_cu = prop.multiProcessorCount;
    }
    xfh_ctx* c = new xfh_ctx();
    c->cfg = *cfg;
    if (num_cu > 0) c->num_cu = num_cu;
    if (c->cfg.nms_threshold <= 0.f) c->cfg.nms_threshold = 0.05f;
    c->Hmax = (cfg->max_height / 32) * 32; c->Wmax = (cfg->max_width / 32) * 32;
#ifdef XFH_TEST_KNOBS
    // DEBUG BUILD ONLY (make knobs -> libxfeat_hip_knobs.so, loaded by tests/workers/knob_worker.py and tools/flake_hunt.sh): kernel forms that the
    // shipped library selects by batch size alone can be forced here, to bisect a box-dependent failure.  The default build has one path.
    if (const char* e = getenv("XFH_SELECT_LEGACY")) c->select_legacy = e[0] == '1';      // k_select's fallback form for every frame
    if (const char* e = getenv("XFH_NO_NMS_HEAT")) c->no_nms_heat = e[0] == '1';          // k_heads_heat as a launch of its own for every batch size
    if (const char* e = getenv("XFH_NO_RIDE")) c->no_ride = e[0] == '1';                  // the keypoint branch on the second stream for every batch size
#endif
    const int B = cfg->max_batch;
    int rc = XFH_OK;
    auto fail = [&](int code) { xfh_destroy(c); return code; };
#define A(ptr, bytes) do { if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) return fail(XFH_ERR_OUT_OF_MEMORY); } while (0)
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(XFH_ERR_HIP);
    {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool masked = false;
#ifdef XFH_TEST_KNOBS
        // XFH_CU_MASKS="lo-hi,lo-hi,..." (measurement knob of the debug build, tools/cu_mask_ab.sh): the k-th ctx created in this process runs its
        // streams on the CUs [lo, hi] of entry k mod n only (hipExtStreamCreateWithCUMask).  Unset: all CUs.
        static std::atomic<int> created{0};
        if (const char* e = getenv("XFH_CU_MASKS")) {
            std::vector<std::pair<int, int>> rg;
            for (const char* p = e; *p;) {
                int lo = 0, hi = 0, n = 0;
                if (sscanf(p, "%d-%d%n", &lo, &hi, &n) == 2 && lo >= 0 && hi >= lo && hi < 256) rg.push_back({lo, hi});
                p += n > 0 ? n : 1;
                while (*p == ',') ++p;
            }
            if (!rg.empty()) {
                const auto r = rg[(size_t)created.fetch_add(1) % rg.size()];
                for (int cu = r.first; cu <= r.second; ++cu) mask[cu >> 5] |= 1u << (cu & 31);
                masked = true;
            }
        }
#endif
        if (masked) {
            if (hipExtStreamCreateWithCUMask(&c->own_stream, 8, mask) != hipSuccess) return fail(XFH_ERR_HIP);
            if (hipExtStreamCreateWithCUMask(&c->aux_stream, 8, mask) != hipSuccess) return fail(XFH_ERR_HIP);
        } else {
            if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(XFH_ERR_HIP);
            if (hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess) return fail(XFH_ERR_HIP);
        }
    }
    c->stream = c->own_stream;
    if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) return fail(XFH_ERR_HIP);
    if (hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) return fail(XFH_ERR_HIP);
    if (hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) return fail(XFH_ERR_HIP);
    const size_t xs = (size_t)c->Hmax * c->Wmax;
    A(c->d_gray, (size_t)B * cfg->max_height * cfg->max_width);
    A(c->X, sizeof(float) * B * xs);
    c->pre_npart = (int)((xs + 1023) / 1024);
    A(c->pre_part, sizeof(double) * B * c->pre_npart * 2);
    A(c->xstat, sizeof(float) * B * 2);
    int lh[XFH_NUM_LAYERS], lw[XFH_NUM_LAYERS];
    layer_dims(c->Hmax, c->Wmax, lh, lw);
    for (int i = 0; i < XFH_NUM_LAYERS; ++i) {
        const int C = XFH_LAYERS[i].cout;
        c->raw_stride[i] = (size_t)lh[i] * lw[i] * C;
        if (i != 0) A(c->raw[i], sizeof(float) * B * c->raw_stride[i]);        // block1.0 is never materialised (k_block1_stats / PRO_L0)
        c->part_stride[i] = (size_t)conv_layer_npart(i, lh[i], lw[i]) * C * 2;
        A(c->part[i], sizeof(double) * B * c->part_stride[i]);
        A(c->stat[i], sizeof(float) * B * 2 * C);
    }
    A(c->skip_pool, sizeof(float) * B * (xs / 16));
    A(c->feats, sizeof(float) * B * c->raw_stride[17]);
    if (B > 32) { A(c->act4, sizeof(float) * B * c->raw_stride[11]); A(c->act5, sizeof(float) * B * c->raw_stride[15]); }
    A(c->feat_nrm, sizeof(float) * B * (xs / 64));
    A(c->H1, sizeof(float) * B * (xs / 64));
    A(c->K1h, sizeof(float) * B * xs);
    c->cand_cap = 1024;
    while (c->cand_cap < xs) c->cand_cap <<= 1;
    A(c->cand, sizeof(u64) * B * c->cand_cap);
    A(c->cand_count, sizeof(int) * B * CAND_CNT_STRIDE);
    A(c->slot_src, sizeof(int) * B * cfg->nfeatures);
    A(c->sel_key, sizeof(u64) * B * cfg->nfeatures);
    A(c->sel_n, sizeof(int) * B);
    const size_t rec = xfh_record_bytes(cfg->nfeatures);
    A(c->d_records, rec * B);
    // pinned slot 0 of the submit / collect ring: ONE frame and ONE record (the batch pipeline copies between the caller's
    // buffers and HBM directly, pipeline.cpp)
    if (hipHostMalloc((void**)&c->h_records, rec, hipHostMallocDefault) != hipSuccess) return fail(XFH_ERR_OUT_OF_MEMORY);
    if (hipHostMalloc((void**)&c->h_gray, (size_t)cfg->max_height * cfg->max_width, hipHostMallocDefault) != hipSuccess) return fail(XFH_ERR_OUT_OF_MEMORY);
    c->s_hgray[0] = c->h_gray; c->s_dgray[0] = c->d_gray; c->s_hrec[0] = c->h_records;
    for (int k = 1; k < xfh_ctx::SLOTS; ++k) {
        A(c->s_dgray[k], (size_t)cfg->max_height * cfg->max_width);
        if (hipHostMalloc((void**)&c->s_hgray[k], (size_t)cfg->max_height * cfg->max_width, hipHostMallocDefault) != hipSuccess) return fail(XFH_ERR_OUT_OF_MEMORY);
        if (hipHostMalloc((void**)&c->s_hrec[k], rec, hipHostMallocDefault) != hipSuccess) return fail(XFH_ERR_OUT_OF_MEMORY);
    }
    for (int k = 0; k < xfh_ctx::SLOTS; ++k)
        if (hipEventCreateWithFlags(&c->s_done[k], hipEventDisableTiming) != hipSuccess) return fail(XFH_ERR_HIP);
#undef A
    (void)rc;
    // matcher workspace for frame-against-frame calls and the pinned output mirror of xfh_match_mnn: no allocation on the call path
    if (match_ws_reserve(c, cfg->nfeatures, cfg->nfeatures) != hipSuccess) return fail(XFH_ERR_OUT_OF_MEMORY);
    {
        MatchWs& w = c->mws;
        w.cap_out = (size_t)cfg->nfeatures * 12 + 256;
        if (hipMalloc((void**)&w.o_buf, w.cap_out) != hipSuccess) return fail(XFH_ERR_OUT_OF_MEMORY);
        if (hipHostMalloc((void**)&w.h_out, w.cap_out, hipHostMallocDefault) != hipSuccess) return fail(XFH_ERR_OUT_OF_MEMORY);
        w.cap_hout = w.cap_out;
        w.cap_in = 2 * (((size_t)cfg->nfeatures * 256 + 255) & ~(size_t)255);
        if (hipMalloc((void**)&w.h_d1, w.cap_in) != hipSuccess) return fail(XFH_ERR_OUT_OF_MEMORY);
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(XFH_ERR_HIP);
    if (xfh_verbose()) fprintf(stderr, "[xfh] ctx %p: device %d, %dx%d (x32: %dx%d), nfeatures %d, max_batch %d, bn_mode %d, flags %d\n", (void*)c, cfg->device,
                               cfg->max_height, cfg->max_width, c->Hmax, c->Wmax, cfg->nfeatures, cfg->max_batch, cfg->bn_mode, cfg->flags);
    *out = c;
    return XFH_OK;
}

int xfh_destroy(xfh_ctx* c) {
    if (!c) return XFH_OK;
    hipSetDevice(c->cfg.device);
    if (c->twin) { xfh_destroy(c->twin); c->twin = nullptr; }
    pipe_destroy(c);
    xfh_comm_destroy(c);
    if (c->stream && c->stream != c->own_stream) hipStreamSynchronize(c->stream);     // work queued on a caller's stream
    if (c->own_stream) hipStreamSynchronize(c->own_stream);
    if (c->aux_stream) hipStreamSynchronize(c->aux_stream);
    auto F = [](void* p) { if (p) hipFree(p); };
    F(c->d_gray); F(c->X); F(c->pre_part); F(c->xstat);
    for (int i = 0; i < XFH_NUM_LAYERS; ++i) { F(c->raw[i]); F(c->part[i]); F(c->stat[i]); }
    if (!c->is_twin && !c->is_lane) {                    // a twin / pipeline lane borrows the weights of its parent
        for (int i = 0; i < XFH_NUM_LAYERS; ++i) { F(c->w.mfma[i]); F(c->w.alt[i]); F(c->w.alt2[i]); F(c->w.m16[i]); F(c->w.bn_bias[i]); }
        for (int i = 0; i < 3; ++i) F(c->w.direct[i]);
        F(c->w.fus2); F(c->w.fus2_bias); F(c->w.skip_w); F(c->w.skip_b); F(c->w.heat2_w); F(c->w.heat2_b); F(c->w.kp3_w); F(c->w.kp3_b);
    }
    F(c->skip_pool); F(c->feats); F(c->act4); F(c->act5); F(c->feat_nrm); F(c->H1); F(c->K1h);
    F(c->cand); F(c->cand_count); F(c->slot_src); F(c->sel_key); F(c->sel_n); F(c->d_records);
    if (c->h_records) hipHostFree(c->h_records);
    if (c->h_gray) hipHostFree(c->h_gray);
    for (int k = 1; k < xfh_ctx::SLOTS; ++k) { F(c->s_dgray[k]); if (c->s_hgray[k]) hipHostFree(c->s_hgray[k]); if (c->s_hrec[k]) hipHostFree(c->s_hrec[k]); }
    for (int k = 0; k < xfh_ctx::SLOTS; ++k) if (c->s_done[k]) hipEventDestroy(c->s_done[k]);
    MatchWs& w = c->mws;
    F(w.img1); F(w.keys); F(w.b2_buf); F(w.h_d1); F(w.o_buf); F(w.o_tab); F(w.bkeys);
    if (w.h_out) hipHostFree(w.h_out);
    if (c->timer.ev) { for (int i = 0; i < 2 * KTimer::MAXEV; ++i) if (c->timer.ev[i]) hipEventDestroy(c->timer.ev[i]); free(c->timer.ev); }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->ev_out) hipEventDestroy(c->ev_out);
    if (c->aux_stream) hipStreamDestroy(c->aux_stream);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
    return XFH_OK;
}

// ------------------------------------------------------------------------- weights
struct BlobEntry { const float* p; uint32_t dims[4]; uint32_t ndim; };
static bool blob_find(const void* blob, size_t nbytes, const char* name, BlobEntry* e) {
    const unsigned char* p = (const unsigned char*)blob;
    if (nbytes < 16 || memcmp(p, "XFHWGT01", 8) != 0) return false;
    uint32_t n; memcpy(&n, p + 8, 4);
    const size_t esz = 48 + 4 + 16 + 8;
    if (16 + (size_t)n * esz > nbytes) return false;
    const unsigned char* data = p + 16 + (size_t)n * esz;
    for (uint32_t i = 0; i < n; ++i) {
        const unsigned char* q = p + 16 + (size_t)i * esz;
        if (strncmp((const char*)q, name, 48) == 0) {
            memcpy(&e->ndim, q + 48, 4); memcpy(e->dims, q + 52, 16);
            uint64_t off; memcpy(&off, q + 68, 8);
            if (e->ndim > 4) return false;
            const uint64_t total = (nbytes - (size_t)(data - p)) / 4;            // floats in the data section
            uint64_t cnt = 1;
            for (uint32_t k = 0; k < e->ndim; ++k) { if (e->dims[k] != 0 && cnt > total / e->dims[k]) return false; cnt *= e->dims[k]; }
            if (off > total || cnt > total - off) return false;                  // overflow-safe: off + cnt <= total
            e->p = (const float*)(data + 4 * off);
            return true;
        }
    }
    return false;
}

static int upload(xfh_ctx* c, float** dst, const std::vector<float>& v) {
    if (*dst) { hipFree(*dst); *dst = nullptr; }
    HIPCK(c, hipMalloc((void**)dst, v.size() * sizeof(float)));
    HIPCK(c, hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return XFH_OK;
}

// OIHW -> [chunk][n (COUTP)][KC] with the k permutation of the MFMA kernels: inside each group of 8
// channels, channel e sits at position 4*(e&1) + (e>>1).  A chunk holds CB = min(cin, cbmax) channels of
// tpc consecutive taps (tpc > 1 only when CB == cin): chunk = (tap / tpc) * NCB + cb, KC = tpc * CB,
// position inside the chunk row = (tap % tpc) * CB + pos.
static std::vector<float> pack_mfma(const float* w, int cout, int cin, int ks, int coutp, int cbmax = 64, int tpc = 1, bool perm16 = false) {
    const int CB = cin > cbmax ? cbmax : cin, NCB = cin / CB, KC = tpc * CB;
    std::vector<float> o((size_t)ks * ks * NCB * coutp * CB, 0.f);
    for (int ky = 0; ky < ks; ++ky) for (int kx = 0; kx < ks; ++kx) for (int cb = 0; cb < NCB; ++cb)
        for (int n = 0; n < cout; ++n) for (int lc = 0; lc < CB; ++lc) {
            // 32x32x2 kernels: k = 8g + e at 8g + 4(e & 1) + (e >> 1); 16x16x4 (k_conv_mfma16): k = 16g + e at 16g + 4(e & 3) + (e >> 2)
            const int g = lc / 8, e = lc % 8, pos = perm16 ? (lc / 16) * 16 + 4 * (lc % 16 & 3) + (lc % 16 >> 2) : g * 8 + 4 * (e & 1) + (e >> 1);
            const int ci = cb * CB + lc, tap = ky * ks + kx;
            const size_t chunk = (size_t)(tap / tpc) * NCB + cb;
            o[(chunk * coutp + n) * KC + (tap % tpc) * CB + pos] = w[(((size_t)n * cin + ci) * ks + ky) * ks + kx];
        }
    return o;
}


static int load_weights_impl(xfh_ctx* c, const void* blob, size_t nbytes) {
    BlobEntry e; char nm[80];
    for (int i = 0; i < XFH_NUM_LAYERS; ++i) {
        const LayerSpec& L = XFH_LAYERS[i];
        snprintf(nm, sizeof nm, "%s.layer.0.weight", L.name);
        if (!blob_find(blob, nbytes, nm, &e) || e.ndim != 4 || (int)e.dims[0] != L.cout || (int)e.dims[1] != L.cin || (int)e.dims[2] != L.ks || (int)e.dims[3] != L.ks) return XFH_ERR_BAD_WEIGHTS;
        const bool running = c->cfg.bn_mode != XFH_BN_BATCH_STATS, folded = c->cfg.bn_mode == XFH_BN_RUNNING_FOLDED;
        std::vector<float> mean, rstd;
        if (running) {
            // eval() semantics: (running_mean, 1/sqrt(running_var + eps)) replace the per-frame statistics
            BlobEntry em, ev; char n2[80];
            snprintf(nm, sizeof nm, "%s.layer.1.running_mean", L.name);
            snprintf(n2, sizeof n2, "%s.layer.1.running_var", L.name);
            if (!blob_find(blob, nbytes, nm, &em) || !blob_find(blob, nbytes, n2, &ev) || (int)em.dims[0] != L.cout || (int)ev.dims[0] != L.cout)
                return XFH_ERR_BAD_WEIGHTS;
            mean.assign(em.p, em.p + L.cout); rstd.resize(L.cout);
            for (int ch = 0; ch < L.cout; ++ch) rstd[ch] = (float)(1.0 / sqrt((double)ev.p[ch] + 1e-5));
        }
        const size_t wcount = (size_t)L.cout * L.cin * L.ks * L.ks;
        std::vector<float> wf(e.p, e.p + wcount);
        if (folded)                                         // BatchNorm into the convolution: W' = W * rstd, b' = -mean * rstd
            for (int co = 0; co < L.cout; ++co)
                for (size_t q = 0; q < wcount / L.cout; ++q) wf[(size_t)co * (wcount / L.cout) + q] *= rstd[co];
        const float* wp = wf.data();
        int rc;
        if (i < 3) {
            std::vector<float> o((size_t)9 * L.cin * L.cout);
            for (int co = 0; co < L.cout; ++co) for (int ci = 0; ci < L.cin; ++ci) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx)
                o[(((size_t)ky * 3 + kx) * L.cin + ci) * L.cout + co] = wp[(((size_t)co * L.cin + ci) * 3 + ky) * 3 + kx];
            rc = upload(c, &c->w.direct[i], o);
        } else {
            const int coutp = (L.cout + 31) / 32 * 32;
            rc = upload(c, &c->w.mfma[i], pack_mfma(wp, L.cout, L.cin, L.ks, coutp));
            if (rc == XFH_OK && L.ks == 3 && L.cin >= 64)                                                // every 3x3 layer with >= 64 input channels: 32-channel chunks
                rc = upload(c, &c->w.alt[i], pack_mfma(wp, L.cout, L.cin, L.ks, coutp, 32));
            if (rc == XFH_OK && L.ks == 3 && L.cin == 64 && L.cout == 64 && L.stride == 1)                 // 7, 10, 11, 16, 17: three taps per chunk
                rc = upload(c, &c->w.alt2[i], pack_mfma(wp, L.cout, L.cin, L.ks, coutp, 64, 3));
            if (rc == XFH_OK && L.ks == 3 && L.cin >= 64)                                                // 7, 9-14, 16, 17: single-frame form
                rc = upload(c, &c->w.m16[i], pack_mfma(wp, L.cout, L.cin, L.ks, L.cout, 64, 1, true));
            if (rc == XFH_OK && L.ks == 3 && L.cin == 24 && L.cout == 64)                              // 6 (block3.0): three taps per chunk for batches <= 8
                rc = upload(c, &c->w.alt[i], pack_mfma(wp, L.cout, L.cin, L.ks, coutp, 64, 3));
            if (rc == XFH_OK && L.ks == 3 && L.cin <= 24 && L.cout == 24)                              // 3, 4, 5: all nine taps in one chunk (persistent kernels, block1.3)
                rc = upload(c, &c->w.alt[i], pack_mfma(wp, L.cout, L.cin, L.ks, coutp, 64, 9));
        }
        if (rc != XFH_OK) return rc;
        if (running) {
            // statistics slots of every frame: the file's values, or the identity when they are folded into the weights (the stored
            // maps are then already activated, and relu(fma(x, 1, -0)) leaves them unchanged in every consumer); k_bn_finalize is never launched
            std::vector<float> st((size_t)c->cfg.max_batch * 2 * L.cout);
            for (int b = 0; b < c->cfg.max_batch; ++b)
                for (int ch = 0; ch < L.cout; ++ch) {
                    st[(size_t)b * 2 * L.cout + ch] = folded ? -0.f : -(mean[ch] * rstd[ch]);       // (beta, alpha) as bn_fold stores them: x * alpha + beta in one fma
                    st[(size_t)b * 2 * L.cout + L.cout + ch] = folded ? 1.f : rstd[ch];
                }
            HIPCK(c, hipMemcpy(c->stat[i], st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
            if (folded) {
                std::vector<float> bb((size_t)((L.cout + 31) / 32 * 32), 0.f);
                for (int ch = 0; ch < L.cout; ++ch) bb[ch] = -mean[ch] * rstd[ch];
                if ((rc = upload(c, &c->w.bn_bias[i], bb)) != XFH_OK) return rc;
            }
        }
    }
    int rc;
    if (!blob_find(blob, nbytes, "block_fusion.2.weight", &e) || e.dims[0] != 64 || e.dims[1] != 64) return XFH_ERR_BAD_WEIGHTS;
    if ((rc = upload(c, &c->w.fus2, pack_mfma(e.p, 64, 64, 1, 64))) != XFH_OK) return rc;
    if (!blob_find(blob, nbytes, "block_fusion.2.bias", &e) || e.dims[0] != 64) return XFH_ERR_BAD_WEIGHTS;
    if ((rc = upload(c, &c->w.fus2_bias, std::vector<float>(e.p, e.p + 64))) != XFH_OK) return rc;
    if (!blob_find(blob, nbytes, "skip1.1.weight", &e) || e.dims[0] != 24) return XFH_ERR_BAD_WEIGHTS;
    if ((rc = upload(c, &c->w.skip_w, std::vector<float>(e.p, e.p + 24))) != XFH_OK) return rc;
    if (!blob_find(blob, nbytes, "skip1.1.bias", &e) || e.dims[0] != 24) return XFH_ERR_BAD_WEIGHTS;
    if ((rc = upload(c, &c->w.skip_b, std::vector<float>(e.p, e.p + 24))) != XFH_OK) return rc;
    if (!blob_find(blob, nbytes, "heatmap_head.2.weight", &e) || e.dims[0] != 1 || e.dims[1] != 64) return XFH_ERR_BAD_WEIGHTS;
    if ((rc = upload(c, &c->w.heat2_w, std::vector<float>(e.p, e.p + 64))) != XFH_OK) return rc;
    if (!blob_find(blob, nbytes, "heatmap_head.2.bias", &e)) return XFH_ERR_BAD_WEIGHTS;
    if ((rc = upload(c, &c->w.heat2_b, std::vector<float>(e.p, e.p + 1))) != XFH_OK) return rc;
    if (!blob_find(blob, nbytes, "keypoint_head.3.weight", &e) || e.dims[0] != 65 || e.dims[1] != 64) return XFH_ERR_BAD_WEIGHTS;
    {
        std::vector<float> o((size_t)64 * 68, 0.f);
        for (int n = 0; n < 65; ++n) for (int k = 0; k < 64; ++k) o[(size_t)k * 68 + n] = e.p[(size_t)n * 64 + k];
        if ((rc = upload(c, &c->w.kp3_w, o)) != XFH_OK) return rc;
    }
    if (!blob_find(blob, nbytes, "keypoint_head.3.bias", &e) || e.dims[0] != 65) return XFH_ERR_BAD_WEIGHTS;
    if ((rc = upload(c, &c->w.kp3_b, std::vector<float>(e.p, e.p + 65))) != XFH_OK) return rc;
    c->w.loaded = true;
    if (xfh_verbose()) fprintf(stderr, "[xfh] ctx %p: weights loaded (%zu bytes, bn_mode %d)\n", (void*)c, nbytes, c->cfg.bn_mode);
    return XFH_OK;
}
int xfh_load_weights(xfh_ctx* c, const void* blob, size_t nbytes) {
    if (!c || !blob) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    // Nothing of this ctx may be in flight while the buffers are replaced: the pipeline lanes are HOST threads holding copies of the weight pointers
    // (a lane between its upload and its kernels is not covered by hipFree's device synchronisation), the twin and the ctx' own streams may still run
    // a submitted frame.  Wait for all of them first -- BEFORE any state changes: a failed synchronisation returns with the old
    // weights still loaded everywhere, parent and children alike (the flag used to be cleared first, and an early return then left
    // the parent "not loaded" while the twin and the lanes still reported the old buffers).
    pipe_wait_idle(c);
    for (int l = 0; l < c->pipe.nlanes; ++l) if (c->pipe.lane[l].ctx) HIPCK(c, hipStreamSynchronize(c->pipe.lane[l].ctx->stream));
    if (c->twin) HIPCK(c, hipStreamSynchronize(c->twin->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (c->aux_stream) HIPCK(c, hipStreamSynchronize(c->aux_stream));
    // A reload replaces the packed buffers one by one (upload() frees and reallocates), and the twin / the pipeline lanes hold COPIES of the
    // pointers: until the reload has succeeded nobody has weights, and from here on there is no early return -- whatever load_weights_impl does, the
    // children get the parent's current state; after a failed reload that is "not loaded" (xfh_extract* then return XFH_ERR_NO_WEIGHTS) instead of
    // pointers into freed memory.
    c->w.loaded = false;
    const int rc = load_weights_impl(c, blob, nbytes);
    int rs = XFH_OK;
    if (c->twin) rs = ctx_share_weights(c, c->twin);
    const int rp = pipe_reshare_weights(c);
    return rc != XFH_OK ? rc : (rs != XFH_OK ? rs : rp);
}

int xfh_load_weights_file(xfh_ctx* c, const char* path) {
    if (!c || !path) return XFH_ERR_INVALID_ARG;
    FILE* f = fopen(path, "rb");
    if (!f) return XFH_ERR_IO;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> buf((size_t)(n > 0 ? n : 0));
    const size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    if (n <= 0 || got != (size_t)n) return XFH_ERR_IO;
    return xfh_load_weights(c, buf.data(), buf.size());
}

// ------------------------------------------------------------------------- extraction
static int check_extract(xfh_ctx* c, const void* gray, int B, int H, int W) {
    if (!c) return XFH_ERR_INVALID_ARG;
    if (!gray || H <= 0 || W <= 0) return XFH_ERR_EMPTY_IMAGE;
    if (B < 1) return XFH_ERR_INVALID_ARG;
    if (B > c->cfg.max_batch) return XFH_ERR_BATCH_TOO_LARGE;
    if (H < 32 || W < 32 || H > c->cfg.max_height || W > c->cfg.max_width) return XFH_ERR_BAD_SIZE;
    if (!c->w.loaded) return XFH_ERR_NO_WEIGHTS;
    return XFH_OK;
}

int xfh_extract_batch_device(xfh_ctx* c, const uint8_t* d_gray, int B, int H, int W, int lap0, int lap1, void* d_records) {
    int rc = check_extract(c, d_gray, B, H, W);
    if (rc != XFH_OK) return rc;
    if (!d_records) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    XfhRange range("xfh:extract_batch_device");
    HIPCK(c, run_extract(c, d_gray, B, H, W, lap0, lap1, (uint8_t*)d_records));
    return XFH_OK;
}

int xfh_extract_batch_device_images(xfh_ctx* c, const uint8_t* d_gray, int B, int H, int W, int lap0, int lap1, void* d_records, void* d_images) {
    int rc = check_extract(c, d_gray, B, H, W);
    if (rc != XFH_OK) return rc;
    if (!d_records || !d_images || (((uintptr_t)d_images) & 15)) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    XfhRange range("xfh:extract_batch_device_images");
    HIPCK(c, run_extract(c, d_gray, B, H, W, lap0, lap1, (uint8_t*)d_records, true, (float*)d_images));
    return XFH_OK;
}

// second single-frame ctx of the submission ring (ctx.h): same configuration with max_batch 1, the parent's weights
}  // extern "C"
int ctx_share_weights(xfh_ctx* c, xfh_ctx* t) {
    t->w = c->w;
    if (c->cfg.bn_mode != XFH_BN_BATCH_STATS && c->w.loaded) {
        // eval() modes: the statistics slots of every frame of the child (the file's values, or the identity when folded)
        const int nb = t->cfg.max_batch < c->cfg.max_batch ? t->cfg.max_batch : c->cfg.max_batch;
        for (int i = 0; i < XFH_NUM_LAYERS; ++i)
            HIPCK(c, hipMemcpy(t->stat[i], c->stat[i], sizeof(float) * 2 * XFH_LAYERS[i].cout * nb, hipMemcpyDeviceToDevice));
    }
    return XFH_OK;
}
extern "C" {
static int twin_ready(xfh_ctx* c) {
    if (c->twin) return XFH_OK;
    xfh_config cfg = c->cfg;
    cfg.max_batch = 1;
    xfh_ctx* t = nullptr;
    const int rc = xfh_create(&cfg, &t);
    if (rc != XFH_OK) return rc;
    t->is_twin = true;
    c->twin = t;
    return ctx_share_weights(c, t);
}

// xfh_extract = submit + collect.  The split form lets the caller overlap its own work (the other camera of a stereo
// rig on a second ctx, tracking of the previous frame, or simply the next frame: up to XFH_SLOTS submissions may be
// outstanding, collected in order) with the GPU.  The record is written by the kernels straight into
// pinned host memory; collect waits for the slot's event and copies only the valid rows (header + front / back
// segments), padding the rest on the host (the reference's default KeyPoint / zero rows, XFextractor.cc:310-312).
int xfh_extract_submit(xfh_ctx* c, const uint8_t* gray, int H, int W, int stride, int lap0, int lap1) {
    int rc = check_extract(c, gray, 1, H, W);
    if (rc != XFH_OK) return rc;
    if (stride < W) return XFH_ERR_INVALID_ARG;
    if (c->s_count >= xfh_ctx::SLOTS) return XFH_ERR_INVALID_ARG;        // collect first
    HIPCK(c, hipSetDevice(c->cfg.device));
    const int k = (c->s_head + c->s_count) % xfh_ctx::SLOTS;
    uint8_t* hg = c->s_hgray[k];
    if (stride == W) memcpy(hg, gray, (size_t)H * W);
    else for (int y = 0; y < H; ++y) memcpy(hg + (size_t)y * W, gray + (size_t)y * stride, (size_t)W);
    xfh_ctx* run = c;
    if (k != 0) {                                            // slot 1 executes on the twin, beside slot 0
        rc = twin_ready(c);
        if (rc != XFH_OK) return rc;
        run = c->twin;
    }
    XfhRange range("xfh:extract_submit");
    // a VGA frame is read from the pinned buffer by k_preproc itself (300 KB over PCIe inside the kernel cost less than the copy command
    // and its hand-over to the compute queue: 0.377 -> 0.370 ms per call); a 1280x720 frame is copied first (0.482 vs 0.499 ms)
    const uint8_t* src = hg;
    if ((size_t)H * W > (size_t)384 * 1024) {
        HIPCK(c, hipMemcpyAsync(c->s_dgray[k], hg, (size_t)H * W, hipMemcpyHostToDevice, run->stream));
        src = c->s_dgray[k];
    }
    HIPCK(c, run_extract(run, src, 1, H, W, lap0, lap1, c->s_hrec[k], false));
    HIPCK(c, hipEventRecord(c->s_done[k], run->stream));
    ++c->s_count;
    return XFH_OK;
}

int xfh_extract_collect(xfh_ctx* c, xfh_keypoint* kps, float* desc, int* n_valid, int* mono_index) {
    if (!c || !kps || !desc) return XFH_ERR_INVALID_ARG;
    if (c->s_count <= 0) return XFH_ERR_INVALID_ARG;           // nothing was submitted
    HIPCK(c, hipSetDevice(c->cfg.device));
    const int k = c->s_head;
    HIPCK(c, hipEventSynchronize(c->s_done[k]));
    c->s_head = (c->s_head + 1) % xfh_ctx::SLOTS; --c->s_count;
    const int nf = c->cfg.nfeatures;
    const uint8_t* r = c->s_hrec[k];
    const RecordHeader* h = (const RecordHeader*)r;
    const xfh_keypoint* rk = (const xfh_keypoint*)(r + xfh_record_kps_offset());
    const float* rd = (const float*)(r + xfh_record_desc_offset(nf));
    int front = h->mono_index, back = h->n_valid - h->mono_index;
    if (front < 0 || back < 0 || front + back > nf) {                                   // a torn header: the padding rows were never written, so there is nothing safe to copy
        c->hip_err = "xfh_extract_collect: inconsistent record header (n_valid / mono_index)";
        return XFH_ERR_HIP;
    }
    memcpy(kps, rk, (size_t)front * sizeof(xfh_keypoint));
    memcpy(desc, rd, (size_t)front * 256);
    const int pad = nf - front - back;
    if (pad > 0) {
        const xfh_keypoint dk = {0.f, 0.f, 0.f, -1.f, 0.f, 0, -1};                       // cv::KeyPoint()
        for (int i = front; i < front + pad; ++i) kps[i] = dk;
        memset(desc + (size_t)front * 64, 0, (size_t)pad * 256);
    }
    if (back > 0) {
        memcpy(kps + (nf - back), rk + (nf - back), (size_t)back * sizeof(xfh_keypoint));
        memcpy(desc + (size_t)(nf - back) * 64, rd + (size_t)(nf - back) * 64, (size_t)back * 256);
    }
    if (n_valid) *n_valid = h->n_valid;
    if (mono_index) *mono_index = h->mono_index;
    return XFH_OK;
}

int xfh_extract(xfh_ctx* c, const uint8_t* gray, int H, int W, int stride, int lap0, int lap1,
                xfh_keypoint* kps, float* desc, int* n_valid, int* mono_index) {
    if (!kps || !desc) return XFH_ERR_INVALID_ARG;
    if (c && c->s_count != 0) return XFH_ERR_INVALID_ARG;      // a submission is outstanding: collect would hand back THAT frame, not this one
    const int rc = xfh_extract_submit(c, gray, H, W, stride, lap0, lap1);
    if (rc != XFH_OK) return rc;
    return xfh_extract_collect(c, kps, desc, n_valid, mono_index);
}
int xfh_detect_and_compute(xfh_ctx* c, const uint8_t* gray, int H, int W, int stride, int lap0, int lap1,
                           xfh_keypoint* kps, float* desc, int* n_valid, int* mono_index) {
    return xfh_extract(c, gray, H, W, stride, lap0, lap1, kps, desc, n_valid, mono_index);
}

// ------------------------------------------------------------------------- matching
int xfh_descriptor_distance(const float* a, const float* b) {
    double s = 0.0;
    for (int k = 0; k < 64; ++k) { const double d = (double)(a[k] - b[k]); s = fma(d, d, s); }      // the device kernels' expression (k_dist_i32)
    const float nd = (float)s;
    return (int)(nd * 512);
}

static int grow(xfh_ctx* c, void** p, size_t* cap, size_t need) {
    if (*p && *cap >= need) return XFH_OK;
    if (*p) { hipFree(*p); *p = nullptr; *cap = 0; }
    HIPCK(c, hipMalloc(p, need));
    *cap = need;
    return XFH_OK;
}

int xfh_match_mnn_device(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, float min_cossim,
                         int* idx1, int* idx2, float* dist, int* n_matches) {
    if (!c || n1 < 0 || n2 < 0 || !n_matches) return XFH_ERR_INVALID_ARG;
    if ((n1 > 0 && !d1) || (n2 > 0 && !d2)) return XFH_ERR_INVALID_ARG;
    if (n1 > 0 && n2 > 0 && (!idx1 || !idx2 || !dist)) return XFH_ERR_INVALID_ARG;
    if ((((uintptr_t)d1) | ((uintptr_t)d2)) & 15) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    XfhRange range("xfh:match_mnn_device");
    HIPCK(c, launch_mnn(c, d1, n1, d2, n2, min_cossim, idx1, idx2, dist, n_matches));
    return XFH_OK;
}

size_t xfh_match_image_bytes(int n) { return n <= 0 ? 0 : (size_t)((n + 255) / 256) * 256 * 64 * sizeof(float); }

int xfh_match_prepare_device(xfh_ctx* c, const float* d, int n, void* image) {
    if (!c || n < 0) return XFH_ERR_INVALID_ARG;
    if (n == 0) return XFH_OK;
    if (!d || !image || ((((uintptr_t)d) | ((uintptr_t)image)) & 15)) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, launch_match_prepare(c, d, n, (float*)image));
    return XFH_OK;
}

int xfh_match_mnn_prepared_device(xfh_ctx* c, const void* image1, int n1, const void* image2, int n2, float min_cossim,
                                  int* idx1, int* idx2, float* dist, int* n_matches) {
    if (!c || n1 < 0 || n2 < 0 || !n_matches) return XFH_ERR_INVALID_ARG;
    if ((n1 > 0 && !image1) || (n2 > 0 && !image2)) return XFH_ERR_INVALID_ARG;
    if (n1 > 0 && n2 > 0 && (!idx1 || !idx2 || !dist)) return XFH_ERR_INVALID_ARG;
    if ((((uintptr_t)image1) | ((uintptr_t)image2)) & 15) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    XfhRange range("xfh:match_mnn_prepared_device");
    HIPCK(c, launch_mnn_prepared(c, (const float*)image1, n1, (const float*)image2, n2, min_cossim, idx1, idx2, dist, n_matches));
    return XFH_OK;
}

// Many pairs in one call (ORBmatcher::match once per frame pair in the reference, ORBmatcher.cc:358-372; its consumers meet one frame with
// several partners): one persistent GEMM launch over the tiles of all pairs + one post launch (kernels_match.hip: launch_mnn_batch).
static int gather_pairs(xfh_ctx* c, int n_pairs, const void* const* image1, const int* n1, const void* const* image2, const int* n2,
                        int* const* idx1, int* const* idx2, float* const* dist, int* n_matches, bool need_out, std::vector<XfhMatchPair>& v) {
    if (!c || n_pairs < 0 || (n_pairs > 0 && (!image1 || !n1 || !image2 || !n2))) return XFH_ERR_INVALID_ARG;
    if (need_out && n_pairs > 0 && (!idx1 || !idx2 || !dist || !n_matches)) return XFH_ERR_INVALID_ARG;
    v.resize((size_t)n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        if (n1[p] < 0 || n2[p] < 0) return XFH_ERR_INVALID_ARG;
        if ((n1[p] > 0 && !image1[p]) || (n2[p] > 0 && !image2[p])) return XFH_ERR_INVALID_ARG;
        if ((((uintptr_t)image1[p]) | ((uintptr_t)image2[p])) & 15) return XFH_ERR_INVALID_ARG;
        if (need_out && n1[p] > 0 && n2[p] > 0 && (!idx1[p] || !idx2[p] || !dist[p])) return XFH_ERR_INVALID_ARG;
        v[p] = XfhMatchPair{(const float*)image1[p], n1[p], (const float*)image2[p], n2[p], need_out ? idx1[p] : nullptr, need_out ? idx2[p] : nullptr,
                            need_out ? dist[p] : nullptr, need_out ? n_matches + p : nullptr};
    }
    return XFH_OK;
}
int xfh_match_mnn_prepared_batch_device(xfh_ctx* c, int n_pairs, const void* const* image1, const int* n1, const void* const* image2, const int* n2,
                                        float min_cossim, int* const* idx1, int* const* idx2, float* const* dist, int* n_matches) {
    std::vector<XfhMatchPair> v;
    const int rc = gather_pairs(c, n_pairs, image1, n1, image2, n2, idx1, idx2, dist, n_matches, true, v);
    if (rc != XFH_OK) return rc;
    if (n_pairs == 0) return XFH_OK;
    HIPCK(c, hipSetDevice(c->cfg.device));
    XfhRange range("xfh:match_mnn_prepared_batch_device");
    HIPCK(c, launch_mnn_batch(c, v.data(), n_pairs, min_cossim));
    return XFH_OK;
}

// n_valid-aware form: the two sets are the nfeatures slots of two extraction records whose prepared images came
// out of xfh_extract_batch_device_images; pairs that touch a padding slot are not reported (the reference's match() would report
// them: zero rows have similarity 0 with everything, ORBmatcher.cc:358-372).  Otherwise xfh_match_mnn_prepared_device.
int xfh_match_records_device(xfh_ctx* c, const void* d_record1, const void* image1, const void* d_record2, const void* image2, float min_cossim,
                             int* idx1, int* idx2, float* dist, int* n_matches) {
    if (!c || !d_record1 || !d_record2 || !image1 || !image2 || !idx1 || !idx2 || !dist || !n_matches) return XFH_ERR_INVALID_ARG;
    if ((((uintptr_t)image1) | ((uintptr_t)image2)) & 15) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    const int nf = c->cfg.nfeatures;
    HIPCK(c, launch_mnn_prepared(c, (const float*)image1, nf, (const float*)image2, nf, min_cossim, idx1, idx2, dist, n_matches,
                                 (const int*)d_record1, (const int*)d_record2));
    return XFH_OK;
}

int xfh_match_mnn(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, float min_cossim,
                  int* idx1, int* idx2, float* dist, int* n_matches) {
    if (!c || n1 < 0 || n2 < 0 || !n_matches) return XFH_ERR_INVALID_ARG;
    if (n1 == 0 || n2 == 0) { *n_matches = 0; return XFH_OK; }
    if (!d1 || !d2 || !idx1 || !idx2 || !dist) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    XfhRange range("xfh:match_mnn");
    MatchWs& w = c->mws;
    const size_t b1 = (size_t)n1 * 64 * 4, b2 = (size_t)n2 * 64 * 4;
    const size_t b1p = (b1 + 255) & ~(size_t)255;
    int rc = grow(c, (void**)&w.h_d1, &w.cap_in, b1p + b2);              // sized for nfeatures x nfeatures in xfh_create
    if (rc != XFH_OK) return rc;
    w.h_d2 = (float*)((char*)w.h_d1 + b1p);
    const int nm = n1 < n2 ? n1 : n2;
    const size_t ob = (size_t)nm * 12 + 256;                              // n at 0, idx1 / idx2 / dist from byte 256 on
    if ((rc = grow(c, (void**)&w.o_buf, &w.cap_out, ob)) != XFH_OK) return rc;
    if (w.cap_hout < ob) {
        if (w.h_out) { hipHostFree(w.h_out); w.h_out = nullptr; w.cap_hout = 0; }
        HIPCK(c, hipHostMalloc((void**)&w.h_out, ob, hipHostMallocDefault));
        w.cap_hout = ob;
    }
    int* o_n = w.o_buf; int* o_idx1 = w.o_buf + 64; int* o_idx2 = o_idx1 + nm; float* o_dist = (float*)(o_idx2 + nm);
    HIPCK(c, hipMemcpyAsync(w.h_d1, d1, b1, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(w.h_d2, d2, b2, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, launch_mnn(c, w.h_d1, n1, w.h_d2, n2, min_cossim, o_idx1, o_idx2, o_dist, o_n));
    // one asynchronous copy of the whole output block into pinned memory (<= 48 KB at 4096 rows), then one wait
    HIPCK(c, hipMemcpyAsync(w.h_out, w.o_buf, ob, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    const int n = w.h_out[0];
    if (n < 0 || n > nm) { c->hip_err = "k_mnn_post: collector timed out"; return XFH_ERR_HIP; }
    memcpy(idx1, w.h_out + 64, (size_t)n * 4);
    memcpy(idx2, w.h_out + 64 + nm, (size_t)n * 4);
    memcpy(dist, w.h_out + 64 + 2 * (size_t)nm, (size_t)n * 4);
    *n_matches = n;
    return XFH_OK;
}

int xfh_distance_i32_device(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, int32_t* out) {
    if (!c || n1 < 0 || n2 < 0) return XFH_ERR_INVALID_ARG;
    if (n1 > 0 && n2 > 0 && (!d1 || !d2 || !out)) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, launch_dist_i32(c, d1, n1, d2, n2, out));
    return XFH_OK;
}

int xfh_distance_i32(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, int32_t* out) {
    if (!c || n1 < 0 || n2 < 0) return XFH_ERR_INVALID_ARG;
    if (n1 == 0 || n2 == 0) return XFH_OK;
    if (!d1 || !d2 || !out) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    MatchWs& w = c->mws;
    const size_t b1 = (size_t)n1 * 64 * 4, b2 = (size_t)n2 * 64 * 4;
    const size_t b1p = (b1 + 255) & ~(size_t)255;
    int rc = grow(c, (void**)&w.h_d1, &w.cap_in, b1p + b2);
    if (rc != XFH_OK) return rc;
    w.h_d2 = (float*)((char*)w.h_d1 + b1p);
    rc = grow(c, (void**)&w.o_tab, &w.cap_tab, (size_t)n1 * n2 * 4);
    if (rc != XFH_OK) return rc;
    HIPCK(c, hipMemcpyAsync(w.h_d1, d1, b1, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(w.h_d2, d2, b2, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, launch_dist_i32(c, w.h_d1, n1, w.h_d2, n2, w.o_tab));
    HIPCK(c, hipMemcpyAsync(out, w.o_tab, (size_t)n1 * n2 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return XFH_OK;
}

int xfh_best2_csr_device(xfh_ctx* c, const float* q, int nq, const float* tg, int nt, const int* offsets, const int* indices, int init_dist,
                         int* best_idx, int* best_dist, int* second_idx, int* second_dist) {
    if (!c || nq < 0 || nt < 0) return XFH_ERR_INVALID_ARG;
    if (nq == 0) return XFH_OK;
    if (!q || !offsets || !indices || !tg || !best_idx || !best_dist || !second_idx || !second_dist) return XFH_ERR_INVALID_ARG;
    if ((((uintptr_t)q) | ((uintptr_t)tg)) & 15) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, launch_best2(c, q, nq, tg, offsets, indices, init_dist, best_idx, best_dist, second_idx, second_dist));
    return XFH_OK;
}

int xfh_best2_csr(xfh_ctx* c, const float* q, int nq, const float* tg, int nt, const int* offsets, const int* indices, int init_dist,
                  int* best_idx, int* best_dist, int* second_idx, int* second_dist) {
    if (!c || nq < 0 || nt < 0) return XFH_ERR_INVALID_ARG;
    if (nq == 0) return XFH_OK;
    if (!q || !offsets || !best_idx || !best_dist || !second_idx || !second_dist) return XFH_ERR_INVALID_ARG;
    const int nnz = offsets[nq];
    if (nnz < 0 || (nnz > 0 && (!indices || !tg))) return XFH_ERR_INVALID_ARG;
    for (int i = 0; i < nq; ++i) if (offsets[i] > offsets[i + 1] || offsets[i] < 0) return XFH_ERR_INVALID_ARG;
    for (int p = 0; p < nnz; ++p) if (indices[p] < 0 || indices[p] >= nt) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    MatchWs& w = c->mws;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t bq = al((size_t)nq * 256), bt = al((size_t)nt * 256 + 16), bo = al((size_t)(nq + 1) * 4), bi = al((size_t)nnz * 4 + 16), br = al((size_t)nq * 4);
    int rc = grow(c, &w.b2_buf, &w.cap_b2, bq + bt + bo + bi + 4 * br);
    if (rc != XFH_OK) return rc;
    char* p0 = (char*)w.b2_buf;
    float* dq = (float*)p0; float* dt = (float*)(p0 + bq); int* doff = (int*)(p0 + bq + bt); int* dind = (int*)(p0 + bq + bt + bo);
    int* o0 = (int*)(p0 + bq + bt + bo + bi); int* o1 = (int*)((char*)o0 + br); int* o2 = (int*)((char*)o1 + br); int* o3 = (int*)((char*)o2 + br);
    HIPCK(c, hipMemcpyAsync(dq, q, (size_t)nq * 256, hipMemcpyHostToDevice, c->stream));
    if (nt > 0) HIPCK(c, hipMemcpyAsync(dt, tg, (size_t)nt * 256, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(doff, offsets, (size_t)(nq + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (nnz > 0) HIPCK(c, hipMemcpyAsync(dind, indices, (size_t)nnz * 4, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, launch_best2(c, dq, nq, dt, doff, dind, init_dist, o0, o1, o2, o3));
    HIPCK(c, hipMemcpyAsync(best_idx, o0, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(best_dist, o1, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(second_idx, o2, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(second_dist, o3, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return XFH_OK;
}

int xfh_distinctive_csr_device(xfh_ctx* c, const float* table, int n_rows, const int* offsets, const int* indices, int n_groups,
                               int max_group, int* best_pos, int* best_median) {
    if (!c || n_rows < 0 || n_groups < 0 || max_group < 0 || max_group > XFH_MAX_GROUP) return XFH_ERR_INVALID_ARG;
    if (n_groups == 0) return XFH_OK;
    if (!offsets || !best_pos || !best_median || (max_group > 0 && (!table || !indices))) return XFH_ERR_INVALID_ARG;
    if (((uintptr_t)table) & 15) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, launch_distinctive(c, table, offsets, indices, n_groups, max_group, best_pos, best_median));
    return XFH_OK;
}

int xfh_distinctive_csr(xfh_ctx* c, const float* table, int n_rows, const int* offsets, const int* indices, int n_groups,
                        int* best_pos, int* best_median) {
    if (!c || n_rows < 0 || n_groups < 0) return XFH_ERR_INVALID_ARG;
    if (n_groups == 0) return XFH_OK;
    if (!offsets || !best_pos || !best_median) return XFH_ERR_INVALID_ARG;
    const int nnz = offsets[n_groups];
    if (nnz < 0 || (nnz > 0 && (!indices || !table))) return XFH_ERR_INVALID_ARG;
    int max_group = 0;
    for (int g = 0; g < n_groups; ++g) {
        if (offsets[g] > offsets[g + 1] || offsets[g] < 0) return XFH_ERR_INVALID_ARG;
        if (offsets[g + 1] - offsets[g] > max_group) max_group = offsets[g + 1] - offsets[g];
    }
    if (max_group > XFH_MAX_GROUP) return XFH_ERR_INVALID_ARG;
    for (int p = 0; p < nnz; ++p) if (indices[p] < 0 || indices[p] >= n_rows) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    MatchWs& w = c->mws;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t bt = al((size_t)n_rows * 256 + 16), bo = al((size_t)(n_groups + 1) * 4), bi = al((size_t)nnz * 4 + 16), br = al((size_t)n_groups * 4);
    int rc = grow(c, &w.b2_buf, &w.cap_b2, bt + bo + bi + 2 * br);
    if (rc != XFH_OK) return rc;
    char* p0 = (char*)w.b2_buf;
    float* dt = (float*)p0; int* doff = (int*)(p0 + bt); int* dind = (int*)(p0 + bt + bo);
    int* o0 = (int*)(p0 + bt + bo + bi); int* o1 = (int*)((char*)o0 + br);
    if (n_rows > 0) HIPCK(c, hipMemcpyAsync(dt, table, (size_t)n_rows * 256, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(doff, offsets, (size_t)(n_groups + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (nnz > 0) HIPCK(c, hipMemcpyAsync(dind, indices, (size_t)nnz * 4, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, launch_distinctive(c, dt, doff, dind, n_groups, max_group, o0, o1));
    HIPCK(c, hipMemcpyAsync(best_pos, o0, (size_t)n_groups * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(best_median, o1, (size_t)n_groups * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return XFH_OK;
}

// ------------------------------------------------------------------------- plumbing
int xfh_synchronize(xfh_ctx* c) {
    if (!c) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (c->twin) HIPCK(c, hipStreamSynchronize(c->twin->stream));
    // (the batch pipeline's lanes are driven by their worker threads: xfh_extract_batch_wait / _drain are the calls that wait for them)
    return XFH_OK;
}
int xfh_set_stream(xfh_ctx* c, void* s) {
    if (!c) return XFH_ERR_INVALID_ARG;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return XFH_OK;
}
const char* xfh_last_hip_error(xfh_ctx* c) { return c ? c->hip_err.c_str() : ""; }

int xfh_dev_alloc(void** p, size_t n) { return hipMalloc(p, n) == hipSuccess ? XFH_OK : XFH_ERR_OUT_OF_MEMORY; }
int xfh_dev_free(void* p) { return hipFree(p) == hipSuccess ? XFH_OK : XFH_ERR_HIP; }
int xfh_memcpy_h2d(void* d, const void* s, size_t n) { return hipMemcpy(d, s, n, hipMemcpyHostToDevice) == hipSuccess ? XFH_OK : XFH_ERR_HIP; }
int xfh_memcpy_d2h(void* d, const void* s, size_t n) { return hipMemcpy(d, s, n, hipMemcpyDeviceToHost) == hipSuccess ? XFH_OK : XFH_ERR_HIP; }

// ------------------------------------------------------------------------- timing
int xfh_bench_mnn_gemm(xfh_ctx* c, const void* image1, int n1, const void* image2, int n2, int iters, double* us_per_launch) {
    if (!c || !image1 || !image2 || n1 < 1 || n2 < 1 || iters < 1 || !us_per_launch) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, bench_mnn_gemm(c, (const float*)image1, n1, (const float*)image2, n2, iters, us_per_launch));
    return XFH_OK;
}
static int bench_match(xfh_ctx* c, bool prepared, const void* a1, int n1, const void* a2, int n2, float min_cossim,
                       int* idx1, int* idx2, float* dist, int* n_matches, int iters, double* us_per_call) {
    if (!c || !a1 || !a2 || n1 < 1 || n2 < 1 || iters < 1 || !us_per_call || !idx1 || !idx2 || !dist || !n_matches) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    hipEvent_t e0, e1;
    HIPCK(c, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return XFH_ERR_HIP; }
    auto call = [&]() {
        return prepared ? launch_mnn_prepared(c, (const float*)a1, n1, (const float*)a2, n2, min_cossim, idx1, idx2, dist, n_matches)
                        : launch_mnn(c, (const float*)a1, n1, (const float*)a2, n2, min_cossim, idx1, idx2, dist, n_matches);
    };
    hipError_t e = hipSuccess;
    for (int i = 0; i < 200 && e == hipSuccess; ++i) e = call();            // the clocks settle over a few hundred of these ~30 us calls
    if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = call();
    if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    *us_per_call = (double)ms * 1e3 / iters;
    HIPCK(c, e);
    return XFH_OK;
}
int xfh_bench_match_prepared(xfh_ctx* c, const void* image1, int n1, const void* image2, int n2, float min_cossim,
                             int* idx1, int* idx2, float* dist, int* n_matches, int iters, double* us_per_call) {
    return bench_match(c, true, image1, n1, image2, n2, min_cossim, idx1, idx2, dist, n_matches, iters, us_per_call);
}
int xfh_bench_match_raw(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, float min_cossim,
                        int* idx1, int* idx2, float* dist, int* n_matches, int iters, double* us_per_call) {
    return bench_match(c, false, d1, n1, d2, n2, min_cossim, idx1, idx2, dist, n_matches, iters, us_per_call);
}
int xfh_debug_match_plan(int n_pairs, const int* n1, const int* n2, int num_cu, int* tiles, int* workgroups, int* tile0, int* planes_max, int* wg_lo, unsigned long long* keys) {
    if (n_pairs < 1 || n_pairs > MNN_MAX_JOBS || !n1 || !n2 || num_cu < 1 || !tiles || !workgroups || !tile0 || !planes_max || !wg_lo || !keys) return XFH_ERR_INVALID_ARG;
    MnnPairIn in[MNN_MAX_JOBS];
    for (int p = 0; p < n_pairs; ++p) { if (n1[p] < 1 || n2[p] < 1) return XFH_ERR_INVALID_ARG; in[p] = MnnPairIn{nullptr, n1[p], nullptr, n2[p]}; }
    MnnBatch jb;
    *keys = (unsigned long long)mnn_seg_plan(in, n_pairs, num_cu, nullptr, &jb);
    *tiles = jb.T; *workgroups = jb.G;
    for (int p = 0; p < n_pairs; ++p) { tile0[p] = jb.job[p].tile0; planes_max[p] = mnn_seg_planes_max(jb.job[p].P2, jb.T, jb.G); }
    for (int w = 0; w <= jb.G; ++w) wg_lo[w] = mnn_seg_lo(w, jb.T, jb.G);
    return XFH_OK;
}
int xfh_bench_mnn_gemm_batch(xfh_ctx* c, int n_pairs, const void* const* image1, const int* n1, const void* const* image2, const int* n2, int iters, double* us_per_launch,
                             double* sclk_mhz) {
    std::vector<XfhMatchPair> v;
    const int rc = gather_pairs(c, n_pairs, image1, n1, image2, n2, nullptr, nullptr, nullptr, nullptr, false, v);
    if (rc != XFH_OK) return rc;
    if (n_pairs < 1 || iters < 1 || !us_per_launch) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, bench_mnn_gemm_batch(c, v.data(), n_pairs, iters, us_per_launch, sclk_mhz));
    return XFH_OK;
}
int xfh_bench_match_batch(xfh_ctx* c, int n_pairs, const void* const* image1, const int* n1, const void* const* image2, const int* n2, float min_cossim,
                          int* const* idx1, int* const* idx2, float* const* dist, int* n_matches, int iters, double* us_per_call) {
    std::vector<XfhMatchPair> v;
    const int rc = gather_pairs(c, n_pairs, image1, n1, image2, n2, idx1, idx2, dist, n_matches, true, v);
    if (rc != XFH_OK) return rc;
    if (n_pairs < 1 || iters < 1 || !us_per_call) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    hipEvent_t e0, e1;
    HIPCK(c, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return XFH_ERR_HIP; }
    hipError_t e = hipSuccess;
    for (int i = 0; i < 50 && e == hipSuccess; ++i) e = launch_mnn_batch(c, v.data(), n_pairs, min_cossim);
    if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = launch_mnn_batch(c, v.data(), n_pairs, min_cossim);
    if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    *us_per_call = (double)ms * 1e3 / iters;
    HIPCK(c, e);
    return XFH_OK;
}
int xfh_timing_enable(xfh_ctx* c, int kernel_id, unsigned layer_mask) {
    if (!c || kernel_id < 0 || kernel_id >= XFH_K_COUNT) return XFH_ERR_INVALID_ARG;
    KTimer& t = c->timer;
    if (kernel_id != XFH_K_NONE && !t.ev) {
        t.ev = (hipEvent_t*)calloc(2 * KTimer::MAXEV, sizeof(hipEvent_t));
        for (int i = 0; i < 2 * KTimer::MAXEV; ++i) HIPCK(c, hipEventCreate(&t.ev[i]));
    }
    t.kernel_id = kernel_id; t.layer_mask = layer_mask; t.nev = 0; t.launches = 0; t.dropped = 0;
    return XFH_OK;
}
int xfh_timing_read(xfh_ctx* c, int* launches, double* total_ms) {
    if (!c) return XFH_ERR_INVALID_ARG;
    KTimer& t = c->timer;
    HIPCK(c, hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (int i = 0; i < t.nev; ++i) {
        float ms = 0.f;
        HIPCK(c, hipEventElapsedTime(&ms, t.ev[2 * i], t.ev[2 * i + 1]));
        tot += ms;
    }
    if (launches) *launches = t.nev;
    if (total_ms) *total_ms = tot;
    const bool overflow = t.dropped > 0;
    t.nev = 0; t.dropped = 0;
    return overflow ? XFH_ERR_BATCH_TOO_LARGE : XFH_OK;       // more than 4096 matching launches since xfh_timing_enable: the sums cover the first 4096 only
}

// ------------------------------------------------------------------------- debug tensors
int xfh_debug_tensor(xfh_ctx* c, int id, int frame, float* out, size_t cap, size_t* count_out) {
    if (!c || frame < 0 || frame >= c->B || !count_out) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipStreamSynchronize(c->stream));
    const size_t xs = (size_t)c->Hmax * c->Wmax;
    const int H = c->H, W = c->W, h8 = H / 8, w8 = W / 8, h4 = H / 4, w4 = W / 4;
    const float* src = nullptr; size_t n = 0;
    switch (id) {
        case XFH_T_X: src = c->X + frame * xs; n = (size_t)H * W; break;
        case XFH_T_XSTAT: src = c->xstat + frame * 2; n = 2; break;
        case XFH_T_SKIP_POOL: src = c->skip_pool + frame * (xs / 16); n = (size_t)h4 * w4; break;
        case XFH_T_FEATS: src = c->feats + frame * c->raw_stride[17]; n = (size_t)h8 * w8 * 64; break;
        case XFH_T_H1: src = c->H1 + frame * (xs / 64); n = (size_t)h8 * w8; break;
        case XFH_T_K1H: src = c->K1h + frame * xs; n = (size_t)H * W; break;
        default:
            if (id >= XFH_T_RAW0 && id < XFH_T_RAW0 + XFH_NUM_LAYERS) {
                const int i = id - XFH_T_RAW0;
                if (!c->raw[i]) return XFH_ERR_INVALID_ARG;                    // block1.0
                src = c->raw[i] + frame * c->raw_stride[i]; n = (size_t)c->lh[i] * c->lw[i] * XFH_LAYERS[i].cout;
            } else if (id >= XFH_T_STAT0 && id < XFH_T_STAT0 + XFH_NUM_LAYERS) {
                const int i = id - XFH_T_STAT0;
                src = c->stat[i] + (size_t)frame * 2 * XFH_LAYERS[i].cout; n = 2 * (size_t)XFH_LAYERS[i].cout;
            } else if (id == XFH_T_SEL) {
                int N = 0;
                HIPCK(c, hipMemcpy(&N, c->sel_n + frame, sizeof(int), hipMemcpyDeviceToHost));
                *count_out = (size_t)N * 3;
                if (!out || cap < (size_t)N * 3) return out ? XFH_ERR_INVALID_ARG : XFH_OK;
                std::vector<u64> keys((size_t)N);
                if (N > 0) HIPCK(c, hipMemcpy(keys.data(), c->sel_key + (size_t)frame * c->cfg.nfeatures, (size_t)N * 8, hipMemcpyDeviceToHost));
                for (int i = 0; i < N; ++i) {
                    const unsigned idx = (unsigned)(keys[i] & 0xFFFFFFFFull);
                    out[i * 3 + 0] = (float)(idx % (unsigned)W); out[i * 3 + 1] = (float)(idx / (unsigned)W);
                    out[i * 3 + 2] = ord2f(~(unsigned)(keys[i] >> 32));
                }
                return XFH_OK;
            } else return XFH_ERR_INVALID_ARG;
    }
    *count_out = n;
    if (!out) return XFH_OK;
    if (cap < n) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
    return XFH_OK;
}

}  // extern "C"

bool ktimer_slot(xfh_ctx* c, int kernel_id, int layer, hipEvent_t* e0, hipEvent_t* e1) {
    KTimer& t = c->timer;
    if (t.kernel_id == XFH_K_NONE || t.kernel_id != kernel_id) return false;
    if (t.layer_mask != 0 && layer >= 0 && !((t.layer_mask >> layer) & 1u)) return false;
    if (!t.ev) return false;
    if (t.nev >= KTimer::MAXEV) { ++t.dropped; return false; }      // reported by xfh_timing_read: never silently
    *e0 = t.ev[2 * t.nev]; *e1 = t.ev[2 * t.nev + 1];
    ++t.nev;
    return true;
}
