// pipeline.cpp -- the batched HOST-VISIBLE extraction path ("u8 in -> 4096-row keypoints + descriptors out,
// host-visible"; reference consumer: Frame.cc:611-618 receives host std::vector<cv::KeyPoint> / cv::Mat from
// XFextractor::operator(), src/XFextractor.cc:310-356).
//
//   xfh_extract_batch_submit   B frames in host memory -> B records in host memory, asynchronous
//   xfh_extract_batch_wait     the OLDEST outstanding submit has landed in its records_out
//   xfh_extract_batch_drain    everything submitted so far has
//   xfh_extract_batch          submit + drain
//
// A call is cut into sub-batches of cfg.max_batch frames which rotate over the lanes of the ctx (ctx.h: PipeLane).  Per
// sub-batch: one H2D copy on the lane's upload stream (SDMA), the kernels on the lane ctx' streams, one D2H copy of the
// padded records on the lane's download stream (SDMA); two generations of device buffers per lane, events only, no host
// synchronisation before xfh_extract_batch_wait.  PCIe carries 0.31 MB in + 1.16 MB out per VGA frame at nfeatures 4096.
// The caller's buffers should be pinned (xfh_host_alloc / xfh_host_register): with pageable memory the HIP runtime stages
// every copy through its own bounce buffers and the copies serialise against the kernels -- still correct, much slower.
#include "ctx.h"
#include <string.h>

#define HIPCK(c, x) do { hipError_t _e = (x); if (_e != hipSuccess) { (c)->hip_err = std::string(#x) + ": " + hipGetErrorString(_e); return XFH_ERR_HIP; } } while (0)

void pipe_destroy(xfh_ctx* c) {
    Pipe& P = c->pipe;
    for (int l = 0; l < P.nlanes; ++l) {
        PipeLane& L = P.lane[l];
        if (L.h2d) hipStreamSynchronize(L.h2d);
        if (L.d2h) hipStreamSynchronize(L.d2h);
        if (L.ctx && L.ctx != c) xfh_destroy(L.ctx);          // synchronises the lane's own streams first
        else if (c->stream) hipStreamSynchronize(c->stream);
        if (L.d_gray[1]) hipFree(L.d_gray[1]);
        if (L.d_rec[1]) hipFree(L.d_rec[1]);
        for (int g = 0; g < 2; ++g) {
            if (L.ev_h2d[g]) hipEventDestroy(L.ev_h2d[g]);
            if (L.ev_k[g]) hipEventDestroy(L.ev_k[g]);
            if (L.ev_d2h[g]) hipEventDestroy(L.ev_d2h[g]);
        }
        if (L.h2d) hipStreamDestroy(L.h2d);
        if (L.d2h) hipStreamDestroy(L.d2h);
        L = PipeLane();
    }
    P.nlanes = 0;
    if (P.join) { hipStreamSynchronize(P.join); hipStreamDestroy(P.join); P.join = nullptr; }
    for (int k = 0; k < XFH_PIPE_MAX_BATCHES; ++k) if (P.batch_ev[k]) { hipEventDestroy(P.batch_ev[k]); P.batch_ev[k] = nullptr; }
    P.b_head = P.b_count = 0;
}

int pipe_reshare_weights(xfh_ctx* c) {
    for (int l = 1; l < c->pipe.nlanes; ++l) {
        xfh_ctx* t = c->pipe.lane[l].ctx;
        HIPCK(c, hipStreamSynchronize(t->stream));
        const int rc = ctx_share_weights(c, t);
        if (rc != XFH_OK) return rc;
    }
    return XFH_OK;
}

// lanes [0, want) exist afterwards
static int pipe_ready(xfh_ctx* c, int want) {
    Pipe& P = c->pipe;
    const size_t rec = xfh_record_bytes(c->cfg.nfeatures), gb = (size_t)c->cfg.max_batch * c->cfg.max_height * c->cfg.max_width;
    if (!P.join) {
        HIPCK(c, hipStreamCreateWithFlags(&P.join, hipStreamNonBlocking));
        for (int k = 0; k < XFH_PIPE_MAX_BATCHES; ++k) HIPCK(c, hipEventCreateWithFlags(&P.batch_ev[k], hipEventDisableTiming));
    }
    while (P.nlanes < want) {
        PipeLane& L = P.lane[P.nlanes];
        L = PipeLane();
        if (P.nlanes == 0) L.ctx = c;
        else {
            xfh_ctx* t = nullptr;
            const int rc = xfh_create(&c->cfg, &t);
            if (rc != XFH_OK) return rc;
            t->is_lane = true;
            L.ctx = t;
            const int rs = ctx_share_weights(c, t);
            if (rs != XFH_OK) { xfh_destroy(t); L = PipeLane(); return rs; }
        }
        ++P.nlanes;                                   // from here on pipe_destroy cleans the lane up
        L.d_gray[0] = L.ctx->d_gray; L.d_rec[0] = L.ctx->d_records;
        if (hipMalloc((void**)&L.d_gray[1], gb) != hipSuccess || hipMalloc((void**)&L.d_rec[1], rec * c->cfg.max_batch) != hipSuccess) return XFH_ERR_OUT_OF_MEMORY;
        HIPCK(c, hipStreamCreateWithFlags(&L.h2d, hipStreamNonBlocking));
        HIPCK(c, hipStreamCreateWithFlags(&L.d2h, hipStreamNonBlocking));
        for (int g = 0; g < 2; ++g) {
            HIPCK(c, hipEventCreateWithFlags(&L.ev_h2d[g], hipEventDisableTiming));
            HIPCK(c, hipEventCreateWithFlags(&L.ev_k[g], hipEventDisableTiming));
            HIPCK(c, hipEventCreateWithFlags(&L.ev_d2h[g], hipEventDisableTiming));
        }
    }
    return XFH_OK;
}

extern "C" {

int xfh_host_alloc(void** p, size_t nbytes) {
    if (!p || nbytes == 0) return XFH_ERR_INVALID_ARG;
    return hipHostMalloc(p, nbytes, hipHostMallocDefault) == hipSuccess ? XFH_OK : XFH_ERR_OUT_OF_MEMORY;
}
int xfh_host_free(void* p) { return !p || hipHostFree(p) == hipSuccess ? XFH_OK : XFH_ERR_HIP; }
int xfh_host_register(void* p, size_t nbytes) {
    if (!p || nbytes == 0) return XFH_ERR_INVALID_ARG;
    return hipHostRegister(p, nbytes, hipHostRegisterDefault) == hipSuccess ? XFH_OK : XFH_ERR_HIP;
}
int xfh_host_unregister(void* p) { return !p || hipHostUnregister(p) == hipSuccess ? XFH_OK : XFH_ERR_HIP; }

int xfh_pipeline_lanes(xfh_ctx* c, int lanes) {
    if (!c || lanes < 1 || lanes > XFH_PIPE_MAX_LANES) return XFH_ERR_INVALID_ARG;
    c->pipe.max_lanes = lanes;                         // lanes already built stay; a smaller number simply leaves them unused
    return XFH_OK;
}

int xfh_extract_batch_submit(xfh_ctx* c, const uint8_t* gray, int B, int H, int W, int lap0, int lap1, void* records_out) {
    if (!c) return XFH_ERR_INVALID_ARG;
    if (!gray || H <= 0 || W <= 0) return XFH_ERR_EMPTY_IMAGE;
    if (B < 1 || !records_out) return XFH_ERR_INVALID_ARG;
    if (H < 32 || W < 32 || H > c->cfg.max_height || W > c->cfg.max_width) return XFH_ERR_BAD_SIZE;
    if (!c->w.loaded) return XFH_ERR_NO_WEIGHTS;
    if (c->s_count != 0) return XFH_ERR_INVALID_ARG;           // slot 0 of the submit / collect ring shares this ctx' frame buffer: collect first
    HIPCK(c, hipSetDevice(c->cfg.device));
    Pipe& P = c->pipe;
    if (P.b_count >= XFH_PIPE_MAX_BATCHES) return XFH_ERR_INVALID_ARG;      // wait for the oldest batch first
    const int S = c->cfg.max_batch, nsub = (B + S - 1) / S, nl = nsub < P.max_lanes ? nsub : P.max_lanes;
    int rc = pipe_ready(c, nl);
    if (rc != XFH_OK) return rc;
    const size_t rec = xfh_record_bytes(c->cfg.nfeatures), fb = (size_t)H * W;
    int last_gen[XFH_PIPE_MAX_LANES];
    for (int l = 0; l < XFH_PIPE_MAX_LANES; ++l) last_gen[l] = -1;
    for (int j = 0; j < nsub; ++j) {
        const int n = B - j * S < S ? B - j * S : S;
        const int li = nl == 1 ? 0 : (int)(P.next % (unsigned)nl);             // a call of one sub-batch always runs on the ctx itself
        PipeLane& L = P.lane[li];
        ++P.next;
        const int g = (int)(L.uses & 1);
        ++L.uses;
        xfh_ctx* lc = L.ctx;
        // upload: d_gray[g] is free once the kernels of its previous use have finished
        HIPCK(c, hipStreamWaitEvent(L.h2d, L.ev_k[g], 0));
        HIPCK(c, hipMemcpyAsync(L.d_gray[g], gray + (size_t)j * S * fb, (size_t)n * fb, hipMemcpyHostToDevice, L.h2d));
        HIPCK(c, hipEventRecord(L.ev_h2d[g], L.h2d));
        // kernels: after the upload, and after the download that last read d_rec[g]
        HIPCK(c, hipStreamWaitEvent(lc->stream, L.ev_h2d[g], 0));
        HIPCK(c, hipStreamWaitEvent(lc->stream, L.ev_d2h[g], 0));
        {
            const hipError_t e = run_extract(lc, L.d_gray[g], n, H, W, lap0, lap1, L.d_rec[g]);
            if (e != hipSuccess) { c->hip_err = std::string("run_extract: ") + hipGetErrorString(e); return XFH_ERR_HIP; }
        }
        HIPCK(c, hipEventRecord(L.ev_k[g], lc->stream));
        // download of the padded records
        HIPCK(c, hipStreamWaitEvent(L.d2h, L.ev_k[g], 0));
        HIPCK(c, hipMemcpyAsync((uint8_t*)records_out + (size_t)j * S * rec, L.d_rec[g], (size_t)n * rec, hipMemcpyDeviceToHost, L.d2h));
        HIPCK(c, hipEventRecord(L.ev_d2h[g], L.d2h));
        L.busy = true;
        last_gen[li] = g;
    }
    // the batch is complete when the last download of every lane it touched is (downloads of one lane finish in order)
    for (int l = 0; l < nl; ++l) if (last_gen[l] >= 0) HIPCK(c, hipStreamWaitEvent(P.join, P.lane[l].ev_d2h[last_gen[l]], 0));
    HIPCK(c, hipEventRecord(P.batch_ev[(P.b_head + P.b_count) % XFH_PIPE_MAX_BATCHES], P.join));
    ++P.b_count;
    return XFH_OK;
}

int xfh_extract_batch_wait(xfh_ctx* c) {
    if (!c) return XFH_ERR_INVALID_ARG;
    Pipe& P = c->pipe;
    if (P.b_count <= 0) return XFH_ERR_INVALID_ARG;             // nothing outstanding
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipEventSynchronize(P.batch_ev[P.b_head]));
    P.b_head = (P.b_head + 1) % XFH_PIPE_MAX_BATCHES; --P.b_count;
    return XFH_OK;
}

int xfh_extract_batch_drain(xfh_ctx* c) {
    if (!c) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    for (int l = 0; l < c->pipe.nlanes; ++l) {
        PipeLane& L = c->pipe.lane[l];
        if (!L.busy) continue;
        HIPCK(c, hipStreamSynchronize(L.d2h));                  // the download is the last command of every sub-batch
        L.busy = false;
    }
    if (c->pipe.join) HIPCK(c, hipStreamSynchronize(c->pipe.join));
    c->pipe.b_head = c->pipe.b_count = 0;
    return XFH_OK;
}

int xfh_extract_batch(xfh_ctx* c, const uint8_t* gray, int B, int H, int W, int lap0, int lap1, void* records_out) {
    const int rc = xfh_extract_batch_submit(c, gray, B, H, W, lap0, lap1, records_out);
    const int rw = c ? xfh_extract_batch_drain(c) : XFH_OK;       // also after a failed submit: nothing of this call may still be in flight
    return rc != XFH_OK ? rc : rw;
}

}  // extern "C"
