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
// sub-batch, IN ORDER ON THE LANE'S ONE STREAM: the H2D copy of its frames (SDMA), the kernels, the D2H copy of the padded
// records (SDMA).  Overlap comes from the lanes: while one lane's stream sits in a copy, the kernels of the others own the
// CUs; PCIe carries 0.31 MB in + 1.16 MB out per VGA frame at nfeatures 4096.  The host never waits before
// xfh_extract_batch_wait, and there is NO cross-stream event:
//
// the classic three-stream form (upload stream | kernel stream | download stream, two buffer generations, events between them)
// was built first and traced (rocprofv3 --kernel-trace --memory-copy-trace, profiles/r03_host_pipeline.md): the kernels of
// sub-batch t+1 did not start before the DOWNLOAD of sub-batch t had finished although nothing orders them.  The runtime
// multiplexes all HIP streams of a process onto four hardware queues; an event wait is a barrier packet in the waiting stream's
// queue, and a barrier packet that waits for a millisecond-long SDMA copy stalls every other stream that happens to share
// that queue -- 15-16 k frames/s whatever the shape, the same as doing the three stages one after the other.  (Kernels that
// read / write the caller's pinned memory themselves, no copy commands at all, were measured too: 15-16 k, k_desc then stalls on
// PCIe writes while it occupies the CUs.)  With in-order lanes a copy can only ever delay its own lane.
// The caller's buffers should be pinned (xfh_host_alloc / xfh_host_register): with pageable memory the HIP runtime stages
// every copy through its own bounce buffers and blocks the calling thread -- still correct, much slower.
#include "ctx.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define HIPCK(c, x) do { hipError_t _e = (x); if (_e != hipSuccess) { (c)->hip_err = std::string(#x) + ": " + hipGetErrorString(_e); return XFH_ERR_HIP; } } while (0)

void pipe_destroy(xfh_ctx* c) {
    Pipe& P = c->pipe;
    for (int l = 0; l < P.nlanes; ++l) {
        PipeLane& L = P.lane[l];
        if (L.ctx && L.ctx != c) xfh_destroy(L.ctx);          // synchronises the lane's own streams first
        else if (c->stream) hipStreamSynchronize(c->stream);
        for (int k = 0; k < XFH_PIPE_MAX_BATCHES; ++k) if (L.done[k]) hipEventDestroy(L.done[k]);
        L = PipeLane();
    }
    P.nlanes = 0;
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
    while (P.nlanes < want) {
        PipeLane& L = P.lane[P.nlanes];
        L = PipeLane();
        if (P.nlanes == 0) L.ctx = c;
        else {
            xfh_ctx* t = nullptr;
            xfh_config cfg = c->cfg;
            cfg.flags |= XFH_FLAG_SERIAL_BRANCH;       // one stream per lane (the keypoint branch inline): the lanes are the concurrency here; measured
                                                        // 25.2-25.7 k frames/s against 18-23 k with a second stream per lane (profiles/r03_host_pipeline.md)
            const int rc = xfh_create(&cfg, &t);
            if (rc != XFH_OK) return rc;
            t->is_lane = true;
            L.ctx = t;
            const int rs = ctx_share_weights(c, t);
            if (rs != XFH_OK) { xfh_destroy(t); L = PipeLane(); return rs; }
        }
        ++P.nlanes;                                   // from here on pipe_destroy cleans the lane up
        if (xfh_verbose()) fprintf(stderr, "[xfh] ctx %p: pipeline lane %d = ctx %p\n", (void*)c, P.nlanes - 1, (void*)L.ctx);
        for (int k = 0; k < XFH_PIPE_MAX_BATCHES; ++k) HIPCK(c, hipEventCreateWithFlags(&L.done[k], hipEventDisableTiming));
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
    XfhRange range("xfh:batch_submit");
    const size_t rec = xfh_record_bytes(c->cfg.nfeatures), fb = (size_t)H * W;
    const int slot = (P.b_head + P.b_count) % XFH_PIPE_MAX_BATCHES;
    unsigned used = 0;
    // a failure in the middle leaves earlier sub-batches queued (they will still copy into records_out): the lanes touched so far are waited for
    // before the error is returned, so that a failed submit never leaves anything of this call in flight (xfh_extract_batch_wait could not cover it)
    auto fail = [&](int code) {
        for (int l = 0; l < nl; ++l) if (used >> l & 1) hipStreamSynchronize(P.lane[l].ctx->stream);
        return code;
    };
    for (int j = 0; j < nsub; ++j) {
        const int n = B - j * S < S ? B - j * S : S;
        const int li = nl == 1 ? 0 : (int)(P.next % (unsigned)nl);             // a call of one sub-batch always runs on the ctx itself
        ++P.next;
        PipeLane& L = P.lane[li];
        xfh_ctx* lc = L.ctx;
        L.busy = true;
        used |= 1u << li;
        hipError_t e = hipMemcpyAsync(lc->d_gray, gray + (size_t)j * S * fb, (size_t)n * fb, hipMemcpyHostToDevice, lc->stream);
        if (e != hipSuccess) { c->hip_err = std::string("hipMemcpyAsync (frames): ") + hipGetErrorString(e); return fail(XFH_ERR_HIP); }
        {
            const int flags = lc->cfg.flags;
            if (nl > 1) lc->cfg.flags |= XFH_FLAG_SERIAL_BRANCH;            // lane 0 is the caller's ctx: same rule while it works as a lane
            e = run_extract(lc, lc->d_gray, n, H, W, lap0, lap1, lc->d_records);
            lc->cfg.flags = flags;
            if (e != hipSuccess) { c->hip_err = std::string("run_extract: ") + hipGetErrorString(e); return fail(XFH_ERR_HIP); }
        }
        e = hipMemcpyAsync((uint8_t*)records_out + (size_t)j * S * rec, lc->d_records, (size_t)n * rec, hipMemcpyDeviceToHost, lc->stream);
        if (e != hipSuccess) { c->hip_err = std::string("hipMemcpyAsync (records): ") + hipGetErrorString(e); return fail(XFH_ERR_HIP); }
    }
    // the batch is complete when the last download of every lane it touched is: one event per lane, waited for by the HOST
    for (int l = 0; l < nl; ++l) if (used >> l & 1) HIPCK(c, hipEventRecord(P.lane[l].done[slot], P.lane[l].ctx->stream));
    P.lanes_of[slot] = used;
    ++P.b_count;
    return XFH_OK;
}

int xfh_extract_batch_wait(xfh_ctx* c) {
    if (!c) return XFH_ERR_INVALID_ARG;
    Pipe& P = c->pipe;
    if (P.b_count <= 0) return XFH_ERR_INVALID_ARG;             // nothing outstanding
    HIPCK(c, hipSetDevice(c->cfg.device));
    for (int l = 0; l < P.nlanes; ++l) if (P.lanes_of[P.b_head] >> l & 1) HIPCK(c, hipEventSynchronize(P.lane[l].done[P.b_head]));
    P.b_head = (P.b_head + 1) % XFH_PIPE_MAX_BATCHES; --P.b_count;
    return XFH_OK;
}

int xfh_extract_batch_drain(xfh_ctx* c) {
    if (!c) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    for (int l = 0; l < c->pipe.nlanes; ++l) {
        PipeLane& L = c->pipe.lane[l];
        if (!L.busy) continue;
        HIPCK(c, hipStreamSynchronize(L.ctx->stream));          // the download is the last command of every sub-batch
        L.busy = false;
    }
    c->pipe.b_head = c->pipe.b_count = 0;
    return XFH_OK;
}

int xfh_extract_batch(xfh_ctx* c, const uint8_t* gray, int B, int H, int W, int lap0, int lap1, void* records_out) {
    const int rc = xfh_extract_batch_submit(c, gray, B, H, W, lap0, lap1, records_out);
    const int rw = c ? xfh_extract_batch_drain(c) : XFH_OK;       // also after a failed submit: nothing of this call may still be in flight
    return rc != XFH_OK ? rc : rw;
}

}  // extern "C"
