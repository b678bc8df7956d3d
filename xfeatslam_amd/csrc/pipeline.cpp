// pipeline.cpp -- the batched HOST-VISIBLE extraction path ("u8 in -> 4096-row keypoints + descriptors out,
// host-visible"; reference consumer: Frame.cc:611-618 receives host std::vector<cv::KeyPoint> / cv::Mat from
// XFextractor::operator(), src/XFextractor.cc:310-356).
//
//   xfh_extract_batch_submit   B frames in host memory -> B records in host memory, asynchronous
//   xfh_extract_batch_wait     the OLDEST outstanding submit has landed in its records_out
//   xfh_extract_batch_drain    everything submitted so far has
//   xfh_extract_batch          submit + drain
//
// A call is cut into sub-batches of cfg.max_batch frames.  PCIe carries 0.31 MB in + 1.16 MB out per VGA frame at nfeatures 4096 against 33 us of
// kernels: everything depends on the copies of one sub-batch running beside the kernels of others, and on nothing else getting in the way.
//
// Three designs were built and traced before this one (profiles/r03_host_pipeline.md, profiles/r04_host_batch_probe.log):
//  (1) upload stream | kernel stream | download stream per lane, events between them: the kernels of sub-batch t+1 did not start before the DOWNLOAD of
//      sub-batch t had finished although nothing orders them.  The runtime multiplexes all HIP streams of a process onto four hardware queues; an event
//      wait is a barrier packet in the waiting stream's queue, and a barrier that waits for a millisecond-long copy stalls every stream that shares
//      the queue: 15-21 k frames/s.  (2) kernels that read / write the caller's pinned memory themselves: 14-17 k (k_desc stalls on PCIe writes while
//      it occupies the CUs).  (3) round 3: in-order lanes -- a lane's H2D, kernels and D2H on ONE stream, four lanes side by side, no cross-stream
//      event: 22-26 k by box and by step size.  Its flaw: a lane's next kernels sit BEHIND its own download in the stream, and lanes that started
//      together stay in phase -- all four compute, then all four copy (22.4 k at 256 frames per step, 25.6 k at 512 on one box).
//  (4) this file, round 4: the ordering moves to the HOST.  Every lane has a worker thread that drives its sub-batch with blocking waits -- copy in
//      (copy stream), kernels (the lane ctx' streams), copy out (copy stream) -- so a stream never holds a command that waits for a copy engine (a lane's
//      two kernel streams still fork / join its keypoint branch by events for sub-batches above 8 frames, kernel to kernel: run_extract, as on any ctx),
//      and a lane that is copying simply is not in anybody's way.  Sub-batches sit in one queue; whichever lane is free takes the next.  Measured with
//      tools/host_thread_probe.py before it was built: 28.5 k (4 lanes) - 29.2 k (8) against 30.5-31 k device resident on the same box.
// A submit of ONE sub-batch (B <= cfg.max_batch: the latency case) runs on the ctx itself, in order on its stream, as before: no thread hand-off.
// The caller's buffers should be pinned (xfh_host_alloc / xfh_host_register): with pageable memory the HIP runtime stages every copy through its own
// bounce buffers -- still correct, much slower.
#include "ctx.h"
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define HIPCK(c, x) do { hipError_t _e = (x); if (_e != hipSuccess) { (c)->hip_err = std::string(#x) + ": " + hipGetErrorString(_e); return XFH_ERR_HIP; } } while (0)

struct PipeJob { const uint8_t* gray; uint8_t* rec_out; int n, H, W, lap0, lap1, slot; };
struct PipeShared {
    std::mutex m;
    std::mutex out_m;                            // ONE download at a time (see lane_main)
    std::condition_variable cv_job, cv_done;
    std::deque<PipeJob> q;
    bool stop = false;
    int remaining[XFH_PIPE_MAX_BATCHES] = {};    // sub-batches of the slot's submit that have not landed yet
    int status[XFH_PIPE_MAX_BATCHES] = {};       // first error of the slot's sub-batches
    int busy = 0;                                // jobs queued or running
    std::string err;
};

// one lane: take the next sub-batch, whoever's it is, and drive it to the caller's memory
static void lane_main(xfh_ctx* parent, int li) {
    Pipe& P = parent->pipe;
    PipeShared& S = *P.sh;
    PipeLane& L = P.lane[li];
    xfh_ctx* lc = L.ctx;
    hipSetDevice(parent->cfg.device);
    const size_t rec = xfh_record_bytes(parent->cfg.nfeatures);
    for (;;) {
        PipeJob j;
        {
            std::unique_lock<std::mutex> lk(S.m);
            // (a lane beyond xfh_pipeline_lanes' current number stays asleep; at shutdown every lane helps to empty the queue)
            S.cv_job.wait(lk, [&] { return S.stop || (!S.q.empty() && li < P.max_lanes); });
            if (S.q.empty()) return;             // stop, and nothing left to do
            j = S.q.front(); S.q.pop_front();
        }
        const size_t fb = (size_t)j.H * j.W;
        const char* what = "hipMemcpyAsync (frames)";
        hipError_t e = hipMemcpyAsync(lc->d_gray, j.gray, (size_t)j.n * fb, hipMemcpyHostToDevice, L.copy);
        if (e == hipSuccess) e = hipStreamSynchronize(L.copy);
        if (e == hipSuccess) { what = "run_extract"; e = run_extract(lc, lc->d_gray, j.n, j.H, j.W, j.lap0, j.lap1, lc->d_records); }
        if (e == hipSuccess) e = hipStreamSynchronize(lc->stream);
        if (e == hipSuccess) {
            // One download at a time.  Lanes that start together finish their kernels together, and then their downloads share the one PCIe link, nobody
            // computes, and the next round starts in phase again -- the pipeline is bistable (the same configuration read 28 k and 19 k frames/s in two
            // runs).  Taking turns costs a lane nothing it would not lose anyway (the link is the shared resource) and staggers the lanes for good: the
            // first one through starts its next sub-batch while the second one copies.
            std::lock_guard<std::mutex> lk(S.out_m);
            what = "hipMemcpyAsync (records)";
            e = hipMemcpyAsync(j.rec_out, lc->d_records, (size_t)j.n * rec, hipMemcpyDeviceToHost, L.copy);
            if (e == hipSuccess) e = hipStreamSynchronize(L.copy);
        }
        {
            std::lock_guard<std::mutex> lk(S.m);
            if (e != hipSuccess && S.status[j.slot] == XFH_OK) { S.status[j.slot] = XFH_ERR_HIP; S.err = std::string(what) + ": " + hipGetErrorString(e); }
            --S.remaining[j.slot]; --S.busy;
        }
        S.cv_done.notify_all();
    }
}

void pipe_wait_idle(xfh_ctx* c) {                 // every queued sub-batch has landed (also called by xfh_load_weights before it touches a buffer)
    Pipe& P = c->pipe;
    if (!P.sh) return;
    std::unique_lock<std::mutex> lk(P.sh->m);
    P.sh->cv_done.wait(lk, [&] { return P.sh->busy == 0; });
}

void pipe_destroy(xfh_ctx* c) {
    Pipe& P = c->pipe;
    if (P.sh) {
        { std::lock_guard<std::mutex> lk(P.sh->m); P.sh->stop = true; }
        P.sh->cv_job.notify_all();
    }
    for (int l = 0; l < P.nlanes; ++l) {
        PipeLane& L = P.lane[l];
        if (L.thread) { std::thread* t = (std::thread*)L.thread; t->join(); delete t; }      // drains the queue first (lane_main only leaves on an empty queue)
        if (L.copy) { hipStreamSynchronize(L.copy); hipStreamDestroy(L.copy); }
        if (L.ctx) xfh_destroy(L.ctx);            // synchronises the lane's own streams first
        L = PipeLane();
    }
    P.nlanes = 0;
    if (c->stream && P.inline_busy) hipStreamSynchronize(c->stream);
    for (int k = 0; k < XFH_PIPE_MAX_BATCHES; ++k) if (P.inline_done[k]) { hipEventDestroy(P.inline_done[k]); P.inline_done[k] = nullptr; }
    delete P.sh; P.sh = nullptr;
    P.b_head = P.b_count = 0; P.inline_busy = false;
}

int pipe_reshare_weights(xfh_ctx* c) {
    pipe_wait_idle(c);                            // no lane is in the middle of a sub-batch while its weight pointers change
    for (int l = 0; l < c->pipe.nlanes; ++l) {
        xfh_ctx* t = c->pipe.lane[l].ctx;
        HIPCK(c, hipStreamSynchronize(t->stream));
        const int rc = ctx_share_weights(c, t);
        if (rc != XFH_OK) return rc;
    }
    return XFH_OK;
}

// worker lanes [0, want) exist and run afterwards
static int pipe_ready(xfh_ctx* c, int want) {
    Pipe& P = c->pipe;
    if (!P.sh) P.sh = new PipeShared();
    while (P.nlanes < want) {
        PipeLane& L = P.lane[P.nlanes];
        L = PipeLane();
        xfh_ctx* t = nullptr;
        xfh_config cfg = c->cfg;
        const int rc = xfh_create(&cfg, &t);
        if (rc != XFH_OK) return rc;
        t->is_lane = true;
        const int rs = ctx_share_weights(c, t);
        if (rs != XFH_OK) { xfh_destroy(t); return rs; }
        if (hipStreamCreateWithFlags(&L.copy, hipStreamNonBlocking) != hipSuccess) { xfh_destroy(t); L = PipeLane(); c->hip_err = "hipStreamCreateWithFlags (lane copy stream)"; return XFH_ERR_HIP; }
        L.ctx = t;
        const int li = P.nlanes;
        ++P.nlanes;                                   // from here on pipe_destroy cleans the lane up
        L.thread = new std::thread(lane_main, c, li);
        if (xfh_verbose()) fprintf(stderr, "[xfh] ctx %p: pipeline lane %d = ctx %p + worker thread\n", (void*)c, li, (void*)L.ctx);
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
    // lanes already built stay; with a smaller number the surplus ones sleep.  The workers read max_lanes inside their wait predicate under the
    // shared mutex: the store takes the same mutex, so a wake-up can not fall between a worker's check and its sleep
    if (c->pipe.sh) {
        { std::lock_guard<std::mutex> lk(c->pipe.sh->m); c->pipe.max_lanes = lanes; }
        c->pipe.sh->cv_job.notify_all();
    } else {
        c->pipe.max_lanes = lanes;
    }
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
    const int S = c->cfg.max_batch, nsub = (B + S - 1) / S;
    const size_t rec = xfh_record_bytes(c->cfg.nfeatures), fb = (size_t)H * W;
    const int slot = (P.b_head + P.b_count) % XFH_PIPE_MAX_BATCHES;
    XfhRange range("xfh:batch_submit");
    if (nsub == 1) {
        // one sub-batch: on the ctx itself, in order on its stream.  A failure in the middle leaves earlier commands queued (they may still copy
        // into records_out): the stream is waited for before the error goes back, so that a failed submit never leaves anything of this call in flight
        if (!P.inline_done[slot]) HIPCK(c, hipEventCreateWithFlags(&P.inline_done[slot], hipEventDisableTiming));
        auto fail = [&](const char* what, hipError_t e) { c->hip_err = std::string(what) + ": " + hipGetErrorString(e); hipStreamSynchronize(c->stream); return XFH_ERR_HIP; };
        P.inline_busy = true;
        hipError_t e = hipMemcpyAsync(c->d_gray, gray, (size_t)B * fb, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) return fail("hipMemcpyAsync (frames)", e);
        if ((e = run_extract(c, c->d_gray, B, H, W, lap0, lap1, c->d_records)) != hipSuccess) return fail("run_extract", e);
        if ((e = hipMemcpyAsync(records_out, c->d_records, (size_t)B * rec, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return fail("hipMemcpyAsync (records)", e);
        if ((e = hipEventRecord(P.inline_done[slot], c->stream)) != hipSuccess) return fail("hipEventRecord", e);
        P.slot_inline[slot] = true;
        ++P.b_count;
        return XFH_OK;
    }
    const int nl = nsub < P.max_lanes ? nsub : P.max_lanes;
    const int rc = pipe_ready(c, nl);
    if (rc != XFH_OK) return rc;
    {
        std::lock_guard<std::mutex> lk(P.sh->m);
        P.sh->remaining[slot] = nsub; P.sh->status[slot] = XFH_OK;
        for (int j = 0; j < nsub; ++j) {
            const int n = B - j * S < S ? B - j * S : S;
            P.sh->q.push_back(PipeJob{gray + (size_t)j * S * fb, (uint8_t*)records_out + (size_t)j * S * rec, n, H, W, lap0, lap1, slot});
        }
        P.sh->busy += nsub;
    }
    P.sh->cv_job.notify_all();
    P.slot_inline[slot] = false;
    ++P.b_count;
    return XFH_OK;
}

int xfh_extract_batch_wait(xfh_ctx* c) {
    if (!c) return XFH_ERR_INVALID_ARG;
    Pipe& P = c->pipe;
    if (P.b_count <= 0) return XFH_ERR_INVALID_ARG;             // nothing outstanding
    HIPCK(c, hipSetDevice(c->cfg.device));
    const int slot = P.b_head;
    int rc = XFH_OK;
    if (P.slot_inline[slot]) {
        const hipError_t e = hipEventSynchronize(P.inline_done[slot]);
        if (e != hipSuccess) { c->hip_err = std::string("hipEventSynchronize: ") + hipGetErrorString(e); rc = XFH_ERR_HIP; }
    } else {
        std::unique_lock<std::mutex> lk(P.sh->m);
        P.sh->cv_done.wait(lk, [&] { return P.sh->remaining[slot] == 0; });
        rc = P.sh->status[slot];
        if (rc != XFH_OK) c->hip_err = P.sh->err;
    }
    P.b_head = (P.b_head + 1) % XFH_PIPE_MAX_BATCHES; --P.b_count;
    return rc;
}

int xfh_extract_batch_drain(xfh_ctx* c) {
    if (!c) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    Pipe& P = c->pipe;
    int rc = XFH_OK;
    while (P.b_count > 0) { const int r = xfh_extract_batch_wait(c); if (rc == XFH_OK) rc = r; }
    pipe_wait_idle(c);
    if (P.inline_busy) { HIPCK(c, hipStreamSynchronize(c->stream)); P.inline_busy = false; }
    return rc;
}

int xfh_extract_batch(xfh_ctx* c, const uint8_t* gray, int B, int H, int W, int lap0, int lap1, void* records_out) {
    const int rc = xfh_extract_batch_submit(c, gray, B, H, W, lap0, lap1, records_out);
    const int rw = c ? xfh_extract_batch_drain(c) : XFH_OK;       // also after a failed submit: nothing of this call may still be in flight
    return rc != XFH_OK ? rc : rw;
}

}  // extern "C"
