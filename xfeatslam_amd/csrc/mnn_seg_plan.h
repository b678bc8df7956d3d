// mnn_seg_plan.h -- the work plan of k_mnn_gemm_seg (mnn_gemm_seg.hip.h): the job table a launch carries in its kernel arguments and the
// integer arithmetic that assigns tiles to workgroups and key planes to d1 panels.  Shared by the host (launch_mnn_batch,
// kernels_mnn_gemm.hip), the GEMM and k_mnn_post (mnn_prepost.hip.h), so the three can never disagree about where a key lives.
#pragma once
#include "mnn_layout.h"

#define MNN_MAX_JOBS 16
struct MnnJob {
    const float* img1; const float* img2;            // panel images (mnn_layout.h)
    u64* partR; u64* partC; u64* pairs;              // partR: [planes][ldr], partC: [P1][ldc], pairs: [P1 * 256]
    size_t ldr, ldc;
    int n1, n2, P1, P2, tile0, pad;                  // panels of d1 / d2; tile0: index of the job's first tile in the launch
};
struct MnnBatch { MnnJob job[MNN_MAX_JOBS]; int njobs, T, G, pad; u64* dbg; };   // dbg: probes only (phase stamps)

// The 256 x 256 tiles of all jobs form one sequence (job, d1 panel, d2 panel), T tiles; workgroup w of G owns the tiles with
// tile * G / T == w, i.e. the range [mnn_seg_lo(w), mnn_seg_lo(w + 1)): floor(T / G) or ceil(T / G) tiles each.
__host__ __device__ inline int mnn_seg_wg(int tile, int T, int G) { return (int)(((long long)tile * G) / T); }
__host__ __device__ inline int mnn_seg_lo(int w, int T, int G) { return (int)(((long long)w * T + G - 1) / G); }
// row-key planes of the d1 panel whose first tile is t0 and that has P2 tiles: two (one per wave group) per workgroup that touches it
__host__ __device__ inline int mnn_seg_planes(int t0, int P2, int T, int G) { return 2 * (mnn_seg_wg(t0 + P2 - 1, T, G) - mnn_seg_wg(t0, T, G) + 1); }
// the most planes any d1 panel of a job can have in any launch with G workgroups: a panel's P2 tiles touch at most this many workgroups
__host__ __device__ inline int mnn_seg_planes_max(int P2, int T, int G) { return 2 * (int)((((long long)P2 * G + T - 1) / T) + 1); }

// ---- host: lay a launch out.  `in`: the pairs; keys: one allocation of at least the returned number of u64 (pass nullptr to size it).
// Per job: partR (mnn_seg_planes_max planes of P1 * 256 keys), partC (P1 planes of P2 * 256), pairs (P1 * 256).  Every plane entry a
// launch reads is written by that launch (the GEMM writes whole 256-row planes of every panel it touches), so nothing is cleared between calls.
struct MnnPairIn { const float* img1; int n1; const float* img2; int n2; };
inline size_t mnn_seg_plan(const MnnPairIn* in, int njobs, int num_cu, u64* keys, MnnBatch* jb) {
    int T = 0;
    for (int p = 0; p < njobs; ++p) {
        MnnJob& J = jb->job[p];
        J.img1 = in[p].img1; J.img2 = in[p].img2; J.n1 = in[p].n1; J.n2 = in[p].n2;
        J.P1 = (in[p].n1 + MNN_PANEL - 1) / MNN_PANEL; J.P2 = (in[p].n2 + MNN_PANEL - 1) / MNN_PANEL;
        J.tile0 = T; J.pad = 0;
        J.ldr = (size_t)J.P1 * MNN_PANEL; J.ldc = (size_t)J.P2 * MNN_PANEL;
        T += J.P1 * J.P2;
    }
    const int G = T < num_cu ? T : num_cu;
    jb->njobs = njobs; jb->T = T; jb->G = G; jb->pad = 0; jb->dbg = nullptr;
    size_t used = 0;
    for (int p = 0; p < njobs; ++p) {
        MnnJob& J = jb->job[p];
        const size_t nR = (size_t)mnn_seg_planes_max(J.P2, T, G) * J.ldr, nC = (size_t)J.P1 * J.ldc, nP = J.ldr;
        J.partR = keys ? keys + used : nullptr; used += nR;
        J.partC = keys ? keys + used : nullptr; used += nC;
        J.pairs = keys ? keys + used : nullptr; used += nP;
    }
    return used;
}
