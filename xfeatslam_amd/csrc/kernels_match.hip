// kernels_match.hip -- descriptor matching on gfx950.
//
//   k_rownorm_img : F::normalize of both descriptor sets (ORBmatcher.cc:358-359), stored as the panel images the
//                   GEMM copies into LDS with LDS-DMA (mnn_layout.h, mnn_prepost.hip.h).
//   k_mnn_gemm_img: cosine-similarity GEMM (N1 x 64) . (64 x N2) on v_mfma_f32_32x32x2_f32 with the first level of
//                   the row / column arg-max (:363-368) fused into the epilogue (mnn_gemm.hip.h; its own translation
//                   unit kernels_mnn_gemm.hip).
//   k_mnn_post    : second level (names the member of the winning group by recomputing its 16 dot products
//                   bit-identically), mutual check (:372), min_cossim gate, ordered compaction and distances (:371-403).
//   k_dist_mfma   : dense (int)(512 * ||a-b||^2), ORBmatcher::DescriptorDistance (:2246-2247): MFMA bulk + exact fix-up.
//   k_best2_csr   : best / second-best distance over candidate lists (the SearchBy* inner loop, :75-119).
//   k_distinctive_csr : MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:329-403), one wave per map point.
//
// Numerics: normalised rows and the 64-term dot products are bit-identical to the oracle
// (fp64 sum of squares -> fp32 sqrt/max/div; one fp32 fma chain in k order, which is what
// the f32 MFMA computes), so the arg-max decisions including ties agree by construction:
// ties resolve to the lowest index (packed key ordered(value) << 32 | ~group, then the first
// member of the group that reaches the value).
#include "ctx.h"
#include "mnn_prepost.hip.h"
#include <utility>
#include <stdlib.h>

#include "dist_mfma.hip.h"

// ---- k_best2_csr: best / second-best integer distance over per-query candidate lists -----------
// One wave per query (the query row comes through the scalar cache); lane l visits candidates
// l, l+64, ... and keeps its two smallest (dist << 32 | position) keys, which reproduces the
// reference's sequential rule exactly (strict '<' in list order = smallest (dist, position)); a
// butterfly merges the 64 lane pairs.  Distances are the exact DescriptorDistance arithmetic.
__device__ __forceinline__ void top2_merge(u64& b, u64& s, u64 ob, u64 os) {
    const u64 lo = b < ob ? b : ob, hi = b < ob ? ob : b;
    const u64 ms = s < os ? s : os;
    b = lo; s = hi < ms ? hi : ms;
}
__global__ __launch_bounds__(256)
void k_best2_csr(const float* __restrict__ q, int nq, const float* __restrict__ tg, const int* __restrict__ offsets,
                 const int* __restrict__ indices, int init_dist, int* __restrict__ best_idx, int* __restrict__ best_dist,
                 int* __restrict__ second_idx, int* __restrict__ second_dist) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qi = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (qi >= nq) return;
    const float* qr = q + (size_t)qi * 64;
    const int beg = offsets[qi], end = offsets[qi + 1];
    const u64 NONE = ~0ull;
    u64 b = NONE, s2 = NONE;
    for (int p = beg + lane; p < end; p += 64) {
        const int idx = indices[p];
        const f32x4* tr = (const f32x4*)(tg + (size_t)idx * 64);
        double acc = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const f32x4 tv = tr[g];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double df = (double)(qr[g * 4 + e] - tv[e]); acc = fma(df, df, acc); }
        }
        const float nd = (float)acc;
        const int dist = (int)(nd * 512.0f);
        const u64 key = ((u64)(unsigned)dist << 32) | (u64)(unsigned)(p - beg);
        if (key < b) { s2 = b; b = key; } else if (key < s2) s2 = key;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const u64 ob = __shfl_xor(b, m), os = __shfl_xor(s2, m);
        top2_merge(b, s2, ob, os);
    }
    if (lane == 0) {
        // apply the reference's initial values: a candidate only counts if dist < init_dist
        int bd = init_dist, bi = -1, sd = init_dist, si = -1;
        if (b != NONE && (int)(b >> 32) < init_dist) {
            bd = (int)(b >> 32); bi = indices[beg + (int)(b & 0xFFFFFFFFull)];
            if (s2 != NONE && (int)(s2 >> 32) < init_dist) { sd = (int)(s2 >> 32); si = indices[beg + (int)(s2 & 0xFFFFFFFFull)]; }
        }
        best_idx[qi] = bi; best_dist[qi] = bd; second_idx[qi] = si; second_dist[qi] = sd;
    }
}

hipError_t launch_best2(xfh_ctx* c, const float* q, int nq, const float* tg, const int* offsets, const int* indices, int init_dist,
                        int* best_idx, int* best_dist, int* second_idx, int* second_dist) {
    if (nq <= 0) return hipSuccess;
    launch_k(c, XFH_K_BEST2, -1, k_best2_csr, dim3((nq + 3) / 4), dim3(256), 0, q, nq, tg, offsets, indices, init_dist,
             best_idx, best_dist, second_idx, second_dist);
    return hipGetLastError();
}

// ---- k_distinctive_csr: MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:329-403), batched -------
// One wave per group (map point).  Rows are taken 64 at a time, lane = row: the lane keeps its descriptor in
// registers and walks the group (row j comes through the scalar cache), storing the exact integer distances of
// its row in LDS; the median sorted[(N-1)/2] is then found by rank counting (distances are small integers, N is
// the number of observations of a map point, typically < 30), and a (median, row) key minimum over the wave
// keeps the FIRST row with the least median (:392-400, strict '<').
__global__ __launch_bounds__(64)
void k_distinctive_csr(const float* __restrict__ table, const int* __restrict__ offsets, const int* __restrict__ indices,
                       int n_groups, int ldn, int* __restrict__ best_pos, int* __restrict__ best_median) {
    extern __shared__ int sd[];                        // [64][ldn], ldn odd
    const int g = blockIdx.x, lane = threadIdx.x;
    const int beg = offsets[g], N = offsets[g + 1] - beg;
    if (N <= 0) { if (lane == 0) { best_pos[g] = -1; best_median[g] = 0x7fffffff; } return; }
    const int k = (N - 1) / 2;                         // (size_t)(0.5 * (N - 1)), :394
    u64 best = ~0ull;
    int* row = sd + lane * ldn;
    for (int r0 = 0; r0 < N; r0 += 64) {
        const int i = r0 + lane;
        if (i < N) {
            f32x4 a[16];
            const f32x4* ar = (const f32x4*)(table + (size_t)indices[beg + i] * 64);
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = ar[q];
            for (int j = 0; j < N; ++j) {
                const f32x4* br = (const f32x4*)(table + (size_t)indices[beg + j] * 64);      // wave-uniform
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const f32x4 bv = br[q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const double df = (double)(a[q][e] - bv[e]); acc = fma(df, df, acc); }
                }
                const float nd = (float)acc;
                row[j] = (j == i) ? 0 : (int)(nd * 512.0f);                                    // :375, DescriptorDistance
            }
            int med = 0;
            for (int j = 0; j < N; ++j) {
                const int v = row[j];
                int less = 0, leq = 0;
                for (int m = 0; m < N; ++m) { const int x = row[m]; less += x < v ? 1 : 0; leq += x <= v ? 1 : 0; }
                if (less <= k && k < leq) med = v;
            }
            const u64 key = ((u64)(unsigned)med << 32) | (u64)(unsigned)i;
            best = key < best ? key : best;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const u64 o = __shfl_xor(best, m); best = o < best ? o : best; }
    if (lane == 0) { best_pos[g] = (int)(best & 0xFFFFFFFFull); best_median[g] = (int)(best >> 32); }
}

hipError_t launch_distinctive(xfh_ctx* c, const float* table, const int* offsets, const int* indices, int n_groups, int max_group,
                              int* best_pos, int* best_median) {
    if (n_groups <= 0) return hipSuccess;
    const int ldn = (max_group < 1 ? 1 : max_group) | 1;
    const size_t lds = (size_t)64 * ldn * sizeof(int);
    XFH_SET_LDS_ATTR_ONCE(c, k_distinctive_csr, (size_t)64 * (XFH_MAX_GROUP | 1) * sizeof(int));
    launch_k(c, XFH_K_DISTINCTIVE, -1, k_distinctive_csr, dim3(n_groups), dim3(64), lds, table, offsets, indices, n_groups, ldn, best_pos, best_median);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Matcher workspace: two panel images, the arg-max keys and the (column, value) pairs.  Reserved once in xfh_create for
// cfg.nfeatures x cfg.nfeatures (the SLAM case: frame against frame); a larger call grows it once (the only
// allocation that can happen on the call path, and only the first time a size is seen).
hipError_t match_ws_reserve(xfh_ctx* c, int n1, int n2) {
    MatchWs& w = c->mws;
    const size_t P1 = ((size_t)n1 + MNN_PANEL - 1) / MNN_PANEL, P2 = ((size_t)n2 + MNN_PANEL - 1) / MNN_PANEL;
    hipError_t e;
    if (w.cap_p1 < P1 || w.cap_p2 < P2) {
        const size_t c1 = P1 > w.cap_p1 ? P1 : w.cap_p1, c2 = P2 > w.cap_p2 ? P2 : w.cap_p2;
        if (w.img1) { if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return e; hipFree(w.img1); w.img1 = nullptr; }
        if (w.keys) { hipFree(w.keys); w.keys = nullptr; }
        w.cap_p1 = w.cap_p2 = 0;
        if ((e = hipMalloc((void**)&w.img1, (c1 + c2) * MNN_PANEL_FLOATS * sizeof(float))) != hipSuccess) return e;
        // partR: c2 planes of c1 panels of rows, partC: c1 planes of c2 panels, pairs: c1 panels
        const size_t nkeys = (2 * c1 * c2 + c1) * MNN_PANEL;
        if ((e = hipMalloc((void**)&w.keys, nkeys * sizeof(u64))) != hipSuccess) return e;
        w.cap_p1 = c1; w.cap_p2 = c2;
    }
    w.img2 = w.img1 + w.cap_p1 * MNN_PANEL_FLOATS;
    w.partR = w.keys; w.partC = w.partR + w.cap_p1 * w.cap_p2 * MNN_PANEL; w.pairs = w.partC + w.cap_p1 * w.cap_p2 * MNN_PANEL;
    return hipSuccess;
}

// GEMM + post on two panel images: the keys of block (by, bx) go to plane bx of partR / plane by of partC
static hipError_t launch_gemm_post(xfh_ctx* c, const float* img1, int n1, const float* img2, int n2, float min_cossim,
                                   int* idx1, int* idx2, float* dist, int* n_matches, const int* hdr1 = nullptr, const int* hdr2 = nullptr) {
    MatchWs& w = c->mws;
    hipError_t e;
    const int P1 = (n1 + MNN_PANEL - 1) / MNN_PANEL, P2 = (n2 + MNN_PANEL - 1) / MNN_PANEL;
    const size_t ldr = (size_t)P1 * MNN_PANEL, ldc = (size_t)P2 * MNN_PANEL;
    if ((e = launch_mnn_gemm(c, img1, n1, img2, n2, w.partR, ldr, w.partC, ldc, w.pairs)) != hipSuccess) return e;
    XFH_SET_LDS_ATTR_ONCE(c, k_mnn_post<0>, MNN_POST_LDS);
    const int nb = (n1 + 15) / 16, ncoll = mnn_ncoll(n1);
    hipLaunchKernelGGL(k_mnn_post<0>, dim3(nb + ncoll), dim3(256), MNN_POST_LDS, c->stream, img1, n1, img2, n2, (const u64*)w.partR, ldr, P2,
                       (const u64*)w.partC, ldc, P1, min_cossim, w.pairs, nb, ncoll, idx1, idx2, dist, n_matches, (long long*)nullptr, hdr1, hdr2);
    return hipGetLastError();
}

// measurement hook (xfh_bench_mnn_gemm): `iters` launches of k_mnn_gemm_img alone, back to back, on two prepared images;
// returns the wall time per launch between two stream events.  In a busy stream the dispatch-attached timestamps of
// consecutive kernels overlap (their sum exceeds the wall time, tools/probes/mnn_probe), so this is the kernel's
// steady-state cost; xfh_timing_* reports the dispatch-attached view that rocprofv3 shows.
hipError_t bench_mnn_gemm(xfh_ctx* c, const float* img1, int n1, const float* img2, int n2, int iters, double* us_per_launch) {
    hipError_t e;
    if ((e = match_ws_reserve(c, n1, n2)) != hipSuccess) return e;
    MatchWs& w = c->mws;
    const int P1 = (n1 + MNN_PANEL - 1) / MNN_PANEL, P2 = (n2 + MNN_PANEL - 1) / MNN_PANEL;
    const size_t ldr = (size_t)P1 * MNN_PANEL, ldc = (size_t)P2 * MNN_PANEL;
    hipEvent_t e0, e1;
    if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
    if ((e = hipEventCreate(&e1)) != hipSuccess) { hipEventDestroy(e0); return e; }
    for (int i = 0; i < 20 && e == hipSuccess; ++i) e = launch_mnn_gemm(c, img1, n1, img2, n2, w.partR, ldr, w.partC, ldc, w.pairs);
    if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = launch_mnn_gemm(c, img1, n1, img2, n2, w.partR, ldr, w.partC, ldc, w.pairs);
    if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    *us_per_launch = (double)ms * 1e3 / (iters > 0 ? iters : 1);
    return e;
}

// ORBmatcher::match on raw descriptor rows: normalise + images, GEMM, post (three launches)
hipError_t launch_mnn(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, float min_cossim,
                      int* idx1, int* idx2, float* dist, int* n_matches) {
    hipError_t e;
    if (n1 <= 0 || n2 <= 0) return hipMemsetAsync(n_matches, 0, sizeof(int), c->stream);
    if ((e = match_ws_reserve(c, n1, n2)) != hipSuccess) return e;
    MatchWs& w = c->mws;
    const int P1 = (n1 + MNN_PANEL - 1) / MNN_PANEL, P2 = (n2 + MNN_PANEL - 1) / MNN_PANEL;
    hipLaunchKernelGGL(k_rownorm_img, dim3((P1 + P2) * 16), dim3(256), 0, c->stream, d1, n1, d2, n2, P1, w.img1, w.img2);
    return launch_gemm_post(c, w.img1, n1, w.img2, n2, min_cossim, idx1, idx2, dist, n_matches);
}

// one descriptor set -> its panel image (xfh_match_prepare_device): the per-frame half of launch_mnn
hipError_t launch_match_prepare(xfh_ctx* c, const float* d, int n, float* img) {
    if (n <= 0) return hipSuccess;
    const int P = (n + MNN_PANEL - 1) / MNN_PANEL;
    hipLaunchKernelGGL(k_rownorm_img, dim3(P * 16), dim3(256), 0, c->stream, d, n, (const float*)nullptr, 0, P, img, (float*)nullptr);
    return hipGetLastError();
}

// ORBmatcher::match on two prepared images: GEMM + post (two launches)
hipError_t launch_mnn_prepared(xfh_ctx* c, const float* img1, int n1, const float* img2, int n2, float min_cossim,
                               int* idx1, int* idx2, float* dist, int* n_matches, const int* hdr1, const int* hdr2) {
    hipError_t e;
    if (n1 <= 0 || n2 <= 0) return hipMemsetAsync(n_matches, 0, sizeof(int), c->stream);
    if ((e = match_ws_reserve(c, n1, n2)) != hipSuccess) return e;
    return launch_gemm_post(c, img1, n1, img2, n2, min_cossim, idx1, idx2, dist, n_matches, hdr1, hdr2);
}

// ---- many pairs in one call: k_mnn_gemm_seg over the tiles of all pairs + k_mnn_post_batch (at most MNN_MAX_JOBS pairs per launch pair; longer lists
// go out in chunks).  Key planes and pairs live in one buffer of the ctx, laid out per chunk by mnn_seg_plan and grown when a call needs more.
static hipError_t batch_chunk(xfh_ctx* c, const XfhMatchPair* pr, int n, float min_cossim, bool gemm_only, u64* dbg = nullptr) {
    MatchWs& w = c->mws;
    MnnPairIn in[MNN_MAX_JOBS];
    int m = 0; int src[MNN_MAX_JOBS];
    hipError_t e;
    for (int p = 0; p < n; ++p) {
        if (pr[p].n1 <= 0 || pr[p].n2 <= 0) {             // an empty side: no matches, nothing to multiply
            if (!gemm_only && (e = hipMemsetAsync(pr[p].n_matches, 0, sizeof(int), c->stream)) != hipSuccess) return e;
            continue;
        }
        in[m] = MnnPairIn{pr[p].img1, pr[p].n1, pr[p].img2, pr[p].n2}; src[m++] = p;
    }
    if (m == 0) return hipSuccess;
    MnnBatch jb;
    const size_t need = mnn_seg_plan(in, m, c->num_cu, nullptr, &jb);
    if (w.cap_bkeys < need) {
        if (w.bkeys) { if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return e; hipFree(w.bkeys); w.bkeys = nullptr; w.cap_bkeys = 0; }
        if ((e = hipMalloc((void**)&w.bkeys, need * sizeof(u64))) != hipSuccess) return e;
        w.cap_bkeys = need;
    }
    mnn_seg_plan(in, m, c->num_cu, w.bkeys, &jb);
    jb.dbg = dbg;
    if ((e = launch_mnn_gemm_seg(c, jb)) != hipSuccess) return e;
    if (gemm_only) return hipSuccess;
    MnnPostBatch pb;
    int gx = 0;
    for (int q = 0; q < m; ++q) {
        const MnnJob& J = jb.job[q]; const XfhMatchPair& P = pr[src[q]]; MnnPostArgs& a = pb.job[q];
        a.img1 = J.img1; a.img2 = J.img2; a.partR = J.partR; a.partC = J.partC; a.pairs = J.pairs;
        a.idx1 = P.idx1; a.idx2 = P.idx2; a.dist = P.dist; a.n_matches = P.n_matches;
        a.stamps = nullptr; a.hdr1 = nullptr; a.hdr2 = nullptr;
        a.ldr = J.ldr; a.ldc = J.ldc; a.n1 = J.n1; a.n2 = J.n2; a.npr = 0; a.npc = J.P1; a.nb = (J.n1 + 15) / 16; a.ncoll = mnn_ncoll(J.n1);
        a.segT = jb.T; a.segG = jb.G; a.tile0 = J.tile0; a.P2 = J.P2; a.min_cossim = min_cossim; a.pad = 0;
        if (a.nb + a.ncoll > gx) gx = a.nb + a.ncoll;
    }
    for (int q = m; q < MNN_MAX_JOBS; ++q) pb.job[q] = pb.job[0];
    XFH_SET_LDS_ATTR_ONCE(c, k_mnn_post_batch, MNN_POST_LDS);
    hipLaunchKernelGGL(k_mnn_post_batch, dim3(gx, m), dim3(256), MNN_POST_LDS, c->stream, pb);
    return hipGetLastError();
}
hipError_t launch_mnn_batch(xfh_ctx* c, const XfhMatchPair* pairs, int n_pairs, float min_cossim) {
    hipError_t e;
    // chunks share the key buffer: the stream orders a chunk's post before the next chunk's GEMM
    for (int p0 = 0; p0 < n_pairs; p0 += MNN_MAX_JOBS)
        if ((e = batch_chunk(c, pairs + p0, n_pairs - p0 < MNN_MAX_JOBS ? n_pairs - p0 : MNN_MAX_JOBS, min_cossim, false)) != hipSuccess) return e;
    return hipSuccess;
}
// measurement hook (xfh_bench_mnn_gemm_batch): `iters` launches of k_mnn_gemm_seg alone on the first <= MNN_MAX_JOBS pairs, wall time per launch
hipError_t bench_mnn_gemm_batch(xfh_ctx* c, const XfhMatchPair* pairs, int n_pairs, int iters, double* us_per_launch, double* sclk_mhz) {
    const int n = n_pairs < MNN_MAX_JOBS ? n_pairs : MNN_MAX_JOBS;
    hipError_t e = hipSuccess;
    hipEvent_t e0, e1;
    u64* dbg = nullptr;
    if ((e = hipMalloc((void**)&dbg, 2 * sizeof(u64))) != hipSuccess) return e;
    if ((e = hipEventCreate(&e0)) != hipSuccess) { hipFree(dbg); return e; }
    if ((e = hipEventCreate(&e1)) != hipSuccess) { hipEventDestroy(e0); hipFree(dbg); return e; }
    for (int i = 0; i < 10 && e == hipSuccess; ++i) e = batch_chunk(c, pairs, n, -1.0f, true);
    if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = batch_chunk(c, pairs, n, -1.0f, true, dbg);       // the last launch's clocks are read below
    if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    u64 h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost);
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(dbg);
    *us_per_launch = (double)ms * 1e3 / (iters > 0 ? iters : 1);
    if (sclk_mhz) *sclk_mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    return e;
}

hipError_t launch_dist_i32(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, int32_t* out) {
    if (n1 <= 0 || n2 <= 0) return hipSuccess;
    const int nt = dist_tiles_per_block(n1, n2, c->num_cu);
    launch_k(c, XFH_K_DIST_I32, -1, k_dist_mfma<0>, dim3((n2 + nt * DTC - 1) / (nt * DTC), (n1 + DT - 1) / DT), dim3(512), 0, d1, n1, d2, n2, out, nt);
    return hipGetLastError();
}
