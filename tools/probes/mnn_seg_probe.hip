// mnn_seg_probe.hip -- k_mnn_gemm_seg (mnn_gemm_seg.hip.h): correctness against a brute-force reference with the same arithmetic
// and an A/B against the one-tile-per-workgroup kernel k_mnn_gemm_img, for one pair and for batches of pairs (development probe,
// not part of the library).  Build: tools/probes/build_mnn_probe.sh ; run on the GPU box: tools/probes/mnn_seg_probe [iters]
#include "../../xfeatslam_amd/csrc/mnn_prepost.hip.h"
#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

// variants live in mnn_seg_probe_gemm.hip (compiled with -fno-honor-nans)
void probe_seg(int skew, int dbg, int G, hipStream_t s, hipEvent_t e0, hipEvent_t e1, const MnnBatch& jb);
void probe_img(hipStream_t s, hipEvent_t e0, hipEvent_t e1, const float* i1, int n1, const float* i2, int n2, u64* pR, size_t ldr, u64* pC, size_t ldc, u64* pairs);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_ref_norm(const float* d, int n, float* o) {
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (row >= n) return;
    const f32x4 v = *(const f32x4*)(d + (size_t)row * 64 + sub * 4);
    double ss = (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z + (double)v.w * (double)v.w;
    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4); ss += __shfl_xor(ss, 8);
    const float nrm = fmaxf((float)sqrt(ss), 1e-12f);
    *(f32x4*)(o + (size_t)row * 64 + sub * 4) = f32x4{v.x / nrm, v.y / nrm, v.z / nrm, v.w / nrm};
}
__global__ void k_ref_best(const float* a, int na, const float* b, int nb, float* bv, int* bi) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= na) return;
    float ar[64];
    for (int k = 0; k < 64; ++k) ar[k] = a[(size_t)row * 64 + k];
    float best = -INFINITY; int idx = 0;
    for (int j = 0; j < nb; ++j) {
        float acc = 0.f;
        for (int k = 0; k < 64; ++k) acc = fmaf(ar[k], b[(size_t)j * 64 + k], acc);
        if (acc > best) { best = acc; idx = j; }
    }
    bv[row] = best; bi[row] = idx;
}

static unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
static float gauss(unsigned& s) {
    float u1 = ((lcg(s) >> 8) + 1) / 16777217.0f, u2 = (lcg(s) >> 8) / 16777216.0f;
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

struct Set { int n = 0; float* d = nullptr; float* img = nullptr; float* nrm = nullptr; };     // raw rows, panel image, plain normalised rows
struct Ref { std::vector<int> i1, i2; std::vector<float> d; };

static hipStream_t S;

static Set make_set(int n, unsigned seed, const Set* noisy_copy_of, int zero, int dup) {
    Set s; s.n = n;
    std::vector<float> h((size_t)n * 64);
    std::vector<float> src;
    if (noisy_copy_of) { src.resize((size_t)noisy_copy_of->n * 64); CK(hipMemcpy(src.data(), noisy_copy_of->d, src.size() * 4, hipMemcpyDeviceToHost)); }
    for (int j = 0; j < n; ++j) {
        if (noisy_copy_of) { const int r = (int)(lcg(seed) % (unsigned)noisy_copy_of->n); for (int k = 0; k < 64; ++k) h[(size_t)j * 64 + k] = src[(size_t)r * 64 + k] + 0.3f * gauss(seed); }
        else for (int k = 0; k < 64; ++k) h[(size_t)j * 64 + k] = gauss(seed);
    }
    for (int z = 0; z < zero && z < n; ++z) memset(&h[(size_t)((z * 37 + 1) % n) * 64], 0, 256);
    for (int z = 0; z < dup; ++z) { const int a = (z * 131 + 2) % n, b = (z * 977 + 300) % n; memcpy(&h[(size_t)b * 64], &h[(size_t)a * 64], 256); }
    const int P = (n + MNN_PANEL - 1) / MNN_PANEL;
    CK(hipMalloc(&s.d, h.size() * 4)); CK(hipMalloc(&s.img, (size_t)P * MNN_PANEL_FLOATS * 4)); CK(hipMalloc(&s.nrm, h.size() * 4));
    CK(hipMemcpy(s.d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rownorm_img, dim3(P * 16), dim3(256), 0, S, (const float*)s.d, n, (const float*)nullptr, 0, P, s.img, (float*)nullptr);
    hipLaunchKernelGGL(k_ref_norm, dim3((n + 15) / 16), dim3(256), 0, S, (const float*)s.d, n, s.nrm);
    CK(hipStreamSynchronize(S));
    return s;
}
static Ref reference(const Set& a, const Set& b) {
    float *rv1, *rv2; int *ri1, *ri2;
    CK(hipMalloc(&rv1, a.n * 4)); CK(hipMalloc(&rv2, b.n * 4)); CK(hipMalloc(&ri1, a.n * 4)); CK(hipMalloc(&ri2, b.n * 4));
    hipLaunchKernelGGL(k_ref_best, dim3((a.n + 255) / 256), dim3(256), 0, S, (const float*)a.nrm, a.n, (const float*)b.nrm, b.n, rv1, ri1);
    hipLaunchKernelGGL(k_ref_best, dim3((b.n + 255) / 256), dim3(256), 0, S, (const float*)b.nrm, b.n, (const float*)a.nrm, a.n, rv2, ri2);
    CK(hipStreamSynchronize(S));
    std::vector<float> v1(a.n); std::vector<int> i1(a.n), i2(b.n);
    CK(hipMemcpy(v1.data(), rv1, a.n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(i1.data(), ri1, a.n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(i2.data(), ri2, b.n * 4, hipMemcpyDeviceToHost));
    Ref r;
    for (int i = 0; i < a.n; ++i) if (i2[i1[i]] == i) { r.i1.push_back(i); r.i2.push_back(i1[i]); r.d.push_back(sqrtf(2.0f * (1.0f - v1[i]))); }
    hipFree(rv1); hipFree(rv2); hipFree(ri1); hipFree(ri2);
    return r;
}

struct Out { int* idx1; int* idx2; float* dist; int* nm; int cap; };
static Out make_out(int cap) { Out o; o.cap = cap; CK(hipMalloc(&o.idx1, cap * 4)); CK(hipMalloc(&o.idx2, cap * 4)); CK(hipMalloc(&o.dist, cap * 4)); CK(hipMalloc(&o.nm, 4)); return o; }
static bool check_out(const Out& o, const Ref& r, const char* what) {
    int n = -1; CK(hipMemcpy(&n, o.nm, 4, hipMemcpyDeviceToHost));
    bool ok = n == (int)r.i1.size();
    if (ok && n > 0) {
        std::vector<int> a(n), b(n); std::vector<float> d(n);
        CK(hipMemcpy(a.data(), o.idx1, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o.idx2, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), o.dist, n * 4, hipMemcpyDeviceToHost));
        ok = a == r.i1 && b == r.i2 && memcmp(d.data(), r.d.data(), n * 4) == 0;
    }
    if (!ok) printf("    %s: MISMATCH (got %d matches, ref %zu)\n", what, n, r.i1.size());
    return ok;
}

static void post_batch(const MnnBatch& jb, const std::vector<Out>& outs, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
    MnnPostBatch pb; memset(&pb, 0, sizeof pb);
    int gx = 0;
    for (int p = 0; p < jb.njobs; ++p) {
        const MnnJob& J = jb.job[p]; MnnPostArgs& a = pb.job[p];
        a.img1 = J.img1; a.img2 = J.img2; a.partR = J.partR; a.partC = J.partC; a.pairs = J.pairs;
        a.idx1 = outs[p].idx1; a.idx2 = outs[p].idx2; a.dist = outs[p].dist; a.n_matches = outs[p].nm;
        a.ldr = J.ldr; a.ldc = J.ldc; a.n1 = J.n1; a.n2 = J.n2; a.npr = 0; a.npc = J.P1; a.nb = (J.n1 + 15) / 16; a.ncoll = mnn_ncoll(J.n1);
        a.segT = jb.T; a.segG = jb.G; a.tile0 = J.tile0; a.P2 = J.P2; a.min_cossim = -1.0f;
        gx = std::max(gx, a.nb + a.ncoll);
    }
    if (e0) hipExtLaunchKernelGGL(k_mnn_post_batch, dim3(gx, jb.njobs), dim3(256), MNN_POST_LDS, S, e0, e1, 0, pb);
    else hipLaunchKernelGGL(k_mnn_post_batch, dim3(gx, jb.njobs), dim3(256), MNN_POST_LDS, S, pb);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int NCU = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.name, NCU);
    CK(hipFuncSetAttribute((const void*)k_mnn_post<0>, hipFuncAttributeMaxDynamicSharedMemorySize, MNN_POST_LDS));
    CK(hipFuncSetAttribute((const void*)k_mnn_post_batch, hipFuncAttributeMaxDynamicSharedMemorySize, MNN_POST_LDS));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    bool all_ok = true;

    // ---- descriptor sets: four of 4096 rows (B = noisy permuted copies of A, so that mutual matches exist), ragged ones, ties, zero rows
    Set A0 = make_set(4096, 11u, nullptr, 0, 0), A1 = make_set(4096, 12u, nullptr, 100, 64);
    Set B0 = make_set(4096, 21u, &A0, 0, 0), B1 = make_set(4096, 22u, &A1, 100, 64), B2 = make_set(4096, 23u, &A0, 0, 9);
    Set R0 = make_set(1000, 31u, nullptr, 5, 9), R1 = make_set(777, 32u, &R0, 5, 9), R2 = make_set(300, 33u, nullptr, 7, 3), R3 = make_set(200, 34u, &R2, 7, 3);
    Set R4 = make_set(257, 35u, nullptr, 0, 2), R5 = make_set(4097, 36u, &R4, 0, 2), R6 = make_set(1, 37u, nullptr, 0, 0), R7 = make_set(5, 38u, &R6, 0, 0);
    Set R8 = make_set(129, 39u, nullptr, 0, 0), R9 = make_set(127, 40u, &R8, 0, 0), R10 = make_set(2500, 41u, &A0, 3, 3);

    struct PairSpec { const Set* a; const Set* b; };
    auto run_case = [&](const char* name, std::vector<PairSpec> ps, int G_override) {
        const int P = (int)ps.size();
        std::vector<MnnPairIn> in(P);
        for (int p = 0; p < P; ++p) in[p] = MnnPairIn{ps[p].a->img, ps[p].a->n, ps[p].b->img, ps[p].b->n};
        MnnBatch jb; memset(&jb, 0, sizeof jb);
        const int G = G_override > 0 ? G_override : NCU;
        const size_t nk = mnn_seg_plan(in.data(), P, G, nullptr, &jb);
        u64* keys; CK(hipMalloc(&keys, nk * 8)); CK(hipMemset(keys, 0xA5, nk * 8));      // garbage: every key the post reads must have been written
        mnn_seg_plan(in.data(), P, G, keys, &jb);
        std::vector<Out> outs(P); std::vector<Ref> refs(P);
        for (int p = 0; p < P; ++p) { outs[p] = make_out(std::min(ps[p].a->n, ps[p].b->n)); refs[p] = reference(*ps[p].a, *ps[p].b); }
        for (int skew = 0; skew < 2; ++skew) {
            bool ok = true;
            for (int rep = 0; rep < 3; ++rep) {
                probe_seg(skew, 0, jb.G, S, nullptr, nullptr, jb);
                post_batch(jb, outs);
                CK(hipStreamSynchronize(S));
                for (int p = 0; p < P; ++p) { char w[64]; snprintf(w, sizeof w, "pair %d rep %d", p, rep); ok &= check_out(outs[p], refs[p], w); }
            }
            size_t nm = 0; for (auto& r : refs) nm += r.i1.size();
            printf("case %-44s T %4d G %3d skew %d: %zu matches over %d pairs  %s\n", name, jb.T, jb.G, skew, nm, P, ok ? "OK" : "MISMATCH");
            all_ok &= ok;
        }
        for (auto& o : outs) { hipFree(o.idx1); hipFree(o.idx2); hipFree(o.dist); hipFree(o.nm); }
        hipFree(keys);
    };
    run_case("4096 x 4096", {{&A0, &B0}}, 0);
    run_case("4096 x 4096 zero rows, duplicates", {{&A1, &B1}}, 0);
    run_case("1000 x 777", {{&R0, &R1}}, 0);
    run_case("300 x 200", {{&R2, &R3}}, 0);
    run_case("257 x 4097", {{&R4, &R5}}, 0);
    run_case("1 x 5", {{&R6, &R7}}, 0);
    run_case("129 x 127", {{&R8, &R9}}, 0);
    run_case("4096 x 4096 on 100 workgroups", {{&A0, &B0}}, 100);
    run_case("4096 x 4096 on 7 workgroups", {{&A1, &B1}}, 7);
    run_case("one frame, three partners", {{&A0, &B0}, {&A0, &B2}, {&A0, &R10}}, 0);
    run_case("8 pairs of 4096 x 4096", {{&A0, &B0}, {&A1, &B1}, {&A0, &B2}, {&A1, &B0}, {&A0, &B1}, {&A1, &B2}, {&B0, &A0}, {&B1, &A1}}, 0);
    run_case("mixed: 4096^2, 1000x777, 257x4097, 1x5, 129x127, 300x200", {{&A0, &B0}, {&R0, &R1}, {&R4, &R5}, {&R6, &R7}, {&R8, &R9}, {&R2, &R3}}, 0);
    run_case("mixed on 13 workgroups", {{&R0, &R1}, {&R4, &R5}, {&R6, &R7}, {&R8, &R9}, {&R2, &R3}, {&R5, &R4}}, 13);
    run_case("16 pairs", {{&A0, &B0}, {&A1, &B1}, {&A0, &B2}, {&A1, &B0}, {&A0, &B1}, {&A1, &B2}, {&B0, &A0}, {&B1, &A1}, {&R0, &R1}, {&R4, &R5}, {&R6, &R7}, {&R8, &R9}, {&R2, &R3}, {&R5, &R4}, {&R10, &A1}, {&B2, &R10}}, 0);

    // ---- which SIMD do the waves of a workgroup sit on?  (the two groups must meet pairwise on the SIMDs for SKEW to pay)
    {
        MnnPairIn in{A0.img, 4096, B0.img, 4096};
        MnnBatch jb; memset(&jb, 0, sizeof jb);
        const size_t nk = mnn_seg_plan(&in, 1, NCU, nullptr, &jb);
        u64* keys; CK(hipMalloc(&keys, nk * 8)); mnn_seg_plan(&in, 1, NCU, keys, &jb);
        probe_seg(1, 3, jb.G, S, nullptr, nullptr, jb); CK(hipStreamSynchronize(S));
        std::vector<u64> h(8 * 8); CK(hipMemcpy(h.data(), jb.job[0].partR, h.size() * 8, hipMemcpyDeviceToHost));
        for (int w = 0; w < 4; ++w) {
            printf("  HW_ID of workgroup %d, waves 0..7 (simd/wave-slot/cu/se): ", w);
            for (int v = 0; v < 8; ++v) { const unsigned x = (unsigned)h[w * 8 + v]; printf("%u/%u/%u/%u ", (x >> 4) & 3, x & 15, (x >> 8) & 15, (x >> 13) & 7); }
            printf("\n");
        }
        hipFree(keys);
    }

    // ---- timing
    auto time_it = [&](const char* name, double gflop, auto&& launch) {
        for (int i = 0; i < 10; ++i) launch(nullptr, nullptr);
        CK(hipStreamSynchronize(S));
        double tot = 0, best = 1e9;
        for (int i = 0; i < iters; ++i) {
            launch(e0, e1); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; best = std::min(best, (double)ms);
        }
        CK(hipEventRecord(e0, S));
        for (int i = 0; i < iters; ++i) launch(nullptr, nullptr);
        CK(hipEventRecord(e1, S)); CK(hipEventSynchronize(e1));
        float msg; CK(hipEventElapsedTime(&msg, e0, e1));
        // continuous stream: dispatch-event durations of every launch inside a busy stream (what bench.py's match.roofline reads)
        const int NREP = 100;
        std::vector<hipEvent_t> ev(2 * NREP);
        for (auto& e : ev) CK(hipEventCreate(&e));
        for (int i = 0; i < 20; ++i) launch(nullptr, nullptr);
        for (int i = 0; i < NREP; ++i) launch(ev[2 * i], ev[2 * i + 1]);
        CK(hipStreamSynchronize(S));
        double cont = 0;
        for (int i = 0; i < NREP; ++i) { float ms; CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); cont += ms; }
        for (auto& e : ev) hipEventDestroy(e);
        const double us = tot / iters * 1e3, usb = msg / iters * 1e3, usc = cont / NREP * 1e3;
        printf("  %-46s alone %8.2f us (min %8.2f) = %.3f | back-to-back wall %8.2f us = %.3f | events in a busy stream %8.2f us = %.3f of 157.3 TF\n", name, us, best * 1e3,
               gflop / us * 1e3 / 157.3, usb, gflop / usb * 1e3 / 157.3, usc, gflop / usc * 1e3 / 157.3);
    };
    {
        // one pair
        MnnPairIn in{A0.img, 4096, B0.img, 4096};
        MnnBatch jb; memset(&jb, 0, sizeof jb);
        const size_t nk = mnn_seg_plan(&in, 1, NCU, nullptr, &jb);
        u64* keys; CK(hipMalloc(&keys, nk * 8)); mnn_seg_plan(&in, 1, NCU, keys, &jb);
        u64 *bR, *bC, *prs; CK(hipMalloc(&bR, 16 * 4096 * 8)); CK(hipMalloc(&bC, 16 * 4096 * 8)); CK(hipMalloc(&prs, 4096 * 8));
        std::vector<Out> outs(1); outs[0] = make_out(4096);
        const double GF = 2.0 * 4096 * 4096 * 64 / 1e9;
        printf("one pair, 4096 x 4096 (%.3f GFLOP):\n", GF);
        time_it("k_mnn_gemm_img (one tile per workgroup)", GF, [&](hipEvent_t a, hipEvent_t b) { probe_img(S, a, b, A0.img, 4096, B0.img, 4096, bR, 4096, bC, 4096, prs); });
        time_it("k_mnn_gemm_seg lockstep", GF, [&](hipEvent_t a, hipEvent_t b) { probe_seg(0, 0, jb.G, S, a, b, jb); });
        time_it("k_mnn_gemm_seg skew", GF, [&](hipEvent_t a, hipEvent_t b) { probe_seg(1, 0, jb.G, S, a, b, jb); });
        time_it("k_mnn_gemm_seg lockstep, no epilogue", GF, [&](hipEvent_t a, hipEvent_t b) { probe_seg(0, 1, jb.G, S, a, b, jb); });
        time_it("k_mnn_gemm_seg skew, no epilogue", GF, [&](hipEvent_t a, hipEvent_t b) { probe_seg(1, 1, jb.G, S, a, b, jb); });
        time_it("whole call: k_mnn_gemm_img + k_mnn_post", GF, [&](hipEvent_t a, hipEvent_t b) {
            if (a) CK(hipEventRecord(a, S));
            probe_img(S, nullptr, nullptr, A0.img, 4096, B0.img, 4096, bR, 4096, bC, 4096, prs);
            hipLaunchKernelGGL(k_mnn_post<0>, dim3(256 + 16), dim3(256), MNN_POST_LDS, S, (const float*)A0.img, 4096, (const float*)B0.img, 4096, (const u64*)bR, (size_t)4096, 16, (const u64*)bC, (size_t)4096, 16, -1.0f, prs, 256, 16,
                               outs[0].idx1, outs[0].idx2, outs[0].dist, outs[0].nm, (long long*)nullptr, (const int*)nullptr, (const int*)nullptr);
            if (b) CK(hipEventRecord(b, S));
        });
        for (int skew = 0; skew < 2; ++skew) {
            char nm[64]; snprintf(nm, sizeof nm, "whole call: k_mnn_gemm_seg %s + post", skew ? "skew" : "lockstep");
            time_it(nm, GF, [&](hipEvent_t a, hipEvent_t b) { if (a) CK(hipEventRecord(a, S)); probe_seg(skew, 0, jb.G, S, nullptr, nullptr, jb); post_batch(jb, outs); if (b) CK(hipEventRecord(b, S)); });
        }
        hipFree(keys); hipFree(bR); hipFree(bC); hipFree(prs);
    }
    for (int P : {2, 4, 8, 16}) {
        const Set* as[4] = {&A0, &A1, &B0, &B1}; const Set* bs[4] = {&B0, &B1, &B2, &A0};
        std::vector<MnnPairIn> in(P);
        for (int p = 0; p < P; ++p) in[p] = MnnPairIn{as[p & 3]->img, 4096, bs[(p >> 2) & 3]->img, 4096};
        MnnBatch jb; memset(&jb, 0, sizeof jb);
        const size_t nk = mnn_seg_plan(in.data(), P, NCU, nullptr, &jb);
        u64* keys; CK(hipMalloc(&keys, nk * 8)); mnn_seg_plan(in.data(), P, NCU, keys, &jb);
        std::vector<Out> outs(P); for (auto& o : outs) o = make_out(4096);
        const double GF = P * 2.0 * 4096 * 4096 * 64 / 1e9;
        printf("%d pairs of 4096 x 4096 in one launch (%.2f GFLOP):\n", P, GF);
        time_it("k_mnn_gemm_seg lockstep", GF, [&](hipEvent_t a, hipEvent_t b) { probe_seg(0, 0, jb.G, S, a, b, jb); });
        time_it("k_mnn_gemm_seg skew", GF, [&](hipEvent_t a, hipEvent_t b) { probe_seg(1, 0, jb.G, S, a, b, jb); });
        time_it("k_mnn_gemm_seg skew, no epilogue", GF, [&](hipEvent_t a, hipEvent_t b) { probe_seg(1, 1, jb.G, S, a, b, jb); });
        time_it("k_mnn_post_batch", GF, [&](hipEvent_t a, hipEvent_t b) { post_batch(jb, outs, a, b); });
        time_it("whole call: k_mnn_gemm_seg skew + k_mnn_post_batch", GF, [&](hipEvent_t a, hipEvent_t b) { if (a) CK(hipEventRecord(a, S)); probe_seg(1, 0, jb.G, S, nullptr, nullptr, jb); post_batch(jb, outs); if (b) CK(hipEventRecord(b, S)); });
        for (auto& o : outs) { hipFree(o.idx1); hipFree(o.idx2); hipFree(o.dist); hipFree(o.nm); }
        hipFree(keys);
    }
    {   // phase stamps of workgroup 0 (wall clock, 100 MHz), 8 pairs: [K start, K end, E start, E end] per tile and group
        const Set* as[4] = {&A0, &A1, &B0, &B1}; const Set* bs[4] = {&B0, &B1, &B2, &A0};
        std::vector<MnnPairIn> in(8);
        for (int p = 0; p < 8; ++p) in[p] = MnnPairIn{as[p & 3]->img, 4096, bs[(p >> 2) & 3]->img, 4096};
        MnnBatch jb; memset(&jb, 0, sizeof jb);
        const size_t nk = mnn_seg_plan(in.data(), 8, NCU, nullptr, &jb);
        u64* keys; CK(hipMalloc(&keys, nk * 8)); mnn_seg_plan(in.data(), 8, NCU, keys, &jb);
        u64* dbg; CK(hipMalloc(&dbg, 128 * 8)); jb.dbg = dbg;
        for (int skew = 0; skew < 2; ++skew) {
            for (int i = 0; i < 3; ++i) probe_seg(skew, 2, jb.G, S, nullptr, nullptr, jb);
            CK(hipStreamSynchronize(S));
            std::vector<u64> h(128); CK(hipMemcpy(h.data(), dbg, 128 * 8, hipMemcpyDeviceToHost));
            const u64 t0 = std::min(h[0], h[64]);
            printf("phase stamps, 8 pairs, %s (us after the first stamp; per tile: K start, K end | E start, E end):\n", skew ? "skew" : "lockstep");
            for (int g = 0; g < 2; ++g) {
                printf("  group %c:", g ? 'Y' : 'X');
                for (int k = 0; k < 33; ++k) printf("%s%.2f", (k % 4 == 0) ? "  |  " : " ", (double)(h[g * 64 + k] - t0) / 100.0);
                printf("\n");
            }
        }
        hipFree(keys);
    }
    printf("%s\n", all_ok ? "ALL OK" : "SOME MISMATCH");
    return all_ok ? 0 : 1;
}
