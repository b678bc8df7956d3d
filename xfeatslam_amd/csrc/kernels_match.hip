// kernels_match.hip -- descriptor matching on gfx950.
//
//   k_mnn_gemm    : cosine-similarity GEMM (N1 x 64) . (64 x N2) on v_mfma_f32_32x32x2_f32 with
//                   the row L2-normalisation fused into the tile load and the row / column
//                   arg-max fused into the epilogue (reference: the commented-out
//                   ORBmatcher::match, src/ORBmatcher.cc:358-368).
//   k_mnn_reduce  : merges the per-tile arg-max partials.
//   k_mnn_final   : mutual check, min_cossim gate, ordered compaction, distances (:371-403).
//   k_dist_i32    : dense (int)(512 * ||a-b||^2), ORBmatcher::DescriptorDistance (:2246-2247).
//
// Numerics: normalised rows and the 64-term dot products are bit-identical to the oracle
// (fp64 sum of squares -> fp32 sqrt/max/div; one fp32 fma chain in k order, which is what
// the f32 MFMA computes), so the arg-max decisions including ties agree by construction.
// Ties resolve to the lowest index through the packed key (ordered(value) << 32 | ~index).
#include "ctx.h"

#define MT 128          // tile edge (rows of d1 / rows of d2 per workgroup)
#define LDK 68          // padded LDS row (floats): 64 + 4 keeps ds_read_b128 conflict free

__device__ __forceinline__ u64 pack_key(float v, unsigned idx) {
    return ((u64)f2ord(v) << 32) | (u64)(0xFFFFFFFFu - idx);
}
__device__ __forceinline__ u64 umax64(u64 a, u64 b) { return a > b ? a : b; }

// k_rownorm: L2-normalise every descriptor row once (F::normalize, eps 1e-12: fp64 sum of
// squares, fp32 sqrt / max / divide) and store it with the k permutation the MFMA loop wants:
// inside each group of 8, element e sits at position 4*(e&1) + (e>>1), so that lane-half h
// reads k = 8g+2j+h for MFMA j out of one ds_read_b128.  16 lanes per row, 16 rows per block.
__global__ __launch_bounds__(256)
void k_rownorm(const float* __restrict__ d1, int n1, const float* __restrict__ d2, int n2,
               float* __restrict__ o1, float* __restrict__ o2) {
    const int t = threadIdx.x, sub = t & 15;
    int row = blockIdx.x * 16 + (t >> 4);
    const float* d; float* o;
    if (row < n1) { d = d1; o = o1; }
    else { row -= n1; if (row >= n2) return; d = d2; o = o2; }      // 16-lane groups exit together
    const f32x4 v = *(const f32x4*)(d + (size_t)row * 64 + sub * 4);
    double ss = (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z + (double)v.w * (double)v.w;
    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4); ss += __shfl_xor(ss, 8);
    const float nrm = fmaxf((float)sqrt(ss), 1e-12f);
    const float a = v.x / nrm, b = v.y / nrm, c = v.z / nrm, e = v.w / nrm;
    // this lane holds elements e0..e0+3 of group g; its pair lane (sub^1) holds the other four.
    // positions 0..3 = even-half lane's {0,2} + odd-half lane's {0,2} ... resolved with one exchange:
    //   group positions: [e0 e2 e4 e6 | e1 e3 e5 e7]
    const bool odd = sub & 1;
    // even lane keeps (a,c) -> pos 0,1 and gets partner's (a,c) -> pos 2,3 ; odd lane: (b,e) pairs -> pos 4..7
    const float sx = odd ? a : b, sy = odd ? c : e;            // what the partner needs from me
    const float rx = __shfl_xor(sx, 1), ry = __shfl_xor(sy, 1);
    const f32x4 outv = odd ? f32x4{rx, ry, b, e} : f32x4{a, c, rx, ry};
    *(f32x4*)(o + (size_t)row * 64 + (sub >> 1) * 8 + (odd ? 4 : 0)) = outv;
}

__global__ __launch_bounds__(256, 2)
void k_mnn_gemm(const float* __restrict__ d1, int n1, const float* __restrict__ d2, int n2,
                u64* __restrict__ partR, u64* __restrict__ partC, int n1pad, int n2pad) {
    // d1/d2: normalised, k-permuted rows from k_rownorm
    __shared__ __attribute__((aligned(16))) float smem[2 * MT * LDK];
    float* sA = smem;
    float* sB = smem + MT * LDK;
    const int t = threadIdx.x;
    const int bx = blockIdx.x, by = blockIdx.y;
    {
        const int sub = t & 15, r0 = t >> 4;
        f32x4 va[8], vb[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int ra = by * MT + p * 16 + r0, rb = bx * MT + p * 16 + r0;
            va[p] = (ra < n1) ? *(const f32x4*)(d1 + (size_t)ra * 64 + sub * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            vb[p] = (rb < n2) ? *(const f32x4*)(d2 + (size_t)rb * 64 + sub * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            *(f32x4*)(sA + (p * 16 + r0) * LDK + sub * 4) = va[p];
            *(f32x4*)(sB + (p * 16 + r0) * LDK + sub * 4) = vb[p];
        }
    }
    __syncthreads();

    const int wave = t >> 6, lane = t & 63, i = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const float* pa = sA + (wr * 64 + i) * LDK + 4 * h;
    const float* pb = sB + (wc * 64 + i) * LDK + 4 * h;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const f32x4 a0 = *(const f32x4*)(pa + g * 8);
        const f32x4 a1 = *(const f32x4*)(pa + 32 * LDK + g * 8);
        const f32x4 b0 = *(const f32x4*)(pb + g * 8);
        const f32x4 b1 = *(const f32x4*)(pb + 32 * LDK + g * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
    }

    // ---- epilogue -----------------------------------------------------------------------
    // C/D layout of 32x32: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Column arg-max (over rows) is lane-local; for the row arg-max the 64x64 wave tile goes
    // through LDS once so that lane l can scan row l in ascending column order.
    const float NEG = -__builtin_huge_valf();
    const int grow0 = by * MT + wr * 64, gcol0 = bx * MT + wc * 64;
    const bool full = (by * MT + MT <= n1) && (bx * MT + MT <= n2);     // block-uniform
    if (!full) {
        // mask rows/columns outside the problem with -inf (never selected against a finite value)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool vr = grow0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < n1;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const bool vc = gcol0 + ct * 32 + i < n2;
                    if (!(vr && vc)) acc[rt][ct][r] = NEG;
                }
            }
    }
    // column best: ascending row order + strict '>' keeps the lowest row among equals
    float cbv[2] = {NEG, NEG};
    int cbr[2] = {0, 0};
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const float v = acc[rt][ct][r];
                const bool gt = v > cbv[ct];
                cbv[ct] = gt ? v : cbv[ct];
                cbr[ct] = gt ? lrow : cbr[ct];
            }
        }
    u64 ck[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const u64 k = (cbv[ct] > NEG) ? pack_key(cbv[ct], (unsigned)(grow0 + cbr[ct])) : 0ull;
        ck[ct] = umax64(k, __shfl_xor(k, 32));
    }

    __syncthreads();                       // every wave is done with sA/sB: reuse as scratch
    float* sT = smem + wave * (64 * LDK);  // this wave's 64 x 64 tile, row stride LDK
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            sT[lrow * LDK + i] = acc[rt][0][r];
            sT[lrow * LDK + 32 + i] = acc[rt][1][r];
        }
    // the tile is private to the wave: a wave-level fence is enough
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    float rbv = NEG; int rbc = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const f32x4 v = *(const f32x4*)(sT + lane * LDK + q * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool gt = v[e] > rbv;
            rbv = gt ? v[e] : rbv;
            rbc = gt ? (q * 4 + e) : rbc;
        }
    }
    const u64 rkey = (rbv > NEG) ? pack_key(rbv, (unsigned)(gcol0 + rbc)) : 0ull;
    __syncthreads();                       // all tiles consumed: reuse the head of smem for the merges
    u64* sRow = (u64*)smem;                // [4 waves][64]
    u64* sCol = sRow + 4 * 64;             // [4 waves][64]
    sRow[wave * 64 + lane] = rkey;         // lane = local row
    if (lane < 32) { sCol[wave * 64 + lane] = ck[0]; sCol[wave * 64 + 32 + lane] = ck[1]; }
    __syncthreads();
    if (wc == 0) {
        const u64 k = umax64(sRow[wave * 64 + lane], sRow[(wave + 1) * 64 + lane]);
        partR[(size_t)bx * n1pad + by * MT + wr * 64 + lane] = k;
    }
    if (wr == 0) {
        const u64 k = umax64(sCol[wave * 64 + lane], sCol[(wave + 2) * 64 + lane]);
        partC[(size_t)by * n2pad + bx * MT + wc * 64 + lane] = k;
    }
}

__global__ void k_mnn_reduce(const u64* __restrict__ partR, const u64* __restrict__ partC, int nbR, int nbC,
                             int n1, int n2, int n1pad, int n2pad,
                             int* __restrict__ best12, float* __restrict__ val12, int* __restrict__ best21) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n1) {
        u64 k = 0;
        for (int b = 0; b < nbC; ++b) k = umax64(k, partR[(size_t)b * n1pad + t]);
        best12[t] = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
        val12[t] = ord2f((unsigned)(k >> 32));
    } else if (t >= n1pad && t - n1pad < n2) {
        const int c = t - n1pad;
        u64 k = 0;
        for (int b = 0; b < nbR; ++b) k = umax64(k, partC[(size_t)b * n2pad + c]);
        best21[c] = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
    }
}

// one workgroup: mutual check + gate + ordered compaction (ascending idx1)
__global__ __launch_bounds__(1024)
void k_mnn_final(const int* __restrict__ best12, const float* __restrict__ val12, const int* __restrict__ best21,
                 int n1, float min_cossim, int* __restrict__ idx1, int* __restrict__ idx2,
                 float* __restrict__ dist, int* __restrict__ n_matches) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n1; i0 += 1024) {
        const int i = i0 + t;
        bool keep = false; int j = 0; float v = 0.f;
        if (i < n1) {
            j = best12[i]; v = val12[i];
            keep = (best21[j] == i);
            if (min_cossim > 0.f) keep = keep && (v > min_cossim);
        }
        const u64 m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (keep) {
            const int o = off + before;
            idx1[o] = i; idx2[o] = j;
            const float cd = 1.0f - v;
            dist[o] = sqrtf(2.0f * cd);
        }
        __syncthreads();
        if (t == 0) { int s = 0; for (int w = 0; w < 16; ++w) s += wsum[w]; base += s; }
        __syncthreads();
    }
    if (t == 0) *n_matches = base;
}

// dense integer metric: fp32 difference, fp64 square-accumulate, fp32 * 512, truncate
__global__ __launch_bounds__(256)
void k_dist_i32(const float* __restrict__ d1, int n1, const float* __restrict__ d2, int n2, int32_t* __restrict__ out) {
    __shared__ float sa[64 * 65];
    __shared__ float sb[64 * 65];
    const int t = threadIdx.x;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int e = t; e < 64 * 64; e += 256) {
        const int r = e >> 6, k = e & 63;
        sa[r * 65 + k] = (r0 + r < n1) ? d1[(size_t)(r0 + r) * 64 + k] : 0.f;
        sb[r * 65 + k] = (c0 + r < n2) ? d2[(size_t)(c0 + r) * 64 + k] : 0.f;
    }
    __syncthreads();
    const int tx = t & 15, ty = t >> 4;       // 16 x 16 threads, 4 x 4 outputs each
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int k = 0; k < 64; ++k) {
        float av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) av[a] = sa[(ty + 16 * a) * 65 + k];
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[b] = sb[(tx + 16 * b) * 65 + k];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double df = (double)(av[a] - bv[b]);
                acc[a][b] = fma(df, df, acc[a][b]);
            }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = r0 + ty + 16 * a, c = c0 + tx + 16 * b;
            if (r < n1 && c < n2) {
                const float nd = (float)acc[a][b];
                out[(size_t)r * n2 + c] = (int)(nd * 512.0f);
            }
        }
}

// ---------------------------------------------------------------------------------------
static hipError_t ensure(void** p, size_t* cap, size_t need_bytes) {
    if (*cap >= need_bytes && *p) return hipSuccess;
    if (*p) { hipError_t e = hipFree(*p); if (e != hipSuccess) return e; *p = nullptr; }
    hipError_t e = hipMalloc(p, need_bytes);
    if (e == hipSuccess) *cap = need_bytes;
    return e;
}

hipError_t launch_mnn(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, float min_cossim,
                      int* idx1, int* idx2, float* dist, int* n_matches) {
    hipError_t e;
    if (n1 <= 0 || n2 <= 0) return hipMemsetAsync(n_matches, 0, sizeof(int), c->stream);
    const int nbR = (n1 + MT - 1) / MT, nbC = (n2 + MT - 1) / MT;
    const int n1pad = nbR * MT, n2pad = nbC * MT;
    MatchWs& w = c->mws;
    const size_t need_part = (size_t)nbC * n1pad + (size_t)nbR * n2pad;
    if (w.cap_part < need_part) {
        if (w.partR) hipFree(w.partR);
        if ((e = hipMalloc((void**)&w.partR, need_part * sizeof(u64))) != hipSuccess) return e;
        w.cap_part = need_part;
    }
    w.partC = w.partR + (size_t)nbC * n1pad;
    const size_t need_best = (size_t)n1pad + n2pad;
    if (w.cap_best < need_best) {
        if (w.best12) hipFree(w.best12);
        if ((e = hipMalloc((void**)&w.best12, need_best * (2 * sizeof(int) + sizeof(float)))) != hipSuccess) return e;
        w.cap_best = need_best;
    }
    w.val12 = (float*)(w.best12 + w.cap_best);
    w.best21 = (int*)(w.val12 + w.cap_best);

    const size_t need_norm = ((size_t)n1 + n2) * 64;
    if (w.cap_norm < need_norm) {
        if (w.norm1) hipFree(w.norm1);
        if ((e = hipMalloc((void**)&w.norm1, need_norm * sizeof(float))) != hipSuccess) return e;
        w.cap_norm = need_norm;
    }
    w.norm2 = w.norm1 + (size_t)n1 * 64;
    hipLaunchKernelGGL(k_rownorm, dim3((n1 + n2 + 15) / 16 + 1), dim3(256), 0, c->stream, d1, n1, d2, n2, w.norm1, w.norm2);
    bool armed = ktimer_begin(c, XFH_K_MNN_GEMM, -1);
    hipLaunchKernelGGL(k_mnn_gemm, dim3(nbC, nbR), dim3(256), 0, c->stream, (const float*)w.norm1, n1, (const float*)w.norm2, n2,
                       w.partR, w.partC, n1pad, n2pad);
    ktimer_end(c, armed);
    const int nthr = n1pad + n2pad;
    hipLaunchKernelGGL(k_mnn_reduce, dim3((nthr + 255) / 256), dim3(256), 0, c->stream, w.partR, w.partC, nbR, nbC,
                       n1, n2, n1pad, n2pad, w.best12, w.val12, w.best21);
    hipLaunchKernelGGL(k_mnn_final, dim3(1), dim3(1024), 0, c->stream, w.best12, w.val12, w.best21, n1, min_cossim,
                       idx1, idx2, dist, n_matches);
    return hipGetLastError();
}

hipError_t launch_dist_i32(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, int32_t* out) {
    if (n1 <= 0 || n2 <= 0) return hipSuccess;
    bool armed = ktimer_begin(c, XFH_K_DIST_I32, -1);
    hipLaunchKernelGGL(k_dist_i32, dim3((n2 + 63) / 64, (n1 + 63) / 64), dim3(256), 0, c->stream, d1, n1, d2, n2, out);
    ktimer_end(c, armed);
    return hipGetLastError();
}
