// kernels_match.hip -- descriptor matching on gfx950.
//
//   k_mnn_gemm    : cosine-similarity GEMM (N1 x 64) . (64 x N2) on v_mfma_f32_32x32x2_f32 with
//                   the row L2-normalisation fused into the tile load and the row / column
//                   arg-max fused into the epilogue (reference: the commented-out
//                   ORBmatcher::match, src/ORBmatcher.cc:358-368).
//   k_mnn_final   : mutual check, min_cossim gate, ordered compaction, distances (:371-403).
//   k_dist_i32    : dense (int)(512 * ||a-b||^2), ORBmatcher::DescriptorDistance (:2246-2247).
//
// Numerics: normalised rows and the 64-term dot products are bit-identical to the oracle
// (fp64 sum of squares -> fp32 sqrt/max/div; one fp32 fma chain in k order, which is what
// the f32 MFMA computes), so the arg-max decisions including ties agree by construction.
// Ties resolve to the lowest index through the packed key (ordered(value) << 32 | ~index).
#include "ctx.h"
#include <utility>
#include <stdlib.h>

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
               float* __restrict__ o1, float* __restrict__ o2, u64* __restrict__ bestR, u64* __restrict__ bestC) {
    const int t = threadIdx.x, sub = t & 15;
    int row = blockIdx.x * 16 + (t >> 4);
    const float* d; float* o;
    if (row < n1) { d = d1; o = o1; if (sub == 0) bestR[row] = 0ull; }       // arg-max keys start at "nothing"
    else { row -= n1; if (row >= n2) return; d = d2; o = o2; if (sub == 0) bestC[row] = 0ull; }   // 16-lane groups exit together
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

// reduce-scatter butterfly step on packed keys, expanded with an integer_sequence fold: with
// ordinary (pragma-unrolled) loops the compiler turns `up ? k[q+s] : k[q]` into a dynamically
// indexed array read, i.e. a select chain over the whole array.
template <int S, int N, int... Q>
__device__ __forceinline__ void bfly_step(u64 (&k)[N], bool up, std::integer_sequence<int, Q...>) {
    ((k[Q] = umax64(up ? k[Q + S] : k[Q], __shfl_xor(up ? k[Q] : k[Q + S], S))), ...);
}

template <int... Q>
__device__ __forceinline__ void pack_rows(u64 (&rk)[32], const float (&rbv)[32], const int (&rbc)[32], std::integer_sequence<int, Q...>) {
    ((rk[Q] = (rbv[Q] > -__builtin_huge_valf()) ? pack_key(rbv[Q], (unsigned)rbc[Q]) : 0ull), ...);
}

// k_mnn_gemm: one workgroup = 128 rows of d1 against TPW = 2 consecutive 128-row blocks of d2.
//   - the second d2 block is prefetched into registers while the first one is on the MFMAs, so
//     only the first global load of a workgroup is exposed;
//   - grid = (ceil(nbC/2), nbR): 512 workgroups at 4096 x 4096, all resident at 2 per CU;
//   - per tile, every wave owns 64 x 64 outputs as 2 x 2 MFMA tiles (four independent chains);
//   - column arg-max (over rows) is lane-local in the C/D layout; the row arg-max is carried
//     lane-locally across both tiles (value + column per row slot) and reduced across lanes ONCE
//     at the end with a packed-key butterfly.
// Measured on gfx950 (tools/probes/mfma_probe.hip, profiles/r01_gemm_notes.md): the f32 MFMA and
// the ordinary VALU instructions of a SIMD do not overlap -- every epilogue instruction costs
// matrix time -- which is why the arg-max is kept at 3 VALU per value and direction.  Variants
// with the epilogue interleaved between MFMAs, 8-wave and persistent 4-tile workgroups were
// measured slower (DESIGN.md "Match kernel: what was tried").
#define TPW 2
__global__ __launch_bounds__(256, 2)
void k_mnn_gemm(const float* __restrict__ d1, int n1, const float* __restrict__ d2, int n2,
                u64* __restrict__ bestR, u64* __restrict__ bestC, int nbC) {
    // d1/d2: normalised, k-permuted rows from k_rownorm; bestR/bestC: packed arg-max keys, merged with
    // 64-bit atomic max (order independent, so the result is deterministic)
    __shared__ __attribute__((aligned(16))) float smem[2 * MT * LDK + 2 * 4 * 64 * 2];
    float* sA = smem;
    float* sB = smem + MT * LDK;
    u64* sRow = (u64*)(smem + 2 * MT * LDK);     // [4 waves][64]
    u64* sCol = sRow + 4 * 64;                    // [4 waves][64]
    const int t = threadIdx.x;
    const int bx2 = blockIdx.x, by = blockIdx.y;
    const int sub = t & 15, r0 = t >> 4;
    const bool has_b1 = bx2 * 2 + 1 < nbC;
    f32x4 vb[8];
    {
        f32x4 va[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int ra = by * MT + p * 16 + r0, rb = bx2 * 2 * MT + p * 16 + r0;
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
    if (has_b1) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int rb = (bx2 * 2 + 1) * MT + p * 16 + r0;
            vb[p] = (rb < n2) ? *(const f32x4*)(d2 + (size_t)rb * 64 + sub * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    const int wave = t >> 6, lane = t & 63, i = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const float NEG = -__builtin_huge_valf();
    const int grow0 = by * MT + wr * 64;
    const float* pa = sA + (wr * 64 + i) * LDK + 4 * h;
    const float* pb = sB + (wc * 64 + i) * LDK + 4 * h;

    // running row best of this lane: slot q = rt*16 + r  <->  row rt*32 + (r&3) + 8*(r>>2) + 4h
    float rbv[32];
    int rbc[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) { rbv[q] = NEG; rbc[q] = 0; }

#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        if (tile == 1 && !has_b1) break;
        const int bx = bx2 * 2 + tile;
        const int gcol0 = bx * MT + wc * 64;
        const bool full = (by * MT + MT <= n1) && (bx * MT + MT <= n2);     // block-uniform
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
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
        // ---- tile epilogue: C/D layout = column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
        if (!full) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool vr = grow0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < n1;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        if (!(vr && gcol0 + ct * 32 + i < n2)) acc[rt][ct][r] = NEG;
                }
        }
        float cbv[2] = {NEG, NEG};
        int cbr[2] = {0, 0};
        const int c0 = gcol0 + i, c1 = gcol0 + 32 + i;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v0 = acc[rt][0][r], v1 = acc[rt][1][r];
                // columns: ascending row order + strict '>' keeps the lowest row among equals
                bool gt = v0 > cbv[0]; cbv[0] = gt ? v0 : cbv[0]; cbr[0] = gt ? lrow : cbr[0];
                gt = v1 > cbv[1]; cbv[1] = gt ? v1 : cbv[1]; cbr[1] = gt ? lrow : cbr[1];
                // rows: ascending column order (tile 0 before tile 1, c0 < c1) + strict '>'
                const int q = rt * 16 + r;
                gt = v0 > rbv[q]; rbv[q] = gt ? v0 : rbv[q]; rbc[q] = gt ? c0 : rbc[q];
                gt = v1 > rbv[q]; rbv[q] = gt ? v1 : rbv[q]; rbc[q] = gt ? c1 : rbc[q];
            }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const u64 k = (cbv[ct] > NEG) ? pack_key(cbv[ct], (unsigned)(grow0 + cbr[ct])) : 0ull;
            const u64 kk = umax64(k, __shfl_xor(k, 32));
            if (lane < 32) sCol[wave * 64 + ct * 32 + lane] = kk;
        }
        __syncthreads();                   // sB is free, column keys are visible
        if (wr == 0) {
            const u64 k = umax64(sCol[wave * 64 + lane], sCol[(wave + 2) * 64 + lane]);
            if (k) __hip_atomic_fetch_max(bestC + bx * MT + wc * 64 + lane, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tile == 0 && has_b1) {
#pragma unroll
            for (int p = 0; p < 8; ++p) *(f32x4*)(sB + (p * 16 + r0) * LDK + sub * 4) = vb[p];
            __syncthreads();
        }
    }

    // ---- rows: one packed-key reduce-scatter over the 32 lanes of each half --------------
    u64 rk[32];
    pack_rows(rk, rbv, rbc, std::make_integer_sequence<int, 32>{});
    bfly_step<16>(rk, (lane & 16) != 0, std::make_integer_sequence<int, 16>{});
    bfly_step<8>(rk, (lane & 8) != 0, std::make_integer_sequence<int, 8>{});
    bfly_step<4>(rk, (lane & 4) != 0, std::make_integer_sequence<int, 4>{});
    bfly_step<2>(rk, (lane & 2) != 0, std::make_integer_sequence<int, 2>{});
    bfly_step<1>(rk, (lane & 1) != 0, std::make_integer_sequence<int, 1>{});
    {
        // lane (i,h) now holds slot q = i: row (q>>4)*32 + (q&3) + 8*((q&15)>>2) + 4h
        const int q = i;
        const int lr = (q >> 4) * 32 + (q & 3) + 8 * ((q & 15) >> 2) + 4 * h;
        sRow[wave * 64 + lr] = rk[0];
    }
    __syncthreads();
    if (wc == 0) {
        const u64 k = umax64(sRow[wave * 64 + lane], sRow[(wave + 1) * 64 + lane]);
        if (k) __hip_atomic_fetch_max(bestR + by * MT + wr * 64 + lane, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// one workgroup: mutual check + gate + ordered compaction (ascending idx1).  Thread t owns rows
// 4t..4t+3 of every 4096-row chunk, so one ballot scan per chunk orders the output.
__global__ __launch_bounds__(1024)
void k_mnn_final(const u64* __restrict__ bestR, const u64* __restrict__ bestC, int n1, float min_cossim,
                 int* __restrict__ idx1, int* __restrict__ idx2, float* __restrict__ dist, int* __restrict__ n_matches) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n1; i0 += 4096) {
        u64 kr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int i = i0 + t * 4 + q; kr[q] = (i < n1) ? bestR[i] : 0ull; }
        int j[4]; float v[4]; u64 kc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            j[q] = (int)(0xFFFFFFFFu - (unsigned)(kr[q] & 0xFFFFFFFFull));
            v[q] = ord2f((unsigned)(kr[q] >> 32));
            kc[q] = kr[q] ? bestC[j[q]] : 0ull;
        }
        bool keep[4]; int cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + t * 4 + q;
            keep[q] = kr[q] != 0ull && (int)(0xFFFFFFFFu - (unsigned)(kc[q] & 0xFFFFFFFFull)) == i;
            if (min_cossim > 0.f) keep[q] = keep[q] && (v[q] > min_cossim);
            cnt += keep[q] ? 1 : 0;
        }
        // exclusive scan of cnt over the 1024 threads
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int off = base + incl - cnt;
        for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (keep[q]) {
                idx1[off] = i0 + t * 4 + q; idx2[off] = j[q];
                const float cd = 1.0f - v[q];
                dist[off] = sqrtf(2.0f * cd);
                ++off;
            }
        __syncthreads();
        if (t == 0) { int sacc = 0; for (int w = 0; w < 16; ++w) sacc += wsum[w]; base += sacc; }
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
    const int nbR = (n1 + MT - 1) / MT, nbC = (n2 + MT - 1) / MT, nbC2 = (nbC + TPW - 1) / TPW;
    MatchWs& w = c->mws;
    const size_t need_best = (size_t)n1 + n2;
    if (w.cap_best < need_best) {
        if (w.bestR) hipFree(w.bestR);
        if ((e = hipMalloc((void**)&w.bestR, need_best * sizeof(u64))) != hipSuccess) return e;
        w.cap_best = need_best;
    }
    w.bestC = w.bestR + n1;
    const size_t need_norm = ((size_t)n1 + n2) * 64;
    if (w.cap_norm < need_norm) {
        if (w.norm1) hipFree(w.norm1);
        if ((e = hipMalloc((void**)&w.norm1, need_norm * sizeof(float))) != hipSuccess) return e;
        w.cap_norm = need_norm;
    }
    w.norm2 = w.norm1 + (size_t)n1 * 64;
    hipLaunchKernelGGL(k_rownorm, dim3((n1 + n2 + 15) / 16 + 1), dim3(256), 0, c->stream, d1, n1, d2, n2, w.norm1, w.norm2, w.bestR, w.bestC);
    launch_k(c, XFH_K_MNN_GEMM, -1, k_mnn_gemm, dim3(nbC2, nbR), dim3(256), 0, (const float*)w.norm1, n1, (const float*)w.norm2, n2,
             w.bestR, w.bestC, nbC);
    hipLaunchKernelGGL(k_mnn_final, dim3(1), dim3(1024), 0, c->stream, (const u64*)w.bestR, (const u64*)w.bestC, n1, min_cossim,
                       idx1, idx2, dist, n_matches);
    return hipGetLastError();
}

hipError_t launch_dist_i32(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, int32_t* out) {
    if (n1 <= 0 || n2 <= 0) return hipSuccess;
    launch_k(c, XFH_K_DIST_I32, -1, k_dist_i32, dim3((n2 + 63) / 64, (n1 + 63) / 64), dim3(256), 0, d1, n1, d2, n2, out);
    return hipGetLastError();
}
