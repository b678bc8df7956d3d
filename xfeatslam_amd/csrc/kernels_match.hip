// kernels_match.hip -- descriptor matching on gfx950.
//
//   k_rownorm     : F::normalize of both descriptor sets (ORBmatcher.cc:358-359), stored k-permuted.
//   k_mnn_gemm    : cosine-similarity GEMM (N1 x 64) . (64 x N2) on v_mfma_f32_32x32x2_f32 with the
//                   first level of the row / column arg-max (:363-368) fused into the epilogue:
//                   value maxima over small index groups, merged with 64-bit atomic max.
//   k_mnn_fix     : second level: names the member of the winning group by recomputing its few dot
//                   products bit-identically, and does the mutual check (:372) + min_cossim gate.
//   k_mnn_final   : ordered compaction and distances (:371-403).
//   k_dist_i32    : dense (int)(512 * ||a-b||^2), ORBmatcher::DescriptorDistance (:2246-2247).
//   k_best2_csr   : best / second-best distance over candidate lists (the SearchBy* inner loop, :75-119).
//   k_distinctive_csr : MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:329-403), one wave per map point.
//
// Numerics: normalised rows and the 64-term dot products are bit-identical to the oracle
// (fp64 sum of squares -> fp32 sqrt/max/div; one fp32 fma chain in k order, which is what
// the f32 MFMA computes), so the arg-max decisions including ties agree by construction:
// ties resolve to the lowest index (packed key ordered(value) << 32 | ~group, then the first
// member of the group that reaches the value).
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

// k_mnn_gemm: one workgroup = 128 rows of d1 against one block of TPW*128 = 256 rows of d2.
//   - the second d2 tile is prefetched into registers while the first one is on the MFMAs, so
//     only the first global load of a workgroup is exposed;
//   - grid = (ceil(n2/256), ceil(n1/128)): 512 workgroups at 4096 x 4096, all resident at 2 per CU;
//   - per tile, every wave owns 64 x 64 outputs as 2 x 2 MFMA tiles (four independent chains).
//
// Two-level exact arg-max.  On gfx950 the f32 MFMA and the ordinary VALU instructions of a SIMD
// do not overlap (tools/probes/mfma_probe.hip), so every epilogue instruction costs matrix time;
// carrying an index next to every running maximum (compare + two selects per value and direction)
// held the previous kernel at 50 % of the MFMA peak.  Here the epilogue only takes VALUE maxima
// (v_max3_f32: half an instruction per value and direction) over small candidate groups that are
// fixed by the lane / wave position:
//     row i of d1   -> best value over the 4 consecutive d2 rows   4*gc .. 4*gc+3   (gc: column group)
//     row j of d2   -> best value over the 16 consecutive d1 rows 16*gr .. 16*gr+15 (gr: row group)
// and merges (value, group) keys with 64-bit atomic max (order independent => deterministic).  The
// d1 / d2 rows are assigned to MFMA rows / columns through a permutation of the LDS tile rows such
// that every lane's candidates are consecutive and the groups ascend with the index, so "largest
// value, then lowest group, then first member equal to that value" is exactly "first index of the
// maximum".  k_mnn_fix recomputes the few candidate dot products (same fp32 fma chain in k order as
// the MFMA, hence the same bits) to name the member.
//
// Tile-row permutations (block-local indices):
//     d1: MFMA row  wr*64 + rt*32 + (r&3) + 8*(r>>2) + 4h   <->  d1 row  wr*64 + h*32 + rt*16 + r
//     d2: MFMA col  wc*64 + ct*32 + i  of tile `tile`       <->  d2 row  i*8 + wc*4 + tile*2 + ct
#define TPW 2
#define MNN_NC 4        // candidates per d1 row (see k_mnn_fix)
#define TLD 68          // row stride (floats) of the per-wave transposition scratch
__global__ __launch_bounds__(256, 2)
void k_mnn_gemm(const float* __restrict__ d1, int n1, const float* __restrict__ d2, int n2,
                u64* __restrict__ bestR, u64* __restrict__ bestC) {
    // d1/d2: normalised, k-permuted rows from k_rownorm; bestR/bestC: packed (value, group) keys
    __shared__ __attribute__((aligned(16))) float smem[2 * MT * LDK + 2 * 4 * 64 * 2];
    float* sA = smem;
    float* sB = smem + MT * LDK;
    u64* sRow = (u64*)(smem + 2 * MT * LDK);     // [4 waves][64]
    u64* sCol = sRow + 4 * 64;                    // [4 waves][64]
    const int t = threadIdx.x;
    const int bx2 = blockIdx.x, by = blockIdx.y;
    const int sub = t & 15, r0 = t >> 4;
    const int row_base = by * MT, col_base = bx2 * (TPW * MT);
    // staging: thread (r0 = t>>4, sub = t&15) moves 16 bytes of tile row rho = p*16 + r0, p = 0..7.  With the
    // permutations above the global row splits into a per-thread part and a compile-time part of p:
    //   d1 row = row_base + [((r0>>2)&1)*32 + (r0&3) + 4*(r0>>3)] + [(p>>2)*64 + ((p>>1)&1)*16 + (p&1)*8]
    //   d2 row = col_base + [r0*8] + [(p&1)*128 + (p>>2)*4 + ((p>>1)&1) + tile*2]
    const bool full = (row_base + MT <= n1) && (col_base + TPW * MT <= n2);      // block-uniform
    const int ra0 = row_base + ((r0 >> 2) & 1) * 32 + (r0 & 3) + 4 * (r0 >> 3), rb0 = col_base + r0 * 8;
    const float* gA = d1 + (size_t)ra0 * 64 + sub * 4;
    const float* gB = d2 + (size_t)rb0 * 64 + sub * 4;
    const f32x4 Z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 vb[8];
    {
        f32x4 va[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int ca = (p >> 2) * 64 + ((p >> 1) & 1) * 16 + (p & 1) * 8, cb = (p & 1) * 128 + (p >> 2) * 4 + ((p >> 1) & 1);
            va[p] = (full || ra0 + ca < n1) ? *(const f32x4*)(gA + ca * 64) : Z4;
            vb[p] = (full || rb0 + cb < n2) ? *(const f32x4*)(gB + cb * 64) : Z4;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            *(f32x4*)(sA + (p * 16 + r0) * LDK + sub * 4) = va[p];
            *(f32x4*)(sB + (p * 16 + r0) * LDK + sub * 4) = vb[p];
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 8; ++p) {                 // tile 1 -> registers
        const int cb = (p & 1) * 128 + (p >> 2) * 4 + ((p >> 1) & 1) + 2;
        vb[p] = (full || rb0 + cb < n2) ? *(const f32x4*)(gB + cb * 64) : Z4;
    }

    const int wave = t >> 6, lane = t & 63, i = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const float NEG = -__builtin_huge_valf();
    const float* pa = sA + (wr * 64 + i) * LDK + 4 * h;
    const float* pb = sB + (wc * 64 + i) * LDK + 4 * h;
    const int grow_lane = row_base + wr * 64 + h * 32;        // + rt*16 + r
    const f32x16 Z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // running row maximum of this lane: slot q = rt*16 + r  <->  d1 row grow_lane + q
    float rbv[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) rbv[q] = NEG;

#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        f32x16 acc[2][2];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 a0 = *(const f32x4*)(pa + g * 8);
            const f32x4 a1 = *(const f32x4*)(pa + 32 * LDK + g * 8);
            const f32x4 b0 = *(const f32x4*)(pb + g * 8);
            const f32x4 b1 = *(const f32x4*)(pb + 32 * LDK + g * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // the first k step takes the literal zero as C: no accumulator clearing instructions
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], (g | j) ? acc[0][0] : Z16, 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], (g | j) ? acc[0][1] : Z16, 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], (g | j) ? acc[1][0] : Z16, 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], (g | j) ? acc[1][1] : Z16, 0, 0, 0);
            }
        }
        // ---- tile epilogue: acc[rt][ct][r] = <d1 row grow_lane + rt*16 + r, d2 row gcol0 + ct>
        const int gcol0 = col_base + i * 8 + wc * 4 + tile * 2;
        if (!full) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool vr = grow_lane + rt * 16 + r < n1;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        if (!(vr && gcol0 + ct < n2)) acc[rt][ct][r] = NEG;
                }
        }
        // rows: value maximum over this lane's (up to) 4 columns
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = rt * 16 + r;
                rbv[q] = fmaxf(fmaxf(rbv[q], acc[rt][0][r]), acc[rt][1][r]);
            }
        // columns: value maximum over the 16 rows of each (rt) group, key = (value, row group)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            float m[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                m[rt] = acc[rt][ct][0];
#pragma unroll
                for (int r = 1; r < 15; r += 2) m[rt] = fmaxf(fmaxf(m[rt], acc[rt][ct][r]), acc[rt][ct][r + 1]);
                m[rt] = fmaxf(m[rt], acc[rt][ct][15]);
            }
            const bool second = m[1] > m[0];                   // tie -> the lower row group
            const float mm = second ? m[1] : m[0];
            const unsigned gr = (unsigned)(grow_lane >> 4) + (second ? 1u : 0u);
            const u64 k = (mm > NEG) ? pack_key(mm, gr) : 0ull;
            const u64 kk = umax64(k, __shfl_xor(k, 32));
            if (lane < 32) sCol[wave * 64 + ct * 32 + lane] = kk;
        }
        __syncthreads();                   // sB is free, column keys are visible
        if (wr == 0) {
            const u64 k = umax64(sCol[wave * 64 + lane], sCol[(wave + 2) * 64 + lane]);
            const int col = col_base + (lane & 31) * 8 + wc * 4 + tile * 2 + (lane >> 5);
            if (k) __hip_atomic_fetch_max(bestC + col, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tile == 0) {
#pragma unroll
            for (int p = 0; p < 8; ++p) *(f32x4*)(sB + (p * 16 + r0) * LDK + sub * 4) = vb[p];
            __syncthreads();
        }
    }

    // ---- rows: transpose the 64 lanes x 32 slots through LDS (the tiles are dead; wave w owns
    // floats [w*32*TLD, (w+1)*32*TLD)), then every lane scans the 32 lanes of its half for slot q = i
    __syncthreads();                                   // all waves are done with sA / sB
    float* T = smem + wave * (32 * TLD);
#pragma unroll
    for (int q = 0; q < 32; ++q) T[q * TLD + lane] = rbv[q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float best = NEG; int bi = 0;
    {
        const float* src = T + i * TLD + h * 32;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 v = *(const f32x4*)(src + g * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const bool gt = v[e] > best; best = gt ? v[e] : best; bi = gt ? g * 4 + e : bi; }
        }
    }
    // lane (i, h) holds slot q = i of half h: d1 row row_base + wr*64 + h*32 + i = row_base + wr*64 + lane
    sRow[wave * 64 + lane] = (best > NEG) ? pack_key(best, (unsigned)(bx2 * 64 + bi * 2 + wc)) : 0ull;
    __syncthreads();
    if (wc == 0) {
        const u64 k = umax64(sRow[wave * 64 + lane], sRow[(wave + 1) * 64 + lane]);
        if (k) __hip_atomic_fetch_max(bestR + row_base + wr * 64 + lane, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// <a, b> over 64 elements as ONE fp32 fma chain in k order from 0 -- the arithmetic of the MFMA loop
// above.  a, b: k-permuted rows (element e of group g at 8g + 4(e&1) + (e>>1)).
__device__ __forceinline__ float dot64_chain(const float* __restrict__ a, const float* __restrict__ b) {
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const f32x4 a0 = *(const f32x4*)(a + g * 8), a1 = *(const f32x4*)(a + g * 8 + 4);
        const f32x4 b0 = *(const f32x4*)(b + g * 8), b1 = *(const f32x4*)(b + g * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc = fmaf(a0[j], b0[j], acc); acc = fmaf(a1[j], b1[j], acc); }
    }
    return acc;
}

// k_mnn_fix: second level of the arg-max and the mutual check.  Sixteen lanes per d1 row.
//   bestR[row] = (M, gc): the row maximum M sits in d2 rows NC*gc .. NC*gc+NC-1 -> lanes 0..NC-1
//   recompute those dot products, the first one equal to M is m12[row] (ORBmatcher.cc:367).
//   bestC[col] = (Mc, gr): m21[col] is the first d1 row of 16*gr .. 16*gr+15 whose dot product equals
//   Mc (:368).  row is a mutual match (:372) iff Mc == M, row lies in that group and no earlier row of
//   the group reaches Mc -- lane l recomputes <d1 row 16*gr + l, col> for the rows before `row` only.
__global__ __launch_bounds__(256)
void k_mnn_fix(const float* __restrict__ d1, int n1, const float* __restrict__ d2, int n2, int NC,
               const u64* __restrict__ bestR, const u64* __restrict__ bestC, float min_cossim,
               int* __restrict__ mcol, float* __restrict__ mval) {
    const int t = threadIdx.x, c = t & 15;
    const int row = blockIdx.x * 16 + (t >> 4);
    const u64 kr = (row < n1) ? bestR[row] : 0ull;
    const float M = ord2f((unsigned)(kr >> 32));
    const int gc = (int)(0xFFFFFFFFu - (unsigned)(kr & 0xFFFFFFFFull));
    const int col = gc * NC + c;
    const bool have = kr != 0ull && c < NC && col < n2;
    const float dv = have ? dot64_chain(d1 + (size_t)row * 64, d2 + (size_t)col * 64) : 0.f;
    unsigned eq = (have && dv == M) ? (1u << c) : 0u;
    eq |= __shfl_xor(eq, 1); eq |= __shfl_xor(eq, 2); eq |= __shfl_xor(eq, 4); eq |= __shfl_xor(eq, 8);
    int cs;
    if (eq) cs = __builtin_ctz(eq);
    else {      // cannot happen while the recomputation is bit-identical; stay deterministic anyway
        u64 k = have ? pack_key(dv, (unsigned)c) : 0ull;
        k = umax64(k, __shfl_xor(k, 1)); k = umax64(k, __shfl_xor(k, 2)); k = umax64(k, __shfl_xor(k, 4)); k = umax64(k, __shfl_xor(k, 8));
        cs = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)) & 15;
    }
    const int cstar = gc * NC + cs;
    bool mutual = false;
    if (kr != 0ull) {                                   // uniform over the 16 lanes of a row
        const u64 kc = bestC[cstar];
        const float Mc = ord2f((unsigned)(kc >> 32));
        const int gr = (int)(0xFFFFFFFFu - (unsigned)(kc & 0xFFFFFFFFull));
        if (kc != 0ull && Mc == M && (row >> 4) == gr) {
            const int r = gr * 16 + c;
            unsigned earlier = (r < row && dot64_chain(d1 + (size_t)r * 64, d2 + (size_t)cstar * 64) == Mc) ? 1u : 0u;
            earlier |= __shfl_xor(earlier, 1); earlier |= __shfl_xor(earlier, 2); earlier |= __shfl_xor(earlier, 4); earlier |= __shfl_xor(earlier, 8);
            mutual = earlier == 0u;
        }
    }
    if (min_cossim > 0.f) mutual = mutual && (M > min_cossim);
    if (c == 0 && row < n1) { mcol[row] = mutual ? cstar : -1; mval[row] = M; }
}

// one workgroup: ordered compaction (ascending idx1) of the mutual pairs and their distances
// (:371-403).  Thread t owns rows 4t..4t+3 of every 4096-row chunk, so one scan per chunk orders
// the output.
__global__ __launch_bounds__(1024)
void k_mnn_final(const int* __restrict__ mcol, const float* __restrict__ mval, int n1,
                 int* __restrict__ idx1, int* __restrict__ idx2, float* __restrict__ dist, int* __restrict__ n_matches) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n1; i0 += 4096) {
        int j[4]; float v[4]; int cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + t * 4 + q;
            j[q] = (i < n1) ? mcol[i] : -1;
            v[q] = (i < n1) ? mval[i] : 0.f;
            cnt += j[q] >= 0 ? 1 : 0;
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
            if (j[q] >= 0) {
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
hipError_t launch_mnn(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, float min_cossim,
                      int* idx1, int* idx2, float* dist, int* n_matches) {
    hipError_t e;
    if (n1 <= 0 || n2 <= 0) return hipMemsetAsync(n_matches, 0, sizeof(int), c->stream);
    MatchWs& w = c->mws;
    const size_t need_best = 2 * (size_t)n1 + n2;          // bestR[n1], bestC[n2], then mcol[n1] (int) + mval[n1] (float)
    if (w.cap_best < need_best) {
        if (w.bestR) hipFree(w.bestR);
        if ((e = hipMalloc((void**)&w.bestR, need_best * sizeof(u64))) != hipSuccess) return e;
        w.cap_best = need_best;
    }
    w.bestC = w.bestR + n1;
    int* mcol = (int*)(w.bestC + n2);
    float* mval = (float*)(mcol + n1);
    const size_t need_norm = ((size_t)n1 + n2) * 64;
    if (w.cap_norm < need_norm) {
        if (w.norm1) hipFree(w.norm1);
        if ((e = hipMalloc((void**)&w.norm1, need_norm * sizeof(float))) != hipSuccess) return e;
        w.cap_norm = need_norm;
    }
    w.norm2 = w.norm1 + (size_t)n1 * 64;
    hipLaunchKernelGGL(k_rownorm, dim3((n1 + n2 + 15) / 16 + 1), dim3(256), 0, c->stream, d1, n1, d2, n2, w.norm1, w.norm2, w.bestR, w.bestC);
    const int NC = MNN_NC;
    launch_k(c, XFH_K_MNN_GEMM, -1, k_mnn_gemm, dim3((n2 + TPW * MT - 1) / (TPW * MT), (n1 + MT - 1) / MT), dim3(256), 0,
             (const float*)w.norm1, n1, (const float*)w.norm2, n2, w.bestR, w.bestC);
    hipLaunchKernelGGL(k_mnn_fix, dim3((n1 + 15) / 16), dim3(256), 0, c->stream, (const float*)w.norm1, n1, (const float*)w.norm2, n2, NC,
                       (const u64*)w.bestR, (const u64*)w.bestC, min_cossim, mcol, mval);
    hipLaunchKernelGGL(k_mnn_final, dim3(1), dim3(1024), 0, c->stream, (const int*)mcol, (const float*)mval, n1,
                       idx1, idx2, dist, n_matches);
    return hipGetLastError();
}

hipError_t launch_dist_i32(xfh_ctx* c, const float* d1, int n1, const float* d2, int n2, int32_t* out) {
    if (n1 <= 0 || n2 <= 0) return hipSuccess;
    launch_k(c, XFH_K_DIST_I32, -1, k_dist_i32, dim3((n2 + 63) / 64, (n1 + 63) / 64), dim3(256), 0, d1, n1, d2, n2, out);
    return hipGetLastError();
}
