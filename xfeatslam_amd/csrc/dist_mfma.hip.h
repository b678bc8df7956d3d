// dist_mfma.hip.h -- k_dist_mfma: the dense table of ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:2246-2247) on gfx950.
// Included by kernels_match.hip (the product instance, DBG = 0) and tools/probes/dist_probe.hip (phases switched off one by one, timing only).
#pragma once
#include "common.h"

// ---- k_dist_mfma: dense table of ORBmatcher::DescriptorDistance (ORBmatcher.cc:2246-2247) -----------------------
//     out[i][j] = (int)(float(sum_k double(a_k - b_k)^2) * 512)          (cv::norm NORM_L2SQR: fp32 difference, fp64 accumulate)
// The exact expression is fp64 VALU work ten times above anything that bounds the table (64 MB written at 4096^2: 8.4 us of HBM;
// 2.1 GFLOP of f32 MFMA: 13.7 us).  So the bulk goes through the matrix cores: v = 512 (|a|^2 + |b|^2) - 1024 <a, b> with the dot
// product from v_mfma_f32_32x32x2_f32.  Round 5: the test "is floor(v) already the reference's integer" is packed fp32 arithmetic
// (rounds 2-4 did it in fp64: ~20 double-precision instructions and three branches per output; on gfx950 every VALU instruction
// takes its cycles from the MFMAs' pipe, and the kernel sat at 67 us):
//     c = fl(ca[row] + cb[col]),  ca = fl(512 |a|^2) from the fp64 norm;      v32 = fma(-1024, dot, c)
//     |v32 - 512 float(s)| <= E = ea[row] + eb[col],   ea = 512 |a|^2 (72 * 2^-24 + 2^-20)
//        64 * 2^-24 (|a|^2+|b|^2) * 512      fp32 fma chain of the dot product (gamma_64 |a||b| <= gamma_64 (|a|^2+|b|^2)/2, times 2)
//        2^-22 v                             fp32 rounding of each difference (2^-23 relative on the sum) and of float(s) (2^-24)
//        2^-23 (|a|^2+|b|^2) * 512 + 2^-24 v the roundings of ca, cb, c and of the final fma            (4 of the 72 units are slack)
//        the two terms in v (together < 2^-21 |v32|) are bounded through |v32| <= 2 c: 2^-20 * 512 (|a|^2+|b|^2), so that E is a sum
//        of a row part and a column part and costs half a packed add per output
//        + 2^-23 ABSOLUTE in each of ea, eb (round 6): the test itself rounds -- u = fl(fl(0.5 - ea) - eb) and t = fl(fract - 0.5),
//        up to 3 * 2^-26 together -- and for rows of norm below ~2e-3 a purely relative E fell under 2^-26: u rounded to exactly 0.5 and a v32 of
//        -1e-10 between two identical tiny rows (fract clamps to 1 - 2^-24, t = 0.5 - 2^-24 < u) passed as "safe" with floor(v32) = -1 against the
//        reference's 0.  With the absolute term u <= 0.5 - 2^-22, so every value within 2^-22 of an integer (the clamp included) is marked.
// Wherever the fractional part of v32 is further than E from 0 and 1, floor(v32) IS the reference's integer (v32 >= 2^23 has no
// fractional part, infinities and NaN give NaN: they fail the test).  The other entries (about 1 % for unit descriptors; every entry
// that is an exact integer, e.g. against zero-padded rows) are collected per wave into a dense list (LDS) and recomputed with the exact
// expression from the tiles still in LDS -- one 64-term fp64 chain per LANE of a full wave instead of per marked bit of a sparse mask.
// Every output is stored once in the bulk pass without a branch (tiles inside the table) and the listed ones are overwritten afterwards
// (behind s_waitcnt vmcnt(0): the first store has reached L2).  512 threads per 128 x 128 tile (a wave: 32 x 64), two tiles per CU:
// four waves per SIMD hide each other's staging and LDS latencies.
// Identical integers by construction; the C oracle (sequential fp64) is the checker in tests/test_gpu_match.py::test_distance_i32_exact.
#define DT 128          // rows of a tile (one d1 panel, resident in LDS while the workgroup walks its column tiles)
#define DTC 64          // columns of a tile
#define DNT 4           // column tiles per workgroup of a table large enough to fill the GPU that way (dist_tiles_per_block)
#define DLDK 68         // padded LDS row (floats)
#define DLIST 512       // entries of a workgroup's fix-up list per round (= one per thread)
// one compare folded into a bit mask: m = 2 m + (|t| >= u), i.e. "not safe" enters at bit 0 and earlier outputs move up
__device__ __forceinline__ void dist_mark(unsigned& m, float t, float u) {
    asm volatile("v_cmp_nlt_f32 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(t), "v"(u) : "vcc");
}
__device__ __forceinline__ int dist_floor_i32(float v) {          // (int)floorf(v) in one instruction
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
// store to (wave-uniform 64-bit row pointer in scalar registers) + (32-bit lane offset): no vector instruction spent on the address
// (left to the compiler, the stores of a lane walk a 64-bit vector pointer: one v_lshl_add_u64 each, on the MFMAs' pipe)
__device__ __forceinline__ void dist_store(const char* row_ptr, unsigned lane_off, int val) {
    asm volatile("global_store_dword %0, %1, %2" :: "v"(lane_off), "v"(val), "s"(row_ptr) : "memory");
}
// bulk pass of one wave: 32 rows x 32 columns.  acc[r] = < row wr*32 + (r&3) + 8*(r>>2) + 4*h , column wc*32 + i >
template <bool FULL, bool NOSTORE = false>
__device__ __forceinline__ unsigned dist_bulk(const f32x16& acc, const f32x4 (&c4)[4], const f32x4 (&u4)[4], float cb, float eb,
                                              int wr, int wc, int i, int h, int row_base, int col_base, int n1, int n2, int32_t* __restrict__ out) {
    unsigned m = 0u;                                  // output r ends at bit 15 - r
    // a store's address = wave-uniform row pointer (scalar registers) + a 32-bit lane offset that is the same for all 16 outputs
    const unsigned lane_off = (unsigned)((4 * h) * n2 + i) * 4u;
    const char* const tile = (const char*)(out + (size_t)(row_base + wr * 32) * n2 + col_base + wc * 32);
    const int col = col_base + wc * 32 + i;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 a4 = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        const f32x4 v = __builtin_elementwise_fma(f32x4{-1024.f, -1024.f, -1024.f, -1024.f}, a4, c4[q] + cb);
        const f32x4 u = u4[q] - eb;                   // 0.5 - E
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rl = wr * 32 + 8 * q + 4 * h + j;
            // v_fract_f32 = v - floor(v) (exact; clamped below 1, NaN for infinities and NaN): |fr - 0.5| < 0.5 - E  <=>  E < fr < 1 - E
            const float tj = __builtin_amdgcn_fractf(v[j]) - 0.5f;
            const char* rowp = tile + (size_t)(8 * q + j) * n2 * 4;
            if (FULL) {
                dist_mark(m, tj, u[j]);
                if (!NOSTORE) dist_store(rowp, lane_off, dist_floor_i32(v[j])); else if (v[j] == 123.456f) dist_store(rowp, lane_off, 1);
            } else {
                const bool inb = (row_base + rl < n1) & (col < n2);
                dist_mark(m, inb ? tj : 0.f, inb ? u[j] : 1.f);          // outside the table: never marked (0 < 1)
                if (inb) dist_store(rowp, lane_off, dist_floor_i32(v[j]));
            }
        }
    }
    return m;
}
// LANES lanes hold one row of a tile (lane `sub`: the 64 / LANES consecutive elements from sub * 64 / LANES on, i.e. whole groups of 8) -> LDS in the
// MFMA's k order (element e of a group of 8 at 4 (e & 1) + (e >> 1)), its 512 |x|^2 and its share of the error bound (fp64 norm; any summation order
// will do: the bound carries 2^-23 for it).  Whole groups per lane: no exchange between lanes, and log2(LANES) shuffle rounds for the norm.
template <int LANES, int DBG>
__device__ __forceinline__ void dist_stage_row(const f32x4 (&v)[16 / LANES], int rl, int sub, float* sT, float* sC, float* sE) {
    constexpr int NV = 16 / LANES;                       // f32x4 pieces per lane; NV / 2 groups of 8
    double ss = 0.0;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        ss = fma((double)v[p].x, (double)v[p].x, ss); ss = fma((double)v[p].y, (double)v[p].y, ss);
        ss = fma((double)v[p].z, (double)v[p].z, ss); ss = fma((double)v[p].w, (double)v[p].w, ss);
    }
    if (!(DBG & 8)) {
#pragma unroll
        for (int d = 1; d < LANES; d <<= 1) ss += __shfl_xor(ss, d);
    }
    if (sub == 0) {
        sC[rl] = (float)(512.0 * ss);
        sE[rl] = (float)(512.0 * (72.0 * 5.9604644775390625e-8 + 9.5367431640625e-7) * 1.00002 * ss + 1.1920928955078125e-7);      // rounded up: the bound stays a bound; + 2^-23 absolute (header)
    }
#pragma unroll
    for (int g = 0; g < NV / 2; ++g) {                   // group of 8 = pieces 2g (e0..e3), 2g+1 (e4..e7) -> (e0 e2 e4 e6), (e1 e3 e5 e7)
        float* dst = sT + rl * DLDK + (sub * (NV / 2) + g) * 8;
        *(f32x4*)dst = f32x4{v[2 * g].x, v[2 * g].z, v[2 * g + 1].x, v[2 * g + 1].z};
        *(f32x4*)(dst + 4) = f32x4{v[2 * g].y, v[2 * g].w, v[2 * g + 1].y, v[2 * g + 1].w};
    }
}
// Workgroup = one d1 panel of 128 rows (staged once) x up to DNT column tiles of 64 d2 rows, walked one after the other.  The global loads of tile
// k + 1 are issued before the MFMAs of tile k and go to the OTHER half of a double-buffered sB after its bulk pass: one barrier per tile, no memory
// latency exposed from the second tile on, and the stores of tile k drain under the arithmetic of tile k + 1.  The entries that need the exact
// expression are only LISTED while the tiles go by (per workgroup, LDS) and recomputed after every second tile, while both halves of sB still hold
// their d2 rows: one 64-term fp64 chain per thread on dense lanes, everything read from LDS, one latency-bound phase per two tiles.  (Rounds 2-4 and the first
// form of round 5: one 128 x 128 tile per workgroup, two workgroups per CU in lock step -- staging, MFMAs, stores and fix-up followed each other
// with nothing to overlap them: 12 + 14 + 8 + 4 us.)  512 threads = 8 waves as 4 (rows) x 2 (columns), a wave owns 32 x 32 outputs of a tile.
template <int DBG>          // probes only: 1 = no exact fix-up, 2 = no bulk epilogue either, 4 = no MFMAs, 8 = no norm shuffles (bits combine)
__global__ __launch_bounds__(512, 4)        // HIP: the second figure is waves per SIMD -- 4 = two workgroups of 8 waves per CU, at most 128 VGPRs
void k_dist_mfma(const float* __restrict__ d1, int n1, const float* __restrict__ d2, int n2, int32_t* __restrict__ out, int nt) {
    __shared__ __attribute__((aligned(16))) float sA[DT * DLDK];
    __shared__ __attribute__((aligned(16))) float sB[2][DTC * DLDK];
    __shared__ __attribute__((aligned(16))) float sCa[DT], sEa[DT], sCb[2][DTC], sEb[2][DTC];      // 512 |x|^2 and the row's share of the error bound
    __shared__ unsigned sList[DLIST];                      // (row in the panel << 16 | column in the workgroup's strip) of the entries to recompute exactly
    __shared__ int sCnt[3];                                // entries listed; "lanes still hold entries" raised while listing an even / an odd column tile
    const int t = threadIdx.x;
    const int row_base = blockIdx.y * DT;
    const int col0 = blockIdx.x * (nt * DTC);
    const int ntile = (n2 - col0 + DTC - 1) / DTC < nt ? (n2 - col0 + DTC - 1) / DTC : nt;        // >= 1 by the grid
    if (t < 3) sCnt[t] = 0;
    if ((DBG & 64) && (blockIdx.y * gridDim.x + blockIdx.x) * 2 >= gridDim.x * gridDim.y) __builtin_amdgcn_s_sleep(127);
    if ((DBG & 128) && (blockIdx.x & 1)) __builtin_amdgcn_s_sleep(127);
    f32x4 vb[2];                                            // this thread's piece of the NEXT column tile: 8 lanes per row, row t >> 3
    auto load_b = [&](int kt) {
        const int row = col0 + kt * DTC + (t >> 3);
        vb[0] = vb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < n2) { const f32x4* src = (const f32x4*)(d2 + (size_t)row * 64 + (t & 7) * 8); vb[0] = src[0]; vb[1] = src[1]; }
    };
    load_b(0);
    {
        f32x4 va[4];                                        // d1 panel: 4 lanes per row, row t >> 2; in flight together with the first column tile
        const int row = row_base + (t >> 2);
#pragma unroll
        for (int p = 0; p < 4; ++p) va[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < n1) {
            const f32x4* src = (const f32x4*)(d1 + (size_t)row * 64 + (t & 3) * 16);
#pragma unroll
            for (int p = 0; p < 4; ++p) va[p] = src[p];
        }
        dist_stage_row<4, DBG>(va, t >> 2, t & 3, sA, sCa, sEa);
    }
    dist_stage_row<8, DBG>(vb, t >> 3, t & 7, sB[0], sCb[0], sEb[0]);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, i = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;                // wave: rows wr*32 .. +31, columns wc*32 .. +31 of the tile
    const float* pa = sA + (wr * 32 + i) * DLDK + 4 * h;
    const f32x16 Z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x4 c4[4], u4[4];                                     // the lane's 16 rows: 512 |a|^2 and 0.5 - (the rows' share of E)
#pragma unroll
    for (int q = 0; q < 4; ++q) {                          // rows 8q + 4h + (0..3) of the wave's 32
        c4[q] = *(const f32x4*)(sCa + wr * 32 + 8 * q + 4 * h);
        u4[q] = 0.5f - *(const f32x4*)(sEa + wr * 32 + 8 * q + 4 * h);
    }
    // List a wave's marked entries of column tile kt: the wave reserves a range of the workgroup's list (one LDS atomic), a lane's place is the exclusive
    // prefix of the counts over the lanes; what does not fit stays in `mask` and raises the overflow flag.  About 90 of a tile's 8192 entries for unit descriptors.
    auto list_marked = [&](unsigned& mask, int kt) {
        if (!__ballot(mask != 0u)) return;
        const int cnt = __popc(mask);
        int pre = cnt;                                       // inclusive prefix over the lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(pre, d); if (lane >= d) pre += o; }
        const int wtotal = __shfl(pre, 63);
        int base = 0;
        if (lane == 63) base = atomicAdd(&sCnt[0], wtotal);
        base = __shfl(base, 63);
        pre += base - cnt;
        while (mask && pre < DLIST) {
            const int bit = 31 - __builtin_clz(mask); mask &= ~(1u << bit);
            const int r = 15 - bit;                          // dist_bulk's bit order
            const int rl = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, cs = kt * DTC + wc * 32 + i;
            sList[pre++] = ((unsigned)rl << 16) | (unsigned)cs;
        }
        // the list is full: flush, then list the rest.  The flag of tile kt's PARITY: a wave that has already left tile kt - 1 (an even, non-flush tile ends
        // without a barrier behind its loop condition) raises the other flag than the one a slower wave of the workgroup is still reading there
        if (__ballot(mask != 0u) && lane == 0) atomicOr(&sCnt[1 + (kt & 1)], 1);
    };
    for (int kt = 0; kt < ntile; ++kt) {
        const int col_base = col0 + kt * DTC, b = kt & 1;
        if (kt + 1 < ntile) load_b(kt + 1);                // in flight through the MFMAs and the bulk pass of this tile
        const float* pb = sB[b] + (wc * 32 + i) * DLDK + 4 * h;
        f32x16 acc;
        if (DBG & 4) acc = Z16 + pa[0] + pb[0];
        else {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 a0 = *(const f32x4*)(pa + g * 8);
                const f32x4 b0 = *(const f32x4*)(pb + g * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], (g | j) ? acc : Z16, 0, 0, 0);
            }
        }
        XFH_MFMA_SETTLE();                                  // common.h: the epilogue branches
        unsigned mask;
        const float cb = sCb[b][wc * 32 + i], eb = sEb[b][wc * 32 + i];
        if (DBG & 2) { mask = 0u; if (acc[3] + acc[5] == 123.456f) out[t] = 1; }
        else if (row_base + DT <= n1 && col_base + DTC <= n2) mask = dist_bulk<true, (DBG & 32) != 0>(acc, c4, u4, cb, eb, wr, wc, i, h, row_base, col_base, n1, n2, out);
        else mask = dist_bulk<false>(acc, c4, u4, cb, eb, wr, wc, i, h, row_base, col_base, n1, n2, out);
        // list the marked entries (list_marked above)
        if (!(DBG & 1)) list_marked(mask, kt);
        // The exact fix-up runs after every SECOND tile (and after the last one): then both halves of sB still hold the d2 rows its entries name (tile kt in
        // sB[b], tile kt - 1 in the other half), so the chains read everything from LDS.  On those tiles the next column tile is staged AFTER the fix-up; on
        // the others right here (its half's last readers were the MFMAs of tile kt - 1: a barrier ago).
        const bool flush_tile = (kt & 1) || kt + 1 == ntile;               // block-uniform
        if (!flush_tile) dist_stage_row<8, DBG>(vb, t >> 3, t & 7, sB[b ^ 1], sCb[b ^ 1], sEb[b ^ 1]);        // (an even tile that is not the last one has a successor)
        __syncthreads();
        // ---- exact expression for the listed entries (block-uniform: a flush tile, or the list has overflowed -- then it holds entries of this tile only, the
        // previous flush having emptied it).  One entry per thread: both rows from LDS (element k = 8g + 2j + hh of a row sits at 8g + 4hh + j), one fp64 fma
        // chain in k order = the oracle's.
        while (!(DBG & 1) && (flush_tile || sCnt[1 + b]) && sCnt[0] > 0) {
            const int total = sCnt[0], more = sCnt[1 + b];
            const int n = total < DLIST ? total : DLIST;
            int fix_val = 0; size_t fix_at = 0;
            if (t < n) {
                const unsigned code = sList[t];
                const int rl = (int)(code >> 16), cs = (int)(code & 0xFFFFu);          // cs = column inside the workgroup's strip = tile * 64 + column of the tile
                const float* ra = sA + rl * DLDK;
                const float* rb = sB[(cs >> 6) & 1] + (cs & 63) * DLDK;
                double s = 0.0;
#pragma unroll 2
                for (int g = 0; g < 8; ++g) {
                    const f32x4 a0 = *(const f32x4*)(ra + g * 8), a1 = *(const f32x4*)(ra + g * 8 + 4);
                    const f32x4 b0 = *(const f32x4*)(rb + g * 8), b1 = *(const f32x4*)(rb + g * 8 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double df0 = (double)(a0[j] - b0[j]); s = fma(df0, df0, s);
                        const double df1 = (double)(a1[j] - b1[j]); s = fma(df1, df1, s);
                    }
                }
                const float nd = (float)s;
                fix_val = (int)(nd * 512.0f); fix_at = (size_t)(row_base + rl) * n2 + col0 + cs;
            }
            // An entry is overwritten by whichever thread got it from the list, i.e. by another wave than the one whose bulk pass stored there first:
            // every wave waits until its bulk stores have reached L2 (vmcnt(0) -- by now they have; the prefetched loads of the next tile are older), then
            // the barrier, then the exact values go out.
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();                                 // also: everybody has read the counters and the list
            if (t < n) out[fix_at] = fix_val;
            if (t == 0) { sCnt[0] = 0; sCnt[1 + b] = 0; }
            __syncthreads();
            if (!more) break;
            list_marked(mask, kt);                          // the lanes that kept entries list them now
            __syncthreads();
        }
        if (flush_tile && kt + 1 < ntile) {                 // every thread is done with both halves of sB
            dist_stage_row<8, DBG>(vb, t >> 3, t & 7, sB[b ^ 1], sCb[b ^ 1], sEb[b ^ 1]);
            __syncthreads();
        }
    }
}
// column tiles per workgroup: DNT when that still leaves two workgroups per CU, fewer for small tables (their time is one workgroup's latency)
static inline int dist_tiles_per_block(int n1, int n2, int num_cu) {
    int nt = DNT;
    while (nt > 1 && (long long)((n2 + nt * DTC - 1) / (nt * DTC)) * ((n1 + DT - 1) / DT) < 2LL * num_cu) nt >>= 1;
    return nt;
}
