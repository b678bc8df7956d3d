// mnn_gemm.hip.h -- k_mnn_gemm_img: the cosine-similarity GEMM of ORBmatcher::match (reference src/ORBmatcher.cc:358-368)
// on v_mfma_f32_32x32x2_f32 with the first level of the row / column arg-max fused in (layout: mnn_layout.h).
// Included by kernels_mnn_gemm.hip (the product instance) and tools/probes/mnn_probe_gemm.hip (variants under
// measurement); both are compiled with -fno-honor-nans: under IEEE mode hipcc otherwise canonicalises every operand of
// a two-input fmaxf with an extra v_max_f32 x, x, x (236 instead of 148 maxima in the epilogue, and every epilogue
// instruction costs matrix time).  The hardware max still ignores a NaN operand; NaN descriptors are outside the contract.
#pragma once
#include "mnn_layout.h"

__device__ __forceinline__ float mnn_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// one 1-KB LDS-DMA: lanes read gsrc + lane*16 .. +16, LDS receives them at lds_dst + lane*16
__device__ __forceinline__ void mnn_dma1k(const float* gsrc_lane, float* lds_dst_wave) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave, 16, 0, 0);
}
#define MNN_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// k_mnn_gemm_img: one workgroup = one d1 panel x one d2 panel (256 x 256 similarities), 8 waves as 4 (wr) x 2 (wc);
// a wave owns 64 d1 rows x 128 d2 rows = 2 x 4 MFMA tiles of 32 x 32 (eight independent accumulator chains).
//
// Two-level exact arg-max.  On gfx950 the f32 MFMA issues on the SIMD's vector pipe, so every epilogue VALU
// instruction costs matrix time.  The epilogue therefore takes only VALUE maxima (v_max3_f32: half an instruction
// per value and direction) over candidate groups that are fixed by the lane position:
//     d1 row  -> best value over a group of 4 consecutive d2 rows    (key: value, d2 row / 4: the four rows one lane holds; found in two steps,
//                the maximum over 16 first, then which quarter of that group reaches it)
//     d2 row  -> best value over a group of 16 consecutive d1 rows   (key: value, d1 row / 16)
// as keys (ordered(value) << 32 | ~group), merged across waves through LDS and written to this block's plane:
//     partR[blockIdx.x][d1 row]   (ldr = rows per plane),   partC[blockIdx.y][d2 row]   (ldc)
// k_mnn_post takes the maximum over the planes: largest value, then lowest group = torch.max's "first index of the
// maximum" once it has named the first member of the group that reaches the value (it recomputes the 4 / <= 16 dot
// products with the MFMA's own arithmetic).  Every plane entry of a launched block is written (0 = no valid product),
// so nothing has to be cleared between calls.
// TAIL (probes only, tools/probes/mnn_tail_probe.hip): the key planes go out as agent-scope (write-through) stores and the workgroup then runs
// mnn_tail_hook<TAIL> -- the skeleton of a match finished INSIDE this launch (arrival counters, last arriver per panel); the product instance is
// TAIL = 0 and contains none of it (if constexpr).
template <int TAIL> __device__ void mnn_tail_hook(float* smem, const float* img2, int n1, int n2, const u64* partR, size_t ldr, const u64* partC, size_t ldc, u64* pairs);
template <int PRIO, int PIPE, int STG, int DBG = 0, int TAIL = 0>      // DBG (probes only): 1 = no epilogue, 2 = no staging
__global__ __launch_bounds__(512, 2)
void k_mnn_gemm_img(const float* __restrict__ img1, int n1, const float* __restrict__ img2, int n2,
                    u64* __restrict__ partR, size_t ldr, u64* __restrict__ partC, size_t ldc, u64* __restrict__ pairs) {
    __shared__ __attribute__((aligned(1024))) float smem[2 * MNN_PANEL_FLOATS];     // 128 KB: d1 image, d2 image
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, i = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int row_base = blockIdx.y * MNN_PANEL, col_base = blockIdx.x * MNN_PANEL;
    const bool full = (row_base + MNN_PANEL <= n1) && (col_base + MNN_PANEL <= n2);      // block-uniform

    // ---- staging: 16 pieces of 1 KB per quarter and operand; wave w moves pieces 2w, 2w+1 of each
    const float* gA = img1 + (size_t)blockIdx.y * MNN_PANEL_FLOATS + wave * 512 + lane * 4;
    const float* gB = img2 + (size_t)blockIdx.x * MNN_PANEL_FLOATS + wave * 512 + lane * 4;
    float* lA = smem + wave * 512;
    float* lB = smem + MNN_PANEL_FLOATS + wave * 512;
    auto issue_quarter = [&](int kc) {
        if (DBG == 2) return;
        mnn_dma1k(gA + kc * 4096, lA + kc * 4096);
        mnn_dma1k(gA + kc * 4096 + 256, lA + kc * 4096 + 256);
        mnn_dma1k(gB + kc * 4096, lB + kc * 4096);
        mnn_dma1k(gB + kc * 4096 + 256, lB + kc * 4096 + 256);
    };
    if (STG == 0) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) issue_quarter(kc);
    } else { issue_quarter(0); issue_quarter(1); }
    // arm the (column, value) pairs of this d1 panel for k_mnn_post's collectors (mnn_prepost.hip.h)
    if (blockIdx.x == 0 && t < MNN_PANEL && row_base + t < n1) pairs[row_base + t] = 0xFFFFFFFE00000000ull;
    if (PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 2 && wave < 4) __builtin_amdgcn_s_setprio(1);

    // operand addresses (mnn_layout.h): as d1, lane i of tile rt reads the row h'*32 + rt*16 + r of the wave's 64-row strip
    const int rsA = ((i >> 2) & 1) * 32 + (i & 3) + 4 * ((i >> 3) & 3);
    const float* pa[2][2];                                                   // [rt][gg]: + kc*4096
    const float* pb[4][2];                                                   // [ct][gg]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int pos = mnn_pos(wr * 64 + rsA + rt * 16), sw = mnn_swz(pos);
        pa[rt][0] = smem + pos * 16 + (((0 | h) ^ sw) << 2);
        pa[rt][1] = smem + pos * 16 + (((2 | h) ^ sw) << 2);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int pos = wc * 128 + ct * 32 + i, sw = mnn_swz(pos);               // = mnn_pos(wc*128 + i*4 + ct)
        pb[ct][0] = smem + MNN_PANEL_FLOATS + pos * 16 + (((0 | h) ^ sw) << 2);
        pb[ct][1] = smem + MNN_PANEL_FLOATS + pos * 16 + (((2 | h) ^ sw) << 2);
    }
    const f32x16 Z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    f32x16 acc[2][4];
    f32x4 fa[2][2][2], fb[2][2][4];               // [buffer][group of 8 inside the quarter][tile]
    auto wait_quarter = [&](int kc) {
        // STG 0: all 16 DMAs were issued up front; STG 1: quarter kc+1 was issued before this wait (except for the last)
        if (STG == 0) { if (kc == 0) MNN_WAIT_VM(12); else if (kc == 1) MNN_WAIT_VM(8); else if (kc == 2) MNN_WAIT_VM(4); else MNN_WAIT_VM(0); }
        else { if (kc < 3) MNN_WAIT_VM(4); else MNN_WAIT_VM(0); }
        __builtin_amdgcn_s_barrier();                 // quarter kc of both panels is in LDS
    };
    auto load_quarter = [&](int kc, int buf) {        // 12 ds_read_b128: both groups of 8 of the quarter
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) fa[buf][gg][rt] = *(const f32x4*)(pa[rt][gg] + kc * 4096);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) fb[buf][gg][ct] = *(const f32x4*)(pb[ct][gg] + kc * 4096);
        }
    };
    auto mfma_group = [&](int kc, int buf, int gg) {  // 32 MFMAs; the first k step takes the literal zero as C
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][gg][rt][j], fb[buf][gg][ct][j], (kc | gg | j) ? acc[rt][ct] : Z16, 0, 0, 0);
    };
    if (PIPE == 0) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            if (STG == 1 && kc >= 1 && kc < 3) issue_quarter(kc + 1);
            wait_quarter(kc);
            load_quarter(kc, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(kc, 0, 0);
            mfma_group(kc, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // the arrival barrier and the operand reads of quarter kc+1 sit in the middle of the MFMAs of quarter kc
        wait_quarter(0);
        load_quarter(0, 0);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(kc, kc & 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kc < 3) { if (STG == 1 && kc < 2) issue_quarter(kc + 2); wait_quarter(kc + 1); load_quarter(kc + 1, (kc + 1) & 1); }
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(kc, kc & 1, 1);
        }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    XFH_MFMA_SETTLE();                                      // common.h: the epilogue starts with a branch
    if (DBG == 1) {
        float sdbg = 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) sdbg += acc[rt][ct][0] + acc[rt][ct][7] + acc[rt][ct][15];
        if (sdbg == 123.456f) partR[t] = 1ull;
        return;
    }
    // ---- epilogue.  acc[rt][ct][r] = < d1 row R0 + rt*16 + r , d2 row C0 + ct >
    const float NEG = -__builtin_huge_valf();
    const int R0 = row_base + wr * 64 + h * 32;
    const int C0 = col_base + wc * 128 + i * 4;
    if (!full) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool vr = R0 + rt * 16 + r < n1;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    if (!(vr && C0 + ct < n2)) acc[rt][ct][r] = NEG;
            }
    }
    // rows: value maximum over this lane's 4 consecutive d2 rows; slot q = rt*16 + r <-> d1 row R0 + q
    float rv[32];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            rv[rt * 16 + r] = fmaxf(mnn_max3(acc[rt][0][r], acc[rt][1][r], acc[rt][2][r]), acc[rt][3][r]);
    // columns: value maximum over the 16 consecutive d1 rows of each rt, key = (value, d1 row group)
    u64 ck[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        float m[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            m[rt] = acc[rt][ct][0];
#pragma unroll
            for (int r = 1; r < 15; r += 2) m[rt] = mnn_max3(m[rt], acc[rt][ct][r], acc[rt][ct][r + 1]);
            m[rt] = fmaxf(m[rt], acc[rt][ct][15]);
        }
        const bool second = m[1] > m[0];                   // tie -> the lower row group
        const float mm = second ? m[1] : m[0];
        const unsigned gr = (unsigned)(R0 >> 4) + (second ? 1u : 0u);
        const u64 k = (mm > NEG) ? mnn_pack_key(mm, gr) : 0ull;
        ck[ct] = mnn_umax64(k, __shfl_xor(k, 32));
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): this wave's operand reads are done
    __builtin_amdgcn_s_barrier();                          // every wave is done with the images

    // scratch in the dead image region: per wave a 32 x 68 float transposition tile, then the key exchange
    float* T = smem + wave * (32 * 68);                                  // 8 x 8704 B
    u64* sCol = (u64*)(smem + 8 * 32 * 68);                              // [wr 4][256]
    u64* sRow = sCol + 4 * 256;                                          // [wc 2][256]
    if (lane < 32) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) sCol[wr * 256 + wc * 128 + ct * 32 + lane] = ck[ct];
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) T[q * 68 + lane] = rv[q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        // lane (i, h) <-> d1 row R0' = row_base + wr*64 + lane: slot q = i of the 32 lanes of half h
        const float* src = T + i * 68 + h * 32;
        float mg[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 v = *(const f32x4*)(src + g * 4);
            mg[g] = fmaxf(mnn_max3(v.x, v.y, v.z), v.w);
        }
        const float M = fmaxf(mnn_max3(mnn_max3(mg[0], mg[1], mg[2]), mnn_max3(mg[3], mg[4], mg[5]), mg[6]), mg[7]);
        int gi = 7;
#pragma unroll
        for (int g = 6; g >= 0; --g) gi = (mg[g] == M) ? g : gi;
        // source lanes 4*gi .. 4*gi+3 of this wave column <-> d2 rows col_base + wc*128 + 16*gi .. +15; the first of the four that reaches M names the
        // candidate group: source lane s holds the d2 rows col_base + wc*128 + 4s .. 4s+3 (one more LDS read at a computed address, three selects)
        const f32x4 wv = *(const f32x4*)(src + gi * 4);
        const int li = (wv.x == M) ? 0 : ((wv.y == M) ? 1 : ((wv.z == M) ? 2 : 3));
        const unsigned grp = (unsigned)((col_base + wc * 128) >> 2) + (unsigned)(gi * 4 + li);
        sRow[wc * 256 + wr * 64 + lane] = (M > NEG) ? mnn_pack_key(M, grp) : 0ull;
    }
    __syncthreads();
    // one key per row / column of this workgroup's block, written to the plane of this block (coalesced 2 KB each);
    // k_mnn_post takes the maximum over the planes.  (64-bit atomic max straight into one bestR / bestC array was
    // measured at ~15 ns per 1000 atomics: 131 K of them cost the kernel 2 us.)
    if (t < 256) {
        // d2 row col_base + t  <->  position wc*128 + ct*32 + i  with  t = wc*128 + i*4 + ct
        const int p = (t & 128) | ((t & 3) << 5) | ((t & 127) >> 2);
        const u64 kc = mnn_umax64(mnn_umax64(sCol[p], sCol[256 + p]), mnn_umax64(sCol[512 + p], sCol[768 + p]));
        if constexpr (TAIL != 0) __hip_atomic_store(&partC[(size_t)blockIdx.y * ldc + col_base + t], kc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else partC[(size_t)blockIdx.y * ldc + col_base + t] = kc;
    } else {
        const int r = t - 256;
        const u64 kr = mnn_umax64(sRow[r], sRow[256 + r]);
        if constexpr (TAIL != 0) __hip_atomic_store(&partR[(size_t)blockIdx.x * ldr + row_base + r], kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else partR[(size_t)blockIdx.x * ldr + row_base + r] = kr;
    }
    if constexpr (TAIL != 0) mnn_tail_hook<TAIL>(smem, img2, n1, n2, partR, ldr, partC, ldc, pairs);
}
