// mnn_layout.h -- the cosine-similarity GEMM of ORBmatcher::match (reference src/ORBmatcher.cc:358-368)
// with the first level of the row / column arg-max fused in, for gfx950.  This header: the operand layout shared by k_rownorm_img, k_mnn_gemm_img and k_mnn_post.
//
// Operand layout ("panel image", written by k_rownorm_img): the normalised rows of a descriptor set are
// stored in panels of 256 rows, and a panel is stored exactly as the LDS image the MFMA loop reads:
//
//     float index inside a panel = kc * 4096 + pos * 16 + ((c ^ swz(pos)) * 4) + s
//
//   kc  = k >> 4               quarter of the 64-element row (one 64-byte piece per row and quarter)
//   c   = ((k >> 3) & 1) * 2 + (k & 1), s = (k & 7) >> 1
//                              inside a group of 8 the even elements come first, so that lane-half h of an
//                              MFMA reads k = 8g + 2j + h for j = 0..3 with one ds_read_b128
//   pos = (row & 128) | (row & 3) << 5 | (row & 127) >> 2     position of the row inside the panel
//   swz(pos) = ((pos >> 2) ^ (pos >> 5)) & 3                  spreads the 16-lane groups of a ds_read_b128 over all 64
//                              banks for BOTH access patterns below (checked exhaustively: tests/test_abi_and_host.py)
//
// A workgroup copies one d1 panel and one d2 panel (64 KB each) into LDS with 1-KB LDS-DMA instructions
// (global_load_lds_dwordx4: wave-uniform LDS base + lane * 16, which is why the global image IS the LDS
// image), quarter by quarter, and starts the MFMAs of quarter kc as soon as that quarter has landed.
// No registers and no ds_write are spent on staging and nothing is re-staged: after the four arrival
// barriers the K loop runs without synchronisation.
//
// The image is the same whichever side of the match a set is used on (xfh_match_prepare_device builds it once per
// frame); the two sides READ it differently, which is what makes the arg-max epilogue cheap (k_mnn_gemm_img):
//   as d1 (MFMA rows), strip of 64 rows of wave row wr: lane i of tile rt reads row  h'*32 + rt*16 + r  with
//        h' = (i >> 2) & 1, r = (i & 3) + 4 * ((i >> 3) & 3): the C/D layout of v_mfma_f32_32x32x2_f32 gives lane-half h
//        the MFMA rows (r&3) + 8*(r>>2) + 4*h, so a lane's 16 accumulator values of one tile are 16 CONSECUTIVE d1 rows;
//   as d2 (MFMA columns), strip of 128 rows of wave column wc: lane i of tile ct reads row i*4 + ct (position
//        ct*32 + i): lane i of the four column tiles holds 4 CONSECUTIVE d2 rows.
#pragma once
#include "common.h"

#define MNN_PANEL 256                 // rows per panel
#define MNN_PANEL_FLOATS (MNN_PANEL * 64)
#define MNN_RGROUP 16                 // d1 rows per column-candidate group  (bestC key)
#define MNN_CGROUP 4                  // d2 rows per row-candidate group     (bestR key): the 4 consecutive d2 rows one lane of the GEMM holds (round 5; 16 before:
                                      // k_mnn_post fetched 16 candidate rows and 16 x 16 column keys per d1 row -- 6 KB per row, and at 8 pairs it ran at the cache's rate)

__host__ __device__ inline int mnn_pos(int r256) {             // row inside the panel -> position
    return (r256 & 128) | ((r256 & 3) << 5) | ((r256 & 127) >> 2);
}
__host__ __device__ inline int mnn_swz(int pos) { return ((pos >> 2) ^ (pos >> 5)) & 3; }
// float offset of the 16-byte piece (g = k >> 3, half = k & 1) of the row at position pos
__host__ __device__ inline int mnn_piece(int pos, int g, int half) {
    return (g >> 1) * 4096 + pos * 16 + (((((g & 1) << 1) | half) ^ mnn_swz(pos)) << 2);
}

__device__ __forceinline__ u64 mnn_pack_key(float v, unsigned idx) {
    return ((u64)f2ord(v) << 32) | (u64)(0xFFFFFFFFu - idx);
}
__device__ __forceinline__ u64 mnn_umax64(u64 a, u64 b) { return a > b ? a : b; }


// One descriptor row -> its normalised row in the panel image.  Sixteen lanes hold the row (lane `sub` the elements 4*sub .. 4*sub+3;
// a row of zeros for padding).  F::normalize as the oracle states it: fp64 sum of squares, fp32 sqrt / max(., 1e-12) / divide.
// Shared by k_rownorm_img (rows from a caller's descriptor array) and k_desc (the rows it has just produced: the extraction then
// hands the matcher a prepared image and k_rownorm_img disappears from the frame-to-frame match) -- one function, the same bits.
__device__ __forceinline__ void mnn_emit_row(const f32x4 v, int row, int sub, float* __restrict__ img) {
    double ss = (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z + (double)v.w * (double)v.w;
    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4); ss += __shfl_xor(ss, 8);
    const float nrm = fmaxf((float)sqrt(ss), 1e-12f);
    const float a = v.x / nrm, b = v.y / nrm, c = v.z / nrm, e = v.w / nrm;
    // this lane holds elements 4*sub .. 4*sub+3; its pair lane (sub^1) holds the other half of the group of 8.
    // even lane writes the piece of the even elements (e0 e2 e4 e6), odd lane the piece of the odd ones.
    const bool odd = sub & 1;
    const float sx = odd ? a : b, sy = odd ? c : e;            // what the partner needs from me
    const float rx = __shfl_xor(sx, 1), ry = __shfl_xor(sy, 1);
    const f32x4 outv = odd ? f32x4{rx, ry, b, e} : f32x4{a, c, rx, ry};
    const int pos = mnn_pos(row & (MNN_PANEL - 1));
    *(f32x4*)(img + (size_t)(row >> 8) * MNN_PANEL_FLOATS + mnn_piece(pos, sub >> 1, odd ? 1 : 0)) = outv;
}
