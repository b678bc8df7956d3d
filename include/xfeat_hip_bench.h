/*
 * xfeat_hip_bench.h -- measurement and debugging entry points of libxfeat_hip.so.  NOT part of the drop-in boundary
 * (include/xfeat_hip.h): nothing a SLAM consumer calls lives here.  bench.py, tools/ and tests/ use these through ctypes.
 */
#ifndef XFEAT_HIP_BENCH_H
#define XFEAT_HIP_BENCH_H

#include "xfeat_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- kernel timers --------------------------------------------------------------------
 * Kernel timing: while enabled, every launch of the kernel family `kernel_id` on the ctx
 * stream is bracketed by hipEvents; xfh_timing_read returns launches and total ms since
 * the last reset.  bench.py uses it for the roofline line. */
enum {
    XFH_K_NONE = 0, XFH_K_MNN_GEMM = 1, XFH_K_CONV_MFMA = 2, XFH_K_CONV_DIRECT = 3,
    XFH_K_NMS = 4, XFH_K_SELECT = 5, XFH_K_DESC = 6, XFH_K_HEADS = 7, XFH_K_DIST_I32 = 8,
    XFH_K_PREPROC = 9, XFH_K_BEST2 = 10, XFH_K_DISTINCTIVE = 11, XFH_K_MNN_GEMM_SEG = 12, XFH_K_COUNT = 13
};
/* layer_mask selects conv layers for XFH_K_CONV_*: 0 = every layer, else bit i = BasicLayer i
 * (0..22 in XFeatModel order) and bit 23 = block_fusion.2 */
int xfh_timing_enable(xfh_ctx* ctx, int kernel_id, unsigned layer_mask);
int xfh_timing_read(xfh_ctx* ctx, int* launches, double* total_ms);   /* XFH_ERR_BATCH_TOO_LARGE: more than 4096 launches matched since
                                                                          * xfh_timing_enable; *launches / *total_ms cover the first 4096 */
/* `iters` launches of the match GEMM alone, back to back, on two prepared images: wall time per launch between two stream
 * events.  (Dispatch-attached timestamps of consecutive kernels in a busy stream overlap; this is the steady-state cost.) */
int xfh_bench_mnn_gemm(xfh_ctx* ctx, const void* d_image1, int n1, const void* d_image2, int n2, int iters, double* us_per_launch);
/* `iters` whole xfh_match_mnn_prepared_device calls (both launches; _raw: xfh_match_mnn_device, three) back to back from C: wall time per call between two stream
 * events -- what a C++ caller's loop sees, without the per-call cost of a foreign-function binding. */
int xfh_bench_match_prepared(xfh_ctx* ctx, const void* d_image1, int n1, const void* d_image2, int n2, float min_cossim,
                             int* d_idx1, int* d_idx2, float* d_dist, int* d_n_matches, int iters, double* us_per_call);
int xfh_bench_match_raw(xfh_ctx* ctx, const float* d_d1, int n1, const float* d_d2, int n2, float min_cossim,
                        int* d_idx1, int* d_idx2, float* d_dist, int* d_n_matches, int iters, double* us_per_call);   /* the same for xfh_match_mnn_device */
/* the many-pairs call (xfh_match_mnn_prepared_batch_device): `iters` launches of its GEMM alone (k_mnn_gemm_seg) back to back -> wall time per
 * launch, and (sclk_mhz, may be NULL) the shader clock workgroup 0 saw INSIDE the last launch (shader clocks per 100 MHz reference tick: the clock
 * the GPU holds on these operands, which prices the f32 MFMA peak this kernel can reach); and `iters` whole calls (GEMM + k_mnn_post_batch) back
 * to back from C -> wall time per call.  Arguments as the call itself. */
int xfh_bench_mnn_gemm_batch(xfh_ctx* ctx, int n_pairs, const void* const* d_image1, const int* n1, const void* const* d_image2, const int* n2,
                             int iters, double* us_per_launch, double* sclk_mhz);
int xfh_bench_match_batch(xfh_ctx* ctx, int n_pairs, const void* const* d_image1, const int* n1, const void* const* d_image2, const int* n2, float min_cossim,
                          int* const* d_idx1, int* const* d_idx2, float* const* d_dist, int* d_n_matches, int iters, double* us_per_call);
/* The work plan of the many-pairs GEMM for a list of pair shapes on `num_cu` workgroups (host arithmetic only, no GPU, no ctx: xfeatslam_amd/csrc/mnn_seg_plan.h).
 * Outputs: *tiles, *workgroups; per pair tile0[p] (first tile), planes_max[p] (row-key planes reserved per d1 row); wg_lo[w] for w = 0 .. *workgroups
 * (workgroup w owns the tiles [wg_lo[w], wg_lo[w + 1])); *keys = u64 entries of the key buffer.  n_pairs <= 16; wg_lo needs num_cu + 1 slots. */
int xfh_debug_match_plan(int n_pairs, const int* n1, const int* n2, int num_cu, int* tiles, int* workgroups, int* tile0, int* planes_max, int* wg_lo,
                         unsigned long long* keys);
const char* xfh_kernel_name(int kernel_id);

/* intermediate tensors of frame `frame` of the last extract call, copied to host as float
 * (ids match oracle/xfeat_oracle.h; image-like tensors are NHWC).  count_out = floats.
 * Statistics (XSTAT, STAT0 + layer) are stored as the fma operands they are applied with: [C] beta = -(mean * rstd), then [C] alpha = rstd. */
enum {
    XFH_T_X = 0, XFH_T_XSTAT = 1, XFH_T_SKIP_POOL = 2,                               /* 3, 4, 5, 7 (unfold2d(x), x1 + skip, fusion input, normalised features) */
    XFH_T_FEATS = 6, XFH_T_H1 = 8, XFH_T_K1H = 9,                                   /* are never materialised on the GPU: fused into consumers */
    XFH_T_RAW0 = 16, XFH_T_STAT0 = 48, XFH_T_SEL = 80                               /* RAW0 + 0 (block1.0) likewise: recomputed inside block1.1 */
};
int xfh_debug_tensor(xfh_ctx* ctx, int id, int frame, float* out, size_t capacity, size_t* count_out);

/* The top-k stage alone on a caller-made candidate set (tests): `keys` = n unique 64-bit keys as the NMS stage writes them
 * ((~ordered(score)) << 32 | y * width + x: ascending key = descending score, then ascending pixel index; n <= the ctx's candidate
 * capacity, nfeatures <= 4096).  sel_out (capacity nfeatures) receives the selected keys in rank order, *n_out their number,
 * hdr_out[4] the record header the stage writes (n_valid, mono_index, n_candidates, 0).  form: 0 = as the ctx runs it,
 * 1 = bucket ranking (falls back by itself when a bucket is too large), 2 = radix select + bitonic sort. */
int xfh_debug_select(xfh_ctx* ctx, const unsigned long long* keys, int n, int width, int lap0, int lap1, int form,
                     unsigned long long* sel_out, int* n_out, int* hdr_out);


/* Counter calibration (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern before
 * trusting an absolute"): `iters` launches of a kernel that moves exactly `nbytes` per launch (a buffer far larger than the
 * 256 MB Infinity Cache: use >= 1 GiB), named k_calib<mode> in a kernel trace.  mode 0 / 1 / 4 = read with 4 / 16 / 32 bytes per
 * lane, 2 / 3 / 5 = write with 4 / 16 / 32 bytes per lane -- the access widths of the extraction kernels.  tools/pmc_calib.sh runs
 * them under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE; tools/summarize_profiles.py turns counter / nbytes into the
 * correction factors of profiles/pmc_traffic.json. */
int xfh_bench_calib(xfh_ctx* ctx, int mode, size_t nbytes, int iters);
/* The shader clock under f32 MFMA load: every SIMD of the device issues `mfmas` v_mfma_f32_32x32x2_f32 per wave back to back (two waves per SIMD);
 * *sclk_mhz = shader clocks per 100 MHz reference tick, *cycles_per_mfma = the issue period seen by one wave (128 = the matrix pipe never idles).
 * The f32 MFMA peak at THIS clock is 256 CUs x 256 flop x sclk; the nominal 157.3 TFLOP/s assumes 2.4 GHz. */
int xfh_bench_sclk(xfh_ctx* ctx, int mfmas, double* sclk_mhz, double* cycles_per_mfma);

#ifdef __cplusplus
}
#endif
#endif /* XFEAT_HIP_BENCH_H */
