// mnn_seg_probe_gemm.hip -- the kernel instances tools/probes/mnn_seg_probe.hip measures (development probe, not part of the library).
// Compiled with -fno-honor-nans like the product instances (kernels_mnn_gemm.hip).
#include "../../xfeatslam_amd/csrc/mnn_gemm_seg.hip.h"
#include <hip/hip_ext.h>

template <int SKEW, int DBG>
static void go(int G, hipStream_t s, hipEvent_t e0, hipEvent_t e1, const MnnBatch& jb) {
    if (e0) hipExtLaunchKernelGGL((k_mnn_gemm_seg<SKEW, DBG>), dim3(G), dim3(512), 0, s, e0, e1, 0, jb);
    else hipLaunchKernelGGL((k_mnn_gemm_seg<SKEW, DBG>), dim3(G), dim3(512), 0, s, jb);
}
void probe_seg(int skew, int dbg, int G, hipStream_t s, hipEvent_t e0, hipEvent_t e1, const MnnBatch& jb) {
    if (skew == 0 && dbg == 0) go<0, 0>(G, s, e0, e1, jb);
    else if (skew == 1 && dbg == 0) go<1, 0>(G, s, e0, e1, jb);
    else if (skew == 0 && dbg == 1) go<0, 1>(G, s, e0, e1, jb);
    else if (skew == 1 && dbg == 1) go<1, 1>(G, s, e0, e1, jb);
    else if (skew == 0 && dbg == 2) go<0, 2>(G, s, e0, e1, jb);
    else if (skew == 1 && dbg == 2) go<1, 2>(G, s, e0, e1, jb);
    else go<1, 3>(G, s, e0, e1, jb);
}
void probe_img(hipStream_t s, hipEvent_t e0, hipEvent_t e1, const float* i1, int n1, const float* i2, int n2, u64* pR, size_t ldr, u64* pC, size_t ldc, u64* pairs) {
    const dim3 grid((n2 + MNN_PANEL - 1) / MNN_PANEL, (n1 + MNN_PANEL - 1) / MNN_PANEL);
    if (e0) hipExtLaunchKernelGGL((k_mnn_gemm_img<0, 1, 1, 0>), grid, dim3(512), 0, s, e0, e1, 0, i1, n1, i2, n2, pR, ldr, pC, ldc, pairs);
    else hipLaunchKernelGGL((k_mnn_gemm_img<0, 1, 1, 0>), grid, dim3(512), 0, s, i1, n1, i2, n2, pR, ldr, pC, ldc, pairs);
}
