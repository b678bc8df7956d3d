// mnn_probe_gemm.hip -- variants of k_mnn_gemm_img for tools/probes/mnn_probe.hip (development probe, not part of
// the library).  Compiled with -fno-honor-nans like the product instance (kernels_mnn_gemm.hip).
#include "../../xfeatslam_amd/csrc/mnn_gemm.hip.h"
#include <hip/hip_ext.h>

template <int PRIO, int PIPE, int STG, int DBG = 0>
static void go(dim3 grid, hipStream_t s, hipEvent_t e0, hipEvent_t e1, const float* i1, int n1, const float* i2, int n2, u64* pR, size_t ldr, u64* pC, size_t ldc, u64* pairs) {
    if (e0) hipExtLaunchKernelGGL((k_mnn_gemm_img<PRIO, PIPE, STG, DBG>), grid, dim3(512), 0, s, e0, e1, 0, i1, n1, i2, n2, pR, ldr, pC, ldc, pairs);
    else hipLaunchKernelGGL((k_mnn_gemm_img<PRIO, PIPE, STG, DBG>), grid, dim3(512), 0, s, i1, n1, i2, n2, pR, ldr, pC, ldc, pairs);
}
const char* probe_gemm_name(int v) {
    static const char* n[] = {"pipe0 stg0", "pipe1 stg0", "pipe0 stg1", "pipe1 stg1", "prio2 pipe1 stg1", "pipe1 stg1 NO EPILOGUE (timing only)", "pipe1 stg1 NO STAGING (timing only)"};
    return v >= 0 && v < 7 ? n[v] : nullptr;
}
void probe_gemm(int v, hipStream_t s, hipEvent_t e0, hipEvent_t e1, const float* i1, int n1, const float* i2, int n2, u64* pR, size_t ldr, u64* pC, size_t ldc, u64* pairs) {
    const dim3 grid((n2 + MNN_PANEL - 1) / MNN_PANEL, (n1 + MNN_PANEL - 1) / MNN_PANEL);
    switch (v) {
        case 0: go<0, 0, 0>(grid, s, e0, e1, i1, n1, i2, n2, pR, ldr, pC, ldc, pairs); break;
        case 1: go<0, 1, 0>(grid, s, e0, e1, i1, n1, i2, n2, pR, ldr, pC, ldc, pairs); break;
        case 2: go<0, 0, 1>(grid, s, e0, e1, i1, n1, i2, n2, pR, ldr, pC, ldc, pairs); break;
        case 3: go<0, 1, 1>(grid, s, e0, e1, i1, n1, i2, n2, pR, ldr, pC, ldc, pairs); break;
        case 4: go<2, 1, 1>(grid, s, e0, e1, i1, n1, i2, n2, pR, ldr, pC, ldc, pairs); break;
        case 5: go<0, 1, 1, 1>(grid, s, e0, e1, i1, n1, i2, n2, pR, ldr, pC, ldc, pairs); break;
        case 6: go<0, 1, 1, 2>(grid, s, e0, e1, i1, n1, i2, n2, pR, ldr, pC, ldc, pairs); break;
    }
}
