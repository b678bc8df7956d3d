// kernels_mnn_gemm.hip -- the product instance of k_mnn_gemm_img (mnn_gemm.hip.h): the descriptor-distance GEMM of
// ORBmatcher::match (reference src/ORBmatcher.cc:358-368).  Its own translation unit because it is compiled with
// -fno-honor-nans (see mnn_gemm.hip.h); everything else in the library keeps IEEE NaN semantics.
#include "ctx.h"
#include "mnn_gemm.hip.h"
#include "mnn_gemm_seg.hip.h"

// measured on MI355X (tools/probes/mnn_probe, profiles/r02_mnn_probe.log): no wave priorities, arrival barrier + operand
// reads of quarter kc+1 in the middle of the MFMAs of quarter kc, LDS-DMAs of quarter kc+2 issued before the wait for quarter kc+1
hipError_t launch_mnn_gemm(xfh_ctx* c, const float* img1, int n1, const float* img2, int n2, u64* partR, size_t ldr, u64* partC, size_t ldc, u64* pairs) {
    const dim3 grid((n2 + MNN_PANEL - 1) / MNN_PANEL, (n1 + MNN_PANEL - 1) / MNN_PANEL);
    launch_k(c, XFH_K_MNN_GEMM, -1, k_mnn_gemm_img<0, 1, 1>, grid, dim3(512), 0, img1, n1, img2, n2, partR, ldr, partC, ldc, pairs);
    return hipGetLastError();
}

// the persistent many-pairs form (mnn_gemm_seg.hip.h): one workgroup per CU walks its share of the tiles of all pairs.  SKEW = 1: the two
// wave groups half a tile apart (tools/probes/mnn_seg_probe, profiles/r04_mnn_probe.log: 8 pairs of 4096 x 4096 133 us against 146 in lockstep)
hipError_t launch_mnn_gemm_seg(xfh_ctx* c, const MnnBatch& jb) {
    launch_k(c, XFH_K_MNN_GEMM_SEG, -1, k_mnn_gemm_seg<1, 0>, dim3(jb.G), dim3(512), 0, jb);
    return hipGetLastError();
}
