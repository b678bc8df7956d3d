// dist_probe.hip -- where k_dist_mfma's time goes (development probe, not part of the library): the product kernel and variants with phases
// switched off, 4096 x 4096 unit descriptors, back to back on a settled clock.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
//   tools/probes/dist_probe.hip -o tools/probes/dist_probe && tools/probes/dist_probe
#include "../../xfeatslam_amd/csrc/dist_mfma.hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

template <int DBG> static void go(dim3 g, hipStream_t s, const float* a, int n1, const float* b, int n2, int32_t* o) {
    hipLaunchKernelGGL((k_dist_mfma<DBG>), g, dim3(512), 0, s, a, n1, b, n2, o, dist_tiles_per_block(n1, n2, 256));
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    std::vector<float> h((size_t)n * 64);
    unsigned x = 12345u;
    for (int r = 0; r < n; ++r) {
        double ss = 0;
        for (int k = 0; k < 64; ++k) { x = x * 1664525u + 1013904223u; const float v = (float)((x >> 8) & 0xffff) / 65536.f - 0.5f; h[(size_t)r * 64 + k] = v; ss += (double)v * v; }
        for (int k = 0; k < 64; ++k) h[(size_t)r * 64 + k] /= (float)sqrt(ss);
    }
    float *a, *b; int32_t* o;
    hipMalloc((void**)&a, h.size() * 4); hipMalloc((void**)&b, h.size() * 4); hipMalloc((void**)&o, (size_t)n * n * 4);
    hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < h.size(); i += 64) std::swap(h[i], h[i + 7]);
    hipMemcpy(b, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nt = dist_tiles_per_block(n, n, 256);
    const dim3 g((n + nt * DTC - 1) / (nt * DTC), (n + DT - 1) / DT);
    const char* names[] = {"product", "no fix-up", "no fix-up, no bulk epilogue", "staging + MFMA without the norm shuffles", "staging only (no MFMA, no epilogue)", "no MFMA (staging + epilogue + fix-up)", "no stores in the bulk pass (timing)", "no stores, no fix-up", "second half of the grid starts 3.4 us late", "odd blockIdx.x starts 3.4 us late"};
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 10; ++v) {
            auto launch = [&]() {
                switch (v) {
                    case 0: go<0>(g, s, a, n, b, n, o); break;
                    case 1: go<1>(g, s, a, n, b, n, o); break;
                    case 2: go<3>(g, s, a, n, b, n, o); break;
                    case 3: go<11>(g, s, a, n, b, n, o); break;
                    case 4: go<7>(g, s, a, n, b, n, o); break;
                    case 5: go<4>(g, s, a, n, b, n, o); break;
                    case 6: go<32>(g, s, a, n, b, n, o); break;
                    case 7: go<33>(g, s, a, n, b, n, o); break;
                    case 8: go<64>(g, s, a, n, b, n, o); break;
                    case 9: go<128>(g, s, a, n, b, n, o); break;
                }
            };
            for (int i = 0; i < 300; ++i) launch();
            hipEventRecord(e0, s);
            for (int i = 0; i < 300; ++i) launch();
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-48s %7.2f us per launch\n", names[v], ms * 1e3 / 300);
        }
    return 0;
}
