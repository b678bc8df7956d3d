// asan_host_test.cpp -- drives the entry points of libxfeat_hip that need no GPU, linked against the sanitizer build
// (make -C xfeatslam_amd/csrc asan): AddressSanitizer / UBSan abort the process on any finding, so exit code 0 = clean.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "xfeat_hip.h"
#include "xfeat_hip_bench.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "asan_host_test: %s failed (line %d)\n", #x, __LINE__); return 1; } } while (0)

int main() {
    xfh_config cfg;
    xfh_config_default(&cfg);
    CHECK(cfg.nfeatures == 4096 && cfg.max_batch == 1);
    for (int s = -2; s < 16; ++s) CHECK(xfh_strerror(s) != nullptr);
    for (int k = -1; k < 16; ++k) CHECK(xfh_kernel_name(k) != nullptr);
    CHECK(xfh_record_desc_offset(4096) % 256 == 0 && xfh_record_bytes(1) >= 16 + 28 + 256);
    CHECK(xfh_match_image_bytes(0) == 0 && xfh_match_image_bytes(257) == 512 * 256);
    CHECK(xfh_compact_bytes_max(8, 2) > 256);
    // no device in this process' world (or a device: then create works and is destroyed again): never a crash
    xfh_ctx* ctx = nullptr;
    const int rc = xfh_create(&cfg, &ctx);
    CHECK(rc == XFH_OK || rc == XFH_ERR_NO_DEVICE);
    if (ctx) xfh_destroy(ctx);
    CHECK(xfh_create(nullptr, &ctx) == XFH_ERR_INVALID_ARG && xfh_destroy(nullptr) == XFH_OK);
    CHECK(xfh_synchronize(nullptr) == XFH_ERR_INVALID_ARG && xfh_extract_batch_wait(nullptr) == XFH_ERR_INVALID_ARG);
    // descriptor distance: unit vectors, zero rows
    std::vector<float> a(64, 0.f), b(64, 0.f);
    a[0] = 1.f; b[1] = 1.f;
    CHECK(xfh_descriptor_distance(a.data(), b.data()) == 1024 && xfh_descriptor_distance(a.data(), a.data()) == 0);
    std::fill(b.begin(), b.end(), 0.f);
    CHECK(xfh_descriptor_distance(a.data(), b.data()) == 512);
    // compact shard reader on well-formed and hostile input
    const int nf = 8, B = 2, total = 5;
    const size_t hdr_b = 256, kps_b = 256;
    std::vector<unsigned char> shard(256 + hdr_b + kps_b + (size_t)total * 256, 0);
    int* h0 = (int*)shard.data(); h0[0] = B; h0[1] = nf; h0[2] = total;
    int* f0 = (int*)(shard.data() + 256); f0[0] = 5; f0[1] = 3; f0[4] = 0; f0[5] = 0;
    std::vector<xfh_keypoint> kp(nf); std::vector<float> desc((size_t)nf * 64);
    int nv = -1, mono = -1;
    CHECK(xfh_unpack_compact(shard.data(), shard.size(), 0, nf, kp.data(), desc.data(), &nv, &mono) == XFH_OK && nv == 5 && mono == 3);
    CHECK(xfh_unpack_compact(shard.data(), shard.size(), 1, nf, kp.data(), desc.data(), &nv, &mono) == XFH_OK && nv == 0);
    CHECK(xfh_unpack_compact(shard.data(), 300, 0, nf, kp.data(), desc.data(), nullptr, nullptr) == XFH_ERR_INVALID_ARG);     // truncated
    CHECK(xfh_unpack_compact(shard.data(), shard.size(), 2, nf, kp.data(), desc.data(), nullptr, nullptr) == XFH_ERR_INVALID_ARG);   // frame out of range
    f0[0] = 1 << 30; f0[1] = -7;                                      // absurd counts in a frame header: clamped, never read out of bounds
    (void)xfh_unpack_compact(shard.data(), shard.size(), 0, nf, kp.data(), desc.data(), &nv, &mono);
    h0[2] = 1 << 28;                                                  // absurd total
    CHECK(xfh_unpack_compact(shard.data(), shard.size(), 0, nf, kp.data(), desc.data(), &nv, &mono) == XFH_ERR_INVALID_ARG);
    h0[0] = 1 << 30;                                                  // absurd frame count
    (void)xfh_unpack_compact(shard.data(), shard.size(), 1 << 29, nf, kp.data(), desc.data(), &nv, &mono);
    printf("asan_host_test ok\n");
    return 0;
}
