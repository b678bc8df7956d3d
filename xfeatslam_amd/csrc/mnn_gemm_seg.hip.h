// mnn_gemm_seg.hip.h -- k_mnn_gemm_seg: the cosine-similarity GEMM of ORBmatcher::match (reference src/ORBmatcher.cc:358-368)
// for ONE OR MANY descriptor-set pairs in one launch, as a persistent kernel (layout: mnn_layout.h; second arg-max level, mutual
// check and output: k_mnn_post, mnn_prepost.hip.h).  The reference's consumers meet one frame with several partners -- the
// previous frame, key frames, loop candidates (ORBmatcher.cc:408-610, one match() per frame pair) -- and a 4096 x 4096 pair
// is exactly one 256 x 256 tile per CU: a launch per pair pays the launch ramp, the staging wait and the epilogue's latency
// with nothing to overlap them (k_mnn_gemm_img, mnn_gemm.hip.h).  Here:
//
//   * the 256 x 256 tiles of all pairs form one row-major sequence (pair, d1 panel, d2 panel); workgroup w of G walks the
//     contiguous range { tile : tile * G / T == w } (T tiles in total, G = min(T, CUs)): equal work whatever the shapes are;
//   * the d1 strip of a wave (64 rows x 64 k = 16 KB) lives in REGISTERS for as long as the workgroup stays in one d1 panel
//     (K = 64 makes that 64 VGPRs), so LDS holds d2 panels only: two 64-KB buffers, the next tile's panel arrives by LDS-DMA
//     (global_load_lds_dwordx4: the global image IS the LDS image) while the current tile's MFMAs run;
//   * the 8 waves are two GROUPS of four, one wave of each per SIMD: group X owns the d2 rows 0..127 of every panel, group Y
//     the rows 128..255 (own half of each buffer, own DMAs, own epilogue scratch).  With SKEW the groups run half a tile apart:
//     while X's waves issue the MFMAs of tile s, Y's waves are in the epilogue of tile s-1 -- LDS transposition, barrier waits,
//     key stores -- and vice versa.  On gfx950 the f32 MFMA and the VALU are ONE pipe (tools/probes/pipe_probe.hip,
//     profiles/r04_pipe_probe.log: a plain VALU instruction costs its full time in the same wave, and the other wave's VALU
//     instructions starve while MFMAs are ready), so this hides the epilogue's LATENCY, not its ~300 VALU instructions; it is
//     worth 133 us against 137-147 in lockstep at 8 pairs.  Both groups execute the same number of s_barriers (a barrier closes
//     every half-tile phase), so one WG-wide barrier serves both;
//   * arg-max level 1 as in k_mnn_gemm_img (value maxima only, v_max3), but the d1-row result stays in registers across the
//     tiles of a d1 panel (a running (value, d2 row group) per lane: lane <-> d1 row) and goes to the row's key plane once, and
//     the d2-row maxima are merged across the group's waves from VALUES in LDS (no key packing / 64-bit shuffles per lane and tile).
//
//   planes: partC[d1 panel][d2 row]  (as before), partR[2 * (w - w_first(row panel)) + group][d1 row]; k_mnn_post derives the
//   number of planes of a d1 panel from (T, G, tile0, P2) with the same integer arithmetic (mnn_seg_plan.h).
// Compiled with -fno-honor-nans like k_mnn_gemm_img (kernels_mnn_gemm.hip).
#pragma once
#include "mnn_seg_plan.h"
#include "mnn_gemm.hip.h"

#define MNN_SEG_CV_PITCH 272                      // floats per row of the column-value scratch (h = 1 lanes land 32 banks away)
#define MNN_SEG_LDS_FLOATS (2 * MNN_PANEL_FLOATS + 16 * MNN_SEG_CV_PITCH)

template <int SKEW, int DBG = 0>                  // DBG (probes only): 1 = no epilogue (timing), 2 = phase stamps of workgroup 0, 3 = HW_ID per wave (into job 0's partR)
__global__ __launch_bounds__(512, 2)
void k_mnn_gemm_seg(const MnnBatch jb) {
    __shared__ __attribute__((aligned(1024))) float smem[MNN_SEG_LDS_FLOATS];   // two d2 panel buffers, column-value scratch
    float* const sColV = smem + 2 * MNN_PANEL_FLOATS;
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int wr = wave & 3, wc = wave >> 2;          // waves w and w + 4 share a SIMD: one wave of each group per SIMD
    const int w = blockIdx.x;
    const int tau_lo = mnn_seg_lo(w, jb.T, jb.G), tau_hi = mnn_seg_lo(w + 1, jb.T, jb.G);
    const int NS = tau_hi - tau_lo;
    if (NS <= 0) return;
    const float NEG = -__builtin_huge_valf();
    // measurement (xfh_bench_mnn_gemm_batch): workgroup 0 reports its span in shader clocks and in 100 MHz reference ticks -> the clock the
    // GPU holds INSIDE this kernel, on these operands (an MFMA loop on constant operands clocks higher)
    long long sclk0 = 0, wall0 = 0;
    if (DBG == 0 && jb.dbg && w == 0) { sclk0 = clock64(); wall0 = wall_clock64(); }
    // The per-lane addresses of the epilogue are recomputed from the lane id inside the phase that uses them (LAUNDER keeps hipcc from
    // hoisting them out of the tile loop): hoisted, they sit in registers across the K loop -- accumulators 128 + strip 64 + operands 32
    // are live there -- and get spilled; a reload at the start of a K phase waits vmcnt(0), i.e. for the key stores before it.
#define MNN_LAUNDER(x) asm volatile("" : "+v"(x))
    int nstamp = 0;
#define MNN_SEG_STAMP() do { if (DBG == 2 && w == 0 && lane == 0 && wr == 0 && nstamp < 60) jb.dbg[wc * 64 + nstamp++] = (u64)wall_clock64(); } while (0)

    // ---- tiles (wave-uniform; the job table sits in the kernel arguments).  Everything a phase needs from the table is fetched one
    // phase ahead (scalar loads, free beside the other work) so that no K phase starts with a chain of scalar-memory round trips.
    struct Tile { int p, by, bx; };
    auto first_tile = [&](int tau) {
        Tile r; r.p = 0;
        for (int q = 1; q < jb.njobs; ++q) if (tau >= jb.job[q].tile0) r.p = q;
        const int loc = tau - jb.job[r.p].tile0, P2 = jb.job[r.p].P2;
        r.by = loc / P2; r.bx = loc - r.by * P2;
        return r;
    };
    auto next_tile = [&](const Tile& c) {
        Tile r = c;
        if (++r.bx == jb.job[c.p].P2) { r.bx = 0; if (++r.by == jb.job[c.p].P1) { r.by = 0; ++r.p; } }
        return r;
    };
    auto panel_src = [&](const Tile& tl) { return jb.job[tl.p].img2 + (size_t)tl.bx * MNN_PANEL_FLOATS; };

    // ---- d2 panel staging: wave (wr, wc) moves, per quarter, the two 1-KB pieces wr*2, wr*2+1 of its group's half
    const int stage_off = wc * 2048 + wr * 512;
    auto issue_panel = [&](const float* src, int buf) {
        const float* g = src + stage_off + lane * 4;
        float* l = smem + buf * MNN_PANEL_FLOATS + stage_off;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            mnn_dma1k(g + kc * 4096, l + kc * 4096);
            mnn_dma1k(g + kc * 4096 + 256, l + kc * 4096 + 256);
        }
    };
    const f32x16 Z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    f32x4 fa[4][2][2];                                // the wave's d1 strip: [quarter][group of 8][rt]
    f32x4 fb[2][4];                                   // d2 operands of the current / next group of 8 k (loop carried: a tile's first group is read one phase early)
    f32x16 acc[2][4];
    float rkv = -__builtin_huge_valf(); unsigned rkg = 0u;   // running (value, d2 row group) of d1 row row_base + wr*64 + lane over this group's d2 rows
    int row_wfirst = 0;                               // first workgroup of the current d1 panel

    auto flush_rows = [&](const Tile& tl) {           // the row keys of tile tl's d1 panel go to this workgroup's plane
        if (DBG == 1) return;
        const MnnJob& J = jb.job[tl.p];
        J.partR[(size_t)(2 * (w - row_wfirst) + wc) * J.ldr + (size_t)tl.by * MNN_PANEL + wr * 64 + lane] = (rkv > NEG) ? mnn_pack_key(rkv, rkg) : 0ull;
    };
    // new d1 panel: strip into registers (mnn_layout.h: lane i of tile rt reads row h'*32 + rt*16 + r of the wave's 64 rows), keys
    // reset, pairs armed by the panel's first tile
    auto enter_row = [&](const Tile& tl) {
        const MnnJob& J = jb.job[tl.p];
        const float* a = J.img1 + (size_t)tl.by * MNN_PANEL_FLOATS;
        int ln = lane; MNN_LAUNDER(ln);
        const int i = ln & 31, h = ln >> 5;
        const int rsA = ((i >> 2) & 1) * 32 + (i & 3) + 4 * ((i >> 3) & 3);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int pos = mnn_pos(wr * 64 + rsA + rt * 16), sw = mnn_swz(pos);
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int off = pos * 16 + ((((gg << 1) | h) ^ sw) << 2);
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) fa[kc][gg][rt] = *(const f32x4*)(a + kc * 4096 + off);
            }
        }
        rkv = NEG; rkg = 0u;
        row_wfirst = mnn_seg_wg(J.tile0 + tl.by * J.P2, jb.T, jb.G);
        // arm the (column, value) pairs of this d1 panel for k_mnn_post's collectors (mnn_prepost.hip.h)
        if (tl.bx == 0 && wc == 0) {
            const int r = tl.by * MNN_PANEL + wr * 64 + lane;
            if (r < J.n1) J.pairs[r] = 0xFFFFFFFE00000000ull;
        }
    };

    // ---- M: the d2-row keys of a finished tile, merged over the 16 d1 row groups from the values in LDS.  All four waves of the group:
    // two lanes per d2 row, eight row groups each, combined by one quad exchange.  The LDS reads go out together with the K phase's first
    // operand reads.
    float mv[8];
    auto merge_load = [&]() {
        int ln = lane; MNN_LAUNDER(ln);
        const int tq = wr * 64 + ln, c = tq >> 1, u = tq & 1;            // d2 row c of the group's 128, half u of the row groups
        const int p = ((c & 3) << 5) | (c >> 2);                         // its position (ct*32 + i)
        const float* src = sColV + wc * 128 + p + u * 8 * MNN_SEG_CV_PITCH;
#pragma unroll
        for (int g = 0; g < 8; ++g) mv[g] = src[g * MNN_SEG_CV_PITCH];
    };
    auto merge_store = [&](u64* dst, int keybase) {                    // dst: partC[d1 panel][this tile's d2 rows of the group], keybase: d1 panel * 16
        int ln = lane; MNN_LAUNDER(ln);
        const int tq = wr * 64 + ln, c = tq >> 1, u = tq & 1;
        const float m = fmaxf(mnn_max3(mnn_max3(mv[0], mv[1], mv[2]), mnn_max3(mv[3], mv[4], mv[5]), mv[6]), mv[7]);
        int gi = 7;
#pragma unroll
        for (int g = 6; g >= 0; --g) gi = (mv[g] == m) ? g : gi;        // first group that reaches the maximum
        const float mo = __shfl_xor(m, 1);
        const int gio = __shfl_xor(gi, 1);
        if (u == 0) {
            const bool second = mo > m;                                  // tie -> the lower row groups
            const float M = second ? mo : m;
            const int G = second ? gio + 8 : gi;
            dst[c] = (M > NEG) ? mnn_pack_key(M, (unsigned)(keybase + G)) : 0ull;
        }
    };

    // ---- operand positions of the d2 panel (mnn_layout.h): lane i of column tile ct reads position wc*128 + ct*32 + i
    int offB[4][2];                                   // [ct][gg]: float offset inside a buffer, + quarter * 4096
    {
        const int i = lane & 31, h = lane >> 5;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int pos = wc * 128 + ct * 32 + i, sw = mnn_swz(pos);   // = mnn_pos(wc*128 + i*4 + ct)
            offB[ct][0] = pos * 16 + (((0 | h) ^ sw) << 2);
            offB[ct][1] = pos * 16 + (((2 | h) ^ sw) << 2);
        }
    }

    // ---- prologue: first panel, first strip
    Tile cur = first_tile(tau_lo), prev = cur, nxt = cur;
    const float* nxt_src = nullptr;
    if (NS > 1) { nxt = next_tile(cur); nxt_src = panel_src(nxt); }
    u64* prev_dst = nullptr; int prev_kb = 0;
    issue_panel(panel_src(cur), 0);
    enter_row(cur);
    __builtin_amdgcn_s_waitcnt(0x0f70);                                      // vmcnt(0) as a builtin: the compiler's scoreboard sees it (an asm wait would
    __builtin_amdgcn_s_barrier();                                            // leave it waiting again, for everything, at the first use of the strip)
    if (SKEW && wc == 1) __builtin_amdgcn_s_barrier();                       // Y idles through X's first K phase

    for (int s = 0; s < NS; ++s) {
        const int buf = s & 1;
        MNN_SEG_STAMP();
        // ================= K phase of tile s
        const bool merge = DBG != 1 && s > 0;
        if (s > 0 && (cur.p != prev.p || cur.by != prev.by)) { flush_rows(prev); enter_row(cur); __builtin_amdgcn_s_waitcnt(0x0f70); }
        {
            const float* bq = smem + buf * MNN_PANEL_FLOATS;
            auto load_group = [&](int g, int b) {                            // 4 ds_read_b128: group of 8 k (quarter g>>1, half g&1)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) fb[b][ct] = *(const f32x4*)(bq + offB[ct][g & 1] + (g >> 1) * 4096);
            };
            auto mfma_half = [&](int g, int b, int j0) {                     // 16 MFMAs; the first k step takes the literal zero as C
#pragma unroll
                for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g >> 1][g & 1][rt][j], fb[b][ct][j], (g | j) ? acc[rt][ct] : Z16, 0, 0, 0);
            };
            // the operand reads of group g+1 sit in the MIDDLE of the MFMAs of group g: hipcc waits lgkmcnt(0), never a counted wait,
            // before the first MFMA that needs an operand, so reads issued right in front of that wait are exposed in full (and with
            // SKEW one wave per SIMD is alone in its K phase), while reads sunk behind the group's MFMAs are exposed as well
            if (s == 0) load_group(0, 0);                                    // (later tiles: read at the end of the previous E phase, in front of the barrier)
            if (merge) { merge_load(); merge_store(prev_dst, prev_kb); }    // the accumulators are dead here: registers to spare
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                mfma_half(g, g & 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (g < 7) load_group(g + 1, (g + 1) & 1);
                if (g == 0 && nxt_src) issue_panel(nxt_src, buf ^ 1);       // lands during this K phase and the E phase
                __builtin_amdgcn_sched_barrier(0);
                mfma_half(g, g & 1, 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        XFH_MFMA_SETTLE();                                                   // common.h: the epilogue branches
        __builtin_amdgcn_s_waitcnt(0x0070);                                  // lgkmcnt(0): this wave's operand reads are done; vmcnt(0): its pieces of the next panel
        MNN_SEG_STAMP();                                                     // (issued a whole K phase ago) have landed, so that behind this barrier the next panel is
        __builtin_amdgcn_s_barrier();                                        // readable by every wave.  ---- end of the K phase: buffer `buf` is dead
        MNN_SEG_STAMP();
        // ================= E phase of tile s.  acc[rt][ct][r] = < d1 row R0 + rt*16 + r , d2 row C0 + ct >
        // scalar work for the phases ahead: where this tile's d2-row keys go (M, next K phase), the tile after next and its panel
        {
            const MnnJob& J = jb.job[cur.p];
            prev_dst = J.partC + (size_t)cur.by * J.ldc + (size_t)cur.bx * MNN_PANEL + wc * 128;
            prev_kb = cur.by * (MNN_PANEL / MNN_RGROUP);
        }
        Tile nn = nxt; const float* nn_src = nullptr;
        if (s + 2 < NS) { nn = next_tile(nxt); nn_src = panel_src(nn); }
        if (DBG == 1) {
            float sdbg = 0.f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) sdbg += acc[rt][ct][0] + acc[rt][ct][7] + acc[rt][ct][15];
            if (sdbg == 123.456f) jb.job[0].partR[t] = 1ull;
        } else {
            int ln = lane; MNN_LAUNDER(ln);
            const int i = ln & 31, h = ln >> 5;
            const MnnJob& J = jb.job[cur.p];
            const int row_base = cur.by * MNN_PANEL, col_base = cur.bx * MNN_PANEL;
            const int n1 = J.n1, n2 = J.n2;
            const bool full = (row_base + MNN_PANEL <= n1) && (col_base + wc * 128 + 128 <= n2);      // wave-uniform
            if (!full) {
                const int R0 = row_base + wr * 64 + h * 32;
                const int C0 = col_base + wc * 128 + i * 4;
                bool vc[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) vc[ct] = C0 + ct < n2;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool vr = R0 + rt * 16 + r < n1;
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) acc[rt][ct][r] = (vr && vc[ct]) ? acc[rt][ct][r] : NEG;
                    }
            }
            // d2 rows: value maximum over the 16 consecutive d1 rows of each rt -> sColV[d1 row group][position]
            {
                float* dst = sColV + (wr * 4 + h * 2) * MNN_SEG_CV_PITCH + wc * 128 + i;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        float m = acc[rt][ct][0];
#pragma unroll
                        for (int r = 1; r < 15; r += 2) m = mnn_max3(m, acc[rt][ct][r], acc[rt][ct][r + 1]);
                        dst[rt * MNN_SEG_CV_PITCH + ct * 32] = fmaxf(m, acc[rt][ct][15]);
                    }
            }
            // d1 rows: value maximum over this lane's 4 consecutive d2 rows; slot q = rt*16 + r <-> d1 row R0 + q; transposed through
            // the wave's 8-KB piece of the dead buffer (T[q][lane ^ 4*(q & 15)]: linear writes, conflict-free ds_read_b128)
            // (the piece is 8-KB aligned, so the swizzle can be applied to the whole index: one v_xor per term, the slot offset is an immediate)
            const int tbase = buf * MNN_PANEL_FLOATS + wr * 4096 + wc * 2048;
            const int tl4 = (tbase + ln) * 4;                                // byte offset of T[0][lane]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                char* const tx = (char*)smem + (tl4 ^ (16 * r));
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    *(float*)(tx + (rt * 16 + r) * 256) = fmaxf(mnn_max3(acc[rt][0][r], acc[rt][1][r], acc[rt][2][r]), acc[rt][3][r]);
            }
            const float* T = smem + tbase;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                // lane (i, h) <-> d1 row row_base + wr*64 + lane: slot q = i, source lanes h*32 .. h*32+31
                const float* src = T + i * 64;
                const int x = 4 * (i & 15);
                float mg[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 v = *(const f32x4*)(src + ((h * 32 + g * 4) ^ x));
                    mg[g] = fmaxf(mnn_max3(v.x, v.y, v.z), v.w);
                }
                const float M = fmaxf(mnn_max3(mnn_max3(mg[0], mg[1], mg[2]), mnn_max3(mg[3], mg[4], mg[5]), mg[6]), mg[7]);
                int gi = 7;
#pragma unroll
                for (int g = 6; g >= 0; --g) gi = (mg[g] == M) ? g : gi;
                // source lanes 4*gi .. 4*gi+3 <-> d2 rows col_base + wc*128 + 16*gi .. +15; the first of the four that reaches M names the candidate
                // group of 4 d2 rows (source lane s holds col_base + wc*128 + 4s .. 4s+3): one more LDS read at a computed address, three selects
                const f32x4 wv = *(const f32x4*)(src + ((h * 32 + gi * 4) ^ x));
                const int li = (wv.x == M) ? 0 : ((wv.y == M) ? 1 : ((wv.z == M) ? 2 : 3));
                // a later tile only wins with a larger value (the tiles of a d1 panel come in ascending d2 order): ties keep the lower d2 group
                const bool better = M > rkv;
                rkg = better ? (unsigned)((col_base + wc * 128) >> 2) + (unsigned)(gi * 4 + li) : rkg;
                rkv = better ? M : rkv;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0070);                                  // this wave's LDS writes (column values) are done
        if (s + 1 < NS) {                                                    // the next K phase's first operands, read in front of the barrier this phase ends with (the wait
            const float* bq = smem + (buf ^ 1) * MNN_PANEL_FLOATS;           // for them then overlaps the wait for the other waves): the next panel is complete since the K-end barrier
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) fb[0][ct] = *(const f32x4*)(bq + offB[ct][0]);
        }
        MNN_SEG_STAMP();
        __builtin_amdgcn_s_barrier();                                        // ---- end of the E phase
        prev = cur; cur = nxt; nxt = nn; nxt_src = nn_src;
    }
    // ---- the last tile's d2-row keys, this d1 panel's row keys
    if (DBG != 1) { merge_load(); merge_store(prev_dst, prev_kb); }
    flush_rows(prev);
    MNN_SEG_STAMP();
    if (SKEW && wc == 0) __builtin_amdgcn_s_barrier();                       // X's closing phase pairs with Y's last E phase
    if (DBG == 0 && jb.dbg && w == 0 && t == 0) { jb.dbg[0] = (u64)(clock64() - sclk0); jb.dbg[1] = (u64)(wall_clock64() - wall0); }
    if (DBG == 3 && lane == 0) jb.job[0].partR[(size_t)w * 8 + wave] = (u64)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // hwreg(HW_REG_HW_ID, 0, 32)
#undef MNN_LAUNDER
#undef MNN_SEG_STAMP
}
