// mnn_prepost.hip.h -- the kernels around the match GEMM: k_rownorm_img (normalise + panel images) and k_mnn_post
// (second arg-max level, mutual check, ordered output).  Included by kernels_match.hip and tools/probes/mnn_probe.hip.
#pragma once
#include "mnn_layout.h"
#include "mnn_seg_plan.h"

// k_rownorm_img: F::normalize of both descriptor sets (ORBmatcher.cc:358-359: fp64 sum of squares, fp32
// sqrt / max(.,1e-12) / divide) written as panel images; rows past the end of a set are written as zeros
// (the GEMM masks their products).  n2 == 0: one set only (the prepare form).  16 lanes per row, 16 rows per block;
// blocks [0, P1*16) serve d1, the rest d2 (P = panels of the set).
__global__ __launch_bounds__(256)
void k_rownorm_img(const float* __restrict__ d1, int n1, const float* __restrict__ d2, int n2, int P1,
                   float* __restrict__ img1, float* __restrict__ img2) {
    const int t = threadIdx.x, sub = t & 15;
    int blk = blockIdx.x;
    const bool second = blk >= P1 * 16;
    if (second) blk -= P1 * 16;
    const int row = blk * 16 + (t >> 4);
    const float* d = second ? d2 : d1;
    const int n = second ? n2 : n1;
    float* img = second ? img2 : img1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < n) v = *(const f32x4*)(d + (size_t)row * 64 + sub * 4);
    mnn_emit_row(v, row, sub, img);
}

// panel base + 16 * position of a row, and its swizzle term
__device__ __forceinline__ const float* mnn_row(const float* img, int row, int& s) {
    const int pos = mnn_pos(row & (MNN_PANEL - 1));
    s = mnn_swz(pos);
    return img + (size_t)(row >> 8) * MNN_PANEL_FLOATS + pos * 16;
}

// k_mnn_post: second level of the arg-max, the mutual check (ORBmatcher.cc:367-372), the min_cossim gate (:361)
// and the ordered output (:371-403) in one launch.  Blocks 0 .. gridDim.x-2: sixteen lanes per d1 row, sixteen rows
// per workgroup (= one d1 row group of the bestC keys).
//   bestR[row] = max over the planes partR[.][row] = (M, gc): the row maximum M sits in the MNN_CGROUP = 4 d2 rows 4*gc .. 4*gc+3 -> lanes 0 .. 3 of the
//   row recompute <row, 4*gc + c>; the first one equal to M is m12[row] (torch.max returns the first index of the maximum).
//   bestC[col] = max over the planes partC[.][col] = (Mc, gr): m21[col] is the first d1 row of 16*gr .. 16*gr+15 whose dot product equals Mc.  `row` is a
//   mutual match iff Mc == M, row lies in that group and no earlier row of the group reaches Mc -- lane c recomputes
//   <16*gr + c, col> for the rows before `row` only; those rows are the workgroup's own 16 rows.
// All global reads are cooperative and coalesced (one 16-byte piece per lane; a 16-lane group fetches one 256-byte row per instruction)
// and go through LDS: the workgroup's own 16 d1 rows once (sA), and for every row its 4 candidate d2 rows (sB); the column keys of the four
// candidates are dealt out over the row's 16 lanes (4 planes each at 4096 rows).  Two dependent round trips remain: bestR -> candidates (+ their
// bestC keys).  Round 5: candidate groups of 4 instead of 16 d2 rows (the GEMMs name the quarter of the winning group of 16: one LDS read and three
// selects per row in their epilogues) -- a d1 row fetches 1.7 KB instead of 6.3 KB from L2 / Infinity Cache, which is what the kernel's time was
// (k_mnn_post_batch: 25 us for 8 pairs = 8 TB/s), and the workgroup's LDS drops from 74 to 22 KB.
// Each row publishes (column or -1, value) as one 8-byte agent-scope atomic store.  The last `ncoll` blocks are the
// collectors (grid = nb + ncoll, nb = ceil(n1/16)): they poll the pairs (agent-scope atomic loads; MNN_PAIR_EMPTY = not
// yet written; the data is its own flag, so no ticket and no fence) and write the matches in ascending idx1 with
// dist = sqrt(2 (1 - cos)).  The writers never wait, so the collectors cannot dead-lock whatever the dispatch order
// is; their spin is bounded.  mnn_ncoll(n1) = min(16, blocks of 256 rows).
#define MNN_PAIR_EMPTY 0xFFFFFFFE00000000ull
#define MNN_SPIN_LIMIT (1 << 22)
__host__ __device__ inline int mnn_ncoll(int n1) { const int nq = (n1 + 255) >> 8; return nq < 16 ? (nq < 1 ? 1 : nq) : 16; }
// LDS rows of k_mnn_post: 64 floats, no padding; the 16-byte chunk q of row r sits at chunk q ^ (r & 15), so that the lanes of a wave that read chunk q of 16
// DIFFERENT rows (the candidate dot products, the mutual check) hit 16 different chunk positions -- two lanes per bank pair instead of sixteen.  Without padding
// a workgroup takes (16 + 64) x 256 B = exactly 20 KB: eight workgroups per CU (with the 56 registers the kernel needs: eight waves per SIMD), so the 2048
// writer workgroups of an 8-pair call are resident in ONE round (the padded form, 21.25 KB, fitted seven: the last 256 workgroups paid a second full latency chain).
#define MNN_POST_LD 64
#define MNN_POST_LDS ((16 + 16 * MNN_CGROUP) * MNN_POST_LD * 4)
__device__ __forceinline__ float* mnn_post_chunk(float* row_base, int row, int q) { return row_base + ((q ^ (row & 15)) << 2); }
__device__ __forceinline__ const float* mnn_post_chunk(const float* row_base, int row, int q) { return row_base + ((q ^ (row & 15)) << 2); }
// everything one match needs from k_mnn_post.  segG > 0: the key planes come from k_mnn_gemm_seg (mnn_gemm_seg.hip.h) -- the number of
// row planes of a d1 panel follows from (segT, segG, tile0, P2) with the GEMM's own arithmetic and `npr` is ignored.
struct MnnPostArgs {
    const float* img1; const float* img2;
    const u64* partR; const u64* partC; u64* pairs;
    int* idx1; int* idx2; float* dist; int* n_matches;
    long long* stamps; const int* hdr1; const int* hdr2;
    size_t ldr, ldc;
    int n1, n2, npr, npc, nb, ncoll;
    int segT, segG, tile0, P2;
    float min_cossim; int pad;
};
template <int TS>            // TS (probes only): wall-clock stamps of block 0 / the collector into `stamps`
__device__ __forceinline__ void mnn_post_body(const MnnPostArgs& a, const int bid, const int nblk) {
    const float* __restrict__ img1 = a.img1; const float* __restrict__ img2 = a.img2;
    const u64* __restrict__ partR = a.partR; const u64* __restrict__ partC = a.partC; u64* __restrict__ pairs = a.pairs;
    int* __restrict__ idx1 = a.idx1; int* __restrict__ idx2 = a.idx2; float* __restrict__ dist = a.dist; int* __restrict__ n_matches = a.n_matches;
    long long* __restrict__ stamps = a.stamps; const int* __restrict__ hdr1 = a.hdr1; const int* __restrict__ hdr2 = a.hdr2;
    const size_t ldr = a.ldr, ldc = a.ldc;
    const int n1 = a.n1, n2 = a.n2, npc = a.npc, nb = a.nb, ncoll = a.ncoll;
    const float min_cossim = a.min_cossim;
    extern __shared__ __attribute__((aligned(16))) float spost[];
    const int t = threadIdx.x;
#define MNN_STAMP(k) do { if (TS && t == 0 && (bid == 0 || bid + 1 == nblk)) stamps[(bid ? 16 : 0) + (k)] = wall_clock64(); } while (0)
    MNN_STAMP(0);
    if (bid < nb) {
        float* sA = spost;                               // [16 rows][64]: row in natural piece order (piece p = 2g + half is chunk p), chunks swizzled (mnn_post_chunk)
        float* sB = spost + 16 * MNN_POST_LD;            // [16 rows][MNN_CGROUP candidates][64], chunks swizzled
        const int c = t & 15, grp = t >> 4;
        const int row = bid * 16 + grp;                  // the image holds whole panels: rows up to the panel end are readable (zeros)
        // bestR[row] = maximum over the npr planes the GEMM blocks of this d1 panel wrote: lane c takes planes c, c+16, ...
        int npr = a.npr;
        if (a.segG > 0) npr = mnn_seg_planes(a.tile0 + (row >> 8) * a.P2, a.P2, a.segT, a.segG);      // uniform over the workgroup (16 rows of one panel)
        u64 kr = 0ull;
        if (row < n1) for (int pl = c; pl < npr; pl += 16) kr = mnn_umax64(kr, partR[(size_t)pl * ldr + row]);
        kr = mnn_umax64(kr, __shfl_xor(kr, 1)); kr = mnn_umax64(kr, __shfl_xor(kr, 2)); kr = mnn_umax64(kr, __shfl_xor(kr, 4)); kr = mnn_umax64(kr, __shfl_xor(kr, 8));
        {
            int sa;
            const float* ra = mnn_row(img1, row, sa);
            *(f32x4*)mnn_post_chunk(sA + grp * MNN_POST_LD, grp, c) = *(const f32x4*)(ra + (c >> 2) * 4096 + (((c & 3) ^ sa) << 2));
        }
        const float M = ord2f((unsigned)(kr >> 32));
        const int gc = (int)(0xFFFFFFFFu - (unsigned)(kr & 0xFFFFFFFFull));
        // the row maximum sits in the MNN_CGROUP = 4 d2 rows 4*gc .. 4*gc+3 (a group never straddles a panel).  Lane c serves candidate cj = c & 3:
        // bestC[its column] = maximum over the npc planes, the planes dealt out over the four lanes that share a candidate (c >> 2, + 4, ...)
        const int cj = c & 3;
        const int col = gc * MNN_CGROUP + cj;
        const bool have = kr != 0ull && col < n2;
        u64 kcand = 0ull;
        if (have) {
            int pl = c >> 2;
            for (; pl + 12 < npc; pl += 16) {             // four independent loads in flight
                u64 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = partC[(size_t)(pl + 4 * u) * ldc + col];
#pragma unroll
                for (int u = 0; u < 4; ++u) kcand = mnn_umax64(kcand, v[u]);
            }
            for (; pl < npc; pl += 4) kcand = mnn_umax64(kcand, partC[(size_t)pl * ldc + col]);
        }
        kcand = mnn_umax64(kcand, __shfl_xor(kcand, 4)); kcand = mnn_umax64(kcand, __shfl_xor(kcand, 8));      // every lane: bestC of candidate c & 3
        if (kr != 0ull) {                                // the four candidate rows (inside the panel image: readable), one 256-byte row per instruction and 16 lanes
            f32x4 pv[MNN_CGROUP];
#pragma unroll
            for (int j = 0; j < MNN_CGROUP; ++j) {
                int sb;
                const float* rb = mnn_row(img2, gc * MNN_CGROUP + j, sb);
                pv[j] = *(const f32x4*)(rb + (c >> 2) * 4096 + (((c & 3) ^ sb) << 2));
            }
#pragma unroll
            for (int j = 0; j < MNN_CGROUP; ++j) *(f32x4*)mnn_post_chunk(sB + (grp * MNN_CGROUP + j) * MNN_POST_LD, grp * MNN_CGROUP + j, c) = pv[j];
        }
        MNN_STAMP(1);
        __syncthreads();
        // <row, candidate>: one fp32 fma chain in k order -- the arithmetic of the MFMA loop (k = 2j of lane-half 0, then 2j+1); lanes 0 .. 3 of the row
        const float* pa = sA + grp * MNN_POST_LD;
        const int rbi = grp * MNN_CGROUP + cj;
        const float* pb = sB + rbi * MNN_POST_LD;
        float dv = 0.f;
        if (c < MNN_CGROUP) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 a0 = *(const f32x4*)mnn_post_chunk(pa, grp, 2 * g), a1 = *(const f32x4*)mnn_post_chunk(pa, grp, 2 * g + 1);
                const f32x4 b0 = *(const f32x4*)mnn_post_chunk(pb, rbi, 2 * g), b1 = *(const f32x4*)mnn_post_chunk(pb, rbi, 2 * g + 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) { dv = fmaf(a0[j], b0[j], dv); dv = fmaf(a1[j], b1[j], dv); }
            }
        }
        MNN_STAMP(2);
        unsigned eq = (c < MNN_CGROUP && have && dv == M) ? (1u << c) : 0u;
        eq |= __shfl_xor(eq, 1); eq |= __shfl_xor(eq, 2); eq |= __shfl_xor(eq, 4); eq |= __shfl_xor(eq, 8);
        int cs;
        if (eq) cs = __builtin_ctz(eq);
        else {      // cannot happen while the recomputation is bit-identical (NaN rows aside); stay deterministic anyway
            u64 k = (c < MNN_CGROUP && have) ? mnn_pack_key(dv, (unsigned)c) : 0ull;
            k = mnn_umax64(k, __shfl_xor(k, 1)); k = mnn_umax64(k, __shfl_xor(k, 2)); k = mnn_umax64(k, __shfl_xor(k, 4)); k = mnn_umax64(k, __shfl_xor(k, 8));
            cs = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)) & (MNN_CGROUP - 1);
        }
        const int cstar = gc * MNN_CGROUP + cs;
        const u64 kc = __shfl(kcand, (t & 48) | cs);                       // the lane of this wave that holds column cstar
        const float Mc = ord2f((unsigned)(kc >> 32));
        const int gr = (int)(0xFFFFFFFFu - (unsigned)(kc & 0xFFFFFFFFull));
        bool mutual = false;
        if (kr != 0ull && cstar < n2 && kc != 0ull && Mc == M && (row >> 4) == gr) {      // uniform over the 16 lanes of a row
            const float* pg = sA + c * MNN_POST_LD;                        // d1 row 16*gr + c is row c of this workgroup
            const int rsi = grp * MNN_CGROUP + cs;
            const float* ps = sB + rsi * MNN_POST_LD;
            float d2v = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 a0 = *(const f32x4*)mnn_post_chunk(pg, c, 2 * g), a1 = *(const f32x4*)mnn_post_chunk(pg, c, 2 * g + 1);
                const f32x4 b0 = *(const f32x4*)mnn_post_chunk(ps, rsi, 2 * g), b1 = *(const f32x4*)mnn_post_chunk(ps, rsi, 2 * g + 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) { d2v = fmaf(a0[j], b0[j], d2v); d2v = fmaf(a1[j], b1[j], d2v); }
            }
            unsigned earlier = (c < grp && d2v == Mc) ? 1u : 0u;
            earlier |= __shfl_xor(earlier, 1); earlier |= __shfl_xor(earlier, 2); earlier |= __shfl_xor(earlier, 4); earlier |= __shfl_xor(earlier, 8);
            mutual = earlier == 0u;
        }
        MNN_STAMP(3);
        if (min_cossim > 0.f) mutual = mutual && (M > min_cossim);
        // n_valid-aware option: the rows are the nfeatures slots of two extraction records; a pair that touches a padding
        // slot is not reported.  Valid slots of a record: [0, mono_index) and [n - (n_valid - mono_index), n) (header: n_valid, mono_index).
        if (hdr1) {
            const int nv1 = hdr1[0], mo1 = hdr1[1], nv2 = hdr2[0], mo2 = hdr2[1];
            mutual = mutual && (row < mo1 || row >= n1 - (nv1 - mo1)) && (cstar < mo2 || cstar >= n2 - (nv2 - mo2));
        }
        if (c == 0 && row < n1) {
            const u64 pr = ((u64)(unsigned)(mutual ? cstar : -1) << 32) | (u64)__builtin_bit_cast(unsigned, M);
            __hip_atomic_store(pairs + row, pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        MNN_STAMP(4);
        return;
    }
    // ---- collectors.  Collector k owns the rows [k*span, (k+1)*span) (span = a multiple of 256) and writes their matches;
    // it polls every row below its upper end: the rows before its range are only counted (they give its output offset),
    // so the collectors need nothing from each other.  Thread t reads rows 256 q + t (coalesced); inside a block of 256
    // rows the output order comes from a ballot scan in the wave plus the counts of the lower waves (LDS).
    // The pairs are re-armed by the NEXT call's k_mnn_gemm_img (a collector must not: the others still read them).
    int* wcnt = (int*)spost;                             // [qb own][4 waves], then [4] for the count of the rows before
    const int lane = t & 63, wave = t >> 6;
    const int k = bid - nb;                              // collector index
    const int nq = (n1 + 255) >> 8;                      // blocks of 256 rows
    const int qspan = (nq + ncoll - 1) / ncoll;
    const int q_lo = (k * qspan < nq) ? k * qspan : nq, q_hi = (q_lo + qspan < nq) ? q_lo + qspan : nq;
    const u64 below = (1ull << lane) - 1ull;
    int before_w = 0;                                    // matches in rows < 256*q_lo seen by THIS wave (same in all its lanes)
    bool timeout = false;
    for (int q0 = 0; q0 < q_lo; q0 += 8) {               // (8 blocks of 256 rows at a time: 16 cost the writers' path its eighth wave per SIMD in registers)
        u64 pr[8];
        for (int spin = 0;; ++spin) {                   // 8 loads unconditionally: they are in flight together
            bool pending = false;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = (q0 + u) * 256 + t;
                pr[u] = (q0 + u < q_lo && i < n1) ? __hip_atomic_load(pairs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) pending = pending || pr[u] == MNN_PAIR_EMPTY;
            if (!pending) break;
            if (spin > MNN_SPIN_LIMIT) { timeout = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) before_w += __popcll(__ballot((int)(unsigned)(pr[u] >> 32) >= 0 && pr[u] != MNN_PAIR_EMPTY));
    }
    if (TS) MNN_STAMP(1);
    int run = 0;
    for (int q0 = q_lo; q0 < q_hi; q0 += 4) {            // own rows, four blocks of 256 at a time
        u64 pr[4];
        for (int spin = 0;; ++spin) {
            bool pending = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = (q0 + u) * 256 + t;
                pr[u] = (q0 + u < q_hi && i < n1) ? __hip_atomic_load(pairs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) pending = pending || pr[u] == MNN_PAIR_EMPTY;
            if (!pending) break;
            if (spin > MNN_SPIN_LIMIT) { timeout = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (TS) MNN_STAMP(2);
        int pre[4];
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();      // wcnt of the previous round has been read (LDS only: no vmcnt wait)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool m = (int)(unsigned)(pr[u] >> 32) >= 0 && pr[u] != MNN_PAIR_EMPTY;
            const u64 bal = __ballot(m);
            pre[u] = m ? __popcll(bal & below) : -1;
            if (lane == 0) wcnt[u * 4 + wave] = __popcll(bal);
        }
        if (q0 == q_lo && lane == 0) wcnt[16 + wave] = before_w;
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
        if (q0 == q_lo) run = wcnt[16] + wcnt[17] + wcnt[18] + wcnt[19];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int bef = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const int v = wcnt[u * 4 + w]; if (w < wave) bef += v; tot += v; }
            if (pre[u] >= 0) {
                const int off = run + bef + pre[u];
                idx1[off] = (q0 + u) * 256 + t; idx2[off] = (int)(unsigned)(pr[u] >> 32);
                const float cd = 1.0f - __builtin_bit_cast(float, (unsigned)(pr[u] & 0xFFFFFFFFull));
                dist[off] = sqrtf(2.0f * cd);
            }
            run += tot;
        }
    }
    if (k == ncoll - 1) {                                // the last collector has seen every row
        if (q_lo >= q_hi) {                              // (it owns no rows when nq < ncoll * qspan: its count is all in before_w)
            if (lane == 0) wcnt[16 + wave] = before_w;
            __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
            run = wcnt[16] + wcnt[17] + wcnt[18] + wcnt[19];
        }
        if (__syncthreads_or(timeout ? 1 : 0)) run = -1;   // a writer never showed up (cannot happen): report it instead of hanging
        if (t == 0) *n_matches = run;
    }
    MNN_STAMP(10);
#undef MNN_STAMP
}

template <int TS>
__global__ __launch_bounds__(256, 8)
void k_mnn_post(const float* __restrict__ img1, int n1, const float* __restrict__ img2, int n2,
                const u64* __restrict__ partR, size_t ldr, int npr, const u64* __restrict__ partC, size_t ldc, int npc, float min_cossim,
                u64* __restrict__ pairs, int nb, int ncoll, int* __restrict__ idx1, int* __restrict__ idx2, float* __restrict__ dist, int* __restrict__ n_matches,
                long long* __restrict__ stamps, const int* __restrict__ hdr1, const int* __restrict__ hdr2) {
    MnnPostArgs a;
    a.img1 = img1; a.img2 = img2; a.partR = partR; a.partC = partC; a.pairs = pairs; a.idx1 = idx1; a.idx2 = idx2; a.dist = dist; a.n_matches = n_matches;
    a.stamps = stamps; a.hdr1 = hdr1; a.hdr2 = hdr2; a.ldr = ldr; a.ldc = ldc; a.n1 = n1; a.n2 = n2; a.npr = npr; a.npc = npc; a.nb = nb; a.ncoll = ncoll;
    a.segT = 0; a.segG = 0; a.tile0 = 0; a.P2 = 0; a.min_cossim = min_cossim; a.pad = 0;
    mnn_post_body<TS>(a, (int)blockIdx.x, (int)gridDim.x);
}

// the matches of up to MNN_MAX_JOBS pairs in one launch: blockIdx.y = pair (the table sits in the kernel arguments), blockIdx.x = block of that
// pair's own grid (writers 0 .. nb-1, then its collectors); blocks past a pair's count leave at once.  Workgroups are dispatched in
// ascending (y, x) order, so a pair's collectors only ever wait for writers that were dispatched before them.
// ASSUMPTION: HIP does not promise that dispatch order.  What the kernel relies on is weaker: the writers never wait for anybody, so a
// collector can only be kept waiting by writers that have not been dispatched yet, and that needs every slot of the GPU to be held by spinning
// collectors -- at most 16 per pair x 16 pairs = 256 workgroups of 4 waves against 2048 workgroup slots (MNN_POST_LDS = 20 KB of LDS each: eight per CU), so writers always
// find room.  Should a writer still never show up, the collectors' spin is bounded (MNN_SPIN_LIMIT) and the pair reports n_matches = -1, which the host
// API turns into an error instead of a hang.
struct MnnPostBatch { MnnPostArgs job[MNN_MAX_JOBS]; };
__global__ __launch_bounds__(256, 8)
void k_mnn_post_batch(const MnnPostBatch pb) {
    const MnnPostArgs& a = pb.job[blockIdx.y];
    const int nblk = a.nb + a.ncoll;
    if ((int)blockIdx.x >= nblk) return;
    mnn_post_body<0>(a, (int)blockIdx.x, nblk);
}
