// conv_halo.hip — halo-staged ping-pong implicit GEMM for the generator's deep 3x3 layers (split-fp16 arithmetic; 56^2 x 128, 28^2 x 256,
// 14^2 x 512 channels): conv_pp.hip's 8-wave schedule with the A (im2col) operand staged ONCE per 32-channel chunk instead of once per tap.
//
// Why.  conv_pp_kernel streams nine shifted copies of every input pixel through LDS (one 256-row x 128-byte A image per (tap, channel chunk)):
// 36 of a wave's 54 LDS-DMA instructions per channel chunk, the lever its own ablation identified (no A DMA: -19 % time) and
// the 2x over-fetch the PMC counters show.  Here GEMM rows are image pixels in RASTER order, so the tile's rows m0 .. m0+255 are consecutive
// NHWC pixels and tap (ky, kx) of row r is the pixel (ky-1) W + (kx-1) further on: ONE halo of 256 + 2 (W + 1) consecutive pixels x 32 channels
// serves all nine taps.  A lane reads its A fragment of tap t from halo row  r + (W+1) + dy_t W + dx_t; at image borders dy / dx are folded back
// (reflect padding) or the lane is pointed at a row of zeros (zero padding) — a per-lane table of 9 x 2 LDS offsets computed once per tile.
//   * A traffic L2 -> LDS per channel chunk: 1.12-1.45 x the tile (W = 14 .. 56) instead of 9 x; DMA instructions per wave and channel
//     chunk 5-6 + 18 (A + B) instead of 54; no per-chunk address VALU (the halo's per-lane source offset is one shift-add of a constant).
//   * B (weights) keeps conv_pp's 3-stage ring (one 128 x 32-k chunk per (tap, channel chunk)); the halo is double-buffered across channel
//     chunks: the pieces of chunk cc+1 are issued one per tap during the first taps of chunk cc.
//   * Same K order (channel-chunk-major, taps back to back) and the same per-accumulator MFMA sequence as conv_igemm_kernel's channel-major
//     walk and conv_pp_kernel: BIT-IDENTICAL results (tests/test_conv_gpu.py).
// LDS: B ring 48 KB + 2 halo buffers of 8 NPA KB (+ a 128-byte zero row each) = 128 KB (NPA 5: W <= 31) or 144 KB (NPA 6: W <= 63); one
// 512-thread workgroup per CU, the two waves of a SIMD one phase apart (conv_pp.hip header has the ordering rules; they apply unchanged:
// the halo buffer of chunk cc+1 was last read in chunk cc-1, whose last load phase every wave left through a barrier with lgkmcnt(0)).
//
// Bound: MFMA (dense fp16 2.5 PF, 3 MFMAs per product).  Algorithmic flop = 2*M*N*K per launch.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "conv_common.h"

#define HS_BM 256
#define HS_BN 128
#define HS_NSTAGE 3
#define HS_BRING_BYTES (HS_NSTAGE * HS_BN * 128) /* 49,152 B */
#define HS_EPI_LD (2 * 32 + 4)
#define HS_DEFAULT_EB 0

typedef __attribute__((address_space(3))) void* hs_lptr_t;

// Phase timeline (tools/halo_timeline.py), compiled only into -DSMIRK_DEBUG_HOOKS variant builds (tools/build_variant.sh): waves 0 and 4 of a few
// workgroups stamp s_memtime at the four phase boundaries of every chunk into spare LDS and dump it at the end of the kernel.
#ifdef SMIRK_DEBUG_HOOKS
__device__ long long* g_hs_dbg = nullptr;                            // [8 workgroups][2 waves][HS_DBG_N] stamps
#define HS_DBG_N 1024
#define HS_DBG_LDS (2 * HS_DBG_N * 8)
#define HS_STAMP()                                                                                             \
    do {                                                                                                       \
        if (dbg_on && dbg_i < HS_DBG_N) { dbg_lds[dbg_i] = (long long)__builtin_readcyclecounter(); ++dbg_i; } \
    } while (0)
#else
#define HS_DBG_LDS 0
#define HS_STAMP() do {} while (0)
#endif

template <int K>
using hs_ic = std::integral_constant<int, K>;

// byte offset inside the B ring of (row, stage, 16-byte piece): blocks of 8 rows, the three stages of a block adjacent (conv_pp.hip)
__device__ __forceinline__ int hs_b_off(int row, int stage, int piece) {
    return ((((row >> 3) * HS_NSTAGE + stage) << 8) + ((row & 7) << 5) + ((piece ^ ((row >> 1) & 7)) << 2)) * 4;
}

// NPA: halo pieces (1 KiB = 8 pixel rows of 128 B) per wave and channel chunk; the halo buffer holds 64 NPA rows >= 256 + 2 (W + 1)
// EB:  "early barrier" — how many of a matrix phase's 24 MFMAs are issued AFTER the barrier that ends it (0, 4 or 8).  The phase timeline
//      (tools/halo_timeline.py) shows the matrix pipe idle for ~130 cycles at every hand-over: the finishing wave reaches the barrier, the barrier
//      releases, the partner wakes up and issues its first MFMA.  With the barrier placed EB MFMAs before the end, the finishing wave's tail keeps the
//      pipe busy while the partner wakes up.  Nothing the barrier orders depends on the position of a wave's MFMAs (they only touch registers).
// X1:  one fp16 MFMA per product block on the hi halves only (the training path's "f16x1" arithmetic, generator_train.TRAIN_ARITH): the lo fragments are neither
//      read from LDS nor multiplied — 8 of the 24 MFMAs and 8 of the 16 fragment reads of a chunk remain.  Same K order as conv_igemm_kernel's X1 instantiation.
template <int NPA, int EB, bool X1, bool STATS = false>
__device__ __forceinline__ void conv_halo_body(const ConvArgs& a) {
    constexpr int NL = 3;   // DMA instructions issued in the load phase: 1-3 measured equal, 0 (all among the MFMAs) 2-3 % slower (profiles/r03b_halo_nl_sweep.txt)
    extern __shared__ __attribute__((aligned(16))) float hs_smem[];
    char* lds = (char*)hs_smem;
    constexpr int ABUF = NPA * 8 * 1024;                            // one halo buffer
    constexpr int ASTRIDE = ABUF + 128;                             // + its row of zeros (byte offset ABUF inside the buffer)
    constexpr int A0 = HS_BRING_BYTES;                              // halo buffer u starts at A0 + u * ASTRIDE
    const SmirkConvDesc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;                       // 4 (M) x 2 (N) waves of 64 x 64
    const int group = swave >> 2;                                    // 0: leads, 1: one phase behind (shares each SIMD with a wave of group 0)
    const int ntn = a.N / HS_BN;
    const int logical = xcd_logical(blockIdx.x, gridDim.x);
    const int m0 = (logical / ntn) * HS_BM, n0 = (logical % ntn) * HS_BN;
    const int W = d.W, HW = d.H * d.W;
    const int fr = lane & 31, hb = lane >> 5;

#ifdef SMIRK_DEBUG_HOOKS
    const int dbg_slot = (blockIdx.x % 97 == 0) ? (int)(blockIdx.x / 97) : -1;
    const bool dbg_on = g_hs_dbg != nullptr && dbg_slot >= 0 && dbg_slot < 8 && (tid == 0 || tid == 256);
    long long* dbg_lds = (long long*)(lds + A0 + 2 * ASTRIDE) + (tid >> 8) * HS_DBG_N;
    int dbg_i = 0;
    const long long dbg_t_entry = (long long)__builtin_readcyclecounter();
#endif
    // ---- zero rows (read by every tap that falls into the zero padding) ------------------------------------------------------------------------
    if (tid < 64) *(float*)(lds + A0 + (tid >> 5) * ASTRIDE + ABUF + (tid & 31) * 4) = 0.f;

    unsigned tabA[9];                                              // per-lane tap table, filled in the prologue while the first DMAs fly
    // B fragment bases: (k-step 0, hi) piece of rows (wn*2 + j)*32 + fr in stage 0
    unsigned bB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bB[j] = (unsigned)hs_b_off((wn * 2 + j) * 32 + fr, 0, 2 * hb);

    // ---- DMA coordinates ------------------------------------------------------------------------------------------------------------------------
    const int pos = lane & 7, lrow = lane >> 3;                    // a wave-wide LDS-DMA writes 8 rows x 8 pieces lane-linearly
    // halo piece k of this wave = block swave + 8k = halo rows 8 (swave + 8k) + lrow; (row >> 1) & 7 does not depend on k
    const unsigned colA16 = (unsigned)(pos ^ ((4 * (swave & 1) + (lrow >> 1)) & 7)) * 16u;
    const int qv0 = m0 - (W + 1) + 8 * swave + lrow;                // linear pixel of piece 0's row (may be negative / beyond the tensor)
    const int npix = d.B * HW;
    unsigned voffB[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int brow = 8 * (swave + 8 * p) + lrow;
        voffB[p] = ((unsigned)min(n0 + brow, a.N - 1) * (unsigned)a.K + (unsigned)(pos ^ ((brow >> 1) & 7)) * 4u) * 4u;
    }
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.in0, (short)0, (int)((long long)npix * d.C0 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(d.C1 > 0 ? a.in1 : a.in0), (short)0,
                                                                         (int)((long long)npix * (d.C1 > 0 ? d.C1 : d.C0) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, (int)((long long)a.N * a.K * 4), 0x00020000);
    const int sh0 = 31 - __builtin_clz((unsigned)d.C0) + 2, sh1 = d.C1 > 0 ? 31 - __builtin_clz((unsigned)d.C1) + 2 : sh0;
    const int ncc = (d.C0 + d.C1) / CV_BK;

    // halo piece KP of channel chunk cc -> halo buffer at byte offset abase
    auto dmaA = [&](int cc, int abase, auto kc) {
        constexpr int KP = decltype(kc)::value;
        const int c0 = cc * CV_BK;
        const bool s1 = c0 >= d.C0;
        const int q = qv0 + 64 * KP;
        unsigned vo = ((unsigned)q << (s1 ? sh1 : sh0)) + colA16;
        vo = ((unsigned)q < (unsigned)npix) ? vo : 0x80000000u;  // outside the tensor: beyond num_records => the DMA writes zeros
        const int so = (s1 ? c0 - d.C0 : c0) * 4;
        char* dst = lds + abase + (swave + 8 * KP) * 1024;
        if (s1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (hs_lptr_t)dst, 16, vo, so, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (hs_lptr_t)dst, 16, vo, so, 0, 0);
    };
    // weight chunk (cc, TAP) -> ring stage ST: piece P (of 2) of this wave
    auto dmaB1 = [&](int cc, auto tapc, auto stc, auto pc) {
        constexpr int TAP = decltype(tapc)::value, ST = decltype(stc)::value, P = decltype(pc)::value;
        const int so = (TAP * a.Cin + cc * CV_BK) * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (hs_lptr_t)(lds + (((swave + 8 * P) * HS_NSTAGE + ST) << 10)), 16, voffB[P], so, 0, 0);
    };
    auto dmaB = [&](int cc, auto tapc, auto stc) {
        dmaB1(cc, tapc, stc, hs_ic<0>{});
        dmaB1(cc, tapc, stc, hs_ic<1>{});
    };

    f32x16 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    half8 ah[2][2], al[2][2], bh[2][2], bl[2][2];                   // [k-step][block]

    // one (load phase, matrix phase) pair for chunk (cc, TAP): A from the halo buffer at `acur`, B from ring stage TAP % 3
    auto body = [&](int cc, int acur, int anext, auto tapc) {
        constexpr int TAP = decltype(tapc)::value, ST = TAP % 3;
        // ---- load phase ------------------------------------------------------------------------------------------------------------------------
        HS_STAMP();
        {
            const unsigned t = tabA[TAP];
            const unsigned a0 = (t & 0xffffu) + (unsigned)acur, a1 = (t >> 16) + (unsigned)acur;
            ah[0][0] = *(const half8*)(lds + a0);
            ah[1][0] = *(const half8*)(lds + (a0 ^ 64u));
            ah[0][1] = *(const half8*)(lds + a1);
            ah[1][1] = *(const half8*)(lds + (a1 ^ 64u));
            if constexpr (!X1) {
                al[0][0] = *(const half8*)(lds + (a0 ^ 16u)); al[1][0] = *(const half8*)(lds + (a0 ^ 80u));
                al[0][1] = *(const half8*)(lds + (a1 ^ 16u)); al[1][1] = *(const half8*)(lds + (a1 ^ 80u));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned b0 = bB[j];
                bh[0][j] = *(const half8*)(lds + b0 + ST * 1024);
                bh[1][j] = *(const half8*)(lds + (b0 ^ 64u) + ST * 1024);
                if constexpr (!X1) {
                    bl[0][j] = *(const half8*)(lds + (b0 ^ 16u) + ST * 1024);
                    bl[1][j] = *(const half8*)(lds + (b0 ^ 80u) + ST * 1024);
                }
            }
        }
        // this phase pair's DMA list: [halo piece TAP of the NEXT channel chunk (taps 0 .. NPA-1),] the two weight pieces of the chunk two ahead.  Past
        // the end of K the clamped last chunk is fetched again into a buffer / stage nobody reads: no branch in the loop bodies.
        constexpr int T2 = (TAP + 2) % 9;
        constexpr int HASA = TAP < NPA ? 1 : 0, NDMA = 2 + HASA, NLD = NL < NDMA ? NL : NDMA;
        const int ccA = min(cc + 1, ncc - 1), ccB = min(cc + (TAP + 2 >= 9 ? 1 : 0), ncc - 1);
        auto dma = [&](auto kc) {                                    // DMA instruction K of the list
            constexpr int K = decltype(kc)::value;
            if constexpr (HASA && K == 0) dmaA(ccA, anext, hs_ic<(TAP < NPA ? TAP : 0)>{});
            else dmaB1(ccB, hs_ic<T2>{}, hs_ic<T2 % 3>{}, hs_ic<K - HASA>{});
        };
        if constexpr (NLD > 0) dma(hs_ic<0>{});
        if constexpr (NLD > 1) dma(hs_ic<1>{});
        if constexpr (NLD > 2) dma(hs_ic<2>{});
        // everything older than this load phase's DMA instructions has landed (this wave's share of chunk q+1's weights, and every halo piece issued
        // in an earlier phase); fragment reads done
        if constexpr (NLD == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else if constexpr (NLD == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else if constexpr (NLD == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        HS_STAMP();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        HS_STAMP();
        // ---- matrix phase: 24 MFMAs, operands in registers; dependent accumulations four instructions apart; the DMA instructions left over from
        //      the load phase go out one after every fourth MFMA (an LDS-DMA costs ~60 issue cycles among bare MFMAs, MI355X_MICROARCH.md) ---------
        __builtin_amdgcn_s_setprio(1);
        auto mm_dma = [&](auto gc) {
            constexpr int G = decltype(gc)::value;
            if constexpr (NLD + G < NDMA) {
                __builtin_amdgcn_sched_barrier(0);
                dma(hs_ic<(NLD + G < NDMA ? NLD + G : 0)>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto mblock = [&](auto bc) {                                 // MFMA block B (0-5) of the chunk: 4 independent accumulators
            constexpr int BK = decltype(bc)::value;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if constexpr (BK == 0) acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][i], bh[0][j], acc0[i][j], 0, 0, 0);
                    if constexpr (BK == 1 && !X1) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][i], bl[0][j], acc1[i][j], 0, 0, 0);
                    if constexpr (BK == 2 && !X1) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0][i], bh[0][j], acc1[i][j], 0, 0, 0);
                    if constexpr (BK == 3) acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][i], bh[1][j], acc0[i][j], 0, 0, 0);
                    if constexpr (BK == 4 && !X1) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][i], bl[1][j], acc1[i][j], 0, 0, 0);
                    if constexpr (BK == 5 && !X1) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[1][i], bh[1][j], acc1[i][j], 0, 0, 0);
                }
        };
        auto hand_over = [&]() {                                     // the barrier that ends the matrix phase
            HS_STAMP();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        mblock(hs_ic<0>{});
        mm_dma(hs_ic<0>{});
        mblock(hs_ic<1>{});
        mm_dma(hs_ic<1>{});
        mblock(hs_ic<2>{});
        mm_dma(hs_ic<2>{});
        mblock(hs_ic<3>{});
        if constexpr (EB == 8) hand_over();
        mblock(hs_ic<4>{});
        if constexpr (EB == 4) hand_over();
        mblock(hs_ic<5>{});
        __builtin_amdgcn_s_setprio(0);
        if constexpr (EB == 0) hand_over();
        else __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: the halo of chunk 0 and weight chunks 0 and 1 in flight, everything but weight chunk 1 landed; group 1 drops one phase behind ----
    dmaA(0, A0, hs_ic<0>{}); dmaA(0, A0, hs_ic<1>{}); dmaA(0, A0, hs_ic<2>{}); dmaA(0, A0, hs_ic<3>{}); dmaA(0, A0, hs_ic<4>{});
    if constexpr (NPA > 5) dmaA(0, A0, hs_ic<5>{});
    dmaB(0, hs_ic<0>{}, hs_ic<0>{});
    dmaB(0, hs_ic<1>{}, hs_ic<1>{});
    __builtin_amdgcn_sched_barrier(0);                             // the first operands are in flight while the tap table (4 integer divisions) is computed
    // ---- per-lane tap table: LDS byte offset (inside a halo buffer) of the (k-step 0, hi) piece of this lane's A row for each tap; rows i = 0 / 1
    //      of the wave tile in the low / high 16 bits.  The other three pieces of a row are offset ^ 16 (lo), ^ 64 (k-step 1), ^ 80. --------------
    {
        unsigned ent[2][9];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rl = (wm * 2 + i) * 32 + fr;
            const int m = m0 + rl;
            const int b = m / HW, rem = m - b * HW, y = rem / W, x = rem - y * W;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                int dy = t / 3 - 1, dx = t % 3 - 1;
                bool zero = m >= a.M;
                if (d.pad_mode == SMIRK_PAD_REFLECT) {
                    if (y + dy < 0 || y + dy >= d.H) dy = -dy;
                    if (x + dx < 0 || x + dx >= W) dx = -dx;
                } else {
                    zero = zero || y + dy < 0 || y + dy >= d.H || x + dx < 0 || x + dx >= W;
                }
                const int p = rl + (W + 1) + dy * W + dx;         // halo row
                const unsigned off = (unsigned)p * 128u + (unsigned)(((2 * hb) ^ ((p >> 1) & 7)) << 4);
                ent[i][t] = zero ? (unsigned)ABUF : off;
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) tabA[t] = ent[0][t] | (ent[1][t] << 16);
    }
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (group == 1) __builtin_amdgcn_s_barrier();
    for (int cc = 0; cc < ncc; ++cc) {
        const int acur = A0 + (cc & 1) * ASTRIDE, anext = A0 + ((cc + 1) & 1) * ASTRIDE;
        body(cc, acur, anext, hs_ic<0>{}); body(cc, acur, anext, hs_ic<1>{}); body(cc, acur, anext, hs_ic<2>{});
        body(cc, acur, anext, hs_ic<3>{}); body(cc, acur, anext, hs_ic<4>{}); body(cc, acur, anext, hs_ic<5>{});
        body(cc, acur, anext, hs_ic<6>{}); body(cc, acur, anext, hs_ic<7>{}); body(cc, acur, anext, hs_ic<8>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the over-fetched pieces must have landed before LDS is reused
    // ---- epilogue operands, requested BEFORE the closing barriers so that the loads fly while the workgroup drains ------------------------------
    // Every lane serves ONE 8-channel group (g = lane % 8) of 4 pixel rows per 32-row block: scale / shift are loaded once, and all eight residual
    // groups (2 blocks x 4 rows) are requested up front: the epilogue pays one memory round trip, not one per block (the fragment registers are free now).
    float* ebuf = hs_smem + wave * 32 * HS_EPI_LD;
    constexpr int GPR = 8, ITEMS = 32 * GPR / 64;
    const int eg = lane & 7, erow = lane >> 3;                       // item = it * 64 + lane: row = it * 8 + erow, group = eg
    const int en = n0 + wn * 64 + eg * 8;
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sf0 = {0.f, 0.f, 0.f, 0.f}, sf1 = sf0;
    if (a.scale) { sc0 = *(const f32x4*)(a.scale + en); sc1 = *(const f32x4*)(a.scale + en + 4); }
    if (a.shift) { sf0 = *(const f32x4*)(a.shift + en); sf1 = *(const f32x4*)(a.shift + en + 4); }
    half8 resh[2][ITEMS], resl[2][ITEMS];
    if (a.residual) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int m = min(m0 + (wm * 2 + i) * 32 + it * 8 + erow, a.M - 1);
                const size_t o = (size_t)m * d.Cout + en;
                resh[i][it] = *(const half8*)(a.residual + o);
                resl[i][it] = *(const half8*)(a.residual + o + 4);
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (group == 0) __builtin_amdgcn_s_barrier();                   // both groups have executed the same number of barriers ...
    __builtin_amdgcn_s_barrier();                                   // ... and every wave's DMA has retired

#ifdef SMIRK_DEBUG_HOOKS
    HS_STAMP();
    if (dbg_on)
        for (int k = 0; k < dbg_i; ++k) g_hs_dbg[(dbg_slot * 2 + (tid >> 8)) * HS_DBG_N + k] = dbg_lds[k];
    __builtin_amdgcn_s_barrier();
#endif
    // ---- epilogue: per-wave transpose through LDS, whole 8-channel groups, BN scale/shift + residual + ReLU, re-split ---------------------------
    HS_STAMP();
    SmirkRangeAcc rng;                                              // split-fp16 range audit (common.h): one running max per lane, tested once after the stores
    float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};                 // STATS (train mode): this lane's column sums over the wave's 64 rows (conv_common.h stats_block)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float x16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x16[r] = X1 ? acc0[i][j][r] : acc0[i][j][r] + acc1[i][j][r] * (1.0f / 2048.0f);
                ebuf[mfma32_row(r, lane) * HS_EPI_LD + j * 32 + fr] = x16[r];
            }
            if constexpr (STATS) stats_block(x16, lane, a.M - (m0 + (wm * 2 + i) * 32), st1[j], st2[j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int row = it * 8 + erow;
            const int m = m0 + (wm * 2 + i) * 32 + row;
            if (m < a.M) {
                float v[8];
                *(f32x4*)v = *(const f32x4*)(ebuf + row * HS_EPI_LD + eg * 8);
                *(f32x4*)(v + 4) = *(const f32x4*)(ebuf + row * HS_EPI_LD + eg * 8 + 4);
                const size_t o = (size_t)m * d.Cout + en;            // raster order: GEMM row m IS the NHWC pixel index
                if (a.scale) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] *= sc0[q]; v[4 + q] *= sc1[q]; }
                }
                if (a.shift) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] += sf0[q]; v[4 + q] += sf1[q]; }
                }
                if (a.residual) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += join1(resh[i][it][q], resl[i][it][q]);
                }
                if (d.act == SMIRK_ACT_RELU) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                half8 hi, lo;
                split8(v, hi, lo, rng);
                *(half8*)(a.out + o) = hi;
                *(half8*)(a.out + o + 4) = lo;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    rng.commit();
    if constexpr (STATS) {                                           // partial row (M tile, wave row 0..3): [N][2] floats, each (row, channel) written by one lane
        float* prow = a.stats + (size_t)((m0 / HS_BM) * 4 + wm) * a.N * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float t1 = stats_pair(st1[j]), t2 = stats_pair(st2[j]);
            const int n = n0 + wn * 64 + j * 32 + fr;
            if ((lane >> 5) == 0) { prow[n * 2] = t1; prow[n * 2 + 1] = t2; }
        }
    }
#ifdef SMIRK_DEBUG_HOOKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HS_STAMP();
    if (dbg_on) {
        long long* g = g_hs_dbg + (dbg_slot * 2 + (tid >> 8)) * HS_DBG_N;
        g[HS_DBG_N - 3] = dbg_lds[dbg_i - 2]; g[HS_DBG_N - 2] = dbg_lds[dbg_i - 1]; g[HS_DBG_N - 4] = dbg_t_entry;
    }
#endif
}

template <int NPA, int EB>
__global__ __launch_bounds__(512, 2) void conv_halo_kernel(ConvArgs a) { conv_halo_body<NPA, EB, false>(a); }
template <int NPA>
__global__ __launch_bounds__(512, 2) void conv_halo_x1_kernel(ConvArgs a) { conv_halo_body<NPA, 0, true>(a); }
// train mode: the same kernels leaving the BatchNorm partial sums of their raw output (ConvArgs::stats; one row per M tile and wave row)
template <int NPA, int EB>
__global__ __launch_bounds__(512, 2) void conv_halo_stats_kernel(ConvArgs a) { conv_halo_body<NPA, EB, false, true>(a); }
template <int NPA>
__global__ __launch_bounds__(512, 2) void conv_halo_x1_stats_kernel(ConvArgs a) { conv_halo_body<NPA, 0, true, true>(a); }

static int hs_env_mode() {                                           // $SMIRK_IGEMM_HALO: "0" off; unset / anything else = every eligible geometry
    const char* env = getenv("SMIRK_IGEMM_HALO");                   // read per call: tests toggle it
    return !env ? 1 : env[0] == '0' ? 0 : env[0] == 'a' ? 2 : 1;
}

// Serves: split-fp16, 3x3, stride 1, pad 1, NHWC out, both sources power-of-two multiples of 32 channels, N a multiple of 128, W <= 63, operands < 2 GiB.
bool smirk_conv_halo_eligible(const ConvArgs& a) {
    const SmirkConvDesc& d = a.d;
    const int mode = hs_env_mode();
    if (mode == 0) return false;
    if (d.KH != 3 || d.KW != 3 || d.stride != 1 || d.out_mode != SMIRK_OUT_NHWC || d.pad_t != 1 || d.pad_l != 1) return false;
    if (d.Ho != d.H || d.Wo != d.W || d.W > 63 || d.W < 2 || d.H < 2) return false;
    if (d.C0 % CV_BK || d.C1 % CV_BK || (d.C0 & (d.C0 - 1)) || (d.C1 & (d.C1 - 1))) return false;
    if (a.N % HS_BN || a.N < HS_BN) return false;
    const long long b0 = (long long)d.B * d.H * d.W * d.C0 * 4, b1 = (long long)d.B * d.H * d.W * d.C1 * 4, bw = (long long)a.N * a.K * 4;
    if (b0 >= (1ll << 31) || b1 >= (1ll << 31) || bw >= (1ll << 31)) return false;
    if (a.M < 4 * HS_BM) return false;                               // tiny problems stay on the 128-row tiles
    // Measured per layer against the kernels it replaces (tools/conv_sweep.py, profiles/r03c_halo_sweep.txt): at 1024 frames per pass every deep
    // layer gains 12-18 % (14x14: 421 -> 502-513 TFLOP/s, 28x28: 405-420 -> 487-523, 56x56: 361-398 -> 431-483); at 128 frames the layers with
    // K >= 2304 gain 3-13 % and the short-K layers (18-36 chunks per tile: the lone workgroup's prologue / epilogue are exposed) lose up to 8 % when timed
    // ALONE.  Rounds 3-4 therefore kept short-K layers with < 2048 tiles on the 128 x 128 tiles; inside the pipeline (other streams fill what a lone
    // workgroup per CU leaves idle) every eligible layer on this kernel measures +1.1 % at 128 frames per pass and +-0.1 % at 1024
    // (profiles/r04x_kernel_selection.txt; round 4 had seen the same +1 % on 4 hardware queues), so the threshold is gone.
    return true;
}

template <int NPA, int EB, bool X1>
static int hs_launch(const ConvArgs& a, hipStream_t st, size_t lds, int dev) {
    static bool attr_done[64] = {};                                  // hipFuncSetAttribute is per-device state (one process may drive several GPUs)
    const void* fn = X1 ? (const void*)conv_halo_x1_kernel<NPA> : (const void*)conv_halo_kernel<NPA, EB>;
    const void* fns = X1 ? (const void*)conv_halo_x1_stats_kernel<NPA> : (const void*)conv_halo_stats_kernel<NPA, EB>;
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(fns, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return SMIRK_ERR_LAUNCH;
        attr_done[dev] = true;
    }
    const int ntm = (a.M + HS_BM - 1) / HS_BM, ntn = a.N / HS_BN;
    if (g_smirk_prof_on) {
        const double px = (double)a.d.B * a.d.H * a.d.W;
        char nm[64];
        if (X1) snprintf(nm, sizeof(nm), "conv_halo_x1_kernel<%d>[256x128,8w,halo,f16x1]", NPA);
        else snprintf(nm, sizeof(nm), "conv_halo_kernel<%d,%d>[256x128,8w,halo]", NPA, EB);
        smirk_prof_next(nm, 2.0 * a.M * a.N * a.K,
                        4.0 * (px * a.Cin + (double)a.M * a.N + (double)a.N * a.K + (a.residual ? (double)a.M * a.N : 0.0)));
    }
    if (a.stats) {
        if constexpr (X1) SMIRK_LAUNCH((conv_halo_x1_stats_kernel<NPA>), dim3(ntm * ntn), dim3(512), lds, st, a);
        else SMIRK_LAUNCH((conv_halo_stats_kernel<NPA, EB>), dim3(ntm * ntn), dim3(512), lds, st, a);
        return smirk_launch_status();
    }
    if constexpr (X1) SMIRK_LAUNCH((conv_halo_x1_kernel<NPA>), dim3(ntm * ntn), dim3(512), lds, st, a);
    else SMIRK_LAUNCH((conv_halo_kernel<NPA, EB>), dim3(ntm * ntn), dim3(512), lds, st, a);
    return smirk_launch_status();
}

int smirk_conv_halo_launch(const ConvArgs& a, hipStream_t st, bool x1) {
    const int npa = (HS_BM + 2 * (a.d.W + 1) + 63) / 64;             // 64 halo rows per piece index (8 waves x 8 rows)
    const size_t lds = HS_BRING_BYTES + 2 * ((size_t)(npa <= 5 ? 5 : 6) * 8 * 1024 + 128) + HS_DBG_LDS;
#ifdef SMIRK_DEBUG_HOOKS
    {
        const char* e = getenv("SMIRK_HALO_DBG");                   // hex device address of a zeroed int64 buffer [8][2][HS_DBG_N], tools/halo_timeline.py
        long long* p = e ? (long long*)strtoull(e, nullptr, 16) : nullptr;
        if (hipMemcpyToSymbolAsync(HIP_SYMBOL(g_hs_dbg), &p, sizeof(p), 0, hipMemcpyHostToDevice, st) != hipSuccess) return SMIRK_ERR_LAUNCH;
    }
#endif
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return SMIRK_ERR_UNSUPPORTED;
    // (EB = MFMAs issued after the hand-over barrier: 4 / 8 measured 3-7 % slower without the time stamps in, profiles/r03b_halo_nl_sweep.txt, r04x_kernel_selection.txt;
    // the $SMIRK_HALO_EB switch and its instantiations left the library in round 5)
    if (x1) return npa <= 5 ? hs_launch<5, HS_DEFAULT_EB, true>(a, st, lds, dev) : hs_launch<6, HS_DEFAULT_EB, true>(a, st, lds, dev);
    return npa <= 5 ? hs_launch<5, HS_DEFAULT_EB, false>(a, st, lds, dev) : hs_launch<6, HS_DEFAULT_EB, false>(a, st, lds, dev);
}
