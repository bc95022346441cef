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
#define HS_DEFAULT_MAX_PIXELS 256                /* geometries the dispatcher sends here by default (H*W); "all" lifts it — set from measurements */

typedef __attribute__((address_space(3))) void* hs_lptr_t;

template <int K>
using hs_ic = std::integral_constant<int, K>;

// byte offset inside the B ring of (row, stage, 16-byte piece): blocks of 8 rows, the three stages of a block adjacent (conv_pp.hip)
__device__ __forceinline__ int hs_b_off(int row, int stage, int piece) {
    return ((((row >> 3) * HS_NSTAGE + stage) << 8) + ((row & 7) << 5) + ((piece ^ ((row >> 1) & 7)) << 2)) * 4;
}

// NPA: halo pieces (1 KiB = 8 pixel rows of 128 B) per wave and channel chunk; the halo buffer holds 64 NPA rows >= 256 + 2 (W + 1)
template <int NPA>
__global__ __launch_bounds__(512, 2) void conv_halo_kernel(ConvArgs a) {
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

    // ---- zero rows (read by every tap that falls into the zero padding) ------------------------------------------------------------------------
    if (tid < 64) *(float*)(lds + A0 + (tid >> 5) * ASTRIDE + ABUF + (tid & 31) * 4) = 0.f;

    // ---- per-lane tap table: LDS byte offset (inside a halo buffer) of the (k-step 0, hi) piece of this lane's A row for each tap; rows i = 0 / 1
    //      of the wave tile in the low / high 16 bits.  The other three pieces of a row are offset ^ 16 (lo), ^ 64 (k-step 1), ^ 80. --------------
    unsigned tabA[9];
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
    // weight chunk (cc, TAP) -> ring stage ST: 2 pieces per wave
    auto dmaB = [&](int cc, auto tapc, auto stc) {
        constexpr int TAP = decltype(tapc)::value, ST = decltype(stc)::value;
        const int so = (TAP * a.Cin + cc * CV_BK) * 4;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (hs_lptr_t)(lds + (((swave + 8 * p) * HS_NSTAGE + ST) << 10)), 16, voffB[p], so, 0, 0);
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
        {
            const unsigned t = tabA[TAP];
            const unsigned a0 = (t & 0xffffu) + (unsigned)acur, a1 = (t >> 16) + (unsigned)acur;
            ah[0][0] = *(const half8*)(lds + a0);        al[0][0] = *(const half8*)(lds + (a0 ^ 16u));
            ah[1][0] = *(const half8*)(lds + (a0 ^ 64u)); al[1][0] = *(const half8*)(lds + (a0 ^ 80u));
            ah[0][1] = *(const half8*)(lds + a1);        al[0][1] = *(const half8*)(lds + (a1 ^ 16u));
            ah[1][1] = *(const half8*)(lds + (a1 ^ 64u)); al[1][1] = *(const half8*)(lds + (a1 ^ 80u));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned b0 = bB[j];
                bh[0][j] = *(const half8*)(lds + b0 + ST * 1024);          bl[0][j] = *(const half8*)(lds + (b0 ^ 16u) + ST * 1024);
                bh[1][j] = *(const half8*)(lds + (b0 ^ 64u) + ST * 1024);  bl[1][j] = *(const half8*)(lds + (b0 ^ 80u) + ST * 1024);
            }
        }
        // halo piece TAP of the NEXT channel chunk (past the end of K the clamped last chunk is fetched again into the buffer nobody reads), then
        // the weight chunk two ahead
        if constexpr (TAP < NPA) dmaA(min(cc + 1, ncc - 1), anext, hs_ic<TAP>{});
        constexpr int T2 = (TAP + 2) % 9;
        dmaB(min(cc + (TAP + 2 >= 9 ? 1 : 0), ncc - 1), hs_ic<T2>{}, hs_ic<T2 % 3>{});
        // everything older than this phase's DMA instructions has landed (this wave's share of chunk q+1's weights, and every halo piece issued
        // in an earlier phase); fragment reads done
        if constexpr (TAP < NPA) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- matrix phase: 24 MFMAs, operands in registers; dependent accumulations four instructions apart -------------------------------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][i], bh[0][j], acc0[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][i], bl[0][j], acc1[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0][i], bh[0][j], acc1[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][i], bh[1][j], acc0[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][i], bl[1][j], acc1[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[1][i], bh[1][j], acc1[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: the halo of chunk 0 and weight chunks 0 and 1 in flight, everything but weight chunk 1 landed; group 1 drops one phase behind ----
    dmaA(0, A0, hs_ic<0>{}); dmaA(0, A0, hs_ic<1>{}); dmaA(0, A0, hs_ic<2>{}); dmaA(0, A0, hs_ic<3>{}); dmaA(0, A0, hs_ic<4>{});
    if constexpr (NPA > 5) dmaA(0, A0, hs_ic<5>{});
    dmaB(0, hs_ic<0>{}, hs_ic<0>{});
    dmaB(0, hs_ic<1>{}, hs_ic<1>{});
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
    if (group == 0) __builtin_amdgcn_s_barrier();                   // both groups have executed the same number of barriers ...
    __builtin_amdgcn_s_barrier();                                   // ... and every wave's DMA has retired

    // ---- epilogue (conv_pp.hip's): per-wave transpose through LDS, whole 8-channel groups, BN scale/shift + residual + ReLU, re-split ---------
    float* ebuf = hs_smem + wave * 32 * HS_EPI_LD;
    constexpr int GPR = 8, ITEMS = 32 * GPR / 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ebuf[mfma32_row(r, lane) * HS_EPI_LD + j * 32 + fr] = acc0[i][j][r] + acc1[i][j][r] * (1.0f / 2048.0f);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int item = it * 64 + lane, row = item / GPR, g = item % GPR;
            const int m = m0 + (wm * 2 + i) * 32 + row, n = n0 + wn * 64 + g * 8;
            if (m < a.M && n < a.N) {
                float v[8];
                *(f32x4*)v = *(const f32x4*)(ebuf + row * HS_EPI_LD + g * 8);
                *(f32x4*)(v + 4) = *(const f32x4*)(ebuf + row * HS_EPI_LD + g * 8 + 4);
                const size_t o = (size_t)m * d.Cout + n;             // raster order: GEMM row m IS the NHWC pixel index
                if (a.scale) {
                    const f32x4 s0 = *(const f32x4*)(a.scale + n), s1 = *(const f32x4*)(a.scale + n + 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] *= s0[q]; v[4 + q] *= s1[q]; }
                }
                if (a.shift) {
                    const f32x4 s0 = *(const f32x4*)(a.shift + n), s1 = *(const f32x4*)(a.shift + n + 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] += s0[q]; v[4 + q] += s1[q]; }
                }
                if (a.residual) {
                    const half8 rh = *(const half8*)(a.residual + o), rl = *(const half8*)(a.residual + o + 4);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += join1(rh[q], rl[q]);
                }
                if (d.act == SMIRK_ACT_RELU) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                half8 hi, lo;
                split8(v, hi, lo);
                *(half8*)(a.out + o) = hi;
                *(half8*)(a.out + o + 4) = lo;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

static int hs_env_mode() {                                           // $SMIRK_IGEMM_HALO: "0" off, "all" every eligible geometry, unset = measured default
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
    if (mode != 2 && (long long)d.Ho * d.Wo > HS_DEFAULT_MAX_PIXELS) return false;
    return a.M >= 4 * HS_BM;                                          // tiny problems stay on the 128-row tiles
}

int smirk_conv_halo_launch(const ConvArgs& a, hipStream_t st) {
    const int npa = (HS_BM + 2 * (a.d.W + 1) + 63) / 64;             // 64 halo rows per piece index (8 waves x 8 rows)
    const size_t lds = HS_BRING_BYTES + 2 * ((size_t)(npa <= 5 ? 5 : 6) * 8 * 1024 + 128);
    static bool attr_done[64] = {};                                  // hipFuncSetAttribute is per-device state (one process may drive several GPUs)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return SMIRK_ERR_UNSUPPORTED;
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute((const void*)conv_halo_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv_halo_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return SMIRK_ERR_LAUNCH;
        attr_done[dev] = true;
    }
    const int ntm = (a.M + HS_BM - 1) / HS_BM, ntn = a.N / HS_BN;
    if (g_smirk_prof_on) {
        const double px = (double)a.d.B * a.d.H * a.d.W;
        char nm[64];
        snprintf(nm, sizeof(nm), "conv_halo_kernel<%d>[256x128,8w,halo]", npa <= 5 ? 5 : 6);
        smirk_prof_next(nm, 2.0 * a.M * a.N * a.K,
                        4.0 * (px * a.Cin + (double)a.M * a.N + (double)a.N * a.K + (a.residual ? (double)a.M * a.N : 0.0)));
    }
    if (npa <= 5) SMIRK_LAUNCH(conv_halo_kernel<5>, dim3(ntm * ntn), dim3(512), lds, st, a);
    else SMIRK_LAUNCH(conv_halo_kernel<6>, dim3(ntm * ntn), dim3(512), lds, st, a);
    return smirk_launch_status();
}
