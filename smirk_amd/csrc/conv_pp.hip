// conv_pp.hip — "ping-pong" implicit-GEMM kernel for the generator's deep 3x3 layers (split-fp16 arithmetic, conv.hip header): the layers that
// are 60 % of the step (56^2 x 128, 28^2 x 256, 14^2 x 512 channels).
//
// Why a second kernel.  conv_igemm_kernel<128,128> runs two 4-wave workgroups per CU, each with a load -> barrier -> 24 MFMA loop; measured
//  it sustains ~41 % MFMA issue because (a) its operand DMA alone needs 67 % of the kernel's time (32 KB of L2 -> LDS traffic
// per 128x128x32 tile-chunk, ~16 TB/s chip-wide), and (b) nothing makes the two co-resident workgroups alternate between their load and their
// matrix phases, so the matrix pipe idles whenever both are loading.  This kernel changes both:
//   * ONE 8-wave workgroup per CU computes a 256(M) x 128(N) tile: 48 KB of DMA per tile-chunk for twice the work of a 128x128 chunk
//     (-25 % L2 -> LDS bytes per flop), wave tiles stay 64x64 (2 x 2 x 2 accumulator blocks = 128 VGPRs, 256-register budget at 2 waves/SIMD).
//   * The two waves that share a SIMD (wave w and w+4: MI355X_MICROARCH.md "Two waves per SIMD") run the SAME code one phase apart: while
//     waves 0-3 issue their 24 MFMAs of chunk q, waves 4-7 read their fragments of chunk q from LDS and enqueue the DMA of chunk q+2 — then
//     the roles swap.  Every phase boundary is one workgroup barrier; the offset is created by one extra barrier that waves 4-7 execute
//     before the loop (and waves 0-3 after it).  The matrix pipe of a SIMD is therefore always owned by exactly one of its two waves.
//   * Operands live in a 3-stage LDS ring (3 x 48 KB): chunk q is in stage q % 3, its DMA is issued two chunks ahead and only the DMA of chunk
//     q+1 is waited for (counted vmcnt: the 6 instructions of chunk q+2 stay in flight across the barrier).
// Ordering rules the schedule relies on (cdna_hip_programming.md §5 "256^2 8-phase template"; MI355X_MICROARCH.md item 7):
//   RAW  chunk q+1 is read (ds_read) in the load phase AFTER the barrier that follows every wave's `s_waitcnt vmcnt(6)` for it;
//   WAR  stage (q+2) % 3 = (q-1) % 3 is refilled only after a barrier that every reader of chunk q-1 reached with `lgkmcnt(0)`.
// K order: channel-chunk-major with the 9 taps of a 32-channel chunk back to back (the shifted re-reads of the same pixels hit L2), the same
// order as conv_igemm's KW_LEAN_CM walk, so both kernels accumulate in the same order and give the same bits.
// Operand addressing: `buffer_load_dwordx4 ... offen lds` — per-lane 32-bit byte offsets built from per-row tables, a scalar channel offset,
// out-of-range offsets for rows in the zero padding (the buffer unit writes zeros), XOR piece swizzle on the source side (conv_common.h).
//
// Bound: MFMA (dense fp16 2.5 PF, 3 MFMAs per product).  Algorithmic flop = 2*M*N*K per launch.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "conv_common.h"

#define PP_BM 256
#define PP_BN 128
#define PP_ROWS (PP_BM + PP_BN)
#define PP_STAGE (PP_ROWS * 32)                 /* dwords per ring stage */
#define PP_NSTAGE 3
#define PP_LDS_BYTES (PP_NSTAGE * PP_STAGE * 4) /* 147,456 B */
#define PP_EPI_LD (2 * 32 + 4)                  /* per-wave transpose buffer row (64 columns + pad) */

typedef __attribute__((address_space(3))) void* pp_lptr_t;

// Ring layout: rows are grouped in blocks of 8 (= the 1 KiB one wave-wide DMA instruction writes lane-linearly) and the three stages of a block
// are adjacent:  dword index of (row, stage, 16-byte piece) = ((row / 8) * 3 + stage) * 256 + (row % 8) * 32 + piece' * 4, piece' = piece ^
// ((row >> 1) & 7) as everywhere else.  Rows 0-255 = A (im2col), rows 256-383 = B (weights).  A lane's fragment reads of every stage / block
// row / k-step then differ from 8 per-lane base addresses only by IMMEDIATE offsets (stage: 1 KiB, next 32 rows: 12 KiB) — with stage-major
// storage the 144 KiB ring exceeds the 64 KiB immediate range and costs ~24 address registers.  Bank mapping is unchanged (blocks are 3 KiB
// = 12 x 256 B apart).
__device__ __forceinline__ int pp_lds(int row, int stage, int piece) {
    return (((row >> 3) * PP_NSTAGE + stage) << 8) + ((row & 7) << 5) + ((piece ^ ((row >> 1) & 7)) << 2);
}

// NL: how many of a wave's 6 DMA instructions per chunk are issued in the load phase; the other 6 - NL are issued in the matrix phase, one after
// every fourth MFMA (an LDS-DMA costs its wave ~60 issue cycles among MFMAs but 100-185 in a phase that also carries 16 ds_read_b128:
// MI355X_MICROARCH.md price list), so that neither phase of the ping-pong is the long pole.
template <int NL>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pp_smem[];
    float* smem = pp_smem;
    const SmirkConvDesc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;                       // 4 (M) x 2 (N) waves of 64 x 64; waves 0-3 = rows 0-127, waves 4-7 = rows 128-255
    const int group = swave >> 2;                                    // 0: leads, 1: runs one phase behind (shares each SIMD with a wave of group 0)
    const int ntn = a.N / PP_BN;
    const int logical = xcd_logical(blockIdx.x, gridDim.x);
    const int m0 = (logical / ntn) * PP_BM, n0 = (logical % ntn) * PP_BN;

    // ---- staging coordinates: thread (srow, pos) fills ring rows srow + 64 p with physical 16-byte piece `pos` ------------------------------
    const int pos = tid & 7, srow = tid >> 3;
    const int col4 = pos ^ ((srow >> 1) & 7);                      // logical piece fetched (source-side swizzle; (srow + 64p) >> 1 has the same low bits)
    const int HoWo = d.Ho * d.Wo;
    // per staged row: pixel index of the CENTRE tap (always inside the image) and a flag word — bit 0 / 1: the ky = 0 / ky = 2 row lies on the
    // other side of the centre (reflection at the top / bottom border), bit 2 / 3: same for kx = 0 / kx = 2, bits 4-6: row ky is readable,
    // bits 7-9: column kx is readable (zero padding: outside rows / columns are not; reflect padding: everything is)
    int pixc[4];
    unsigned flags[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int m = min(m0 + srow + 64 * p, a.M - 1);           // ragged last tile: clamped rows compute garbage that is never stored
        int b, oy, ox;
        row_to_pixel(m, HoWo, d.Wo, a.psh, b, oy, ox);
        pixc[p] = (b * d.H + oy) * d.W + ox;                       // pad 1, stride 1: the centre tap reads input pixel (oy, ox)
        const bool top = oy == 0, bot = oy == d.H - 1, lef = ox == 0, rig = ox == d.W - 1;
        unsigned f;
        if (d.pad_mode == SMIRK_PAD_REFLECT) f = (unsigned)top | ((unsigned)bot << 1) | ((unsigned)lef << 2) | ((unsigned)rig << 3) | (0x3Fu << 4);
        else f = ((unsigned)!top << 4) | (1u << 5) | ((unsigned)!bot << 6) | ((unsigned)!lef << 7) | (1u << 8) | ((unsigned)!rig << 9);
        flags[p] = f;
    }
    unsigned voffB[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) voffB[p] = ((unsigned)min(n0 + srow + 64 * p, a.N - 1) * (unsigned)a.K + (unsigned)col4 * 4u) * 4u;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.in0, (short)0, (int)((long long)d.B * d.H * d.W * d.C0 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(d.C1 > 0 ? a.in1 : a.in0), (short)0,
                                                                         (int)((long long)d.B * d.H * d.W * (d.C1 > 0 ? d.C1 : d.C0) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, (int)((long long)a.N * a.K * 4), 0x00020000);
    const int sh0 = 31 - __builtin_clz((unsigned)d.C0) + 2, sh1 = d.C1 > 0 ? 31 - __builtin_clz((unsigned)d.C1) + 2 : sh0;
    const int ncc = (d.C0 + d.C1) / CV_BK;
    const int twoW = 2 * d.W;
    const unsigned col16 = (unsigned)col4 * 16u;

    // DMA of chunk (cc, TAP) into ring stage ST: 4 A pieces + 2 B pieces per wave (1 KiB each).  `prep` computes the four per-lane A offsets (VALU,
    // load phase), `piece` issues DMA instruction k (0-3: A rows 8 swave + 64 k ..., 4-5: B) — so the issue points can be spread over both phases.
    unsigned vo[4];
    int dma_sa = 0, dma_sw = 0, dma_s1 = 0;
    auto prep = [&](int cc, auto tapc) {
        constexpr int TAP = decltype(tapc)::value, KY = TAP / 3, KX = TAP % 3;
        const int c0 = cc * CV_BK;
        const bool s1 = c0 >= d.C0;
        const int sh = s1 ? sh1 : sh0;
        dma_s1 = s1; dma_sa = (s1 ? c0 - d.C0 : c0) * 4; dma_sw = (TAP * a.Cin + c0) * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int px = pixc[p];
            unsigned f = flags[p];
            asm volatile("" : "+v"(px), "+v"(f));                  // nothing derived from the per-row state may be hoisted out of the K loop:
                                                                   // 36 (row, tap) offsets in registers spill; ~12 VALU per row here ride in the
                                                                   // load phase, under the partner wave's MFMAs
            if constexpr (KY == 0) px += (int)(f & 1u) * twoW - d.W;           // -W, or +W when reflected at the top border
            if constexpr (KY == 2) px += d.W - (int)((f >> 1) & 1u) * twoW;
            if constexpr (KX == 0) px += (int)((f >> 2) & 1u) * 2 - 1;
            if constexpr (KX == 2) px += 1 - (int)((f >> 3) & 1u) * 2;
            const unsigned inval = (((f >> (4 + KY)) & (f >> (7 + KX))) & 1u) ^ 1u;    // 1: this tap of this row lies in the zero padding
            vo[p] = (((unsigned)px << sh) + col16) | (inval << 31);                    // bit 31 set => beyond num_records => the DMA writes zeros
#ifdef SMIRK_DEBUG_HOOKS
            if (a.ablate == 1) vo[p] = (unsigned)(tid & 63) * 16u;                    // timing experiment only: every A piece re-reads one L1-resident KiB
#endif
        }
    };
    auto piece = [&](auto kc, auto stc) {
        constexpr int KP = decltype(kc)::value, ST = decltype(stc)::value;
        if constexpr (KP < 4) {
#ifdef SMIRK_DEBUG_HOOKS
            if (a.ablate == 2) return;                                               // timing experiment only: no A traffic at all
#endif
            float* dst = smem + (((swave + 8 * KP) * PP_NSTAGE + ST) << 8);          // block swave + 8 KP (rows 8 swave + 64 KP ...), stage ST
            if (dma_s1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (pp_lptr_t)dst, 16, vo[KP], dma_sa, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (pp_lptr_t)dst, 16, vo[KP], dma_sa, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (pp_lptr_t)(smem + (((PP_BM / 8 + swave + 8 * (KP - 4)) * PP_NSTAGE + ST) << 8)), 16,
                                                     voffB[KP - 4], dma_sw, 0, 0);
        }
    };
    auto pieces = [&](auto loc, auto hic, auto stc) {             // DMA instructions [LO, HI)
        constexpr int LO = decltype(loc)::value, HI = decltype(hic)::value;
        if constexpr (LO <= 0 && 0 < HI) piece(std::integral_constant<int, 0>{}, stc);
        if constexpr (LO <= 1 && 1 < HI) piece(std::integral_constant<int, 1>{}, stc);
        if constexpr (LO <= 2 && 2 < HI) piece(std::integral_constant<int, 2>{}, stc);
        if constexpr (LO <= 3 && 3 < HI) piece(std::integral_constant<int, 3>{}, stc);
        if constexpr (LO <= 4 && 4 < HI) piece(std::integral_constant<int, 4>{}, stc);
        if constexpr (LO <= 5 && 5 < HI) piece(std::integral_constant<int, 5>{}, stc);
    };
    auto issue = [&](int cc, auto tapc, auto stc) {
        prep(cc, tapc);
        pieces(std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{}, stc);
    };

    f32x16 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    const int fr = lane & 31, hb = lane >> 5;
    half8 ah[2][2], al[2][2], bh[2][2], bl[2][2];                   // [k-step][block]: the 16 operand fragments of one chunk (64 VGPRs)

    // one (load phase, matrix phase) pair for chunk (cc, TAP); the chunk sits in ring stage TAP % 3 (9 taps per channel chunk, 9 % 3 == 0)
    auto body = [&](int cc, auto tapc) {
        constexpr int TAP = decltype(tapc)::value, ST = TAP % 3;
        // ---- load phase: fragments of this chunk, DMA of the chunk two ahead --------------------------------------------------------------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pc = 2 * (2 * s + hb);                       // logical piece of this lane's hi halves (lo = pc + 1)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wm * 2 + i) * 32 + fr;
                ah[s][i] = *(const half8*)(smem + pp_lds(row, ST, pc)); al[s][i] = *(const half8*)(smem + pp_lds(row, ST, pc + 1));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = PP_BM + (wn * 2 + j) * 32 + fr;
                bh[s][j] = *(const half8*)(smem + pp_lds(row, ST, pc)); bl[s][j] = *(const half8*)(smem + pp_lds(row, ST, pc + 1));
            }
        }
        // the chunk two ahead; past the end of K the (clamped) last channel chunk is fetched again into a stage nobody reads any more — two
        // wasted chunk loads per tile instead of a branch (and its duplicated code) in seven of nine loop bodies
        constexpr int T2 = (TAP + 2) % 9, S2 = T2 % 3;
        prep(min(cc + (TAP + 2 >= 9 ? 1 : 0), ncc - 1), std::integral_constant<int, T2>{});
        pieces(std::integral_constant<int, 0>{}, std::integral_constant<int, NL>{}, std::integral_constant<int, S2>{});
        // everything issued before these NL instructions has landed: the whole next chunk (this wave's share); fragment reads done
        if constexpr (NL == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else if constexpr (NL == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if constexpr (NL == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else if constexpr (NL == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- matrix phase: 24 MFMAs, operands already in registers; dependent accumulations are four instructions apart ----------------------
        __builtin_amdgcn_s_setprio(1);
        constexpr int NM = 6 - NL;                                     // DMA instructions left for this phase: one after every fourth MFMA
        auto mm_dma = [&](auto gc) {
            constexpr int G = decltype(gc)::value;
            if constexpr (G < NM) {
                __builtin_amdgcn_sched_barrier(0);
                piece(std::integral_constant<int, NL + G>{}, std::integral_constant<int, S2>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][i], bh[0][j], acc0[i][j], 0, 0, 0);
        mm_dma(std::integral_constant<int, 0>{});
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][i], bl[0][j], acc1[i][j], 0, 0, 0);
        mm_dma(std::integral_constant<int, 1>{});
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0][i], bh[0][j], acc1[i][j], 0, 0, 0);
        mm_dma(std::integral_constant<int, 2>{});
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][i], bh[1][j], acc0[i][j], 0, 0, 0);
        mm_dma(std::integral_constant<int, 3>{});
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][i], bl[1][j], acc1[i][j], 0, 0, 0);
        mm_dma(std::integral_constant<int, 4>{});
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[1][i], bh[1][j], acc1[i][j], 0, 0, 0);
        mm_dma(std::integral_constant<int, 5>{});
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: chunks 0 and 1 in flight, chunk 0 landed; group 1 drops one phase behind ------------------------------------------------
    issue(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    issue(0, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (group == 1) __builtin_amdgcn_s_barrier();
    for (int cc = 0; cc < ncc; ++cc) {
        body(cc, std::integral_constant<int, 0>{}); body(cc, std::integral_constant<int, 1>{}); body(cc, std::integral_constant<int, 2>{});
        body(cc, std::integral_constant<int, 3>{}); body(cc, std::integral_constant<int, 4>{}); body(cc, std::integral_constant<int, 5>{});
        body(cc, std::integral_constant<int, 6>{}); body(cc, std::integral_constant<int, 7>{}); body(cc, std::integral_constant<int, 8>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the two over-fetched chunks must have landed before the ring is reused
    if (group == 0) __builtin_amdgcn_s_barrier();                   // both groups have executed the same number of barriers ...
    __builtin_amdgcn_s_barrier();                                   // ... and every wave's DMA has retired: the ring is free

    // ---- epilogue: per-wave transpose through LDS, whole 8-channel groups, BN scale/shift + residual + ReLU, re-split --------------------------
    float* ebuf = smem + wave * 32 * PP_EPI_LD;
    constexpr int GPR = 8, ITEMS = 32 * GPR / 64;                   // 8-channel groups per buffer row; items per lane
    SmirkRangeAcc rng;                                              // split-fp16 range audit (common.h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ebuf[mfma32_row(r, lane) * PP_EPI_LD + j * 32 + fr] = acc0[i][j][r] + acc1[i][j][r] * (1.0f / 2048.0f);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int item = it * 64 + lane, row = item / GPR, g = item % GPR;
            const int m = m0 + (wm * 2 + i) * 32 + row, n = n0 + wn * 64 + g * 8;
            if (m < a.M && n < a.N) {
                float v[8];
                *(f32x4*)v = *(const f32x4*)(ebuf + row * PP_EPI_LD + g * 8);
                *(f32x4*)(v + 4) = *(const f32x4*)(ebuf + row * PP_EPI_LD + g * 8 + 4);
                int b, y, x;
                row_to_pixel(m, HoWo, d.Wo, a.psh, b, y, x);
                const size_t o = (((size_t)b * d.Ho + y) * d.Wo + x) * d.Cout + n;
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
                split8(v, hi, lo, rng);
                *(half8*)(a.out + o) = hi;
                *(half8*)(a.out + o + 4) = lo;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    rng.commit();
}

// Serves: split-fp16, 3x3, stride 1, NHWC out, both sources multiples of 32 channels and powers of two, N a multiple of 128, operands < 2 GiB.
bool smirk_conv_pp_eligible(const ConvArgs& a) {
    const SmirkConvDesc& d = a.d;
    const char* env = getenv("SMIRK_IGEMM_PP");                      // "0" keeps these layers on conv_igemm_kernel (A/B switch; read per call: tests toggle it)
    if (env && env[0] == '0') return false;
    if (d.KH != 3 || d.KW != 3 || d.stride != 1 || d.out_mode != SMIRK_OUT_NHWC || d.pad_t != 1 || d.pad_l != 1) return false;
    if (d.Ho != d.H || d.Wo != d.W || d.W > 1023) return false;
    if (d.C0 % CV_BK || d.C1 % CV_BK || (d.C0 & (d.C0 - 1)) || (d.C1 & (d.C1 - 1))) return false;
    if (a.N % PP_BN || a.N < PP_BN) return false;
    const long long b0 = (long long)d.B * d.H * d.W * d.C0 * 4, b1 = (long long)d.B * d.H * d.W * d.C1 * 4, bw = (long long)a.N * a.K * 4;
    if (b0 >= (1ll << 31) || b1 >= (1ll << 31) || bw >= (1ll << 31)) return false;
    // Measured per layer at B = 128 (tools/conv_sweep.py --ab-pp, profiles/r02_conv_sweep_pp.txt): 14x14 layers 0.320-0.325 -> 0.277-0.279 ms
    // (K = 2304 / 4608: 72 / 144 chunks per tile amortise the exposed prologue + epilogue of the single workgroup per CU), 28x28 and 56x56
    // layers 1-9 % SLOWER (36-72 chunks per tile; the 128x128 kernel hides one workgroup's epilogue behind its co-resident twin's main
    // loop).  SMIRK_IGEMM_PP=all lifts the restriction for experiments.
    const bool all = env && env[0] == 'a';
    if (!all && (long long)d.Ho * d.Wo > 256) return false;
    return a.M >= 4 * PP_BM;                                          // tiny problems stay on the 128-row tiles
}

int smirk_conv_pp_launch(const ConvArgs& a_in, hipStream_t st) {
    static bool attr_done_dev[64] = {};                              // hipFuncSetAttribute is per-device state: one flag per device ordinal
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return SMIRK_ERR_UNSUPPORTED;
    bool& attr_done = attr_done_dev[dev];
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)conv_pp_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES) != hipSuccess)
            return SMIRK_ERR_LAUNCH;
        attr_done = true;
    }
    ConvArgs a = a_in;
#ifdef SMIRK_DEBUG_HOOKS                                                 /* -DSMIRK_DEBUG_HOOKS variant builds only (tools/build_variant.sh) */
    if (const char* ab = getenv("SMIRK_PP_ABLATE")) a.ablate = atoi(ab);        // WRONG RESULTS: operand-traffic timing experiments
#endif
    // NL = 3 of a wave's 6 LDS-DMA instructions per chunk are issued in its load phase, 3 among the MFMAs.  0 / 2 / 6 measured the same within 1 % (0.279-0.281 ms on
    // the 14 x 14 layer, profiles/r02_conv_pp_dma_split.txt); the $SMIRK_PP_NL switch and those instantiations left the library in round 5.
    const int ntm = (a.M + PP_BM - 1) / PP_BM, ntn = a.N / PP_BN;
    if (g_smirk_prof_on) {
        const double px = (double)a.d.B * a.d.H * a.d.W;
        smirk_prof_next("conv_pp_kernel<3>[256x128,8w,3stage]", 2.0 * a.M * a.N * a.K,
                        4.0 * (px * a.Cin + (double)a.M * a.N + (double)a.N * a.K + (a.residual ? (double)a.M * a.N : 0.0)));
    }
    SMIRK_LAUNCH(conv_pp_kernel<3>, dim3(ntm * ntn), dim3(512), PP_LDS_BYTES, st, a);
    return smirk_launch_status();
}
