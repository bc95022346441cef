// Implicit-GEMM convolution for MI355X (gfx950) on the matrix cores, two arithmetic modes sharing one loader:
//
//   F32    v_mfma_f32_32x32x2_f32 — an exact, k-ordered fp32 FMA chain at the 157 TFLOP/s f32 matrix rate.
//   F16X3  "split-fp16": CDNA4 has no TF32, so fp32-class accuracy is obtained on the 2.5 PFLOP/s 16-bit pipe by carrying
//          every value as an fp16 pair  x = hi + lo * 2^-11  (hi = fp16(x), lo = fp16((x - hi) * 2^11); 22 significand bits)
//          and issuing three v_mfma_f32_32x32x16_f16 per product block:  acc0 += a_hi.w_hi ;  acc1 += a_hi.w_lo + a_lo.w_hi ;
//          result = acc0 + acc1 * 2^-11  (fp16 x fp16 products are exact in fp32; only the 2^-22 lo.lo term is dropped).
//          Measured error vs fp64 equals that of an fp32 GEMM.  Activations live in HBM already split:
//          [B][H][W][C/8][2][8] halves = 4 bytes per element, the same footprint and the same 16-byte vector geometry as fp32
//          NHWC, so the loader is shared verbatim; the epilogue re-splits its outputs.
//
// GEMM view:  out[m][n] = sum_k A[m][k] * Wt[n][k]
//     m = (b, oy, ox) output pixel (NHWC), n = output channel, k = (ky, kx, c) — c contiguous, matching NHWC activations, so an
//     im2col row segment is a contiguous run of channels and every global access is a 16-byte vector.
// * A is gathered on the fly (zero or REFLECT padding = index arithmetic, no padded copy; a channel-concatenated input is two
//   source tensors, so torch.cat((up, skip), 1) of the U-Net decoder is never materialised).
// * Block tile BM x BN x 32, 4 waves; both operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers, no
//   ds_write) into a 2-stage ring: K-contiguous 128-byte rows with an XOR piece swizzle (source side + read side) so the 16-lane
//   groups of ds_read_b128 hit 16 distinct 16-byte slots; ONE barrier per K chunk, the next chunk's DMA runs under the MFMAs.
//   F32: one ds_read_b128 per lane feeds FOUR MFMAs — lanes 0-31 take k = 4t..4t+3 and lanes 32-63 k = 4t+4..4t+7 of each 8-k
//   group (the same permutation on A and B leaves the sum unchanged).  F16X3: a lane's 16-byte read is the 8 hi (or 8 lo)
//   halves of one 8-channel group = one MFMA operand.
// * Epilogue fuses BatchNorm(eval) scale/shift, residual add, ReLU, and (for ConvTranspose2d k2 s2) the 2x2 pixel scatter; the
//   F16X3 epilogue transposes each wave's tile through LDS so every lane stores whole 8-channel groups (2 x 16 bytes).
// * blockIdx -> tile mapping is XCD-aware: the 8 XCDs each get a contiguous run of tiles, so blocks that share an A row
//   panel or a weight panel hit the same private L2.
// * K is walked in one of several orders / addressing schemes (enum KW_* below).  The shipped defaults for channel-aligned layers are
//   the buffer-addressed walks: operands arrive through `buffer_load_dwordx4 ... offen lds` with a per-lane 32-bit offset that is
//   constant per (tap, source) segment, a scalar channel offset, an out-of-range offset for rows in the zero padding (the buffer unit
//   writes zeros), scalar LDS destinations and a loop unrolled by two — no vector ALU work between the barrier and the first fragment
//   read.  3x3 layers on the 128x128 tile walk K channel-chunk-major (the 9 taps of a 32-channel chunk back to back) so that the
//   shifted re-reads of the same input pixels hit L2: measured with tools/igemm_micro.py, the hot loop runs a chunk in 1300 cycles on
//   L2-resident operands but 2850 when they come back from the Infinity Cache.
//
// Bound: MFMA (fp32 157.3 TF; f16 2.5 PF issuing 3 MFMA-flop per algorithmic flop) once the operands hit L2; the Infinity Cache's
// ~7 TB/s otherwise.  Algorithmic flop = 2*M*N*K.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include <type_traits>

#include "conv_common.h"

// 16 zero bytes in global memory: the source of a direct-to-LDS load whose im2col element is padding / out of range
__device__ __attribute__((aligned(16))) float g_zero16_conv[4] = {0.f, 0.f, 0.f, 0.f};
#define g_zero16 g_zero16_conv

// branch-free pointer select (hipcc turns `c ? p : q` feeding a DMA into exec-masked branches around each load)
__device__ __forceinline__ const float* psel(bool c, const float* p, const float* q) {
    const unsigned long long m = 0ull - (unsigned long long)c;
    return (const float*)(((unsigned long long)p & m) | ((unsigned long long)q & ~m));
}

// K-walk modes of conv_tile
//   KW_GENERIC  any geometry: tap / channel / source recomputed per chunk with divides (3x3 with C % 32 != 0, e.g. an 8-channel input)
//   KW_FAST     every source has C % 32 == 0: a chunk never straddles a tap or a concat source; im2col pointers are set up once per
//               (tap, source) SEGMENT and advance by 128 bytes per chunk
//   KW_FAST_KT  1x1 convolutions with any C % 4 == 0 (the encoders' pointwise layers): one segment per source whose last chunk is
//               partial — lanes whose 16-byte piece lies beyond the segment read the zero page
//   KW_CMAJOR   3x3 with every source C % 32 == 0, walked channel-chunk-major with the 9 taps of a chunk back to back: the taps re-read
//               (shifted) the same input pixels, so in this order the re-reads are L2 hits; tap-major order puts a whole pass over the
//               channels (> the 4 MB L2 of an XCD) between them and every tap's rows come back from the Infinity Cache, whose ~7 TB/s
//               (tools/igemm_micro.py) — not the matrix pipe — then sets the pace
//   KW_LEAN     KW_FAST's walk with buffer-addressed DMA (`buffer_load_dwordx4 ... offen lds`): the per-lane part of every operand address is a
//               32-bit VGPR offset that is constant for a whole (tap, source) segment, the advancing channel offset is a scalar, rows in the
//               zero padding carry an out-of-range offset (the buffer unit returns zeros: no select, no zero page), the LDS destination
//               (M0) is pure scalar arithmetic, and the loop is unrolled by two so the stage is a compile-time constant.  Cuts the ~54 VALU
//               + 20 SALU per chunk and wave that KW_FAST spends on 64-bit pointers.
//   KW_LEAN_CM  KW_CMAJOR's channel-major walk with KW_LEAN's buffer-addressed DMA (channel counts must be powers of two: the pixel index
//               becomes a byte offset with one shift-add per row and chunk)
enum { KW_GENERIC = 0, KW_FAST = 1, KW_FAST_KT = 2, KW_CMAJOR = 3, KW_LEAN = 4, KW_LEAN_CM = 5 };

// one BM x BN output tile at (m0, n0); `smem` holds 2 pipeline stages of (BM + BN) x 32 dwords
template <int BM, int BN, int WGM, int WGN, bool SPLIT, int KWALK, bool X1 = false, bool STATS = false>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, float* smem, const int m0, const int n0) {
    constexpr int NW = WGM * WGN, NT = 64 * NW;                 // waves / threads per workgroup (4 or 8 waves)
    constexpr int RP = NT / 8;                                  // operand rows filled per DMA pass (8 lanes x 16 B per 128-byte row)
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int PA = BM / RP, PB = BN / RP;
    constexpr int STAGE = (BM + BN) * 32;                       // dwords per pipeline stage
    constexpr int EPI_LD = TN * 32 + 4;                         // per-wave transpose buffer [32][EPI_LD] (vector epilogue)
    static_assert(NW * 32 * EPI_LD <= 2 * STAGE, "epilogue buffer must fit in the operand LDS");
    static_assert(BM % RP == 0 && BN % RP == 0, "tile must be a whole number of DMA passes");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const SmirkConvDesc& d = a.d;

    // ---- per-thread staging coordinates: thread (srow, pos) fills LDS row srow + RP*p, physical piece pos -------------------------
    const int pos = tid & 7, srow = tid >> 3;
    const int col4 = pos ^ ((srow >> 1) & 7);                   // logical 16-byte piece this lane fetches (source-side swizzle)
    int pyx[PA], boff[PA];                                      // (iy0 << 16 | ix0 & 0xffff) and image offset of each staged row
    const int HoWo = d.Ho * d.Wo;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        // rows beyond M (ragged last tile) are clamped to the last valid row: they compute garbage that the epilogue never stores,
        // and no per-load predicate is needed for them
        const int m = min(m0 + srow + RP * p, a.M - 1);
        int b, oy, ox;
        row_to_pixel(m, HoWo, d.Wo, a.psh, b, oy, ox);
        pyx[p] = ((oy * d.stride - d.pad_t) << 16) | ((ox * d.stride - d.pad_l) & 0xffff);
        boff[p] = b * d.H * d.W;
    }
    unsigned wrow[PB];                                          // dword offset of weight row min(n0+srow+RP*p, N-1) (columns beyond N: clamped, never stored)
#pragma unroll
    for (int p = 0; p < PB; ++p) wrow[p] = (unsigned)min(n0 + srow + RP * p, a.N - 1) * (unsigned)a.K;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    // ---- KW_GENERIC: issue the direct-to-LDS loads of K chunk k0 into pipeline stage `st` ------------------------------------------
    auto issue_generic = [&](int k0, int st) {
        float* As = smem + st * STAGE;
        float* Bs = As + BM * 32;
        const int k = k0 + col4 * 4;
        const bool kval = k < a.K;
        int tap = 0, c = 0;
        if (kval) { tap = k / a.Cin; c = k - tap * a.Cin; }
        const int ky = tap / d.KW, kx = tap - ky * d.KW;
        const bool src1 = c >= d.C0;
        const float* src = src1 ? a.in1 : a.in0;
        const int cs = src1 ? d.C1 : d.C0, cc = src1 ? c - d.C0 : c;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            int iy = (pyx[p] >> 16) + ky, ix = (int)(short)(pyx[p] & 0xffff) + kx;
            bool ok = kval;
            if (d.pad_mode == SMIRK_PAD_REFLECT) { iy = reflect_idx(iy, d.H); ix = reflect_idx(ix, d.W); }
            else ok = ok && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
            iy = min(max(iy, 0), d.H - 1); ix = min(max(ix, 0), d.W - 1);
            const float* g = psel(ok, src + ((size_t)(boff[p] + iy * d.W + ix) * cs + cc), g_zero16);
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(As + (wave * 8 + RP * p) * 32), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const float* g = psel(kval, a.w + wrow[p] + k, g_zero16);
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Bs + (wave * 8 + RP * p) * 32), 16, 0, 0);
        }
    };

    // ---- KW_CMAJOR: per staged row, the (reflected / clamped) pixel-row and column offsets of the three ky / kx, and a 9-bit tap validity mask
    int rowoff[PA][3], coloff[PA][3];
    unsigned vmask[PA];
    if constexpr (KWALK == KW_CMAJOR || KWALK == KW_LEAN_CM) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int iy0 = pyx[p] >> 16, ix0 = (int)(short)(pyx[p] & 0xffff);
            unsigned oky = 0, okx = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int iy = iy0 + k, ix = ix0 + k;
                if (d.pad_mode == SMIRK_PAD_REFLECT) { iy = reflect_idx(iy, d.H); ix = reflect_idx(ix, d.W); oky |= 1u << k; okx |= 1u << k; }
                else { oky |= (unsigned)(iy >= 0 && iy < d.H) << k; okx |= (unsigned)(ix >= 0 && ix < d.W) << k; }
                rowoff[p][k] = boff[p] + min(max(iy, 0), d.H - 1) * d.W;
                coloff[p][k] = min(max(ix, 0), d.W - 1);
            }
            vmask[p] = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) vmask[p] |= (((oky >> (t / 3)) & (okx >> (t % 3))) & 1u) << t;
        }
    }
    auto issue_cmajor = [&](int cc, int tap, int st) {           // tap is a compile-time constant at every call site
        float* As = smem + st * STAGE;
        float* Bs = As + BM * 32;
        const int c0 = cc * CV_BK;
        const bool s1 = c0 >= d.C0;
        const float* src = s1 ? a.in1 : a.in0;
        const int cs = s1 ? d.C1 : d.C0, cb = (s1 ? c0 - d.C0 : c0) + col4 * 4;
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const float* g = src + ((size_t)(rowoff[p][ky] + coloff[p][kx]) * cs + cb);
            g = psel((vmask[p] >> tap) & 1u, g, g_zero16);
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(As + (wave * 8 + RP * p) * 32), 16, 0, 0);
        }
        const float* wb = a.w + (tap * a.Cin + c0 + col4 * 4);
#pragma unroll
        for (int p = 0; p < PB; ++p)
            __builtin_amdgcn_global_load_lds((gptr_t)(wb + wrow[p]), (lptr_t)(Bs + (wave * 8 + RP * p) * 32), 16, 0, 0);
    };

#if defined(__HIP_DEVICE_COMPILE__)      // the buffer-resource type and builtins exist in the device pass only
    // ---- KW_LEAN: buffer resources + per-lane byte offsets -------------------------------------------------------------------------
    [[maybe_unused]] const int swave = __builtin_amdgcn_readfirstlane(wave);
    [[maybe_unused]] unsigned voffA[PA], voffB[PB];
    [[maybe_unused]] int sA = 0, sW = 0, lseg_left = 0, lseg_tap = 0, lseg_src = 0;
    auto lean_rsrc = [&](const float* base, long long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)bytes, 0x00020000);
    };
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs0 = lean_rsrc(a.in0, (long long)d.B * d.H * d.W * d.C0 * 4);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs1 = lean_rsrc(d.C1 > 0 ? a.in1 : a.in0, (long long)d.B * d.H * d.W * (d.C1 > 0 ? d.C1 : d.C0) * 4);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsw = lean_rsrc(a.w, (long long)a.N * a.K * 4);
    if constexpr (KWALK == KW_LEAN || KWALK == KW_LEAN_CM) {
#pragma unroll
        for (int p = 0; p < PB; ++p) voffB[p] = (wrow[p] + (unsigned)col4 * 4u) * 4u;
    }
    auto open_segment_lean = [&]() {
        const int ky = lseg_tap / d.KW, kx = lseg_tap - ky * d.KW;
        const int cs = lseg_src ? d.C1 : d.C0;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            int iy = (pyx[p] >> 16) + ky, ix = (int)(short)(pyx[p] & 0xffff) + kx;
            bool ok = true;
            if (d.pad_mode == SMIRK_PAD_REFLECT) { iy = reflect_idx(iy, d.H); ix = reflect_idx(ix, d.W); }
            else ok = iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
            iy = min(max(iy, 0), d.H - 1); ix = min(max(ix, 0), d.W - 1);
            const unsigned off = ((unsigned)(boff[p] + iy * d.W + ix) * (unsigned)cs + (unsigned)col4 * 4u) * 4u;
            voffA[p] = ok ? off : 0x80000000u;                   // beyond num_records: the load returns zeros
        }
        sA = 0;
        sW = (lseg_tap * a.Cin + (lseg_src ? d.C0 : 0)) * 4;
        lseg_left = cs / CV_BK;
    };
    auto issue_lean = [&](auto stc, bool src1) {
        constexpr int ST = decltype(stc)::value;
        float* As = smem + ST * STAGE;
        float* Bs = As + BM * 32;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            if (src1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr_t)(As + (swave * 8 + RP * p) * 32), 16, voffA[p], sA, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lptr_t)(As + (swave * 8 + RP * p) * 32), 16, voffA[p], sA, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p)
        {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr_t)(Bs + (swave * 8 + RP * p) * 32), 16, voffB[p], sW, 0, 0);
        }
        sA += CV_BK * 4; sW += CV_BK * 4;
        --lseg_left;
    };
    // channel-major: chunk (cc, tap) with compile-time tap and stage
    [[maybe_unused]] const int sh0 = 31 - __builtin_clz((unsigned)d.C0) + 2, sh1 = d.C1 > 0 ? 31 - __builtin_clz((unsigned)d.C1) + 2 : sh0;
    auto issue_lean_cm = [&](int cc, auto tapc, auto stc) {
        constexpr int TAP = decltype(tapc)::value, ST = decltype(stc)::value, KY = TAP / 3, KX = TAP % 3;
        float* As = smem + ST * STAGE;
        float* Bs = As + BM * 32;
        const int c0 = cc * CV_BK;
        const bool s1 = c0 >= d.C0;
        const int sh = s1 ? sh1 : sh0, sa = (s1 ? c0 - d.C0 : c0) * 4, sw = (TAP * a.Cin + c0) * 4;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const unsigned off = ((unsigned)(rowoff[p][KY] + coloff[p][KX]) << sh) + (unsigned)col4 * 16u;
            const unsigned vo = ((vmask[p] >> TAP) & 1u) ? off : 0x80000000u;
            if (s1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr_t)(As + (swave * 8 + RP * p) * 32), 16, vo, sa, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lptr_t)(As + (swave * 8 + RP * p) * 32), 16, vo, sa, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr_t)(Bs + (swave * 8 + RP * p) * 32), 16, voffB[p], sw, 0, 0);
        }
    };
    // a segment ends: next (tap, source)
    auto lean_advance = [&]() {
        if (d.C1 > 0 && lseg_src == 0) lseg_src = 1; else { lseg_src = 0; ++lseg_tap; }
        open_segment_lean();
    };

#endif

    // ---- KW_FAST*: segment state ------------------------------------------------------------------------------------------------------
    const float* pa[PA];
    int inca[PA];                                               // 32 dwords, or 0 when the row reads the zero page
    const float* pbase = a.w;                                   // a.w + k offset of the next chunk to issue (this lane's piece)
    int seg_left = 0, seg_tap = 0, seg_src = 0;                 // chunks left to ISSUE in the open segment; next segment to open
    int kleft = 0;                                              // KW_FAST_KT: dwords of the segment still ahead of this lane's piece
    auto open_segment = [&]() {
        const int ky = seg_tap / d.KW, kx = seg_tap - ky * d.KW;
        const float* src = seg_src ? a.in1 : a.in0;
        const int cs = seg_src ? d.C1 : d.C0;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            int iy = (pyx[p] >> 16) + ky, ix = (int)(short)(pyx[p] & 0xffff) + kx;
            bool ok = true;
            if (d.pad_mode == SMIRK_PAD_REFLECT) { iy = reflect_idx(iy, d.H); ix = reflect_idx(ix, d.W); }
            else ok = iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
            iy = min(max(iy, 0), d.H - 1); ix = min(max(ix, 0), d.W - 1);
            pa[p] = psel(ok, src + ((size_t)(boff[p] + iy * d.W + ix) * cs + col4 * 4), g_zero16);
            inca[p] = ok ? CV_BK : 0;
        }
        pbase = a.w + (seg_tap * a.Cin + (seg_src ? d.C0 : 0) + col4 * 4);
        seg_left = (cs + CV_BK - 1) / CV_BK;
        kleft = cs - col4 * 4;
        if (d.C1 > 0 && seg_src == 0) seg_src = 1; else { seg_src = 0; ++seg_tap; }
    };
    // straight-line (branch-free) issue of the next chunk: 1 KiB DMA per wave instruction + pointer bumps; meant to sit in the same basic
    // block as the MFMAs of the current chunk so the scheduler can hide it in their shadow
    auto issue_fast = [&](int st) {
        float* As = smem + st * STAGE;
        float* Bs = As + BM * 32;
        const bool kv = (KWALK == KW_FAST_KT) ? (kleft > 0) : true;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const float* g = (KWALK == KW_FAST_KT) ? psel(kv, pa[p], g_zero16) : pa[p];
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(As + (wave * 8 + RP * p) * 32), 16, 0, 0);
            pa[p] += inca[p];
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const float* g = (KWALK == KW_FAST_KT) ? psel(kv, pbase + wrow[p], g_zero16) : pbase + wrow[p];
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Bs + (wave * 8 + RP * p) * 32), 16, 0, 0);
        }
        pbase += CV_BK;
        kleft -= CV_BK;
        --seg_left;
    };

    constexpr int NACC = SPLIT ? 2 : 1;
    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;

    const int fr = lane & 31, hb = lane >> 5;
    // all LDS operand reads of a chunk (both 16-k steps) go out up front into two fragment sets, then the MFMAs
    auto compute = [&](const float* As, const float* Bs) {
        if constexpr (SPLIT) {
            if constexpr (X1) {                                   // F16X1 (its own instantiations: the three-MFMA kernels stay as they are): hi halves only
                half8 xh[2][TM], wh[2][TN];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int pc = 2 * (2 * s + hb);
#pragma unroll
                    for (int i = 0; i < TM; ++i) xh[s][i] = *(const half8*)(As + lds_piece((wm * TM + i) * 32 + fr, pc));
#pragma unroll
                    for (int j = 0; j < TN; ++j) wh[s][j] = *(const half8*)(Bs + lds_piece((wn * TN + j) * 32 + fr, pc));
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[s][i], wh[s][j], acc[0][i][j], 0, 0, 0);
            } else {
            half8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pc = 2 * (2 * s + hb);                  // logical piece of this lane's hi halves (lo = pc + 1)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = (wm * TM + i) * 32 + fr;
                    ah[s][i] = *(const half8*)(As + lds_piece(row, pc)); al[s][i] = *(const half8*)(As + lds_piece(row, pc + 1));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = (wn * TN + j) * 32 + fr;
                    bh[s][j] = *(const half8*)(Bs + lds_piece(row, pc)); bl[s][j] = *(const half8*)(Bs + lds_piece(row, pc + 1));
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bh[s][j], acc[0][i][j], 0, 0, 0);
                        acc[NACC - 1][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bl[s][j], acc[NACC - 1][i][j], 0, 0, 0);
                        acc[NACC - 1][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][i], bh[s][j], acc[NACC - 1][i][j], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < CV_BK / 8; ++kk) {
                f32x4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *(const f32x4*)(As + lds_piece((wm * TM + i) * 32 + fr, kk * 2 + hb));
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *(const f32x4*)(Bs + lds_piece((wn * TN + j) * 32 + fr, kk * 2 + hb));
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[0][i][j], 0, 0, 0);
            }
        }
    };
    auto chunk_ready = [&]() {
        // chunk data has landed once THIS wave's DMAs retire and every wave has passed the barrier; the barrier also means every wave
        // finished reading the other stage (used by the previous chunk), so it may be refilled right away, under the MFMAs
        // vmcnt: this wave's DMAs of the chunk have landed.  lgkmcnt: this wave's fragment reads of the PREVIOUS chunk have completed —
        // the compiler may sink their consumers (MFMAs) below the barrier, and the stage they read is refilled right after it
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                           // s_barrier is IntrNoMem: stop the compiler moving LDS reads above it
    };

    if constexpr (KWALK == KW_GENERIC) {
        const int nchunk = (a.K + CV_BK - 1) / CV_BK;
        issue_generic(0, 0);
        for (int ch = 0; ch < nchunk; ++ch) {
            chunk_ready();
            if (ch + 1 < nchunk) issue_generic((ch + 1) * CV_BK, (ch + 1) & 1);
            compute(smem + (ch & 1) * STAGE, smem + (ch & 1) * STAGE + BM * 32);
        }
    } else if constexpr (KWALK == KW_LEAN) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int nloop = d.KH * d.KW * (d.C0 / CV_BK + d.C1 / CV_BK);
        open_segment_lean();
        issue_lean(std::integral_constant<int, 0>{}, false);
        for (int ch = 0; ch + 1 < nloop; ch += 2) {
            if (lseg_left == 0) lean_advance();
            chunk_ready();
            issue_lean(std::integral_constant<int, 1>{}, lseg_src != 0);
            compute(smem, smem + BM * 32);
            const bool more = ch + 2 < nloop;
            if (more && lseg_left == 0) lean_advance();
            chunk_ready();
            if (more) issue_lean(std::integral_constant<int, 0>{}, lseg_src != 0);
            compute(smem + STAGE, smem + STAGE + BM * 32);
        }
        if (nloop & 1) {
            chunk_ready();
            compute(smem, smem + BM * 32);
        }
#endif
    } else if constexpr (KWALK == KW_LEAN_CM) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int ncc = (d.C0 + d.C1) / CV_BK;                    // even (dispatch guarantees it): two chunks x 9 taps = 18 static (tap, stage) bodies
        auto body = [&](int cc, auto tapc, auto parc) {           // chunk (cc, TAP) sits in stage (PAR*9 + TAP) & 1
            constexpr int TAP = decltype(tapc)::value, PAR = decltype(parc)::value, ST = (PAR * 9 + TAP) & 1;
            chunk_ready();
            if constexpr (TAP < 8) issue_lean_cm(cc, std::integral_constant<int, TAP + 1>{}, std::integral_constant<int, ST ^ 1>{});
            else if (cc + 1 < ncc) issue_lean_cm(cc + 1, std::integral_constant<int, 0>{}, std::integral_constant<int, ST ^ 1>{});
            compute(smem + ST * STAGE, smem + ST * STAGE + BM * 32);
        };
        auto nine = [&](int cc, auto parc) {
            body(cc, std::integral_constant<int, 0>{}, parc); body(cc, std::integral_constant<int, 1>{}, parc);
            body(cc, std::integral_constant<int, 2>{}, parc); body(cc, std::integral_constant<int, 3>{}, parc);
            body(cc, std::integral_constant<int, 4>{}, parc); body(cc, std::integral_constant<int, 5>{}, parc);
            body(cc, std::integral_constant<int, 6>{}, parc); body(cc, std::integral_constant<int, 7>{}, parc);
            body(cc, std::integral_constant<int, 8>{}, parc);
        };
        issue_lean_cm(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        for (int cc = 0; cc < ncc; cc += 2) {
            nine(cc, std::integral_constant<int, 0>{});
            nine(cc + 1, std::integral_constant<int, 1>{});
        }
#endif
    } else if constexpr (KWALK == KW_CMAJOR) {
        const int ncc = (d.C0 + d.C1) / CV_BK;
        issue_cmajor(0, 0, 0);
        int par = 0;
        for (int cc = 0; cc < ncc; ++cc) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                chunk_ready();
                const float* As = smem + par * STAGE;
                par ^= 1;
                if (tap < 8) issue_cmajor(cc, tap + 1, par);
                else if (cc + 1 < ncc) issue_cmajor(cc + 1, 0, par);
                compute(As, As + BM * 32);
            }
        }
    } else {
        const int nloop = d.KH * d.KW * ((d.C0 + CV_BK - 1) / CV_BK + (d.C1 + CV_BK - 1) / CV_BK);
        open_segment();
        issue_fast(0);
        for (int ch = 0; ch + 1 < nloop; ++ch) {                 // every iteration issues the NEXT chunk: no guard inside the hot block
            if (seg_left == 0) open_segment();                   // rare (once per tap / source), before the hot block
            chunk_ready();
            const float* As = smem + (ch & 1) * STAGE;
            issue_fast((ch + 1) & 1);
            compute(As, As + BM * 32);
        }
        chunk_ready();
        compute(smem + ((nloop - 1) & 1) * STAGE, smem + ((nloop - 1) & 1) * STAGE + BM * 32);
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------------
    const bool convt = d.out_mode == SMIRK_OUT_CONVT2X2;
    if (SPLIT || (a.N % 8 == 0)) {                              // whole 8-channel groups: transpose through LDS, 2 x 16-byte stores per lane
        __syncthreads();                                          // every wave is done with As/Bs: reuse as transpose buffers
        float* ebuf = smem + wave * 32 * EPI_LD;
        SmirkRangeAcc rng;                                        // split-fp16 range audit: running max of what this lane stores, tested once below
        constexpr int GPR = TN * 4;                               // 8-channel groups per buffer row
        constexpr int ITEMS = 32 * GPR / 64;
        float st1[TN], st2[TN];                                   // STATS: this lane's column sums (sum z, sum z^2) over the wave's rows
#pragma unroll
        for (int j = 0; j < TN; ++j) { st1[j] = 0.f; st2[j] = 0.f; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float x16[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    x16[r] = SPLIT ? acc[0][i][j][r] + acc[NACC - 1][i][j][r] * (1.0f / 2048.0f) : acc[0][i][j][r];
                    ebuf[mfma32_row(r, lane) * EPI_LD + j * 32 + fr] = x16[r];
                }
                if constexpr (STATS) stats_block(x16, lane, a.M - (m0 + (wm * TM + i) * 32), st1[j], st2[j]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // per-wave buffer: no workgroup barrier, and no vmcnt drain (a
                                                                  // __syncthreads here would wait for the previous pass's output stores)
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = it * 64 + lane, row = item / GPR, g = item % GPR;
                const int m = m0 + (wm * TM + i) * 32 + row, n = n0 + wn * TN * 32 + g * 8;
                if (m < a.M && n < a.N) {
                    float v[8];
                    *(f32x4*)v = *(const f32x4*)(ebuf + row * EPI_LD + g * 8);
                    *(f32x4*)(v + 4) = *(const f32x4*)(ebuf + row * EPI_LD + g * 8 + 4);
                    const int co = convt ? n % d.Cout : n;
                    size_t o;
                    int b, y, x;
                    row_to_pixel(m, HoWo, d.Wo, a.psh, b, y, x);
                    if (convt) {
                        const int q = n / d.Cout, dy = q >> 1, dx = q & 1;
                        o = (((size_t)b * 2 * d.Ho + 2 * y + dy) * 2 * d.Wo + 2 * x + dx) * d.Cout + co;
                    } else {
                        o = (((size_t)b * d.Ho + y) * d.Wo + x) * d.Cout + n;
                    }
                    if (a.scale) {
                        const f32x4 s0 = *(const f32x4*)(a.scale + co), s1 = *(const f32x4*)(a.scale + co + 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v[q] *= s0[q]; v[4 + q] *= s1[q]; }
                    }
                    if (a.shift) {
                        const f32x4 s0 = *(const f32x4*)(a.shift + co), s1 = *(const f32x4*)(a.shift + co + 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v[q] += s0[q]; v[4 + q] += s1[q]; }
                    }
                    if (a.residual) {
                        if constexpr (SPLIT) {
                            const half8 rh = *(const half8*)(a.residual + o), rl = *(const half8*)(a.residual + o + 4);
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] += join1(rh[q], rl[q]);
                        } else {
                            const f32x4 r0 = *(const f32x4*)(a.residual + o), r1 = *(const f32x4*)(a.residual + o + 4);
#pragma unroll
                            for (int q = 0; q < 4; ++q) { v[q] += r0[q]; v[4 + q] += r1[q]; }
                        }
                    }
                    if (d.act == SMIRK_ACT_RELU) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                    }
                    if constexpr (SPLIT) {
                        half8 hi, lo;
                        split8(v, hi, lo, rng);
                        *(half8*)(a.out + o) = hi;
                        *(half8*)(a.out + o + 4) = lo;
                    } else {
                        *(f32x4*)(a.out + o) = *(f32x4*)v;
                        *(f32x4*)(a.out + o + 4) = *(f32x4*)(v + 4);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (SPLIT) rng.commit();
        if constexpr (STATS) {                                    // partial row (tile, wave row): [N][2] floats, every (row, channel) written by exactly one lane
            float* prow = a.stats + (size_t)((m0 / BM) * WGM + wm) * a.N * 2;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float t1 = stats_pair(st1[j]), t2 = stats_pair(st2[j]);
                const int n = n0 + (wn * TN + j) * 32 + fr;
                if (hb == 0 && n < a.N) { prow[n * 2] = t1; prow[n * 2 + 1] = t2; }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + fr;
            if (n >= a.N) continue;
            const int co = convt ? n % d.Cout : n;
            const float sc = a.scale ? a.scale[co] : 1.0f, sh = a.shift ? a.shift[co] : 0.0f;
            int dy = 0, dx = 0;
            if (convt) { const int q = n / d.Cout; dy = q >> 1; dx = q & 1; }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + mfma32_row(r, lane);
                    if (m >= a.M) continue;
                    float v = acc[0][i][j][r] * sc + sh;
                    size_t o;
                    int b, y, x;
                    row_to_pixel(m, HoWo, d.Wo, a.psh, b, y, x);
                    if (convt) o = (((size_t)b * 2 * d.Ho + 2 * y + dy) * 2 * d.Wo + 2 * x + dx) * d.Cout + co;
                    else o = (((size_t)b * d.Ho + y) * d.Wo + x) * d.Cout + n;
                    if (a.residual) v += a.residual[o];
                    if (d.act == SMIRK_ACT_RELU) v = fmaxf(v, 0.f);
                    a.out[o] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN, bool SPLIT, int KWALK, bool X1 = false, bool STATS = false>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void conv_igemm_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * 32];
    const int ntn = (a.N + BN - 1) / BN;
    const int logical = xcd_logical(blockIdx.x, gridDim.x);
    const int mt = logical / ntn, nt = logical % ntn;
    conv_tile<BM, BN, WGM, WGN, SPLIT, KWALK, X1, STATS>(a, smem, mt * BM, nt * BN);
}

template <int BM, int BN, int WGM, int WGN, bool SPLIT, int KWALK, bool X1 = false>
static void launch_igemm_kw(const ConvArgs& a, hipStream_t st) {
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    if (g_smirk_prof_on) {                                       // the profiler's label is the instantiation that is launched HERE
        char nm[120];
        snprintf(nm, sizeof(nm), "conv_igemm_kernel<%d,%d,%d,%d,%s,%d>%s", BM, BN, WGM, WGN, SPLIT ? "true" : "false", KWALK, X1 ? "[f16x1]" : "");
        const double px = (double)a.d.B * a.d.H * a.d.W;
        smirk_prof_next(nm, 2.0 * a.M * a.N * a.K, 4.0 * (px * a.Cin + (double)a.M * a.N + (double)a.N * a.K + (a.residual ? (double)a.M * a.N : 0.0)));
    }
    if constexpr (SPLIT && (KWALK == KW_LEAN || KWALK == KW_LEAN_CM || KWALK == KW_FAST_KT)) {      // the walks the training step's layers take
        if (a.stats) {
            SMIRK_LAUNCH((conv_igemm_kernel<BM, BN, WGM, WGN, SPLIT, KWALK, X1, true>), dim3(ntm * ntn), dim3(64 * WGM * WGN), 0, st, a);
            return;
        }
    }
    SMIRK_LAUNCH((conv_igemm_kernel<BM, BN, WGM, WGN, SPLIT, KWALK, X1>), dim3(ntm * ntn), dim3(64 * WGM * WGN), 0, st, a);
}

template <int BM, int BN, int WGM, int WGN, bool SPLIT, bool X1 = false>
static void launch_igemm(const ConvArgs& a, hipStream_t st) {
    const SmirkConvDesc& d = a.d;
    // measured (B=128, same box, tap-major -> channel-major): 56x56 64->128 0.221 -> 0.201 ms, 128->128 0.367 -> 0.344, 256->128 0.725 -> 0.655;
    // 28x28 128->256 0.203 -> 0.184, 256->256 0.357 -> 0.336, 512->256 0.680 -> 0.652; 14x14 layers unchanged; 112x112 64->64 (128x64 tile) 0.51 -> 0.54
    // default ON (measured, B=128, same box, old paths -> lean: 14x14x512 0.346 -> 0.327 ms, 28x28 512->256 0.665 -> 0.590, 56x56 256->128
    // 0.715 -> 0.605, 112x112 64->64 0.50 -> 0.46); the pointer-based K walks below remain for operands the 32-bit buffer offsets cannot address
    const bool lean = true;
    if constexpr (SPLIT && WGM * WGN == 4) {
        const long long b0 = (long long)d.B * d.H * d.W * d.C0 * 4, b1 = (long long)d.B * d.H * d.W * d.C1 * 4, bw = (long long)a.N * a.K * 4;
        if (lean && (d.C0 % CV_BK == 0) && (d.C1 % CV_BK == 0) && b0 < (1ll << 31) && b1 < (1ll << 31) && bw < (1ll << 31)) {
            const bool pow2 = (d.C0 & (d.C0 - 1)) == 0 && (d.C1 & (d.C1 - 1)) == 0, even = ((d.C0 + d.C1) / CV_BK) % 2 == 0;
            const bool want_cm = BN == 128;        // (the 128 x 64 tile keeps the tap-major lean walk: DESIGN.md section 6, the channel-major instantiation of that tile)
            if (want_cm && d.KH == 3 && d.KW == 3 && pow2 && even) launch_igemm_kw<BM, BN, WGM, WGN, SPLIT, KW_LEAN_CM, X1>(a, st);
            else launch_igemm_kw<BM, BN, WGM, WGN, SPLIT, KW_LEAN, X1>(a, st);
            return;
        }
    }
    const bool cmajor = BN == 128 && d.Ho * d.Wo >= 400;
    if constexpr (SPLIT && WGM * WGN == 4) {
        if (cmajor && d.KH == 3 && d.KW == 3 && (d.C0 % CV_BK == 0) && (d.C1 % CV_BK == 0)) {
            launch_igemm_kw<BM, BN, WGM, WGN, SPLIT, KW_CMAJOR, X1>(a, st);
            return;
        }
    }
    if ((d.C0 % CV_BK == 0) && (d.C1 % CV_BK == 0)) launch_igemm_kw<BM, BN, WGM, WGN, SPLIT, KW_FAST, X1>(a, st);
    else if (d.KH * d.KW == 1) launch_igemm_kw<BM, BN, WGM, WGN, SPLIT, KW_FAST_KT, X1>(a, st);
    else launch_igemm_kw<BM, BN, WGM, WGN, SPLIT, KW_GENERIC, X1>(a, st);
}

// conv_pp.hip: 8-wave ping-pong kernel (256 x 128 tile, 3-stage ring) for the deep split-fp16 3x3 layers
bool smirk_conv_pp_eligible(const ConvArgs& a);
int smirk_conv_pp_launch(const ConvArgs& a, hipStream_t st);

// conv_halo.hip: the ping-pong schedule with the A operand staged once per channel chunk (one pixel halo serves all nine taps)
bool smirk_conv_halo_eligible(const ConvArgs& a);
int smirk_conv_halo_launch(const ConvArgs& a, hipStream_t st, bool x1);

// conv_patch.hip: persistent halo-patch kernel for the large-image / few-channel 3x3 layers (split-fp16 only)
bool smirk_conv3x3_patch_eligible(const SmirkConvDesc* d, bool has_residual);
int smirk_conv3x3_patch_launch(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, const float* scale,
                               const float* shift, void* out, hipStream_t st, const float* fw, const float* fb, float* fout,
                               int fcout);

// conv_ring.hip: 16 x 16 patches, weights streamed through an LDS ring, two workgroups per CU (Cout = 64 at 112 x 112), optional fused 2 x 2 max-pool
bool smirk_conv3x3_ring64_eligible(const SmirkConvDesc* d, bool has_residual);
int smirk_conv3x3_ring64_launch(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, const float* scale, const float* shift, void* out,
                                void* pooled, hipStream_t st, float* stats = nullptr, int* stats_rows = nullptr);

static int conv_dispatch_one(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, const float* scale,
                             const float* shift, const void* residual, void* out, void* stream, bool split, bool x1 = false, float* stats = nullptr,
                             int* stats_rows = nullptr);

// The buffer-addressed operand DMA of the split-fp16 kernels (32-bit per-lane offsets, out-of-range rows as the zero padding) needs every
// input tensor below 2 GiB.  A whole 1024-frame shard in one pass exceeds that on the 224^2 / 112^2 layers (6.6 / 3.3 GB): such a layer is
// launched as a few batch chunks that each fit — frames are independent and a frame's result does not depend on the batch it travels in
// (tests/test_scale_gpu.py), so this is the same computation, and each chunk still holds >= 10^4 tiles.  Returns the images per launch.
static int conv_batch_chunk(const SmirkConvDesc* d, bool split) {
    if (!split || d->B <= 1) return d->B;
    const long long px = (long long)d->H * d->W, per = 4 * px * (d->C0 > d->C1 ? d->C0 : d->C1), lim = (1ll << 31) - 1;
    if (per * d->B <= lim || per > lim) return d->B;
    return (int)(lim / per);
}

static int conv_dispatch(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, const float* scale,
                         const float* shift, const void* residual, void* out, void* stream, bool split, bool x1 = false, float* stats = nullptr,
                         int* stats_rows = nullptr) {
    if (!d || !in0 || !w || !out) return SMIRK_ERR_BAD_ARG;
    if (stats_rows) *stats_rows = 0;                            // (a layer run as several batch chunks writes no statistics: the caller reduces the stored tensor)
    const int bc = (d->B > 0 && d->H > 0 && d->W > 0 && d->C0 > 0 && d->C1 >= 0) ? conv_batch_chunk(d, split) : d->B;
    if (bc >= d->B) return conv_dispatch_one(d, in0, in1, w, scale, shift, residual, out, stream, split, x1, stats, stats_rows);
    const size_t ipx = (size_t)d->H * d->W, opx = (size_t)d->Ho * d->Wo * (d->out_mode == SMIRK_OUT_CONVT2X2 ? 4 : 1);
    for (int b0 = 0; b0 < d->B; b0 += bc) {
        SmirkConvDesc dc = *d;
        dc.B = d->B - b0 < bc ? d->B - b0 : bc;
        const int rc = conv_dispatch_one(&dc, (const float*)in0 + b0 * ipx * d->C0, in1 ? (const float*)in1 + b0 * ipx * d->C1 : nullptr, w, scale, shift,
                                         residual ? (const float*)residual + b0 * opx * d->Cout : nullptr, (float*)out + b0 * opx * d->Cout, stream, split, x1);
        if (rc != SMIRK_OK) return rc;
    }
    return SMIRK_OK;
}

static int conv_dispatch_one(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, const float* scale,
                             const float* shift, const void* residual, void* out, void* stream, bool split, bool x1, float* stats, int* stats_rows) {
    if (!d || !in0 || !w || !out) return SMIRK_ERR_BAD_ARG;
    if (stats_rows) *stats_rows = 0;
    if (x1 && !split) return SMIRK_ERR_BAD_ARG;
    const int cq = split ? 8 : 4;                               // channel granule: one 16-byte vector (fp32) / one hi+lo group
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 || d->C0 <= 0 || d->C0 % cq || d->C1 % cq || d->C1 < 0 ||
        (d->C1 > 0 && !in1) || d->Ho <= 0 || d->Wo <= 0 || d->stride <= 0 || (split && d->Cout % 8))
        return SMIRK_ERR_BAD_ARG;
    if (!((d->KH == 3 && d->KW == 3) || (d->KH == 1 && d->KW == 1))) return SMIRK_ERR_UNSUPPORTED;
    if (d->pad_mode == SMIRK_PAD_REFLECT && (d->H < 2 || d->W < 2 || d->pad_t > 1 || d->pad_l > 1)) return SMIRK_ERR_UNSUPPORTED;
    ConvArgs a;
    a.d = *d; a.in0 = (const float*)in0; a.in1 = (const float*)in1; a.w = (const float*)w; a.scale = scale; a.shift = shift;
    a.residual = (const float*)residual; a.out = (float*)out;
    a.Cin = d->C0 + d->C1;
    a.K = d->KH * d->KW * a.Cin;
    a.N = d->Cout;
    if (d->out_mode == SMIRK_OUT_CONVT2X2) {
        if (d->KH != 1 || d->C1 != 0 || d->Ho != d->H || d->Wo != d->W || d->stride != 1 || residual) return SMIRK_ERR_UNSUPPORTED;
        a.N = 4 * d->Cout;
    }
    const long long M = (long long)d->B * d->Ho * d->Wo;
    if (M > (1ll << 30) || (long long)d->B * d->H * d->W > (1ll << 30)) return SMIRK_ERR_UNSUPPORTED;
    a.M = (int)M;
    a.ablate = 0;
    a.stats = nullptr;
    a.psh = 0;
    if (d->KH == 3)                                             // only convs with a halo profit from patch ordering
        while (a.psh < 4 && d->Ho % (2 << a.psh) == 0 && d->Wo % (2 << a.psh) == 0) ++a.psh;
    hipStream_t st = (hipStream_t)stream;
    static const bool no_patch = getenv("SMIRK_DISABLE_PATCH_KERNEL") != nullptr;   // A/B switch for tools/ and tests: neither the patch nor the ring kernels
    // F16X1 lives in conv_igemm_kernel only: the specialised kernels below issue the three-MFMA product unconditionally
    if (split && !x1 && !no_patch && smirk_conv3x3_ring64_eligible(d, residual != nullptr))         // conv_ring.hip: the 64-output-channel layers on large images
        return smirk_conv3x3_ring64_launch(d, in0, in1, w, scale, shift, out, nullptr, st, stats, stats_rows);      // (they write statistics only for a raw epilogue)
    if (split && !x1 && !no_patch && smirk_conv3x3_patch_eligible(d, residual != nullptr))
        return smirk_conv3x3_patch_launch(d, in0, in1, w, scale, shift, out, st, nullptr, nullptr, nullptr, 0);
    // train mode: BatchNorm statistics of the RAW output from the accumulators (ConvArgs::stats), for the kernel families that carry the STATS epilogue
    const bool want_stats = split && stats && stats_rows && d->out_mode == SMIRK_OUT_NHWC && !scale && !shift && !residual && d->act == SMIRK_ACT_NONE;
    if (split && smirk_conv_halo_eligible(a)) {                    // (the halo kernel has an X1 instantiation: conv_halo_x1_kernel)
        if (want_stats) { a.stats = stats; *stats_rows = ((a.M + 255) / 256) * 4; }            // 256-row tiles x 4 wave rows
        return smirk_conv_halo_launch(a, st, x1);
    }
    if (split && !x1 && smirk_conv_pp_eligible(a)) return smirk_conv_pp_launch(a, st);
    if (want_stats) {
        // the K walks launch_igemm picks for operands below 2 GiB (KW_LEAN / KW_LEAN_CM when
        // every source has C % 32 == 0, KW_FAST_KT for 1x1 layers otherwise) carry the STATS epilogue; one partial row per (M tile, wave row)
        const long long b0 = (long long)d->B * d->H * d->W * d->C0 * 4, b1 = (long long)d->B * d->H * d->W * d->C1 * 4, bw = (long long)a.N * a.K * 4;
        const bool lean = (d->C0 % CV_BK == 0) && (d->C1 % CV_BK == 0) && b0 < (1ll << 31) && b1 < (1ll << 31) && bw < (1ll << 31);
        const bool kt = !((d->C0 % CV_BK == 0) && (d->C1 % CV_BK == 0)) && d->KH * d->KW == 1;
        if (lean || kt) {
            const int BMt = a.N > 32 ? 128 : 256, WGMt = a.N > 32 ? 2 : 4;
            a.stats = stats;
            *stats_rows = ((a.M + BMt - 1) / BMt) * WGMt;
        }
    }
    if (split && x1) {                                           // F16X1: the same three tile shapes, one MFMA per block
        if (a.N > 64) launch_igemm<128, 128, 2, 2, true, true>(a, st);
        else if (a.N > 32) launch_igemm<128, 64, 2, 2, true, true>(a, st);
        else launch_igemm<256, 32, 4, 1, true, true>(a, st);
    } else if (split) {
        if (a.N > 64) launch_igemm<128, 128, 2, 2, true>(a, st);
        else if (a.N > 32) launch_igemm<128, 64, 2, 2, true>(a, st);
        else launch_igemm<256, 32, 4, 1, true>(a, st);
    } else {
        if (a.N > 64) launch_igemm<128, 128, 2, 2, false>(a, st);
        else if (a.N > 32) launch_igemm<128, 64, 2, 2, false>(a, st);
        else launch_igemm<256, 32, 4, 1, false>(a, st);
    }
    return smirk_launch_status();
}

extern "C" int smirk_conv_igemm_f32(const SmirkConvDesc* d, const float* in0, const float* in1, const float* w,
                                    const float* scale, const float* shift, const float* residual, float* out,
                                    void* stream) {
    return conv_dispatch(d, in0, in1, w, scale, shift, residual, out, stream, false);
}

// Network tail in one launch: conv3x3 (split16 in) + BN + ReLU + 1x1 conv (Cout 32 -> fcout <= 4) + bias + sigmoid -> NCHW fp32.
// conv3x3 + BN + ReLU with the 2 x 2 max-pool of its output produced by the same launch (the U-Net's encoder blocks: `enc = block(x); pool(enc)`,
// smirk_generator.py:52-59).  SMIRK_ERR_UNSUPPORTED when no kernel with a fused pool serves the shape (the caller then runs conv and pool separately).
extern "C" int smirk_conv3x3_pool_f16x3(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, const float* scale, const float* shift, void* out,
                                        void* pooled, void* stream) {
    if (!d || !in0 || !w || !out || !pooled) return SMIRK_ERR_BAD_ARG;
    if (d->C0 % 8 || d->C1 % 8 || (d->C1 > 0 && !in1) || d->H % 2 || d->W % 2) return SMIRK_ERR_BAD_ARG;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->C0 <= 0) return SMIRK_ERR_BAD_ARG;
    const int bc = conv_batch_chunk(d, true);                      // batch chunks below 2 GiB each: eligibility is a property of the chunk
    if (bc <= 0) return SMIRK_ERR_UNSUPPORTED;
    {
        SmirkConvDesc d0 = *d;
        d0.B = d->B < bc ? d->B : bc;
        if (d0.Cout != 64 || !smirk_conv3x3_ring64_eligible(&d0, false)) return SMIRK_ERR_UNSUPPORTED;
    }
    const size_t px = (size_t)d->H * d->W;
    for (int b0 = 0; b0 < d->B; b0 += bc) {
        SmirkConvDesc dc = *d;
        dc.B = d->B - b0 < bc ? d->B - b0 : bc;
        const int rc = smirk_conv3x3_ring64_launch(&dc, (const float*)in0 + b0 * px * d->C0, in1 ? (const float*)in1 + b0 * px * d->C1 : nullptr, w, scale, shift,
                                                   (float*)out + b0 * px * d->Cout, (float*)pooled + b0 * (px / 4) * d->Cout, (hipStream_t)stream);
        if (rc != SMIRK_OK) return rc;
    }
    return SMIRK_OK;
}

extern "C" int smirk_conv3x3_tail_f16x3(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, const float* scale,
                                        const float* shift, const float* fw, const float* fb, float* out_nchw, int fcout,
                                        void* stream) {
    if (!d || !in0 || !w || !fw || !out_nchw || fcout <= 0 || fcout > 4 || d->Cout != 32) return SMIRK_ERR_BAD_ARG;
    if (d->C0 % 8 || d->C1 % 8 || (d->C1 > 0 && !in1) || d->act != SMIRK_ACT_RELU) return SMIRK_ERR_BAD_ARG;
    if (!smirk_conv3x3_patch_eligible(d, false)) return SMIRK_ERR_UNSUPPORTED;
    const int bc = (d->B > 0 && d->H > 0 && d->W > 0 && d->C0 > 0 && d->C1 >= 0) ? conv_batch_chunk(d, true) : d->B;
    if (bc >= d->B) return smirk_conv3x3_patch_launch(d, in0, in1, w, scale, shift, nullptr, (hipStream_t)stream, fw, fb, out_nchw, fcout);
    const size_t px = (size_t)d->H * d->W;
    for (int b0 = 0; b0 < d->B; b0 += bc) {                     // batch chunks below 2 GiB each (see conv_batch_chunk)
        SmirkConvDesc dc = *d;
        dc.B = d->B - b0 < bc ? d->B - b0 : bc;
        const int rc = smirk_conv3x3_patch_launch(&dc, (const float*)in0 + b0 * px * d->C0, in1 ? (const float*)in1 + b0 * px * d->C1 : nullptr, w, scale, shift,
                                                  nullptr, (hipStream_t)stream, fw, fb, out_nchw + b0 * px * fcout, fcout);
        if (rc != SMIRK_OK) return rc;
    }
    return SMIRK_OK;
}

extern "C" int smirk_conv_igemm_f16x3(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w,
                                      const float* scale, const float* shift, const void* residual, void* out,
                                      void* stream) {
    return conv_dispatch(d, in0, in1, w, scale, shift, residual, out, stream, true);
}

// The same convolution on the same split16 operands with ONE MFMA per product block: only the hi halves (fp16(x), 11 significand bits) enter the product,
// accumulation stays fp32.  This is the 16-bit class of BASELINE config 5 (the reference trains under bf16 autocast: 8 significand bits); the output is
// written split16 like every activation, so it feeds either mode.  Always conv_igemm_kernel (the three-MFMA product is compiled into the specialised kernels).
extern "C" int smirk_conv_igemm_f16x1(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w,
                                      const float* scale, const float* shift, const void* residual, void* out,
                                      void* stream) {
    return conv_dispatch(d, in0, in1, w, scale, shift, residual, out, stream, true, true);
}

// Train mode: the raw convolution (no BatchNorm / activation / residual in the epilogue) that ALSO leaves the per-tile partial column sums (sum z, sum z^2) of its
// output for the BatchNorm that follows (nn.BatchNorm2d in training mode after every convolution of smirk_generator.py:88-119 and of the timm backbones,
// smirk_trainer.py:349-355): `stats` [smirk_conv_stats_rows_max(d)][Cout][2] floats; *rows = partial rows written, 0 when the kernel family that serves this shape
// does not produce them (the caller then reduces the stored tensor: smirk_bn_train_forward_split16).  x1: one MFMA per product block (f16x1).
extern "C" size_t smirk_conv_stats_rows_max(const SmirkConvDesc* d) {
    if (!d || d->B <= 0 || d->Ho <= 0 || d->Wo <= 0) return 0;
    const size_t M = (size_t)d->B * d->Ho * d->Wo;
    const size_t tiles = (M + 127) / 128 * 2 + 4;                // 128-row tiles x 2 wave rows (= 256-row tiles x 4)
    return tiles > 4096 ? tiles : 4096;                          // the persistent patch / ring kernels: <= 3 workgroups per CU x 4 waves (one row each)
}
extern "C" int smirk_conv_igemm_stats_split16(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, void* out, float* stats, int* rows, int x1,
                                              void* stream) {
    if (!stats || !rows) return SMIRK_ERR_BAD_ARG;
    return conv_dispatch(d, in0, in1, w, nullptr, nullptr, nullptr, out, stream, true, x1 != 0, stats, rows);
}

// ------------------------------------------------------------------------------------------------------------------
// split-fp16 companions: conversion, pooling, generator input pack, final 1x1 + sigmoid
// ------------------------------------------------------------------------------------------------------------------
// fp32 [n_groups][8] -> split [n_groups][2][8] halves (and back)
__global__ __launch_bounds__(256) void f32_to_split16_kernel(const float* __restrict__ in, float* __restrict__ out, size_t ng) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ng; i += (size_t)gridDim.x * blockDim.x) {
        float v[8];
        *(f32x4*)v = *(const f32x4*)(in + i * 8);
        *(f32x4*)(v + 4) = *(const f32x4*)(in + i * 8 + 4);
        half8 hi, lo;
        split8(v, hi, lo);
        *(half8*)(out + i * 8) = hi;
        *(half8*)(out + i * 8 + 4) = lo;
    }
}
__global__ __launch_bounds__(256) void split16_to_f32_kernel(const float* __restrict__ in, float* __restrict__ out, size_t ng) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ng; i += (size_t)gridDim.x * blockDim.x) {
        const half8 hi = *(const half8*)(in + i * 8), lo = *(const half8*)(in + i * 8 + 4);
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = join1(hi[q], lo[q]);
        *(f32x4*)(out + i * 8) = *(f32x4*)v;
        *(f32x4*)(out + i * 8 + 4) = *(f32x4*)(v + 4);
    }
}
static unsigned grid_for(size_t total, unsigned cap) { const size_t g = (total + 255) / 256; return (unsigned)(g > cap ? cap : (g ? g : 1)); }

extern "C" int smirk_f32_to_split16(const float* in, void* out, size_t n_elems, void* stream) {
    if (!in || !out || n_elems % 8) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(f32_to_split16_kernel, dim3(grid_for(n_elems / 8, 16384)), dim3(256), 0, (hipStream_t)stream, in, (float*)out, n_elems / 8);
    return smirk_launch_status();
}
extern "C" int smirk_split16_to_f32(const void* in, float* out, size_t n_elems, void* stream) {
    if (!in || !out || n_elems % 8) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(split16_to_f32_kernel, dim3(grid_for(n_elems / 8, 16384)), dim3(256), 0, (hipStream_t)stream, (const float*)in, out, n_elems / 8);
    return smirk_launch_status();
}

static unsigned grid_for(size_t total, unsigned cap);

// Range audit of a split16 tensor (debugging aid behind $SMIRK_F16X3_RANGE_CHECK): the split-fp16 format carries |x| < 65504 only (hi is an fp16);
// counts the values whose hi half is non-finite or whose magnitude reaches `limit`.  counts[0] += offenders, counts[1] = max |x| seen (as float bits).
__global__ __launch_bounds__(256) void split16_range_kernel(const float* __restrict__ in, size_t ng, float limit, unsigned* __restrict__ counts) {
    unsigned bad = 0;
    float mx = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ng; i += (size_t)gridDim.x * blockDim.x) {
        const half8 hi = *(const half8*)(in + i * 8), lo = *(const half8*)(in + i * 8 + 4);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float v = fabsf(join1(hi[q], lo[q]));
            if (!(v < limit)) ++bad;                                  // also catches NaN
            else mx = fmaxf(mx, v);
        }
    }
    if (bad) atomicAdd(counts, bad);
    atomicMax(counts + 1, __float_as_uint(mx));
}

extern "C" int smirk_split16_range_check(const void* in, size_t n_elems, float limit, uint32_t* counts, void* stream) {
    if (!in || !counts || n_elems % 8) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(split16_range_kernel, dim3(grid_for(n_elems / 8, 4096)), dim3(256), 0, (hipStream_t)stream, (const float*)in, n_elems / 8, limit, counts);
    return smirk_launch_status();
}

// 2x2/2 max pool on split tensors: the winning element's (hi, lo) pair is copied unchanged (exact).
__global__ __launch_bounds__(256) void maxpool2x2_split_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H,
                                                               int W, int G) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        size_t t = i / G;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const float* p = in + ((((size_t)b * H + 2 * oy) * W + 2 * ox) * G + g) * 8;
        const size_t dx = (size_t)G * 8, dy = (size_t)W * G * 8;
        half8 bh = *(const half8*)p, bl = *(const half8*)(p + 4);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float* q = p + (k & 1) * dx + (k >> 1) * dy;
            const half8 h = *(const half8*)q, l = *(const half8*)(q + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (join1(h[e], l[e]) > join1(bh[e], bl[e])) { bh[e] = h[e]; bl[e] = l[e]; }
        }
        *(half8*)(out + i * 8) = bh;
        *(half8*)(out + i * 8 + 4) = bl;
    }
}

extern "C" int smirk_maxpool2x2_split16(const void* in, void* out, int B, int H, int W, int C, void* stream) {
    if (!in || !out || B <= 0 || H % 2 || W % 2 || C % 8 || C <= 0) return SMIRK_ERR_BAD_ARG;
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    smirk_prof_next(nullptr, 0.0, 5.0 * total * 32);          // reads 4 groups, writes 1 (32 bytes per split16 group)
    SMIRK_LAUNCH(maxpool2x2_split_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, (const float*)in,
                       (float*)out, B, H, W, C / 8);
    return smirk_launch_status();
}

// two NCHW sources (Ca + Cb <= 8 channels) -> split tensor [B][H][W][1 group]: cat + layout change + split in one pass
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b2, int Cb,
                                                         float* __restrict__ out, int B, int HW) {
    const size_t total = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, p = i % HW;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float x = 0.f;
            if (c < Ca) x = a[(b * Ca + c) * HW + p];
            else if (c < Ca + Cb) x = b2[(b * Cb + (c - Ca)) * HW + p];
            smirk_range_audit1(x);                                // the network INPUT: the one place a NaN can enter from outside (`!(|x| < limit)` catches it)
            v[c] = x;
        }
        half8 hi, lo;
        split8(v, hi, lo);
        *(half8*)(out + i * 8) = hi;
        *(half8*)(out + i * 8 + 4) = lo;
    }
}

extern "C" int smirk_pack_generator_input_split16(const float* a, int Ca, const float* b, int Cb, void* out, int B, int H, int W,
                                                  void* stream) {
    if (!a || !out || B <= 0 || Ca <= 0 || Cb < 0 || Ca + Cb > 8 || (Cb > 0 && !b)) return SMIRK_ERR_BAD_ARG;
    smirk_prof_next(nullptr, 0.0, (double)B * H * W * ((Ca + Cb) * 4 + 32));
    SMIRK_LAUNCH(pack_split_kernel, dim3(grid_for((size_t)B * H * W, 16384)), dim3(256), 0, (hipStream_t)stream, a, Ca, b, Cb,
                       (float*)out, B, H * W);
    return smirk_launch_status();
}

// final 1x1 conv C -> Cout (<=4) + bias + sigmoid on a split input; NCHW fp32 out.
__global__ __launch_bounds__(256) void conv1x1_sigmoid_split_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                                    int B, int HW, int C, int Cout) {
    extern __shared__ float sw[];   // [Cout][C] + [Cout]
    for (int i = threadIdx.x; i < Cout * C; i += blockDim.x) sw[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) sw[Cout * C + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const size_t total = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, p = i % HW;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* src = in + i * C;
        for (int g = 0; g < C / 8; ++g) {
            const half8 hi = *(const half8*)(src + g * 8), lo = *(const half8*)(src + g * 8 + 4);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = join1(hi[k], lo[k]);
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < Cout) acc[o] = fmaf(v, sw[o * C + g * 8 + k], acc[o]);
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < Cout) {
                const float z = acc[o] + sw[Cout * C + o];
                out[(b * Cout + o) * HW + p] = 1.0f / (1.0f + expf(-z));
            }
    }
}

extern "C" int smirk_conv1x1_sigmoid_nchw_split16(const void* in, const float* w, const float* bias, float* out, int B, int H,
                                                  int W, int C, int Cout, void* stream) {
    if (!in || !w || !out || B <= 0 || C % 8 || C <= 0 || Cout <= 0 || Cout > 4) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(conv1x1_sigmoid_split_kernel, dim3(grid_for((size_t)B * H * W, 16384)), dim3(256),
                       (size_t)(Cout * C + Cout) * 4, (hipStream_t)stream, (const float*)in, w, bias, out, B, H * W, C, Cout);
    return smirk_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// memory-bound companions (HBM roofline): 16-byte vectors, one pass
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int B,
                                                         int H, int W, int C4) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const f32x4* p = in + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C4 + c;
        const f32x4 v00 = p[0], v01 = p[C4], v10 = p[(size_t)W * C4], v11 = p[(size_t)W * C4 + C4];
        f32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = fmaxf(fmaxf(v00[k], v01[k]), fmaxf(v10[k], v11[k]));
        out[i] = r;
    }
}

extern "C" int smirk_maxpool2x2_nhwc(const float* in, float* out, int B, int H, int W, int C, void* stream) {
    if (!in || !out || B <= 0 || H % 2 || W % 2 || C % 4 || C <= 0) return SMIRK_ERR_BAD_ARG;
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    SMIRK_LAUNCH(maxpool2x2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x4*)in, (f32x4*)out, B,
                       H, W, C / 4);
    return smirk_launch_status();
}

// NCHW (two optional sources of C0a / C0b channels) -> NHWC with Cpad channels; lanes walk x so both sides coalesce.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b2,
                                                           int Cb, float* __restrict__ out, int B, int HW, int Cpad) {
    const size_t total = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, p = i % HW;
        float* o = out + i * Cpad;
        for (int c4 = 0; c4 < Cpad; c4 += 4) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c4 + k;
                float x = 0.f;
                if (c < Ca) x = a[(b * Ca + c) * HW + p];
                else if (c < Ca + Cb) x = b2[(b * Cb + (c - Ca)) * HW + p];
                v[k] = x;
            }
            *(f32x4*)(o + c4) = v;
        }
    }
}

extern "C" int smirk_nchw_to_nhwc_pad(const float* in, float* out, int B, int Cin, int H, int W, int Cpad, void* stream) {
    if (!in || !out || B <= 0 || Cin <= 0 || Cpad < Cin || Cpad % 4) return SMIRK_ERR_BAD_ARG;
    const size_t total = (size_t)B * H * W;
    const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    SMIRK_LAUNCH(nchw_to_nhwc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, Cin, (const float*)nullptr, 0,
                       out, B, H * W, Cpad);
    return smirk_launch_status();
}

extern "C" int smirk_pack_generator_input(const float* rendered, const float* masked, float* out, int B, int H, int W,
                                          void* stream) {
    if (!rendered || !masked || !out || B <= 0) return SMIRK_ERR_BAD_ARG;
    const size_t total = (size_t)B * H * W;
    const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    SMIRK_LAUNCH(nchw_to_nhwc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, rendered, 3, masked, 3, out, B,
                       H * W, 8);
    return smirk_launch_status();
}

// final 1x1 conv C -> Cout (<=4) + bias + sigmoid; NHWC in, NCHW out.  One pixel per lane: reads C contiguous floats.
__global__ __launch_bounds__(256) void conv1x1_sigmoid_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int B, int HW, int C, int Cout) {
    extern __shared__ float sw[];   // [Cout][C] + [Cout]
    for (int i = threadIdx.x; i < Cout * C; i += blockDim.x) sw[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) sw[Cout * C + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const size_t total = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, p = i % HW;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const f32x4* src = (const f32x4*)(in + i * C);
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const f32x4 v = src[c4];
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < Cout) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[o] = fmaf(v[k], sw[o * C + c4 * 4 + k], acc[o]);
                }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < Cout) {
                const float z = acc[o] + sw[Cout * C + o];
                out[(b * Cout + o) * HW + p] = 1.0f / (1.0f + expf(-z));
            }
    }
}

extern "C" int smirk_conv1x1_sigmoid_nchw(const float* in, const float* w, const float* bias, float* out, int B, int H,
                                          int W, int C, int Cout, void* stream) {
    if (!in || !w || !out || B <= 0 || C % 4 || C <= 0 || Cout <= 0 || Cout > 4) return SMIRK_ERR_BAD_ARG;
    const size_t total = (size_t)B * H * W;
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    SMIRK_LAUNCH(conv1x1_sigmoid_kernel, dim3(grid), dim3(256), (size_t)(Cout * C + Cout) * 4, (hipStream_t)stream, in,
                       w, bias, out, B, H * W, C, Cout);
    return smirk_launch_status();
}
