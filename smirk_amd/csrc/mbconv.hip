// Fused MobileNetV3 block for gfx950 — one launch per InvertedResidual / DepthwiseSeparable block of the timm "minimal" backbones
// (reference call site smirk_encoder.py:11-21 `self.encoder(img)[-1]`):
//
//      x --1x1 expand + BN + ReLU--> E --3x3 depthwise (stride 1|2, TF-SAME) + BN + ReLU--> D --1x1 project + BN (+x)--> out
//
// The unfused path writes E and D to HBM and reads them back (E is 3-6x wider than x: 4E + x + out bytes per block).  Here a workgroup
// owns an 8x8 (stride 1) or 4x8 (stride 2) output tile, stages the matching input halo tile of x in LDS ONCE, and walks the expanded
// channels in chunks of 32:
//      phase 1   E_c[halo px][32]  = relu(bn1(X . Wexp_c))      split-fp16 x3 MFMA (v_mfma_f32_32x32x16_f16), A from LDS, B from L1/L2;
//                                     rows outside the image are forced to 0 (the depthwise conv zero-pads E, not x)      -> LDS fp32
//      phase 2   D_c[out px][32]   = relu(bn2(dw3x3(E_c)))       fp32 VALU, lane = channel                                  -> LDS split16
//      phase 3   P[out px][Cout]  += D_c . Wproj_c               MFMA, accumulators stay in registers across the chunks
// and finally out = bn3(P) (+ x) re-split to the split16 activation layout.  HBM traffic per block: the input tile with its halo
// ((10/8)^2 = 1.56x at stride 1, 1.13x at stride 2) + the output — 5-8x less than the unfused sequence at 112^2..28^2.
// Bound: HBM at 112^2/56^2 (then latency: 2 barriers per 32-channel chunk); arithmetic identical in kind to the unfused kernels
// (f16x3 products, fp32 accumulation, fp32 depthwise), E is kept in fp32 instead of being rounded to split16 in between.
#include <stdio.h>

#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#ifndef MB_ES2
#define MB_ES2 36                 // stride-2 E row stride; 48 is the conflict-free value (variant build: tools/build_variant.sh "-DMB_ES2=48" mbconv.hip)
#endif
#define MB_ES(S) ((S) == 1 ? 32 : MB_ES2)

struct MBArgs {
    const char* x;          // split16 NHWC, Cin*4 bytes per pixel
    const char* wexp;       // [mid][Cin] split16 rows (Cin*4 bytes), NULL for a DepthwiseSeparable block (mid == Cin, E = x)
    const float *s1, *b1;   // [mid]
    const float* wdw;       // [9][mid]
    const float *s2, *b2;   // [mid]
    const char* wproj;      // [Cout][mid] split16 rows
    const float *s3, *b3;   // [Cout]
    char* out;              // split16 NHWC [B][Ho][Wo][Cout]
    int B, H, W, Cin, mid, Cout, Ho, Wo, pt, pl, residual, cinp, coutp, tiles_x, tiles_y, es_floats;
};

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains vmcnt, which would expose the latency of the
// weight-fragment prefetches that are deliberately left in flight across the phases
__device__ __forceinline__ void mb_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ float mb_join(_Float16 hi, _Float16 lo) { return (float)hi + (float)lo * (1.0f / 2048.0f); }

// KS = Cin rounded up to 16, in 16-k MFMA steps (1..3: the fused path serves the blocks with Cin <= 48, i.e. everything down to 14x14)
template <int S, bool EXP, int KS>
__global__ __launch_bounds__(256, 3) void mbconv_fused_kernel(MBArgs a) {
    SmirkRangeAccS rng;                                 // split-fp16 range audit (common.h), scalar-register form (the <1,true,3> instantiation sits at its 168-VGPR budget)
    constexpr int THO = (S == 1) ? 8 : 4, TWO = 8;
    constexpr int HI = (THO - 1) * S + 3, WI = (TWO - 1) * S + 3, NH = HI * WI, MH = (NH + 31) / 32 * 32, MO = THO * TWO;
    // Es row stride (floats).  Phase 2 reads E with ds_read_b128, lane = (pixel p = tid >> 3, channel quad c4 = tid & 7); the hardware services a b128 read in the lane
    // groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (MI355X_MICROARCH.md), i.e. quads 0-3 of pixel j, 4-7 of j + 1, 4-7 of j + 2, 0-3 of j + 3: with a pixel
    // stride of S * ES floats these 16 quads fall into 16 distinct 16-byte bank slots iff S * ES = 32 (mod 64) - 32 | 48 floats.  The former 36 (chosen for 16
    // CONSECUTIVE lanes) made every depthwise tap read a 2-way (stride 1) or 3-way (stride 2) conflict: 33-37 % of the kernel's LDS cycles (profiles/r03v_pmc_census_full.txt).
    // Stride 1 takes 32.  Stride 2 stays at 36 (48 costs 7 KB more LDS per workgroup).  Measured effect of either: none outside the noise - the four fused-MBConv
    // entries of the config-4 bench table sum to 13.6-13.8 ms per 1024 frames before and after (profiles/r03z_, r03s_, r03s2_bench_full.json; the three backbones run on
    // concurrent streams, so single entries trade +-25 % between runs).  The conflict cycles are gone from the depthwise reads; the kernel is bound by parked waves and
    // instruction issue, not by the LDS.
    constexpr int ES = MB_ES(S), DSB = 144;               // Ds row stride (bytes: 32 ch x 4 B + 16)
    constexpr int P3 = (S == 1) ? 2 : 1;                  // project tiles per wave: (MO/32) x (Cout<=96)/32 = 6 | 3 tiles over 4 waves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int strideX = a.cinp * 4 + 16;                  // odd multiple of 16 B: the 16 rows of a ds_read_b128 group hit 16 distinct slots
    const int nchunks = (a.mid + 31) / 32, midp = nchunks * 32;
    char* Xs = smem;
    float* Es = (float*)(Xs + MH * strideX);
    char* Ds = (char*)(Es + a.es_floats);
    float* Wd = (float*)(Ds + MO * DSB);                  // [13][midp]: 9 depthwise taps, s2, b2, s1, b1 (zero beyond mid)
    unsigned char* ok = (unsigned char*)(Wd + 13 * midp);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, hb = lane >> 5;
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int oy0 = ty * THO, ox0 = tx * TWO, iy0 = oy0 * S - a.pt, ix0 = ox0 * S - a.pl;
    const size_t xrow = (size_t)a.Cin * 4;
    const int gmax = a.Cin / 8;
    const int ntn = a.coutp / 32, ntiles = (MO / 32) * ntn;
    const half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};

    half8 we[KS][2];                                      // this lane's expand-weight fragments of the current chunk (column = 32c + fr)
    auto load_we = [&](int c) {
        const int ch = 32 * c + fr;
        const char* wrow = a.wexp + (size_t)(ch < a.mid ? ch : 0) * xrow;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int g = 2 * s + hb;
            const bool v = ch < a.mid && g < gmax;
            we[s][0] = v ? *(const half8*)(wrow + g * 32) : hz;
            we[s][1] = v ? *(const half8*)(wrow + g * 32 + 16) : hz;
        }
    };
    if constexpr (EXP) load_we(0);

    // ---- phase 0: halo tile of x -> LDS (zeros outside the image / beyond Cin / beyond NH); depthwise + BN constants -> LDS -----------
    {
        // Both staging loops request ALL of a lane's elements before the first one is stored: rolled load -> store loops pay one dependent global round trip per
        // iteration (3-8 for the halo tile, 3-12 for the constants), which was most of a tile's ~10 us (24 k cycles for ~30 MFMAs; same defect and fix as the
        // first version of encoder_head.hip).  Addresses are clamped and the value selected afterwards, so the loads carry no branches.
        constexpr int P = KS * 4;                         // 16-byte vectors per staged row (cinp = 16 KS)
        const int Preal = a.Cin / 4;
        const char* xb = a.x + (size_t)b * a.H * a.W * xrow;
        constexpr int NIT = (MH * P + 255) / 256;
        f32x4 xv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it, row = idx / P, pc = idx - row * P;
            const int iy = iy0 + row / WI, ix = ix0 + row % WI;
            const bool in = idx < MH * P && row < NH && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && pc < Preal;
            const int yc = min(max(iy, 0), a.H - 1), xc = min(max(ix, 0), a.W - 1), pcc = min(pc, Preal - 1);
            const f32x4 t = *(const f32x4*)(xb + ((size_t)yc * a.W + xc) * xrow + pcc * 16);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            xv[it] = in ? t : z;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it, row = idx / P, pc = idx - row * P;
            if (idx < MH * P) {
                const int iy = iy0 + row / WI, ix = ix0 + row % WI;
                *(f32x4*)(Xs + row * strideX + pc * 16) = xv[it];
                if (pc == 0) ok[row] = (row < NH && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? 1 : 0;
            }
        }
        for (int base = 0; base < 13 * midp; base += 256 * 4) {
            float cv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + 256 * u + tid, k = min(idx / midp, 12), ch = idx - (idx / midp) * midp, chc = min(ch, a.mid - 1);
                const float* src = k < 9 ? a.wdw + (size_t)k * a.mid : k == 9 ? a.s2 : k == 10 ? a.b2 : (k == 11 ? (EXP ? a.s1 : a.s2) : (EXP ? a.b1 : a.b2));
                const float t = src[chc];
                cv[u] = (idx < 13 * midp && ch < a.mid && (k < 11 || EXP)) ? t : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + 256 * u + tid;
                if (idx < 13 * midp) Wd[idx] = cv[u];
            }
        }
    }
    mb_barrier();

    // validity of the 16 accumulator rows this lane owns in each of its (at most 2) phase-1 tiles: bit r of okm[i]
    unsigned okm[2] = {0u, 0u};
    if constexpr (EXP) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = wave + 4 * i;
            if (t < MH / 32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) okm[i] |= (unsigned)ok[t * 32 + mfma32_row(r, lane)] << r;
            }
        }
    }

    // project-weight fragments of a chunk: fetched BEFORE the depthwise phase where the register budget allows (168 VGPRs at 3 workgroups per CU), so the
    // L2 round trip hides under the depthwise arithmetic instead of sitting in front of the project MFMAs of every chunk
    constexpr bool PREF = (P3 == 1);                    // stride 2: one project tile per wave (stride 1 with prefetch spilled 37-64 VGPRs)
    half8 wpre[PREF ? P3 : 1][2][2];
    auto load_wp = [&](int c, int q, half8 (*dst)[2]) {
        const int id = wave + 4 * q;
        const int nt = id - (id / ntn) * ntn, co = nt * 32 + fr;
        const char* wrow = a.wproj + (size_t)(co < a.Cout ? co : 0) * a.mid * 4;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int kg = min(4 * c + 2 * s + hb, a.mid / 8 - 1);
            const bool v = id < ntiles && co < a.Cout && (4 * c + 2 * s + hb) * 8 < a.mid;
            const half8 t0 = *(const half8*)(wrow + kg * 32), t1 = *(const half8*)(wrow + kg * 32 + 16);
            dst[s][0] = v ? t0 : hz;
            dst[s][1] = v ? t1 : hz;
        }
    };

    f32x16 pacc[P3][2];
#pragma unroll
    for (int q = 0; q < P3; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[q][h][r] = 0.f;

    for (int c = 0; c < nchunks; ++c) {
        // ---- phase 1 ----------------------------------------------------------------------------------------------------------
        if constexpr (EXP) {
            const float s1 = Wd[11 * midp + 32 * c + fr], b1 = Wd[12 * midp + 32 * c + fr];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const int t = wave + 4 * ti;
                if (t >= MH / 32) break;
                f32x16 e0, e1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { e0[r] = 0.f; e1[r] = 0.f; }
                unsigned om = okm[ti];
                if constexpr (S == 1) asm volatile("" : "+v"(om));   // opaque per chunk: the 16 per-row masks are re-derived here (2 VALU each) instead of being hoisted out of the chunk loop and spilled
                const char* arow = Xs + (t * 32 + fr) * strideX;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int g = 2 * s + hb;
                    const half8 ah = *(const half8*)(arow + g * 32), al = *(const half8*)(arow + g * 32 + 16);
                    e0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, we[s][0], e0, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, we[s][1], e1, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, we[s][0], e1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // stride 1: all MH rows are written (Es holds MH rows; rows >= NH are zeros nobody reads): 16 straight-line stores off one base register instead of
                    // 16 exec-masked branches with separately kept addresses
                    const int row = t * 32 + mfma32_row(r, lane);
                    if (S == 1 || row < NH) {                                    // stride 2 keeps the round-2 form (Es holds NH rows there)
                        const float v = fmaxf((e0[r] + e1[r] * (1.0f / 2048.0f)) * s1 + b1, 0.f);
                        Es[row * ES + fr] = ((om >> r) & 1u) ? v : 0.f;          // the depthwise conv zero-pads E (not x)
                    }
                }
            }
            if (c + 1 < nchunks) load_we(c + 1);          // same registers: the fetch runs under phases 2-3
        } else {                                          // DepthwiseSeparable: the depthwise conv reads x itself
            for (int idx = tid; idx < NH * 4; idx += 256) {
                const int row = idx >> 2, g = (idx & 3) + 4 * c;
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
                if (g < gmax) {
                    const half8 hi = *(const half8*)(Xs + row * strideX + g * 32), lo = *(const half8*)(Xs + row * strideX + g * 32 + 16);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v0[k] = mb_join(hi[k], lo[k]); v1[k] = mb_join(hi[4 + k], lo[4 + k]); }
                }
                *(f32x4*)(Es + row * ES + (idx & 3) * 8) = v0;
                *(f32x4*)(Es + row * ES + (idx & 3) * 8 + 4) = v1;
            }
        }
        mb_barrier();
        if constexpr (PREF) {
#pragma unroll
            for (int q = 0; q < P3; ++q) load_wp(c, q, wpre[q]);
        }
        // ---- phase 2: depthwise 3x3 + BN + ReLU; a lane owns 4 consecutive channels of a pixel, 32 pixels per pass ------------------------
        {
            const int c4 = tid & 7, cb = 32 * c + c4 * 4;
            f32x4 w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w[k] = *(const f32x4*)(Wd + k * midp + cb);
            const f32x4 s2 = *(const f32x4*)(Wd + 9 * midp + cb), b2 = *(const f32x4*)(Wd + 10 * midp + cb);
#pragma unroll
            for (int q = 0; q < MO / 32; ++q) {
                const int p = (tid >> 3) + 32 * q, oy = p / TWO, ox = p % TWO;
                const float* e = Es + ((oy * S) * WI + ox * S) * ES + c4 * 4;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 v = *(const f32x4*)(e + (ky * WI + kx) * ES);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] = fmaf(v[k], w[ky * 3 + kx][k], acc[k]);
                    }
                typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                half4 hi, lo;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = fmaxf(acc[k] * s2[k] + b2[k], 0.f);       // (intermediate: an overflow here turns into inf / NaN in every output channel, audited below)
                    _Float16 h, l;
                    smirk_split1(v, h, l);
                    hi[k] = h; lo[k] = l;
                }
                char* d = Ds + p * DSB + (c4 >> 1) * 32 + (c4 & 1) * 8;
                *(half4*)d = hi;
                *(half4*)(d + 16) = lo;
            }
        }
        mb_barrier();
        // ---- phase 3: project, accumulate over chunks --------------------------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < P3; ++q) {
            const int id = wave + 4 * q;
            if (id < ntiles) {
                const int mt = id / ntn;
                const char* arow = Ds + (mt * 32 + fr) * DSB;
                half8 wl[2][2];
                if constexpr (!PREF) load_wp(c, q, wl);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int g = 2 * s + hb;
                    const half8 ah = *(const half8*)(arow + g * 32), al = *(const half8*)(arow + g * 32 + 16);
                    const half8 w0 = PREF ? wpre[q][s][0] : wl[s][0], w1 = PREF ? wpre[q][s][1] : wl[s][1];
                    pacc[q][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w0, pacc[q][0], 0, 0, 0);
                    pacc[q][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w1, pacc[q][1], 0, 0, 0);
                    pacc[q][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w0, pacc[q][1], 0, 0, 0);
                }
            }
        }
        // no barrier: the next chunk's phase 1 writes Es only (last read in phase 2, fenced above); its phase 2 rewrites Ds after the
        // barrier that follows phase 1, which every wave reaches only once its phase 3 is done.
    }
    // ---- epilogue: bn3 -> LDS (re-using Es) -> (+ residual) -> split16 stores ---------------------------------------------------------
    float* Os = Es;
    // the lane id is re-derived (mbcnt) for the epilogue: <1, true, 3> kept a lane-derived value alive across the chunk loop in scratch otherwise
    const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
    for (int q = 0; q < P3; ++q) {
        const int id = wave + 4 * q;
        if (id < ntiles) {
            const int mt = id / ntn, nt = id - mt * ntn, co = nt * 32 + (lane_e & 31);
            const float s3 = co < a.Cout ? a.s3[co] : 0.f, b3 = co < a.Cout ? a.b3[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + mfma32_row(r, lane_e);
                Os[row * a.coutp + co] = (pacc[q][0][r] + pacc[q][1][r] * (1.0f / 2048.0f)) * s3 + b3;
            }
        }
    }
    mb_barrier();
    {
        const int G = a.Cout / 8;
        char* ob = a.out + (size_t)b * a.Ho * a.Wo * a.Cout * 4;
        for (int idx = tid; idx < MO * G; idx += 256) {
            const int p = idx / G, g8 = idx - p * G;
            const int oy = oy0 + p / TWO, ox = ox0 + p % TWO;
            if (oy >= a.Ho || ox >= a.Wo) continue;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = Os[p * a.coutp + g8 * 8 + k];
            if (a.residual) {                             // stride 1, Cin == Cout: x at the same pixel = halo row (y + pt, x + pl)
                const char* xr = Xs + ((p / TWO + a.pt) * WI + (p % TWO) + a.pl) * strideX + g8 * 32;
                const half8 hi = *(const half8*)xr, lo = *(const half8*)(xr + 16);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += mb_join(hi[k], lo[k]);
            }
            half8 hi, lo;
            rng.see8(v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                _Float16 h, l;
                smirk_split1(v[k], h, l);
                hi[k] = h; lo[k] = l;
            }
            char* o = ob + ((size_t)oy * a.Wo + ox) * a.Cout * 4 + g8 * 32;
            *(half8*)o = hi;
            *(half8*)(o + 16) = lo;
        }
    }
    rng.commit();
}

static int mb_same_pad_lead(int n, int s) {
    const int o = (n + s - 1) / s;
    int t = (o - 1) * s + 3 - n;
    if (t < 0) t = 0;
    return t / 2;
}

extern "C" size_t smirk_mbconv_lds_bytes(int Cin, int mid, int Cout, int stride) {
    const int cinp = (Cin + 15) / 16 * 16, coutp = (Cout + 31) / 32 * 32, midp = (mid + 31) / 32 * 32;
    const int NH = stride == 1 ? 100 : 153, MH = (NH + 31) / 32 * 32, MO = stride == 1 ? 64 : 32;
    const int ES = MB_ES(stride);
    const int erows = stride == 1 ? MH : NH;
    const size_t es = (size_t)((erows * ES > MO * coutp) ? erows * ES : MO * coutp);
    return (size_t)MH * (cinp * 4 + 16) + es * 4 + (size_t)MO * 144 + (size_t)13 * midp * 4 + MH;
}

/* 1 if smirk_mbconv_fused_split16 serves this block shape (the caller keeps the unfused kernel sequence otherwise) */
extern "C" int smirk_mbconv_supported(int Cin, int mid, int Cout, int stride) {
    if ((stride != 1 && stride != 2) || Cin % 8 || mid % 8 || Cout % 8 || Cin <= 0 || mid <= 0 || Cout <= 0) return 0;
    return Cin <= 48 && Cout <= 96 && smirk_mbconv_lds_bytes(Cin, mid, Cout, stride) <= 64 * 1024;
}

template <int S, bool EXP>
static void mb_launch(const MBArgs& a, dim3 grid, size_t lds, hipStream_t st, double flop, double bytes) {
    const int ks = a.cinp / 16;
    if (g_smirk_prof_on) {
        char nm[64];
        snprintf(nm, sizeof(nm), "mbconv_fused_kernel<%d,%s,%d>", S, EXP ? "true" : "false", ks);
        smirk_prof_next(nm, flop, bytes);
    }
    if (ks == 1) SMIRK_LAUNCH((mbconv_fused_kernel<S, EXP, 1>), grid, dim3(256), lds, st, a);
    else if (ks == 2) SMIRK_LAUNCH((mbconv_fused_kernel<S, EXP, 2>), grid, dim3(256), lds, st, a);
    else SMIRK_LAUNCH((mbconv_fused_kernel<S, EXP, 3>), grid, dim3(256), lds, st, a);
}

extern "C" int smirk_mbconv_fused_split16(const void* x, const void* wexp, const float* s1, const float* b1, const float* wdw,
                                          const float* s2, const float* b2, const void* wproj, const float* s3, const float* b3,
                                          int residual, void* out, int B, int H, int W, int Cin, int mid, int Cout, int stride,
                                          void* stream) {
    if (!x || !wdw || !s2 || !b2 || !wproj || !s3 || !b3 || !out || B <= 0 || H <= 0 || W <= 0) return SMIRK_ERR_BAD_ARG;
    if (wexp && (!s1 || !b1)) return SMIRK_ERR_BAD_ARG;
    if (!wexp && mid != Cin) return SMIRK_ERR_BAD_ARG;
    if (residual && (stride != 1 || Cin != Cout)) return SMIRK_ERR_BAD_ARG;
    if (!smirk_mbconv_supported(Cin, mid, Cout, stride)) return SMIRK_ERR_UNSUPPORTED;
    MBArgs a;
    a.x = (const char*)x; a.wexp = (const char*)wexp; a.s1 = s1; a.b1 = b1; a.wdw = wdw; a.s2 = s2; a.b2 = b2;
    a.wproj = (const char*)wproj; a.s3 = s3; a.b3 = b3; a.out = (char*)out;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.mid = mid; a.Cout = Cout; a.residual = residual;
    a.Ho = (H + stride - 1) / stride; a.Wo = (W + stride - 1) / stride;
    a.pt = stride == 1 ? 1 : mb_same_pad_lead(H, stride); a.pl = stride == 1 ? 1 : mb_same_pad_lead(W, stride);
    a.cinp = (Cin + 15) / 16 * 16; a.coutp = (Cout + 31) / 32 * 32;
    const int THO = stride == 1 ? 8 : 4, NH = stride == 1 ? 100 : 153, MO = stride == 1 ? 64 : 32;
    a.tiles_x = (a.Wo + 7) / 8; a.tiles_y = (a.Ho + THO - 1) / THO;
    const int erows = stride == 1 ? (NH + 31) / 32 * 32 : NH;
    a.es_floats = (erows * MB_ES(stride) > MO * a.coutp) ? erows * MB_ES(stride) : MO * a.coutp;
    const size_t lds = smirk_mbconv_lds_bytes(Cin, mid, Cout, stride);
    const dim3 grid((unsigned)((size_t)B * a.tiles_x * a.tiles_y));
    hipStream_t st = (hipStream_t)stream;
    const double pin = (double)B * H * W, pout = (double)B * a.Ho * a.Wo;
    const double flop = 2.0 * (wexp ? pin * Cin * mid : 0.0) + 2.0 * pout * mid * 9 + 2.0 * pout * mid * Cout;
    const double bytes = 4.0 * (pin * Cin * (residual ? 2 : 1) + pout * Cout);
    if (stride == 1) { if (wexp) mb_launch<1, true>(a, grid, lds, st, flop, bytes); else mb_launch<1, false>(a, grid, lds, st, flop, bytes); }
    else { if (wexp) mb_launch<2, true>(a, grid, lds, st, flop, bytes); else mb_launch<2, false>(a, grid, lds, st, flop, bytes); }
    return smirk_launch_status();
}
