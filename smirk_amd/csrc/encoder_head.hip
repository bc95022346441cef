// encoder_head.hip — the first two layers of a timm MobileNetV3-"minimal" backbone in ONE launch (gfx950):
//
//      img --3x3 s2 stem conv (3 -> 16) + BN + ReLU--> S --3x3 depthwise (stride 1 | 2, TF-SAME) + BN + ReLU--> D --1x1 (16 -> 16) + BN (+ S)--> out
//
// (`conv_stem` + `bn1` + `blocks[0][0]` = DepthwiseSeparableConv ds_r1_k3_s1_e1_c16 in the large backbone, ds_r1_k3_s2_e1_c16 in
// the small one; reference call site smirk_encoder.py:18-21,52-55,80-83 `self.encoder(img)[-1]`.)
//
// Why.  Unfused these are three launches over 16-channel tensors at 112 x 112 — 822 MB each at 1024 frames — and the pointwise layer sat on a 256 x 32
// GEMM tile at 0.11 of the HBM roof (round-2 verdict: `conv_igemm_kernel<256,32,4,1,true,2>` 3 x 2.06 ms): five passes over that tensor for 0.4 % of the
// path's flops.  Fused, a workgroup owns a 16 x 16 (stride 1) / 8 x 8 (stride 2) OUTPUT tile:
//      phase A  the (2 TS + 1)^2 x 3 image patch                       -> LDS (fp32; zeros outside the image: TF-SAME pads with zeros)
//      phase B  stem conv on the TS x TS halo the depthwise needs      -> LDS (fp32; zeros outside the 112 x 112 stem image: the depthwise pads S, not img)
//               lane = (pixel, 8-channel half); the half is wave-uniform, so the 216 weights of a lane's outputs are SCALAR operands (s_load -> SGPR),
//               not LDS traffic: 27 LDS reads for 216 FMAs
//      phase C  depthwise 3x3 + BN + ReLU, lane = (pixel, 4-channel quad), quad wave-uniform (36 scalar weights)      -> LDS (fp32)
//      phase D  pointwise 16 -> 16 + BN (+ residual S), lane = (pixel, 8-channel half), weights broadcast from LDS    -> split16 NHWC store
// HBM traffic: the image patch ((37/32)^2 = 1.34x of the tile's pixels at stride 1, 1.20x at stride 2) + the output; S and D never leave the CU.
// Arithmetic: fp32 FMA throughout (the intermediate tensors are NOT rounded to the 22-bit split16 storage in between, unlike the unfused sequence).
// Bound: VALU issue (~950 FMA per lane and tile) at ~0.4 ms per 1024 frames against ~0.3 ms of HBM time; 3-4 workgroups per CU (39 / 34 KB of LDS: the depthwise output re-uses the image patch's LDS) hide the memory and LDS latencies.
#include <stdio.h>

#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// TF 'SAME' leading pad for kernel 3: total = max((ceil(n/s)-1)*s + 3 - n, 0); leading = total/2   (timm layers/padding.py)
__host__ __device__ static inline int eh_same_pad_lead(int n, int s) {
    const int o = (n + s - 1) / s;
    int t = (o - 1) * s + 3 - n;
    if (t < 0) t = 0;
    return t / 2;
}

struct EncHeadArgs {
    const float* img;                                   // [B][3][H][W] fp32
    const float *ws, *ss, *bs;                          // stem  [16][27] (k = (ky,kx,c)), folded BN scale / shift [16]
    const float *wd, *sd, *bd;                          // dw    [9][16], scale / shift [16]
    const char* wp;                                     // pw    split16 rows [16][16] (64 bytes per output channel)
    const float *sp, *bp;                               // pw    scale / shift [16]
    char* out;                                          // split16 NHWC [B][Ho][Wo][16]
    int B, H, W, Hs, Ws, Ho, Wo, residual, tiles_x, tiles_y, pts, pls, ptd, pld;
};

// float offsets of 16-byte quad `quad` of halo position q (column hx) in Ss, and of output pixel p in Ds (see the layout notes in the kernel)
__device__ __forceinline__ int ss_off(int q, int hx, int quad) { return q * 16 + ((quad ^ ((hx >> 1) & 3)) << 2); }
__device__ __forceinline__ int ds_off(int p, int quad) { return p * 16 + ((quad ^ ((p + (p >> 2)) & 3)) << 2); }

template <int S>
__global__ __launch_bounds__(256, 3) void encoder_head_fused_kernel(EncHeadArgs a) {
    SmirkRangeAcc rng;                                  // split-fp16 range audit (common.h)
    constexpr int TO = (S == 1) ? 16 : 8;               // output tile
    constexpr int TS = (TO - 1) * S + 3;                // stem-image halo tile (18 | 17)
    constexpr int TI = (TS - 1) * 2 + 3;                // image patch (37 | 35)
    // LDS layouts (bank rules of MI355X_MICROARCH.md: ds_read_b32 = 2 x 32 lanes over 32 banks; ds_read_b128 = the lane groups {0-3, 12-15, 20-27} /
    // {4-11, 16-19, 28-31} over 64 banks; ds_write_b128 = 8 x 8 lanes over 32 banks).  Round 3's census (profiles/r03v_pmc_census_full.txt) found 65 % of this
    // kernel's LDS cycles in bank conflicts: the stem's stride-2 reads were 2-way, every [pixel][16-float] access 4- to 8-way (a pixel is 64 bytes, so 16 lanes
    // only reach 4 of the 16 16-byte bank slots).
    //   image patch, stride-1 variant: columns split into an even and an odd plane per row (tap kx reads plane kx & 1 at x + (kx >> 1): consecutive lanes ->
    //     consecutive banks) and a row pitch of 41 (2 * 41 = 18 mod 32: the lanes that wrap into the next stem row continue on the next banks): conflict-free;
    //   Ss [halo pixel][16]: the 16-byte quad is XOR-swizzled with (column >> 1) & 3: depthwise tap reads conflict-free at stride 1, 2-way at stride 2 (was 4 / 8);
    //   Ds [pixel][16]: quad ^ ((p + (p >> 2)) & 3): pointwise reads and depthwise writes conflict-free (were 4-way).
    constexpr bool SPLIT = (S == 1);
    constexpr int TIP = SPLIT ? 41 : TI + 1;            // LDS row stride of the patch
    constexpr int HALF = 21;                            // SPLIT: offset of the odd-column plane inside a row (even plane: 19 entries)
    constexpr int ISZ = 3 * TI * TIP, DSZ = TO * TO * 16;
    __shared__ __attribute__((aligned(16))) float IDs[ISZ > DSZ ? ISZ : DSZ];   // image patch [c][y][x]; dead after phase B, then the depthwise output [p][16]
    __shared__ __attribute__((aligned(16))) float Ss[TS * TS * 16];   // stem output [y][x][16]
    float* Is = IDs;
    float* Ds = IDs;
    __shared__ __attribute__((aligned(16))) float Wp[16 * 16];        // pointwise weights [cin][cout] (transposed: a lane reads 8 couts of one cin)
    const int tid = threadIdx.x, wave = tid >> 6;
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int oy0 = ty * TO, ox0 = tx * TO;
    const int sy0 = oy0 * S - a.ptd, sx0 = ox0 * S - a.pld;         // stem-image coordinates of Ss[0][0]
    const int iy0 = sy0 * 2 - a.pts, ix0 = sx0 * 2 - a.pls;         // image coordinates of Is[.][0][0]

    // ---- phase A: image patch, pointwise weights ------------------------------------------------------------------------------------------------
    {
        // all of a lane's patch elements are requested before the first one is stored: one memory round trip per tile instead of one per element
        // (first version: a rolled load -> store loop, 17 dependent global round trips per tile, 2.2 ms per 1024 frames)
        const float* ib = a.img + (size_t)b * 3 * a.H * a.W;
        constexpr int NIT = (3 * TI * TI + 255) / 256;
        float v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + 256 * it;
            const int c = i / (TI * TI), r = i - c * TI * TI, y = r / TI, x = r - y * TI;
            const int iy = iy0 + y, ix = ix0 + x;
            const bool ok = i < 3 * TI * TI && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            // branch-free: always load from a clamped in-image address, select afterwards (predicated loads came out as 17 exec-masked branches)
            const int cc = min(c, 2), yc = min(max(iy, 0), a.H - 1), xc = min(max(ix, 0), a.W - 1);
            const float t = ib[(cc * a.H + yc) * a.W + xc];
            v[it] = ok ? t : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + 256 * it;
            const int c = i / (TI * TI), r = i - c * TI * TI, y = r / TI, x = r - y * TI;
            if (i < 3 * TI * TI) Is[(c * TI + y) * TIP + (SPLIT ? (x & 1) * HALF + (x >> 1) : x)] = v[it];
        }
        {   // thread = (cout, cin): split16 row of cout, group cin / 8, element cin % 8
            const int co = tid >> 4, ci = tid & 15;
            const _Float16* row = (const _Float16*)(a.wp + co * 64) + (ci >> 3) * 16;
            Wp[ci * 16 + co] = (float)row[ci & 7] + (float)row[8 + (ci & 7)] * (1.0f / 2048.0f);
        }
    }
    __syncthreads();

    // ---- phase B: stem conv + BN + ReLU on the TS x TS halo; (pixel, half) with the half uniform per wave -------------------------------------------
    // A lane keeps the accumulators of its (up to) NP pixels and walks the taps in the OUTER loop: the 8 weights of a tap are wave-uniform scalars
    // (s_load -> SGPR operands of the FMAs), fetched once per wave and tap.  The ky loop is kept rolled so that only 72 of the 216 weights are live at a time
    // (fully unrolled, the compiler hoisted all 216 scalar loads and spilled 248 SGPRs).
    {
        constexpr int NP = (TS * TS + 127) / 128;
        const int half = __builtin_amdgcn_readfirstlane(wave >> 1);  // waves 0-1: channels 0-7, waves 2-3: channels 8-15
        const float* w = a.ws + half * 8 * 27;                       // uniform address
        float acc[NP][8];
        int base[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int p = min((tid & 127) + 128 * j, TS * TS - 1);   // clamped lanes recompute the last pixel; they do not store
            const int y = p / TS, x = p - y * TS;
            base[j] = 2 * y * TIP + (SPLIT ? x : 2 * x);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[j][q] = 0.f;
        }
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            const float* wk = w + ky * 9;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float wq[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) wq[q] = wk[q * 27 + kx * 3 + c];
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        const float v = Is[c * TI * TIP + base[j] + ky * TIP + (SPLIT ? (kx & 1) * HALF + (kx >> 1) : kx)];
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc[j][q] = fmaf(v, wq[q], acc[j][q]);
                    }
                }
        }
        float sc[8], sh[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { sc[q] = a.ss[half * 8 + q]; sh[q] = a.bs[half * 8 + q]; }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int p = (tid & 127) + 128 * j;
            if (p < TS * TS) {
                const int y = p / TS, x = p - y * TS;
                const int sy = sy0 + y, sx = sx0 + x;
                const bool in = sy >= 0 && sy < a.Hs && sx >= 0 && sx < a.Ws;   // outside the stem image: the depthwise conv's zero padding
                f32x4 o0, o1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o0[q] = in ? fmaxf(acc[j][q] * sc[q] + sh[q], 0.f) : 0.f;
                    o1[q] = in ? fmaxf(acc[j][4 + q] * sc[4 + q] + sh[4 + q], 0.f) : 0.f;
                }
                *(f32x4*)(Ss + ss_off(p, x, 2 * half)) = o0;
                *(f32x4*)(Ss + ss_off(p, x, 2 * half + 1)) = o1;
            }
        }
    }
    __syncthreads();

    // ---- phase C: depthwise 3x3 + BN + ReLU; (pixel, quad) with the quad uniform per wave -----------------------------------------------------------
    {
        const int quad = __builtin_amdgcn_readfirstlane(wave);
        const float* w = a.wd + quad * 4;                            // [9][16]: tap k at w[k * 16 + q]
        float sc[4], sh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { sc[q] = a.sd[quad * 4 + q]; sh[q] = a.bd[quad * 4 + q]; }
        for (int p = tid & 63; p < TO * TO; p += 64) {
            const int y = p / TO, x = p - y * TO;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 v = *(const f32x4*)(Ss + ss_off((y * S + ky) * TS + x * S + kx, x * S + kx, quad));
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(v[q], w[(ky * 3 + kx) * 16 + q], acc[q]);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaxf(acc[q] * sc[q] + sh[q], 0.f);
            *(f32x4*)(Ds + ds_off(p, quad)) = acc;
        }
    }
    __syncthreads();

    // ---- phase D: pointwise 16 -> 16 + BN (+ residual), split16 store; (pixel, half), half uniform per wave -------------------------------------------
    {
        const int half = __builtin_amdgcn_readfirstlane(wave >> 1);
        float sc[8], sh[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { sc[q] = a.sp[half * 8 + q]; sh[q] = a.bp[half * 8 + q]; }
        char* ob = a.out + (size_t)b * a.Ho * a.Wo * 64;
        for (int p = tid & 127; p < TO * TO; p += 128) {
            const int y = p / TO, x = p - y * TO;
            const int oy = oy0 + y, ox = ox0 + x;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const f32x4 v = *(const f32x4*)(Ds + ds_off(p, c4));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 w0 = *(const f32x4*)(Wp + (c4 * 4 + k) * 16 + half * 8), w1 = *(const f32x4*)(Wp + (c4 * 4 + k) * 16 + half * 8 + 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { acc[q] = fmaf(v[k], w0[q], acc[q]); acc[4 + q] = fmaf(v[k], w1[q], acc[4 + q]); }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = acc[q] * sc[q] + sh[q];
            if (a.residual) {                                        // stride 1: x of the block = S at the same pixel = halo position (y + ptd, x + pld)
                const int rq = (y + a.ptd) * TS + x + a.pld;
                const f32x4 r0 = *(const f32x4*)(Ss + ss_off(rq, x + a.pld, 2 * half)), r1 = *(const f32x4*)(Ss + ss_off(rq, x + a.pld, 2 * half + 1));
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[q] += r0[q]; acc[4 + q] += r1[q]; }
            }
            if (oy < a.Ho && ox < a.Wo) {
                half8 hi, lo;
                rng.see8(acc);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    _Float16 h, l;
                    smirk_split1(acc[q], h, l);
                    hi[q] = h; lo[q] = l;
                }
                char* o = ob + ((size_t)oy * a.Wo + ox) * 64 + half * 32;
                *(half8*)o = hi;
                *(half8*)(o + 16) = lo;
            }
        }
    }
    rng.commit();
}

/* 1 if smirk_encoder_head_fused_split16 serves this stem + first block (the caller keeps the unfused kernel sequence otherwise) */
extern "C" int smirk_encoder_head_supported(int stem_cout, int kind, int cin, int mid, int cout, int stride, int skip) {
    if (stem_cout != 16 || kind != 0 || cin != 16 || mid != 16 || cout != 16 || (stride != 1 && stride != 2)) return 0;
    return !(skip && stride != 1);
}

extern "C" int smirk_encoder_head_fused_split16(const float* img, const float* stem_w, const float* stem_scale, const float* stem_shift,
                                                const float* dw_w, const float* dw_scale, const float* dw_shift, const void* pw_w,
                                                const float* pw_scale, const float* pw_shift, int residual, void* out, int B, int H, int W,
                                                int stride, void* stream) {
    if (!img || !stem_w || !stem_scale || !stem_shift || !dw_w || !dw_scale || !dw_shift || !pw_w || !pw_scale || !pw_shift || !out || B <= 0 ||
        H < 3 || W < 3 || (stride != 1 && stride != 2) || (residual && stride != 1))
        return SMIRK_ERR_BAD_ARG;
    EncHeadArgs a;
    a.img = img; a.ws = stem_w; a.ss = stem_scale; a.bs = stem_shift; a.wd = dw_w; a.sd = dw_scale; a.bd = dw_shift; a.wp = (const char*)pw_w;
    a.sp = pw_scale; a.bp = pw_shift; a.out = (char*)out; a.B = B; a.H = H; a.W = W; a.residual = residual;
    a.Hs = (H + 1) / 2; a.Ws = (W + 1) / 2;
    a.Ho = (a.Hs + stride - 1) / stride; a.Wo = (a.Ws + stride - 1) / stride;
    a.pts = eh_same_pad_lead(H, 2); a.pls = eh_same_pad_lead(W, 2);
    a.ptd = stride == 1 ? 1 : eh_same_pad_lead(a.Hs, 2); a.pld = stride == 1 ? 1 : eh_same_pad_lead(a.Ws, 2);
    const int TO = stride == 1 ? 16 : 8;
    a.tiles_x = (a.Wo + TO - 1) / TO; a.tiles_y = (a.Ho + TO - 1) / TO;
    const size_t nblk = (size_t)B * a.tiles_x * a.tiles_y;
    if (nblk > 0x7fffffffull) return SMIRK_ERR_UNSUPPORTED;
    const double ps = (double)B * a.Hs * a.Ws, po = (double)B * a.Ho * a.Wo;
    smirk_prof_next(nullptr, 2.0 * (ps * 16 * 27 + po * 16 * 9 + po * 256), 4.0 * ((double)B * 3 * H * W + po * 16));
    if (stride == 1) SMIRK_LAUNCH(encoder_head_fused_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a);
    else SMIRK_LAUNCH(encoder_head_fused_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a);
    return smirk_launch_status();
}
