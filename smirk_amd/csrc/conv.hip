// fp32 implicit-GEMM convolution for MI355X (gfx950) on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: an exact,
// k-ordered fp32 FMA chain at the 157 TFLOP/s matrix rate — there is no TF32 on CDNA4 and none is emulated here).
//
// GEMM view:  out[m][n] = sum_k A[m][k] * Wt[n][k]
//     m = (b, oy, ox) output pixel (NHWC), n = output channel, k = (ky, kx, c) — c contiguous, matching NHWC activations, so an
//     im2col row segment is a contiguous run of channels and every global access is a 16-byte vector.
// * A is gathered on the fly (zero or REFLECT padding = index arithmetic, no padded copy; a channel-concatenated input is two
//   source tensors, so torch.cat((up, skip), 1) of the U-Net decoder is never materialised).
// * Block tile BM x BN x 32, 4 waves; both operands are staged K-contiguous in LDS with a 36-float row stride so one
//   ds_read_b128 per lane feeds FOUR MFMAs: lanes 0-31 take k = 4t..4t+3 and lanes 32-63 take k = 4t+4..4t+7 of each 8-k
//   group (the MFMA's two k slots per instruction are fed the same permutation on A and B, so the sum is unchanged).
// * Next K chunk is prefetched into registers while the current one is consumed from LDS.
// * Epilogue fuses BatchNorm(eval) scale/shift, residual add, ReLU, and (for ConvTranspose2d k2 s2) the 2x2 pixel scatter.
// * blockIdx -> tile mapping is XCD-aware: the 8 XCDs each get a contiguous run of tiles, so blocks that share an A row
//   panel or a weight panel hit the same private L2.
//
// Bound: MFMA fp32.  Algorithmic flop = 2*M*N*K.
#include "common.h"

#define CV_BK 32
#define CV_LDS (CV_BK + 4)

struct ConvArgs {
    SmirkConvDesc d;
    const float *in0, *in1, *w, *scale, *shift, *residual;
    float* out;
    int M, N, K, Cin;
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = (i < 0) ? -i : i;
    return (i >= n) ? (2 * n - 2 - i) : i;
}

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int PA = BM / 32, PB = BN / 32;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * CV_LDS];
    float* As = smem;
    float* Bs = smem + BM * CV_LDS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const SmirkConvDesc& d = a.d;

    // ---- XCD-aware tile id: hardware places block id on XCD id%8; give each XCD a contiguous run of logical tiles ----
    const int ntn = (a.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    int logical;
    {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int m0 = (logical / ntn) * BM, n0 = (logical % ntn) * BN;

    // ---- per-thread staging coordinates -------------------------------------------------------------------------------
    const int col4 = tid & 7, srow = tid >> 3;
    int iy0[PA], ix0[PA], boff[PA];
    const int HoWo = d.Ho * d.Wo;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m0 + srow + 32 * p;
        if (m < a.M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
            iy0[p] = oy * d.stride - d.pad_t;
            ix0[p] = ox * d.stride - d.pad_l;
            boff[p] = b * d.H * d.W;
        } else {
            iy0[p] = -(1 << 28); ix0[p] = 0; boff[p] = 0;      // out of range => always zero-filled (never reflected: guarded below)
        }
    }
    f32x4 ra[PA], rb[PB];
    auto load_chunk = [&](int k0) {
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
            int iy = iy0[p] + ky, ix = ix0[p] + kx;
            bool ok = kval && (iy0[p] > -(1 << 27));
            if (d.pad_mode == SMIRK_PAD_REFLECT) {
                if (ok) { iy = reflect_idx(iy, d.H); ix = reflect_idx(ix, d.W); }
            } else {
                ok = ok && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
            }
            ra[p] = ok ? *(const f32x4*)(src + ((size_t)(boff[p] + iy * d.W + ix) * cs + cc)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int n = n0 + srow + 32 * p;
            rb[p] = (kval && n < a.N) ? *(const f32x4*)(a.w + (size_t)n * a.K + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunk = (a.K + CV_BK - 1) / CV_BK;
    const int fr = lane & 31, kh = (lane >> 5) * 4;
    load_chunk(0);
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PA; ++p) *(f32x4*)(As + (srow + 32 * p) * CV_LDS + col4 * 4) = ra[p];
#pragma unroll
        for (int p = 0; p < PB; ++p) *(f32x4*)(Bs + (srow + 32 * p) * CV_LDS + col4 * 4) = rb[p];
        __syncthreads();
        if (ch + 1 < nchunk) load_chunk((ch + 1) * CV_BK);
#pragma unroll
        for (int kk = 0; kk < CV_BK / 8; ++kk) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const f32x4*)(As + ((wm * TM + i) * 32 + fr) * CV_LDS + kk * 8 + kh);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const f32x4*)(Bs + ((wn * TN + j) * 32 + fr) * CV_LDS + kk * 8 + kh);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------------
    const bool convt = d.out_mode == SMIRK_OUT_CONVT2X2;
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
                float v = acc[i][j][r] * sc + sh;
                size_t o;
                if (convt) {
                    const int b = m / HoWo, rem = m - b * HoWo;
                    const int y = rem / d.Wo, x = rem - y * d.Wo;
                    o = (((size_t)b * 2 * d.Ho + 2 * y + dy) * 2 * d.Wo + 2 * x + dx) * d.Cout + co;
                } else {
                    o = (size_t)m * d.Cout + n;
                }
                if (a.residual) v += a.residual[o];
                if (d.act == SMIRK_ACT_RELU) v = fmaxf(v, 0.f);
                a.out[o] = v;
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN>
static void launch_igemm(const ConvArgs& a, hipStream_t st) {
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN>), dim3(ntm * ntn), dim3(256), 0, st, a);
}

extern "C" int smirk_conv_igemm_f32(const SmirkConvDesc* d, const float* in0, const float* in1, const float* w,
                                    const float* scale, const float* shift, const float* residual, float* out,
                                    void* stream) {
    if (!d || !in0 || !w || !out) return SMIRK_ERR_BAD_ARG;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 || d->C0 <= 0 || d->C0 % 4 || d->C1 % 4 || d->C1 < 0 ||
        (d->C1 > 0 && !in1) || d->Ho <= 0 || d->Wo <= 0 || d->stride <= 0)
        return SMIRK_ERR_BAD_ARG;
    if (!((d->KH == 3 && d->KW == 3) || (d->KH == 1 && d->KW == 1))) return SMIRK_ERR_UNSUPPORTED;
    if (d->pad_mode == SMIRK_PAD_REFLECT && (d->H < 2 || d->W < 2 || d->pad_t > 1 || d->pad_l > 1)) return SMIRK_ERR_UNSUPPORTED;
    ConvArgs a;
    a.d = *d; a.in0 = in0; a.in1 = in1; a.w = w; a.scale = scale; a.shift = shift; a.residual = residual; a.out = out;
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
    hipStream_t st = (hipStream_t)stream;
    if (a.N > 64) launch_igemm<128, 128, 2, 2>(a, st);
    else if (a.N > 32) launch_igemm<128, 64, 2, 2>(a, st);
    else launch_igemm<256, 32, 4, 1>(a, st);
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
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x4*)in, (f32x4*)out, B,
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
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, Cin, (const float*)nullptr, 0,
                       out, B, H * W, Cpad);
    return smirk_launch_status();
}

extern "C" int smirk_pack_generator_input(const float* rendered, const float* masked, float* out, int B, int H, int W,
                                          void* stream) {
    if (!rendered || !masked || !out || B <= 0) return SMIRK_ERR_BAD_ARG;
    const size_t total = (size_t)B * H * W;
    const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, rendered, 3, masked, 3, out, B,
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
    hipLaunchKernelGGL(conv1x1_sigmoid_kernel, dim3(grid), dim3(256), (size_t)(Cout * C + Cout) * 4, (hipStream_t)stream, in,
                       w, bias, out, B, H * W, C, Cout);
    return smirk_launch_status();
}
