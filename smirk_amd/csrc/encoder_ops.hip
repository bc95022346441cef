// Memory-bound pieces of the MobileNetV3-minimal encoders (timm tf_mobilenetv3_*_minimal_100; smirk_encoder.py:7-12,34-110)
// for MI355X (gfx950).  The 1x1 (pointwise) convolutions run on the MFMA implicit-GEMM kernel in conv.hip; here are the
// 3->16 stride-2 stem, the depthwise 3x3 stencils (TF 'same' padding), global-average-pool + Linear heads and the
// ExpressionEncoder output clamps.  All NHWC fp32, 16-byte vectors over channels, BatchNorm(eval) folded into scale/shift.
// Bound: HBM (each activation read once, written once).
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
// split-fp16 activation format (see conv.hip): 8 channels = 8 hi halves + 8 lo halves; value = hi + lo * 2^-11
__device__ __forceinline__ void enc_split8(const float* v, half8& hi, half8& lo) {
    smirk_range_audit8(v);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        _Float16 h, l;
        smirk_split1(v[q], h, l);
        hi[q] = h; lo[q] = l;
    }
}
__device__ __forceinline__ float enc_join(_Float16 hi, _Float16 lo) { return (float)hi + (float)lo * (1.0f / 2048.0f); }

// TF 'SAME' leading pad for kernel 3: total = max((ceil(n/s)-1)*s + 3 - n, 0); leading = total/2
__host__ __device__ static inline int same_pad_lead(int n, int s) {
    const int o = (n + s - 1) / s;
    int t = (o - 1) * s + 3 - n;
    if (t < 0) t = 0;
    return t / 2;
}

__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ out, int B, int H, int W, int Cout, int split) {
    extern __shared__ float sw[];   // [27][Cout] transposed weights, then scale[Cout], shift[Cout]
    for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) { const int k = i / Cout, co = i % Cout; sw[i] = w[co * 27 + k]; }
    const bool raw = (split & 2) != 0;                 // training: the bare convolution (BatchNorm takes batch statistics afterwards)
    split &= 1;
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) { sw[27 * Cout + i] = raw ? 1.f : scale[i]; sw[28 * Cout + i] = raw ? 0.f : shift[i]; }
    __syncthreads();
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int pt = same_pad_lead(H, 2), pl = same_pad_lead(W, 2);
    const size_t total = (size_t)B * Ho * Wo, HW = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
        const size_t b = i / ((size_t)Wo * Ho);
        float x[27];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * 2 - pt + ky, ix = ox * 2 - pl + kx;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
                for (int c = 0; c < 3; ++c) x[(ky * 3 + kx) * 3 + c] = ok ? img[(b * 3 + c) * HW + (size_t)iy * W + ix] : 0.f;
            }
        float* o = out + i * Cout;
        for (int co = 0; co < Cout; co += 8) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 27; ++k)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = fmaf(x[k], sw[k * Cout + co + q], acc[q]);
            if (!raw) {
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = fmaxf(acc[q] * sw[27 * Cout + co + q] + sw[28 * Cout + co + q], 0.f);
            }
            if (split) {
                half8 hi, lo;
                enc_split8(acc, hi, lo);
                *(half8*)(o + co) = hi;
                *(half8*)(o + co + 4) = lo;
            } else {
                *(f32x4*)(o + co) = *(f32x4*)acc;
                *(f32x4*)(o + co + 4) = *(f32x4*)(acc + 4);
            }
        }
    }
}

static int stem_launch(const float* img, const float* w, const float* scale, const float* shift, float* out, int B, int H, int W,
                       int Cout, void* stream, int split) {
    if (!img || !w || (!(split & 2) && (!scale || !shift)) || !out || B <= 0 || Cout % 8 || Cout <= 0 || Cout > 64) return SMIRK_ERR_BAD_ARG;
    const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2);
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    smirk_prof_next(nullptr, 54.0 * B * ((H + 1) / 2) * ((W + 1) / 2) * Cout, 4.0 * ((double)B * 3 * H * W + (double)B * ((H + 1) / 2) * ((W + 1) / 2) * Cout));
    SMIRK_LAUNCH(stem_conv_kernel, dim3(grid), dim3(256), (size_t)29 * Cout * 4, (hipStream_t)stream, img, w, scale,
                       shift, out, B, H, W, Cout, split);
    return smirk_launch_status();
}
extern "C" int smirk_stem_conv_s2(const float* img, const float* w, const float* scale, const float* shift, float* out,
                                  int B, int H, int W, int Cout, void* stream) {
    return stem_launch(img, w, scale, shift, out, B, H, W, Cout, stream, 0);
}
extern "C" int smirk_stem_conv_s2_split16(const float* img, const float* w, const float* scale, const float* shift, void* out,
                                          int B, int H, int W, int Cout, void* stream) {
    return stem_launch(img, w, scale, shift, (float*)out, B, H, W, Cout, stream, 1);
}
extern "C" int smirk_stem_conv_s2_raw_split16(const float* img, const float* w, void* out, int B, int H, int W, int Cout, void* stream) {
    return stem_launch(img, w, nullptr, nullptr, (float*)out, B, H, W, Cout, stream, 3);
}

__global__ __launch_bounds__(256) void dwconv3x3_kernel(const f32x4* __restrict__ in, const f32x4* __restrict__ w,
                                                        const f32x4* __restrict__ scale, const f32x4* __restrict__ shift,
                                                        f32x4* __restrict__ out, int B, int H, int W, int C4, int stride,
                                                        int relu) {
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const int pt = stride == 1 ? 1 : same_pad_lead(H, stride), pl = stride == 1 ? 1 : same_pad_lead(W, stride);
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const size_t b = t / Ho;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride - pt + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride - pl + kx;
                if (ix < 0 || ix >= W) continue;
                const f32x4 v = in[((b * H + iy) * W + ix) * C4 + c];
                const f32x4 ww = w[(ky * 3 + kx) * C4 + c];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(v[q], ww[q], acc[q]);
            }
        }
        const f32x4 sc = scale[c], sh = shift[c];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = acc[q] * sc[q] + sh[q];
            acc[q] = relu ? fmaxf(v, 0.f) : v;
        }
        out[i] = acc;
    }
}

// depthwise 3x3 on split16 activations: one lane = one pixel x one 8-channel group (32 bytes in, 32 bytes out), fp32 arithmetic
__global__ __launch_bounds__(256) void dwconv3x3_split_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ out, int B, int H, int W, int G, int stride,
                                                              int relu) {
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const int pt = stride == 1 ? 1 : same_pad_lead(H, stride), pl = stride == 1 ? 1 : same_pad_lead(W, stride);
    const int C = G * 8;
    // a workgroup walks output ROWS (image, oy): row decomposition is scalar work, a lane needs ONE 32-bit division (e / G) per item.  The flat 64-bit
    // index this replaces cost three 64-bit divisions per item — ~45 % of the kernel's instructions, and the kernel is VALU-bound (2.2 TB/s measured)
    const int rows = B * Ho, per_row = Wo * G;
    for (int r = blockIdx.x; r < rows; r += gridDim.x)
    for (int e = threadIdx.x; e < per_row; e += blockDim.x) {
        const size_t b = (size_t)(r / Ho);
        const int oy = r - (int)b * Ho;
        const int ox = e / G, g = e - ox * G;
        const size_t i = (size_t)r * per_row + e;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride - pt + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride - pl + kx;
                if (ix < 0 || ix >= W) continue;
                const float* p = in + (((b * H + iy) * W + ix) * G + g) * 8;
                const half8 hi = *(const half8*)p, lo = *(const half8*)(p + 4);
                const float* ww = w + (ky * 3 + kx) * C + g * 8;
                const f32x4 w0 = *(const f32x4*)ww, w1 = *(const f32x4*)(ww + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[q] = fmaf(enc_join(hi[q], lo[q]), w0[q], acc[q]);
                    acc[4 + q] = fmaf(enc_join(hi[4 + q], lo[4 + q]), w1[q], acc[4 + q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float v = acc[q] * scale[g * 8 + q] + shift[g * 8 + q];
            acc[q] = relu ? fmaxf(v, 0.f) : v;
        }
        half8 hi, lo;
        enc_split8(acc, hi, lo);
        *(half8*)(out + i * 8) = hi;
        *(half8*)(out + i * 8 + 4) = lo;
    }
}

extern "C" int smirk_dwconv3x3_split16(const void* in, const float* w, const float* scale, const float* shift, void* out, int B,
                                       int H, int W, int C, int stride, int relu, void* stream) {
    if (!in || !w || !scale || !shift || !out || B <= 0 || C % 8 || C <= 0 || (stride != 1 && stride != 2)) return SMIRK_ERR_BAD_ARG;
    const size_t total = (size_t)B * ((H + stride - 1) / stride) * ((W + stride - 1) / stride) * (C / 8);
    const size_t rows = (size_t)B * ((H + stride - 1) / stride);
    if (rows > 0x7fffffff) return SMIRK_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)(rows > 16384 ? 16384 : rows);
    smirk_prof_next(nullptr, 18.0 * total * 8, 4.0 * ((double)B * H * W * C + (double)total * 8));
    SMIRK_LAUNCH(dwconv3x3_split_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)in, w, scale, shift,
                       (float*)out, B, H, W, C / 8, stride, relu);
    return smirk_launch_status();
}

extern "C" int smirk_dwconv3x3(const float* in, const float* w, const float* scale, const float* shift, float* out, int B,
                               int H, int W, int C, int stride, int relu, void* stream) {
    if (!in || !w || !scale || !shift || !out || B <= 0 || C % 4 || C <= 0 || (stride != 1 && stride != 2))
        return SMIRK_ERR_BAD_ARG;
    const size_t total = (size_t)B * ((H + stride - 1) / stride) * ((W + stride - 1) / stride) * (C / 4);
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    SMIRK_LAUNCH(dwconv3x3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x4*)in, (const f32x4*)w,
                       (const f32x4*)scale, (const f32x4*)shift, (f32x4*)out, B, H, W, C / 4, stride, relu);
    return smirk_launch_status();
}

// global-average-pool + Linear head in two launches: (1) pooled[b][c] = mean over HW — one workgroup per (face, 256-channel slab), lanes over
// channels so every load is coalesced; (2) out[b][n] = pooled[b] . W[n] + bias — grid (n-tile of 64, face), one wave per output neuron,
// lane-strided dot product + shuffle reduction.  `ws` holds the pooled vectors ([B][C] floats).
__global__ __launch_bounds__(256) void gap_pool_kernel(const float* __restrict__ feat, float* __restrict__ pooled, int HW, int C, int split) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* f = feat + (size_t)b * HW * C;
    float s = 0.f;
    if (split) {
        const _Float16* h = (const _Float16*)f;
        const int off = (c >> 3) * 16 + (c & 7);                        // halves: group base + hi lane; lo is 8 halves further
        for (int p = 0; p < HW; ++p) s += enc_join(h[(size_t)p * C * 2 + off], h[(size_t)p * C * 2 + off + 8]);
    } else {
        for (int p = 0; p < HW; ++p) s += f[(size_t)p * C + c];
    }
    pooled[(size_t)b * C + c] = s / (float)HW;
}

__global__ __launch_bounds__(256) void gap_linear_kernel(const float* __restrict__ pooled_g, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out, int C, int N) {
    extern __shared__ float pooled[];
    const int b = blockIdx.y, n_lo = blockIdx.x * 64, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += blockDim.x) pooled[c] = pooled_g[(size_t)b * C + c];
    __syncthreads();
    const int n_hi = min(n_lo + 64, N);
    for (int n = n_lo + wave; n < n_hi; n += 4) {
        const float* wr = w + (size_t)n * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = fmaf(pooled[c], wr[c], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) out[(size_t)b * N + n] = s + (bias ? bias[n] : 0.f);
    }
}

static int gap_launch(const float* feat, const float* w, const float* bias, float* out, float* ws, int B, int HW, int C, int N,
                      void* stream, int split) {
    if (!feat || !w || !out || !ws || B <= 0 || HW <= 0 || C <= 0 || N <= 0 || C > 8192 || (split && C % 8)) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(gap_pool_kernel, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, feat, ws, HW, C, split);
    SMIRK_LAUNCH(gap_linear_kernel, dim3((N + 63) / 64, B), dim3(256), (size_t)C * 4, (hipStream_t)stream, (const float*)ws, w, bias,
                       out, C, N);
    return smirk_launch_status();
}
extern "C" int smirk_gap_linear(const float* feat, const float* w, const float* bias, float* out, float* ws, int B, int HW, int C,
                                int N, void* stream) {
    return gap_launch(feat, w, bias, out, ws, B, HW, C, N, stream, 0);
}
extern "C" int smirk_gap_linear_split16(const void* feat, const float* w, const float* bias, float* out, float* ws, int B, int HW, int C,
                                        int N, void* stream) {
    return gap_launch((const float*)feat, w, bias, out, ws, B, HW, C, N, stream, 1);
}

__global__ void expression_clamps_kernel(float* __restrict__ p, int B, int n_exp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float* r = p + (size_t)b * (n_exp + 5);
    r[n_exp + 0] = fminf(fmaxf(r[n_exp + 0], 0.f), 1.f);
    r[n_exp + 1] = fminf(fmaxf(r[n_exp + 1], 0.f), 1.f);
    r[n_exp + 2] = fmaxf(r[n_exp + 2], 0.f);
    r[n_exp + 3] = fminf(fmaxf(r[n_exp + 3], -0.2f), 0.2f);
    r[n_exp + 4] = fminf(fmaxf(r[n_exp + 4], -0.2f), 0.2f);
}

extern "C" int smirk_expression_clamps(float* params, int B, int n_exp, void* stream) {
    if (!params || B <= 0 || n_exp < 0) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(expression_clamps_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, B, n_exp);
    return smirk_launch_status();
}
