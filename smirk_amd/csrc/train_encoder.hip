// Training-mode companions of the SmirkEncoder backbones on MI355X (BASELINE config 5, encoder slice):  the `tf_mobilenetv3_*_minimal_100`
// feature extractors the reference builds in smirk_encoder.py:7-12 are stem conv 3x3/s2 + depthwise-separable / inverted-residual blocks of
// {pointwise 1x1, depthwise 3x3, BatchNorm, ReLU} + global average pool + Linear head (smirk_encoder.py:14-22, :48-56, :76-85).
// Pointwise convolutions (forward, data gradient, weight gradient) and BatchNorm run on the kernels the generator's training path already has
// (conv.hip / train.hip).  This file adds what is specific to the encoder:
//   * depthwise 3x3 (stride 1 / TF-'SAME' stride 2) data gradient (+ optional skip gradient) and weight gradient,
//   * the stem's weight gradient and data gradient (the image gradient the cycle path sends back into the generator, smirk_trainer.py:293-297),
//   * global-average-pool + Linear backward.
// All of it is HBM-bound streaming work (a depthwise tap is 1 MAC per 4 bytes): one lane = one pixel x one 8-channel split16 group, fp32
// arithmetic, reductions two-stage with fp64 partials in a fixed order (bit-reproducible).
#include <stdio.h>

#include "conv_common.h"

namespace {

__device__ __forceinline__ void load_group(const float* p, float* v) {
    const half8 hi = *(const half8*)p, lo = *(const half8*)(p + 4);
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = join1(hi[q], lo[q]);
}
__device__ __forceinline__ void store_group(float* p, const float* v) {
    half8 hi, lo;
    split8(v, hi, lo);
    *(half8*)p = hi;
    *(half8*)(p + 4) = lo;
}
inline unsigned blocks_for(size_t items, unsigned cap) {
    const size_t g = (items + 255) / 256;
    return (unsigned)(g > cap ? cap : (g ? g : 1));
}
// TF 'SAME' leading pad for kernel 3 (timm Conv2dSame): total = max((ceil(n/s)-1)*s + 3 - n, 0); leading = total/2.  stride 1 => 1.
__host__ __device__ inline int pad_lead(int n, int s) {
    if (s == 1) return 1;
    const int o = (n + s - 1) / s;
    int t = (o - 1) * s + 3 - n;
    if (t < 0) t = 0;
    return t / 2;
}

// dX[b,iy,ix,c] = sum_{ky,kx} dZ[b,oy,ox,c] * w[ky,kx,c]  over the outputs whose tap (ky,kx) reads (iy,ix):  oy*s - pt + ky = iy
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w /*[9][C]*/, const float* __restrict__ add,
                                                       float* __restrict__ dx, int B, int H, int W, int G, int stride) {
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const int pt = pad_lead(H, stride), pl = pad_lead(W, stride), C = G * 8;
    const size_t total = (size_t)B * H * W * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        size_t t = i / G;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const size_t b = t / H;
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        if (add) load_group(add + i * 8, acc);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + pt - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + pl - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                float v[8];
                load_group(dz + (((b * Ho + oy) * Wo + ox) * G + g) * 8, v);
                const float* ww = w + (ky * 3 + kx) * C + g * 8;
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = fmaf(v[q], ww[q], acc[q]);
            }
        }
        store_group(dx + i * 8, acc);
    }
}
// (Round 6 re-indexed this kernel like dwconv3x3_train_kernel — a lane keeps its channel group, the group's 72 weights in registers, two 32-bit divisions per item —
// and measured it SLOWER: 1.06 -> 1.42 ms per training step (1.9 -> 1.4 TB/s).  The flat index lets consecutive lanes walk consecutive 32-byte pieces of memory whatever G
// is; the (group, lane) mapping leaves 256 mod G lanes idle and costs 72 VGPRs of occupancy.  Reverted.)

// dW[ky,kx,c] = sum over output pixels of dZ[b,oy,ox,c] * X[b, oy*s-pt+ky, ox*s-pl+kx, c].  Stage 1: each block walks its share of the output rows,
// thread = (row lane, channel group), 72 fp32 partial sums per thread (a thread sees <= a few dozen rows), combined over the row lanes in fp64.
#define DW_RED_BLOCKS 256
__global__ __launch_bounds__(256) void dw_wgrad_stage1(const float* __restrict__ dz, const float* __restrict__ x, int B, int H, int W, int G, int stride,
                                                       double* __restrict__ part /*[blocks][C][9]*/) {
    __shared__ float red[256 * 24];
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const int pt = pad_lead(H, stride), pl = pad_lead(W, stride);
    const int tid = threadIdx.x, RPB = 256 / G, g = tid % G, rl = tid / G;
    const bool active = rl < RPB;
    const size_t M = (size_t)B * Ho * Wo;
    float acc[9][8];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[k][q] = 0.f;
    if (active)
        for (size_t r = (size_t)blockIdx.x * RPB + rl; r < M; r += (size_t)gridDim.x * RPB) {
            const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho);
            const size_t b = r / ((size_t)Wo * Ho);
            float d[8];
            load_group(dz + (r * G + g) * 8, d);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * stride - pt + ky;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox * stride - pl + kx;
                    if (ix < 0 || ix >= W) continue;
                    float v[8];
                    load_group(x + (((b * H + iy) * W + ix) * G + g) * 8, v);
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[ky * 3 + kx][q] = fmaf(d[q], v[q], acc[ky * 3 + kx][q]);
                }
            }
        }
    // combine the row lanes (fixed order) three taps at a time: 24 sums per thread through LDS, G * 24 threads each add one column in fp64
    for (int k0 = 0; k0 < 9; k0 += 3) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int q = 0; q < 8; ++q) red[tid * 24 + k * 8 + q] = active ? acc[k0 + k][q] : 0.f;
        __syncthreads();
        for (int o = tid; o < G * 24; o += 256) {
            const int gg = o / 24, e = o % 24, k = k0 + e / 8, q = e % 8;
            double sum = 0.0;
            for (int j = 0; j < RPB; ++j) sum += (double)red[(j * G + gg) * 24 + e];
            part[((size_t)blockIdx.x * G * 8 + gg * 8 + q) * 9 + k] = sum;
        }
    }
}
template <bool PARAM>     // PARAM: write nn.Conv2d's depthwise parameter layout [C][1][3][3] instead of the kernels' tap-major [9][C]
__global__ __launch_bounds__(256) void dw_wgrad_stage2(const double* __restrict__ part, int nblocks, int C, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * 9) return;
    const int c = i / 9, k = i % 9;
    double s[4] = {0.0, 0.0, 0.0, 0.0};                                    // 4 independent chains keep the loads in flight; fixed combination order
    int b = 0;
    for (; b + 4 <= nblocks; b += 4)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += part[((size_t)(b + j) * C + c) * 9 + k];
    for (; b < nblocks; ++b) s[0] += part[((size_t)b * C + c) * 9 + k];
    dw[PARAM ? i : k * C + c] = (float)((s[0] + s[1]) + (s[2] + s[3]));
}

// ---- stem: Conv2d(3, Cout, 3, stride 2, TF 'SAME'), image NCHW fp32, output gradient split16 [B][Ho][Wo][Cout] ----------------------------
// weight gradient dW[co][ky][kx][c]: 27*Cout outputs; a block stages 16 output pixels (their 27 image taps and Cout gradients) in LDS, every thread
// owns outputs {tid, tid+256, ...}; per-block fp32 partials -> fp64 stage 2.
#define STEM_PIX 16
#define STEM_BLOCKS 1024
__global__ __launch_bounds__(256) void stem_wgrad_stage1(const float* __restrict__ img, const float* __restrict__ dz, int B, int H, int W, int Cout,
                                                         float* __restrict__ part /*[blocks][27*Cout]*/) {
    __shared__ float sx[STEM_PIX][28], sd[STEM_PIX][64];
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, pt = pad_lead(H, 2), pl = pad_lead(W, 2), tid = threadIdx.x;
    const size_t total = (size_t)B * Ho * Wo, HW = (size_t)H * W;
    const int nout = 27 * Cout;
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                    // nout <= 27*64 = 1728 <= 7*256
    for (size_t base = (size_t)blockIdx.x * STEM_PIX; base < total; base += (size_t)gridDim.x * STEM_PIX) {
        for (int e = tid; e < STEM_PIX * 27; e += 256) {
            const int p = e / 27, k = e % 27, c = k % 3, kx = (k / 3) % 3, ky = k / 9;
            const size_t i = base + p;
            float v = 0.f;
            if (i < total) {
                const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
                const size_t b = i / ((size_t)Wo * Ho);
                const int iy = oy * 2 - pt + ky, ix = ox * 2 - pl + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = img[(b * 3 + c) * HW + (size_t)iy * W + ix];
            }
            sx[p][k] = v;
        }
        for (int e = tid; e < STEM_PIX * Cout; e += 256) {
            const int p = e / Cout, co = e % Cout;
            const size_t i = base + p;
            float v = 0.f;
            if (i < total) {
                const _Float16* h = (const _Float16*)(dz + i * Cout);
                const int off = (co >> 3) * 16 + (co & 7);
                v = join1(h[off], h[off + 8]);
            }
            sd[p][co] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int o = tid + j * 256;
            if (o < nout) {
                const int co = o / 27, k = o % 27;
#pragma unroll
                for (int p = 0; p < STEM_PIX; ++p) acc[j] = fmaf(sd[p][co], sx[p][k], acc[j]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int o = tid + j * 256;
        if (o < nout) part[(size_t)blockIdx.x * nout + o] = acc[j];
    }
}
__global__ __launch_bounds__(256) void stem_wgrad_stage2(const float* __restrict__ part, int nblocks, int nout, float* __restrict__ dw) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nout) return;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    int b = 0;
    for (; b + 4 <= nblocks; b += 4)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += (double)part[(size_t)(b + j) * nout + o];
    for (; b < nblocks; ++b) s[0] += (double)part[(size_t)b * nout + o];
    dw[o] = (float)((s[0] + s[1]) + (s[2] + s[3]));
}
// data gradient: dImg[b,c,iy,ix] = sum_{ky,kx,co} dZ[b,oy,ox,co] * w[co][ky][kx][c]
__global__ __launch_bounds__(256) void stem_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w /*[Cout][27]*/, float* __restrict__ dimg,
                                                         int B, int H, int W, int Cout) {
    extern __shared__ float sw[];                                          // [27][Cout]
    for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) { const int k = i / Cout, co = i % Cout; sw[i] = w[co * 27 + k]; }
    __syncthreads();
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, pt = pad_lead(H, 2), pl = pad_lead(W, 2), G = Cout / 8;
    const size_t total = (size_t)B * H * W, HW = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ix = (int)(i % W), iy = (int)((i / W) % H);
        const size_t b = i / HW;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + pt - ky;
            if (ty < 0 || (ty & 1) || (ty >> 1) >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + pl - kx;
                if (tx < 0 || (tx & 1) || (tx >> 1) >= Wo) continue;
                const float* src = dz + ((b * Ho + (ty >> 1)) * Wo + (tx >> 1)) * Cout;
                for (int g = 0; g < G; ++g) {
                    float v[8];
                    load_group(src + g * 8, v);
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[c] = fmaf(v[q], sw[((ky * 3 + kx) * 3 + c) * Cout + g * 8 + q], acc[c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dimg[(b * 3 + c) * HW + (size_t)iy * W + ix] = acc[c];
    }
}

// ---- global average pool + Linear backward ------------------------------------------------------------------------------------------------
// dW[n][c] = sum_b dOut[b][n] * pooled[b][c];  db[n] = sum_b dOut[b][n]     (fp64 accumulation, fixed order)
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ dout, const float* __restrict__ pooled, float* __restrict__ dw,
                                                         float* __restrict__ db, int B, int C, int N) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)N * C) {
        const int n = (int)(i / C), c = (int)(i % C);
        double s = 0.0;
        for (int b = 0; b < B; ++b) s += (double)dout[(size_t)b * N + n] * (double)pooled[(size_t)b * C + c];
        dw[i] = (float)s;
    }
    if (i < (size_t)N) {
        double s = 0.0;
        for (int b = 0; b < B; ++b) s += (double)dout[(size_t)b * N + i];
        db[i] = (float)s;
    }
}
// dFeat[b,p,c] = (sum_n dOut[b][n] * W[n][c]) / HW   for every pixel p
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ w, float* __restrict__ dfeat, int B, int HW,
                                                         int C, int N) {
    const int G = C / 8;
    const size_t total = (size_t)B * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        const size_t b = i / G;
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        for (int n = 0; n < N; ++n) {
            const float d = dout[b * N + n];
            const float* wr = w + (size_t)n * C + g * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = fmaf(d, wr[q], acc[q]);
        }
        const float inv = 1.0f / (float)HW;
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] *= inv;
        for (int p = 0; p < HW; ++p) store_group(dfeat + ((b * HW + p) * G + g) * 8, acc);
    }
}

// Train-mode depthwise 3x3 (raw output: BatchNorm takes batch statistics afterwards) that ALSO leaves the statistics' stage-1 partial sums (round 6).
// Thread = (8-channel group g = tid % G, row lane tid / G), like the column reductions of train.hip: a thread keeps its group for the whole launch, so the 72
// weights of the group are loaded ONCE into registers (the inference kernel re-fetches eight per tap and item and spends a 32-bit division per item on
// e -> (pixel, group)), walks output pixels r = block * RPB + lane + k * stride, and adds what it stores into per-thread fp64 sums; the block's sums are combined in
// a fixed order through LDS exactly as colsum_stage1 does and written as one partial row [block][C][2] (sum z, sum z^2) — the layout bn_finalize_kernel /
// bn_apply_fin_kernel<double> consume.  The separate statistics pass over the depthwise output (41 launches and a full read of the tensor per step) is gone.
__global__ __launch_bounds__(256) void dwconv3x3_train_kernel(const float* __restrict__ in, const float* __restrict__ w /*[9][C]*/, float* __restrict__ out, int B,
                                                              int H, int W, int G, int stride, double* __restrict__ part /*[blocks][C][2]*/) {
    __shared__ double red[256 * 16];
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const int pt = pad_lead(H, stride), pl = pad_lead(W, stride), C = G * 8;
    const int tid = threadIdx.x, g = tid % G, rl = tid / G, RPB = 256 / G;
    const bool active = rl < RPB;
    float wt[9][8];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const f32x4 w0 = *(const f32x4*)(w + k * C + g * 8), w1 = *(const f32x4*)(w + k * C + g * 8 + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { wt[k][q] = w0[q]; wt[k][4 + q] = w1[q]; }
    }
    double s1[8], s2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { s1[q] = 0.0; s2[q] = 0.0; }
    const unsigned M = (unsigned)B * Ho * Wo, HoWo = (unsigned)Ho * Wo;
    // All 18 loads of a pixel go out before the first is consumed: taps outside the image read through a buffer resource at an out-of-range offset (zeros) instead of
    // branching around the load (a branch per tap serialises the nine round trips: the first version of this kernel measured 1.1 TB/s, like the inference kernel).
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, (short)0, (int)((size_t)B * H * W * C * 4), 0x00020000);
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    for (unsigned r = blockIdx.x * RPB + rl; active && r < M; r += gridDim.x * RPB) {
        const unsigned b = r / HoWo, rem = r - b * HoWo;
        const int oy = (int)(rem / Wo), ox = (int)(rem - (unsigned)oy * Wo);
        u32x4_t th[9], tl[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * stride - pt + ky, ix = ox * stride - pl + kx;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const unsigned off = ok ? (unsigned)((((b * H + iy) * W + ix) * G + g) * 32u) : 0x80000000u;
                th[ky * 3 + kx] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                tl[ky * 3 + kx] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 16, 0);
            }
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            union { u32x4_t u; half8 h; } uh, ul;
            uh.u = th[k]; ul.u = tl[k];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = fmaf(join1(uh.h[q], ul.h[q]), wt[k][q], acc[q]);
        }
        store_group(out + ((size_t)r * G + g) * 8, acc);
#pragma unroll
        for (int q = 0; q < 8; ++q) { s1[q] += (double)acc[q]; s2[q] += (double)acc[q] * (double)acc[q]; }
    }
    for (int q = 0; q < 8; ++q) { red[tid * 16 + q * 2] = s1[q]; red[tid * 16 + q * 2 + 1] = s2[q]; }
    __syncthreads();
    for (int o = tid; o < G * 16; o += 256) {                         // fixed order: row lane 0, 1, ... of each (group, channel, which) column
        const int gg = o >> 4, k = o & 15;
        double a = 0.0;
        for (int rr = 0; rr < RPB; ++rr) a += red[(rr * G + gg) * 16 + k];
        part[((size_t)blockIdx.x * G * 8 + gg * 8 + (k >> 1)) * 2 + (k & 1)] = a;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------------------
/* Train-mode depthwise 3x3 (smirk_encoder.py:7-12's timm blocks after `self.train()`): the raw convolution + the stage-1 partial sums of its output for the
 * BatchNorm that follows.  part: [rows][C][2] fp64, room for 1024 rows (1024 * C * 16 bytes); *rows = rows written.  SMIRK_ERR_UNSUPPORTED for inputs of 2 GiB and
 * more (the caller then runs smirk_dwconv3x3_split16 + smirk_bn_train_forward_split16). */
extern "C" int smirk_dwconv3x3_stats_split16(const void* in, const float* w, void* out, int B, int H, int W, int C, int stride, double* part, int* rows,
                                             void* stream) {
    if (!in || !w || !out || !part || !rows || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || C / 8 > 256 || (stride != 1 && stride != 2)) return SMIRK_ERR_BAD_ARG;
    const int G = C / 8, RPB = 256 / G;
    const size_t M = (size_t)B * ((H + stride - 1) / stride) * ((W + stride - 1) / stride);
    if (M > 0x7fffffffull || (size_t)B * H * W * C * 4 >= (1ull << 31)) return SMIRK_ERR_UNSUPPORTED;     // (32-bit buffer offsets into the input)
    size_t nb = (M + RPB - 1) / RPB;
    if (nb > 1024) nb = 1024;                                        // four 4-wave workgroups per CU; <= 1024 fp64 partial rows for the finalisation
    *rows = (int)nb;
    smirk_prof_next(nullptr, 18.0 * (double)M * C, 4.0 * ((double)B * H * W * C + (double)M * C));
    SMIRK_LAUNCH(dwconv3x3_train_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float*)in, w, (float*)out, B, H, W, G, stride, part);
    return smirk_launch_status();
}

extern "C" int smirk_dwconv3x3_dgrad_split16(const void* dz, const float* w, const void* add, void* dx, int B, int H, int W, int C, int stride, void* stream) {
    if (!dz || !w || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || (stride != 1 && stride != 2)) return SMIRK_ERR_BAD_ARG;
    smirk_prof_next(nullptr, 18.0 * B * H * W * C / (stride * stride), 4.0 * B * H * W * C * (1.0 + 1.0 / (stride * stride) + (add ? 1.0 : 0.0)));
    SMIRK_LAUNCH(dw_dgrad_kernel, dim3(blocks_for((size_t)B * H * W * (C / 8), 16384)), dim3(256), 0, (hipStream_t)stream, (const float*)dz, w,
                 (const float*)add, (float*)dx, B, H, W, C / 8, stride);
    return smirk_launch_status();
}

extern "C" size_t smirk_dwconv3x3_wgrad_workspace_bytes(int C) { return (size_t)DW_RED_BLOCKS * (size_t)C * 9 * sizeof(double); }

static int dw_wgrad_impl(const void* dz, const void* x, float* dw, int B, int H, int W, int C, int stride, void* ws, size_t ws_bytes, void* stream,
                         bool param_layout) {
    if (!dz || !x || !dw || !ws || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || C / 8 > 256 || (stride != 1 && stride != 2)) return SMIRK_ERR_BAD_ARG;
    if (ws_bytes < smirk_dwconv3x3_wgrad_workspace_bytes(C)) return SMIRK_ERR_WORKSPACE;
    const int G = C / 8, RPB = 256 / G;
    const size_t M = (size_t)B * ((H + stride - 1) / stride) * ((W + stride - 1) / stride);
    const unsigned nb = (unsigned)((M + RPB - 1) / RPB > DW_RED_BLOCKS ? DW_RED_BLOCKS : (M + RPB - 1) / RPB);
    hipStream_t st = (hipStream_t)stream;
    smirk_prof_next(nullptr, 18.0 * (double)M * C, 4.0 * ((double)B * H * W * C + (double)M * C));
    SMIRK_LAUNCH(dw_wgrad_stage1, dim3(nb), dim3(256), 0, st, (const float*)dz, (const float*)x, B, H, W, G, stride, (double*)ws);
    if (param_layout) SMIRK_LAUNCH(dw_wgrad_stage2<true>, dim3((C * 9 + 255) / 256), dim3(256), 0, st, (const double*)ws, (int)nb, C, dw);
    else SMIRK_LAUNCH(dw_wgrad_stage2<false>, dim3((C * 9 + 255) / 256), dim3(256), 0, st, (const double*)ws, (int)nb, C, dw);
    return smirk_launch_status();
}
extern "C" int smirk_dwconv3x3_wgrad_split16(const void* dz, const void* x, float* dw, int B, int H, int W, int C, int stride, void* ws, size_t ws_bytes,
                                             void* stream) {
    return dw_wgrad_impl(dz, x, dw, B, H, W, C, stride, ws, ws_bytes, stream, false);
}
/* the same gradient written in the parameter's layout [C][1][3][3] (what autograd hands the optimiser for the depthwise `conv_dw.weight.grad`) */
extern "C" int smirk_dwconv3x3_wgrad_param_split16(const void* dz, const void* x, float* dw, int B, int H, int W, int C, int stride, void* ws, size_t ws_bytes,
                                                   void* stream) {
    return dw_wgrad_impl(dz, x, dw, B, H, W, C, stride, ws, ws_bytes, stream, true);
}

extern "C" size_t smirk_stem_conv_s2_wgrad_workspace_bytes(int Cout) { return (size_t)STEM_BLOCKS * 27 * (size_t)Cout * sizeof(float); }

extern "C" int smirk_stem_conv_s2_wgrad_split16(const float* img, const void* dz, float* dw, int B, int H, int W, int Cout, void* ws, size_t ws_bytes,
                                                void* stream) {
    if (!img || !dz || !dw || !ws || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 8 || Cout > 64) return SMIRK_ERR_BAD_ARG;
    if (ws_bytes < smirk_stem_conv_s2_wgrad_workspace_bytes(Cout)) return SMIRK_ERR_WORKSPACE;
    const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2);
    const unsigned nb = (unsigned)((total + STEM_PIX - 1) / STEM_PIX > STEM_BLOCKS ? STEM_BLOCKS : (total + STEM_PIX - 1) / STEM_PIX);
    hipStream_t st = (hipStream_t)stream;
    smirk_prof_next(nullptr, 54.0 * (double)total * Cout, 4.0 * ((double)B * 3 * H * W + (double)total * Cout));
    SMIRK_LAUNCH(stem_wgrad_stage1, dim3(nb), dim3(256), 0, st, img, (const float*)dz, B, H, W, Cout, (float*)ws);
    SMIRK_LAUNCH(stem_wgrad_stage2, dim3((27 * Cout + 255) / 256), dim3(256), 0, st, (const float*)ws, (int)nb, 27 * Cout, dw);
    return smirk_launch_status();
}

extern "C" int smirk_stem_conv_s2_dgrad_split16(const void* dz, const float* w, float* dimg, int B, int H, int W, int Cout, void* stream) {
    if (!dz || !w || !dimg || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 8 || Cout > 64) return SMIRK_ERR_BAD_ARG;
    smirk_prof_next(nullptr, 54.0 * B * ((H + 1) / 2) * ((W + 1) / 2) * Cout, 4.0 * ((double)B * 3 * H * W + (double)B * ((H + 1) / 2) * ((W + 1) / 2) * Cout));
    SMIRK_LAUNCH(stem_dgrad_kernel, dim3(blocks_for((size_t)B * H * W, 16384)), dim3(256), (size_t)27 * Cout * 4, (hipStream_t)stream, (const float*)dz, w,
                 dimg, B, H, W, Cout);
    return smirk_launch_status();
}

extern "C" int smirk_gap_linear_backward_split16(const float* dout, const float* w, const float* pooled, float* dw, float* db, void* dfeat, int B, int HW,
                                                 int C, int N, void* stream) {
    if (!dout || !w || !pooled || B <= 0 || HW <= 0 || C <= 0 || C % 8 || N <= 0 || (!dw != !db)) return SMIRK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dw) SMIRK_LAUNCH(head_wgrad_kernel, dim3((unsigned)(((size_t)N * C + 255) / 256)), dim3(256), 0, st, dout, pooled, dw, db, B, C, N);
    if (dfeat) SMIRK_LAUNCH(head_dgrad_kernel, dim3(blocks_for((size_t)B * (C / 8), 16384)), dim3(256), 0, st, dout, w, (float*)dfeat, B, HW, C, N);
    return smirk_launch_status();
}
