// Training-mode building blocks of the SmirkGenerator on MI355X (BASELINE config 5, first slice: the generator is 97 % of the step's FLOPs):
//   * BatchNorm2d in TRAIN mode — batch statistics, normalise + affine (+ residual) (+ ReLU), running-stat update    (nn.BatchNorm2d inside
//     smirk_generator.py:88-119 `_block` and :121-178 `ResnetBlock` after `self.train()`, base_trainer.py:108-111) — and its backward;
//   * the backward companions of the forward kernels: weight gradient of 3x3 / 1x1 / transposed convolutions (split-fp16 x3 on the fp16 matrix pipe
//     with LDS transpose reads; exact fp32 MFMA selectable), 2x2 max-pool backward, reflection-pad fold, space-to-depth of the ConvTranspose2d output gradient, final 1x1 conv + sigmoid backward.
// Data gradients of the convolutions need no kernel of their own: dX = conv(dZ, W rotated by 180 degrees with Cin <-> Cout swapped) runs on
// the forward implicit-GEMM / ping-pong / halo-patch kernels (the host repacks the weights once per step).
//
// Activations and their gradients are split16 NHWC tensors (conv_common.h): every kernel here decodes 8-channel groups to fp32, computes in
// fp32 (reductions in fp64, two-stage and in a fixed order => bit-reproducible) and re-splits.  All kernels are HBM-bound streaming kernels
// except the weight gradient, a GEMM whose reduction index is K = B*H*W pixels, so both operands are "k-major" in memory.  That is exactly the operand
// layout of v_mfma_f32_32x32x2_f32 (one fp32 per lane: lanes 0-31 / 32-63 hold k, k+1) — the exact-fp32 kernels need no transposition but run at the fp32
// matrix rate (157 TFLOP/s); the default kernels consume the split16 operands as stored on the 16x faster fp16 pipe and let ds_read_b64_tr_b16 produce
// the k-contiguous fragments that pipe wants (wgrad_f16_kernel, wgrad3x3_halo_f16_kernel).
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

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

inline unsigned row_blocks(size_t M, int RPB) {                             // blocks of 256 / G rows, about 8 per CU at most
    const size_t b = (M + RPB - 1) / RPB;
    return (unsigned)(b > 2048 ? 2048 : (b ? b : 1));
}
#define RED_BLOCKS 512
// blocks of a two-stage column reduction over a [M][C] split16 tensor.  (Round 6 tried up to 1024 blocks chosen by tensor size, eight rows in flight per thread in the
// statistics pass and four in the apply kernels: bn_apply 3.96 -> 4.40 ms per training step, the backward sums 3.26 -> 3.69 ms, the statistics pass unchanged — the
// extra registers cost more occupancy than the loads in flight bought — and went back to this.)
inline unsigned red_blocks(size_t M, int C, int RPB) {
    (void)C;
    const size_t by_rows = (M + RPB - 1) / RPB;
    return (unsigned)(by_rows > RED_BLOCKS ? RED_BLOCKS : (by_rows ? by_rows : 1));
}
// (Round 6 swept rows in flight per thread {1, 2, 4} x stage-1 blocks {512 ... 4096} on the step's tensor shapes, profiles/r06_bn_sweep.txt: timed ALONE the three launches
// of a BatchNorm move their 3 (forward) / 5 (backward) passes at 4.9-5.9 TB/s on the U-Net's 100-400 MB tensors whatever U is, and 512 blocks is the best count at every size —
// more blocks only lengthen stage 2.  These kernels are at the HBM rate the part delivers; what is left to take out of BatchNorm is passes, not kernel tuning.)
inline unsigned blocks_for(size_t items, unsigned cap) {
    const size_t g = (items + 255) / 256;
    return (unsigned)(g > cap ? cap : (g ? g : 1));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// per-channel sums over the rows of a [M][C] split16 tensor: S1 = sum f(x), S2 = sum g(x); two stages, fp64, fixed order
//   MODE 0 (statistics)      f = z,            g = z*z
//   MODE 1 (BN backward)     f = dyh,          g = dyh * xhat      with dyh = dy * [relu ? (xhat*gamma+beta > 0) : 1], xhat = (z-mean)*invstd
// ---------------------------------------------------------------------------------------------------------------------------------
// DEV = true: device-coherent accesses (relaxed agent-scope atomics) for values that workgroups of ONE launch hand to each other.  Unused by the shipped
// three-launch BatchNorm; the one-launch cooperative form that needed it was measured slower on this 8-XCD part (a grid barrier costs 15-20 us, a dependent
// launch less: DESIGN.md 8.9, profiles/r02ai_bn_one_launch_experiment.txt) and was removed from the library in round 3 (git history: train.hip @ 9afbeda).
template <bool DEV, typename T>
__device__ __forceinline__ void st_x(T* p, T v) {
    if (DEV) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool DEV, typename T>
__device__ __forceinline__ T ld_x(const T* p) {
    if (DEV) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <int MODE, bool DEV, int U = 4>
__device__ __forceinline__ void colsum_partial(const float* __restrict__ z, const float* __restrict__ dy, size_t M, int G,
                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                               const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                               double* __restrict__ part /*[blocks][C][2]*/, double* red /*[256 * 16] in LDS*/) {
    const int tid = threadIdx.x, g = tid % G, rl = tid / G, RPB = 256 / G;          // RPB rows in flight per block iteration; when G does not
    const bool active = rl < RPB;                                                   // divide 256 the last 256 - RPB*G threads only attend the barriers
    double s1[8], s2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { s1[q] = 0.0; s2[q] = 0.0; }
    float mu[8], is[8], ga[8], be[8];
    if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { mu[q] = mean[g * 8 + q]; is[q] = invstd[g * 8 + q]; ga[q] = gamma[g * 8 + q]; be[q] = beta[g * 8 + q]; }
    }
    // four rows per iteration: all their loads are issued before the first is consumed.  One row in flight per thread (8 waves per CU x 2 KB) is
    // 16 KB per CU against ~2 us of HBM latency = the 2.1-2.6 TB/s this kernel measured; the accumulation order per thread stays row-ascending.
    const size_t S = (size_t)gridDim.x * RPB;
    for (size_t r0 = (size_t)blockIdx.x * RPB + rl; active && r0 < M; r0 += U * S) {
        float v[U][8], d[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = r0 + u * S;
            if (r < M) {
                load_group(z + (r * G + g) * 8, v[u]);
                if (MODE == 1) load_group(dy + (r * G + g) * 8, d[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r0 + u * S >= M) break;
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { s1[q] += (double)v[u][q]; s2[q] += (double)v[u][q] * (double)v[u][q]; }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float xh = (v[u][q] - mu[q]) * is[q];
                    const float dh = (relu && !(xh * ga[q] + be[q] > 0.f)) ? 0.f : d[u][q];
                    s1[q] += (double)dh; s2[q] += (double)dh * (double)xh;
                }
            }
        }
    }
    // reduce over the RPB row lanes of each channel group in a fixed order (row lane 0, 1, ...): all 16 sums of a thread go to LDS at once and
    // G * 16 threads each add up one (group, channel, which) column — one barrier, not sixteen, and no single thread walking 64 entries 8 times
    // (that serial tail, not the streaming loop, was most of this kernel: 2.1 TB/s)
    for (int q = 0; q < 8; ++q) { red[tid * 16 + q * 2] = s1[q]; red[tid * 16 + q * 2 + 1] = s2[q]; }
    __syncthreads();
    for (int o = tid; o < G * 16; o += 256) {
        const int gg = o >> 4, k = o & 15;
        double a = 0.0;
        for (int r = 0; r < RPB; ++r) a += red[(r * G + gg) * 16 + k];
        st_x<DEV>(&part[((size_t)blockIdx.x * G * 8 + gg * 8 + (k >> 1)) * 2 + (k & 1)], a);
    }
}
template <int MODE, int U = 4>
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ z, const float* __restrict__ dy, size_t M, int G,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                     double* __restrict__ part /*[blocks][C][2]*/) {
    __shared__ double red[256 * 16];
    colsum_partial<MODE, false, U>(z, dy, M, G, mean, invstd, gamma, beta, relu, part, red);
}

// stage 2: the <= 512 per-block partials of a channel are summed by 16 threads (strided, fixed order) and combined in LDS in a fixed order;
// one workgroup = 16 channels x 16 partial lanes.  (A single thread per channel walking 512 partials was 0.1 ms of pure load latency per launch.)
template <bool DEV = false>
__device__ __forceinline__ bool stage2_sum(const double* __restrict__ part, int nblocks, int C, int c, double& a, double& b, double (*red)[16][2]) {
    const int j = threadIdx.x >> 4, cl = threadIdx.x & 15;
    double sa = 0.0, sb = 0.0;
    if (c < C)
        // eight partials are loaded before the first is added (same addition order): the rolled loop waited for every load before issuing the next one —
        // 32 dependent L2 round trips, 13 us for a kernel with microseconds of work, 279 times per training step
        for (int k0 = j; k0 < nblocks; k0 += 16 * 8) {
            double va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 16 * u;
                const bool on = k < nblocks;
                va[u] = on ? ld_x<DEV>(&part[((size_t)k * C + c) * 2]) : 0.0;
                vb[u] = on ? ld_x<DEV>(&part[((size_t)k * C + c) * 2 + 1]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { sa += va[u]; sb += vb[u]; }
        }
    red[j][cl][0] = sa; red[j][cl][1] = sb;
    __syncthreads();
    if (j != 0 || c >= C) return false;
    a = 0.0; b = 0.0;
    for (int k = 0; k < 16; ++k) { a += red[k][cl][0]; b += red[k][cl][1]; }
    return true;
}
// MODE 0: mean, biased variance, invstd, running-stat update (momentum; running_var takes the unbiased variance)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ part, int nblocks, int C, double n, float eps, float momentum,
                                                          float* __restrict__ mean, float* __restrict__ var, float* __restrict__ invstd,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt) {
    __shared__ double red[16][16][2];
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;          // nn.BatchNorm2d.forward: num_batches_tracked.add_(1) (one launch less per layer)
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    double a, b;
    if (!stage2_sum(part, nblocks, C, c, a, b, red)) return;
    const double m = a / n;
    double v = b / n - m * m;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)m; var[c] = (float)v; invstd[c] = (float)(1.0 / sqrt(v + (double)eps));
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)m;
    if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(n > 1.0 ? v * n / (n - 1.0) : v);
}

// The same finalisation from the fp32 per-tile partial sums that the convolution kernels leave behind (ConvArgs::stats: [P][C][2] = (sum z, sum z^2) per partial row):
// one workgroup = 16 channels x 2 sums (one 128-byte segment of every partial row) x 32 row lanes; every thread adds its rows k*32 + rl in ascending order into
// eight interleaved fp64 chains (eight loads in flight), the chains and then the 32 row lanes are combined in a fixed order -> bit-reproducible run to run.
// P = M / 64 ... M / 128 partial rows: 98-196 for the 14 x 14 layers (one round trip), 12544 for a 112 x 112 layer at 64 frames (49 round trips of L2 hits).
__global__ __launch_bounds__(1024) void bn_finalize_partials_kernel(const float* __restrict__ part, int P, int C, double n, float eps, float momentum,
                                                                    float* __restrict__ mean, float* __restrict__ var, float* __restrict__ invstd,
                                                                    float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt) {
    __shared__ double red[32][33];
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
    const int t = threadIdx.x & 31, rl = threadIdx.x >> 5, c0 = blockIdx.x * 16;
    const bool on = c0 * 2 + t < C * 2;
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (on) {
        const float* p = part + (size_t)c0 * 2 + t;
        int k = rl;
        for (; k + 7 * 32 < P; k += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + 32 * u) * C * 2];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += (double)v[u];
        }
        for (int u = 0; k < P; k += 32, ++u) s[u] += (double)p[(size_t)k * C * 2];
    }
    red[rl][t] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (rl != 0 || !on) return;
    double a = 0.0;
    for (int k = 0; k < 32; ++k) a += red[k][t];
    const double b = __shfl_xor(a, 1, 64);                               // lanes (2c, 2c + 1) hold (sum z, sum z^2) of channel c0 + c
    if (t & 1) return;
    const int c = c0 + (t >> 1);
    const double m = a / n;
    double v = b / n - m * m;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)m; var[c] = (float)v; invstd[c] = (float)(1.0 / sqrt(v + (double)eps));
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)m;
    if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(n > 1.0 ? v * n / (n - 1.0) : v);
}

// MODE 1 / plain column sums: out1[c] = S1, out2[c] = S2 (as fp32)
__global__ __launch_bounds__(256) void colsum_stage2(const double* __restrict__ part, int nblocks, int C, float* __restrict__ out1, float* __restrict__ out2) {
    __shared__ double red[16][16][2];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    double a, b;
    if (!stage2_sum(part, nblocks, C, c, a, b, red)) return;
    if (out1) out1[c] = (float)a;
    if (out2) out2[c] = (float)b;
}

// The two BatchNorm element-wise kernels give every thread a FIXED 8-channel group and let it walk rows (256 / G rows in flight per block), so
// the per-channel coefficients are loaded once into registers and no per-element index division is needed.
// y = [relu]( (z - mean) * invstd * gamma + beta [+ residual] )
template <bool DEV = false>
__device__ __forceinline__ void bn_apply_body(const float* __restrict__ z, size_t M, int G, const float* mean,
                                              const float* invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                              const float* __restrict__ residual, int relu, float* __restrict__ y) {
    const int tid = threadIdx.x, g = tid % G, rl = tid / G, RPB = 256 / G;
    if (rl >= RPB) return;
    float mu[8], is[8], ga[8], be[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { mu[q] = ld_x<DEV>(mean + g * 8 + q); is[q] = ld_x<DEV>(invstd + g * 8 + q); ga[q] = gamma[g * 8 + q]; be[q] = beta[g * 8 + q]; }
    for (size_t r = (size_t)blockIdx.x * RPB + rl; r < M; r += (size_t)gridDim.x * RPB) {
        const size_t i = r * G + g;
        float v[8];
        load_group(z + i * 8, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (v[q] - mu[q]) * is[q] * ga[q] + be[q];
        if (residual) {
            float rr[8];
            load_group(residual + i * 8, rr);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += rr[q];
        }
        if (relu) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        store_group(y + i * 8, v);
    }
}
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, size_t M, int G, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ residual, int relu, float* __restrict__ y) {
    bn_apply_body(z, M, G, mean, invstd, gamma, beta, residual, relu, y);
}

// dz = gamma * invstd * (dyh - sum_dyh / n - xhat * sum_dyh_xhat / n)
template <bool DEV = false>
__device__ __forceinline__ void bn_backward_apply_body(const float* __restrict__ z, const float* __restrict__ dy, size_t M, int G, float inv_n,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* sum_dy, const float* sum_dy_xhat, int relu,
                                                       float* __restrict__ dz) {
    const int tid = threadIdx.x, g = tid % G, rl = tid / G, RPB = 256 / G;
    if (rl >= RPB) return;
    float mu[8], is[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = g * 8 + q;
        mu[q] = mean[c]; is[q] = invstd[c]; ga[q] = gamma[c]; be[q] = beta[c]; s1[q] = ld_x<DEV>(sum_dy + c) * inv_n; s2[q] = ld_x<DEV>(sum_dy_xhat + c) * inv_n;
    }
    for (size_t r = (size_t)blockIdx.x * RPB + rl; r < M; r += (size_t)gridDim.x * RPB) {
        const size_t i = r * G + g;
        float v[8], d[8];
        load_group(z + i * 8, v);
        load_group(dy + i * 8, d);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float xh = (v[q] - mu[q]) * is[q];
            const float dh = (relu && !(xh * ga[q] + be[q] > 0.f)) ? 0.f : d[q];
            v[q] = ga[q] * is[q] * (dh - s1[q] - xh * s2[q]);
        }
        store_group(dz + i * 8, v);
    }
}
__global__ __launch_bounds__(256) void bn_backward_apply_kernel(const float* __restrict__ z, const float* __restrict__ dy, size_t M, int G, float inv_n,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ sum_dy, const float* __restrict__ sum_dy_xhat, int relu,
                                                                float* __restrict__ dz) {
    bn_backward_apply_body(z, dy, M, G, inv_n, mean, invstd, gamma, beta, sum_dy, sum_dy_xhat, relu, dz);
}

// (Round 6 built a one-launch "finalise + apply" for tensors of <= 64 MB — element-wise workgroups owning a 32-channel column block and a row range, each reducing the
// partial rows of its own channels first — to drop the 279 tiny stage-2 launches of a training step.  Measured on the MI355X: the column-block mapping (four lanes per
// 128-byte row segment, rows C * 4 bytes apart) streams at 0.6-0.8 TB/s where the whole-row mapping below reaches 3-4.5 TB/s; the step went 41.2 -> 45.7 ms.  Removed
// again: on this part a dependent launch stays the cheap way to hand 2 KB of statistics to 2048 workgroups.  profiles/r06d_bench_train64_fin.txt)
// 2x2/2 max-pool backward: the gradient goes to the first maximum of the window in scan order (ATen's max_pool2d picks `val > max`), plus an
// optional second gradient of the same tensor (the U-Net skip connection) added in.  x, dx [B][H][W][G*8]; dy [B][H/2][W/2][G*8]
__global__ __launch_bounds__(256) void maxpool_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ add,
                                                               float* __restrict__ dx, int B, int H, int W, int G) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        size_t t = i / G;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const size_t p00 = ((((size_t)b * H + 2 * oy) * W + 2 * ox) * G + g) * 8, sx = (size_t)G * 8, sy = (size_t)W * G * 8;
        float v[4][8], gy[8], o[4][8];
        load_group(x + p00, v[0]); load_group(x + p00 + sx, v[1]); load_group(x + p00 + sy, v[2]); load_group(x + p00 + sy + sx, v[3]);
        load_group(dy + i * 8, gy);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int best = 0;
            float m = v[0][q];
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k][q] > m) { m = v[k][q]; best = k; }
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k][q] = (k == best) ? gy[q] : 0.f;
        }
        const size_t off[4] = {p00, p00 + sx, p00 + sy, p00 + sy + sx};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (add) {
                float a[8];
                load_group(add + off[k], a);
#pragma unroll
                for (int q = 0; q < 8; ++q) o[k][q] += a[q];
            }
            store_group(dx + off[k], o[k]);
        }
    }
}

// backward of ReflectionPad2d(1): dxp [B][H+2][W+2][G*8] -> dx [B][H][W][G*8] (+ optional add); padded index 0 mirrors row 1, H+1 mirrors row H-2
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ dxp, const float* __restrict__ add, float* __restrict__ dx, int B, int H,
                                                           int W, int G) {
    const size_t total = (size_t)B * H * W * G;
    const int Hp = H + 2, Wp = W + 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        size_t t = i / G;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        int ys[2] = {y + 1, -1}, xs[2] = {x + 1, -1};
        if (y == 1) ys[1] = 0;
        if (y == H - 2) ys[1] = (ys[1] < 0) ? Hp - 1 : ys[1];            // H == 3: both borders mirror into row 1 (handled below)
        if (x == 1) xs[1] = 0;
        if (x == W - 2) xs[1] = (xs[1] < 0) ? Wp - 1 : xs[1];
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        if (add) load_group(add + i * 8, acc);
        // general form (also right for H or W == 3, where a row collects both borders): walk all padded rows / cols that reflect onto (y, x)
        for (int yp = 0; yp < Hp; ++yp) {
            int ry = yp - 1; ry = ry < 0 ? -ry : ry; ry = ry >= H ? 2 * H - 2 - ry : ry;
            if (ry != y) continue;
            for (int xp = 0; xp < Wp; ++xp) {
                int rx = xp - 1; rx = rx < 0 ? -rx : rx; rx = rx >= W ? 2 * W - 2 - rx : rx;
                if (rx != x) continue;
                float v[8];
                load_group(dxp + ((((size_t)b * Hp + yp) * Wp + xp) * G + g) * 8, v);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += v[q];
            }
        }
        (void)ys; (void)xs;
        store_group(dx + i * 8, acc);
    }
}

// ConvTranspose2d(k=2, s=2) output gradient [B][2H][2W][Co] -> [B][H][W][(dy,dx,co)] : the layout in which its input gradient is a 1x1 convolution
__global__ __launch_bounds__(256) void space_to_depth_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int G) {
    const size_t total = (size_t)B * H * W * 4 * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        size_t t = i / G;
        const int q = (int)(t % 4); t /= 4;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        const float* src = in + ((((size_t)b * 2 * H + 2 * y + (q >> 1)) * 2 * W + 2 * x + (q & 1)) * G + g) * 8;
        *(f32x4*)(out + i * 8) = *(const f32x4*)src;
        *(f32x4*)(out + i * 8 + 4) = *(const f32x4*)(src + 4);
    }
}

// final 1x1 conv + sigmoid backward (smirk_generator.py:47-49,76): dl[o] = dy * y * (1 - y) -> dl8 (split16, [pixels][8], channels >= Cout zero: the
// "output gradient" operand from which the generic weight-gradient / column-sum kernels produce dW and db deterministically);
// dd[c] = sum_o dl[o] w[o][c] (split16 out)
__global__ __launch_bounds__(256) void final_backward_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ w,
                                                             float* __restrict__ dd, float* __restrict__ dl8, int B, int HW, int C, int Cout) {
    const size_t total = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, p = i % HW;
        float dl[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < Cout) {
                const float yy = y[(b * Cout + o) * HW + p];
                dl[o] = dy[(b * Cout + o) * HW + p] * yy * (1.0f - yy);
            }
        store_group(dl8 + i * 8, dl);
        for (int g = 0; g < C / 8; ++g) {
            float o8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float s = 0.f;
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < Cout) s = fmaf(dl[o], w[o * C + g * 8 + q], s);
                o8[q] = s;
            }
            store_group(dd + (i * (C / 8) + g) * 8, o8);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient:  dW[co][n] = sum over pixels p of dZ[p][co] * X[p shifted by tap(n)][ci(n)],  n = tap * Cin + ci     (exact fp32, v_mfma_f32_32x32x2_f32)
//   GEMM with M = Cout, N = KH*KH*Cin (the taps are flattened INTO N, so a 32-channel layer still fills a 128-wide tile with four taps) and
//   K = B*H*W pixels split over gridDim.z.  Workgroup tile TM x 128 with TM in {32, 64, 128} chosen from Cout; 4 waves as 2 x 2 (TM = 128: 64 x 64 per
//   wave) or 1 x 4 (TM <= 64: TM x 32 per wave).  16 pixels per chunk, decoded from split16 into fp32 [k][TM + 4] / [k][128 + 4] in a DOUBLE-BUFFERED LDS
//   image: the next chunk's global loads are issued before the current chunk's MFMAs and land in registers underneath them; one barrier per chunk.
// part[split][Cout][N] fp32 partials are summed in split order by wgrad_reduce_kernel  (bit-reproducible).
// ---------------------------------------------------------------------------------------------------------------------------------
#define WG_KC 16
#define WG_LD 132
#define WG_MAX_SPLIT 512
#define SMIRK_WGRAD_F16_DEFAULT 2
struct WgradArgs {
    const float *dz, *x;         // split16 [B][H][W][Cout], [B][H][W][Cin]
    float* part;
    int B, H, W, Cout, Cin, KH, pad, reflect;
    int chunks_per_split;        // 16-pixel chunks per K split
};

template <int TM>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    constexpr int WAVES_M = TM == 128 ? 2 : 1, WAVES_N = 4 / WAVES_M;
    constexpr int BM = TM / 32 / WAVES_M, BN = 4 / WAVES_N;                 // 32 x 32 MFMA blocks per wave
    constexpr int LDA = TM + 4;
    __shared__ __attribute__((aligned(16))) float As[2][WG_KC * LDA], Bs[2][WG_KC * WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int N = a.KH * a.KH * a.Cin;
    const int co0 = blockIdx.x * TM, n0 = blockIdx.y * 128, split = blockIdx.z;
    const long long npix = (long long)a.B * a.H * a.W;
    f32x16 acc[BM][BN];
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int j = 0; j < BN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging: thread -> (pixel k = tid / 16, 8-wide group = tid % 16) of the 16 x 128 B tile and (for tid % 16 < TM / 8) of the 16 x TM A tile
    const int sk = tid >> 4, sg = tid & 15;
    const int Gout = a.Cout / 8, Gin = a.Cin / 8;
    const int gco = co0 / 8 + sg;
    const bool a_on = sg < TM / 8 && gco < Gout;
    const int nb = n0 + sg * 8;                                               // this thread's 8 B columns: one tap, 8 consecutive input channels
    const bool b_on = nb < N;
    const int tap = b_on ? nb / a.Cin : 0, gci = b_on ? (nb % a.Cin) / 8 : 0;
    const int ky = tap / a.KH, kx = tap % a.KH;
    float va[8], vb[8];
    auto fetch = [&](int c) {
        const long long p = ((long long)split * a.chunks_per_split + c) * WG_KC + sk;
#pragma unroll
        for (int q = 0; q < 8; ++q) { va[q] = 0.f; vb[q] = 0.f; }
        if (p < npix) {
            if (a_on) load_group(a.dz + ((size_t)p * Gout + gco) * 8, va);
            if (b_on) {
                const int x0 = (int)(p % a.W), y0 = (int)((p / a.W) % a.H);
                const long long b = p / ((long long)a.W * a.H);
                int iy = y0 + ky - a.pad, ix = x0 + kx - a.pad;
                bool ok = true;
                if (a.reflect) { iy = reflect_idx(iy, a.H); ix = reflect_idx(ix, a.W); }
                else ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                if (ok) load_group(a.x + ((((size_t)b * a.H + iy) * a.W + ix) * Gin + gci) * 8, vb);
            }
        }
    };
    fetch(0);
    for (int c = 0; c < a.chunks_per_split; ++c) {
        float* as = As[c & 1];
        float* bs = Bs[c & 1];
        if (sg < TM / 8) { *(f32x4*)(as + sk * LDA + sg * 8) = *(f32x4*)va; *(f32x4*)(as + sk * LDA + sg * 8 + 4) = *(f32x4*)(va + 4); }
        *(f32x4*)(bs + sk * WG_LD + sg * 8) = *(f32x4*)vb; *(f32x4*)(bs + sk * WG_LD + sg * 8 + 4) = *(f32x4*)(vb + 4);
        __syncthreads();                       // chunk c visible; the buffer written next iteration was last read two iterations ago, behind this barrier
        if (c + 1 < a.chunks_per_split) fetch(c + 1);
        const int kk = lane >> 5, mm = lane & 31;
#pragma unroll
        for (int k2 = 0; k2 < WG_KC; k2 += 2) {
            float fa[BM], fb[BN];
#pragma unroll
            for (int i = 0; i < BM; ++i) fa[i] = as[(k2 + kk) * LDA + (wm * BM + i) * 32 + mm];
#pragma unroll
            for (int j = 0; j < BN; ++j) fb[j] = bs[(k2 + kk) * WG_LD + (wn * BN + j) * 32 + mm];
#pragma unroll
            for (int i = 0; i < BM; ++i)
#pragma unroll
                for (int j = 0; j < BN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    float* out = a.part + (size_t)split * a.Cout * N;
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int j = 0; j < BN; ++j) {
            const int n = n0 + (wn * BN + j) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm * BM + i) * 32 + mfma32_row(r, lane);
                if (co < a.Cout && n < N) out[(size_t)co * N + n] = acc[i][j][r];
            }
        }
}
// ---------------------------------------------------------------------------------------------------------------------------------
// The same GEMM on the fp16 matrix pipe: split-fp16 x3 (acc0 += dz_hi.x_hi ; acc1 += dz_hi.x_lo + dz_lo.x_hi ; dW = acc0 + acc1 * 2^-11 — fp16 x fp16 products
// are exact in fp32, only the 2^-22 lo.lo term is dropped: the arithmetic of the forward convolutions), 3 x v_mfma_f32_32x32x16_f16 per 32 x 32 x 16 block
// = 96 matrix-pipe cycles where the exact-fp32 instruction needs 512.  Both operands are ALREADY split16 in HBM, so nothing is converted; what the fp16
// instruction needs and memory does not offer is k (= pixel) contiguity per lane — 8 consecutive pixels of one channel — while memory is channel-contiguous
// per pixel.  gfx950's LDS transpose read does that re-arrangement for free: `ds_read_b64_tr_b16` lets the 16 lanes of a group fetch a [4 pixels][16 channels]
// block (lane 4r+q supplies the 8-byte address of pixel r, channels 4q..4q+3) and returns to lane c the four pixels of channel c.  Two such reads make one
// MFMA operand (pixels 8*(lane>>5) + 0..7 of channel lane&31); the pixel <-> (lane half, element) assignment is the same for both operands, which is all a
// reduction index has to satisfy.
// LDS image of a 16-pixel chunk of an operand with NG 8-channel groups, in 16-byte slots (one slot = the hi OR the lo halves of one group of one pixel):
//     slot(h, g, k) = ((h * NG/4 + g/4) * 4 + k/4) * 16  +  ((k%4 + g/4) % 4) * 4  +  g%4
// i.e. every aligned 256-byte bank row holds 4 pixels x 4 groups of one half.  A half-wave's transpose read (4 pixels x 32 channels = 4 groups) covers exactly
// one bank row -> conflict-free; the staging writes (`ds_write_b128`, 8 lanes = 8 consecutive groups of one pixel per LDS cycle) land on 8 distinct 16-byte
// bank slots because the row rotation by g/4 flips the slot's bit 2 between groups 0-3 and 4-7.
// Staging is register-based (global 16-byte loads of the raw hi / lo pieces, issued a chunk ahead, under the MFMAs), double-buffered, one barrier per SUB chunks.
// ---------------------------------------------------------------------------------------------------------------------------------
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4_t* lds_fp16x4_ptr;

__device__ __forceinline__ int wgf_slot(int ngq, int kq_per, int h, int g, int k) {      // 16-byte slot of (half h, group g, pixel k); kq_per = pixel quads per image
    return ((h * ngq + (g >> 2)) * kq_per + (k >> 2)) * 16 + ((((k & 3) + (g >> 2)) & 3) << 2) + (g & 3);
}
__device__ __forceinline__ half8 wgf_frag(const char* base, int off) {                    // two transpose reads -> one 32x32x16 MFMA operand (8 pixels of this lane's channel)
    union { fp16x4_t v[2]; half8 h; } u;
    u.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4_ptr)(base + off));
    u.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4_ptr)(base + off + 256));
    return u.h;
}

template <int TM, int SUB, bool X1 = false>   // X1: hi x hi only, one MFMA per block (BASELINE config 5's 16-bit class); its own instantiations
__global__ __launch_bounds__(256, 2) void wgrad_f16_kernel(WgradArgs a, int trmap) {
    constexpr int WAVES_M = TM == 128 ? 2 : 1, WAVES_N = 4 / WAVES_M;
    constexpr int BM = TM / 32 / WAVES_M, BN = 4 / WAVES_N;                 // 32 x 32 MFMA blocks per wave: 2x2 (TM 128), 2x1 (TM 64), 1x1 (TM 32)
    constexpr int GA = TM / 8, AQ = GA / 4, BQ = 4;                         // 8-channel groups / group quads of the A (dz) and B (x, 128 columns) tiles
    constexpr int A_BYTES = GA * 2 * 16 * 16, B_BYTES = 16 * 2 * 16 * 16;   // one 16-pixel chunk image
    __shared__ __attribute__((aligned(256))) char As[2][SUB][A_BYTES];
    __shared__ __attribute__((aligned(256))) char Bs[2][SUB][B_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int N = a.KH * a.KH * a.Cin;
    const int co0 = blockIdx.x * TM, n0 = blockIdx.y * 128, split = blockIdx.z;
    const long long npix = (long long)a.B * a.H * a.W;
    f32x16 acc0[BM][BN], acc1[BM][BN];
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int j = 0; j < BN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    // staging: thread -> (pixel sk, 8-wide group sg) of every 16-pixel chunk: the group's raw 32 bytes (hi piece, lo piece)
    const int sk = tid >> 4, sg = tid & 15;
    const int Gout = a.Cout / 8, Gin = a.Cin / 8;
    const int gco = co0 / 8 + sg;
    const bool a_on = sg < GA && gco < Gout;
    const int nb = n0 + sg * 8;
    const bool b_on = nb < N;
    const int tap = b_on ? nb / a.Cin : 0, gci = b_on ? (nb % a.Cin) / 8 : 0;
    const int ky = tap / a.KH, kx = tap % a.KH;
    const int wa_hi = wgf_slot(AQ, 4, 0, sg, sk) * 16, wa_lo = wgf_slot(AQ, 4, 1, sg, sk) * 16;
    const int wb_hi = wgf_slot(BQ, 4, 0, sg, sk) * 16, wb_lo = wgf_slot(BQ, 4, 1, sg, sk) * 16;
    const int iters = (a.chunks_per_split + SUB - 1) / SUB;
    // the pixel this thread stages in the NEXT chunk, kept decomposed (image, row, column) and advanced by 16 per chunk: no divisions in the loop
    const long long p0 = (long long)split * a.chunks_per_split * WG_KC + sk;
    int fb = (int)(p0 / ((long long)a.W * a.H));
    int fy, fx;
    { const int rem = (int)(p0 - (long long)fb * a.W * a.H); fy = rem / a.W; fx = rem - fy * a.W; }
    int fc = 0;
    // Operand fetch through buffer resources with running 32-bit offsets (both tensors are < 2 GiB here, the dispatcher checks): NHWC pixels are linear in
    // memory, so chunk c's dz piece sits 16 pixels after chunk c-1's, and (zero padding) the x pixel under this thread's tap is the linear pixel
    // p + (ky - pad) * W + (kx - pad) whenever it lies inside the image — one add per chunk instead of a 64-bit multiply chain; a lane with nothing to fetch
    // (past the split / the tensor, tap outside the image, column beyond N) carries an out-of-range offset and the load returns zeros: no branches, no zero
    // fill.  (PMC of the pointer-based first version: 8.8 VALU + 3.2 SALU instructions per MFMA, the VALU pipe as busy as the matrix pipe.)  Reflection
    // padding (three 14x14 layers) computes its source pixel explicitly.
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, (short)0, (int)(npix * a.Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)(npix * a.Cin * 4), 0x00020000);
    const unsigned OOB = 0x80000000u;
    unsigned oa = (unsigned)(((unsigned long long)p0 * Gout + gco) * 32ull);
    unsigned ob = (unsigned)(((long long)p0 + (long long)(ky - a.pad) * a.W + (kx - a.pad)) * Gin + gci) * 32u;
    const unsigned a_step = (unsigned)(WG_KC * Gout * 32), b_step = (unsigned)(WG_KC * Gin * 32);
    u32x4_t ra[SUB][2], rb[SUB][2];
    auto fetch = [&]() {
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            const bool live = fc < a.chunks_per_split && fb < a.B;
            const unsigned va = (live && a_on) ? oa : OOB;
            ra[s][0] = __builtin_amdgcn_raw_buffer_load_b128(rsa, va, 0, 0);
            ra[s][1] = __builtin_amdgcn_raw_buffer_load_b128(rsa, va, 16, 0);
            const int iy = fy + ky - a.pad, ix = fx + kx - a.pad;
            unsigned vb;
            if (a.reflect) vb = (live && b_on) ? (unsigned)((((fb * a.H + reflect_idx(iy, a.H)) * a.W + reflect_idx(ix, a.W)) * Gin + gci) * 32) : OOB;
            else vb = (live && b_on && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) ? ob : OOB;
            rb[s][0] = __builtin_amdgcn_raw_buffer_load_b128(rsb, vb, 0, 0);
            rb[s][1] = __builtin_amdgcn_raw_buffer_load_b128(rsb, vb, 16, 0);
            ++fc;
            oa += a_step; ob += b_step;
            fx += WG_KC;
            while (fx >= a.W) { fx -= a.W; if (++fy == a.H) { fy = 0; ++fb; } }
        }
    };
    // transpose-read lane geometry: 16-lane group -> [4 pixels][16 channels]; lane i of the group supplies pixel r, channel quarter q
    const int li = lane & 15, r4 = trmap ? (li & 3) : (li >> 2), q4 = trmap ? (li >> 2) : (li & 3);
    const int gl = ((lane >> 4) & 1) * 2 + (q4 >> 1), khalf = lane >> 5;    // group within the block's quad; pixels 8*khalf.. of the chunk
    int offA[BM], offB[BN];                                                  // byte offset of (half hi, first read) inside a chunk image
#pragma unroll
    for (int i = 0; i < BM; ++i) {
        const int gq = wm * BM + i;
        offA[i] = (((gq * 4 + khalf * 2) * 16) + (((r4 + gq) & 3) << 2) + gl) * 16 + (q4 & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < BN; ++j) {
        const int gq = wn * BN + j;
        offB[j] = (((gq * 4 + khalf * 2) * 16) + (((r4 + gq) & 3) << 2) + gl) * 16 + (q4 & 1) * 8;
    }
    // (Round 6 kept a SECOND iteration's fetches in flight in a second register set — 207 -> 236 VGPRs, 64 KB outstanding per workgroup instead of 32 — on the theory that
    // the kernel is bound by bytes in flight: 4.903 -> 4.907 ms per training step, i.e. nothing.  It is not latency-bound; what limits it stays open.  Removed again.)
    fetch();
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            if (sg < GA) { *(u32x4_t*)(As[buf][s] + wa_hi) = ra[s][0]; *(u32x4_t*)(As[buf][s] + wa_lo) = ra[s][1]; }
            *(u32x4_t*)(Bs[buf][s] + wb_hi) = rb[s][0]; *(u32x4_t*)(Bs[buf][s] + wb_lo) = rb[s][1];
        }
        __syncthreads();                       // this stage visible; the stage written next iteration was last read two iterations ago, behind this barrier
        if (it + 1 < iters) fetch();
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            half8 ah[BM], al[BM], bh[BN], bl[BN];
#pragma unroll
            for (int i = 0; i < BM; ++i) { ah[i] = wgf_frag(As[buf][s], offA[i]); if constexpr (!X1) al[i] = wgf_frag(As[buf][s], offA[i] + AQ * 1024); }
#pragma unroll
            for (int j = 0; j < BN; ++j) { bh[j] = wgf_frag(Bs[buf][s], offB[j]); if constexpr (!X1) bl[j] = wgf_frag(Bs[buf][s], offB[j] + BQ * 1024); }
#pragma unroll
            for (int i = 0; i < BM; ++i)
#pragma unroll
                for (int j = 0; j < BN; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                    if constexpr (!X1) {
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc1[i][j], 0, 0, 0);
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc1[i][j], 0, 0, 0);
                    }
                }
        }
    }
    float* out = a.part + (size_t)split * a.Cout * N;
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int j = 0; j < BN; ++j) {
            const int n = n0 + (wn * BN + j) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm * BM + i) * 32 + mfma32_row(r, lane);
                if (co < a.Cout && n < N) out[(size_t)co * N + n] = acc0[i][j][r] + acc1[i][j][r] * (1.0f / 2048.0f);
            }
        }
}

// Weight gradient of a 3x3 zero-padded convolution with FEW channels (the U-Net's 224^2 / 112^2 layers: Cout, Cin in {32, 64}).  There the generic kernel is
// bound by the L2 -> CU path, not by MFMA: a 32 x 128 tile re-loads 10 KB of operands for 131 kflop (13 flop/B, 36 TFLOP/s measured).  Here one workgroup owns ALL
// nine taps of its pixel chunk: the chunk is 16 consecutive pixels of one image row, x is staged ONCE as a [3 rows][18 pixels][Cin] halo and every tap reads its
// B fragment from that halo at a pixel offset — 9 KB per chunk for 295 kflop at Cin = Cout = 32 (33 flop/B).  The (Cout/32) x 9 x (Cin/32) MFMA blocks are dealt
// round-robin to the 4 waves; LDS double-buffered, next chunk's loads in flight under the MFMAs, K split over chunks like the generic kernel (same partial layout).
template <int TM, int CIN>
__global__ __launch_bounds__(256) void wgrad3x3_halo_kernel(WgradArgs a) {
    constexpr int LDA = TM + 4, LDB = CIN + 4, HPX = 3 * 18;
    constexpr int NB = (TM / 32) * 9 * (CIN / 32), MAXB = (NB + 3) / 4;
    constexpr int GA = TM / 8, GB = CIN / 8;                                   // 8-channel groups per pixel
    constexpr int NLB = (HPX * GB + 255) / 256;                               // B groups per thread and chunk
    __shared__ __attribute__((aligned(16))) float As[2][WG_KC * LDA], Bs[2][HPX * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x;
    const int cpr = a.W / 16;                                                  // chunks per image row
    const long long nchunk = (long long)a.B * a.H * cpr;
    f32x16 acc[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float va[8], vb[NLB][8];
    auto fetch = [&](int c) {
        const long long ch = (long long)split * a.chunks_per_split + c;
#pragma unroll
        for (int q = 0; q < 8; ++q) va[q] = 0.f;
#pragma unroll
        for (int u = 0; u < NLB; ++u)
#pragma unroll
            for (int q = 0; q < 8; ++q) vb[u][q] = 0.f;
        if (ch >= nchunk) return;
        const int xc = (int)(ch % cpr), y = (int)((ch / cpr) % a.H);
        const long long b = ch / ((long long)cpr * a.H);
        const int x0 = xc * 16;
        if (tid < WG_KC * GA) {                                                // dz: 16 pixels x GA groups
            const int k = tid / GA, g = tid % GA;
            load_group(a.dz + ((((size_t)b * a.H + y) * a.W + x0 + k) * GA + g) * 8, va);
        }
#pragma unroll
        for (int u = 0; u < NLB; ++u) {
            const int e = tid + u * 256;
            if (e < HPX * GB) {
                const int hp = e / GB, g = e % GB, hy = hp / 18, hx = hp % 18;
                const int iy = y - 1 + hy, ix = x0 - 1 + hx;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) load_group(a.x + ((((size_t)b * a.H + iy) * a.W + ix) * GB + g) * 8, vb[u]);
            }
        }
    };
    // this wave's MFMA blocks: blk = wave + 4 i  ->  (m block, tap, n block)
    int boffA[MAXB], boffB[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int blk = wave + 4 * i, mb = blk / (9 * (CIN / 32)), rem = blk % (9 * (CIN / 32)), tap = rem / (CIN / 32), nb = rem % (CIN / 32);
        boffA[i] = mb * 32;
        boffB[i] = ((tap / 3) * 18 + (tap % 3)) * LDB + nb * 32;               // halo pixel offset of the tap: row ky, column kx (pixel k sits at column k + 1 - 1 + kx)
    }
    fetch(0);
    for (int c = 0; c < a.chunks_per_split; ++c) {
        float* as = As[c & 1];
        float* bs = Bs[c & 1];
        if (tid < WG_KC * GA) {
            const int k = tid / GA, g = tid % GA;
            *(f32x4*)(as + k * LDA + g * 8) = *(f32x4*)va; *(f32x4*)(as + k * LDA + g * 8 + 4) = *(f32x4*)(va + 4);
        }
#pragma unroll
        for (int u = 0; u < NLB; ++u) {
            const int e = tid + u * 256;
            if (e < HPX * GB) {
                const int hp = e / GB, g = e % GB;
                *(f32x4*)(bs + hp * LDB + g * 8) = *(f32x4*)vb[u]; *(f32x4*)(bs + hp * LDB + g * 8 + 4) = *(f32x4*)(vb[u] + 4);
            }
        }
        __syncthreads();
        if (c + 1 < a.chunks_per_split) fetch(c + 1);
        const int kk = lane >> 5, mm = lane & 31;
#pragma unroll
        for (int k2 = 0; k2 < WG_KC; k2 += 2) {
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                if (wave + 4 * i < NB) {
                    const float fa = as[(k2 + kk) * LDA + boffA[i] + mm];
                    const float fb = bs[(k2 + kk) * LDB + boffB[i] + mm];      // pixel k of the chunk under tap (ky, kx) = halo (ky, k + kx)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
                }
            }
        }
    }
    const int N = 9 * CIN;
    float* out = a.part + (size_t)split * a.Cout * N;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int blk = wave + 4 * i;
        if (blk < NB) {
            const int mb = blk / (9 * (CIN / 32)), rem = blk % (9 * (CIN / 32)), tap = rem / (CIN / 32), nb = rem % (CIN / 32);
            const int n = tap * CIN + nb * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mb * 32 + mfma32_row(r, lane);
                out[(size_t)co * N + n] = acc[i][r];
            }
        }
    }
}

// The all-taps halo kernel on the fp16 matrix pipe (split-fp16 x3, LDS transpose reads; see wgrad_f16_kernel for the arithmetic and the slot layout).
// Chunk = 16 consecutive pixels of one image row.  A image: dz [16 pixels][TM channels]; B image: the x halo [3 rows x 18 pixels -> 54 halo pixels, padded to 14
// pixel quads][CIN channels].  The B fragment of tap (ky, kx) for chunk pixel k is halo pixel ky*18 + kx + k: a per-lane constant added to the pixel index, so
// the shifted fragments of all nine taps are read from the ONE staged halo (adding 4 to a pixel index moves exactly one pixel quad, hence the second transpose
// read of an operand is again +256 bytes).  The (TM/32) x 9 x (CIN/32) MFMA blocks are dealt round-robin to NW waves; a wave's A fragments are read once per chunk.
template <int TM, int CIN, int NW, int SUB, bool BUF, bool X1 = false>
__global__ __launch_bounds__(NW * 64) void wgrad3x3_halo_f16_kernel(WgradArgs a, int trmap) {
    constexpr int NT = NW * 64, HPX = 54, HQ = 14;
    constexpr int MB = TM / 32, NBQ = CIN / 32;                               // 32-channel quads of A and B
    constexpr int NB = MB * 9 * NBQ, MAXB = (NB + NW - 1) / NW;
    constexpr int GA = TM / 8, GB = CIN / 8;
    constexpr int A_BYTES = GA * 2 * 16 * 16, B_BYTES = GB * 2 * HQ * 4 * 16;
    constexpr int NLA = (16 * GA + NT - 1) / NT, NLB = (HPX * GB + NT - 1) / NT;
    __shared__ __attribute__((aligned(256))) char As[2][SUB][A_BYTES];
    __shared__ __attribute__((aligned(256))) char Bs[2][SUB][B_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x;
    const int cpr = a.W / 16;
    f32x16 acc0[MAXB], acc1[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[i][r] = 0.f; acc1[i][r] = 0.f; }
    // chunk cursor (image, row, chunk of the row) of the NEXT chunk to fetch, advanced by one per chunk
    const long long ch0 = (long long)split * a.chunks_per_split;
    int fb = (int)(ch0 / ((long long)cpr * a.H));
    int fy, fxc;
    { const int rem = (int)(ch0 - (long long)fb * cpr * a.H); fy = rem / cpr; fxc = rem - fy * cpr; }
    int fc = 0;
    // Operand fetch through buffer resources with running 32-bit offsets (see wgrad_f16_kernel): W % 16 == 0, so consecutive chunks are consecutive runs of 16
    // pixels in memory — across row ends and image ends too — and a chunk's operands sit at (first pixel of the chunk) + a per-thread constant; halo pixels
    // outside the image and idle lanes carry an out-of-range offset (zeros).
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const long long npix = (long long)a.B * a.H * a.W;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, (short)0, (int)(npix * TM * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)(npix * CIN * 4), 0x00020000);
    const unsigned OOB = 0x80000000u;
    unsigned ca[NLA];                                   // per-thread constant part of the dz offset: (pixel k of the chunk, group g)
    int cbo[NLB], chy[NLB], chx[NLB];                   // x halo: offset relative to the chunk's first pixel, halo row / column
#pragma unroll
    for (int u = 0; u < NLA; ++u) {
        const int e = tid + u * NT;
        ca[u] = e < 16 * GA ? (unsigned)(e * 32) : OOB;                        // (k * GA + g) * 32 with e = k * GA + g
    }
#pragma unroll
    for (int u = 0; u < NLB; ++u) {
        const int e = tid + u * NT, hp = e / GB, g = e % GB;
        chy[u] = e < HPX * GB ? hp / 18 : -100000;                             // idle lane: a row no image has
        chx[u] = hp % 18;
        cbo[u] = (((chy[u] - 1) * a.W + chx[u] - 1) * GB + g) * 32;
    }
    unsigned pa = (unsigned)(ch0 * 16 * GA * 32), pb = (unsigned)(ch0 * 16 * GB * 32);     // byte offset of the next chunk's first pixel in dz / x
    u32x4_t ra[SUB][NLA][2], rb[SUB][NLB][2];
    auto fetch_buf = [&]() {
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            const bool live = fc < a.chunks_per_split && fb < a.B;
            const int x0 = fxc * 16;
            // waves whose 64 items all lie past the end of the list skip the instruction (wave-uniform test): with 6-12 waves and 64-432 items most waves
            // have nothing to fetch, and an all-out-of-range load still costs its issue slot in the memory pipeline
#pragma unroll
            for (int u = 0; u < NLA; ++u) {
                if (__builtin_amdgcn_readfirstlane(wave * 64 + u * NT) < 16 * GA) {
                    const unsigned va = (live && ca[u] != OOB) ? pa + ca[u] : OOB;
                    ra[s][u][0] = __builtin_amdgcn_raw_buffer_load_b128(rsa, va, 0, 0);
                    ra[s][u][1] = __builtin_amdgcn_raw_buffer_load_b128(rsa, va, 16, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < NLB; ++u) {
                if (__builtin_amdgcn_readfirstlane(wave * 64 + u * NT) < HPX * GB) {
                    const bool ok = live && (unsigned)(fy - 1 + chy[u]) < (unsigned)a.H && (unsigned)(x0 - 1 + chx[u]) < (unsigned)a.W;
                    const unsigned vb = ok ? pb + (unsigned)cbo[u] : OOB;
                    rb[s][u][0] = __builtin_amdgcn_raw_buffer_load_b128(rsb, vb, 0, 0);
                    rb[s][u][1] = __builtin_amdgcn_raw_buffer_load_b128(rsb, vb, 16, 0);
                }
            }
            ++fc;
            pa += 16 * GA * 32; pb += 16 * GB * 32;
            if (++fxc == cpr) { fxc = 0; if (++fy == a.H) { fy = 0; ++fb; } }
        }
    };
    // pointer-based variant (64-bit addresses, exec-masked loads): measured FASTER than the buffer form for the 6- and 12-wave instantiations (64x32: 120 vs 95-103
    // TFLOP/s, 32x64: 134 vs 102-119, 64x64: 204 vs 193-202) and slower for the 3-wave 32x32 one (121 vs 145-154), same box — BUF selects per instantiation
    auto fetch_ptr = [&]() {
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
#pragma unroll
            for (int u = 0; u < NLA; ++u) ra[s][u][0] = ra[s][u][1] = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < NLB; ++u) rb[s][u][0] = rb[s][u][1] = u32x4_t{0u, 0u, 0u, 0u};
            if (fc < a.chunks_per_split && fb < a.B) {
                const int x0 = fxc * 16;
#pragma unroll
                for (int u = 0; u < NLA; ++u) {
                    const int e = tid + u * NT;
                    if (e < 16 * GA) {
                        const int k = e / GA, g = e % GA;
                        const u32x4_t* q = (const u32x4_t*)(a.dz + ((((size_t)fb * a.H + fy) * a.W + x0 + k) * GA + g) * 8);
                        ra[s][u][0] = q[0]; ra[s][u][1] = q[1];
                    }
                }
#pragma unroll
                for (int u = 0; u < NLB; ++u) {
                    const int e = tid + u * NT;
                    if (e < HPX * GB) {
                        const int hp = e / GB, g = e % GB, hy = hp / 18, hx = hp % 18;
                        const int iy = fy - 1 + hy, ix = x0 - 1 + hx;
                        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                            const u32x4_t* q = (const u32x4_t*)(a.x + ((((size_t)fb * a.H + iy) * a.W + ix) * GB + g) * 8);
                            rb[s][u][0] = q[0]; rb[s][u][1] = q[1];
                        }
                    }
                }
            }
            ++fc;
            if (++fxc == cpr) { fxc = 0; if (++fy == a.H) { fy = 0; ++fb; } }
        }
    };
    auto fetch = [&]() { if constexpr (BUF) fetch_buf(); else fetch_ptr(); };
    // transpose-read lane geometry (see wgrad_f16_kernel)
    const int li = lane & 15, r4 = trmap ? (li & 3) : (li >> 2), q4 = trmap ? (li >> 2) : (li & 3);
    const int gl = ((lane >> 4) & 1) * 2 + (q4 >> 1), khalf = lane >> 5;
    int offA[MB], offB[MAXB];
#pragma unroll
    for (int m = 0; m < MB; ++m) offA[m] = (((m * 4 + khalf * 2) * 16) + (((r4 + m) & 3) << 2) + gl) * 16 + (q4 & 1) * 8;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int blk = wave + NW * i, rem = blk % (9 * NBQ), tap = rem / NBQ, nbq = rem % NBQ;
        const int hp0 = (tap / 3) * 18 + (tap % 3) + khalf * 8 + r4;            // halo pixel of chunk pixel 8*khalf + r4 under this tap
        offB[i] = (((nbq * HQ + (hp0 >> 2)) * 16) + ((((hp0 & 3) + nbq) & 3) << 2) + gl) * 16 + (q4 & 1) * 8;
    }
    const int iters = (a.chunks_per_split + SUB - 1) / SUB;
    fetch();
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
#pragma unroll
            for (int u = 0; u < NLA; ++u) {
                const int e = tid + u * NT;
                if (e < 16 * GA) {
                    const int k = e / GA, g = e % GA;
                    *(u32x4_t*)(As[buf][s] + wgf_slot(GA / 4, 4, 0, g, k) * 16) = ra[s][u][0];
                    *(u32x4_t*)(As[buf][s] + wgf_slot(GA / 4, 4, 1, g, k) * 16) = ra[s][u][1];
                }
            }
#pragma unroll
            for (int u = 0; u < NLB; ++u) {
                const int e = tid + u * NT;
                if (e < HPX * GB) {
                    const int hp = e / GB, g = e % GB;
                    *(u32x4_t*)(Bs[buf][s] + wgf_slot(GB / 4, HQ, 0, g, hp) * 16) = rb[s][u][0];
                    *(u32x4_t*)(Bs[buf][s] + wgf_slot(GB / 4, HQ, 1, g, hp) * 16) = rb[s][u][1];
                }
            }
        }
        __syncthreads();
        if (it + 1 < iters) fetch();
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            half8 ah[MB], al[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) { ah[m] = wgf_frag(As[buf][s], offA[m]); if constexpr (!X1) al[m] = wgf_frag(As[buf][s], offA[m] + (GA / 4) * 1024); }
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int blk = wave + NW * i;
                if (blk < NB) {
                    const half8 bh = wgf_frag(Bs[buf][s], offB[i]);
                    half8 fah = ah[0];
                    if (MB == 2 && blk >= 9 * NBQ) fah = ah[MB - 1];
                    acc0[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, bh, acc0[i], 0, 0, 0);
                    if constexpr (!X1) {
                        const half8 bl = wgf_frag(Bs[buf][s], offB[i] + (GB / 4) * HQ * 256);
                        half8 fal = al[0];
                        if (MB == 2 && blk >= 9 * NBQ) fal = al[MB - 1];
                        acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, bl, acc1[i], 0, 0, 0);
                        acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal, bh, acc1[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    const int N = 9 * CIN;
    float* out = a.part + (size_t)split * a.Cout * N;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int blk = wave + NW * i;
        if (blk < NB) {
            const int mb = blk / (9 * NBQ), rem = blk % (9 * NBQ), tap = rem / NBQ, nbq = rem % NBQ;
            const int n = tap * CIN + nbq * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mb * 32 + mfma32_row(r, lane);
                out[(size_t)co * N + n] = acc0[i][r] + acc1[i][r] * (1.0f / 2048.0f);
            }
        }
    }
}

// Sum of the split-K partials, written in the layout the CALLER keeps the gradient in (no permute-copy / torch.cat afterwards):
//   layout 0  packed  dw[co][(t, ci)]                                   (the forward operand layout; what the kernels above accumulate)
//   layout 1  nn.Conv2d parameter  dw[co][cin_off + ci][t]  of a [Cout][cin_total][KH][KH] tensor, ci < cin_real (padded input channels are dropped; the two
//             sources of a decoder convolution write the two channel ranges of ONE gradient tensor)
//   layout 2  nn.ConvTranspose2d(2, 2) parameter  dw[row][co][dydx]  of a [Cin_t][Cout_t][2][2] tensor, from the packed [row][(dydx, co)] (row = input channel)
struct WgradLayout { int mode, T, Cin, cin_total, cin_off, cin_real; };
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, size_t n, float* __restrict__ dw, WgradLayout L) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // 8 independent chains keep 8 loads in flight; combined in a fixed order
        int k = 0;
        for (; k + 8 <= nsplit; k += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += part[(size_t)(k + j) * n + i];
        for (; k < nsplit; ++k) s[0] += part[(size_t)k * n + i];
        const float v = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
        if (L.mode == 0) {
            dw[i] = v;
        } else if (L.mode == 1) {
            const int N = L.T * L.Cin;
            const size_t co = i / (size_t)N;
            const int rem = (int)(i - co * N), t = rem / L.Cin, ci = rem - t * L.Cin;
            if (ci < L.cin_real) dw[((size_t)co * L.cin_total + L.cin_off + ci) * L.T + t] = v;
        } else {
            const int N = L.Cin;                                          // = 4 * Cout_t columns (dydx, co)
            const size_t row = i / (size_t)N;
            const int rem = (int)(i - row * N), ct = N / 4, dydx = rem / ct, co = rem - dydx * ct;
            dw[((size_t)row * ct + co) * 4 + dydx] = v;
        }
    }
}

// One launch per convolution and training step: nn.Conv2d weight [Cout][Cin][KH][KH] fp32 -> the two split16 operand images the step needs,
//   fwd  [Cout][(ky,kx,c)]        c < cin_pad (zero beyond Cin)                      : the forward implicit-GEMM weight
//   dgr  [cin_pad][(ky',kx',co)]  = W[co][ci][KH-1-ky'][KH-1-kx'] (zero rows beyond Cin): the data-gradient convolution's weight (180-degree rotation,
//                                                                                        Cin <-> Cout); for KH = 1 simply the transpose.
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const float* __restrict__ w, int Cout, int cin_total, int cin_off, int Cin, int KH, int cin_pad,
                                                                float* __restrict__ fwd, float* __restrict__ dgr) {
    const int T = KH * KH;
    const size_t nf = (size_t)Cout * T * cin_pad / 8, nd = (size_t)cin_pad * T * Cout / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (size_t)gridDim.x * blockDim.x) {
        float v[8];
        if (i < nf) {
            if (!fwd) continue;
            const size_t k0 = (i * 8) % ((size_t)T * cin_pad);
            const int co = (int)((i * 8) / ((size_t)T * cin_pad)), tap = (int)(k0 / cin_pad), c0 = (int)(k0 % cin_pad);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (c0 + q < Cin) ? w[((size_t)co * cin_total + cin_off + c0 + q) * T + tap] : 0.f;
            store_group(fwd + i * 8, v);
        } else {
            if (!dgr) continue;
            const size_t j = i - nf, k0 = (j * 8) % ((size_t)T * Cout);
            const int ci = (int)((j * 8) / ((size_t)T * Cout)), tap = (int)(k0 / Cout), co0 = (int)(k0 % Cout);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (ci < Cin) ? w[((size_t)(co0 + q) * cin_total + cin_off + ci) * T + (T - 1 - tap)] : 0.f;
            store_group(dgr + j * 8, v);
        }
    }
}

// every weight of a network in ONE launch: job j owns the vector range [start_j, start_{j+1}) of the flattened work list; a thread finds its job by
// binary search (<= 8 steps for 256 jobs).  A training step re-packed its 59 convolution weights with 118 launches of ~13 us each.
__global__ __launch_bounds__(256) void pack_conv_weights_batch_kernel(const SmirkPackJob* __restrict__ jobs, int njobs, unsigned long long total) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].start <= i) lo = mid; else hi = mid - 1;
        }
        const SmirkPackJob J = jobs[lo];
        const size_t li = (size_t)(i - J.start);
        if (J.KH == SMIRK_PACK_DEPTHWISE) {                        // nn.Conv2d(C, C, 3, groups=C) weight [C][1][3][3] -> fp32 [9][C] (the depthwise kernels' tap-major image)
            const int k = (int)((li * 8) / (size_t)J.Cout), c0 = (int)((li * 8) % (size_t)J.Cout);
            float* o = (float*)J.fwd + li * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = J.w[(size_t)(c0 + q) * 9 + k];
            continue;
        }
        if (J.KH == SMIRK_PACK_STEM) {                             // Conv2d(3, Cout, 3) weight [Cout][3][3][3] -> fp32 [Cout][(ky,kx,c)]; one item = one float (27 per row)
            const int co = (int)(li / 27), k = (int)(li % 27), c = k % 3, t = k / 3;
            ((float*)J.fwd)[li] = J.w[((size_t)co * 3 + c) * 9 + t];
            continue;
        }
        if (J.KH == SMIRK_PACK_CONVT2X2) {                         // nn.ConvTranspose2d(Cin_t, Cout_t, 2, 2) weight [Cin_t][Cout_t][2][2]; Cout = Cin_t, Cin = Cout_t here
            const int Ci = J.Cout, Co = J.Cin;                     //   fwd   [(dydx, co)][ci]  split16: the forward 1x1 form (out_mode CONVT2X2)
            const size_t nfw = J.fwd ? (size_t)4 * Co * Ci / 8 : 0; //   dgrad [ci][(dydx, co)]  split16: its data gradient (a 1x1 convolution over the space-to-depth gradient)
            float v[8];
            if (li < nfw) {
                const int row = (int)((li * 8) / (size_t)Ci), ci0 = (int)((li * 8) % (size_t)Ci), t = row / Co, co = row % Co;
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = J.w[((size_t)(ci0 + q) * Co + co) * 4 + t];
                store_group((float*)J.fwd + li * 8, v);
            } else {
                const size_t j = li - nfw;
                const int ci = (int)((j * 8) / (size_t)(4 * Co)), col = (int)((j * 8) % (size_t)(4 * Co)), t = col / Co, co0 = col % Co;
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = J.w[((size_t)ci * Co + co0 + q) * 4 + t];
                store_group((float*)J.dgrad + j * 8, v);
            }
            continue;
        }
        const int T = J.KH * J.KH;
        const size_t nf = J.fwd ? (size_t)J.Cout * T * J.cin_pad / 8 : 0;
        float v[8];
        if (li < nf) {
            const size_t k0 = (li * 8) % ((size_t)T * J.cin_pad);
            const int co = (int)((li * 8) / ((size_t)T * J.cin_pad)), tap = (int)(k0 / J.cin_pad), c0 = (int)(k0 % J.cin_pad);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (c0 + q < J.Cin) ? J.w[((size_t)co * J.cin_total + J.cin_off + c0 + q) * T + tap] : 0.f;
            store_group((float*)J.fwd + li * 8, v);
        } else {
            const size_t j = li - nf, k0 = (j * 8) % ((size_t)T * J.Cout);
            const int ci = (int)((j * 8) / ((size_t)T * J.Cout)), tap = (int)(k0 / J.Cout), co0 = (int)(k0 % J.Cout);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (ci < J.Cin) ? J.w[((size_t)(co0 + q) * J.cin_total + J.cin_off + ci) * T + (T - 1 - tap)] : 0.f;
            store_group((float*)J.dgrad + j * 8, v);
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------------------
extern "C" int smirk_pack_conv_weights_batch_split16(const SmirkPackJob* jobs_device, int njobs, unsigned long long total_vectors, void* stream) {
    if (!jobs_device || njobs <= 0 || njobs > 4096 || total_vectors == 0) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(pack_conv_weights_batch_kernel, dim3(blocks_for((size_t)total_vectors, 8192)), dim3(256), 0, (hipStream_t)stream, jobs_device, njobs, total_vectors);
    return smirk_launch_status();
}

extern "C" int smirk_pack_conv_weights_split16(const float* w, int Cout, int cin_total, int cin_off, int Cin, int KH, int cin_pad, void* fwd, void* dgrad,
                                               void* stream) {
    if (!w || (!fwd && !dgrad) || Cout <= 0 || Cin <= 0 || cin_off < 0 || cin_off + Cin > cin_total || Cout % 8 || cin_pad % 8 || cin_pad < Cin ||
        (KH != 1 && KH != 3))
        return SMIRK_ERR_BAD_ARG;
    const size_t n = ((size_t)Cout * KH * KH * cin_pad + (size_t)cin_pad * KH * KH * Cout) / 8;
    SMIRK_LAUNCH(pack_conv_weights_kernel, dim3(blocks_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, w, Cout, cin_total, cin_off, Cin, KH, cin_pad, (float*)fwd, (float*)dgrad);
    return smirk_launch_status();
}

extern "C" size_t smirk_train_reduce_workspace_bytes(int C) { return (size_t)RED_BLOCKS * (size_t)C * 2 * sizeof(double); }

extern "C" int smirk_bn_train_forward_split16(const void* z, size_t M, int C, const float* gamma, const float* beta, const void* residual, int relu,
                                              float eps, float momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                                              float* save_mean, float* save_var, float* save_invstd, void* y, void* ws, size_t ws_bytes, void* stream) {
    if (!z || !gamma || !beta || !save_mean || !save_var || !save_invstd || !y || !ws || M == 0 || C <= 0 || C % 8 || C / 8 > 256) return SMIRK_ERR_BAD_ARG;
    if (ws_bytes < smirk_train_reduce_workspace_bytes(C)) return SMIRK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int G = C / 8, RPB = 256 / G;
    const unsigned nb = red_blocks(M, C, RPB);
    smirk_prof_next(nullptr, 0.0, (double)M * C * 4);
    SMIRK_LAUNCH(colsum_stage1<0>, dim3(nb), dim3(256), 0, st, (const float*)z, (const float*)nullptr, M, G, (const float*)nullptr, (const float*)nullptr,
                 (const float*)nullptr, (const float*)nullptr, 0, (double*)ws);
    SMIRK_LAUNCH(bn_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, (const double*)ws, (int)nb, C, (double)M, eps, momentum, save_mean, save_var,
                 save_invstd, running_mean, running_var, num_batches_tracked);
    smirk_prof_next(nullptr, 0.0, (double)M * C * 4 * (residual ? 3 : 2));
    SMIRK_LAUNCH(bn_apply_kernel, dim3(row_blocks(M, RPB)), dim3(256), 0, st, (const float*)z, M, G, (const float*)save_mean,
                 (const float*)save_invstd, gamma, beta, (const float*)residual, relu, (float*)y);
    return smirk_launch_status();
}

/* The same BatchNorm forward when the convolution that produced z already left the per-tile partial sums of z (smirk_conv_igemm_stats_split16, rows > 0):
 * finalise from the P partial rows (fixed-order fp64) + apply — the statistics pass over z and one launch are gone. */
extern "C" int smirk_bn_train_forward_partials_split16(const void* z, size_t M, int C, const float* gamma, const float* beta, const void* residual, int relu,
                                                       float eps, float momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                                                       float* save_mean, float* save_var, float* save_invstd, void* y, const void* partials_v, int P,
                                                       int partials_fp64, void* stream) {
    if (!z || !gamma || !beta || !save_mean || !save_var || !save_invstd || !y || !partials_v || P <= 0 || M == 0 || C <= 0 || C % 8 || C / 8 > 256)
        return SMIRK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int G = C / 8, RPB = 256 / G;
    if (partials_fp64) {                                             // stage-1 rows of a reduction kernel (smirk_dwconv3x3_stats_split16): [P][C][2] doubles
        SMIRK_LAUNCH(bn_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, (const double*)partials_v, P, C, (double)M, eps, momentum, save_mean, save_var,
                     save_invstd, running_mean, running_var, num_batches_tracked);
        smirk_prof_next(nullptr, 0.0, (double)M * C * 4 * (residual ? 3 : 2));
        SMIRK_LAUNCH(bn_apply_kernel, dim3(row_blocks(M, RPB)), dim3(256), 0, st, (const float*)z, M, G, (const float*)save_mean, (const float*)save_invstd, gamma,
                     beta, (const float*)residual, relu, (float*)y);
        return smirk_launch_status();
    }
    const float* partials = (const float*)partials_v;
    SMIRK_LAUNCH(bn_finalize_partials_kernel, dim3((C + 15) / 16), dim3(1024), 0, st, partials, P, C, (double)M, eps, momentum, save_mean, save_var, save_invstd,
                 running_mean, running_var, num_batches_tracked);
    smirk_prof_next(nullptr, 0.0, (double)M * C * 4 * (residual ? 3 : 2));
    SMIRK_LAUNCH(bn_apply_kernel, dim3(row_blocks(M, RPB)), dim3(256), 0, st, (const float*)z, M, G, (const float*)save_mean,
                 (const float*)save_invstd, gamma, beta, (const float*)residual, relu, (float*)y);
    return smirk_launch_status();
}

extern "C" int smirk_bn_train_backward_split16(const void* z, const void* dy, size_t M, int C, const float* gamma, const float* beta, const float* save_mean,
                                               const float* save_invstd, int relu, void* dz, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                               void* stream) {
    if (!z || !dy || !gamma || !beta || !save_mean || !save_invstd || !dz || !dgamma || !dbeta || !ws || M == 0 || C <= 0 || C % 8 || C / 8 > 256)
        return SMIRK_ERR_BAD_ARG;
    if (ws_bytes < smirk_train_reduce_workspace_bytes(C)) return SMIRK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int G = C / 8, RPB = 256 / G;
    const unsigned nb = red_blocks(M, C, RPB);
    SMIRK_LAUNCH(colsum_stage1<1>, dim3(nb), dim3(256), 0, st, (const float*)z, (const float*)dy, M, G, save_mean, save_invstd, gamma, beta, relu, (double*)ws);
    SMIRK_LAUNCH(colsum_stage2, dim3((C + 15) / 16), dim3(256), 0, st, (const double*)ws, (int)nb, C, dbeta, dgamma);
    smirk_prof_next(nullptr, 0.0, (double)M * C * 4 * 3);
    SMIRK_LAUNCH(bn_backward_apply_kernel, dim3(row_blocks(M, RPB)), dim3(256), 0, st, (const float*)z, (const float*)dy, M, G,
                 (float)(1.0 / (double)M), save_mean, save_invstd, gamma, beta, (const float*)dbeta, (const float*)dgamma, relu, (float*)dz);
    return smirk_launch_status();
}

// ---- eval-mode BatchNorm of the differentiable generator (smirk_trainer.py:108-113: the frozen generator in .eval() inside a graph that is
//      back-propagated to its input): y = [relu]((z - running_mean) * rsqrt(running_var + eps) * gamma + beta [+ residual]); the backward is the same
//      element-wise kernel as in train mode with the two batch sums set to zero (dz = gamma * invstd * dy * relu mask).  No statistics, no reduction.
__global__ __launch_bounds__(256) void bn_invstd_kernel(const float* __restrict__ var, float eps, int C, float* __restrict__ invstd, float* __restrict__ zeros) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) { invstd[c] = 1.0f / sqrtf(var[c] + eps); zeros[c] = 0.f; }
}

extern "C" int smirk_bn_eval_forward_split16(const void* z, size_t M, int C, const float* gamma, const float* beta, const float* running_mean,
                                             const float* running_var, const void* residual, int relu, float eps, float* save_invstd, float* zeros,
                                             void* y, void* stream) {
    if (!z || !gamma || !beta || !running_mean || !running_var || !save_invstd || !zeros || !y || M == 0 || C <= 0 || C % 8 || C / 8 > 256)
        return SMIRK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int G = C / 8, RPB = 256 / G;
    SMIRK_LAUNCH(bn_invstd_kernel, dim3((C + 255) / 256), dim3(256), 0, st, running_var, eps, C, save_invstd, zeros);
    smirk_prof_next(nullptr, 0.0, (double)M * C * 4 * (residual ? 3 : 2));
    SMIRK_LAUNCH(bn_apply_kernel, dim3(row_blocks(M, RPB)), dim3(256), 0, st, (const float*)z, M, G, running_mean, (const float*)save_invstd, gamma, beta,
                 (const float*)residual, relu, (float*)y);
    return smirk_launch_status();
}

extern "C" int smirk_bn_eval_backward_split16(const void* z, const void* dy, size_t M, int C, const float* gamma, const float* beta,
                                              const float* running_mean, const float* invstd, const float* zeros, int relu, void* dz, void* stream) {
    if (!z || !dy || !gamma || !beta || !running_mean || !invstd || !zeros || !dz || M == 0 || C <= 0 || C % 8 || C / 8 > 256) return SMIRK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int G = C / 8, RPB = 256 / G;
    smirk_prof_next(nullptr, 0.0, (double)M * C * 4 * 3);
    SMIRK_LAUNCH(bn_backward_apply_kernel, dim3(row_blocks(M, RPB)), dim3(256), 0, st, (const float*)z, (const float*)dy, M, G, 0.0f, running_mean, invstd,
                 gamma, beta, zeros, zeros, relu, (float*)dz);
    return smirk_launch_status();
}

extern "C" int smirk_colsum_split16(const void* x, size_t M, int C, float* sums, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !sums || !ws || M == 0 || C <= 0 || C % 8 || C / 8 > 256) return SMIRK_ERR_BAD_ARG;
    if (ws_bytes < smirk_train_reduce_workspace_bytes(C)) return SMIRK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int G = C / 8, RPB = 256 / G;
    const unsigned nb = red_blocks(M, C, RPB);
    SMIRK_LAUNCH(colsum_stage1<0>, dim3(nb), dim3(256), 0, st, (const float*)x, (const float*)nullptr, M, G, (const float*)nullptr, (const float*)nullptr,
                 (const float*)nullptr, (const float*)nullptr, 0, (double*)ws);
    SMIRK_LAUNCH(colsum_stage2, dim3((C + 15) / 16), dim3(256), 0, st, (const double*)ws, (int)nb, C, sums, (float*)nullptr);
    return smirk_launch_status();
}

extern "C" int smirk_maxpool2x2_backward_split16(const void* x, const void* dy, const void* add, void* dx, int B, int H, int W, int C, void* stream) {
    if (!x || !dy || !dx || B <= 0 || H % 2 || W % 2 || C % 8 || C <= 0) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(maxpool_backward_kernel, dim3(blocks_for((size_t)B * (H / 2) * (W / 2) * (C / 8), 16384)), dim3(256), 0, (hipStream_t)stream,
                 (const float*)x, (const float*)dy, (const float*)add, (float*)dx, B, H, W, C / 8);
    return smirk_launch_status();
}

extern "C" int smirk_reflect_pad1_backward_split16(const void* dxp, const void* add, void* dx, int B, int H, int W, int C, void* stream) {
    if (!dxp || !dx || B <= 0 || H < 2 || W < 2 || C % 8 || C <= 0) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(reflect_fold_kernel, dim3(blocks_for((size_t)B * H * W * (C / 8), 16384)), dim3(256), 0, (hipStream_t)stream, (const float*)dxp,
                 (const float*)add, (float*)dx, B, H, W, C / 8);
    return smirk_launch_status();
}

extern "C" int smirk_space_to_depth2_split16(const void* in, void* out, int B, int H, int W, int C, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C % 8 || C <= 0) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(space_to_depth_kernel, dim3(blocks_for((size_t)B * H * W * 4 * (C / 8), 16384)), dim3(256), 0, (hipStream_t)stream, (const float*)in,
                 (float*)out, B, H, W, C / 8);
    return smirk_launch_status();
}

extern "C" int smirk_conv1x1_sigmoid_backward_split16(const float* dy, const float* y, const float* w, void* dd, void* dl8, int B, int H, int W, int C,
                                                      int Cout, void* stream) {
    if (!dy || !y || !w || !dd || !dl8 || B <= 0 || C % 8 || C <= 0 || Cout <= 0 || Cout > 4) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(final_backward_kernel, dim3(blocks_for((size_t)B * H * W, 16384)), dim3(256), 0, (hipStream_t)stream, dy, y, w, (float*)dd, (float*)dl8, B,
                 H * W, C, Cout);
    return smirk_launch_status();
}

static bool wgrad_halo_ok(int W, int Cout, int Cin, int KH, int reflect) {
    return KH == 3 && !reflect && W % 16 == 0 && (Cout == 32 || Cout == 64) && (Cin == 32 || Cin == 64);
}
// $SMIRK_WGRAD_F16: "0" = exact-fp32 MFMA kernel (wgrad_kernel), "1" / "2" = split-fp16 x3 kernel with 1 / 2 chunks per barrier (default 2);
// "+16" (17 / 18) selects the alternative lane geometry of the LDS transpose read (diagnostic)
static int g_wgrad_mode_override = -1;
static std::atomic<unsigned long long> g_wgrad_x1_fallbacks{0};
static int wgrad_f16_mode() {
    static const int mode = [] { const char* e = getenv("SMIRK_WGRAD_F16"); return e ? atoi(e) : SMIRK_WGRAD_F16_DEFAULT; }();
    return g_wgrad_mode_override >= 0 ? g_wgrad_mode_override : mode;
}
extern "C" int smirk_conv_wgrad_set_mode(int mode) {
    const int prev = wgrad_f16_mode();
    g_wgrad_mode_override = mode;
    return prev;
}
static int wgrad_nsplit(long long npix, int Cout, int N) {
    const long long chunks = (npix + WG_KC - 1) / WG_KC;
    if (N % 9 == 0 && (Cout == 32 || Cout == 64) && (N == 288 || N == 576)) {  // halo kernel (when eligible): one workgroup per split, ~6 per CU
        long long want = (Cout == 64 && N == 576) ? 1024 : 1536;               // 38 KB LDS -> 4 per CU, else 20-29 KB -> 6 per CU
        if (want > chunks) want = chunks;
        return (int)want;
    }
    const int TM = Cout <= 32 ? 32 : Cout <= 64 ? 64 : 128;
    const long long tiles = (long long)((Cout + TM - 1) / TM) * ((N + 127) / 128);
    long long want = (TM == 128 ? 1024 : 1536) / tiles;   // 4 (TM = 128: 34 KB LDS each) / 6 workgroups per CU in ONE round (rounding up put 1152 on
                                                                       // 1024 slots for the 512-channel layers: a second, 12 % full round)
    if (want > WG_MAX_SPLIT) want = WG_MAX_SPLIT;
    if (want > chunks) want = chunks;
    return (int)(want < 1 ? 1 : want);
}
extern "C" size_t smirk_conv_wgrad_workspace_bytes(int B, int H, int W, int Cout, int Cin, int KH) {
    const long long npix = (long long)B * H * W;
    return (size_t)wgrad_nsplit(npix, Cout, KH * KH * Cin) * Cout * KH * KH * Cin * 4;
}
/* dW[Cout][(ky,kx,ci)] (fp32, the packed forward layout) = sum over pixels of dz[p][co] * x[p + tap][ci];  KH in {1, 3}, pad = (KH-1)/2 */
static int conv_wgrad_impl(const void* dz, const void* x, float* dw, int B, int H, int W, int Cout, int Cin, int KH, int reflect, void* ws, size_t ws_bytes,
                           void* stream, int x1, WgradLayout L = WgradLayout{0, 0, 0, 0, 0, 0});
extern "C" int smirk_conv_wgrad_f32(const void* dz, const void* x, float* dw, int B, int H, int W, int Cout, int Cin, int KH, int reflect, void* ws, size_t ws_bytes,
                                    void* stream) {
    return conv_wgrad_impl(dz, x, dw, B, H, W, Cout, Cin, KH, reflect, ws, ws_bytes, stream, 0);
}
/* the same weight gradient with ONE MFMA per product block (hi halves of dz and x only, fp32 accumulation): BASELINE config 5's 16-bit class.  Needs the
 * f16 kernels (operands below 2 GiB, $SMIRK_WGRAD_F16 != 0): SMIRK_ERR_UNSUPPORTED otherwise — the caller falls back to smirk_conv_wgrad_f32 knowingly. */
extern "C" int smirk_conv_wgrad_f16x1(const void* dz, const void* x, float* dw, int B, int H, int W, int Cout, int Cin, int KH, int reflect, void* ws, size_t ws_bytes,
                                      void* stream) {
    return conv_wgrad_impl(dz, x, dw, B, H, W, Cout, Cin, KH, reflect, ws, ws_bytes, stream, 1);
}
/* The weight gradient summed straight into the PARAMETER's layout (WgradLayout above): layout 1 = nn.Conv2d weight [Cout][cin_total][KH][KH], channels
 * [cin_off, cin_off + cin_real) (Cin - cin_real padded operand channels are dropped); layout 2 = nn.ConvTranspose2d(2, 2) weight [Cout][Cin / 4][2][2] from the
 * 1x1 form the ConvTranspose backward uses (dz = the layer's INPUT, x = space-to-depth of the output gradient).  x1 != 0: one MFMA per product block. */
extern "C" int smirk_conv_wgrad_param(const void* dz, const void* x, float* dw_param, int B, int H, int W, int Cout, int Cin, int KH, int reflect, int layout,
                                      int cin_total, int cin_off, int cin_real, int x1, void* ws, size_t ws_bytes, void* stream) {
    if (layout == 1) {
        if (cin_real <= 0 || cin_real > Cin || cin_off < 0 || cin_off + cin_real > cin_total) return SMIRK_ERR_BAD_ARG;
    } else if (layout == 2) {
        if (KH != 1 || Cin % 4) return SMIRK_ERR_BAD_ARG;
    } else if (layout != 0) {
        return SMIRK_ERR_BAD_ARG;
    }
    const WgradLayout L{layout, KH * KH, Cin, cin_total, cin_off, cin_real};
    int rc = x1 ? conv_wgrad_impl(dz, x, dw_param, B, H, W, Cout, Cin, KH, reflect, ws, ws_bytes, stream, 1, L) : SMIRK_ERR_UNSUPPORTED;
    if (rc == SMIRK_ERR_UNSUPPORTED) {
        if (x1) g_wgrad_x1_fallbacks.fetch_add(1, std::memory_order_relaxed);     // a step that declared f16x1 ran this layer in the f32-class arithmetic: countable
        rc = conv_wgrad_impl(dz, x, dw_param, B, H, W, Cout, Cin, KH, reflect, ws, ws_bytes, stream, 0, L);
    }
    return rc;
}
/* how many smirk_conv_wgrad_param calls asked for the one-MFMA arithmetic (x1) and were served by the f32-class kernels instead (operands >= 2 GiB, or
 * $SMIRK_WGRAD_F16=0) since the library was loaded: the Python wrapper warns once when this moves, so a step never mixes arithmetics silently */
extern "C" unsigned long long smirk_conv_wgrad_x1_fallbacks(void) { return g_wgrad_x1_fallbacks.load(std::memory_order_relaxed); }
static int conv_wgrad_impl(const void* dz, const void* x, float* dw, int B, int H, int W, int Cout, int Cin, int KH, int reflect, void* ws, size_t ws_bytes,
                           void* stream, int x1, WgradLayout L) {
    if (!dz || !x || !dw || !ws || B <= 0 || H <= 0 || W <= 0 || Cout % 8 || Cin % 8 || Cout <= 0 || Cin <= 0 || (KH != 1 && KH != 3)) return SMIRK_ERR_BAD_ARG;
    if (ws_bytes < smirk_conv_wgrad_workspace_bytes(B, H, W, Cout, Cin, KH)) return SMIRK_ERR_WORKSPACE;
    const long long npix = (long long)B * H * W, chunks = (npix + WG_KC - 1) / WG_KC;
    const int N = KH * KH * Cin, nsplit = wgrad_nsplit(npix, Cout, N);
    if (wgrad_halo_ok(W, Cout, Cin, KH, reflect)) {                          // few-channel 3x3 layers: all nine taps from one staged halo
        WgradArgs h;
        h.dz = (const float*)dz; h.x = (const float*)x; h.part = (float*)ws;
        h.B = B; h.H = H; h.W = W; h.Cout = Cout; h.Cin = Cin; h.KH = 3; h.pad = 1; h.reflect = 0;
        h.chunks_per_split = (int)((chunks + nsplit - 1) / nsplit);          // W % 16 == 0: a 16-pixel chunk never crosses a row
        hipStream_t hs = (hipStream_t)stream;
        const bool fits32h = npix * Cout * 4 < (1ll << 31) && npix * Cin * 4 < (1ll << 31);     // 32-bit buffer offsets (see below)
        const int mode = fits32h ? wgrad_f16_mode() : 0;
        if (x1 && !mode) return SMIRK_ERR_UNSUPPORTED;                       // (before smirk_prof_next: a refused call leaves no pending profile label)
        smirk_prof_next(nullptr, 2.0 * (double)npix * Cout * N, 0.0);
        if (mode) {                                                          // split-fp16 x3 on the fp16 matrix pipe (LDS transpose reads)
            const int trmap = (mode >> 4) & 1;
            if (x1) {
                if (Cout == 32 && Cin == 32) SMIRK_LAUNCH((wgrad3x3_halo_f16_kernel<32, 32, 3, 2, true, true>), dim3(nsplit), dim3(192), 0, hs, h, trmap);
                else if (Cout == 64 && Cin == 32) SMIRK_LAUNCH((wgrad3x3_halo_f16_kernel<64, 32, 6, 2, false, true>), dim3(nsplit), dim3(384), 0, hs, h, trmap);
                else if (Cout == 32 && Cin == 64) SMIRK_LAUNCH((wgrad3x3_halo_f16_kernel<32, 64, 6, 1, false, true>), dim3(nsplit), dim3(384), 0, hs, h, trmap);
                else SMIRK_LAUNCH((wgrad3x3_halo_f16_kernel<64, 64, 12, 1, false, true>), dim3(nsplit), dim3(768), 0, hs, h, trmap);
            } else if (Cout == 32 && Cin == 32) SMIRK_LAUNCH((wgrad3x3_halo_f16_kernel<32, 32, 3, 2, true>), dim3(nsplit), dim3(192), 0, hs, h, trmap);
            else if (Cout == 64 && Cin == 32) SMIRK_LAUNCH((wgrad3x3_halo_f16_kernel<64, 32, 6, 2, false>), dim3(nsplit), dim3(384), 0, hs, h, trmap);
            else if (Cout == 32 && Cin == 64) SMIRK_LAUNCH((wgrad3x3_halo_f16_kernel<32, 64, 6, 1, false>), dim3(nsplit), dim3(384), 0, hs, h, trmap);
            else SMIRK_LAUNCH((wgrad3x3_halo_f16_kernel<64, 64, 12, 1, false>), dim3(nsplit), dim3(768), 0, hs, h, trmap);
        } else if (Cout == 32 && Cin == 32) SMIRK_LAUNCH((wgrad3x3_halo_kernel<32, 32>), dim3(nsplit), dim3(256), 0, hs, h);
        else if (Cout == 64 && Cin == 32) SMIRK_LAUNCH((wgrad3x3_halo_kernel<64, 32>), dim3(nsplit), dim3(256), 0, hs, h);
        else if (Cout == 32 && Cin == 64) SMIRK_LAUNCH((wgrad3x3_halo_kernel<32, 64>), dim3(nsplit), dim3(256), 0, hs, h);
        else SMIRK_LAUNCH((wgrad3x3_halo_kernel<64, 64>), dim3(nsplit), dim3(256), 0, hs, h);
        const size_t nh = (size_t)Cout * N;
        SMIRK_LAUNCH(wgrad_reduce_kernel, dim3(blocks_for(nh, 4096)), dim3(256), 0, hs, (const float*)ws, nsplit, nh, dw, L);
        return smirk_launch_status();
    }
    WgradArgs a;
    a.dz = (const float*)dz; a.x = (const float*)x; a.part = (float*)ws;
    a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.Cin = Cin; a.KH = KH; a.pad = (KH - 1) / 2; a.reflect = reflect;
    a.chunks_per_split = (int)((chunks + nsplit - 1) / nsplit);
    hipStream_t st = (hipStream_t)stream;
    const int TM = Cout <= 32 ? 32 : Cout <= 64 ? 64 : 128;
    const dim3 grid((Cout + TM - 1) / TM, (N + 127) / 128, nsplit);
    // the split-fp16 kernel fetches through buffer resources with 32-bit offsets: operands of 2 GiB and more take the exact-fp32 kernel (64-bit pointers)
    const bool fits32 = npix * Cout * 4 < (1ll << 31) && npix * Cin * 4 < (1ll << 31);
    const int mode = fits32 ? wgrad_f16_mode() : 0;
    if (x1 && !mode) return SMIRK_ERR_UNSUPPORTED;
    smirk_prof_next(nullptr, 2.0 * (double)npix * Cout * N, 0.0);
    if (mode) {                                                              // split-fp16 x3 on the fp16 matrix pipe (LDS transpose reads)
        const int trmap = (mode >> 4) & 1;
        if (x1) {                                                            // (always two chunks per barrier: the measured default)
            if (TM == 32) SMIRK_LAUNCH((wgrad_f16_kernel<32, 2, true>), grid, dim3(256), 0, st, a, trmap);
            else if (TM == 64) SMIRK_LAUNCH((wgrad_f16_kernel<64, 2, true>), grid, dim3(256), 0, st, a, trmap);
            else SMIRK_LAUNCH((wgrad_f16_kernel<128, 2, true>), grid, dim3(256), 0, st, a, trmap);
        } else if ((mode & 15) == 1) {
            if (TM == 32) SMIRK_LAUNCH((wgrad_f16_kernel<32, 1>), grid, dim3(256), 0, st, a, trmap);
            else if (TM == 64) SMIRK_LAUNCH((wgrad_f16_kernel<64, 1>), grid, dim3(256), 0, st, a, trmap);
            else SMIRK_LAUNCH((wgrad_f16_kernel<128, 1>), grid, dim3(256), 0, st, a, trmap);
        } else {
            if (TM == 32) SMIRK_LAUNCH((wgrad_f16_kernel<32, 2>), grid, dim3(256), 0, st, a, trmap);
            else if (TM == 64) SMIRK_LAUNCH((wgrad_f16_kernel<64, 2>), grid, dim3(256), 0, st, a, trmap);
            else SMIRK_LAUNCH((wgrad_f16_kernel<128, 2>), grid, dim3(256), 0, st, a, trmap);
        }
    } else if (TM == 32) SMIRK_LAUNCH(wgrad_kernel<32>, grid, dim3(256), 0, st, a);
    else if (TM == 64) SMIRK_LAUNCH(wgrad_kernel<64>, grid, dim3(256), 0, st, a);
    else SMIRK_LAUNCH(wgrad_kernel<128>, grid, dim3(256), 0, st, a);
    const size_t n = (size_t)Cout * KH * KH * Cin;
    SMIRK_LAUNCH(wgrad_reduce_kernel, dim3(blocks_for(n, 4096)), dim3(256), 0, st, (const float*)ws, nsplit, n, dw, L);
    return smirk_launch_status();
}
