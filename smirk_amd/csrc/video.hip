// Frame pre/post-processing of the reference's video loop (demo_video.py:107-214), so that a batch of decoded frames
// goes uint8 -> hot path -> uint8 without leaving HBM.  All kernels are streaming, one thread per output pixel; HBM-bound.
//
//   warp_affine_u8      skimage.transform.warp(order=1, mode='constant', cval=0, preserve_range=True).astype(uint8) of demo_video.py:124,163,
//                       203 in float64 (the library's working precision), fused with cv2.cvtColor(BGR2RGB), /255 and HWC->NCHW
//                       (demo_video.py:133-135)
//   resize_linear_u8    cv2.resize(img, (224,224)) INTER_LINEAR, 8-bit fixed-point path (demo_video.py:134)
//   u8_to_f32 / f32_to_u8   torch.Tensor(u8).permute(2,0,1).float()/255 (demo_video.py:165,169) and (grid*255).astype(uint8) + channel swap
//                       written into a column range of a wider grid = torch.cat(..., dim=3) (demo_video.py:170-172,209-214)
//   interp_bilinear     F.interpolate(x, (H, W), mode='bilinear') (demo_video.py:167,207), the ATen CUDA kernel's arithmetic
//   hull_mask           datasets/base_dataset.py:9-15 create_mask: convex hull (gift wrapping by one workgroup per frame, LDS) + row-wise fill
#include "common.h"

__device__ __forceinline__ double px_or_zero(const uint8_t* img, int rows, int cols, double r, double c, int ch) {
    if (r < 0 || r >= rows || c < 0 || c >= cols) return 0.0;
    return (double)img[((size_t)(int)r * cols + (int)c) * 3 + ch];
}

__global__ __launch_bounds__(256) void warp_affine_u8_kernel(const uint8_t* __restrict__ src, int N, int Hs, int Ws,
                                                             const double* __restrict__ mats, int Ho, int Wo, int swap_rb,
                                                             float* __restrict__ out_f32, uint8_t* __restrict__ out_u8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)Ho * Wo;
    if (i >= (size_t)N * plane) return;
    const int n = (int)(i / plane), row = (int)((i % plane) / Wo), col = (int)(i % Wo);
    const double* M = mats + (size_t)n * 6;
    const double x = M[0] * col + M[1] * row + M[2];
    const double y = M[3] * col + M[4] * row + M[5];
    const double minr = floor(y), minc = floor(x), maxr = ceil(y), maxc = ceil(x);
    const double dr = y - minr, dc = x - minc;
    const uint8_t* img = src + (size_t)n * Hs * Ws * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const double top = (1 - dc) * px_or_zero(img, Hs, Ws, minr, minc, ch) + dc * px_or_zero(img, Hs, Ws, minr, maxc, ch);
        const double bot = (1 - dc) * px_or_zero(img, Hs, Ws, maxr, minc, ch) + dc * px_or_zero(img, Hs, Ws, maxr, maxc, ch);
        const uint8_t v = (uint8_t)((1 - dr) * top + dr * bot);                  // astype(np.uint8): truncation
        const int oc = swap_rb ? 2 - ch : ch;
        if (out_f32) out_f32[((size_t)n * 3 + oc) * plane + (size_t)row * Wo + col] = (float)v / 255.0f;
        if (out_u8) out_u8[i * 3 + oc] = v;
    }
}

extern "C" int smirk_warp_affine_u8(const uint8_t* src, int N, int Hs, int Ws, const double* mats, int Ho, int Wo, int swap_rb,
                                    float* out_f32_nchw, uint8_t* out_u8_hwc, void* stream) {
    if (!src || !mats || (!out_f32_nchw && !out_u8_hwc) || N <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0) return SMIRK_ERR_BAD_ARG;
    const size_t n = (size_t)N * Ho * Wo;
    SMIRK_LAUNCH(warp_affine_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, N, Hs, Ws, mats,
                       Ho, Wo, swap_rb, out_f32_nchw, out_u8_hwc);
    return smirk_launch_status();
}

// cv2.resize INTER_LINEAR taps: fx = float((d+0.5)*scale-0.5), sx = floor, clamp; coefficients = round(w * 2048) as short
__device__ __forceinline__ void cv_linear_tap(int d, int ssize, int dsize, int* s0, int* s1, int* a0, int* a1) {
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
    *s0 = sx; *s1 = min(sx + 1, ssize - 1);
    *a0 = (int)rintf((1.f - fx) * 2048.f); *a1 = (int)rintf(fx * 2048.f);
}

__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* __restrict__ src, int N, int Hs, int Ws, int Ho, int Wo,
                                                               int swap_rb, float* __restrict__ out_f32, uint8_t* __restrict__ out_u8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)Ho * Wo;
    if (i >= (size_t)N * plane) return;
    const int n = (int)(i / plane), row = (int)((i % plane) / Wo), col = (int)(i % Wo);
    int x0, x1, xa0, xa1, y0, y1, ya0, ya1;
    cv_linear_tap(col, Ws, Wo, &x0, &x1, &xa0, &xa1);
    cv_linear_tap(row, Hs, Ho, &y0, &y1, &ya0, &ya1);
    const uint8_t* img = src + (size_t)n * Hs * Ws * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        uint8_t v;
        if (Hs == Ho && Ws == Wo) v = img[((size_t)row * Ws + col) * 3 + ch];
        else {
            const int r0 = img[((size_t)y0 * Ws + x0) * 3 + ch] * xa0 + img[((size_t)y0 * Ws + x1) * 3 + ch] * xa1;
            const int r1 = img[((size_t)y1 * Ws + x0) * 3 + ch] * xa0 + img[((size_t)y1 * Ws + x1) * 3 + ch] * xa1;
            const int o = (((ya0 * (r0 >> 4)) >> 16) + ((ya1 * (r1 >> 4)) >> 16) + 2) >> 2;
            v = (uint8_t)min(max(o, 0), 255);
        }
        const int oc = swap_rb ? 2 - ch : ch;
        if (out_f32) out_f32[((size_t)n * 3 + oc) * plane + (size_t)row * Wo + col] = (float)v / 255.0f;
        if (out_u8) out_u8[i * 3 + oc] = v;
    }
}

extern "C" int smirk_resize_linear_u8(const uint8_t* src, int N, int Hs, int Ws, int Ho, int Wo, int swap_rb, float* out_f32_nchw,
                                      uint8_t* out_u8_hwc, void* stream) {
    if (!src || (!out_f32_nchw && !out_u8_hwc) || N <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0) return SMIRK_ERR_BAD_ARG;
    const size_t n = (size_t)N * Ho * Wo;
    SMIRK_LAUNCH(resize_linear_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, N, Hs, Ws, Ho,
                       Wo, swap_rb, out_f32_nchw, out_u8_hwc);
    return smirk_launch_status();
}

__global__ __launch_bounds__(256) void f32_to_u8_grid_kernel(const float* __restrict__ src, int N, int H, int W, int swap_rb,
                                                             uint8_t* __restrict__ dst, int grid_w, int col0) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)H * W;
    if (i >= (size_t)N * plane) return;
    const int n = (int)(i / plane), row = (int)((i % plane) / W), col = (int)(i % W);
    uint8_t* o = dst + (((size_t)n * H + row) * grid_w + col0 + col) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float v = src[((size_t)n * 3 + ch) * plane + (size_t)row * W + col] * 255.0f;
        o[swap_rb ? 2 - ch : ch] = (uint8_t)(int)v;                      // astype(np.uint8) of a value in [0, 255]
    }
}

extern "C" int smirk_f32_nchw_to_u8_grid(const float* src, int N, int H, int W, int swap_rb, uint8_t* grid, int grid_w, int col0,
                                         void* stream) {
    if (!src || !grid || N <= 0 || H <= 0 || W <= 0 || col0 < 0 || col0 + W > grid_w) return SMIRK_ERR_BAD_ARG;
    const size_t n = (size_t)N * H * W;
    SMIRK_LAUNCH(f32_to_u8_grid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, N, H, W, swap_rb,
                       grid, grid_w, col0);
    return smirk_launch_status();
}

__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* __restrict__ src, int N, int H, int W, int swap_rb,
                                                        float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)H * W;
    if (i >= (size_t)N * plane) return;
    const int n = (int)(i / plane);
    const size_t p = i % plane;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) dst[((size_t)n * 3 + (swap_rb ? 2 - ch : ch)) * plane + p] = (float)src[i * 3 + ch] / 255.0f;
}

extern "C" int smirk_u8_hwc_to_f32_nchw(const uint8_t* src, int N, int H, int W, int swap_rb, float* dst, void* stream) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0) return SMIRK_ERR_BAD_ARG;
    const size_t n = (size_t)N * H * W;
    SMIRK_LAUNCH(u8_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, N, H, W, swap_rb, dst);
    return smirk_launch_status();
}

// ATen upsample_bilinear2d (CUDA kernel arithmetic), align_corners = False
__global__ __launch_bounds__(256) void interp_bilinear_kernel(const float* __restrict__ src, int NC, int H, int W, int Ho, int Wo,
                                                              float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)Ho * Wo;
    if (i >= (size_t)NC * plane) return;
    const int nc = (int)(i / plane), row = (int)((i % plane) / Wo), col = (int)(i % Wo);
    const float rh = (float)H / (float)Ho, rw = (float)W / (float)Wo;
    const float h1r = fmaxf(rh * ((float)row + 0.5f) - 0.5f, 0.f), w1r = fmaxf(rw * ((float)col + 0.5f) - 0.5f, 0.f);
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = (h1 < H - 1) ? 1 : 0, w1p = (w1 < W - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l, w1l = w1r - (float)w1, w0l = 1.f - w1l;
    const float* s = src + (size_t)nc * H * W;
    const float top = w0l * s[(size_t)h1 * W + w1] + w1l * s[(size_t)h1 * W + w1 + w1p];
    const float bot = w0l * s[(size_t)(h1 + h1p) * W + w1] + w1l * s[(size_t)(h1 + h1p) * W + w1 + w1p];
    dst[i] = h0l * top + h1l * bot;
}

extern "C" int smirk_interp_bilinear_f32(const float* src, int NC, int H, int W, int Ho, int Wo, float* dst, void* stream) {
    if (!src || !dst || NC <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return SMIRK_ERR_BAD_ARG;
    const size_t n = (size_t)NC * Ho * Wo;
    SMIRK_LAUNCH(interp_bilinear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, NC, H, W, Ho, Wo,
                       dst);
    return smirk_launch_status();
}

// ---- create_mask: convex hull of the (truncated-to-int) landmarks, filled with 0 on a field of 1 ---------------------------------------
#define HULL_MAX_PTS 1024
__global__ __launch_bounds__(256) void hull_mask_kernel(const float* __restrict__ lmk, int L, int stride, int H, int W,
                                                        float* __restrict__ out) {
    __shared__ int px[HULL_MAX_PTS], py[HULL_MAX_PTS];
    __shared__ int hx[HULL_MAX_PTS], hy[HULL_MAX_PTS];
    __shared__ long long red[256];
    __shared__ int s_n, s_cur;
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* lp = lmk + (size_t)n * L * stride;
    for (int i = tid; i < L; i += 256) { px[i] = (int)lp[i * stride]; py[i] = (int)lp[i * stride + 1]; }     // astype(np.int32)
    __syncthreads();
    // start: lexicographically smallest (x, y) — certainly a hull vertex
    long long best = 0x7fffffffffffffffll;
    for (int i = tid; i < L; i += 256) {
        const long long key = (((long long)px[i] + (1ll << 30)) << 32) | (unsigned)(py[i] + (1 << 30));
        best = key < best ? key : best;
    }
    red[tid] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = red[tid + o] < red[tid] ? red[tid + o] : red[tid]; __syncthreads(); }
    if (tid == 0) { hx[0] = (int)(red[0] >> 32) - (1 << 30); hy[0] = (int)(unsigned)(red[0] & 0xffffffffll) - (1 << 30); s_n = 1; }
    __syncthreads();
    // gift wrapping: next vertex = the point q such that no point lies strictly to the right of cur->q; among collinear candidates the farthest
    for (int it = 0; it < L; ++it) {
        const int cx = hx[s_n - 1], cy = hy[s_n - 1];
        int q = -1;
        for (int i = tid; i < L; i += 256) {
            if (px[i] == cx && py[i] == cy) continue;
            if (q < 0) { q = i; continue; }
            const long long cr = (long long)(px[q] - cx) * (py[i] - cy) - (long long)(py[q] - cy) * (px[i] - cx);
            if (cr < 0) q = i;
            else if (cr == 0) {
                const long long dq = (long long)(px[q] - cx) * (px[q] - cx) + (long long)(py[q] - cy) * (py[q] - cy);
                const long long di = (long long)(px[i] - cx) * (px[i] - cx) + (long long)(py[i] - cy) * (py[i] - cy);
                if (di > dq) q = i;
            }
        }
        red[tid] = q;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                const int a = (int)red[tid], b = (int)red[tid + o];
                int w = a;
                if (a < 0) w = b;
                else if (b >= 0) {
                    const long long cr = (long long)(px[a] - cx) * (py[b] - cy) - (long long)(py[a] - cy) * (px[b] - cx);
                    if (cr < 0) w = b;
                    else if (cr == 0) {
                        const long long da = (long long)(px[a] - cx) * (px[a] - cx) + (long long)(py[a] - cy) * (py[a] - cy);
                        const long long db = (long long)(px[b] - cx) * (px[b] - cx) + (long long)(py[b] - cy) * (py[b] - cy);
                        if (db > da) w = b;
                    }
                }
                red[tid] = w;
            }
            __syncthreads();
        }
        if (tid == 0) {
            const int w = (int)red[0];
            s_cur = 0;
            if (w >= 0 && !(px[w] == hx[0] && py[w] == hy[0])) { hx[s_n] = px[w]; hy[s_n] = py[w]; s_n = s_n + 1; s_cur = 1; }
        }
        __syncthreads();
        if (!s_cur) break;
    }
    const int hn = s_n;
    int ymin = hy[0], ymax = hy[0];
    for (int i = 1; i < hn; ++i) { ymin = min(ymin, hy[i]); ymax = max(ymax, hy[i]); }
    float* o = out + (size_t)n * H * W;
    for (int y = tid; y < H; y += 256) {
        int a = 1, b = 0;
        if (y >= ymin && y <= ymax) {
            double xl = 1e300, xr = -1e300;
            for (int i = 0; i < hn; ++i) {
                const int x0 = hx[i], y0 = hy[i], x1 = hx[(i + 1) % hn], y1 = hy[(i + 1) % hn];
                if (min(y0, y1) <= y && y <= max(y0, y1)) {
                    if (y0 == y1) { xl = fmin(xl, (double)min(x0, x1)); xr = fmax(xr, (double)max(x0, x1)); }
                    else { const double xx = x0 + (double)(x1 - x0) * (double)(y - y0) / (double)(y1 - y0); xl = fmin(xl, xx); xr = fmax(xr, xx); }
                }
            }
            if (hn == 1) { xl = xr = hx[0]; }
            a = max((int)floor(xl + 0.5), 0); b = min((int)floor(xr + 0.5), W - 1);
        }
        for (int x = 0; x < W; ++x) o[(size_t)y * W + x] = (x >= a && x <= b) ? 0.f : 1.f;
    }
}

extern "C" int smirk_hull_mask(const float* landmarks, int N, int L, int stride, int H, int W, float* out, void* stream) {
    if (!landmarks || !out || N <= 0 || L <= 0 || L > HULL_MAX_PTS || stride < 2 || H <= 0 || W <= 0) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(hull_mask_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, landmarks, L, stride, H, W, out);
    return smirk_launch_status();
}
