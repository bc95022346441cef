// Mesh-based masking utilities for MI355X (gfx950) — the step between Renderer and SmirkGenerator
// (src/utils/masking.py:39-181; callers demo.py:146-167, smirk_trainer.py:76-93,268-293).  All HBM/latency-bound integer + fp32 work:
//   mask_face_weights   per (image, FLAME triangle): mean z of its 3 vertex normals, visibility gate (< 0.05), xy shoelace area,
//                       region probability  -> sampling weight                                  (masking.py:144-160)
//   sample_faces        one workgroup per image: LDS prefix sum of the 9976 weights (the CDF), then every lane draws with a counter-
//                       based Philox4x32-10 generator and binary-searches the CDF (multinomial with replacement) and draws the
//                       folded-uniform barycentrics                                            (masking.py:51-68,163-166)
//   points_to_pixels    .5*(1+p)*S -> int64 (truncation) -> clamp x,y                          (masking.py:172-175)
//   maxpool_sq          (2r+1)^2 stride-1 max pool with -inf padding as two separable passes   (masking.py:78,96)
//   bernoulli_field / masking_compose / transfer_pixels                                        (masking.py:71-102,116-129)
// Random draws are reproducible from (seed, call offset) but are NOT torch's CPU/CUDA streams — the reference itself draws
// different numbers on CPU and GPU; deterministic entry points (coords= / explicit fields) are what the parity tests pin.
#include "common.h"

// ---- Philox4x32-10 (Salmon et al. 2011), counter = (idx, stream, 0, 0), key = seed ------------------------------------------------
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   // [0,1), 24 bits like torch.rand

__global__ __launch_bounds__(256) void mask_face_weights_kernel(const float* __restrict__ tv, const float* __restrict__ normals,
                                                                const int32_t* __restrict__ faces, const float* __restrict__ prob,
                                                                int B, int V, int F, float* __restrict__ w) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * F) return;
    const int b = (int)(i / F), f = (int)(i % F);
    const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    const float* nb = normals + (size_t)b * V * 3;
    const float* vb = tv + (size_t)b * V * 3;
    const float nz = ((nb[i0 * 3 + 2] + nb[i1 * 3 + 2]) + nb[i2 * 3 + 2]) / 3.0f;
    const float x1 = vb[i0 * 3], y1 = vb[i0 * 3 + 1], x2 = vb[i1 * 3], y2 = vb[i1 * 3 + 1], x3 = vb[i2 * 3], y3 = vb[i2 * 3 + 1];
    const float area = 0.5f * fabsf(x1 * y2 + x2 * y3 + x3 * y1 - x2 * y1 - x3 * y2 - x1 * y3);
    const float p = (nz < 0.05f) ? prob[f] : 0.0f;
    w[i] = p * area;
}

__global__ __launch_bounds__(1024) void sample_faces_kernel(const float* __restrict__ w, int F, int num, uint64_t seed,
                                                            uint64_t offset, int32_t* __restrict__ idx, float* __restrict__ bary) {
    extern __shared__ float cdf[];               // [F] inclusive prefix sums, + 32 wave totals
    float* wtot = cdf + F;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const float* wb = w + (size_t)b * F;
    // block scan in chunks of blockDim: sequential carry keeps the summation order fixed (deterministic)
    float carry = 0.f;
    for (int base = 0; base < F; base += blockDim.x) {
        const int i = base + tid;
        float v = (i < F) ? wb[i] : 0.f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
        if (lane == 63) wtot[wave] = v;
        __syncthreads();
        float pre = carry;
        for (int k = 0; k < wave; ++k) pre += wtot[k];
        if (i < F) cdf[i] = v + pre;
        float all = carry;
        for (int k = 0; k < nw; ++k) all += wtot[k];
        __syncthreads();
        carry = all;
    }
    const float total = carry;
    for (int t = tid; t < num; t += blockDim.x) {
        uint32_t r[4];
        const uint64_t ctr = offset + (uint64_t)b * num + t;
        philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        const float target = u01(r[0]) * total;
        int lo = 0, hi = F - 1;                  // first index whose cdf exceeds target (zero-weight faces are never chosen)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] > target) hi = mid; else lo = mid + 1; }
        idx[(size_t)b * num + t] = lo;
        float u = u01(r[1]), v = u01(r[2]);
        if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }                       // masking.py:58-60
        float* o = bary + ((size_t)b * num + t) * 3;
        o[0] = 1.0f - (u + v); o[1] = u; o[2] = v;                             // masking.py:63-68
    }
}

__global__ __launch_bounds__(256) void points_to_pixels_kernel(const float* __restrict__ p, size_t n, int S, long long* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = 0.5f * (1.0f + p[i * 3 + c]) * (float)S;               // masking.py:172
        long long q = (long long)v;                                             // .long(): truncation toward zero
        if (c < 2) q = q < 0 ? 0 : (q > S - 1 ? S - 1 : q);                     // masking.py:174-175
        out[i * 3 + c] = q;
    }
}

// one pass of a separable (2r+1) max filter with -inf padding; dir 0 = along x, 1 = along y.  in/out [B][H][W]
// Generic fallback (any radius / width); the shipped sizes take maxpool_sq_lds_kernel below.  vectorize(disable): the loop vectoriser turned
// the complement + max chain into v_pk_add_f32, an instruction class that must not appear in this library (build.py).
__global__ __launch_bounds__(256) void maxfilter1d_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                                          int r, int dir, int complement_in, int complement_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const size_t base = i - (size_t)y * W - x;
    float m = -INFINITY;
    if (dir == 0) {
        const int lo = max(x - r, 0), hi = min(x + r, W - 1);
#pragma clang loop vectorize(disable) interleave(disable)
        for (int k = lo; k <= hi; ++k) { float v = in[base + (size_t)y * W + k]; if (complement_in) v = 1.0f - v; m = fmaxf(m, v); }
    } else {
        const int lo = max(y - r, 0), hi = min(y + r, H - 1);
#pragma clang loop vectorize(disable) interleave(disable)
        for (int k = lo; k <= hi; ++k) { float v = in[base + (size_t)k * W + x]; if (complement_in) v = 1.0f - v; m = fmaxf(m, v); }
    }
    out[i] = complement_out ? 1.0f - m : m;
}

// (2R+1)^2 max pool, both passes in ONE launch: a workgroup owns MP_ROWS output rows of one image over the full width.  It stages the
// rows [y0-R, y0+MP_ROWS+R) once in LDS (complemented on the way in if asked; -inf outside the image and in an R-wide margin left and
// right), runs the x pass LDS -> LDS with every thread producing 4 neighbouring outputs from 4+2R staged values (6 LDS reads per output
// instead of 21 global ones), then the y pass the same way down a column (lanes are neighbouring x: conflict-free) and writes the result.
// max is exact and order-free, so the output equals the two-pass form bit for bit (tests/test_masking_gpu.py).
#define MP_ROWS 32
__host__ __device__ static inline int mp_row_stride(int W, int R) { return (W + 3) / 4 * 4 + (2 * R + 3) / 4 * 4 + 4; }
template <int R>
__global__ __launch_bounds__(512) void maxpool_sq_lds_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                            int complement_in, int complement_out) {
    extern __shared__ __attribute__((aligned(16))) float mp_lds[];
    // staged row stride: the W + 2R columns (margins included) rounded up so that every thread's 4 + 2R window, read as whole 16-byte vectors, stays inside its
    // row.  The x pass reads the window of a lane with ds_read_b128 (lane stride 16 bytes: conflict-free); as scalar reads at a lane stride of 4 floats every
    // access was a 4-way bank conflict - 58-64 % of this kernel's LDS cycles (profiles/r03v_pmc_census_full.txt).
    const int SW = mp_row_stride(W, R);
    const int NR = MP_ROWS + 2 * R;                     // staged rows
    float* A = mp_lds;                                  // [NR][SW]  input rows
    float* Hx = mp_lds + NR * SW;                       // [NR][W]   after the x pass
    const int tiles = (H + MP_ROWS - 1) / MP_ROWS;
    const int b = blockIdx.x / tiles, y0 = (blockIdx.x % tiles) * MP_ROWS;
    const float* src = in + (size_t)b * H * W;
    // eight requests in flight per lane and trip, clamped addresses + select: a rolled load -> store loop pays one dependent global round trip per iteration (25 of them)
    for (int base = 0; base < NR * SW; base += 512 * 8) {
        float sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + 512 * u + (int)threadIdx.x, rr = i / SW, cx = i - rr * SW - R, y = y0 - R + rr;
            const bool in = i < NR * SW && y >= 0 && y < H && cx >= 0 && cx < W;
            const float t = src[(size_t)min(max(y, 0), H - 1) * W + min(max(cx, 0), W - 1)];
            sv[u] = in ? (complement_in ? 1.0f - t : t) : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + 512 * u + (int)threadIdx.x;
            if (i < NR * SW) A[i] = sv[u];
        }
    }
    __syncthreads();
    const int W4 = (W + 3) / 4;
    for (int i = threadIdx.x; i < NR * W4; i += 512) {
        const int rr = i / W4, x = (i - rr * W4) * 4;
        const f32x4* a4 = (const f32x4*)(A + rr * SW + x);   // a[k] = column x - R + k; 16-byte aligned: SW and x are multiples of 4
        constexpr int NV = (4 + 2 * R + 3) / 4;
        float v[4 * NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const f32x4 t = a4[j];
            v[4 * j] = t[0]; v[4 * j + 1] = t[1]; v[4 * j + 2] = t[2]; v[4 * j + 3] = t[3];
        }
        float c = v[3];
#pragma unroll
        for (int k = 4; k <= 2 * R; ++k) c = fmaxf(c, v[k]);
        const float o0 = fmaxf(fmaxf(c, v[2]), fmaxf(v[1], v[0]));
        const float o1 = fmaxf(fmaxf(c, v[2]), fmaxf(v[1], v[2 * R + 1]));
        const float o2 = fmaxf(fmaxf(c, v[2]), fmaxf(v[2 * R + 2], v[2 * R + 1]));
        const float o3 = fmaxf(fmaxf(c, v[2 * R + 3]), fmaxf(v[2 * R + 2], v[2 * R + 1]));
        float* h = Hx + rr * W + x;
        if ((W & 3) == 0) {                             // uniform: rows of Hx are 16-byte aligned
            const f32x4 o = {o0, o1, o2, o3};
            *(f32x4*)h = o;
        } else {
            h[0] = o0;
            if (x + 1 < W) h[1] = o1;
            if (x + 2 < W) h[2] = o2;
            if (x + 3 < W) h[3] = o3;
        }
    }
    __syncthreads();
    float* dst = out + (size_t)b * H * W;
    for (int i = threadIdx.x; i < (MP_ROWS / 4) * W; i += 512) {
        const int q = i / W, x = i - q * W;
        const float* h = Hx + (q * 4) * W + x;        // h[k * W] = staged row q*4 + k = image row y0 + q*4 - R + k
        float v[4 + 2 * R];
#pragma unroll
        for (int k = 0; k < 4 + 2 * R; ++k) v[k] = h[k * W];
        float c = v[3];
#pragma unroll
        for (int k = 4; k <= 2 * R; ++k) c = fmaxf(c, v[k]);
        float o[4];
        o[0] = fmaxf(fmaxf(c, v[2]), fmaxf(v[1], v[0]));
        o[1] = fmaxf(fmaxf(c, v[2]), fmaxf(v[1], v[2 * R + 1]));
        o[2] = fmaxf(fmaxf(c, v[2]), fmaxf(v[2 * R + 2], v[2 * R + 1]));
        o[3] = fmaxf(fmaxf(c, v[2 * R + 3]), fmaxf(v[2 * R + 2], v[2 * R + 1]));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = y0 + q * 4 + k;
            if (y < H) dst[(size_t)y * W + x] = complement_out ? 1.0f - o[k] : o[k];
        }
    }
}

__global__ __launch_bounds__(256) void bernoulli_field_kernel(float* __restrict__ out, size_t n, float p, uint64_t seed, uint64_t offset) {
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    uint32_t r[4];
    const uint64_t ctr = offset + i4 / 4;
    philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), 1u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i4 + k < n) out[i4 + k] = u01(r[k]) < p ? 1.0f : 0.0f;
}

// masked = img * mask * (1 - rendered_mask); extra = extra_points * (noise ? N(1, 0.05) : 1) * keep; out = extra > 0 ? extra : masked
// img / extra / out [B][C][H][W]; mask, rendered_mask, keep [B][1][H][W] (rendered_mask / keep nullable)
__global__ __launch_bounds__(256) void masking_compose_kernel(const float* __restrict__ img, const float* __restrict__ mask,
                                                              const float* __restrict__ rendered_mask,
                                                              const float* __restrict__ extra, const float* __restrict__ pmask,
                                                              const float* __restrict__ keep,
                                                              const float* __restrict__ noise_mult, int B, int C, int HW,
                                                              int gen_noise, uint64_t seed, uint64_t offset,
                                                              float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * C * HW) return;
    const size_t b = i / ((size_t)C * HW), p = i % HW;
    float m = mask[b * HW + p];
    if (rendered_mask) m = m * (1.0f - rendered_mask[b * HW + p]);
    const float masked = img[i] * m;
    float e = extra ? extra[i] : img[i] * pmask[b * HW + p];                                 // demo.py:163 extra_points = image * pmask
    if (noise_mult) e = e * noise_mult[i];
    else if (gen_noise) {
        uint32_t r[4];
        const uint64_t ctr = offset + i;
        philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), 2u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        const float u1 = fmaxf(u01(r[0]), 5.96e-8f), u2 = u01(r[1]);
        const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);      // Box-Muller
        e = e * (z * 0.05f + 1.0f);                                                          // masking.py:91-92
    }
    if (keep) e = e * keep[b * HW + p];
    out[i] = (e > 0.0f) ? e : masked;                                                        // masking.py:101
}

// retained[b,:,p2y,p2x] = img[b,:,p1y,p1x] for l < rbound[b]; duplicates in points2: the highest l wins (CPU index_put order)
__global__ __launch_bounds__(256) void transfer_winner_kernel(const long long* __restrict__ p2, const long long* __restrict__ rbound,
                                                              int B, int L, int H, int W, int* __restrict__ winner) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * L) return;
    const int b = (int)(i / L), l = (int)(i % L);
    if (rbound && l >= rbound[b]) return;
    const long long x = p2[i * 3 + 0], y = p2[i * 3 + 1];
    if (x < 0 || x >= W || y < 0 || y >= H) return;
    atomicMax(&winner[((size_t)b * H + y) * W + x], l);
}
__global__ __launch_bounds__(256) void transfer_gather_kernel(const float* __restrict__ img, const long long* __restrict__ p1,
                                                              const int* __restrict__ winner, int B, int C, int L, int H, int W,
                                                              float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * C * H * W) return;
    const size_t HW = (size_t)H * W, b = i / (C * HW), c = (i / HW) % C, p = i % HW;
    const int l = winner[b * HW + p];
    float v = 0.f;
    if (l >= 0) {
        long long x = p1[((size_t)b * L + l) * 3 + 0], y = p1[((size_t)b * L + l) * 3 + 1];
        x = x < 0 ? x + W : x; y = y < 0 ? y + H : y;                    // python negative indexing never occurs after the clamps; kept for safety
        v = img[(b * C + c) * HW + (size_t)y * W + x];
    }
    out[i] = v;
}

// pmask[b,0,y,x] = 1 at the first rbound[b] sampled points (demo.py:153-159, smirk_trainer.py:86-88); out must be zero-filled
__global__ __launch_bounds__(256) void scatter_points_kernel(const long long* __restrict__ pts, const long long* __restrict__ rbound,
                                                             int B, int L, int H, int W, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * L) return;
    const int b = (int)(i / L), l = (int)(i % L);
    if (rbound && l >= rbound[b]) return;
    const long long x = pts[i * 3], y = pts[i * 3 + 1];
    if (x >= 0 && x < W && y >= 0 && y < H) out[((size_t)b * H + y) * W + x] = 1.0f;
}
// rendered_mask = 1 - (rendered_img == 0).all(dim=1)   (demo.py:146, smirk_trainer.py:79)
__global__ __launch_bounds__(256) void rendered_mask_kernel(const float* __restrict__ img, int B, int C, int HW, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const size_t b = i / HW, p = i % HW;
    bool all0 = true;
    for (int c = 0; c < C; ++c) all0 = all0 && (img[(b * C + c) * HW + p] == 0.0f);
    out[i] = all0 ? 0.0f : 1.0f;
}

static unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

extern "C" int smirk_scatter_points_mask(const int64_t* points, const int64_t* rbound, int B, int L, int H, int W, float* out, void* stream) {
    if (!points || !out || B <= 0 || L <= 0) return SMIRK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, (size_t)B * H * W * 4, st) != hipSuccess) return SMIRK_ERR_LAUNCH;
    SMIRK_LAUNCH(scatter_points_kernel, dim3(nblk((size_t)B * L)), dim3(256), 0, st, (const long long*)points, (const long long*)rbound,
                       B, L, H, W, out);
    return smirk_launch_status();
}

extern "C" int smirk_rendered_mask(const float* rendered_img, int B, int C, int H, int W, float* out, void* stream) {
    if (!rendered_img || !out || B <= 0 || C <= 0) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(rendered_mask_kernel, dim3(nblk((size_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, rendered_img, B, C, H * W, out);
    return smirk_launch_status();
}

extern "C" int smirk_mask_face_weights(const float* tverts, const float* normals, const int32_t* faces, const float* face_prob,
                                       int B, int V, int F, float* weights, void* stream) {
    if (!tverts || !normals || !faces || !face_prob || !weights || B <= 0 || F <= 0) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(mask_face_weights_kernel, dim3(nblk((size_t)B * F)), dim3(256), 0, (hipStream_t)stream, tverts, normals, faces,
                       face_prob, B, V, F, weights);
    return smirk_launch_status();
}

extern "C" int smirk_sample_faces(const float* weights, int B, int F, int num, uint64_t seed, uint64_t offset, int32_t* idx,
                                  float* bary, void* stream) {
    if (!weights || !idx || !bary || B <= 0 || F <= 0 || num <= 0 || F > 38000) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(sample_faces_kernel, dim3(B), dim3(1024), (size_t)(F + 32) * 4, (hipStream_t)stream, weights, F, num, seed,
                       offset, idx, bary);
    return smirk_launch_status();
}

extern "C" int smirk_points_to_pixels(const float* points, int B, int L, int image_size, int64_t* out, void* stream) {
    if (!points || !out || B <= 0 || L <= 0 || image_size <= 0) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(points_to_pixels_kernel, dim3(nblk((size_t)B * L)), dim3(256), 0, (hipStream_t)stream, points, (size_t)B * L,
                       image_size, (long long*)out);
    return smirk_launch_status();
}

// returns false when the > 64 KB dynamic-LDS opt-in failed on this device (the caller then takes the two-pass kernels instead of a launch that would fail)
template <int R>
static bool launch_maxpool_lds(const float* in, float* out, int B, int H, int W, int complement, hipStream_t st) {
    const size_t lds = (size_t)(MP_ROWS + 2 * R) * ((size_t)mp_row_stride(W, R) + W) * sizeof(float);
    static bool attr_done[64] = {};                    // per device: the attribute is per-device state (one process may drive several GPUs)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute((const void*)maxpool_sq_lds_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return false;                              // not marked done: retried (and reported through the fallback) on the next call
        }
        attr_done[dev] = true;
    }
    const int tiles = (H + MP_ROWS - 1) / MP_ROWS;
    smirk_prof_next(nullptr, 0.0, 2.0 * B * H * W * sizeof(float));
    SMIRK_LAUNCH(maxpool_sq_lds_kernel<R>, dim3((unsigned)(B * tiles)), dim3(512), lds, st, in, out, H, W, complement & 1, (complement >> 1) & 1);
    return true;
}

extern "C" int smirk_maxpool_sq(const float* in, float* tmp, float* out, int B, int H, int W, int radius, int complement, void* stream) {
    if (!in || !tmp || !out || B <= 0 || H <= 0 || W <= 0 || radius < 0) return SMIRK_ERR_BAD_ARG;
    const size_t n = (size_t)B * H * W;
    // the fused LDS form covers the radii the reference uses (masking.py:78 -> 10, :96 -> 5) at widths whose staged rows fit 160 KB of LDS
    const size_t staged = (size_t)(MP_ROWS + 2 * radius) * ((size_t)W + mp_row_stride(W, radius)) * sizeof(float);
    // The fused form reads rows [y0 - R, y0 + 32 + R) of `in` while other workgroups write `out`: in == out (alias-safe in the two-pass form, which goes
    // through tmp) would race, so an aliased call keeps the two-pass kernels.  `tmp` must not alias either tensor in any form.
    if (tmp == in || tmp == out) return SMIRK_ERR_BAD_ARG;
    const bool lds_ok = staged <= 160 * 1024 && (size_t)B * ((H + MP_ROWS - 1) / MP_ROWS) < 0x7fffffffull && in != out;
    if (lds_ok && radius == 10 && launch_maxpool_lds<10>(in, out, B, H, W, complement, (hipStream_t)stream)) return smirk_launch_status();
    if (lds_ok && radius == 5 && launch_maxpool_lds<5>(in, out, B, H, W, complement, (hipStream_t)stream)) return smirk_launch_status();
    SMIRK_LAUNCH(maxfilter1d_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, in, tmp, B, H, W, radius, 0, complement & 1, 0);
    SMIRK_LAUNCH(maxfilter1d_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, (const float*)tmp, out, B, H, W, radius, 1, 0,
                       (complement >> 1) & 1);
    return smirk_launch_status();
}

extern "C" int smirk_bernoulli_field(float* out, size_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
    if (!out || n == 0) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(bernoulli_field_kernel, dim3(nblk((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, p, seed, offset);
    return smirk_launch_status();
}

extern "C" int smirk_masking_compose(const float* img, const float* mask, const float* rendered_mask, const float* extra_points,
                                     const float* pmask, const float* keep, const float* noise_mult, int B, int C, int H, int W,
                                     int gen_noise, uint64_t seed, uint64_t offset, float* out, void* stream) {
    if (!img || !mask || (!extra_points && !pmask) || !out || B <= 0 || C <= 0) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(masking_compose_kernel, dim3(nblk((size_t)B * C * H * W)), dim3(256), 0, (hipStream_t)stream, img, mask,
                       rendered_mask, extra_points, pmask, keep, noise_mult, B, C, H * W, gen_noise, seed, offset, out);
    return smirk_launch_status();
}

extern "C" int smirk_transfer_pixels(const float* img, const int64_t* points1, const int64_t* points2, const int64_t* rbound, int B,
                                     int C, int L, int H, int W, int32_t* winner_ws, float* out, void* stream) {
    if (!img || !points1 || !points2 || !winner_ws || !out || B <= 0 || L <= 0) return SMIRK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(winner_ws, 0xFF, (size_t)B * H * W * 4, st) != hipSuccess) return SMIRK_ERR_LAUNCH;   // -1
    SMIRK_LAUNCH(transfer_winner_kernel, dim3(nblk((size_t)B * L)), dim3(256), 0, st, (const long long*)points2,
                       (const long long*)rbound, B, L, H, W, winner_ws);
    SMIRK_LAUNCH(transfer_gather_kernel, dim3(nblk((size_t)B * C * H * W)), dim3(256), 0, st, img, (const long long*)points1,
                       (const int*)winner_ws, B, C, L, H, W, out);
    return smirk_launch_status();
}

// demo.py:154-156: rsing = randint(0, 2) * 2 - 1, rscale = rand() * (mul - 1) + 1, rbound = (n * (1 / mul) * rscale ** rsing).long() — per image,
// drawn on the device so the step has no host round trip (the reference draws them on the host and copies them over)
__global__ __launch_bounds__(256) void point_budget_kernel(long long* __restrict__ rbound, int B, int n_points, float mul, uint64_t seed,
                                                           uint64_t offset) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t r[4];
    const uint64_t ctr = offset + (uint64_t)b;
    philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), 3u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float rscale = u01(r[0]) * (mul - 1.0f) + 1.0f;
    const float f = (r[1] & 1u) ? rscale : 1.0f / rscale;
    rbound[b] = (long long)((float)n_points * (1.0f / mul) * f);
}

extern "C" int smirk_random_point_budget(int64_t* rbound, int B, int n_points, float mul, uint64_t seed, uint64_t offset, void* stream) {
    if (!rbound || B <= 0 || n_points <= 0 || !(mul >= 1.0f)) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(point_budget_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, (long long*)rbound, B, n_points, mul, seed, offset);
    return smirk_launch_status();
}
