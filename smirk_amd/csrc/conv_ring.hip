// conv_ring.hip — 3x3 stride-1 zero-padded convolution with 64 OUTPUT channels on large images (the U-Net's 112 x 112 layers: 32 -> 64, 64 -> 64,
// (64 ++ 64) -> 64; src/smirk_generator.py:54-55,72-73, _block :88-119), split-fp16 (f16x3) arithmetic — see conv.hip for the number format.
//
// Why another kernel.  These four layers run at 280-320 TFLOP/s (profiles/r04b_conv_sweep_B128_B1024.txt: 14.5 ms of the 1024-frame step), a third of the
// split-fp16 ceiling and far from their HBM floor (6.6-9.9 GB, 1.3-2 ms): with 64 output channels the weight tensor (147 / 295 KB) does not fit in LDS next to a
// halo stage, so round 3 served them with a 128 x 64 implicit-GEMM tile (nine shifted im2col copies through LDS, one workgroup's waves 32 columns wide: 1 LDS
// fragment read per MFMA) or with the streamed-weights patch kernel (two 32-column halves: also 1 read per MFMA, 154 KB of LDS = ONE 4-wave workgroup per CU, every
// phase serial).  Here:
//   * a workgroup owns a 16 x 16 output patch, stages its 18 x 18 x 32-channel input halo ONCE per channel chunk (41.5 KB, all nine taps read it) and each of its
//     four waves computes 64 pixels x ALL 64 output channels (2 x 2 MFMA tiles: 8 fragment reads per 12 MFMAs, the deep-layer kernels' ratio);
//   * the weights stream through a 3-stage ring of [64 couts][32 channels] 8 KB chunks, one per (channel chunk, tap), fetched two phases ahead by LDS-DMA
//     (conv_halo.hip's ring: 9 phases per chunk, so a phase's stage is tap % 3 — a compile-time constant);
//   * LDS = 24.6 KB ring + 42 KB halo = 66.6 KB: TWO workgroups per CU (THREE for the 32-output-channel instantiation: 12 KB ring, 144 VGPRs), whose load / wait / epilogue phases cover each other's matrix phases (what the Cout = 32
//     patch kernels do with resident weights);
//   * optional 2 x 2 max-pool of the output in the epilogue (encoder2 -> pool2): the pooled tensor comes from the fp32 transpose buffer, the separate pooling
//     launch (and its re-read of the 3.3 GB tensor) disappears.
// K order: channel chunk -> tap -> 2 k-steps (as conv3x3_patch_kernel), epilogue arithmetic identical to it: results are bit-identical to the patch kernels'.
//
// Bound: MFMA (dense fp16 2.5 PF, 3 MFMAs per product); algorithmic flop 2 * B*H*W * 64 * 9 * Cin, bytes = input + output (+ pooled) + weights.
#include <stdlib.h>
#include <type_traits>

#include "conv_common.h"

#define RG_PT 16
#define RG_PH (RG_PT + 2)
#define RG_PPIX (RG_PH * RG_PH)                    // 324 halo pixels
#define RG_HALO_SLOTS 328                          // 41 DMA instructions of 8 pixels (the 324 halo pixels + 4 slots of zeros)
#define RG_HALO_BYTES (RG_HALO_SLOTS * 128)        // 41,984
#define RG_LDS_BYTES(COUT) (3 * (COUT) * 128 + RG_HALO_BYTES)   // 66,560 (Cout 64: two workgroups per CU) / 54,272 (Cout 32: three)

struct RingArgs {
    const float *in0, *in1, *w, *scale, *shift;
    float *out, *pool;
    int B, H, W, C0, C1;
    int nchunk, npatch, act;
    float* stats;    // train mode: per (workgroup, wave) partial column sums [gridDim.x * 4][Cout][2] of the raw output (BatchNorm statistics), or nullptr
};

typedef __attribute__((address_space(3))) void* rg_lptr_t;

__device__ __forceinline__ int rg_piece_w(int row, int piece) { return row * 128 + ((piece ^ ((row >> 1) & 7)) << 4); }            // bytes
__device__ __forceinline__ int rg_piece_a(int pix, int x, int piece) { return pix * 128 + ((piece ^ ((x >> 1) & 7)) << 4); }        // bytes; x = halo column

__device__ __forceinline__ void rg_frag_ready(const half8& a, const half8& b, const half8& c, const half8& d) {
    asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d));
}

// TN = output channels / 32 (1: the two-source 64 -> 32 layer at 224 x 224, 2: the 64-channel layers at 112 x 112); POOL: also write the 2 x 2 max-pool of the output
template <int TN, bool POOL, bool STATS = false>
__global__ __launch_bounds__(256, TN == 1 ? 3 : 2) void conv3x3_ring_kernel(RingArgs a) {
    constexpr int COUT = 32 * TN, STAGE = COUT * 128, EPI_LD = COUT + 4, GPR = COUT / 8, ITEMS = 32 * GPR / 64;
    extern __shared__ __attribute__((aligned(16))) char rg_lds[];
    char* const ring = rg_lds;
    constexpr int RING = 3 * STAGE;
    char* const halo = rg_lds + RING;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, hb = lane >> 5;
    const int ry = fr >> 4, rx = fr & 15;
    const int H = a.H, W = a.W, Cin = a.C0 + a.C1, K = 9 * Cin;
    const int tiles_x = W / RG_PT, tiles_per_img = (H / RG_PT) * tiles_x;

    const long long px_all = (long long)a.B * H * W;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.in0, (short)0, (int)(px_all * a.C0 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.C1 > 0 ? a.in1 : a.in0), (short)0, (int)(px_all * (a.C1 > 0 ? a.C1 : a.C0) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, (int)((long long)COUT * K * 4), 0x00020000);
    const int sh0 = 33 - __builtin_clz((unsigned)a.C0), sh1 = a.C1 > 0 ? 33 - __builtin_clz((unsigned)a.C1) : sh0;      // log2(4 C): byte offset of a pixel

    // weight chunk (cc, tap) -> ring stage st: rows 8 j .. 8 j + 7 for j = wave (, wave + 4 when Cout = 64) (64 lanes = 8 rows x 8 pieces, lane-linear in LDS; the XOR swizzle is
    // applied on the source side: position pos of row r receives piece pos ^ ((r >> 1) & 7))
    unsigned voffB[2] = {0u, 0u};                                    // [TN] entries used (a template-sized array captured by the lambdas below made hipcc 7.0 drop the HOST-side instantiation of the kernel: undefined stub at load time)
#pragma unroll
    for (int k = 0; k < TN; ++k) {
        const int row = 8 * (wave + 4 * k) + (lane >> 3);
        voffB[k] = ((unsigned)row * (unsigned)K + (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 4u) * 4u;
    }
    auto dmaB = [&](int cc, int tap, int st) {
        const int so = (tap * Cin + cc * 32) * 4;
#pragma unroll
        for (int k = 0; k < TN; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (rg_lptr_t)(ring + st * STAGE + (wave + 4 * k) * 1024), 16, voffB[k], so, 0, 0);
    };
    // halo of chunk cc of patch (b, oy0, ox0): instruction q of this wave covers halo pixels (4 q + wave) * 8 .. + 7; outside the image -> out-of-range offset -> zeros
    auto dmaA = [&](int b, int oy0, int ox0, int cc) {
        const int c0 = cc * 32;
        const bool s1 = c0 >= a.C0;
        const int cb = s1 ? c0 - a.C0 : c0, sh = s1 ? sh1 : sh0;
        const int basepix = (b * H + oy0 - 1) * W + ox0 - 1;
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            if ((q * 4 + wave) * 8 >= RG_HALO_SLOTS) continue;        // wave-uniform: 41 instructions cover the halo
            int l8 = lane >> 3;
            asm volatile("" : "+v"(l8));                              // slot arithmetic is redone per issue instead of living in 2 x 11 registers (conv_patch.hip)
            const int pix = (q * 4 + wave) * 8 + l8;
            const int hy = (pix * 3641) >> 16, hx = pix - hy * RG_PH;  // pix / 18 for pix < 400
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            const bool ok = pix < RG_PPIX && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int piece = (lane & 7) ^ ((hx >> 1) & 7);
            const unsigned vo = ok ? ((unsigned)(basepix + hy * W + hx) << sh) + (unsigned)piece * 16u : 0x80000000u;
            char* dst = halo + (q * 4 + wave) * 8 * 128;
            if (s1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (rg_lptr_t)dst, 16, vo, cb * 4, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (rg_lptr_t)dst, 16, vo, cb * 4, 0, 0);
        }
    };

    // per-lane fragment bases (bytes): A = pixel (4 wave + ry, rx + kx) of the halo at piece 2 hb (swizzled by the column), B = row fr of a ring stage
    unsigned aB[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
        aB[kx] = (unsigned)RING + (unsigned)((4 * wave + ry) * RG_PH + rx + kx) * 128u + (unsigned)(((((rx + kx) >> 1) & 7) << 4) ^ (hb << 5));
    const unsigned bB = (unsigned)fr * 128u + (unsigned)((((fr >> 1) & 7) << 4) ^ (hb << 5));      // + j * 4096 (N tile), + stage * STAGE; (row >> 1) & 7 == (fr >> 1) & 7

    float ep_sc[8], ep_sh[8];                                        // epilogue coefficients of this lane's channel group (g = lane % GPR), loaded once
#pragma unroll
    for (int q = 0; q < 8; ++q) { ep_sc[q] = a.scale ? a.scale[(lane % GPR) * 8 + q] : 1.f; ep_sh[q] = a.shift ? a.shift[(lane % GPR) * 8 + q] : 0.f; }

    int p = blockIdx.x;
    if (p >= a.npatch) return;
    dmaB(0, 0, 0);
    dmaB(0, 1, 1);
    f32x16 acc0[2][TN], acc1[2][TN];
    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float st1[TN], st2[TN];                                          // STATS: column sums of everything this wave computed (sum of per-patch sums: short fp32 chains)
#pragma unroll
    for (int j = 0; j < TN; ++j) { st1[j] = 0.f; st2[j] = 0.f; }
    SmirkRangeAccS rng;                                               // split-fp16 range audit (common.h): running max over every patch of this workgroup, tested once at the end
    for (; p < a.npatch; p += gridDim.x) {
        const int b = p / tiles_per_img, t = p - b * tiles_per_img, ty = t / tiles_x;
        const int oy0 = ty * RG_PT, ox0 = (t - ty * tiles_x) * RG_PT;
        for (int cc = 0; cc < a.nchunk; ++cc) {
            // ---- this chunk's halo: the stage is single (the second workgroup of the CU computes meanwhile); everyone has left the previous chunk / epilogue
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            dmaA(b, oy0, ox0, cc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // halo + the two weight chunks in flight (+ the previous patch's stores)
            __builtin_amdgcn_s_barrier();
            const int ccn = cc + 1 < a.nchunk ? cc + 1 : 0;
            auto phase = [&](auto tapc, auto firstc) {
                constexpr int TAP = decltype(tapc)::value, ST = TAP % 3, KY = TAP / 3, KX = TAP % 3;
                constexpr bool FIRST = decltype(firstc)::value;       // first phase of a patch: the accumulators start from the MFMA's inline zero
                if constexpr (TAP > 0) {
                    if constexpr (TN == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");   // this phase's weights have landed (the next phase's TN pieces may still fly)
                    else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                     // ... for every wave, and every wave has left the stage that is refilled now
                }
                // the weights of the phase two ahead; past the last phase of this workgroup's last patch chunk 0 is fetched again into a stage nobody reads (the
                // counted wait above relies on every phase issuing exactly TN pieces)
                constexpr int T2 = (TAP + 2) % 9;
                dmaB(TAP + 2 < 9 ? cc : ccn, T2, T2 % 3);
                unsigned av = aB[KX], bv = bB;
                asm volatile("" : "+v"(av), "+v"(bv));                // the XOR variants below are formed here, not hoisted into 40 registers
                half8 ah[2][2], al[2][2], bh[2][TN], bl[2][TN];       // [k-step][tile]
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned ch = (unsigned)(s << 6), cl = ch | 16u;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const unsigned disp = (unsigned)((2 * i + KY) * RG_PH * 128);
                        ah[s][i] = *(const half8*)(rg_lds + ((av ^ ch) + disp));
                        al[s][i] = *(const half8*)(rg_lds + ((av ^ cl) + disp));
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bh[s][j] = *(const half8*)(rg_lds + ((bv ^ ch) + (unsigned)(ST * STAGE + j * 4096)));
                        bl[s][j] = *(const half8*)(rg_lds + ((bv ^ cl) + (unsigned)(ST * STAGE + j * 4096)));
                    }
                }
                // matrix section at priority 1 (the other resident workgroup's loads / address arithmetic yield the issue slots): +0.5-3 % on the 64-channel
                // instantiation, nothing on the 32-channel one (same-box A/B of a -DRG_SETPRIO build, round 4)
                if constexpr (TN == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    rg_frag_ready(ah[s][1], al[s][1], bh[s][TN - 1], bl[s][TN - 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bh[s][j], (FIRST && s == 0) ? z16 : acc0[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bl[s][j], (FIRST && s == 0) ? z16 : acc1[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][i], bh[s][j], acc1[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (TN == 2) __builtin_amdgcn_s_setprio(0);
            };
            if (cc == 0) phase(std::integral_constant<int, 0>{}, std::true_type{});
            else phase(std::integral_constant<int, 0>{}, std::false_type{});
            phase(std::integral_constant<int, 1>{}, std::false_type{}); phase(std::integral_constant<int, 2>{}, std::false_type{});
            phase(std::integral_constant<int, 3>{}, std::false_type{}); phase(std::integral_constant<int, 4>{}, std::false_type{});
            phase(std::integral_constant<int, 5>{}, std::false_type{}); phase(std::integral_constant<int, 6>{}, std::false_type{});
            phase(std::integral_constant<int, 7>{}, std::false_type{}); phase(std::integral_constant<int, 8>{}, std::false_type{});
        }
        // ---- epilogue: the halo stage is dead once every wave is here: per-wave fp32 transpose buffers; BN + ReLU, re-split, whole 8-channel groups ------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* const ebuf = (float*)halo + wave * 32 * EPI_LD;
        float pt1[TN], pt2[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { pt1[j] = 0.f; pt2[j] = 0.f; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float x16[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    x16[r] = acc0[i][j][r] + acc1[i][j][r] * (1.0f / 2048.0f);
                    ebuf[mfma32_row(r, lane) * EPI_LD + j * 32 + fr] = x16[r];
                }
                if constexpr (STATS) stats_block(x16, lane, 32, pt1[j], pt2[j]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // per-wave transpose buffer: LDS operations of one wave execute in order
            const int oy = oy0 + 4 * wave + 2 * i;
            auto finish = [&](const float* src, float* v) {
                *(f32x4*)v = *(const f32x4*)src;
                *(f32x4*)(v + 4) = *(const f32x4*)(src + 4);
                if (a.scale) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= ep_sc[q];
                }
                if (a.shift) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += ep_sh[q];
                }
                if (a.act == SMIRK_ACT_RELU) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                }
            };
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {                     // item e = it * 64 + lane: tile row e / GPR (pixel (row >> 4, row & 15)), channel group e % GPR
                const int e = it * 64 + lane, row = e / GPR, g = e % GPR;
                float v[8];
                finish(ebuf + row * EPI_LD + g * 8, v);
                half8 hi, lo;
                split8(v, hi, lo, rng);
                float* o = a.out + (((size_t)b * H + oy + (row >> 4)) * W + ox0 + (row & 15)) * COUT + g * 8;
                *(half8*)o = hi;
                *(half8*)(o + 4) = lo;
            }
            if constexpr (POOL) {                                     // the tile's 8 pooled pixels x GPR groups (<= one item per lane): max of the 2 x 2 pixels' finished values
                const int pp = lane / GPR, g = lane % GPR;
                float m[8], v[8];
                if (pp < 8) {
                finish(ebuf + (2 * pp) * EPI_LD + g * 8, m);
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    finish(ebuf + (2 * pp + (k & 1) + 16 * (k >> 1)) * EPI_LD + g * 8, v);
#pragma unroll
                    for (int q = 0; q < 8; ++q) m[q] = fmaxf(m[q], v[q]);
                }
                half8 hi, lo;
                split8_noaudit(m, hi, lo);                            // (the maximum of four values audited in the item loop above)
                float* o = a.pool + (((size_t)b * (H >> 1) + (oy >> 1)) * (W >> 1) + (ox0 >> 1) + pp) * COUT + g * 8;
                *(half8*)o = hi;
                *(half8*)(o + 4) = lo;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the buffer is rewritten by the next tile
        }
        if constexpr (STATS) {
#pragma unroll
            for (int j = 0; j < TN; ++j) { st1[j] += pt1[j]; st2[j] += pt2[j]; }
        }
    }
    rng.commit();
    if constexpr (STATS) {
        float* prow = a.stats + (size_t)(blockIdx.x * 4 + wave) * COUT * 2;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float t1 = stats_pair(st1[j]), t2 = stats_pair(st2[j]);
            if (hb == 0) { prow[(j * 32 + fr) * 2] = t1; prow[(j * 32 + fr) * 2 + 1] = t2; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the over-fetched weight pieces must have landed before the workgroup's LDS is released
}

// Serves: split-fp16, 3x3, stride 1, zero pad 1, same size, NHWC out, Cout = 64 (or Cout = 32 with >= 2 channel chunks: resident weights would leave one workgroup per CU), H, W multiples of 16 and >= 64, both sources whole power-of-two multiples of 32
// channels, no residual, operands below 2 GiB (the dispatcher chunks larger batches).  $SMIRK_CONV_RING=0 restores the round-3 kernels (A/B switch, tests).
bool smirk_conv3x3_ring64_eligible(const SmirkConvDesc* d, bool has_residual) {
    const char* env = getenv("SMIRK_CONV_RING");                     // read per call: tests toggle it
    if (env && env[0] == '0') return false;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1 || d->pad_mode != SMIRK_PAD_ZERO) return false;
    if (d->out_mode != SMIRK_OUT_NHWC || d->Ho != d->H || d->Wo != d->W || d->H % RG_PT || d->W % RG_PT || d->H < 64 || has_residual) return false;
    if ((d->Cout != 64 && !(d->Cout == 32 && d->C0 + d->C1 >= 64)) || d->C0 < 32 || d->C0 % 32 || d->C1 % 32 || (d->C0 & (d->C0 - 1)) || (d->C1 & (d->C1 - 1))) return false;
    const long long px = (long long)d->B * d->H * d->W;
    return px * d->C0 * 4 < (1ll << 31) && px * d->C1 * 4 < (1ll << 31);
}

int smirk_conv3x3_ring64_launch(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, const float* scale, const float* shift, void* out,
                                void* pooled, hipStream_t st, float* stats, int* stats_rows) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SMIRK_ERR_LAUNCH;
    static bool attr_done[64] = {};                                  // hipFuncSetAttribute is per-device state
    static int n_cu[64] = {};
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute((const void*)conv3x3_ring_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS_BYTES(64)) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv3x3_ring_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS_BYTES(64)) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv3x3_ring_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS_BYTES(32)) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv3x3_ring_kernel<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS_BYTES(64)) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv3x3_ring_kernel<1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS_BYTES(32)) != hipSuccess)
            return SMIRK_ERR_LAUNCH;
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n_cu[dev] = cus;
        attr_done[dev] = true;
    }
    RingArgs a;
    a.in0 = (const float*)in0; a.in1 = (const float*)in1; a.w = (const float*)w; a.scale = scale; a.shift = shift;
    a.out = (float*)out; a.pool = (float*)pooled;
    a.B = d->B; a.H = d->H; a.W = d->W; a.C0 = d->C0; a.C1 = d->C1; a.act = d->act;
    a.nchunk = (d->C0 + d->C1) / 32;
    a.npatch = d->B * (d->H / RG_PT) * (d->W / RG_PT);
    const int per_cu = d->Cout == 32 ? 3 : 2;                        // resident workgroups per CU (LDS)
    const int grid = a.npatch < per_cu * n_cu[dev] ? a.npatch : per_cu * n_cu[dev];
    a.stats = nullptr;
    if (stats && stats_rows && !pooled && !scale && !shift && d->act == SMIRK_ACT_NONE) { a.stats = stats; *stats_rows = grid * 4; }   // train mode: raw output + its column sums
    if (g_smirk_prof_on) {
        const double px = (double)d->B * d->H * d->W, K = 9.0 * (d->C0 + d->C1);
        smirk_prof_next(d->Cout == 32 ? "conv3x3_ring_kernel<1>[16x16 patch,ring]" : pooled ? "conv3x3_ring_kernel<2,pool>[16x16 patch,ring]" : "conv3x3_ring_kernel<2>[16x16 patch,ring]",
                        2.0 * px * d->Cout * K, px * (d->C0 + d->C1) * 4.0 + px * d->Cout * 4.0 * (pooled ? 1.25 : 1.0) + K * d->Cout * 4.0);
    }
    if (a.stats) {
        if (d->Cout == 32) SMIRK_LAUNCH((conv3x3_ring_kernel<1, false, true>), dim3(grid), dim3(256), RG_LDS_BYTES(32), st, a);
        else SMIRK_LAUNCH((conv3x3_ring_kernel<2, false, true>), dim3(grid), dim3(256), RG_LDS_BYTES(64), st, a);
    } else if (d->Cout == 32) {
        if (pooled) return SMIRK_ERR_UNSUPPORTED;
        SMIRK_LAUNCH((conv3x3_ring_kernel<1, false>), dim3(grid), dim3(256), RG_LDS_BYTES(32), st, a);
    } else if (pooled) SMIRK_LAUNCH((conv3x3_ring_kernel<2, true>), dim3(grid), dim3(256), RG_LDS_BYTES(64), st, a);
    else SMIRK_LAUNCH((conv3x3_ring_kernel<2, false>), dim3(grid), dim3(256), RG_LDS_BYTES(64), st, a);
    return smirk_launch_status();
}
