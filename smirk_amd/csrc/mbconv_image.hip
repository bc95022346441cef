// mbconv_image.hip — image-resident fused MobileNetV3 InvertedResidual block for the <= 14 x 14 stages of the timm "minimal" backbones (gfx950):
//
//      x --1x1 expand + BN + ReLU--> E --3x3 depthwise (stride 1, pad 1) + BN + ReLU--> D --1x1 project + BN (+x)--> out
//
// (reference call site smirk_encoder.py:18-21,52-55,80-83 `self.encoder(img)[-1]`.)  mbconv.hip fuses the blocks with Cin <= 48 on
// 8 x 8 tiles; the 14 x 14 / 7 x 7 blocks with 80-112 input channels ran as three launches (pointwise / depthwise / pointwise) that write and re-read
// the 3-6x wider expanded tensors: at 1024 frames 17 ms of ~2 TB/s kernels for 2 % of the path's flops (round-2 verdict, item 3d).  Here ONE workgroup
// owns WHOLE images — one 14 x 14 image or four 7 x 7 images = 196 pixels = seven 32-row MFMA blocks, one per wave — so there is no halo to recompute:
//   * x (196 px x Cin, split16 as stored) is loaded into LDS once and stays there (A operand of the expand GEMM, residual of the epilogue);
//   * the expanded channels are walked in chunks of 32:
//       beta   P1(c)   E_c = relu(bn1(X . Wexp_c))   3 x v_mfma_f32_32x32x16_f16 per 16-k step (split-fp16 x3), Wexp_c fragments prefetched into registers,
//                      scattered as fp32 into a zero-bordered (H+2) x (W+2) grid in LDS (the depthwise conv pads E, not x);
//       alpha  DW(c)   every lane computes depthwise + BN + ReLU for ITS pixel row and ITS 8 channels of each 16-k step — exactly the MFMA A-operand
//                      fragment (row = lane % 32, k = 8 (lane / 32) ..) of the project GEMM — straight into registers: D never exists in memory;
//       beta   P3(c)   P += D_c . Wproj_c, the wave's four 32-column accumulator tiles (Cout <= 128) stay in registers across the chunks; Wproj_c is
//                      staged through LDS once per workgroup (register-staged copy one chunk ahead, 144-byte rows: conflict-free fragment reads);
//     two barriers per chunk (alpha | beta: the grid is rewritten by P1(c+1) only after every wave has read it for DW(c));
//   * epilogue: bn3 (+ x from LDS), per-wave transpose through LDS, whole 8-channel split16 groups to HBM.
// HBM traffic per block: x in + out (0.16 GB at 1024 frames for 112 -> 672 -> 112 instead of 2.3 GB).  LDS: 91 KB (x) + 32 KB (E grid, XOR-swizzled
// 16-byte slots) + 2 x 16 KB (Wproj chunks) + tables = 156 KB in the largest case: one 512-thread workgroup per CU.
// Bound: latency / issue (1 workgroup per CU, phases separated by barriers); the point is the 14x smaller memory footprint, not MFMA occupancy.
// Arithmetic identical in kind to mbconv.hip (f16x3 products, fp32 accumulation, fp32 depthwise, E kept in fp32).
#include <stdio.h>

#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct MBIArgs {
    const char* x;          // split16 NHWC, Cin*4 bytes per pixel
    const char* wexp;       // [mid][Cin] split16 rows
    const float *s1, *b1;   // [mid]
    const float* wdw;       // [9][mid]
    const float *s2, *b2;   // [mid]
    const char* wproj;      // [Cout][mid] split16 rows
    const float *s3, *b3;   // [Cout]
    char* out;              // split16 NHWC [B][H][W][Cout]
    int B, H, W, Cin, mid, Cout, residual, ipw, rows, gsz, nb;   // ipw images per workgroup, rows = ipw*H*W (<= 224), gsz = ipw*(H+2)*(W+2), nb = ceil(rows/32)
    int off_es, off_wp, off_wc, off_gmap;                        // byte offsets into the dynamic LDS block
    // halo-tile mode (round 5): images larger than 224 pixels whose sides are multiples of 14 (28 x 28, 56 x 56, 112 x 112 — every earlier stage of the two backbones
    // at 224 x 224 input) are cut into 14 x 14 tiles; a workgroup stages the tile's 16 x 16 halo of x, expands ALL 256 halo pixels (eight 32-row blocks: the
    // depthwise conv needs E one pixel beyond the tile; E is zero outside the image, not x), runs depthwise / project / epilogue on the 196 interior pixels exactly
    // as the whole-image mode does on its 196 rows.  The E grid is the 16 x 16 halo itself (pitch 16 instead of W + 2).
    int tile, tiles_x, tiles_y, gpitch;                          // tile = 1: halo tiles; gpitch = row pitch of the E grid in cells
    int rowsE, nbE, off_gmapE;                                   // rows / 32-row blocks of the expand GEMM (= rows / nb in the whole-image mode), its row -> cell map
};

__device__ __forceinline__ void mbi_barrier() {              // LDS-only barrier: global prefetches stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// fp32 offset of (grid position gi, channel c) inside the E grid: rows of 32 floats, the eight 16-byte slots of a row XOR-swizzled with (gi >> 1) & 7 so
// that the 16 lanes of a ds_read_b128 group (16 neighbouring pixels, same channel quad) hit 16 distinct slots
__device__ __forceinline__ int mbi_es(int gi, int c) { return gi * 32 + ((((c >> 2) ^ (gi >> 1)) & 7) << 2) + (c & 3); }

// KS = Cin / 16 (16-k MFMA steps of the expand GEMM), NT = 32-column tiles of the project GEMM (Cout <= 32 NT)
// TILE: halo-tile mode (MBIArgs); PADK: Cin is not a multiple of 16 — both compile-time so that the register-bound whole-image instantiations (KS 4-7, 128 accumulators)
// carry none of the extra code
template <int KS, int NT, bool TILE = false, bool PADK = false>
__global__ __launch_bounds__(512, 2) void mbconv_image_kernel(MBIArgs a) {
    extern __shared__ __attribute__((aligned(16))) char mbi_smem[];
    SmirkRangeAccS rng;                                 // split-fp16 range audit (common.h), scalar-register form: these instantiations sit at 226-256 VGPRs
    const int strideX = KS * 64 + 16;                   // Cin padded to 16 KS channels (zeros): odd multiple of 16 B
    char* Xs = mbi_smem;
    float* Es = (float*)(mbi_smem + a.off_es);
    char* Wp = mbi_smem + a.off_wp;                     // 2 x [Cout][144 B]
    float* Wc = (float*)(mbi_smem + a.off_wc);          // 2 x [11][32]: 9 depthwise taps, s2, b2 of a chunk
    short* gmap = (short*)(mbi_smem + a.off_gmap);      // [nb * 32]: grid cell of output pixel row r (-1: no such row)
    short* gmapE = TILE ? (short*)(mbi_smem + a.off_gmapE) : gmap;    // [nbE * 32]: grid cell c of expand row r; ~c for a cell that must hold 0 (outside the image / no such row)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, hb = lane >> 5;
    const int ntile = a.tiles_x * a.tiles_y;
    const int tb = TILE ? (int)blockIdx.x / ntile : 0, tt = TILE ? (int)blockIdx.x - tb * ntile : 0;
    const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;                     // halo-tile mode: tile (tx, ty) of image tb
    const int b0 = TILE ? tb : blockIdx.x * a.ipw;
    const int nimg = min(a.ipw, a.B - b0), HW = a.H * a.W, vrows = TILE ? 196 : nimg * HW;   // valid rows (the last workgroup may hold fewer images)
    const int nchunks = (a.mid + 31) / 32;
    const size_t xrow = (size_t)a.Cin * 4;
    const half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const bool active = swave < a.nb;                   // one 32-row block per wave (depthwise / project / epilogue rows)
    const bool activeE = TILE ? swave < a.nbE : active; // ... of the expand GEMM (all eight waves in the halo-tile mode)
    const int wpbuf = a.Cout * 144;

    // ---- prefetch helpers ----------------------------------------------------------------------------------------------------------------------------
    half8 we[KS][2];                                    // expand-weight fragments of the chunk P1 computes next (column = 32 c + fr)
    float s1v = 0.f, b1v = 0.f;
    auto load_we = [&](int c) {
        // channels beyond mid (ragged last chunk): the fragments of a clamped row are loaded unconditionally (no branches in the loop) and the chunk's
        // BN constants are zeroed instead, so E = relu(finite * 0 + 0) = 0 there
        int frw = fr;
        asm volatile("" : "+v"(frw));
        const int ch = 32 * c + frw;
        const bool v = ch < a.mid;
        const int chc = min(ch, a.mid - 1);
        const char* wbase = a.wexp;
        asm volatile("" : "+s"(wbase));                  // the base pointer is re-read from its SGPR pair per chunk (a VGPR copy hoisted out of the loop went to scratch at <7, 4>)
        const char* wrow = wbase + (unsigned)chc * (unsigned)(a.Cin * 4);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int g = 2 * s + hb;
            if constexpr (PADK) {                        // Cin not a multiple of 16 (24, 40): the last k-step's upper half is padding -> zero fragments
                const int gc = min(g, a.Cin / 8 - 1);
                const half8 t0 = *(const half8*)(wrow + gc * 32), t1h = *(const half8*)(wrow + gc * 32 + 16);
                we[s][0] = g * 8 < a.Cin ? t0 : hz;
                we[s][1] = g * 8 < a.Cin ? t1h : hz;
            } else {
                we[s][0] = *(const half8*)(wrow + g * 32);
                we[s][1] = *(const half8*)(wrow + g * 32 + 16);
            }
        }
        const float t1 = a.s1[chc], t2 = a.b1[chc];
        s1v = v ? t1 : 0.f;
        b1v = v ? t2 : 0.f;
    };
    // Wproj chunk c: Cout rows x 128 B (4 groups of [8 hi][8 lo]) = Cout x 8 sixteen-byte vectors, <= 2 per thread; Wc chunk: 11 x 32 floats
    f32x4 wpv[2];
    float wcv = 0.f;
    auto fetch_stage = [&](int c) {
        int t0 = tid;
        asm volatile("" : "+v"(t0));                     // addresses re-derived per chunk: hoisted out of the chunk loop they are six more live registers (-> scratch at NT = 4)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = t0 + 512 * u, row = idx >> 3, piece = idx & 7, kg = 4 * c + (piece >> 1);
            const bool ok = row < a.Cout && kg * 8 < a.mid;
            const int rc = min(row, a.Cout - 1), kc = min(kg, a.mid / 8 - 1);
            const f32x4 t = *(const f32x4*)(a.wproj + ((size_t)rc * a.mid + kc * 8) * 4 + (piece & 1) * 16);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            wpv[u] = ok ? t : z;
        }
        {   // 11 x 32 depthwise / BN constants: wdw is [9][mid], s2 and b2 follow the same indexing through a pointer select
            const int k = min(t0 >> 5, 10), ch = 32 * c + (t0 & 31), chc = min(ch, a.mid - 1);
            const float* src = k < 9 ? a.wdw + (size_t)k * a.mid : (k == 9 ? a.s2 : a.b2);
            const float t = src[chc];
            wcv = ch < a.mid ? t : 0.f;
        }
    };
    auto store_stage = [&](int c) {
        char* dst = Wp + (c & 1) * wpbuf;
        int t0 = tid;
        asm volatile("" : "+v"(t0));
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = t0 + 512 * u, row = idx >> 3, piece = idx & 7;
            if (row < a.Cout) *(f32x4*)(dst + row * 144 + piece * 16) = wpv[u];
        }
        if (t0 < 352) Wc[(c & 1) * 352 + t0] = wcv;
    };

    // ---- phase 0: x -> LDS, zero grid, row -> grid maps ---------------------------------------------------------------------------------------------
    {
        const int P = a.Cin / 4;                         // 16-byte vectors per row that exist in memory (KS * 4 in LDS: the rest is zero padding)
        const char* xb = a.x + (size_t)b0 * HW * xrow;
        // four requests in flight per lane and trip (a rolled load -> store loop pays one dependent global round trip per 8 KB: 11 of them for 112 channels);
        // whole-image mode: the images of a workgroup are contiguous, so the copy is a linear stream re-strided to strideX; halo-tile mode: row = halo cell, its
        // pixel may lie outside the image (zeros).  Clamped address + select: no branches around the loads.
        const int rowsE = TILE ? 256 : a.rows, total = rowsE * P, vtotal = vrows * P;
        for (int base = 0; base < total; base += 512 * 4) {
            f32x4 xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + 512 * u + tid;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if (TILE) {
                    const int c = min(idx / P, 255), pc = idx - (idx / P) * P;
                    const int gy = ty * 14 - 1 + (c >> 4), gx = tx * 14 - 1 + (c & 15);
                    const bool ok = idx < total && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                    const int yc = min(max(gy, 0), a.H - 1), xc = min(max(gx, 0), a.W - 1);
                    const f32x4 t = *(const f32x4*)(xb + ((size_t)yc * a.W + xc) * xrow + pc * 16);
                    xv[u] = ok ? t : z;
                } else {
                    const f32x4 t = *(const f32x4*)(xb + (size_t)min(idx, vtotal - 1) * 16);
                    xv[u] = idx < vtotal ? t : z;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + 512 * u + tid, row = idx / P, pc = idx - row * P;
                if (idx < total) *(f32x4*)(Xs + row * strideX + pc * 16) = xv[u];
            }
        }
        if (PADK && KS * 4 > P) {                        // channel padding up to the MFMA k-step: zeros (Cin = 24 / 40: two vectors per row)
            const int PP = KS * 4 - P;
            for (int idx = tid; idx < rowsE * PP; idx += 512) {
                const int row = idx / PP, pc = P + idx - row * PP;
                *(f32x4*)(Xs + row * strideX + pc * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        for (int idx = tid; idx < a.gsz * 8; idx += 512) ((f32x4*)Es)[idx] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = tid; r < a.nb * 32; r += 512) {
            int g = -1;                                 // rows that do not exist
            if (r < vrows) {
                if (TILE) {
                    const int iy = r / 14, ix = r - iy * 14;
                    g = (iy + 1) * 16 + ix + 1;
                } else {
                    const int im = r / HW, rem = r - im * HW, y = rem / a.W, xx = rem - y * a.W;
                    g = (im * (a.H + 2) + y + 1) * (a.W + 2) + xx + 1;
                }
            }
            gmap[r] = (short)g;
        }
        if (TILE)
            for (int r = tid; r < a.nbE * 32; r += 512) {
                const int gy = ty * 14 - 1 + (r >> 4), gx = tx * 14 - 1 + (r & 15);
                const bool ok = r < 256 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                gmapE[r] = (short)(ok ? r : ~min(r, 255));
            }
    }
    mbi_barrier();

    f32x16 pacc[NT][2];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[q][h][r] = 0.f;
    const int myrow = min(wave * 32 + fr, (TILE ? 256 : a.rows) - 1); // clamped: rows beyond the last one duplicate it and are never stored
    const int mygi = active ? (int)gmap[min(wave * 32 + fr, a.nb * 32 - 1)] : -1;

    // P1: expand chunk (fragments in `we`, BN constants in s1v / b1v) -> E grid
    auto expand = [&]() {
        if (!activeE) return;
        f32x16 e0, e1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { e0[r] = 0.f; e1[r] = 0.f; }
        const char* arow = Xs + myrow * strideX;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int g = 2 * s + hb;
            const half8 ah = *(const half8*)(arow + g * 32), al = *(const half8*)(arow + g * 32 + 16);
            e0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, we[s][0], e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, we[s][1], e1, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, we[s][0], e1, 0, 0, 0);
            if ((s & 1) || NT == 4) __builtin_amdgcn_sched_barrier(0);      // at most two k-steps (NT = 4: one) of A fragments in flight: the register file is full (128 accumulators)
        }
        int rbase = wave * 32 + 4 * hb;                 // row of accumulator register r: rbase + (r & 3) + 8 (r >> 2)
        asm volatile("" : "+v"(rbase));                 // re-read the 16 grid positions from LDS every chunk instead of keeping them in registers
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // rows whose cell must hold 0 carry ~cell: rows that do not exist (whole-image mode: -1 = ~0, border cell 0, which is 0 anyway) and halo pixels outside
            // the image (halo-tile mode: the depthwise conv pads E with zeros there) — no branch
            const int gm = gmapE[rbase + (r & 3) + 8 * (r >> 2)];
            const float v = fmaxf((e0[r] + e1[r] * (1.0f / 2048.0f)) * s1v + b1v, 0.f);
            Es[mbi_es(gm >= 0 ? gm : ~gm, fr)] = gm >= 0 ? v : 0.f;
        }
    };


    // chunk loop; iteration c = -1 only stages chunk 0 and expands it (one copy of every phase's code: the kernel is register-bound)
    for (int c = -1; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) fetch_stage(c + 1);                   // global -> registers, lands under the depthwise arithmetic
        // ---- alpha: depthwise 3x3 + BN + ReLU of this lane's pixel row, 8 channels per 16-k step -> MFMA A fragments ----------------------------------
        half8 dh[2] = {hz, hz}, dl[2] = {hz, hz};
        if (c >= 0 && active && mygi >= 0) {
            const float* wc = Wc + (c & 1) * 352;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int c8 = 8 * (2 * s + hb);
                int gc = mygi;
                asm volatile("" : "+v"(gc));            // re-derive the 18 grid addresses per 16-k step and per chunk: kept in registers (LICM across the chunk
                                                        // loop, CSE across the two k-steps) they cost ~60 VGPRs beside the 128 accumulators -> scratch spills
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int gi = gc + (ky - 1) * a.gpitch + (kx - 1);
                        const f32x4 v0 = *(const f32x4*)(Es + mbi_es(gi, c8)), v1 = *(const f32x4*)(Es + mbi_es(gi, c8 + 4));
                        const f32x4 w0 = *(const f32x4*)(wc + (ky * 3 + kx) * 32 + c8), w1 = *(const f32x4*)(wc + (ky * 3 + kx) * 32 + c8 + 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { acc[q] = fmaf(v0[q], w0[q], acc[q]); acc[4 + q] = fmaf(v1[q], w1[q], acc[4 + q]); }
                        if (kx == 2 || NT == 4) __builtin_amdgcn_sched_barrier(0);   // at most one tap row (12 ds_read_b128 = 48 registers; NT = 4: one tap, 16) in flight beside the 128 accumulators
                    }
                const f32x4 sa = *(const f32x4*)(wc + 9 * 32 + c8), sb = *(const f32x4*)(wc + 9 * 32 + c8 + 4);
                const f32x4 ba = *(const f32x4*)(wc + 10 * 32 + c8), bb = *(const f32x4*)(wc + 10 * 32 + c8 + 4);
                // (no range audit on this INTERMEDIATE: a depthwise output beyond the fp16 range becomes inf here and reaches every output channel of the project
                // GEMM as inf / NaN, which the audit of the block's output below reports — one audit site per kernel keeps the 226-256-VGPR instantiations spill-free)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float v = fmaxf(acc[q] * (q < 4 ? sa[q & 3] : sb[q & 3]) + (q < 4 ? ba[q & 3] : bb[q & 3]), 0.f);
                    _Float16 h, l;
                    smirk_split1(v, h, l);
                    dh[s][q] = h; dl[s][q] = l;
                }
            }
        }
        if (more) store_stage(c + 1);                   // the other Wp / Wc buffer: last read in chunk c-1, read next after two barriers
        mbi_barrier();                                  // every wave has read the grid for DW(c): P1(c+1) may overwrite it
        // ---- beta: project chunk c, then expand chunk c+1 ----------------------------------------------------------------------------------------------
        if (more) load_we(c + 1);                       // lands under the project MFMAs
        if (c >= 0 && active) {
            const char* wp = Wp + (c & 1) * wpbuf;
            int frq = fr;
            asm volatile("" : "+v"(frq));               // the NT row addresses are re-derived per chunk (hoisted they are NT more live registers)
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int co = min(q * 32 + frq, a.Cout - 1);     // columns beyond Cout duplicate the last row and are never stored
                const char* brow = wp + co * 144;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int g = 2 * s + hb;
                    const half8 bhh = *(const half8*)(brow + g * 32), bll = *(const half8*)(brow + g * 32 + 16);
                    pacc[q][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh[s], bhh, pacc[q][0], 0, 0, 0);
                    pacc[q][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh[s], bll, pacc[q][1], 0, 0, 0);
                    pacc[q][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dl[s], bhh, pacc[q][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);      // one tile's B fragments at a time (hoisting all 4 NT sets costs 64 registers)
            }
        }
        if (more) expand();
        mbi_barrier();
    }

    // ---- epilogue: bn3 (+ x), per-wave 32 x 32 transposes through the (now free) grid region, split16 stores ---------------------------------------------
    if (active) {
        // lane / wave are re-derived (mbcnt, the wave id kept in an SGPR since the prologue): at NT = 4 the thread-id register was the one value that did not fit
        // beside the accumulators across the chunk loop and went to scratch
        const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), wave = swave, fr = lane & 31;
        float* tb = Es + wave * (32 * 36);              // 32 rows x 36 floats per wave: 7 x 4608 B <= the 32 KB grid
        char* ob = a.out + (size_t)b0 * HW * a.Cout * 4;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            if (q * 32 < a.Cout) {
                const int co = q * 32 + fr;
                const float s3 = co < a.Cout ? a.s3[co] : 0.f, b3 = co < a.Cout ? a.b3[co] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[mfma32_row(r, lane) * 36 + fr] = (pacc[q][0][r] + pacc[q][1][r] * (1.0f / 2048.0f)) * s3 + b3;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int it = 0; it < 2; ++it) {        // 32 rows x 4 groups = 128 items over 64 lanes
                    const int item = it * 64 + lane, row = item >> 2, g8 = q * 4 + (item & 3);
                    const int r = wave * 32 + row;
                    if (r < vrows && g8 * 8 < a.Cout) {
                        // whole-image mode: output row r = pixel r of the workgroup's images, residual = x row r; halo-tile mode: interior pixel (iy, ix) of tile
                        // (tx, ty), residual = the halo row of that pixel
                        const int iy = r / 14, ix = r - iy * 14;
                        const int xrow_i = TILE ? (iy + 1) * 16 + ix + 1 : r;
                        const size_t opix = TILE ? (size_t)(ty * 14 + iy) * a.W + tx * 14 + ix : (size_t)r;
                        float v[8];
                        *(f32x4*)v = *(const f32x4*)(tb + row * 36 + (item & 3) * 8);
                        *(f32x4*)(v + 4) = *(const f32x4*)(tb + row * 36 + (item & 3) * 8 + 4);
                        if (a.residual) {
                            const char* xr = Xs + xrow_i * strideX + g8 * 32;
                            const half8 hi = *(const half8*)xr, lo = *(const half8*)(xr + 16);
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] += (float)hi[k] + (float)lo[k] * (1.0f / 2048.0f);
                        }
                        half8 hi, lo;
                        rng.see8(v);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            _Float16 h, l;
                            smirk_split1(v[k], h, l);
                            hi[k] = h; lo[k] = l;
                        }
                        char* o = ob + (opix * a.Cout + g8 * 8) * 4;
                        *(half8*)o = hi;
                        *(half8*)(o + 16) = lo;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    rng.commit();
}

struct MbiPlan { int ipw, rows, gsz, nb, off_es, off_wp, off_wc, off_gmap, tile, tiles_x, tiles_y, gpitch, rowsE, nbE, off_gmapE; size_t lds; };
static bool mbi_plan(int H, int W, int Cin, int mid, int Cout, MbiPlan& p) {
    if (H <= 0 || W <= 0 || Cin % 8 || mid % 8 || Cout % 8 || Cin < 16 || Cin > 112 || Cout > 128 || mid <= 0 || Cout <= 0) return false;
    const int KS = (Cin + 15) / 16;
    p.tile = H * W > 224 ? 1 : 0;
    if (p.tile) {                                       // 14 x 14 tiles with a 16 x 16 halo
        if (H % 14 || W % 14) return false;
        p.tiles_x = W / 14; p.tiles_y = H / 14; p.gpitch = 16;
        p.ipw = 1; p.rows = 196; p.gsz = 256; p.nb = 7; p.rowsE = 256; p.nbE = 8;
    } else {
        p.tiles_x = p.tiles_y = 1; p.gpitch = W + 2;
        p.ipw = 224 / (H * W);
        if (p.ipw > 8) p.ipw = 8;
        p.rows = p.ipw * H * W;
        p.gsz = p.ipw * (H + 2) * (W + 2);
        p.nb = (p.rows + 31) / 32;
        p.rowsE = p.rows; p.nbE = p.nb;
    }
    if (p.nb > 7 || p.gsz * 128 < p.nb * 32 * 36 * 4) return false;   // one output row block per wave (the epilogue's transposes must fit the grid)
    size_t o = (size_t)p.rowsE * (KS * 64 + 16);
    o = (o + 15) / 16 * 16; p.off_es = (int)o; o += (size_t)p.gsz * 128;
    p.off_wp = (int)o; o += (size_t)2 * Cout * 144;
    p.off_wc = (int)o; o += 2 * 352 * 4;
    p.off_gmap = (int)o; o += (size_t)p.nb * 32 * 2;
    p.off_gmapE = p.off_gmap;                           // whole-image mode: expand rows = output rows, one map serves both
    if (p.tile) { p.off_gmapE = (int)o; o += (size_t)p.nbE * 32 * 2; }
    p.lds = (o + 15) / 16 * 16;
    return p.lds <= 160 * 1024;
}

// (ceil(Cin / 16), ceil(Cout / 32)) pairs that are instantiated: 80 -> 80, 80 -> 112, 96 -> 96, 112 -> 112 are the 14 x 14 / 7 x 7 blocks of the two backbones;
// <2, 1> (24 -> 24) and <3, 2> (40 -> 40, 40 -> 48, 48 -> 48) the stride-1 blocks of the 56 x 56 / 28 x 28 stages (halo tiles) and of the small backbone's 14 x 14 stage
static bool mbi_has_variant(int ks, int nt) {
    return (ks == 5 && (nt == 2 || nt == 3 || nt == 4)) || (ks == 6 && (nt == 3 || nt == 4)) || (ks == 7 && (nt == 3 || nt == 4)) ||
           (ks == 4 && (nt == 2 || nt == 3)) || (ks == 2 && nt == 1) || (ks == 3 && nt == 2);
}

/* 1 if smirk_mbconv_image_split16 serves this InvertedResidual block (stride 1, whole images per workgroup) */
extern "C" int smirk_mbconv_image_supported(int H, int W, int Cin, int mid, int Cout, int stride) {
    MbiPlan p;
    if (stride != 1 || !mbi_plan(H, W, Cin, mid, Cout, p)) return 0;
    const int ks = (Cin + 15) / 16, nt = (Cout + 31) / 32;
    if (!mbi_has_variant(ks, nt)) return 0;
    const bool small = (ks == 2 && nt == 1) || (ks == 3 && nt == 2);       // halo tiles and Cin % 16 != 0 are instantiated for these two only
    return (small || (!p.tile && Cin % 16 == 0)) ? 1 : 0;
}

template <int KS, int NT, bool TILE = false, bool PADK = false>
static int mbi_launch(const MBIArgs& a, unsigned grid, size_t lds, hipStream_t st, double flop, double bytes) {
    static bool attr_done[64] = {};                      // hipFuncSetAttribute is per-device state
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return SMIRK_ERR_UNSUPPORTED;
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute((const void*)mbconv_image_kernel<KS, NT, TILE, PADK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return SMIRK_ERR_LAUNCH;
        attr_done[dev] = true;
    }
    if (g_smirk_prof_on) {
        char nm[64];
        if (TILE || PADK) snprintf(nm, sizeof(nm), "mbconv_image_kernel<%d,%d,%s,%s>", KS, NT, TILE ? "true" : "false", PADK ? "true" : "false");
        else snprintf(nm, sizeof(nm), "mbconv_image_kernel<%d,%d>", KS, NT);
        smirk_prof_next(nm, flop, bytes);
    }
    SMIRK_LAUNCH((mbconv_image_kernel<KS, NT, TILE, PADK>), dim3(grid), dim3(512), lds, st, a);
    return smirk_launch_status();
}

extern "C" int smirk_mbconv_image_split16(const void* x, const void* wexp, const float* s1, const float* b1, const float* wdw, const float* s2,
                                          const float* b2, const void* wproj, const float* s3, const float* b3, int residual, void* out, int B,
                                          int H, int W, int Cin, int mid, int Cout, void* stream) {
    if (!x || !wexp || !s1 || !b1 || !wdw || !s2 || !b2 || !wproj || !s3 || !b3 || !out || B <= 0) return SMIRK_ERR_BAD_ARG;
    if (residual && Cin != Cout) return SMIRK_ERR_BAD_ARG;
    MbiPlan p;
    if (!mbi_plan(H, W, Cin, mid, Cout, p) || !smirk_mbconv_image_supported(H, W, Cin, mid, Cout, 1)) return SMIRK_ERR_UNSUPPORTED;
    MBIArgs a;
    a.x = (const char*)x; a.wexp = (const char*)wexp; a.s1 = s1; a.b1 = b1; a.wdw = wdw; a.s2 = s2; a.b2 = b2; a.wproj = (const char*)wproj;
    a.s3 = s3; a.b3 = b3; a.out = (char*)out; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.mid = mid; a.Cout = Cout; a.residual = residual;
    a.ipw = p.ipw; a.rows = p.rows; a.gsz = p.gsz; a.nb = p.nb; a.off_es = p.off_es; a.off_wp = p.off_wp; a.off_wc = p.off_wc; a.off_gmap = p.off_gmap;
    a.tile = p.tile; a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.gpitch = p.gpitch; a.rowsE = p.rowsE; a.nbE = p.nbE; a.off_gmapE = p.off_gmapE;
    const unsigned grid = p.tile ? (unsigned)B * p.tiles_x * p.tiles_y : (unsigned)((B + p.ipw - 1) / p.ipw);
    hipStream_t st = (hipStream_t)stream;
    const double px = (double)B * H * W;
    const double flop = 2.0 * px * ((double)Cin * mid + 9.0 * mid + (double)mid * Cout);
    const double bytes = 4.0 * px * (Cin * (residual ? 2 : 1) + Cout);
    const int ks = (Cin + 15) / 16, nt = (Cout + 31) / 32;
#define MBI_CASE(K, N) if (ks == K && nt == N) return mbi_launch<K, N>(a, grid, p.lds, st, flop, bytes)
    MBI_CASE(5, 3); MBI_CASE(5, 4); MBI_CASE(6, 3); MBI_CASE(7, 4);          // mbi_has_variant lists the same pairs
    MBI_CASE(4, 2); MBI_CASE(4, 3); MBI_CASE(5, 2); MBI_CASE(6, 4); MBI_CASE(7, 3);
    if (ks == 2 && nt == 1) return p.tile ? mbi_launch<2, 1, true, true>(a, grid, p.lds, st, flop, bytes) : mbi_launch<2, 1, false, true>(a, grid, p.lds, st, flop, bytes);
    if (ks == 3 && nt == 2) return p.tile ? mbi_launch<3, 2, true, true>(a, grid, p.lds, st, flop, bytes) : mbi_launch<3, 2, false, true>(a, grid, p.lds, st, flop, bytes);
#undef MBI_CASE
    return SMIRK_ERR_UNSUPPORTED;
}
