// 3x3 stride-1 zero-padded convolution for LARGE images with FEW channels (the U-Net's 224x224 / 112x112 layers) on MI355X,
// split-fp16 (f16x3) arithmetic — see conv.hip for the number format.
//
// For these layers an im2col GEMM is bound by operand movement, not MFMA: N = Cout is only 32-64 wide, so every input element
// fetched into the CU feeds very few MACs and the 9 taps re-fetch it 9 times.  Here a workgroup owns a 16x16-pixel output patch:
//   * the (16+2)x(16+2) input halo patch of one 32-channel chunk is DMA'd ONCE into LDS (global_load_lds_dwordx4, XOR piece
//     swizzle, zero page for out-of-image halo pixels) and all 9 taps read it from there  -> 1.27x the unique bytes instead of 9x;
//   * the layer's whole weight tensor stays resident in LDS ([chunk][tap][Cout][32 dwords], <= 72 KiB) because workgroups are
//     PERSISTENT: 256 of them (one per CU) walk the patch list, double-buffering the next halo patch under the current MFMAs;
//   * 4 waves x (2 M-tiles of 32 pixels) x (Cout/32 N-tiles); K = 9 taps x 32 channels per chunk = 18 MFMA k-steps of 16;
//   * epilogue: BatchNorm(eval) scale/shift + ReLU, re-split to fp16 pairs, transposed through LDS, 32-byte stores per lane.
// A concatenated input (decoder: up ++ skip) is two source tensors, one chunk sequence each.
//
// Bound: HBM (activation read 1.27x + write 1x) once the DMA is hidden; MFMA work is ~40 % of that time.
#include <type_traits>
#include <stdlib.h>

#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define PT 16                    // output patch edge
#define PH (PT + 2)              // halo patch edge
#define PPIX (PH * PH)           // 324 halo pixels
#define PSTAGE (((PPIX + 7) / 8 * 8) * 32)   // dwords per input stage (one 32-channel chunk); whole 8-pixel DMA instructions => 328 pixels

__device__ __attribute__((aligned(16))) float g_zero16_patch[4] = {0.f, 0.f, 0.f, 0.f};
#define g_zero16 g_zero16_patch

// Workgroup barrier / wave-local fence that wait for LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits for the output
// stores just issued (write latency, 2-3 times per patch) and for the next patch's operand DMA — none of which the epilogue depends on.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// branch-free pointer select (hipcc turns `c ? p : q` feeding a DMA into exec-masked branches around each load: measured 3 branches per
// halo row, ~400 scalar/vector instructions per patch and wave before this)
__device__ __forceinline__ const float* psel_p(bool c, const float* p, const float* q) {
    const unsigned long long m = 0ull - (unsigned long long)c;
    return (const float*)(((unsigned long long)p & m) | ((unsigned long long)q & ~m));
}


// buffer-addressed halo DMA (see conv.hip KW_LEAN): the per-lane part of a halo pixel's address is (patch-relative pixel << log2(4 C)) + piece,
// constant across patches; the patch origin is scalar; pixels outside the image carry an out-of-range offset and come back as zeros
#if defined(__HIP_DEVICE_COMPILE__)
#define PATCH_BUF_DECL()                                                                                                             \
    const __amdgpu_buffer_rsrc_t brs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.in0, (short)0, a.use_buf ? a.B * a.H * a.W * a.C0 * 4 : 0, 0x00020000); \
    const __amdgpu_buffer_rsrc_t brs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.C1 > 0 ? a.in1 : a.in0), (short)0,               \
                                                                          a.use_buf ? a.B * a.H * a.W * (a.C1 > 0 ? a.C1 : a.C0) * 4 : 0, 0x00020000); \
    const int bsh0 = 33 - __builtin_clz((unsigned)a.C0), bsh1 = a.C1 > 0 ? 33 - __builtin_clz((unsigned)a.C1) : bsh0;                  \
    const int swave = __builtin_amdgcn_readfirstlane(wave);
#define PATCH_BUF_LOAD(S1, LDS, VOFF, SOFF)                                                                                          \
    do { if (S1) __builtin_amdgcn_raw_ptr_buffer_load_lds(brs1, (lptr_t)(LDS), 16, VOFF, SOFF, 0, 0);                                 \
         else __builtin_amdgcn_raw_ptr_buffer_load_lds(brs0, (lptr_t)(LDS), 16, VOFF, SOFF, 0, 0); } while (0)
#else
#define PATCH_BUF_DECL() const int bsh0 = 0, bsh1 = 0, swave = 0; (void)bsh0; (void)bsh1; (void)swave;
#define PATCH_BUF_LOAD(S1, LDS, VOFF, SOFF) do { (void)(VOFF); } while (0)
#endif

struct PatchArgs {
    const float *in0, *in1, *w, *scale, *shift;
    float* out;
    int B, H, W, C0, C1, Cout;
    int nchunk;      // 32-channel chunks per patch (over both sources)
    int npatch;      // B * (H/16) * (W/16)
    int act;
    // optional fused network tail (Cout == 32 only): out_nchw[b][o][y][x] = sigmoid(sum_c relu(bn(conv))[c] * fw[o][c] + fb[o]), o < fcout;
    // the 32-channel activation is then never written to HBM (smirk_generator.py:47-49,76)
    const float *fw, *fb;
    float* fout;
    int fcout;
    long long* dbg;  // optional phase time stamps (s_memtime) of wave 0 of a few workgroups: [block][item][5]; tools/patch_timeline.py
    int use_buf;     // operand DMA through buffer resources (32-bit per-lane offsets, OOB rows for the zero padding): C0, C1 powers of two, tensors < 2 GiB
};

__device__ __forceinline__ int lds_piece_p(int row, int piece) { return row * 32 + ((piece ^ ((row >> 1) & 7)) << 2); }
// Halo stages are swizzled by the pixel's COLUMN x in the 18-wide halo row, not by its linear index: a ds_read_b128 is serviced in the lane groups
// {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (MI355X_MICROARCH.md), and lanes 0-15 / 16-31 of a fragment read are 16 consecutive pixels of halo rows
// y / y + 1.  A group therefore holds columns x0 + {0-3, 12-15} of one row and x0 + {4-11} of the other: with (x >> 1) & 7 the eight even and the eight
// odd columns of a group get eight distinct 16-byte slots in their half of the banks for every tap.  Swizzling by the linear index (18 y + x) shifts the
// pattern by one slot per row and made every A read a 2-way conflict: 38 % of this kernel's LDS cycles (profiles/r03v_pmc_census_full.txt).
__device__ __forceinline__ int halo_swz(int x) { return (x >> 1) & 7; }
__device__ __forceinline__ int lds_piece_a(int pix, int x, int piece) { return pix * 32 + ((piece ^ halo_swz(x)) << 2); }

// "The fragments of this step have arrived": an empty asm that READS them.  The compiler's waitcnt pass then places its wait here, i.e. BEFORE the next
// step's reads are issued, where `lgkmcnt(0)` is exact (only this step's reads are outstanding) — placed in front of the first MFMA instead, it emitted
// `lgkmcnt(0)` for every second step although the six newest reads belonged to the following step (ISA of the 18-step loop), exposing the LDS latency.
__device__ __forceinline__ void frag_ready(const half8& a0, const half8& a1, const half8& a2, const half8& a3, const half8& b0, const half8& b1,
                                           const half8& b2, const half8& b3) {
    asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}

template <typename RA>
__device__ __forceinline__ void split8p(const float* v, half8& hi, half8& lo, RA& rng) {
    rng.see8(v);                                                   // split-fp16 range audit (common.h): running max, tested once per kernel
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        smirk_half2 h, l;
        smirk_split2(v[q], v[q + 1], h, l);
        hi[q] = h.x; hi[q + 1] = h.y; lo[q] = l.x; lo[q + 1] = l.y;
    }
}

// NST = number of input stages: 2 = next halo patch DMA'd under the current MFMAs (one workgroup per CU when the weights are large);
// 1 = single stage, used when weights + one stage fit twice into a CU's 160 KiB: TWO workgroups per CU then cover each other's DMA
// wait / epilogue, which hides more latency than double buffering a lone 4-wave workgroup;
// 3 = RING of three stages with 16 x 8-pixel patches (PTH = 8, 23 KiB per stage): the DMA of item n+2 is issued when item n starts, so ~2 stages
// (47 KB) per CU are in flight at all times — the PMC profile of the 1- / 2-stage forms (profiles/r02_pmc_patch11.txt) shows the waves parked on
// `s_waitcnt vmcnt` / the barrier 58 % of their life with MFMA and LDS each ~27 % busy: HBM latency, not bandwidth.  The wait at the top of item n is
// COUNTED (the DMA of item n+1 may stay outstanding): loads return in order among loads, so "at most nq outstanding" proves item n has landed; stores of
// earlier epilogues can only make the wait stricter.
// GROUPS = 2: one 512-thread workgroup = two independent 4-wave pipelines (each its own patches, stages and barriers) sharing ONE copy of the weights
// in LDS — what two workgroups per CU do, minus the second 36-72 KiB weight image, which is what makes room for double-buffered stages.  The groups
// synchronise among their own four waves through an LDS arrival counter (gfx950 has one hardware barrier per workgroup and no named barriers).
__device__ __forceinline__ void group_barrier(unsigned* cnt, unsigned& epoch, int lane) {
    epoch += 4;
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

template <int TN, int NST, int PTH, int GROUPS>
__global__ __launch_bounds__(256 * GROUPS, (NST == 1 && GROUPS == 1) ? 2 : 1) void conv3x3_patch_kernel(PatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int COUT = TN * 32;
    constexpr int EPI_LD = COUT + 4;
    constexpr int PHR = PTH + 2;                                  // halo rows
    constexpr int PPIXT = PHR * PH;                               // halo pixels
    constexpr int PSTG = ((PPIXT + 7) / 8 * 8) * 32;              // dwords per input stage
    constexpr int MT = PTH / 8;                                   // 32-pixel M tiles (2 rows of 16) per wave
    constexpr int WROWS = PTH / 4;                                // output rows per wave
    static_assert(4 * 32 * EPI_LD <= PSTG, "transpose buffers must fit in one input stage");
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6, group = GROUPS == 1 ? 0 : wave_all >> 2, wave = GROUPS == 1 ? wave_all : wave_all & 3;
    unsigned* gcnt = (unsigned*)smem + group * 32;               // GROUPS == 2: arrival counters live in the first 256 bytes
    float* Wl = smem + (GROUPS == 2 ? 64 : 0);                   // [nchunk][9][COUT][32]
    float* St = Wl + a.nchunk * 9 * COUT * 32 + group * NST * PSTG;     // this group's NST input stages
    unsigned epoch = 0;
    if (GROUPS == 2 && tid < 64) ((unsigned*)smem)[tid] = 0u;
    auto gbar = [&]() {
        if (GROUPS == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        else group_barrier(gcnt, epoch, lane);
    };
    const int fr = lane & 31, hb = lane >> 5;
    const int Cin = a.C0 + a.C1, K = 9 * Cin;

    // ---- weights -> LDS once: row (cc, tap, n) = 128 bytes at w[n][tap*Cin + cc*32 ...]; pieces beyond the layer's channels are zero
    {
        const int rows = a.nchunk * 9 * COUT;
        for (int r0 = wave_all * 8; r0 < rows; r0 += 32 * GROUPS) {
            const int row = r0 + (lane >> 3), pos = lane & 7;
            const int piece = pos ^ ((row >> 1) & 7);
            const int cc = row / (9 * COUT), rem = row - cc * 9 * COUT, tap = rem / COUT, n = rem - tap * COUT;
            const int c = cc * 32 + piece * 4;
            const float* g = (row < rows && c < Cin) ? a.w + (size_t)n * K + tap * Cin + c : g_zero16;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Wl + r0 * 32), 16, 0, 0);
        }
    }
    // ---- per-lane halo pixel slots: instruction q of this wave covers halo pixels (q*4 + wave)*8 .. +7.  The slot's (row, column) is recomputed at every
    // issue (a multiply-shift division by 18) instead of being kept in 2 x 11 registers: with them the 2-workgroups-per-CU variant spilled, and the
    // scratch reloads in front of each DMA instruction made ISSUING one halo patch cost 14k cycles — 58 % of a patch's period (tools/patch_timeline.py)
    constexpr int NQ = (PPIXT + 31) / 32;
    auto hp = [&](int q, int& pix, int& y, int& x) {
        int l8 = lane >> 3;
        asm volatile("" : "+v"(l8));                              // opaque per call: nothing derived from the slot may be hoisted out of the patch loop
        pix = (q * 4 + wave) * 8 + l8;
        const int yy = (pix * 3641) >> 16;                        // pix / 18 for pix < 400
        y = pix < PPIXT ? yy : -100000;                           // slots past the halo: row far outside any image -> zero fill
        x = pix - yy * PH;
    };
    const int tiles_x = a.W / PT, tiles_per_img = (a.H / PTH) * tiles_x;
    const int nitem = a.npatch * a.nchunk;
    auto item_patch = [&](int item, int& b, int& oy0, int& ox0, int& cc) {
        const int p = item / a.nchunk;
        cc = item - p * a.nchunk;
        b = p / tiles_per_img;
        const int t = p - b * tiles_per_img;
        oy0 = (t / tiles_x) * PTH;
        ox0 = (t - (t / tiles_x) * tiles_x) * PT;
    };
    PATCH_BUF_DECL()
    // INTERIOR patches (no halo pixel outside its image: 73 % of the patches of a 224^2 map) need no per-lane address or validity arithmetic at all:
    // the lane's offset of slot q relative to the patch's halo origin is a per-launch constant (kept in NQ registers), the origin goes into the
    // buffer load's SCALAR offset.  Issuing a halo patch used to cost ~380 cycles per DMA instruction in VALU address math (4.2 k cycles per patch,
    // tools/patch_timeline.py), more than the DMA's own latency.
    unsigned loff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int pix = (q * 4 + wave) * 8 + (lane >> 3), yy = (pix * 3641) >> 16, xx = pix - yy * PH;
        const int piece = (lane & 7) ^ halo_swz(xx);
        loff[q] = pix < PPIXT ? (((unsigned)(yy * a.W + xx)) << bsh0) + (unsigned)piece * 16u : 0x80000000u;
    }
    auto issue_item = [&](int item, int st) {
        int b, oy0, ox0, cc;
        item_patch(item, b, oy0, ox0, cc);
        const int c0 = cc * 32;
        const bool s1 = c0 >= a.C0;
        const float* src = s1 ? a.in1 : a.in0;
        const int cs = s1 ? a.C1 : a.C0, cb = s1 ? c0 - a.C0 : c0;
        float* dst = St + st * PSTG;
        if (a.use_buf) {
            const int basepix = (b * a.H + oy0 - 1) * a.W + ox0 - 1, sh = s1 ? bsh1 : bsh0;
            if (oy0 > 0 && oy0 + PTH < a.H && ox0 > 0 && ox0 + PT < a.W && cb + 32 <= cs && sh == bsh0) {
                const unsigned so = (unsigned)cb * 4u + ((unsigned)basepix << sh);
                (void)so;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if ((q * 4 + swave) * 8 < PPIXT) PATCH_BUF_LOAD(s1, dst + (q * 4 + swave) * 8 * 32, loff[q], so);
                return;
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if ((q * 4 + swave) * 8 < PPIXT) {
                    int pix, hy, hx;
                    hp(q, pix, hy, hx);
                    const int piece = (lane & 7) ^ halo_swz(hx);
                    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                    const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && (cb + piece * 4) < cs;
                    const unsigned vo = ok ? ((unsigned)(basepix + hy * a.W + hx) << sh) + (unsigned)piece * 16u : 0x80000000u;
                    PATCH_BUF_LOAD(s1, dst + (q * 4 + swave) * 8 * 32, vo, cb * 4);
                }
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pixbase = (q * 4 + wave) * 8;
            if (pixbase < PPIXT) {                               // wave-uniform
                int pix, hy, hx;
                hp(q, pix, hy, hx);
                const int piece = (lane & 7) ^ halo_swz(hx);
                const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && (cb + piece * 4) < cs;
                const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);     // keep the unselected address in range
                const float* g = psel_p(ok, src + ((size_t)(b * a.H + cy) * a.W + cx) * cs + cb + piece * 4, g_zero16);
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + pixbase * 32), 16, 0, 0);
            }
        }
    };

    f32x16 acc0[MT][TN], acc1[MT][TN];
    // epilogue coefficients of THIS lane's channel group (g = lane % (COUT/8) for every item of every tile): loaded once, not once per tile — inside
    // the persistent loop each reload was a global-load round trip in front of the stores
    float ep_sc[8], ep_sh[8];
    {
        const int g = lane % (COUT / 8);
#pragma unroll
        for (int q = 0; q < 8; ++q) { ep_sc[q] = a.scale ? a.scale[g * 8 + q] : 1.f; ep_sh[q] = a.shift ? a.shift[g * 8 + q] : 0.f; }
    }
    // fused network tail: the 1x1 conv's weights [4][COUT] (zero rows beyond fcout) and biases live in LDS behind the stages, staged ONCE per workgroup — read
    // through a.fw inside the epilogue they were re-fetched from global memory for every item (the output stores may alias them as far as the compiler knows):
    // 24 dependent-latency loads in front of each of a patch's four store groups.  (Registers are not an option: this instantiation sits at 256 VGPRs.)
    float* const Tw = St + (GROUPS - group) * NST * PSTG;         // first float behind the last stage of the last group
    if (TN == 1 && a.fout) {
        for (int i = tid; i < 4 * COUT + 4; i += 256 * GROUPS) {
            const int o = i / COUT, c = i - o * COUT;
            Tw[i] = i < 4 * COUT ? (o < a.fcout ? a.fw[o * COUT + c] : 0.f) : ((i - 4 * COUT) < a.fcout && a.fb ? a.fb[i - 4 * COUT] : 0.f);
        }
    }
    const int ry = fr >> 4, rx = fr & 15;
    int item = (blockIdx.x * GROUPS + group) * a.nchunk;         // each 4-wave pipeline walks whole patches
    const int stride = gridDim.x * GROUPS * a.nchunk;
    if (GROUPS == 2) __syncthreads();                            // the zeroed counters are visible before any arrival
    auto next_item = [&](int it) { return (it % a.nchunk == a.nchunk - 1) ? it - (a.nchunk - 1) + stride : it + 1; };
    if (item < nitem) issue_item(item, 0);
    if (NST == 3 && item < nitem && next_item(item) < nitem) issue_item(next_item(item), 1);
    // DMA instructions one item costs THIS wave (the ring's counted wait): q with (4 q + wave) * 8 < PPIXT
    int nq_w = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) nq_w += ((q * 4 + swave) * 8 < PPIXT) ? 1 : 0;
    int st = 0, dbg_n = 0;
    bool first = true;
    SmirkRangeAcc rng;
    if (GROUPS == 2 && !(item < nitem)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // idle group: still owes its weight share + the barrier
    while (item < nitem) {
        int b, oy0, ox0, cc;
        item_patch(item, b, oy0, ox0, cc);
        const int nxt = next_item(item);
        const bool stamp = a.dbg && wave_all == 0 && lane == 0 && (blockIdx.x & 63) == 0 && dbg_n < 48;
        long long* dbgp = stamp ? a.dbg + ((size_t)(blockIdx.x >> 6) * 48 + dbg_n) * 6 : nullptr;
        if (stamp) { dbgp[0] = __builtin_amdgcn_s_memtime(); ++dbg_n; }
        if (NST == 3 && nxt < nitem) {                            // item + 1 may stay in flight
            if (nq_w == NQ) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NQ) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NQ - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // + this wave's LDS reads of the stage about to be refilled
        }
        if (GROUPS == 2 && first) { __syncthreads(); first = false; }   // the OTHER group's share of the weight DMA has landed too
        else gbar();
        // stage st (and, first time, the weights) has landed
        if (stamp) dbgp[1] = __builtin_amdgcn_s_memtime();
        if (NST == 2 && nxt < nitem) issue_item(nxt, st ^ 1);    // the other stage is free: refill it under the MFMAs
        if (NST == 3 && nxt < nitem) {
            const int nn = next_item(nxt);
            if (nn < nitem) issue_item(nn, st == 0 ? 2 : st - 1);     // (st + 2) % 3: the stage item - 1 has just left
        }
        if (cc == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
        }
        const float* A = St + st * PSTG;
        const float* Wc = Wl + cc * 9 * COUT * 32;
        // 18 (tap, 16-k step) stages, software-pipelined in registers: the LDS reads of stage t+1 are issued before the MFMAs of stage t
        {
            half8 ah[2][MT], al[2][MT], bh[2][TN], bl[2][TN];
            auto fetch = [&](int t, int set) {
                const int tap = t >> 1, sstep = t & 1, ky = tap / 3, kx = tap % 3;
                const int pc = 2 * (2 * sstep + hb);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int pix = (WROWS * wave + 2 * i + ry + ky) * PH + rx + kx;
                    ah[set][i] = *(const half8*)(A + lds_piece_a(pix, rx + kx, pc));
                    al[set][i] = *(const half8*)(A + lds_piece_a(pix, rx + kx, pc + 1));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = tap * COUT + j * 32 + fr;
                    bh[set][j] = *(const half8*)(Wc + lds_piece_p(row, pc));
                    bl[set][j] = *(const half8*)(Wc + lds_piece_p(row, pc + 1));
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int t = 0; t < 18; ++t) {
                const int cur = t & 1;
                frag_ready(ah[cur][0], al[cur][0], ah[cur][MT - 1], al[cur][MT - 1], bh[cur][0], bl[cur][0], bh[cur][TN - 1], bl[cur][TN - 1]);
                if (t + 1 < 18) fetch(t + 1, cur ^ 1);           // in flight under this step's 3 * MT * TN MFMAs
                __builtin_amdgcn_sched_barrier(0);               // keep it that way: the machine scheduler otherwise sinks each read next to its MFMA
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bh[cur][j], acc0[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bl[cur][j], acc1[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][i], bh[cur][j], acc1[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);               // the next step's reads stay behind this step's MFMAs in program order
            }
        }
        if (stamp) dbgp[2] = __builtin_amdgcn_s_memtime();
        if (cc == a.nchunk - 1) {
            // ---- epilogue: this stage's LDS is dead now (next DMA went to the other stage): use it as per-wave transpose buffers
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gbar();
            float* ebuf = St + st * PSTG + wave * 32 * EPI_LD;
            constexpr int GPR = COUT / 8, ITEMS = 32 * GPR / 64;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float x = acc0[i][j][r] + acc1[i][j][r] * (1.0f / 2048.0f);
                        ebuf[mfma32_row(r, lane) * EPI_LD + j * 32 + fr] = x;
                    }
                wave_lds_fence();          // per-wave transpose buffer: LDS ops of one wave execute in order
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) {
                    const int e = it * 64 + lane, row = e / GPR, g = e % GPR;
                    const int oy = oy0 + WROWS * wave + 2 * i + (row >> 4), ox = ox0 + (row & 15);
                    float v[8];
                    *(f32x4*)v = *(const f32x4*)(ebuf + row * EPI_LD + g * 8);
                    *(f32x4*)(v + 4) = *(const f32x4*)(ebuf + row * EPI_LD + g * 8 + 4);
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
                    if (TN == 1 && a.fout) {
                        // fused 1x1 conv + sigmoid: the 4 lanes e..e+3 hold one pixel's 4 channel groups -> partial dots + 2 xor shuffles
                        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int o = 0; o < 4; ++o) {                 // outputs >= fcout carry zero weights
                            const f32x4 w0 = *(const f32x4*)(Tw + o * COUT + g * 8), w1 = *(const f32x4*)(Tw + o * COUT + g * 8 + 4);
#pragma unroll
                            for (int q = 0; q < 4; ++q) { part[o] = fmaf(v[q], w0[q], part[o]); part[o] = fmaf(v[4 + q], w1[q], part[o]); }
                        }
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            part[o] += __shfl_xor(part[o], 1, 64);
                            part[o] += __shfl_xor(part[o], 2, 64);
                        }
                        if (g == 0) {
                            const size_t HW = (size_t)a.H * a.W;
#pragma unroll
                            for (int o = 0; o < 4; ++o) {
                                if (o < a.fcout) {
                                    const float z = part[o] + Tw[4 * COUT + o];
                                    a.fout[((size_t)b * a.fcout + o) * HW + (size_t)oy * a.W + ox] = 1.0f / (1.0f + expf(-z));
                                }
                            }
                        }
                    } else {
                        half8 hi, lo;
                        split8p(v, hi, lo, rng);
                        float* o = a.out + (((size_t)b * a.H + oy) * a.W + ox) * COUT + g * 8;
                        *(half8*)o = hi;
                        *(half8*)(o + 4) = lo;
                    }
                }
                wave_lds_fence();          // per-wave transpose buffer: LDS ops of one wave execute in order
            }
        }
        if (stamp) dbgp[3] = __builtin_amdgcn_s_memtime();
        if (NST == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gbar();                                               // everyone is done with the only stage (compute or epilogue)
            if (stamp) dbgp[4] = __builtin_amdgcn_s_memtime();
            if (nxt < nitem) issue_item(nxt, 0);
            if (stamp) dbgp[5] = __builtin_amdgcn_s_memtime();
        } else if (NST == 2) {
            st ^= 1;
        } else {
            st = st == 2 ? 0 : st + 1;
        }
        item = nxt;
    }
    rng.commit();
}

// Streamed-weights variant for Cout = 64 layers whose weight tensor does not fit in LDS next to the input stages (64->64, 128->64 at 112x112).
// The 64 output channels are processed as two halves of 32 (one N-tile each): a pipeline ITEM is (patch, 32-channel chunk, half); the
// half-chunk weights [9 taps][32][32 dwords] = 36 KiB are double-buffered like the input halo stages, so 2 x 36 + 2 x 41 = 154 KiB of the
// CU's 160 KiB LDS are in use.  While item i computes, the DMA for item i+1 (its weights, and its input stage when it starts a new chunk)
// is in flight.  Both halves' accumulators live in registers across the chunks of a patch; one epilogue per patch writes all 64 channels.
#define WSTAGE (9 * 32 * 32)     // dwords per weight stage (one half-chunk)
__global__ __launch_bounds__(256, 1) void conv3x3_patch_stream_kernel(PatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int COUT = 64, EPI_LD = COUT + 4;
    static_assert(4 * 32 * EPI_LD <= PSTAGE, "transpose buffers must fit in one input stage");
    float* Wst = smem;                                           // 2 weight stages
    float* St = smem + 2 * WSTAGE;                               // 2 input stages
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, hb = lane >> 5;
    const int Cin = a.C0 + a.C1, K = 9 * Cin;

    constexpr int NQ = (PPIX + 31) / 32;
    auto hp = [&](int q, int& pix, int& y, int& x) {              // halo slot -> (row, column), recomputed per issue (see conv3x3_patch_kernel)
        int l8 = lane >> 3;
        asm volatile("" : "+v"(l8));
        pix = (q * 4 + wave) * 8 + l8;
        const int yy = (pix * 3641) >> 16;
        y = pix < PPIX ? yy : -100000;
        x = pix - yy * PH;
    };
    const int tiles_x = a.W / PT, tiles_per_img = (a.H / PT) * tiles_x;
    auto patch_origin = [&](int p, int& b, int& oy0, int& ox0) {
        b = p / tiles_per_img;
        const int t = p - b * tiles_per_img;
        oy0 = (t / tiles_x) * PT;
        ox0 = (t - (t / tiles_x) * tiles_x) * PT;
    };
    PATCH_BUF_DECL()
    unsigned loff[NQ];                                           // interior patches: per-lane slot offsets, origin in the scalar offset (see conv3x3_patch_kernel)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int pix = (q * 4 + wave) * 8 + (lane >> 3), yy = (pix * 3641) >> 16, xx = pix - yy * PH;
        const int piece = (lane & 7) ^ halo_swz(xx);
        loff[q] = pix < PPIX ? (((unsigned)(yy * a.W + xx)) << bsh0) + (unsigned)piece * 16u : 0x80000000u;
    }
    auto issue_input = [&](int p, int cc, int st) {
        int b, oy0, ox0;
        patch_origin(p, b, oy0, ox0);
        const int c0 = cc * 32;
        const bool s1 = c0 >= a.C0;
        const float* src = s1 ? a.in1 : a.in0;
        const int cs = s1 ? a.C1 : a.C0, cb = s1 ? c0 - a.C0 : c0;
        float* dst = St + st * PSTAGE;
        if (a.use_buf) {
            const int basepix = (b * a.H + oy0 - 1) * a.W + ox0 - 1, sh = s1 ? bsh1 : bsh0;
            if (oy0 > 0 && oy0 + PT < a.H && ox0 > 0 && ox0 + PT < a.W && sh == bsh0) {
                const unsigned so = (unsigned)cb * 4u + ((unsigned)basepix << sh);
                (void)so;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if ((q * 4 + swave) * 8 < PPIX) PATCH_BUF_LOAD(s1, dst + (q * 4 + swave) * 8 * 32, loff[q], so);
                return;
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if ((q * 4 + swave) * 8 < PPIX) {
                    int pix_, hy, hx;
                    hp(q, pix_, hy, hx);
                    const int piece = (lane & 7) ^ halo_swz(hx);
                    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                    const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                    const unsigned vo = ok ? ((unsigned)(basepix + hy * a.W + hx) << sh) + (unsigned)piece * 16u : 0x80000000u;
                    PATCH_BUF_LOAD(s1, dst + (q * 4 + swave) * 8 * 32, vo, cb * 4);
                }
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pixbase = (q * 4 + wave) * 8;
            if (pixbase < PPIX) {
                int pix_, hy, hx;
                hp(q, pix_, hy, hx);
                const int piece = (lane & 7) ^ halo_swz(hx);
                const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);     // keep the unselected address in range
                const float* g = psel_p(ok, src + ((size_t)(b * a.H + cy) * a.W + cx) * cs + cb + piece * 4, g_zero16);
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + pixbase * 32), 16, 0, 0);
            }
        }
    };
    auto issue_weights = [&](int cc, int h, int st) {
        float* dst = Wst + st * WSTAGE;
        for (int r0 = wave * 8; r0 < 9 * 32; r0 += 32) {         // 288 rows: (tap, n), 8 rows per wave instruction
            const int row = r0 + (lane >> 3), piece = (lane & 7) ^ ((row >> 1) & 7);
            const int tap = row >> 5, n = h * 32 + (row & 31);
            __builtin_amdgcn_global_load_lds((gptr_t)(a.w + (size_t)n * K + tap * Cin + cc * 32 + piece * 4), (lptr_t)(dst + r0 * 32), 16, 0, 0);
        }
    };

    f32x16 acc0[2][2], acc1[2][2];                               // [half][M-tile]
    float ep_sc[8], ep_sh[8];                                    // this lane's epilogue coefficients (g = lane % 8), loaded once
    {
        const int g = lane % (COUT / 8);
#pragma unroll
        for (int q = 0; q < 8; ++q) { ep_sc[q] = a.scale ? a.scale[g * 8 + q] : 1.f; ep_sh[q] = a.shift ? a.shift[g * 8 + q] : 0.f; }
    }
    const int ry = fr >> 4, rx = fr & 15;
    // 18 (tap, 16-k step) stages, software-pipelined in registers: the LDS reads of stage t+1 are issued before the MFMAs of stage t
    auto run = [&](const float* A, const float* Wc, f32x16 (&c0)[2], f32x16 (&c1)[2]) {
        half8 ah[2][2], al[2][2], bh[2], bl[2];
        auto fetch = [&](int t, int set) {
            const int tap = t >> 1, sstep = t & 1, ky = tap / 3, kx = tap % 3;
            const int pc = 2 * (2 * sstep + hb);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pix = (4 * wave + 2 * i + ry + ky) * PH + rx + kx;
                ah[set][i] = *(const half8*)(A + lds_piece_a(pix, rx + kx, pc));
                al[set][i] = *(const half8*)(A + lds_piece_a(pix, rx + kx, pc + 1));
            }
            const int row = tap * 32 + fr;
            bh[set] = *(const half8*)(Wc + lds_piece_p(row, pc));
            bl[set] = *(const half8*)(Wc + lds_piece_p(row, pc + 1));
        };
        fetch(0, 0);
#pragma unroll
        for (int t = 0; t < 18; ++t) {
            const int cur = t & 1;
            frag_ready(ah[cur][0], al[cur][0], ah[cur][1], al[cur][1], bh[cur], bl[cur], bh[cur], bl[cur]);
            if (t + 1 < 18) fetch(t + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) c0[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bh[cur], c0[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) c1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bl[cur], c1[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) c1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][i], bh[cur], c1[i], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }
    };
    auto stage_ready = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // + this wave's LDS reads of the stage about to be refilled
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // pipeline over (patch, chunk); each chunk is two straight-line half items (h = 0, 1) so the accumulator sets are compile-time fixed.
    // Weight stage h holds half h of the current chunk; input stage (chunk_no & 1) holds the current chunk's halo patch.
    int chunk_no = 0;
    int p = blockIdx.x, cc = 0;
    SmirkRangeAcc rng;
    if (p < a.npatch) { issue_input(p, 0, 0); issue_weights(0, 0, 0); }
    while (p < a.npatch) {
        int np = p, ncc = cc + 1;
        if (ncc == a.nchunk) { ncc = 0; np = p + gridDim.x; }
        const float* A = St + (chunk_no & 1) * PSTAGE;
        // ---- half 0: its weights (stage 0) and the chunk's input have landed; prefetch half 1's weights into stage 1 ----------------
        stage_ready();
        issue_weights(cc, 1, 1);
        if (cc == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[0][i][r] = 0.f; acc1[0][i][r] = 0.f; acc0[1][i][r] = 0.f; acc1[1][i][r] = 0.f; }
        }
        run(A, Wst, acc0[0], acc1[0]);
        // ---- half 1: prefetch the NEXT chunk (weights half 0 into stage 0, input into the other input stage) ---------------------------
        stage_ready();
        if (np < a.npatch) { issue_weights(ncc, 0, 0); issue_input(np, ncc, (chunk_no + 1) & 1); }
        run(A, Wst + WSTAGE, acc0[1], acc1[1]);
        if (cc == a.nchunk - 1) {
            int b, oy0, ox0;
            patch_origin(p, b, oy0, ox0);
            lds_barrier();                                        // the current input stage is dead: per-wave transpose buffers
            float* ebuf = St + (chunk_no & 1) * PSTAGE + wave * 32 * EPI_LD;
            constexpr int GPR = COUT / 8, ITEMS = 32 * GPR / 64;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ebuf[mfma32_row(r, lane) * EPI_LD + j * 32 + fr] = acc0[j][i][r] + acc1[j][i][r] * (1.0f / 2048.0f);
                wave_lds_fence();          // per-wave transpose buffer: LDS ops of one wave execute in order
#pragma unroll
                for (int e0 = 0; e0 < ITEMS; ++e0) {
                    const int e = e0 * 64 + lane, row = e / GPR, g = e % GPR;
                    const int oy = oy0 + 4 * wave + 2 * i + (row >> 4), ox = ox0 + (row & 15);
                    float v[8];
                    *(f32x4*)v = *(const f32x4*)(ebuf + row * EPI_LD + g * 8);
                    *(f32x4*)(v + 4) = *(const f32x4*)(ebuf + row * EPI_LD + g * 8 + 4);
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
                    half8 hi, lo;
                    split8p(v, hi, lo, rng);
                    float* o = a.out + (((size_t)b * a.H + oy) * a.W + ox) * COUT + g * 8;
                    *(half8*)o = hi;
                    *(half8*)(o + 4) = lo;
                }
                wave_lds_fence();          // per-wave transpose buffer: LDS ops of one wave execute in order
            }
        }
        ++chunk_no;
        p = np; cc = ncc;
    }
    rng.commit();
}

// Can this layer run on the patch kernel?  (3x3, stride 1, zero pad 1, same size, H,W % 16 == 0, Cout 32/64, weights resident)
static bool patch_common(const SmirkConvDesc* d, bool has_residual) {
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1 || d->pad_mode != SMIRK_PAD_ZERO) return false;
    if (d->out_mode != SMIRK_OUT_NHWC || d->Ho != d->H || d->Wo != d->W || d->H % PT || d->W % PT || has_residual) return false;
    if (d->Cout != 32 && d->Cout != 64) return false;
    if (d->C1 > 0 && d->C0 % 32) return false;
    return d->H >= 64;
}
static bool patch_resident(const SmirkConvDesc* d) {             // whole weight tensor fits next to two input stages
    const int nchunk = (d->C0 + 31) / 32 + (d->C1 + 31) / 32;
    return ((size_t)nchunk * 9 * d->Cout * 32 + 2 * PSTAGE) * 4 <= 160 * 1024;
}
static bool patch_streamed(const SmirkConvDesc* d) {             // Cout = 64, whole 32-channel chunks: weights streamed per half-chunk
    // measured (B=128, 112x112): 128->64 0.90 ms vs 1.02 ms on the implicit-GEMM kernel, 64->64 0.53 vs 0.50 ms => only for >= 4 chunks
    return d->Cout == 64 && d->C0 % 32 == 0 && d->C1 % 32 == 0 && (d->C0 + d->C1) >= 128;
}
static bool patch_eligible(const SmirkConvDesc* d, bool has_residual) {
    return patch_common(d, has_residual) && (patch_resident(d) || patch_streamed(d));
}

int smirk_conv3x3_patch_launch(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, const float* scale,
                               const float* shift, void* out, hipStream_t st, const float* fw, const float* fb, float* fout,
                               int fcout) {
    PatchArgs a;
    a.fw = fw; a.fb = fb; a.fout = fout; a.fcout = fcout;
    {   // debugging aid: SMIRK_PATCH_DBG=<device address (hex) of a zeroed int64 buffer of >= 8*48*6 entries>, see tools/patch_timeline.py
#ifdef SMIRK_DEBUG_HOOKS                                                 /* a raw device address from the environment: variant builds only */
        const char* e = getenv("SMIRK_PATCH_DBG");
        a.dbg = e ? (long long*)strtoull(e, nullptr, 16) : nullptr;
#else
        a.dbg = nullptr;
#endif
    }
    a.in0 = (const float*)in0; a.in1 = (const float*)in1; a.w = (const float*)w; a.scale = scale; a.shift = shift; a.out = (float*)out;
    a.B = d->B; a.H = d->H; a.W = d->W; a.C0 = d->C0; a.C1 = d->C1; a.Cout = d->Cout; a.act = d->act;
    a.nchunk = (d->C0 + 31) / 32 + (d->C1 + 31) / 32;
    a.npatch = d->B * (d->H / PT) * (d->W / PT);
    {
        const long long px = (long long)d->B * d->H * d->W;
        const bool pow2 = (d->C0 & (d->C0 - 1)) == 0 && (d->C1 & (d->C1 - 1)) == 0;
        a.use_buf = (pow2 && px * d->C0 * 4 < (1ll << 31) && px * d->C1 * 4 < (1ll << 31)) ? 1 : 0;
    }
    if (g_smirk_prof_on) {
        const double px = (double)d->B * d->H * d->W, K = 9.0 * (d->C0 + d->C1);
        const double outb = fout ? px * fcout * 4.0 : px * d->Cout * 4.0;
        smirk_prof_next(nullptr, 2.0 * px * d->Cout * K + (fout ? 2.0 * px * d->Cout * fcout : 0.0),
                        px * (d->C0 + d->C1) * 4.0 + outb + K * d->Cout * 4.0);
    }
    if (!patch_resident(d)) {
        static bool attr2_dev[64] = {};                              // per device ordinal (hipFuncSetAttribute is per-device state)
        int dev2 = 0;
        (void)hipGetDevice(&dev2);
        bool& attr2 = attr2_dev[(dev2 >= 0 && dev2 < 64) ? dev2 : 0];
        if (!attr2) { (void)hipFuncSetAttribute((const void*)conv3x3_patch_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr2 = true; }
        const size_t lds2 = (size_t)(2 * WSTAGE + 2 * PSTAGE) * 4;
        SMIRK_LAUNCH(conv3x3_patch_stream_kernel, dim3(a.npatch < 256 ? a.npatch : 256), dim3(256), lds2, st, a);
        return smirk_launch_status();
    }
    const size_t wbytes = (size_t)a.nchunk * 9 * d->Cout * 32 * 4;
    static bool attr_done_dev[64] = {};                              // per device ordinal (hipFuncSetAttribute is per-device state)
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& attr_done = attr_done_dev[(dev >= 0 && dev < 64) ? dev : 0];
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3x3_patch_kernel<1, 1, 16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_patch_kernel<1, 2, 16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_patch_kernel<2, 2, 16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    // (round 4: two 4-wave groups sharing the weights on 16 x 16 patches, <1,1,16,2> with the LDS-counter group barrier: dec1a 7.38 -> 10.72 ms — removed again)
    // (rounds 2-4 kept three-stage-ring and two-group variants of this kernel behind $SMIRK_PATCH_RING: all measured 15-50 % slower than two single-stage workgroups
    // per CU — a lone 4-wave workgroup with one M tile per wave has two accumulator chains per wave and nothing else on its SIMD, so MFMA latency, not DMA latency,
    // decides (profiles/r02_patch_timeline_and_variants.txt) — and left the library in round 5 together with the switch.)
    const size_t tail = fout ? (4 * 32 + 4) * 4 : 0;                                              // the fused tail's 1x1 weights / biases
    const bool one_stage = (wbytes + (size_t)PSTAGE * 4 + tail) * 2 <= 160 * 1024 && d->Cout == 32;    // two workgroups per CU fit
    const size_t lds = wbytes + (size_t)(one_stage ? 1 : 2) * PSTAGE * 4 + tail;
    const int cap = one_stage ? 512 : 256;
    const int grid = a.npatch < cap ? a.npatch : cap;
    // (no statistics epilogue here: the <1,1,16,1> instantiation sits at 253 of its 256 VGPRs and the two running column sums would spill — round 6 tried; the three
    // 32-channel 224 x 224 layers it serves in train mode keep the stand-alone statistics pass, which streams them at HBM rate)
    if (d->Cout == 32 && one_stage) SMIRK_LAUNCH((conv3x3_patch_kernel<1, 1, 16, 1>), dim3(grid), dim3(256), lds, st, a);
    else if (d->Cout == 32) SMIRK_LAUNCH((conv3x3_patch_kernel<1, 2, 16, 1>), dim3(grid), dim3(256), lds, st, a);
    else SMIRK_LAUNCH((conv3x3_patch_kernel<2, 2, 16, 1>), dim3(grid), dim3(256), lds, st, a);
    return smirk_launch_status();
}

bool smirk_conv3x3_patch_eligible(const SmirkConvDesc* d, bool has_residual) { return patch_eligible(d, has_residual); }
