// Shared pieces of the implicit-GEMM convolution kernels (conv.hip, conv_pp.hip): argument block, GEMM-row -> pixel map, the split-fp16
// conversions and the XOR-swizzled LDS operand image.  See conv.hip for the design notes.
#pragma once
#include "common.h"

#define CV_BK 32

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct ConvArgs {
    SmirkConvDesc d;
    const float *in0, *in1, *w, *scale, *shift, *residual;   // F16X3: in0/in1/w/residual/out are split-fp16 tensors viewed as dwords
    float* out;
    int M, N, K, Cin;
    int ablate;    // reserved for ablation experiments (unused in the shipped kernels)
    float* stats;  // train mode (BatchNorm statistics from the accumulators): per-tile partial column sums [rows][N][2] = (sum z, sum z^2) in fp32, or nullptr.
                   // Every kernel family that honours it states its row count through smirk_conv_stats_rows(); the fixed-order fp64 reduction over the rows is
                   // bn_finalize_partials_kernel (train.hip)
    int psh;       // GEMM rows enumerate each image in (2^psh x 2^psh)-pixel patches (tile-major): a BM-row tile is then a compact 2-D
                   // patch whose 3x3 halo is ~1.3x its area instead of 3 full image rows — the im2col re-reads stay in L1/L2
};

// GEMM row -> (image, y, x).  Inside an image rows are ordered patch-major; any bijection is valid because every output
// address is computed from (b, y, x).
__device__ __forceinline__ void row_to_pixel(int m, int HoWo, int Wo, int psh, int& b, int& oy, int& ox) {
    b = m / HoWo;
    const int rem = m - b * HoWo;
    const int t = rem >> (2 * psh), in = rem & ((1 << (2 * psh)) - 1);
    const int tpr = Wo >> psh, ty = t / tpr, tx = t - ty * tpr;
    oy = (ty << psh) + (in >> psh);
    ox = (tx << psh) + (in & ((1 << psh) - 1));
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = (i < 0) ? -i : i;
    return (i >= n) ? (2 * n - 2 - i) : i;
}

// 8 fp32 values -> split-fp16 group: out_hi = fp16(v), out_lo = fp16((v - hi) * 2^11)
__device__ __forceinline__ void split8(const float* v, half8& hi, half8& lo) {
    smirk_range_audit8(v);
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        smirk_half2 h, l;
        smirk_split2(v[q], v[q + 1], h, l);
        hi[q] = h.x; hi[q + 1] = h.y; lo[q] = l.x; lo[q + 1] = l.y;
    }
}
// split WITHOUT an audit: for values that are the maximum / a copy of values audited a few lines earlier (the fused 2 x 2 max-pool outputs)
__device__ __forceinline__ void split8_noaudit(const float* v, half8& hi, half8& lo) {
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        smirk_half2 h, l;
        smirk_split2(v[q], v[q + 1], h, l);
        hi[q] = h.x; hi[q + 1] = h.y; lo[q] = l.x; lo[q + 1] = l.y;
    }
}
// the same split for the hot epilogues: the range audit goes into the lane's running maximum (SmirkRangeAcc, common.h), tested once per kernel
template <typename RA>
__device__ __forceinline__ void split8(const float* v, half8& hi, half8& lo, RA& ra) {
    ra.see8(v);
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        smirk_half2 h, l;
        smirk_split2(v[q], v[q + 1], h, l);
        hi[q] = h.x; hi[q + 1] = h.y; lo[q] = l.x; lo[q + 1] = l.y;
    }
}
__device__ __forceinline__ float join1(_Float16 hi, _Float16 lo) { return (float)hi + (float)lo * (1.0f / 2048.0f); }


// LDS operand image: [rows][32 dwords] (one 128-byte K chunk per row), written by global_load_lds_dwordx4 — 64 lanes x 16 B =
// 8 consecutive rows per wave instruction, lane-linear, so no padding is possible.  Bank conflicts are removed by an XOR
// swizzle applied on the SOURCE side (which 16-byte piece of the row a lane fetches) and on the read side:
// physical piece = logical piece ^ ((row >> 1) & 7)  => the 16 rows of a ds_read_b128 lane group hit 16 distinct 16-byte slots.
__device__ __forceinline__ int lds_piece(int row, int piece) { return row * 32 + ((piece ^ ((row >> 1) & 7)) << 2); }


// XCD-aware tile id: hardware places block id on XCD id%8; give each XCD a contiguous run of logical tiles
__device__ __forceinline__ int xcd_logical(int id, int nblk) {
    const int xcd = id & 7, slot = id >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
