// Renderer for MI355X (gfx950) — compiled with -ffp-contract=off: the rasteriser must reproduce the reference's
// un-fused fp32 mul/sub/add/div sequence bit for bit (pix_to_face parity), see SURVEY.md App. B.
//
//   render_project     orthographic camera + y/z flip                         (util.py:64-78, renderer.py:101-102)
//   render_normals     area-weighted vertex normals by CSR gather in the reference's index_add_ order — deterministic,
//                      no atomics                                               (util.py:30-62)
//   raster_face_setup  per face: 3 screen-space corners (sub-mesh select, z+10, xy negation) + conservative pixel bbox
//                                                                               (renderer.py:139-144,172-173)
//   raster_tile        one workgroup per 32x8 pixel tile: bin the mesh's faces against the tile into an LDS list, then every
//                      lane owns ONE pixel and walks the list (LDS-staged face records) keeping the lexicographic minimum
//                      (z, face) — the reference's K=1 rule — and finally shades its pixel: barycentric normal
//                      interpolation + 5 directional Lambert lights.  No global atomics, every output byte written once,
//                      coalesced.  Replaces pytorch3d rasterize_meshes + the [N,H,W,1,3,6] gather of renderer.py:194-206
//                      (925 MB at B=256 in the reference) + add_directionlight (renderer.py:239-250).
//
// Bound: HBM/L2 (integer + fp32 scan work).  Algorithmic bytes per face (image): verts in 60 KB, image out 602 KB.
#include <atomic>
#include <stdlib.h>

#include "common.h"

#define TILE_W 32
#define TILE_H 8
#define FACE_CHUNK 64
#define FACE_REC 24               // floats per staged face record in raster_tile (21 used; 96-byte rows: six aligned 16-byte reads)
#define K_EPS 1e-8f

struct MeshDev {
    int V, Vf, Ff, nnz;
    const int32_t *keep, *faces, *nrm_ptr, *nrm_face, *nrm_corner;
};

__global__ __launch_bounds__(256) void render_project(const float* __restrict__ verts, const float* __restrict__ cam,
                                                      int B, int V, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * V) return;
    const int b = (int)(i / V);
    const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    const float x = verts[i * 3 + 0], y = verts[i * 3 + 1], z = verts[i * 3 + 2];
    out[i * 3 + 0] = s * (x + tx);
    out[i * 3 + 1] = -(s * (y + ty));
    out[i * 3 + 2] = -(s * z);
}

__global__ __launch_bounds__(256) void project_landmarks(const float* __restrict__ lmk, const float* __restrict__ cam,
                                                         int B, int L, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * L) return;
    const int b = (int)(i / L);
    const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    out[i * 2 + 0] = s * (lmk[i * 3 + 0] + tx);
    out[i * 2 + 1] = -(s * (lmk[i * 3 + 1] + ty));
}

__device__ inline void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__global__ __launch_bounds__(256) void render_normals(MeshDev m, int B, const float* __restrict__ verts,
                                                      float* __restrict__ normals) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * m.Vf) return;
    const int b = (int)(i / m.Vf), vi = (int)(i % m.Vf);
    const float* vb = verts + (size_t)b * m.V * 3;
    float n[3] = {0.f, 0.f, 0.f};
    for (int e = m.nrm_ptr[vi]; e < m.nrm_ptr[vi + 1]; ++e) {
        const int f = m.nrm_face[e], c = m.nrm_corner[e];
        float p[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int g = m.keep[m.faces[f * 3 + k]];
#pragma unroll
            for (int d = 0; d < 3; ++d) p[k][d] = vb[g * 3 + d];
        }
        // corner c contributes cross(p[c+2] - p[c], p[c+1] - p[c])... spelled per corner as in util.py:52-57
        const int i0 = c, i1 = (c + 1) % 3, i2 = (c + 2) % 3;
        float a[3], bb[3], cr[3];
        // corner1: cross(v2-v1, v0-v1); corner2: cross(v0-v2, v1-v2); corner0: cross(v1-v0, v2-v0)  == cross(p[i1]-p[i0], p[i2]-p[i0])
#pragma unroll
        for (int d = 0; d < 3; ++d) { a[d] = p[i1][d] - p[i0][d]; bb[d] = p[i2][d] - p[i0][d]; }
        cross3(a, bb, cr);
        n[0] += cr[0]; n[1] += cr[1]; n[2] += cr[2];
    }
    const float nrm = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    const float den = fmaxf(nrm, 1e-6f);                                 // F.normalize(eps=1e-6)
    normals[i * 3 + 0] = n[0] / den;
    normals[i * 3 + 1] = n[1] / den;
    normals[i * 3 + 2] = n[2] / den;
}

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}
__device__ __forceinline__ float pix_to_ndc(int i, int S) { return -1.0f + (2.0f * (float)i + 1.0f) / (float)S; }

// conservative pixel-index box (xi0,xi1,yi0,yi1) of a face record; empty (xi0 > xi1) for faces the reference skips wholesale
__device__ inline short4 face_pixel_box(const float* p, int H, int W) {
    const float xmin = fminf(p[0], fminf(p[3], p[6])), xmax = fmaxf(p[0], fmaxf(p[3], p[6]));
    const float ymin = fminf(p[1], fminf(p[4], p[7])), ymax = fmaxf(p[1], fmaxf(p[4], p[7]));
    const float zmin = fminf(p[2], fminf(p[5], p[8]));                    // z_invalid = zmin < kEpsilon: a face with ANY vertex not in front of the camera is dropped whole
    const float area = edge_fn(p[0], p[1], p[3], p[4], p[6], p[7]);
    short4 box = make_short4(1, 0, 1, 0);
    const bool finite = (xmin == xmin) && (xmax == xmax) && (ymin == ymin) && (ymax == ymax);
    if (finite && fabsf(area) > K_EPS && !(zmin < K_EPS)) {
        // pixel column xi sees ndc(W-1-xi); ndc(i) = -1 + (2i+1)/W  =>  i in [ (W(xmin+1)-1)/2 , (W(xmax+1)-1)/2 ], widened by 1
        float ilo = floorf((W * (xmin + 1.0f) - 1.0f) * 0.5f) - 1.0f, ihi = ceilf((W * (xmax + 1.0f) - 1.0f) * 0.5f) + 1.0f;
        float jlo = floorf((H * (ymin + 1.0f) - 1.0f) * 0.5f) - 1.0f, jhi = ceilf((H * (ymax + 1.0f) - 1.0f) * 0.5f) + 1.0f;
        ilo = fminf(fmaxf(ilo, 0.f), (float)(W - 1)); ihi = fminf(fmaxf(ihi, -1.f), (float)(W - 1));
        jlo = fminf(fmaxf(jlo, 0.f), (float)(H - 1)); jhi = fminf(fmaxf(jhi, -1.f), (float)(H - 1));
        if (!(xmax < -1.0f || xmin > 1.0f || ymax < -1.0f || ymin > 1.0f))
            box = make_short4((short)(W - 1 - (int)ihi), (short)(W - 1 - (int)ilo), (short)(H - 1 - (int)jhi),
                              (short)(H - 1 - (int)jlo));
    }
    return box;
}

// ---- per-frame face setup, sorted front to back -------------------------------------------------------------------------------------------------------------
// Record of a face in the workspace (FREC_G = 12 floats): x0,y0,z0,x1,y1,z1,x2,y2,z2 in pytorch3d NDC (sub-mesh select, xy negation, z + 10: renderer.py:139-144,
// 172-173), zlow, the face id (int bits), pad; bbox: conservative pixel-index box (xi0,xi1,yi0,yi1), empty (xi0 > xi1) for faces the reference skips wholesale
// (|area| <= eps, zmin < eps).  Records are stored in ASCENDING zlow order (valid faces first; nvalid[b] of them).
//
// zlow is a proven lower bound of the depth pz the per-pixel code of raster_tile can compute for ANY pixel this face covers:
//     pz = fl(fl(fl(w0 z0) + fl(w1 z1)) + fl(w2 z2)),  w_k = fl(e_k / A) > 0,  z_k >= zmin   =>   pz >= zmin (w0 + w1 + w2) (1 - 4u),      u = 2^-24
//     e_k (computed, -ffp-contract=off) differs from the exact edge function by <= 8u ext^2 for a pixel inside the face's box (ext = box width + height in NDC),
//     the exact edge functions sum to the exact doubled area, and A = fl(area_computed + 1e-8)   =>   w0 + w1 + w2 >= 1 - 2u - (1e-8 + 32u ext^2) / |A|
// zlow = zmin (1 - rho)(1 - 2e-6) with rho = (1e-8 + 128u ext^2) / |A| (4x the derived error term), and 0 when rho is not small or z is tiny.  A face whose zlow
// is STRICTLY above the current depth of every pixel of a block cannot win the (pz, face) lexicographic minimum anywhere in it: skipping it leaves pix_to_face,
// the barycentrics and the z-buffer bit-identical (the reference's K = 1 result does not depend on the order faces are visited in).  On SURVEY 8(d)'s synthetic
// basis the projected mesh is a crumpled sheet (3408 triangles of ~240 px^2 over a 17 k px silhouette: ~24 layers), which is what made the face loop 5 % of the step.
#define FREC_G 12
#define SORT_N 4096               // faces per frame the in-LDS bitonic sort handles (SMIRK's sub-mesh: 3408); larger meshes keep the mesh order and zlow = 0

__device__ __forceinline__ void face_corners(const MeshDev& m, const float* tb, int f, float* p) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int g = m.keep[m.faces[f * 3 + k]];
        p[k * 3 + 0] = -tb[g * 3 + 0];
        p[k * 3 + 1] = -tb[g * 3 + 1];
        p[k * 3 + 2] = tb[g * 3 + 2] + 10.0f;
    }
}
__device__ __forceinline__ float face_zlow(const float* p) {
    const float xmin = fminf(p[0], fminf(p[3], p[6])), xmax = fmaxf(p[0], fmaxf(p[3], p[6]));
    const float ymin = fminf(p[1], fminf(p[4], p[7])), ymax = fmaxf(p[1], fmaxf(p[4], p[7]));
    const float zmin = fminf(p[2], fminf(p[5], p[8]));
    const float A = edge_fn(p[6], p[7], p[0], p[1], p[3], p[4]) + K_EPS;           // the value raster_tile divides by
    const float ext = (xmax - xmin) + (ymax - ymin);
    const float rho = (1e-8f + 7.63e-6f * ext * ext) / fabsf(A);                  // 128 u = 7.63e-6
    if (!(rho < 0.25f) || !(zmin > 1e-3f)) return 0.f;                             // (NaN / inf / degenerate faces land here too)
    return zmin * (1.0f - rho) * (1.0f - 2e-6f);
}
__device__ __forceinline__ void store_face(float* frec, short4* fbox, size_t slot, const float* p, float zlow, int f, short4 box) {
    float* r = frec + slot * FREC_G;
    *(float4*)r = make_float4(p[0], p[1], p[2], p[3]);
    *(float4*)(r + 4) = make_float4(p[4], p[5], p[6], p[7]);
    *(float4*)(r + 8) = make_float4(p[8], zlow, __int_as_float(f), 0.f);
    fbox[slot] = box;
}

// meshes beyond SORT_N faces: mesh order, no depth bound (the tile kernel then visits every binned face, as rounds 1-4 did)
__global__ __launch_bounds__(256) void raster_face_setup(MeshDev m, int B, int H, int W, const float* __restrict__ tv,
                                                         float* __restrict__ frec, short4* __restrict__ fbox, int* __restrict__ nvalid) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * m.Ff) return;
    const int b = (int)(i / m.Ff), f = (int)(i % m.Ff);
    float p[9];
    face_corners(m, tv + (size_t)b * m.V * 3, f, p);
    store_face(frec, fbox, i, p, 0.f, f, face_pixel_box(p, H, W));
    if (f == 0) nvalid[b] = m.Ff;
}

// one workgroup per frame: keys (zlow bits << 32 | face) of the valid faces, bitonic sort in LDS, records written in sorted order
__global__ __launch_bounds__(512) void raster_face_setup_sorted(MeshDev m, int B, int H, int W, const float* __restrict__ tv,
                                                                float* __restrict__ frec, short4* __restrict__ fbox, int* __restrict__ nvalid) {
    __shared__ unsigned long long key[SORT_N];
    __shared__ int cnt;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* tb = tv + (size_t)b * m.V * 3;
    if (tid == 0) cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int f = tid; f < SORT_N; f += 512) {
        unsigned long long k = ~0ull;                                               // invalid / padding: sorts last
        if (f < m.Ff) {
            float p[9];
            face_corners(m, tb, f, p);
            const short4 box = face_pixel_box(p, H, W);
            if (box.x <= box.y) { k = ((unsigned long long)__float_as_uint(face_zlow(p)) << 32) | (unsigned)f; ++mine; }   // zlow >= 0: its bits order like the value
        }
        key[f] = k;
    }
    if (mine) atomicAdd(&cnt, mine);
    __syncthreads();
    for (int k2 = 2; k2 <= SORT_N; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < SORT_N / 2; t += 512) {
                const int i0 = ((t & ~(j - 1)) << 1) | (t & (j - 1)), i1 = i0 | j;       // the pair (i0, i0 + j) of this compare-exchange step
                const bool up = (i0 & k2) == 0;
                const unsigned long long a = key[i0], c = key[i1];
                if ((a > c) == up) { key[i0] = c; key[i1] = a; }
            }
            __syncthreads();
        }
    const int nv = cnt;
    if (tid == 0) nvalid[b] = nv;
    for (int s = tid; s < m.Ff; s += 512) {
        const size_t slot = (size_t)b * m.Ff + s;
        if (s < nv) {
            const int f = (int)(unsigned)(key[s] & 0xFFFFFFFFull);
            float p[9];
            face_corners(m, tb, f, p);
            store_face(frec, fbox, slot, p, __uint_as_float((unsigned)(key[s] >> 32)), f, face_pixel_box(p, H, W));
        } else {
            fbox[slot] = make_short4(1, 0, 1, 0);                                   // never binned
        }
    }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void raster_tile(MeshDev m, int B, int H, int W, const float* __restrict__ frec,
                                                   const short4* __restrict__ fbox, const int* __restrict__ nvalid, const float* __restrict__ normals,
                                                   float* __restrict__ img, long long* __restrict__ p2f_out,
                                                   float* __restrict__ bary_out, float* __restrict__ zbuf_out, int qmax, int ablate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    // layout: [FACE_CHUNK*FACE_REC floats][FACE_CHUNK ints][8 ints: per-wave list counts, 4 floats: per-wave block depth][4 x qmax uint16 lists]
    float* sface = (float*)dyn_smem;
    int* sfid = (int*)(sface + FACE_CHUNK * FACE_REC);
    int* scnt = sfid + FACE_CHUNK;
    float* szmx = (float*)(scnt + 4);
    unsigned short* slist = (unsigned short*)(scnt + 8);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (W + TILE_W - 1) / TILE_W;
    const int b = blockIdx.y;
    const int tx0 = (blockIdx.x % tiles_x) * TILE_W, ty0 = (blockIdx.x / tiles_x) * TILE_H;
    const int tx1 = min(tx0 + TILE_W, W) - 1, ty1 = min(ty0 + TILE_H, H) - 1;
    // ---- binning, order-preserving: wave w scans the sorted faces [w q, (w + 1) q) and compacts the ones whose box touches the tile into ITS list with a ballot
    // prefix (no atomics: the concatenation of the four lists is again in ascending zlow order)
    const short4* boxes = fbox + (size_t)b * m.Ff;
    const int nv = (ablate & 2) ? 0 : nvalid[b];
    {
        const int q = ((nv + 3) / 4 + 63) & ~63, lo = wave * q, hi = min(lo + q, nv);
        unsigned short* mylist = slist + wave * qmax;
        int c = 0;
        // eight 64-face rounds of boxes are requested before the first is tested: one global round trip per eight rounds instead of one per round (the ballot of a
        // round needs its boxes, so a rolled loop paid ~14 dependent L2 / HBM latencies per wave — most of a tile's time once the depth culling had removed the face loop)
        for (int base = lo; base < hi; base += 64 * 8) {
            short4 bx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = base + 64 * u + lane;
                bx[u] = boxes[min(s, nv - 1)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = base + 64 * u + lane;
                const bool hit = s < hi && bx[u].x <= tx1 && bx[u].y >= tx0 && bx[u].z <= ty1 && bx[u].w >= ty0 && bx[u].x <= bx[u].y;
                const unsigned long long mk = __ballot(hit);
                if (hit) mylist[c + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short)s;
                c += __popcll(mk);
            }
        }
        if (lane == 0) scnt[wave] = c;
    }
    __syncthreads();
    const int n0 = scnt[0], n1 = n0 + scnt[1], n2 = n1 + scnt[2];
    const int n = (ablate & 1) ? 0 : n2 + scnt[3];

    // a wave owns an 8 x 8 pixel block of the 32 x 8 tile (not a 32 x 2 strip): with face boxes of ~20 px the block is touched by 1.5x fewer faces
    const int xi = tx0 + (tid >> 6) * 8 + (tid & 7), yi = ty0 + ((tid & 63) >> 3);
    const bool live = (xi < W) && (yi < H);
    const float xf = pix_to_ndc(W - 1 - xi, W), yf = pix_to_ndc(H - 1 - yi, H);
    float best_z = 0.f, bw0 = -1.f, bw1 = -1.f, bw2 = -1.f;
    int best_f = -1;
    const float* fr = frec + (size_t)b * m.Ff * FREC_G;
    // NDC box of this wave's 8 x 8 block (NDC decreases with the pixel index): the same pix_to_ndc values its lanes use
    const int wx0 = tx0 + (tid >> 6) * 8;
    const float sx_hi = pix_to_ndc(W - 1 - wx0, W), sx_lo = pix_to_ndc(W - 1 - (wx0 + 7), W);
    const float sy_hi = pix_to_ndc(H - 1 - ty0, H), sy_lo = pix_to_ndc(H - 1 - (ty0 + TILE_H - 1), H);
    const float INF = __uint_as_float(0x7f800000u);
    // the records of a chunk are requested one chunk ahead (by the 64 threads that stage them) and land under the previous chunk's face loop: the LDS list lookup ->
    // global load -> LDS store chain was ~2 us per chunk in front of a face loop that the depth culling had made short
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
    auto fetch_record = [&](int i) {
        const int s = i < n0 ? slist[i] : i < n1 ? slist[qmax + i - n0] : i < n2 ? slist[2 * qmax + i - n1] : slist[3 * qmax + i - n2];
        g0 = *(const float4*)(fr + (size_t)s * FREC_G); g1 = *(const float4*)(fr + (size_t)s * FREC_G + 4); g2 = *(const float4*)(fr + (size_t)s * FREC_G + 8);
    };
    if (tid < min(FACE_CHUNK, n)) fetch_record(tid);
    for (int base = 0; base < n; base += FACE_CHUNK) {
        const int cnt = min(FACE_CHUNK, n - base);
        // depth the block's pixels hold now: +inf while any of them is uncovered (pixels outside the image do not count).  It only falls as faces are
        // visited, so a value read here stays a valid (conservative) bound for the whole chunk.
        const float zblock = wave_max(!live ? -INF : (best_f >= 0 ? best_z : INF));
        __syncthreads();                                          // the previous chunk's records and depth slots have been read by every wave
        if (lane == 0) szmx[wave] = zblock;
        // stage one record per face: everything that does not depend on the pixel is computed ONCE here, with the same operations (and
        // -ffp-contract=off) the per-pixel code used, so every value is bit-identical: vertices, area + eps, the face's NDC box and the
        // (b - a) factors of the three edge functions
        if (tid < cnt) {
            const float x0 = g0.x, y0 = g0.y, z0 = g0.z, x1 = g0.w, y1 = g1.x, z1 = g1.y, x2 = g1.z, y2 = g1.w, z2 = g2.x;
            float* r = sface + tid * FACE_REC;
            r[0] = x0; r[1] = y0; r[2] = z0; r[3] = x1; r[4] = y1; r[5] = z1; r[6] = x2; r[7] = y2; r[8] = z2;
            r[9] = edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS;
            r[10] = fminf(x0, fminf(x1, x2)); r[11] = fmaxf(x0, fmaxf(x1, x2));
            r[12] = fminf(y0, fminf(y1, y2)); r[13] = fmaxf(y0, fmaxf(y1, y2));
            r[14] = y2 - y1; r[15] = x2 - x1;                    // w0: edge_fn(p, v1, v2)
            r[16] = y0 - y2; r[17] = x0 - x2;                    // w1: edge_fn(p, v2, v0)
            r[18] = y1 - y0; r[19] = x1 - x0;                    // w2: edge_fn(p, v0, v1)
            r[20] = g2.y;                                        // zlow
            sfid[tid] = __float_as_int(g2.z);
        }
        __syncthreads();
        // every face from here on has zlow >= this chunk's first (ascending order): once all four blocks are covered in front of it the tile is done
        if (fmaxf(fmaxf(szmx[0], szmx[1]), fmaxf(szmx[2], szmx[3])) < sface[20]) break;
        if (base + FACE_CHUNK + tid < n && tid < FACE_CHUNK) fetch_record(base + FACE_CHUNK + tid);       // the next chunk's records
        // wave-level cull: a wave owns an 8 x 8 pixel block; lane l tests face l of the chunk against the block's NDC box (64 faces in parallel,
        // one LDS read each) and only the faces that can touch the strip are walked.  Conservative w.r.t. the per-pixel box test below (a face
        // whose box misses the strip fails that test at every pixel of it), so the result is unchanged; it removes the serial
        // read-compare-branch per face and pixel that made this loop 90 % of the kernel (tools/raster_time.py).
        bool rel = false;
        float zc = 0.f;                                           // this lane's face: lower bound of its depth INSIDE THIS BLOCK (>= the face's zlow)
        // a block that is already covered in front of this chunk's first (= nearest) face is finished for good — its wave only keeps the barriers company while
        // the tile's other blocks go on (the tile-level exit above needs all four)
        if (lane < cnt && !(zblock < sface[20])) {
            const float* r = sface + lane * FACE_REC;
            rel = !(sx_lo > r[11] || sx_hi < r[10] || sy_lo > r[13] || sy_hi < r[12]) && !(r[20] > zblock);      // depth: see face_zlow
            zc = r[20];
            // Edge test of the whole block (the synthetic FLAME basis stretches triangles to ~21 x 22-pixel boxes: a box overlaps many blocks its
            // triangle never enters).  A pixel is covered only if s e_k > 0 for all three edge functions, s = sign(area + eps); e_k is linear in the
            // pixel centre, so its maximum over the block sits at a corner and separates into an x and a y term.  The face is dropped when that
            // maximum is below MINUS a margin 50x larger than the fp32 rounding of the per-pixel evaluation (|e| <= |A| |dx| + |B| |dy|, error <~ 2e-7
            // of that): whatever the exact per-pixel arithmetic below would have computed for this block is <= 0, i.e. the per-pixel test would have
            // skipped the face at every pixel.  Conservative => pix_to_face / bary / zbuf stay bit-identical (tests/test_render_gpu.py).
            const float area = r[9];
            if (rel && area != 0.0f) {
                const float sg = area < 0.0f ? -1.0f : 1.0f;
                const float vx[3] = {r[3], r[6], r[0]}, vy[3] = {r[4], r[7], r[1]};          // e0 is taken about v1, e1 about v2, e2 about v0
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float A = sg * r[14 + 2 * k], Bc = -sg * r[15 + 2 * k];
                    const float dx0 = sx_lo - vx[k], dx1 = sx_hi - vx[k], dy0 = sy_lo - vy[k], dy1 = sy_hi - vy[k];
                    const float emax = fmaxf(A * dx0, A * dx1) + fmaxf(Bc * dy0, Bc * dy1);
                    const float mag = fabsf(A) * fmaxf(fabsf(dx0), fabsf(dx1)) + fabsf(Bc) * fmaxf(fabsf(dy0), fabsf(dy1));
                    if (emax < -1e-5f * mag) rel = false;
                }
                // Depth of the face's PLANE over this block.  The interpolated depth is the affine function F(p) = sum_k z_k e_k(p) / A of the pixel centre; over the
                // block's centres (within +-3.5 pixels of its middle m) F(p) >= F(m) - |dF/dx| hx - |dF/dy| hy.  The per-pixel code computes pz >= F(p) - zmax (24u ext^2 /
                // |A| + 6u) for a covered pixel (face_zlow's derivation), and F(m), dF/dx, dF/dy computed here in fp32 are within zmax 39u D^2 / |A| of their exact values
                // (D = face extent + block extent: the middle may lie outside the face's box).  err below is 3x their sum; the bound is dropped for ill-conditioned
                // faces (rho2 not small).  A face whose nearest VERTEX is in front of the surface but whose plane passes behind it in this block is culled by this
                // bound and not by zlow (tests/test_raster_zbound_cpu.py checks both against the per-pixel arithmetic).
                if (rel) {
                    const float z0 = r[2], z1 = r[5], z2 = r[8];
                    const float zmx = fmaxf(z0, fmaxf(z1, z2)), zmn = fminf(z0, fminf(z1, z2));
                    const float D = ((r[11] - r[10]) + (r[13] - r[12])) + ((sx_hi - sx_lo) + (sy_hi - sy_lo));
                    const float rho2 = (1e-8f + 1.2e-5f * D * D) / fabsf(area);                  // 200 u = 1.2e-5
                    if (rho2 < 0.01f && zmn > 1e-3f) {
                        const float inva = 1.0f / area;
                        const float cxm = 0.5f * (sx_lo + sx_hi), cym = 0.5f * (sy_lo + sy_hi);
                        const float hx = 0.5f * (sx_hi - sx_lo) * 1.00001f, hy = 0.5f * (sy_hi - sy_lo) * 1.00001f;
                        const float e0m = (cxm - r[3]) * r[14] - (cym - r[4]) * r[15];
                        const float e1m = (cxm - r[6]) * r[16] - (cym - r[7]) * r[17];
                        const float e2m = (cxm - r[0]) * r[18] - (cym - r[1]) * r[19];
                        const float Fm = ((e0m * z0 + e1m * z1) + e2m * z2) * inva;
                        const float gx = ((z0 * r[14] + z1 * r[16]) + z2 * r[18]) * inva, gy = ((z0 * r[15] + z1 * r[17]) + z2 * r[19]) * inva;
                        const float var = fabsf(gx) * hx + fabsf(gy) * hy;
                        const float zb = ((Fm - var) - (zmx * rho2 + (fabsf(Fm) + var) * 1e-5f)) * (1.0f - 2e-6f);
                        zc = fmaxf(zc, zb);                       // (NaN -> zc)
                        if (zc > zblock) rel = false;
                    }
                }
            }
        }
        unsigned long long relmask = __ballot(rel);
        while (relmask) {
            const int j = __builtin_ctzll(relmask);
            relmask &= relmask - 1;
            // the whole record in six 16-byte broadcast reads, then straight-line arithmetic: the box test no longer costs four dependent LDS round trips
            const float4* v4 = (const float4*)(sface + j * FACE_REC);
            const float4 c0 = v4[0], c1 = v4[1], c2 = v4[2], c3 = v4[3], c4 = v4[4];
            const float zl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zc), j));    // lane j's bound for ITS face in this block
            const float x0 = c0.x, y0 = c0.y, x1 = c0.w, y1 = c1.x, x2 = c1.z, y2 = c1.w;
            const float area = c2.y;
            // a pixel of the face's box that already holds a depth strictly below the face's lower depth bound cannot be won by this face: it is not a candidate.
            // A block at the silhouette is never covered as a whole (no block-level cull), but the ~24 layers behind its covered part have no candidate pixel
            // left and are dropped here, before the edge functions and the divisions.
            const bool inbox = !(xf > c2.w || xf < c2.z || yf > c3.y || yf < c3.x) && !(best_f >= 0 && best_z < zl);
            if (__ballot(inbox) == 0ull) continue;
            const float e0 = (xf - x1) * c3.z - (yf - y1) * c3.w;
            const float e1 = (xf - x2) * c4.x - (yf - y2) * c4.y;
            const float e2 = (xf - x0) * c4.z - (yf - y0) * c4.w;
            // w_i = e_i / area > 0 needs e_i != 0 and sign(e_i) == sign(area): decided without the three divisions for the pixels of the box
            // that lie outside the triangle (area == 0 -> inf / NaN quotients: left to the exact path)
            const bool neg = area < 0.0f;
            const bool out = area != 0.0f && (e0 == 0.0f || (e0 < 0.0f) != neg || e1 == 0.0f || (e1 < 0.0f) != neg || e2 == 0.0f || (e2 < 0.0f) != neg);
            if (!inbox || out) continue;
            const float w0 = e0 / area, w1 = e1 / area, w2 = e2 / area;
            const float pz = w0 * c0.z + w1 * c1.y + w2 * c2.x;
            if (pz < 0) continue;
            if (!((w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f))) continue;
            const int f = sfid[j];
            if (best_f < 0 || pz < best_z || (pz == best_z && f < best_f)) {
                best_z = pz; best_f = f; bw0 = w0; bw1 = w1; bw2 = w2;
            }
        }
    }
    if (!live) return;
    // ---- resolve + shade (renderer.py:150-166,194-206,239-250) -------------------------------------------------------
    float shaded = 0.f;
    if (best_f >= 0) {
        const float* nb = normals + (size_t)b * m.Vf * 3;
        const int i0 = m.faces[best_f * 3 + 0], i1 = m.faces[best_f * 3 + 1], i2 = m.faces[best_f * 3 + 2];
        float nimg[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) nimg[c] = (bw0 * nb[i0 * 3 + c] + bw1 * nb[i1 * 3 + c]) + bw2 * nb[i2 * 3 + c];
        const float grey = 180.0f / 255.0f;
        const float albedo = (bw0 * grey + bw1 * grey) + bw2 * grey;
        const float q = 0.57735026918962576f;   // 1/sqrt(3): F.normalize of (+-1,+-1,1)
        const float lx[5] = {-q, q, -q, q, 0.f}, ly[5] = {q, q, -q, -q, 0.f}, lz[5] = {q, q, q, q, 1.f};
        float sh = 0.f;
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            float t = (nimg[0] * lx[l] + nimg[1] * ly[l]) + nimg[2] * lz[l];
            t = fminf(fmaxf(t, 0.f), 1.f) * 1.7f;
            sh = (l == 0) ? t : sh + t;
        }
        sh = sh / 5.0f;
        shaded = albedo * sh;
    }
    const size_t pix = (size_t)yi * W + xi, plane = (size_t)H * W;
    float* ib = img + (size_t)b * 3 * plane;
    ib[pix] = shaded; ib[plane + pix] = shaded; ib[2 * plane + pix] = shaded;
    const size_t o = (size_t)b * plane + pix;
    if (p2f_out) p2f_out[o] = best_f >= 0 ? (long long)b * m.Ff + best_f : -1ll;
    if (zbuf_out) zbuf_out[o] = best_f >= 0 ? best_z : -1.f;
    if (bary_out) { bary_out[o * 3 + 0] = bw0; bary_out[o * 3 + 1] = bw1; bary_out[o * 3 + 2] = bw2; }
}

static MeshDev mesh_dev(const SmirkRenderMesh* m) {
    MeshDev d;
    d.V = m->V; d.Vf = m->Vf; d.Ff = m->Ff; d.nnz = m->nnz;
    d.keep = m->keep; d.faces = m->faces; d.nrm_ptr = m->nrm_ptr; d.nrm_face = m->nrm_face; d.nrm_corner = m->nrm_corner;
    return d;
}

static size_t ws_normals(const SmirkRenderMesh* m, int B) { return smirk_align_up((size_t)B * m->Vf * 12, 256); }
static size_t ws_frec(const SmirkRenderMesh* m, int B) { return smirk_align_up((size_t)B * m->Ff * FREC_G * 4, 256); }
static size_t ws_fbox(const SmirkRenderMesh* m, int B) { return smirk_align_up((size_t)B * m->Ff * 8, 256); }
static size_t ws_nvalid(int B) { return smirk_align_up((size_t)B * 4, 256); }

extern "C" size_t smirk_render_workspace_bytes(const SmirkRenderMesh* mesh, int B, int H, int W) {
    if (!mesh || B <= 0) return 0;
    (void)H; (void)W;
    return ws_normals(mesh, B) + ws_frec(mesh, B) + ws_fbox(mesh, B) + ws_nvalid(B);
}

extern "C" int smirk_render_forward(const SmirkRenderMesh* mesh, int B, int H, int W, const float* verts,
                                    const float* cam, float* transformed, float* img, int64_t* pix_to_face, float* bary,
                                    float* zbuf, float* normals, void* ws, size_t ws_bytes, void* stream) {
    if (!mesh || !verts || !cam || !transformed || !img || !ws || B <= 0) return SMIRK_ERR_BAD_ARG;
    if (H != W || H <= 0 || H > 1024 || mesh->Ff > 65535 || mesh->Ff <= 0) return SMIRK_ERR_UNSUPPORTED;
    if (ws_bytes < smirk_render_workspace_bytes(mesh, B, H, W)) return SMIRK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const MeshDev d = mesh_dev(mesh);
    char* p = (char*)ws;
    float* nrm = (float*)p; p += ws_normals(mesh, B);
    float* frec = (float*)p; p += ws_frec(mesh, B);
    short4* fbox = (short4*)p; p += ws_fbox(mesh, B);
    int* nvalid = (int*)p;
    if (normals) nrm = normals;
    const size_t nv = (size_t)B * mesh->V, nk = (size_t)B * mesh->Vf, nf = (size_t)B * mesh->Ff;
    SMIRK_LAUNCH(render_project, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, verts, cam, B, mesh->V,
                       transformed);
    SMIRK_LAUNCH(render_normals, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, d, B, verts, nrm);
    if (mesh->Ff <= SORT_N)
        SMIRK_LAUNCH(raster_face_setup_sorted, dim3((unsigned)B), dim3(512), 0, st, d, B, H, W, transformed, frec, fbox, nvalid);
    else
        SMIRK_LAUNCH(raster_face_setup, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, d, B, H, W, transformed, frec, fbox, nvalid);
    const int tiles = ((W + TILE_W - 1) / TILE_W) * ((H + TILE_H - 1) / TILE_H);
    const int qmax = (((mesh->Ff + 3) / 4 + 63) & ~63) + 8;          // capacity of a wave's binned list (a quarter of the faces, whole 64-face rounds)
    const size_t smem = FACE_CHUNK * FACE_REC * 4 + FACE_CHUNK * 4 + 32 + smirk_align_up((size_t)4 * qmax * 2, 16);
#ifdef SMIRK_DEBUG_HOOKS                                            /* -DSMIRK_DEBUG_HOOKS variant builds only (tools/build_variant.sh, tools/raster_time.py) */
    static const char* abl = getenv("SMIRK_RASTER_ABLATE");      // 1: no per-pixel face loop, 2: no binning scan (timing experiments; wrong images)
    const int ablate = abl ? atoi(abl) : 0;
#else
    const int ablate = 0;
#endif
    if (smem > 64 * 1024) {
        // the four per-wave bin lists grow with the mesh (2 bytes per face): beyond ~27 k faces the dynamic LDS passes the 64 KB a launch gets by default.
        // gfx950 has 160 KB per workgroup: raise this kernel's limit once per process (the largest size asked so far), refuse what cannot fit at all.
        if (smem > 150 * 1024) return SMIRK_ERR_UNSUPPORTED;
        static std::atomic<size_t> raised{0};
        if (raised.load(std::memory_order_relaxed) < smem) {
            if (hipFuncSetAttribute((const void*)raster_tile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
                (void)hipGetLastError();
                return SMIRK_ERR_UNSUPPORTED;
            }
            raised.store(smem, std::memory_order_relaxed);
        }
    }
    SMIRK_LAUNCH(raster_tile, dim3(tiles, B), dim3(256), smem, st, d, B, H, W, frec, fbox, nvalid, nrm, img,
                       (long long*)pix_to_face, bary, zbuf, qmax, ablate);
    return smirk_launch_status();
}

// ====================================================================================================================================
// Backward pass: dL/dvertices, dL/dcam from dL/drendered_img (+ optional dL/dtransformed_vertices) — what autograd
// through Renderer.forward + pytorch3d's RasterizeMeshes backward produce in the reference's training step (smirk_trainer.py:46-48,362).
// Visibility carries no gradient; the barycentric weights do (geometry_utils.cuh BarycentricCoordsBackward = the analytic derivative of
// w_k = e_k / (area + 1e-8)), as do the interpolated vertex normals through shading.  All sums are gathers in a fixed order: the
// gradients are bit-reproducible run to run (pytorch3d's atomicAdd backward is not).
//   render_bwd_faces    one lane per (image, face): walk the face's pixel box, for pixels it owns accumulate the gradients of its 3
//                       screen-space corners (bary path) and of its 3 corner normals (shading path)            -> fgrad[B][Ff][15]
//   render_bwd_normals  one lane per kept vertex: gather corner-normal gradients (CSR), F.normalize backward     -> dS[B][Vf][3]
//   render_bwd_vertices one workgroup per image: gather screen gradients + cross-product backward per vertex, un-project through the
//                       camera, reduce dL/dcam                                                                     -> d_verts, d_cam
// ====================================================================================================================================
#define FG 15
#define BWD_LANES 16
__global__ __launch_bounds__(256) void render_bwd_faces(MeshDev m, int B, int H, int W, const float* __restrict__ verts,
                                                        const float* __restrict__ cam, const float* __restrict__ normals,
                                                        const long long* __restrict__ p2f, const float* __restrict__ g_img,
                                                        float* __restrict__ fgrad) {
    const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = gt / BWD_LANES;                              // BWD_LANES lanes share one (image, face): balances big boxes
    const int sub = (int)(gt % BWD_LANES);
    if (i >= (size_t)B * m.Ff) return;
    const int b = (int)(i / m.Ff), f = (int)(i % m.Ff);
    const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    const float* vb = verts + (size_t)b * m.V * 3;
    const float* nb = normals + (size_t)b * m.Vf * 3;
    float p[9], N[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int lv = m.faces[f * 3 + k], g = m.keep[lv];
        // same operation sequence as render_project + raster_face_setup so that the box and the weights match the forward pass
        p[k * 3 + 0] = -(s * (vb[g * 3 + 0] + tx));
        p[k * 3 + 1] = -(-(s * (vb[g * 3 + 1] + ty)));
        p[k * 3 + 2] = -(s * vb[g * 3 + 2]) + 10.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) N[k * 3 + c] = nb[lv * 3 + c];
    }
    float acc[FG];
#pragma unroll
    for (int k = 0; k < FG; ++k) acc[k] = 0.f;
    const short4 box = face_pixel_box(p, H, W);
    const float x0 = p[0], y0 = p[1], x1 = p[3], y1 = p[4], x2 = p[6], y2 = p[7];
    const float area = edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS;
    const float grey = 180.0f / 255.0f, q = 0.57735026918962576f;
    const float lx[5] = {-q, q, -q, q, 0.f}, ly[5] = {q, q, -q, -q, 0.f}, lz[5] = {q, q, q, q, 1.f};
    const long long me = (long long)b * m.Ff + f;
    const size_t plane = (size_t)H * W;
    float g_area = 0.f;
    const int bw = box.y - box.x + 1, bh = box.w - box.z + 1;
    const int npx = (bw > 0 && bh > 0) ? bw * bh : 0;
    {
        for (int idx = sub; idx < npx; idx += BWD_LANES) {
            const int yo = idx / bw, yi = box.z + yo, xi = box.x + (idx - yo * bw);
            const size_t pix = (size_t)yi * W + xi;
            if (p2f[(size_t)b * plane + pix] != me) continue;
            const float yf = pix_to_ndc(H - 1 - yi, H);
            const float xf = pix_to_ndc(W - 1 - xi, W);
            const float e0 = edge_fn(xf, yf, x1, y1, x2, y2), e1 = edge_fn(xf, yf, x2, y2, x0, y0), e2 = edge_fn(xf, yf, x0, y0, x1, y1);
            const float w0 = e0 / area, w1 = e1 / area, w2 = e2 / area;
            const float* gi = g_img + (size_t)b * 3 * plane + pix;
            const float g = (gi[0] + gi[plane]) + gi[2 * plane];
            float nimg[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) nimg[c] = (w0 * N[c] + w1 * N[3 + c]) + w2 * N[6 + c];
            const float albedo = (w0 * grey + w1 * grey) + w2 * grey;
            float sh = 0.f, dl[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                const float t = (nimg[0] * lx[l] + nimg[1] * ly[l]) + nimg[2] * lz[l];
                sh += fminf(fmaxf(t, 0.f), 1.f) * 1.7f;
                if (t >= 0.f && t <= 1.f) { dl[0] += lx[l]; dl[1] += ly[l]; dl[2] += lz[l]; }     // clamp passes gradient on [0, 1]
            }
            sh = sh / 5.0f;
            const float d_alb = g * sh, d_sh = g * albedo * (1.7f / 5.0f);
            const float dn[3] = {d_sh * dl[0], d_sh * dl[1], d_sh * dl[2]};
            const float dw0 = d_alb * grey + ((dn[0] * N[0] + dn[1] * N[1]) + dn[2] * N[2]);
            const float dw1 = d_alb * grey + ((dn[0] * N[3] + dn[1] * N[4]) + dn[2] * N[5]);
            const float dw2 = d_alb * grey + ((dn[0] * N[6] + dn[1] * N[7]) + dn[2] * N[8]);
#pragma unroll
            for (int c = 0; c < 3; ++c) { acc[6 + c] += w0 * dn[c]; acc[9 + c] += w1 * dn[c]; acc[12 + c] += w2 * dn[c]; }
            g_area -= ((dw0 * e0 + dw1 * e1) + dw2 * e2) / (area * area);
            const float ge0 = dw0 / area, ge1 = dw1 / area, ge2 = dw2 / area;
            // e0 = E(p; v1, v2), e1 = E(p; v2, v0), e2 = E(p; v0, v1);  dE(p;a,b)/da = (p.y - b.y, b.x - p.x), dE/db = (a.y - p.y, p.x - a.x)
            acc[0] += ge1 * (y2 - yf) + ge2 * (yf - y1);   acc[1] += ge1 * (xf - x2) + ge2 * (x1 - xf);      // v0
            acc[2] += ge0 * (yf - y2) + ge2 * (y0 - yf);   acc[3] += ge0 * (x2 - xf) + ge2 * (xf - x0);      // v1
            acc[4] += ge0 * (y1 - yf) + ge1 * (yf - y0);   acc[5] += ge0 * (xf - x1) + ge1 * (x0 - xf);      // v2
        }
    }
#pragma unroll
    for (int o = BWD_LANES / 2; o > 0; o >>= 1) {                 // fixed-order tree over the lanes of this face: deterministic
#pragma unroll
        for (int k = 0; k < FG; ++k) acc[k] += __shfl_xor(acc[k], o, 64);
        g_area += __shfl_xor(g_area, o, 64);
    }
    if (sub != 0) return;
    // area = E(v2; v0, v1) + eps
    acc[0] += g_area * (y2 - y1); acc[1] += g_area * (x1 - x2);
    acc[2] += g_area * (y0 - y2); acc[3] += g_area * (x2 - x0);
    acc[4] += g_area * (y1 - y0); acc[5] += g_area * (x0 - x1);
#pragma unroll
    for (int k = 0; k < FG; ++k) fgrad[i * FG + k] = acc[k];
}

__global__ __launch_bounds__(256) void render_bwd_normals(MeshDev m, int B, const float* __restrict__ verts,
                                                          const float* __restrict__ fgrad, float* __restrict__ dS) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * m.Vf) return;
    const int b = (int)(i / m.Vf), vi = (int)(i % m.Vf);
    const float* vb = verts + (size_t)b * m.V * 3;
    const float* fg = fgrad + (size_t)b * m.Ff * FG;
    float n[3] = {0.f, 0.f, 0.f}, dN[3] = {0.f, 0.f, 0.f};
    for (int e = m.nrm_ptr[vi]; e < m.nrm_ptr[vi + 1]; ++e) {
        const int f = m.nrm_face[e], c = m.nrm_corner[e];
        float p[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int g = m.keep[m.faces[f * 3 + k]];
#pragma unroll
            for (int d = 0; d < 3; ++d) p[k][d] = vb[g * 3 + d];
        }
        const int i0 = c, i1 = (c + 1) % 3, i2 = (c + 2) % 3;
        float a[3], bb[3], cr[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { a[d] = p[i1][d] - p[i0][d]; bb[d] = p[i2][d] - p[i0][d]; }
        cross3(a, bb, cr);
#pragma unroll
        for (int d = 0; d < 3; ++d) { n[d] += cr[d]; dN[d] += fg[(size_t)f * FG + 6 + c * 3 + d]; }
    }
    const float nrm = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    float* o = dS + i * 3;
    if (nrm >= 1e-6f) {                                      // N = S / |S|:  dS = (dN - N (N.dN)) / |S|
        const float N0 = n[0] / nrm, N1 = n[1] / nrm, N2 = n[2] / nrm;
        const float dot = (N0 * dN[0] + N1 * dN[1]) + N2 * dN[2];
        o[0] = (dN[0] - N0 * dot) / nrm; o[1] = (dN[1] - N1 * dot) / nrm; o[2] = (dN[2] - N2 * dot) / nrm;
    } else {                                                 // N = S / eps
        o[0] = dN[0] / 1e-6f; o[1] = dN[1] / 1e-6f; o[2] = dN[2] / 1e-6f;
    }
}

__device__ inline void block_reduce3(float* v, float* red /*[3][4]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) red[k * 4 + wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = (red[k * 4 + 0] + red[k * 4 + 1]) + (red[k * 4 + 2] + red[k * 4 + 3]);
    __syncthreads();
}

// d_verts / d_cam initialisation with the (optional) gradient of `transformed_vertices` itself
__global__ __launch_bounds__(256) void render_bwd_init(int B, int V, const float* __restrict__ verts, const float* __restrict__ cam,
                                                       const float* __restrict__ g_tv, float* __restrict__ d_verts,
                                                       float* __restrict__ d_cam) {
    __shared__ float red[12];
    const int b = blockIdx.x;
    const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    float dc[3] = {0.f, 0.f, 0.f};
    for (int v = threadIdx.x; v < V; v += 256) {
        const size_t o = ((size_t)b * V + v) * 3;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (g_tv) { g0 = g_tv[o]; g1 = g_tv[o + 1]; g2 = g_tv[o + 2]; }
        d_verts[o] = s * g0; d_verts[o + 1] = -(s * g1); d_verts[o + 2] = -(s * g2);
        dc[0] += ((verts[o] + tx) * g0 - (verts[o + 1] + ty) * g1) - verts[o + 2] * g2;
        dc[1] += s * g0; dc[2] -= s * g1;
    }
    block_reduce3(dc, red);
    if (threadIdx.x < 3) d_cam[b * 3 + threadIdx.x] = dc[threadIdx.x];
}

__global__ __launch_bounds__(256) void render_bwd_vertices(MeshDev m, int B, const float* __restrict__ verts,
                                                           const float* __restrict__ cam, const float* __restrict__ fgrad,
                                                           const float* __restrict__ dS, float* __restrict__ d_verts,
                                                           float* __restrict__ d_cam) {
    __shared__ float red[12];
    const int b = blockIdx.x;
    const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    const float* vb = verts + (size_t)b * m.V * 3;
    const float* fg = fgrad + (size_t)b * m.Ff * FG;
    const float* dSb = dS + (size_t)b * m.Vf * 3;
    float dc[3] = {0.f, 0.f, 0.f};
    for (int vi = threadIdx.x; vi < m.Vf; vi += 256) {
        float gsx = 0.f, gsy = 0.f, dv[3] = {0.f, 0.f, 0.f};
        for (int e = m.nrm_ptr[vi]; e < m.nrm_ptr[vi + 1]; ++e) {
            const int f = m.nrm_face[e], c = m.nrm_corner[e];
            gsx += fg[(size_t)f * FG + c * 2]; gsy += fg[(size_t)f * FG + c * 2 + 1];
            const int l1 = m.faces[f * 3 + (c + 1) % 3], l2 = m.faces[f * 3 + (c + 2) % 3];
            const int g0i = m.keep[vi], g1i = m.keep[l1], g2i = m.keep[l2];
            float p0[3], p1[3], p2[3], a[3], bb[3], t[3], u[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) { p0[d] = vb[g0i * 3 + d]; p1[d] = vb[g1i * 3 + d]; p2[d] = vb[g2i * 3 + d]; }
            const float* G0 = dSb + vi * 3; const float* G1 = dSb + l1 * 3; const float* G2 = dSb + l2 * 3;
            // this vertex's own corner: n = a x b, a = p1 - p0, b = p2 - p0:  d/da = b x G, d/db = G x a, d/dp0 = -(both)
#pragma unroll
            for (int d = 0; d < 3; ++d) { a[d] = p1[d] - p0[d]; bb[d] = p2[d] - p0[d]; }
            cross3(bb, G0, t); cross3(G0, a, u);
#pragma unroll
            for (int d = 0; d < 3; ++d) dv[d] -= t[d] + u[d];
            // corner at p2: n = (p0 - p2) x (p1 - p2): p0 enters the first factor:  d/dp0 = (p1 - p2) x G2
#pragma unroll
            for (int d = 0; d < 3; ++d) a[d] = p1[d] - p2[d];
            cross3(a, G2, t);
            // corner at p1: n = (p2 - p1) x (p0 - p1): p0 enters the second factor: d/dp0 = G1 x (p2 - p1)
#pragma unroll
            for (int d = 0; d < 3; ++d) bb[d] = p2[d] - p1[d];
            cross3(G1, bb, u);
#pragma unroll
            for (int d = 0; d < 3; ++d) dv[d] += t[d] + u[d];
        }
        const int g = m.keep[vi];
        const size_t o = ((size_t)b * m.V + g) * 3;
        // screen = (-(s (x + tx)), s (y + ty))
        d_verts[o] += dv[0] - s * gsx;
        d_verts[o + 1] += dv[1] + s * gsy;
        d_verts[o + 2] += dv[2];
        dc[0] += (vb[g * 3 + 1] + ty) * gsy - (vb[g * 3] + tx) * gsx;
        dc[1] -= s * gsx; dc[2] += s * gsy;
    }
    block_reduce3(dc, red);
    if (threadIdx.x < 3) d_cam[b * 3 + threadIdx.x] += dc[threadIdx.x];
}

__global__ __launch_bounds__(256) void project_landmarks_bwd(const float* __restrict__ lmk, const float* __restrict__ cam,
                                                             const float* __restrict__ g_out, int B, int L,
                                                             float* __restrict__ d_lmk, float* __restrict__ d_cam) {
    __shared__ float red[12];
    const int b = blockIdx.x;
    const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    float dc[3] = {0.f, 0.f, 0.f};
    for (int l = threadIdx.x; l < L; l += 256) {
        const size_t i = (size_t)b * L + l;
        const float g0 = g_out[i * 2], g1 = g_out[i * 2 + 1];
        d_lmk[i * 3] = s * g0; d_lmk[i * 3 + 1] = -(s * g1); d_lmk[i * 3 + 2] = 0.f;
        dc[0] += (lmk[i * 3] + tx) * g0 - (lmk[i * 3 + 1] + ty) * g1;
        dc[1] += s * g0; dc[2] -= s * g1;
    }
    block_reduce3(dc, red);
    if (threadIdx.x < 3) d_cam[b * 3 + threadIdx.x] += dc[threadIdx.x];
}

static size_t ws_fgrad(const SmirkRenderMesh* m, int B) { return smirk_align_up((size_t)B * m->Ff * FG * 4, 256); }

extern "C" size_t smirk_render_backward_workspace_bytes(const SmirkRenderMesh* mesh, int B, int H, int W) {
    if (!mesh || B <= 0) return 0;
    (void)H; (void)W;
    return 2 * ws_normals(mesh, B) + ws_fgrad(mesh, B);
}

extern "C" int smirk_render_backward(const SmirkRenderMesh* mesh, int B, int H, int W, const float* verts, const float* cam,
                                     const int64_t* pix_to_face, const float* g_img, const float* g_transformed, float* d_verts,
                                     float* d_cam, void* ws, size_t ws_bytes, void* stream) {
    if (!mesh || !verts || !cam || !pix_to_face || !d_verts || !d_cam || !ws || B <= 0) return SMIRK_ERR_BAD_ARG;
    if (H != W || H <= 0 || H > 1024 || mesh->Ff > 65535 || mesh->Ff <= 0) return SMIRK_ERR_UNSUPPORTED;
    if (ws_bytes < smirk_render_backward_workspace_bytes(mesh, B, H, W)) return SMIRK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const MeshDev d = mesh_dev(mesh);
    char* p = (char*)ws;
    float* nrm = (float*)p; p += ws_normals(mesh, B);
    float* dS = (float*)p; p += ws_normals(mesh, B);
    float* fgrad = (float*)p;
    const size_t nk = (size_t)B * mesh->Vf, nf = (size_t)B * mesh->Ff;
    SMIRK_LAUNCH(render_bwd_init, dim3(B), dim3(256), 0, st, B, mesh->V, verts, cam, g_transformed, d_verts, d_cam);
    if (g_img) {
        SMIRK_LAUNCH(render_normals, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, d, B, verts, nrm);
        SMIRK_LAUNCH(render_bwd_faces, dim3((unsigned)((nf * BWD_LANES + 255) / 256)), dim3(256), 0, st, d, B, H, W, verts, cam,
                           (const float*)nrm, (const long long*)pix_to_face, g_img, fgrad);
        SMIRK_LAUNCH(render_bwd_normals, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, d, B, verts, (const float*)fgrad, dS);
        SMIRK_LAUNCH(render_bwd_vertices, dim3(B), dim3(256), 0, st, d, B, verts, cam, (const float*)fgrad, (const float*)dS,
                           d_verts, d_cam);
    }
    return smirk_launch_status();
}

extern "C" int smirk_project_landmarks_backward(const float* lmk, const float* cam, const float* g_out, int B, int L, float* d_lmk,
                                                float* d_cam_accum, void* stream) {
    if (!lmk || !cam || !g_out || !d_lmk || !d_cam_accum || B <= 0 || L <= 0) return SMIRK_ERR_BAD_ARG;
    SMIRK_LAUNCH(project_landmarks_bwd, dim3(B), dim3(256), 0, (hipStream_t)stream, lmk, cam, g_out, B, L, d_lmk, d_cam_accum);
    return smirk_launch_status();
}

// vertex normals on their own (used by utils/masking.py:146 on the FULL mesh of the transformed vertices): mesh->keep may be NULL => identity
__global__ __launch_bounds__(256) void normals_full_kernel(MeshDev m, int B, const float* __restrict__ verts, float* __restrict__ normals) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * m.Vf) return;
    const int b = (int)(i / m.Vf), vi = (int)(i % m.Vf);
    const float* vb = verts + (size_t)b * m.V * 3;
    float n[3] = {0.f, 0.f, 0.f};
    for (int e = m.nrm_ptr[vi]; e < m.nrm_ptr[vi + 1]; ++e) {
        const int f = m.nrm_face[e], c = m.nrm_corner[e];
        float p[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int g = m.faces[f * 3 + k];
#pragma unroll
            for (int d = 0; d < 3; ++d) p[k][d] = vb[g * 3 + d];
        }
        const int i0 = c, i1 = (c + 1) % 3, i2 = (c + 2) % 3;
        float a[3], bb[3], cr[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { a[d] = p[i1][d] - p[i0][d]; bb[d] = p[i2][d] - p[i0][d]; }
        cross3(a, bb, cr);
        n[0] += cr[0]; n[1] += cr[1]; n[2] += cr[2];
    }
    const float den = fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-6f);
    normals[i * 3 + 0] = n[0] / den; normals[i * 3 + 1] = n[1] / den; normals[i * 3 + 2] = n[2] / den;
}

extern "C" int smirk_vertex_normals(const SmirkRenderMesh* mesh, int B, const float* verts, float* normals, void* stream) {
    if (!mesh || !verts || !normals || B <= 0 || mesh->Vf != mesh->V) return SMIRK_ERR_BAD_ARG;   // full mesh only (faces index verts directly)
    const MeshDev d = mesh_dev(mesh);
    const size_t n = (size_t)B * mesh->Vf;
    SMIRK_LAUNCH(normals_full_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, B, verts, normals);
    return smirk_launch_status();
}

extern "C" int smirk_project_landmarks(const float* lmk, const float* cam, int B, int L, float* out, void* stream) {
    if (!lmk || !cam || !out || B <= 0 || L <= 0) return SMIRK_ERR_BAD_ARG;
    const size_t n = (size_t)B * L;
    SMIRK_LAUNCH(project_landmarks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lmk, cam,
                       B, L, out);
    return smirk_launch_status();
}
