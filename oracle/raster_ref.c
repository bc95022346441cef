/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of the rasteriser SMIRK calls at src/renderer/renderer.py:185-193:
 *   pytorch3d.renderer.mesh.rasterize_meshes(image_size=224, blur_radius=0.0, faces_per_pixel=1,
 *                                            bin_size=None, perspective_correct=False)
 * pytorch3d (un-pinned; "highest wheel in py39_cu117_pyt201", readme.md:37 => 0.7.4/0.7.5) is NOT on disk,
 * so this follows its published naive CPU algorithm from memory:
 *   csrc/rasterize_meshes/rasterize_meshes_cpu.cpp :: RasterizeMeshesNaiveCpu, ComputeFaceAreas,
 *                                                     ComputeFaceBoundingBoxes, CheckPointOutsideBoundingBox
 *   csrc/utils/geometry_utils.h                     :: EdgeFunctionForward, BarycentricCoordinatesForward (kEpsilon = 1e-8)
 *   csrc/rasterize_meshes/rasterization_utils.h     :: PixToNonSquareNdc
 * z rule (stated because releases differ): a face is skipped for EVERY pixel when its NEAREST vertex is not in front of the camera,
 *   z_invalid = zmin < kEpsilon        ("Faces with at least one vertex behind the camera won't render correctly and should be removed
 *                                        or clipped before calling the rasterizer" — CheckPointOutsideBoundingBox, 0.7.x, CPU and CUDA alike)
 * i.e. a face that STRADDLES z = 0 (zmin < eps <= zmax) is dropped whole, not clipped.  Rounds 1-3 restated this as `zmax < eps` (only faces
 * entirely behind the camera dropped; older releases); it cannot matter for SMIRK (every z is ~10 after renderer.py:144) but the restatement claims
 * literalness, so the 0.7.x rule is implemented and pinned by tests/test_cpu_suite.py::test_raster_kat_face_straddling_z0_is_dropped_whole.
 * PARITY UNPINNED: the reference has no golden vectors for this call; correctness is defended by the analytic
 * known-answer tests in tests/test_cpu_suite.py (test_raster_kat_*, test_raster_c_matches_numpy_on_random_soup) and tests/test_raster_differential.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (plain fp32 mul/sub/add/div, no FMA: the x86 wheel has none).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define K_EPS 1e-8f

static inline float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    /* EdgeFunctionForward(p, a, b) */
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

static inline float pix_to_ndc(int i, int S) {
    /* PixToNonSquareNdc for a square image: range 2, offset 1 */
    return -1.0f + (2.0f * (float)i + 1.0f) / (float)S;
}

/* face_verts: [B, Ff, 3, 3] fp32 (x, y, z per corner; already in pytorch3d NDC: +X left, +Y up)
 * pix_to_face: [B, H, W] int32, face index LOCAL to the mesh (0..Ff-1) or -1
 * zbuf:        [B, H, W] fp32 (-1 background)
 * bary:        [B, H, W, 3] fp32 (-1 background)                                        */
void smirk_oracle_rasterize_naive(const float* face_verts, int B, int Ff, int H, int W,
                                  int32_t* pix_to_face, float* zbuf, float* bary) {
    for (int n = 0; n < B; ++n) {
        const float* fv = face_verts + (size_t)n * Ff * 9;
#pragma omp parallel for schedule(dynamic, 4)
        for (int yi = 0; yi < H; ++yi) {
            const float yf = pix_to_ndc(H - 1 - yi, H);
            for (int xi = 0; xi < W; ++xi) {
                const float xf = pix_to_ndc(W - 1 - xi, W);
                float best_z = 0.f, bw0 = -1.f, bw1 = -1.f, bw2 = -1.f;
                int best_f = -1;
                for (int f = 0; f < Ff; ++f) {
                    const float* v = fv + (size_t)f * 9;
                    const float x0 = v[0], y0 = v[1], z0 = v[2];
                    const float x1 = v[3], y1 = v[4], z1 = v[5];
                    const float x2 = v[6], y2 = v[7], z2 = v[8];
                    /* ComputeFaceAreas: EdgeFunctionForward(v0, v1, v2) */
                    const float face_area = edge_fn(x0, y0, x1, y1, x2, y2);
                    if (fabsf(face_area) <= K_EPS) continue;
                    /* CheckPointOutsideBoundingBox, blur 0 (inclusive), z_invalid = zmin < kEpsilon */
                    const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
                    const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
                    const float zmin = fminf(z0, fminf(z1, z2));
                    if (xf > xmax || xf < xmin || yf > ymax || yf < ymin || zmin < K_EPS) continue;
                    /* BarycentricCoordinatesForward */
                    const float area = edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS;
                    const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / area;
                    const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / area;
                    const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / area;
                    const float pz = w0 * z0 + w1 * z1 + w2 * z2;
                    if (pz < 0) continue;
                    const int inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);
                    if (!inside) continue; /* blur_radius == 0 */
                    /* K = 1: keep the lexicographic minimum of (pz, f); f ascends so strict < keeps the lower f */
                    if (best_f < 0 || pz < best_z) {
                        best_z = pz; best_f = f; bw0 = w0; bw1 = w1; bw2 = w2;
                    }
                }
                const size_t o = ((size_t)n * H + yi) * W + xi;
                pix_to_face[o] = best_f;
                zbuf[o] = best_f >= 0 ? best_z : -1.f;
                bary[o * 3 + 0] = bw0; bary[o * 3 + 1] = bw1; bary[o * 3 + 2] = bw2;
            }
        }
    }
}
