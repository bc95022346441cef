"""The rasteriser's depth culling (csrc/render.hip, round 5) skips a face for a block of pixels when the face's `zlow` is strictly above the depth every pixel of the
block already holds.  That is bit-exact only if zlow really is a LOWER BOUND of the depth pz the per-pixel code computes for ANY pixel the face covers.  The bound is
derived in the kernel's header; this test restates `face_zlow` in numpy float32 and checks it against the fp32 per-pixel arithmetic of the oracle (same operation order
as oracle/raster_ref.c and rasterize_numpy: un-fused mul / sub / add / div) on 20 000 random triangles — ordinary, sliver, tiny, huge, back-facing, close to the z
cut-off — at every pixel centre of the face's box that the inside test accepts.  No GPU needed: the property is arithmetic."""
import numpy as np

f32 = np.float32


def _E(px, py, ax, ay, bx, by):
    return (((px - ax).astype(f32) * (by - ay).astype(f32)).astype(f32) - ((py - ay).astype(f32) * (bx - ax).astype(f32)).astype(f32)).astype(f32)


def face_zlow(p):
    """numpy float32 restatement of csrc/render.hip::face_zlow (p: [n, 9] = x0 y0 z0 x1 y1 z1 x2 y2 z2)"""
    x0, y0, z0, x1, y1, z1, x2, y2, z2 = (p[:, k] for k in range(9))
    xmin, xmax = np.minimum(x0, np.minimum(x1, x2)), np.maximum(x0, np.maximum(x1, x2))
    ymin, ymax = np.minimum(y0, np.minimum(y1, y2)), np.maximum(y0, np.maximum(y1, y2))
    zmin = np.minimum(z0, np.minimum(z1, z2))
    A = (_E(x2, y2, x0, y0, x1, y1) + f32(1e-8)).astype(f32)
    ext = ((xmax - xmin).astype(f32) + (ymax - ymin).astype(f32)).astype(f32)
    with np.errstate(all="ignore"):
        rho = ((f32(1e-8) + (f32(7.63e-6) * ext).astype(f32) * ext).astype(f32) / np.abs(A)).astype(f32)
        z = ((zmin * (f32(1.0) - rho).astype(f32)).astype(f32) * f32(1.0 - 2e-6)).astype(f32)
    return np.where((rho < f32(0.25)) & (zmin > f32(1e-3)), z, f32(0.0)).astype(f32), A


def _triangles(rng, n):
    """a mix of shapes in the NDC square, z around SMIRK's +10 offset and down to the validity cut-off"""
    kind = rng.integers(0, 6, n)
    c = rng.uniform(-1.1, 1.1, (n, 1, 2))
    size = np.choose(kind, [0.2, 0.02, 0.003, 1.5, 0.2, 0.2])[:, None, None]
    v = c + rng.normal(0, 1, (n, 3, 2)) * size
    sl = kind == 4                                                   # slivers: third vertex almost on the line through the first two
    t = rng.uniform(-0.2, 1.2, n)
    v[sl, 2] = v[sl, 0] + (v[sl, 1] - v[sl, 0]) * t[sl, None] + rng.normal(0, 1e-5, (sl.sum(), 2))
    z = np.where(kind[:, None] == 5, rng.uniform(2e-3, 0.5, (n, 3)), 10.0 + rng.normal(0, 0.15, (n, 3)))
    p = np.concatenate([v, z[:, :, None]], 2).reshape(n, 9).astype(f32)
    flip = rng.random(n) < 0.5                                       # back faces are kept by the reference (area < 0 flips all signs)
    p[flip] = p[flip][:, [3, 4, 5, 0, 1, 2, 6, 7, 8]]
    return p


def test_zlow_is_a_lower_bound_of_every_depth_the_per_pixel_code_computes():
    rng = np.random.default_rng(5)
    H = W = 224
    p = _triangles(rng, 20000)
    zlow, A = face_zlow(p)
    ndc = (f32(-1.0) + (f32(2.0) * np.arange(W, dtype=f32)[::-1] + f32(1.0)).astype(f32) / f32(W)).astype(f32)      # pixel index -> NDC (renderer: index i sees W-1-i)
    checked = culled_possible = 0
    worst = np.inf
    for k in range(p.shape[0]):
        x0, y0, z0, x1, y1, z1, x2, y2, z2 = (p[k:k + 1, j] for j in range(9))
        xs = ndc[(ndc >= min(x0[0], x1[0], x2[0])) & (ndc <= max(x0[0], x1[0], x2[0]))]
        ys = ndc[(ndc >= min(y0[0], y1[0], y2[0])) & (ndc <= max(y0[0], y1[0], y2[0]))]
        if xs.size == 0 or ys.size == 0 or A[k] == 0:
            continue
        if xs.size * ys.size > 4096:                                 # huge triangles: a random subset of their pixel centres
            xs, ys = rng.choice(xs, min(xs.size, 64), replace=False), rng.choice(ys, min(ys.size, 64), replace=False)
        px, py = np.meshgrid(xs, ys)
        px, py = px.ravel().astype(f32), py.ravel().astype(f32)
        with np.errstate(all="ignore"):
            w0 = (_E(px, py, x1, y1, x2, y2) / A[k]).astype(f32)
            w1 = (_E(px, py, x2, y2, x0, y0) / A[k]).astype(f32)
            w2 = (_E(px, py, x0, y0, x1, y1) / A[k]).astype(f32)
            pz = (((w0 * z0).astype(f32) + (w1 * z1).astype(f32)).astype(f32) + (w2 * z2).astype(f32)).astype(f32)
        ok = (w0 > 0) & (w1 > 0) & (w2 > 0) & ~(pz < 0)
        if not ok.any():
            continue
        checked += int(ok.sum())
        culled_possible += int(zlow[k] > 0)
        assert pz[ok].min() >= zlow[k], (k, p[k], float(pz[ok].min()), float(zlow[k]))
        if zlow[k] > 0:
            worst = min(worst, float(pz[ok].min() / zlow[k]))
    assert checked > 200000 and culled_possible > 5000, (checked, culled_possible)
    assert worst < 1.01, worst                                       # ... and the bound is tight enough to cull: within 1 % of the smallest depth seen


def test_zlow_gives_up_on_degenerate_faces():
    p = np.array([[0, 0, 10, 1e-4, 0, 10, 0, 1e-4, 10],             # area 1e-8: rho = 1 -> no bound
                  [0, 0, 10, 0.2, 0, 10, 0, 0.2, 10],               # ordinary
                  [0, 0, 5e-4, 0.2, 0, 10, 0, 0.2, 10],             # a vertex almost at the camera plane
                  [0, 0, np.nan, 0.2, 0, 10, 0, 0.2, 10]], f32)
    z, _ = face_zlow(p)
    assert z[0] == 0 and 9.99 < z[1] < 10 and z[2] == 0 and z[3] == 0
