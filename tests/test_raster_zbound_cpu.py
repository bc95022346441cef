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


def block_zc(p, A, zlow, sx_lo, sx_hi, sy_lo, sy_hi):
    """numpy float32 restatement of the per-(face, block) plane bound of raster_tile's cull stage, same operation order; scalars for one face and one block"""
    x0, y0, z0, x1, y1, z1, x2, y2, z2 = (f32(v) for v in p)
    zmx, zmn = max(z0, z1, z2), min(z0, z1, z2)
    ext = f32(f32(max(x0, x1, x2) - min(x0, x1, x2)) + f32(max(y0, y1, y2) - min(y0, y1, y2)))
    D = f32(ext + f32(f32(sx_hi - sx_lo) + f32(sy_hi - sy_lo)))
    with np.errstate(all="ignore"):
        rho2 = f32(f32(f32(1e-8) + f32(f32(f32(1.2e-5) * D) * D)) / abs(A))
    if not (rho2 < f32(0.01) and zmn > f32(1e-3)):
        return zlow
    r14, r15, r16, r17, r18, r19 = f32(y2 - y1), f32(x2 - x1), f32(y0 - y2), f32(x0 - x2), f32(y1 - y0), f32(x1 - x0)
    inva = f32(f32(1.0) / A)
    cxm, cym = f32(f32(0.5) * f32(sx_lo + sx_hi)), f32(f32(0.5) * f32(sy_lo + sy_hi))
    hx, hy = f32(f32(f32(0.5) * f32(sx_hi - sx_lo)) * f32(1.00001)), f32(f32(f32(0.5) * f32(sy_hi - sy_lo)) * f32(1.00001))
    e0m = f32(f32(f32(cxm - x1) * r14) - f32(f32(cym - y1) * r15))
    e1m = f32(f32(f32(cxm - x2) * r16) - f32(f32(cym - y2) * r17))
    e2m = f32(f32(f32(cxm - x0) * r18) - f32(f32(cym - y0) * r19))
    Fm = f32(f32(f32(f32(e0m * z0) + f32(e1m * z1)) + f32(e2m * z2)) * inva)
    gx = f32(f32(f32(f32(z0 * r14) + f32(z1 * r16)) + f32(z2 * r18)) * inva)
    gy = f32(f32(f32(f32(z0 * r15) + f32(z1 * r17)) + f32(z2 * r19)) * inva)
    var = f32(f32(abs(gx) * hx) + f32(abs(gy) * hy))
    zb = f32(f32(f32(Fm - var) - f32(f32(zmx * rho2) + f32(f32(abs(Fm) + var) * f32(1e-5)))) * f32(1.0 - 2e-6))
    return zlow if not (zb > zlow) else zb


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


def test_block_plane_bound_is_a_lower_bound_inside_the_block():
    """the per-(face, 8 x 8 block) bound of the cull stage (plane depth at the block's middle minus its variation over the block minus the error terms) against the
    per-pixel arithmetic, for every 8 x 8 block a triangle's box touches; it must also be USEFUL: for slanted triangles it beats the nearest-vertex bound"""
    rng = np.random.default_rng(11)
    H = W = 224
    p = _triangles(rng, 6000)
    zlow, A = face_zlow(p)
    ndc = (f32(-1.0) + (f32(2.0) * np.arange(W, dtype=f32)[::-1] + f32(1.0)).astype(f32) / f32(W)).astype(f32)      # pixel index i -> NDC
    checked = tighter = blocks = 0
    for k in range(p.shape[0]):
        if A[k] == 0:
            continue
        x0, y0, z0, x1, y1, z1, x2, y2, z2 = (p[k:k + 1, j] for j in range(9))
        xmin, xmax = min(x0[0], x1[0], x2[0]), max(x0[0], x1[0], x2[0])
        ymin, ymax = min(y0[0], y1[0], y2[0]), max(y0[0], y1[0], y2[0])
        xi = np.nonzero((ndc >= xmin) & (ndc <= xmax))[0]
        yi = np.nonzero((ndc >= ymin) & (ndc <= ymax))[0]
        if xi.size == 0 or yi.size == 0:
            continue
        bxs, bys = np.unique(xi // 8), np.unique(yi // 8)
        if bxs.size * bys.size > 12:                                  # huge triangles: a random subset of their blocks
            bxs, bys = rng.choice(bxs, min(bxs.size, 4), replace=False), rng.choice(bys, min(bys.size, 3), replace=False)
        for bx in bxs:
            for by in bys:
                px, py = np.meshgrid(ndc[bx * 8:bx * 8 + 8], ndc[by * 8:by * 8 + 8])
                px, py = px.ravel(), py.ravel()
                inb = (px >= xmin) & (px <= xmax) & (py >= ymin) & (py <= ymax)
                if not inb.any():
                    continue
                with np.errstate(all="ignore"):
                    w0 = (_E(px, py, x1, y1, x2, y2) / A[k]).astype(f32)
                    w1 = (_E(px, py, x2, y2, x0, y0) / A[k]).astype(f32)
                    w2 = (_E(px, py, x0, y0, x1, y1) / A[k]).astype(f32)
                    pz = (((w0 * z0).astype(f32) + (w1 * z1).astype(f32)).astype(f32) + (w2 * z2).astype(f32)).astype(f32)
                ok = inb & (w0 > 0) & (w1 > 0) & (w2 > 0) & ~(pz < 0)
                blocks += 1
                # block NDC box exactly as the kernel forms it: pix_to_ndc of the block's first / last pixel index (NDC decreases with the index)
                zc = block_zc(p[k], A[k], zlow[k], ndc[bx * 8 + 7], ndc[bx * 8], ndc[by * 8 + 7], ndc[by * 8])
                if ok.any():
                    checked += int(ok.sum())
                    assert pz[ok].min() >= zc, (k, bx, by, p[k], float(pz[ok].min()), float(zc), float(zlow[k]))
                tighter += int(zc > zlow[k] * f32(1.0005))
    assert checked > 100000 and blocks > 20000, (checked, blocks)
    assert tighter > blocks // 4, (tighter, blocks)
