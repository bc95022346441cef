"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the per-frame pre/post-processing of the reference's video loop (demo_video.py:16-36,107-214).
The arithmetic there is done by third-party libraries that are NOT on disk (cv2 = opencv-python, skimage = scikit-image, both only
named in the reference's requirements.txt without the sources): every function below restates the PUBLISHED algorithm and is
**PARITY UNPINNED** — it is anchored on the reference's call sites only.

  crop_transform      demo_video.py:16-36 crop_face -> skimage.transform.estimate_transform('similarity', src, dst) = Umeyama (1991),
                      skimage/transform/_geometric.py::_umeyama
  warp_u8             skimage.transform.warp(image, inverse_map, output_shape, preserve_range=True).astype(np.uint8), order=1,
                      mode='constant', cval=0: float64 bilinear, skimage/_shared/interpolation.pxd::bilinear_interpolation
                      (floor/ceil neighbours, constant 0 outside), clipping to the input range is a no-op for convex combinations
  resize_linear_u8    cv2.resize(img, (w, h)) default INTER_LINEAR on 8-bit data: fixed-point separable filter, coefficients scaled by
                      2^11 (modules/imgproc/src/resize.cpp: HResizeLinear / VResizeLinear<uchar,int,short>)
  hull_mask           datasets/base_dataset.py:9-15 create_mask: cv2.convexHull + cv2.fillConvexPoly(mask, hull, 0).  Restated as the
                      closed convex polygon scan-converted row by row with round-half-up ends (fillConvexPoly's XY_SHIFT rounding); its
                      additional 8-connected outline pass and 16-bit fixed-point edge stepping are NOT restated (<= 1 px at the boundary,
                      which the caller then dilates by a 21x21 max filter, masking.py:84-88)
  interp_bilinear     F.interpolate(x, (H, W), mode='bilinear') (align_corners=False) — torch is on disk: pinned in the tests against it
  to_u8 / from_u8     (x * 255.0).astype(np.uint8) and torch.Tensor(u8).float() / 255.0 in float32, channel swap = cv2.cvtColor BGR2RGB
"""
import numpy as np

f32 = np.float32


def umeyama(src, dst):
    """Least-squares similarity transform dst ~ s R src + t (3x3 homogeneous), skimage's _umeyama with estimate_scale=True."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    num, dim = src.shape
    sm, dm = src.mean(0), dst.mean(0)
    sd, dd = src - sm, dst - dm
    A = dd.T @ sd / num
    d = np.ones(dim)
    if np.linalg.det(A) < 0:
        d[dim - 1] = -1
    T = np.eye(dim + 1)
    U, S, V = np.linalg.svd(A)
    rank = np.linalg.matrix_rank(A)
    if rank == 0:
        return np.nan * T
    if rank == dim - 1:
        if np.linalg.det(U) * np.linalg.det(V) > 0:
            T[:dim, :dim] = U @ V
        else:
            s = d[dim - 1]
            d[dim - 1] = -1
            T[:dim, :dim] = U @ np.diag(d) @ V
            d[dim - 1] = s
    else:
        T[:dim, :dim] = U @ np.diag(d) @ V
    scale = 1.0 / sd.var(axis=0).sum() * (S @ d)
    T[:dim, dim] = dm - scale * (T[:dim, :dim] @ sm.T)
    T[:dim, :dim] *= scale
    return T


def crop_transform(landmarks, scale=1.0, image_size=224):
    """demo_video.py:16-36 — returns tform.params (3x3): original-frame (x, y) -> crop (x, y)."""
    lm = np.asarray(landmarks)
    left, right, top, bottom = np.min(lm[:, 0]), np.max(lm[:, 0]), np.min(lm[:, 1]), np.max(lm[:, 1])
    old_size = (right - left + bottom - top) / 2
    center = np.array([right - (right - left) / 2.0, bottom - (bottom - top) / 2.0])
    size = int(old_size * scale)
    src = np.array([[center[0] - size / 2, center[1] - size / 2], [center[0] - size / 2, center[1] + size / 2],
                    [center[0] + size / 2, center[1] - size / 2]])
    dst = np.array([[0, 0], [0, image_size - 1], [image_size - 1, 0]])
    return umeyama(src, dst)


def warp_u8(image, matrix, out_shape):
    """image [H,W,C] uint8, matrix 3x3 mapping OUTPUT (col,row,1) -> INPUT (x,y) (= what skimage calls inverse_map) -> uint8 [Ho,Wo,C]."""
    img = np.asarray(image).astype(np.float64)
    rows, cols = img.shape[:2]
    Ho, Wo = out_shape
    M = np.asarray(matrix, np.float64)
    cc, rr = np.meshgrid(np.arange(Wo, dtype=np.float64), np.arange(Ho, dtype=np.float64))
    x = M[0, 0] * cc + M[0, 1] * rr + M[0, 2]
    y = M[1, 0] * cc + M[1, 1] * rr + M[1, 2]
    minr, minc, maxr, maxc = np.floor(y), np.floor(x), np.ceil(y), np.ceil(x)
    dr, dc = (y - minr)[..., None], (x - minc)[..., None]

    def px(r, c):
        ok = (r >= 0) & (r < rows) & (c >= 0) & (c < cols)
        v = img[np.clip(r, 0, rows - 1).astype(np.int64), np.clip(c, 0, cols - 1).astype(np.int64)]
        return np.where(ok[..., None], v, 0.0)

    top = (1 - dc) * px(minr, minc) + dc * px(minr, maxc)
    bot = (1 - dc) * px(maxr, minc) + dc * px(maxr, maxc)
    return ((1 - dr) * top + dr * bot).astype(np.uint8)


def resize_linear_u8(image, dsize):
    """cv2.resize(image, (w, h)) INTER_LINEAR for uint8 [H,W,C]."""
    src = np.asarray(image, np.uint8)
    sh, sw = src.shape[:2]
    dw, dh = dsize
    if (sw, sh) == (dw, dh):
        return src.copy()

    def taps(ssize, dsize_):
        scale = 1.0 / (dsize_ / ssize)
        ofs, co = np.zeros(dsize_, np.int64), np.zeros((dsize_, 2), np.int64)
        for d in range(dsize_):
            fx = f32((d + 0.5) * scale - 0.5)
            sx = int(np.floor(fx))
            fx = f32(fx - sx)
            if sx < 0:
                fx, sx = f32(0), 0
            if sx >= ssize - 1:
                fx, sx = f32(0), ssize - 1
            ofs[d] = sx
            co[d] = (int(np.rint(f32(f32(1) - fx) * f32(2048))), int(np.rint(fx * f32(2048))))
        return ofs, co

    xo, xa = taps(sw, dw)
    yo, ya = taps(sh, dh)
    s = src.astype(np.int64)
    x1 = np.minimum(xo + 1, sw - 1)
    hr = s[:, xo] * xa[None, :, 0, None] + s[:, x1] * xa[None, :, 1, None]               # [sh, dw, C], scaled by 2^11
    y1 = np.minimum(yo + 1, sh - 1)
    out = (((ya[:, 0, None, None] * (hr[yo] >> 4)) >> 16) + ((ya[:, 1, None, None] * (hr[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def convex_hull(pts):
    """Andrew's monotone chain on integer points -> hull vertices, collinear points dropped."""
    p = sorted(set(map(tuple, np.asarray(pts, np.int64).tolist())))
    if len(p) <= 2:
        return np.array(p, np.int64)
    cross = lambda o, a, b: (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lo, up = [], []
    for q in p:
        while len(lo) >= 2 and cross(lo[-2], lo[-1], q) <= 0:
            lo.pop()
        lo.append(q)
    for q in reversed(p):
        while len(up) >= 2 and cross(up[-2], up[-1], q) <= 0:
            up.pop()
        up.append(q)
    return np.array(lo[:-1] + up[:-1], np.int64)


def hull_mask(landmarks, shape):
    """create_mask: 1 outside / 0 inside the convex hull of landmarks.astype(int32)[..., :2]; rows [ymin, ymax], columns
    [round_half_up(x_left(y)), round_half_up(x_right(y))] of the closed polygon."""
    H, W = shape
    mask = np.ones((H, W), np.uint8)
    hull = convex_hull(np.asarray(landmarks).astype(np.int32)[..., :2])
    if len(hull) == 0:
        return mask
    n = len(hull)
    ymin, ymax = int(hull[:, 1].min()), int(hull[:, 1].max())
    for y in range(max(ymin, 0), min(ymax, H - 1) + 1):
        xl, xr = None, None
        for i in range(n):
            (x0, y0), (x1, y1) = hull[i], hull[(i + 1) % n]
            if min(y0, y1) <= y <= max(y0, y1):
                if y0 == y1:
                    cand = (float(min(x0, x1)), float(max(x0, x1)))
                else:
                    xx = x0 + (x1 - x0) * (y - y0) / (y1 - y0)
                    cand = (xx, xx)
                xl = cand[0] if xl is None else min(xl, cand[0])
                xr = cand[1] if xr is None else max(xr, cand[1])
        if xl is None:
            continue
        a, b = int(np.floor(xl + 0.5)), int(np.floor(xr + 0.5))
        a, b = max(a, 0), min(b, W - 1)
        if a <= b:
            mask[y, a:b + 1] = 0
    return mask


def to_u8(x):
    """(x * 255.0).astype(np.uint8) for float32 x in [0, 1] (truncation toward zero)."""
    return (np.asarray(x, f32) * f32(255.0)).astype(np.uint8)


def from_u8(u):
    return np.asarray(u).astype(f32) / f32(255.0)


def interp_bilinear(x, out_hw):
    """ATen upsample_bilinear2d, align_corners=False, float32.  x [N,C,H,W]."""
    x = np.asarray(x, f32)
    N, C, H, W = x.shape
    Ho, Wo = out_hw

    def idx(insz, outsz):
        scale = f32(insz) / f32(outsz)
        real = np.maximum((scale * (np.arange(outsz, dtype=f32) + f32(0.5)) - f32(0.5)).astype(f32), f32(0))
        i0 = real.astype(np.int64)
        i1 = i0 + (i0 < insz - 1)
        l1 = (real - i0.astype(f32)).astype(f32)
        return i0, i1, (f32(1) - l1).astype(f32), l1

    y0, y1, hy0, hy1 = idx(H, Ho)
    x0, x1, wx0, wx1 = idx(W, Wo)
    g = lambda yy, xx: x[:, :, yy][:, :, :, xx]
    top = (wx0 * g(y0, x0)).astype(f32) + (wx1 * g(y0, x1)).astype(f32)
    bot = (wx0 * g(y1, x0)).astype(f32) + (wx1 * g(y1, x1)).astype(f32)
    return ((hy0[:, None] * top).astype(f32) + (hy1[:, None] * bot).astype(f32)).astype(f32)
