"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of the deterministic parts of src/utils/masking.py:
  :39-47 triangle_area, :71-102 masking (with the random fields passed in), :116-129 transfer_pixels,
  :132-181 mesh_based_mask_uniform_faces (sampling weights :144-160; the coords= path :167-177).
Pinned against the reference functions themselves by tests/golden/masking_golden.npz (oracle/make_golden.py) — the reference's
torch.multinomial / rand / randn / bernoulli draws are not reproducible across devices, so those enter as explicit inputs.
"""
import numpy as np
from scipy.ndimage import maximum_filter

from .render_ref import vertex_normals

f32 = np.float32


def face_weights(tv, faces, prob):
    tv = np.asarray(tv, f32)
    n = vertex_normals(tv, faces)                                     # full mesh, index_add_ order
    nz = ((n[:, faces[:, 0], 2] + n[:, faces[:, 1], 2]).astype(f32) + n[:, faces[:, 2], 2]).astype(f32) / f32(3)
    p = np.where(nz < f32(0.05), np.asarray(prob, f32)[None], f32(0))
    v = tv[:, faces]                                                  # [B,F,3,3]
    x1, y1, x2, y2, x3, y3 = v[..., 0, 0], v[..., 0, 1], v[..., 1, 0], v[..., 1, 1], v[..., 2, 0], v[..., 2, 1]
    s = (x1 * y2).astype(f32)
    for t, sign in ((x2 * y3, 1), (x3 * y1, 1), (x2 * y1, -1), (x3 * y2, -1), (x1 * y3, -1)):
        s = (s + t.astype(f32) if sign > 0 else s - t.astype(f32)).astype(f32)
    return (p * (f32(0.5) * np.abs(s))).astype(f32)


def points_from_coords(tv, faces, idx, bary, S=224):
    tv = np.asarray(tv, f32)
    B = tv.shape[0]
    tri = faces[idx]
    lv = tv[np.arange(B)[:, None, None], tri]                         # [B,N,3,3]
    b = np.asarray(bary, f32)
    p = ((lv[:, :, 0] * b[..., 0:1]).astype(f32) + (lv[:, :, 1] * b[..., 1:2]).astype(f32)).astype(f32) + (lv[:, :, 2] * b[..., 2:3]).astype(f32)
    q = (f32(0.5) * (f32(1) + p.astype(f32)) * f32(S)).astype(f32)
    out = np.trunc(q).astype(np.int64)
    out[..., 0] = np.clip(out[..., 0], 0, S - 1)
    out[..., 1] = np.clip(out[..., 1], 0, S - 1)
    return out, p.astype(f32)


def _maxpool(x, r):
    return maximum_filter(x, size=(1, 1, 2 * r + 1, 2 * r + 1), mode="constant", cval=-np.inf)


def masking(img, mask, extra, wr, rendered_mask=None, noise_mult=None, random_field=None):
    img, mask, extra = np.asarray(img, f32), np.asarray(mask, f32), np.asarray(extra, f32).copy()
    m = (f32(1) - _maxpool(f32(1) - mask, wr)).astype(f32)
    if rendered_mask is not None:
        m = (m * (f32(1) - np.asarray(rendered_mask, f32))).astype(f32)
    masked = (img * m).astype(f32)
    if noise_mult is not None:
        extra = (extra * np.asarray(noise_mult, f32)).astype(f32)
    if random_field is not None:
        extra = (extra * (f32(1) - _maxpool(np.asarray(random_field, f32), 5))).astype(f32)
    return np.where(extra > 0, extra, masked).astype(f32)


def transfer_pixels(img, p1, p2, rbound=None):
    img = np.asarray(img, f32)
    out = np.zeros_like(img)
    for b in range(img.shape[0]):
        n = p1.shape[1] if rbound is None else int(rbound[b])
        for l in range(n):                                             # sequential: the last write to a pixel wins
            out[b, :, p2[b, l, 1], p2[b, l, 0]] = img[b, :, p1[b, l, 1], p1[b, l, 0]]
    return out
