"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy float32 restatement of the reference renderer:
  src/renderer/util.py:10-28 (face_vertices), :30-62 (vertex_normals), :64-78 (batch_orth_proj)
  src/renderer/renderer.py:11-47 (keep_vertices_and_update_faces), :50-98 (ctor), :100-118 (forward),
                           :121-168 (render), :171-207 (rasterize), :239-250 (add_directionlight)
The third-party rasteriser is oracle/raster_ref.c (PARITY UNPINNED, see its header); a slow pure-numpy
twin (`rasterize_numpy`) cross-checks the C build on tiny cases.
"""
import ctypes
import os
import pickle
import subprocess

import numpy as np

from . import assets as A

f32 = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(HERE, "_build")
_LIB = None


def build_c(force=False):
    """gcc build of raster_ref.c -> oracle/_build/libraster_ref.so (no FMA contraction)."""
    os.makedirs(_BUILD, exist_ok=True)
    so = os.path.join(_BUILD, "libraster_ref.so")
    src = os.path.join(HERE, "raster_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared", "-fPIC", src, "-o", so, "-lm"]
        subprocess.check_call(cmd)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_c())
        _LIB.smirk_oracle_rasterize_naive.restype = None
    return _LIB


def rasterize_naive(face_verts, H=224, W=224):
    """face_verts [B,Ff,3,3] f32 (pytorch3d NDC) -> pix_to_face[B,H,W] i32 (local idx / -1), zbuf, bary[B,H,W,3]."""
    fv = np.ascontiguousarray(face_verts, dtype=f32)
    B, Ff = fv.shape[:2]
    p2f = np.empty((B, H, W), np.int32)
    zb = np.empty((B, H, W), f32)
    bary = np.empty((B, H, W, 3), f32)
    c = ctypes.c_void_p
    _lib().smirk_oracle_rasterize_naive(c(fv.ctypes.data), B, Ff, H, W, c(p2f.ctypes.data), c(zb.ctypes.data),
                                        c(bary.ctypes.data))
    return p2f, zb, bary


def rasterize_numpy(face_verts, H, W):
    """Same algorithm in numpy float32, vectorised over faces per pixel (tiny cases only)."""
    fv = np.asarray(face_verts, f32)
    B, Ff = fv.shape[:2]
    p2f = -np.ones((B, H, W), np.int32); zb = -np.ones((B, H, W), f32); bary = -np.ones((B, H, W, 3), f32)
    E = lambda px, py, ax, ay, bx, by: ((px - ax) * (by - ay)).astype(f32) - ((py - ay) * (bx - ax)).astype(f32)
    for n in range(B):
        x0, y0, z0 = fv[n, :, 0, 0], fv[n, :, 0, 1], fv[n, :, 0, 2]
        x1, y1, z1 = fv[n, :, 1, 0], fv[n, :, 1, 1], fv[n, :, 1, 2]
        x2, y2, z2 = fv[n, :, 2, 0], fv[n, :, 2, 1], fv[n, :, 2, 2]
        fa = E(x0, y0, x1, y1, x2, y2)
        area = (E(x2, y2, x0, y0, x1, y1) + f32(1e-8)).astype(f32)
        xmin, xmax = np.minimum(x0, np.minimum(x1, x2)), np.maximum(x0, np.maximum(x1, x2))
        ymin, ymax = np.minimum(y0, np.minimum(y1, y2)), np.maximum(y0, np.maximum(y1, y2))
        zmin = np.minimum(z0, np.minimum(z1, z2))                    # z_invalid = zmin < kEpsilon (raster_ref.c header)
        for yi in range(H):
            yf = f32(-1.0) + f32(f32(2.0) * f32(H - 1 - yi) + f32(1.0)) / f32(H)
            for xi in range(W):
                xf = f32(-1.0) + f32(f32(2.0) * f32(W - 1 - xi) + f32(1.0)) / f32(W)
                with np.errstate(all="ignore"):
                    w0 = (E(xf, yf, x1, y1, x2, y2) / area).astype(f32)
                    w1 = (E(xf, yf, x2, y2, x0, y0) / area).astype(f32)
                    w2 = (E(xf, yf, x0, y0, x1, y1) / area).astype(f32)
                    pz = ((w0 * z0).astype(f32) + (w1 * z1).astype(f32)).astype(f32) + (w2 * z2).astype(f32)
                ok = (np.abs(fa) > f32(1e-8)) & ~((xf > xmax) | (xf < xmin) | (yf > ymax) | (yf < ymin) | (zmin < f32(1e-8)))
                ok &= ~(pz < 0) & (w0 > 0) & (w1 > 0) & (w2 > 0)
                idx = np.nonzero(ok)[0]
                if idx.size:
                    k = idx[np.lexsort((idx, pz[idx]))[0]]
                    p2f[n, yi, xi] = k; zb[n, yi, xi] = pz[k]; bary[n, yi, xi] = (w0[k], w1[k], w2[k])
    return p2f, zb, bary


# ---- util.py:64-78 + renderer.py:101-102 ----------------------------------------------------
def orth_proj_flip(X, cam):
    X = np.asarray(X, f32); cam = np.asarray(cam, f32).reshape(-1, 1, 3)
    xy = (X[:, :, :2] + cam[:, :, 1:]).astype(f32)
    Xn = (cam[:, :, 0:1] * np.concatenate([xy, X[:, :, 2:]], 2)).astype(f32)
    Xn[:, :, 1:] = -Xn[:, :, 1:]
    return Xn


# ---- renderer.py:11-47 -----------------------------------------------------------------------
def keep_vertices_and_update_faces(faces, keep):
    keep = np.unique(np.asarray(keep, np.int64))
    n = int(faces.max()) + 1
    new = -np.ones(n, np.int64)
    new[keep] = np.arange(len(keep))
    valid = (new[faces] != -1).all(1)
    return new[faces[valid]]


# ---- util.py:30-62 ---------------------------------------------------------------------------
def vertex_normals(verts, faces):
    """verts [B,Nv,3] f32, faces [Ff,3] -> [B,Nv,3]; sequential accumulation order of index_add_."""
    verts = np.asarray(verts, f32)
    B, Nv = verts.shape[:2]
    out = np.zeros((B, Nv, 3), f32)
    cr = lambda a, b: np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                                a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], 1).astype(f32)
    for b in range(B):
        vf = verts[b][faces]                                             # [Ff,3,3]
        np.add.at(out[b], faces[:, 1], cr(vf[:, 2] - vf[:, 1], vf[:, 0] - vf[:, 1]))
        np.add.at(out[b], faces[:, 2], cr(vf[:, 0] - vf[:, 2], vf[:, 1] - vf[:, 2]))
        np.add.at(out[b], faces[:, 0], cr(vf[:, 1] - vf[:, 0], vf[:, 2] - vf[:, 0]))
    nrm = np.sqrt((out * out).sum(-1, keepdims=True, dtype=f32)).astype(f32)
    return (out / np.maximum(nrm, f32(1e-6))).astype(f32)


class RendererRef:
    """Mirror of Renderer(render_full_head=False) (renderer.py:50-98)."""

    def __init__(self, assets_root):
        a = os.path.join(assets_root, "assets")
        _, _, faces, _ = A.parse_obj(os.path.join(a, "head_template.obj"))
        masks = pickle.load(open(os.path.join(a, "FLAME_masks", "FLAME_masks.pkl"), "rb"), encoding="latin1")
        self.final_mask = np.asarray(masks["face"], np.int64)
        self.faces = keep_vertices_and_update_faces(faces, self.final_mask)      # [3408,3]
        self.image_size = 224
        L = np.array([[-1, 1, 1], [1, 1, 1], [-1, -1, 1], [1, -1, 1], [0, 0, 1]], f32)
        self.light_dirs = (L / np.maximum(np.sqrt((L * L).sum(1, keepdims=True)), 1e-12)).astype(f32)

    def forward(self, vertices, cam, **landmarks):
        tv = orth_proj_flip(vertices, cam)
        out = {k: orth_proj_flip(v, cam)[..., :2] for k, v in landmarks.items()}
        img, aux = self.render(np.asarray(vertices, f32), tv)
        out.update(rendered_img=img, transformed_vertices=tv, _aux=aux)
        return out

    def raster_input(self, tv):
        """renderer.py:140-144,172-173: sub-mesh, z+10, negate xy -> face_verts [B,Ff,3,3]."""
        t = tv[:, self.final_mask].copy()
        t[:, :, 2] = t[:, :, 2] + f32(10)
        t[..., :2] = -t[..., :2]
        return t[:, self.faces]

    def render(self, vertices, tv, H=224, W=224):
        B = vertices.shape[0]
        v = vertices[:, self.final_mask]
        normals = vertex_normals(v, self.faces)                                  # [B,1787,3]
        fn = normals[:, self.faces]                                              # [B,Ff,3,3]
        col = np.full((B, self.faces.shape[0], 3, 3), f32(180.0) / f32(255.0), f32)
        attr = np.concatenate([col, fn], -1)                                     # [B,Ff,3,6]
        p2f, zb, bary = rasterize_naive(self.raster_input(tv), H, W)
        mask = p2f < 0
        idx = np.where(mask, 0, p2f)
        vals = attr[np.arange(B)[:, None, None], idx]                            # [B,H,W,3,6]
        pv = (bary[..., None] * vals).astype(f32)
        pix = ((pv[..., 0, :] + pv[..., 1, :]).astype(f32) + pv[..., 2, :]).astype(f32)   # sum over corners, in order
        pix[mask] = 0
        albedo, nimg = pix[..., :3], pix[..., 3:6]
        # add_directionlight (renderer.py:239-250): mean over 5 lights of clamp(n.l,0,1)*1.7
        ndl = np.zeros((B, H, W, 5), f32)
        for li in range(5):
            d = self.light_dirs[li]
            t = ((nimg[..., 0] * d[0]).astype(f32) + (nimg[..., 1] * d[1]).astype(f32)).astype(f32) + (nimg[..., 2] * d[2]).astype(f32)
            ndl[..., li] = np.clip(t, 0, 1) * f32(1.7)
        sh = ndl[..., 0]
        for li in range(1, 5):
            sh = (sh + ndl[..., li]).astype(f32)
        sh = (sh / f32(5)).astype(f32)
        img = (albedo * sh[..., None]).astype(f32).transpose(0, 3, 1, 2).copy()
        return img, dict(pix_to_face=p2f, bary=bary, zbuf=zb, normals=normals)
