"""MI355X drop-in for the reference Renderer (src/renderer/renderer.py:49-251, src/renderer/util.py).

Same constructor signature / asset files / forward() contract.  The orthographic projection, vertex normals, triangle
rasterisation (the pytorch3d rasterize_meshes call of renderer.py:185-193), barycentric attribute interpolation and
5-light Lambert shading run in libsmirk_hip.so (smirk_render_forward).  Forward only this round (renderer backward is
SURVEY.md §8 f-2); `rendered_img` background is exactly 0.0 as callers rely on (smirk_trainer.py:79,290; demo.py:146).
"""
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L


def load_obj(path):
    """Wavefront reader for the template mesh: `v`, `vt` and `f a/b c/d e/f` records (what pytorch3d.io.load_obj
    returns for head_template.obj, renderer.py:54-57).  -> verts[V,3] f32, faces[F,3] i64, uvs[T,2] f32, uvfaces[F,3] i64."""
    vs, vts, fs, fts = [], [], [], []
    with open(path, "r") as fh:
        for ln in fh:
            t = ln.split()
            if not t:
                continue
            if t[0] == "v":
                vs.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "vt":
                vts.append((float(t[1]), float(t[2])))
            elif t[0] == "f":
                idx = [tok.split("/") for tok in t[1:4]]
                fs.append([int(i[0]) - 1 for i in idx])
                fts.append([int(i[1]) - 1 if len(i) > 1 and i[1] else 0 for i in idx])
    return (np.asarray(vs, np.float32), np.asarray(fs, np.int64), np.asarray(vts, np.float32).reshape(-1, 2),
            np.asarray(fts, np.int64))


def submesh_faces(faces, keep):
    """Faces whose three corners are all kept, re-indexed into the sorted kept-vertex list (renderer.py:11-47)."""
    keep = np.unique(np.asarray(keep, np.int64))
    remap = np.full(int(faces.max()) + 1, -1, np.int64)
    remap[keep] = np.arange(keep.size)
    f = remap[faces]
    return f[(f >= 0).all(1)], keep


def normal_csr(faces, nv):
    """Per-vertex list of (face, corner) in the accumulation order of util.py:52-57 (corner-1 pass, corner-2 pass,
    corner-0 pass; faces ascending inside a pass) so the GPU gather reproduces the CPU index_add_ summation order."""
    nf = faces.shape[0]
    vert = np.concatenate([faces[:, 1], faces[:, 2], faces[:, 0]])
    face = np.concatenate([np.arange(nf)] * 3)
    corner = np.concatenate([np.full(nf, 1), np.full(nf, 2), np.full(nf, 0)])
    order = np.argsort(vert, kind="stable")
    ptr = np.zeros(nv + 1, np.int64)
    np.add.at(ptr, vert + 1, 1)
    return np.cumsum(ptr).astype(np.int32), face[order].astype(np.int32), corner[order].astype(np.int32)


class Renderer(nn.Module):
    def __init__(self, render_full_head=False, obj_filename='assets/head_template.obj'):
        super().__init__()
        self.image_size = 224
        verts, faces, uvs, uvfaces = load_obj(obj_filename)
        self.render_full_head = render_full_head
        nverts = int(faces.max()) + 1
        colors = torch.full((1, nverts, 3), 180.0).float() / 255.
        with open('assets/FLAME_masks/FLAME_masks.pkl', 'rb') as fh:
            self.flame_masks = pickle.load(fh, encoding='latin1')
        if not render_full_head:
            self.final_mask = self.flame_masks['face'].tolist()
            sub, keep = submesh_faces(faces, self.final_mask)
            colors = colors[:, self.final_mask, :]
        else:
            sub, keep = faces, np.arange(nverts, dtype=np.int64)
        # buffers with the reference's names (renderer.py:75-98)
        self.register_buffer('faces', torch.from_numpy(sub)[None])
        self.register_buffer('face_colors', colors[0][torch.from_numpy(sub)][None])
        uv = torch.from_numpy(uvs)[None]
        self.register_buffer('raw_uvcoords', uv)
        uv3 = torch.cat([uv, uv[:, :, 0:1] * 0. + 1.], -1) * 2 - 1
        uv3[..., 1] = -uv3[..., 1]
        self.register_buffer('uvcoords', uv3)
        self.register_buffer('uvfaces', torch.from_numpy(uvfaces)[None])
        self.register_buffer('face_uvcoords', uv3[0][torch.from_numpy(uvfaces)][None])
        pi = np.pi
        cf = [1 / np.sqrt(4 * pi)] + [((2 * pi) / 3) * (np.sqrt(3 / (4 * pi)))] * 3 + \
             [(pi / 4) * 3 * (np.sqrt(5 / (12 * pi)))] * 3 + \
             [(pi / 4) * (3 / 2) * (np.sqrt(5 / (12 * pi))), (pi / 4) * (1 / 2) * (np.sqrt(5 / (4 * pi)))]
        self.register_buffer('constant_factor', torch.tensor(cf).float())
        # kernel-side topology (non-persistent)
        ptr, nf, nc = normal_csr(sub, keep.size)
        nb = lambda n, a: self.register_buffer(n, torch.from_numpy(np.ascontiguousarray(a)), persistent=False)
        nb('_k_keep', keep.astype(np.int32))
        nb('_k_faces', sub.astype(np.int32))
        nb('_k_nrm_ptr', ptr)
        nb('_k_nrm_face', nf)
        nb('_k_nrm_corner', nc)
        self._dims = dict(V=nverts, Vf=int(keep.size), Ff=int(sub.shape[0]), nnz=int(nf.size))
        self._ws = L.Workspace()
        self._mesh, self._mesh_key = None, None

    def _struct(self):
        key = self._k_keep.data_ptr()
        if self._mesh is None or self._mesh_key != key:
            if not self._k_keep.is_cuda:
                raise L.SmirkHipError("Renderer module is on the CPU: move it to the HIP device; there is no CPU path")
            m = L.SmirkRenderMesh()
            for k, v in self._dims.items():
                setattr(m, k, v)
            I = torch.int32
            m.keep, m.faces = L.ptr(self._k_keep, I), L.ptr(self._k_faces, I)
            m.nrm_ptr, m.nrm_face, m.nrm_corner = L.ptr(self._k_nrm_ptr, I), L.ptr(self._k_nrm_face, I), L.ptr(self._k_nrm_corner, I)
            self._mesh, self._mesh_key = m, key
        return self._mesh

    def forward(self, vertices, cam_params, _aux=False, **landmarks):
        """vertices [B,V,3], cam_params [B,3]=(scale,tx,ty) -> dict(rendered_img [B,3,224,224], transformed_vertices,
        <landmark key> [B,L,2] ...)   (renderer.py:100-118).  `_aux=True` also returns pix_to_face / bary / zbuf / normals."""
        vertices, cam = L.as_f32c(vertices), L.as_f32c(cam_params)
        if vertices.shape[1] != self._dims['V']:
            raise L.SmirkHipError("vertex count does not match the template mesh")
        keys = list(landmarks.keys())
        lms = [L.as_f32c(landmarks[k]) for k in keys]
        if torch.is_grad_enabled() and any(t.requires_grad for t in [vertices, cam] + lms):
            # training step 1 (smirk_trainer.py:46-48,362): image / landmark losses back-propagate to vertices and camera
            if _aux:
                raise L.SmirkHipError("_aux outputs are not available on the differentiable path")
            res = _RenderFunction.apply(self, len(lms), vertices, cam, *lms)
            out = {'rendered_img': res[0], 'transformed_vertices': res[1]}
            out.update({k: v for k, v in zip(keys, res[2:])})
            return out
        img, tv, proj, aux = self._launch(vertices, cam, lms, _aux, False)
        out = {'rendered_img': img, 'transformed_vertices': tv}
        out.update({k: v for k, v in zip(keys, proj)})
        if _aux:
            out['_aux'] = aux
        return out

    def _launch(self, vertices, cam, lms, want_aux, want_p2f):
        B, dev = vertices.shape[0], vertices.device
        H = W = self.image_size
        lib, mesh = L.lib(), self._struct()
        tv = torch.empty_like(vertices)
        img = torch.empty(B, 3, H, W, device=dev)
        p2f = bary = zbuf = nrm = None
        if want_aux or want_p2f:
            p2f = torch.empty(B, H, W, dtype=torch.int64, device=dev)
        if want_aux:
            bary = torch.empty(B, H, W, 3, device=dev)
            zbuf = torch.empty(B, H, W, device=dev)
            nrm = torch.empty(B, self._dims['Vf'], 3, device=dev)
        nws = lib.smirk_render_workspace_bytes(mesh, B, H, W)
        ws = self._ws.get(nws, dev)
        P = L.ptr
        L.check(lib.smirk_render_forward(mesh, B, H, W, P(vertices), P(cam), P(tv), P(img), P(p2f, torch.int64, True),
                                         P(bary, allow_none=True), P(zbuf, allow_none=True), P(nrm, allow_none=True),
                                         P(ws, torch.uint8), nws, L.stream_ptr()))
        proj = []
        for lm in lms:
            o = torch.empty(B, lm.shape[1], 2, device=dev)
            L.check(lib.smirk_project_landmarks(P(lm), P(cam), B, lm.shape[1], P(o), L.stream_ptr()))
            proj.append(o)
        if self.render_full_head:
            # reference quirk (renderer.py:141): render() adds the near-plane shift `z += 10` IN PLACE; with the face sub-mesh that hits
            # an advanced-index copy, in full-head mode it hits the very tensor forward() returns as 'transformed_vertices'
            tv[..., 2] += 10
        aux = dict(pix_to_face=p2f, bary=bary, zbuf=zbuf, normals=nrm) if want_aux else p2f
        return img, tv, proj, aux

    def _launch_backward(self, vertices, cam, p2f, lms, g_img, g_tv, g_lms):
        B, dev = vertices.shape[0], vertices.device
        H = W = self.image_size
        lib, mesh = L.lib(), self._struct()
        d_verts, d_cam = torch.empty_like(vertices), torch.empty_like(cam)
        g_img = None if g_img is None else L.as_f32c(g_img)
        g_tv = None if g_tv is None else L.as_f32c(g_tv)
        nws = lib.smirk_render_backward_workspace_bytes(mesh, B, H, W)
        ws = self._ws.get(nws, dev)
        P = L.ptr
        L.check(lib.smirk_render_backward(mesh, B, H, W, P(vertices), P(cam), P(p2f, torch.int64), P(g_img, allow_none=True),
                                          P(g_tv, allow_none=True), P(d_verts), P(d_cam), P(ws, torch.uint8), nws, L.stream_ptr()))
        d_lms = []
        for lm, g in zip(lms, g_lms):
            if g is None:
                d_lms.append(None)
                continue
            g = L.as_f32c(g)
            d = torch.empty_like(lm)
            L.check(lib.smirk_project_landmarks_backward(P(lm), P(cam), P(g), B, lm.shape[1], P(d), P(d_cam), L.stream_ptr()))
            d_lms.append(d)
        return d_verts, d_cam, d_lms

    def render(self, vertices, transformed_vertices=None):
        """Shaded image only (renderer.py:121-168).  `transformed_vertices` is recomputed from the camera inside
        smirk_render_forward, so this entry exists for signature compatibility and requires it to be None."""
        raise L.SmirkHipError("Renderer.render(vertices, transformed_vertices) is not exposed separately: call forward()")


class _RenderFunction(torch.autograd.Function):
    """autograd bridge: forward = smirk_render_forward (+ landmark projections) keeping pix_to_face, backward = smirk_render_backward."""

    @staticmethod
    def forward(ctx, module, n_lm, vertices, cam, *lms):
        img, tv, proj, p2f = module._launch(vertices, cam, list(lms), False, True)
        ctx.module = module
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(vertices, cam, p2f, *lms)
        return (img, tv, *proj)

    @staticmethod
    def backward(ctx, g_img, g_tv, *g_lms):
        vertices, cam, p2f, *lms = ctx.saved_tensors
        d_verts, d_cam, d_lms = ctx.module._launch_backward(vertices, cam, p2f, lms, g_img, g_tv, g_lms)
        return (None, None, d_verts, d_cam, *d_lms)
