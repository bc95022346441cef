"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Imports the REAL reference modules from /root/reference on CPU torch, with the shims of
SURVEY.md App. D.  Exists only in the build container (the GPU box has no /root/reference):
used by ``oracle/make_golden.py`` to emit ``tests/golden/*.npz`` and by the ``not gpu`` tests
that pin the restatements in ``oracle/*_ref.py`` against the reference's own code.

Shims (nothing else is patched):
  * numpy>=2 removed np.float_/np.complex_/np.unicode_ which FLAME.py:21-24 aliases at import;
  * ``cv2`` (imported, unused, by renderer/util.py:5) -> empty module;
  * ``pytorch3d.{structures,io,renderer.mesh}`` -> Meshes / load_obj / rasterize_meshes backed by
    oracle/raster_ref.c (App. B restatement — pytorch3d itself is NOT on disk: parity unpinned);
  * ``timm`` -> create_model backed by oracle/mobilenet_ref.py (App. A restatement — parity unpinned).
"""
import contextlib
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("SMIRK_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "src", "FLAME"))


def _install_shims():
    import torch
    for name, typ in (("float_", np.float64), ("complex_", np.complex128), ("unicode_", np.str_)):
        if not hasattr(np, name):
            setattr(np, name, typ)
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")

    from . import assets as A
    from . import render_ref as R

    class Meshes:  # the 4 accessors rasterize_meshes needs; equal-sized meshes only
        def __init__(self, verts, faces):
            self._v, self._f = verts, faces

        def verts_packed(self):
            return self._v.reshape(-1, 3)

        def faces_packed(self):
            B, Fn = self._f.shape[:2]
            off = (torch.arange(B, dtype=torch.long) * self._v.shape[1])[:, None, None]
            return (self._f.long() + off).reshape(-1, 3)

        def num_meshes(self):
            return self._v.shape[0]

        def num_faces_per_mesh(self):
            return self._f.shape[1]

    def load_obj(path):
        v, vt, f, ft = A.parse_obj(path)
        faces = types.SimpleNamespace(verts_idx=torch.from_numpy(f), textures_idx=torch.from_numpy(ft))
        aux = types.SimpleNamespace(verts_uvs=torch.from_numpy(vt))
        return torch.from_numpy(v), faces, aux

    def rasterize_meshes(meshes, image_size=224, blur_radius=0.0, faces_per_pixel=1, bin_size=None,
                         max_faces_per_bin=None, perspective_correct=False, **kw):
        assert blur_radius == 0.0 and faces_per_pixel == 1 and not perspective_correct
        vp = meshes.verts_packed().detach().cpu().numpy().astype(np.float32)
        fp = meshes.faces_packed().detach().cpu().numpy()
        fv = vp[fp]  # [B*Ff,3,3]
        B, Ff = meshes.num_meshes(), meshes.num_faces_per_mesh()
        p2f, zbuf, bary = R.rasterize_naive(fv.reshape(B, Ff, 3, 3), image_size, image_size)
        p2f = p2f.astype(np.int64)
        bary_t = torch.from_numpy(bary)
        vpk = meshes.verts_packed()
        if vpk.requires_grad:
            # differentiable barycentrics (oracle/render_torch_ref.py): forward values stay the C rasteriser's, the gradient is
            # autograd of pytorch3d's formula — what its BarycentricCoordsBackward computes analytically.
            from .render_torch_ref import bary_differentiable
            fvt = vpk[meshes.faces_packed()].reshape(B, Ff, 3, 3)
            bd = bary_differentiable(fvt, torch.from_numpy(p2f), image_size, image_size)
            bary_t = bary_t + (bd - bd.detach())
        off = (np.arange(B, dtype=np.int64) * Ff)[:, None, None]
        p2f = np.where(p2f >= 0, p2f + off, -1)
        return (torch.from_numpy(p2f)[..., None], torch.from_numpy(zbuf)[..., None],
                bary_t[:, :, :, None, :], torch.full(p2f.shape + (1,), -1.0))

    p3d = types.ModuleType("pytorch3d")
    st = types.ModuleType("pytorch3d.structures"); st.Meshes = Meshes
    io = types.ModuleType("pytorch3d.io"); io.load_obj = load_obj
    rn = types.ModuleType("pytorch3d.renderer")
    rm = types.ModuleType("pytorch3d.renderer.mesh"); rm.rasterize_meshes = rasterize_meshes
    rn.mesh = rm
    p3d.structures, p3d.io, p3d.renderer = st, io, rn
    sys.modules.update({"pytorch3d": p3d, "pytorch3d.structures": st, "pytorch3d.io": io,
                        "pytorch3d.renderer": rn, "pytorch3d.renderer.mesh": rm})

    from . import mobilenet_ref as M
    timm = types.ModuleType("timm")
    timm.create_model = lambda name, pretrained=True, features_only=True: M.create_model(name)
    sys.modules["timm"] = timm


@contextlib.contextmanager
def reference(sandbox_dir):
    """Context: cwd = sandbox (has assets/), reference ``src`` package importable. Yields a namespace
    with the four reference classes."""
    assert available(), "reference tree not present (expected only in the build container)"
    _install_shims()
    old_cwd, old_path = os.getcwd(), list(sys.path)
    old_bc = sys.dont_write_bytecode
    sys.dont_write_bytecode = True  # reference tree is read-only
    os.chdir(sandbox_dir)
    sys.path.insert(0, REF_ROOT)
    try:
        from src.FLAME.FLAME import FLAME
        from src.FLAME import lbs
        from src.renderer.renderer import Renderer
        from src.renderer import util as render_util
        from src.smirk_generator import SmirkGenerator
        from src.smirk_encoder import SmirkEncoder
        from src.utils import masking
        yield types.SimpleNamespace(FLAME=FLAME, lbs=lbs, Renderer=Renderer, render_util=render_util,
                                    SmirkGenerator=SmirkGenerator, SmirkEncoder=SmirkEncoder, masking=masking)
    finally:
        os.chdir(old_cwd)
        sys.path[:] = old_path
        sys.dont_write_bytecode = old_bc
