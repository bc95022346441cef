"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference-side asset helpers.  The seeded synthetic assets / inputs themselves live in synthdata.py (INPUT generation is
shared so the GPU path and this oracle see identical bits); this module adds the OBJ reader used by the pytorch3d stub and the
script that packs the reference's public assets into tests/golden/assets_bundle.npz.

The reference reads its assets by cwd-relative path (FLAME.py:50-51,54,81-82,94,111;
renderer.py:50,54,65).  The licence-gated FLAME2020/generic_model.pkl is absent, and
/root/reference does not exist on the GPU box at all, so tests build an ``assets/`` tree
from ``tests/golden/assets_bundle.npz`` (topology/embeddings taken from the reference's
public assets by ``oracle/make_golden.py``) plus a seeded synthetic FLAME model
(SURVEY.md §8(d)): every file has the on-disk format the reference loader expects.
"""
import os
import pickle

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

from synthdata import (BUNDLE, F, V, load_bundle, synth_cam, synth_flame_model, synth_flame_params,  # noqa: F401,E402
                             synth_generator_input, synth_images, write_obj, write_sandbox)


def parse_obj(path):
    """Minimal OBJ reader: v / vt / f a/b c/d e/f lines (head_template.obj has nothing else).

    Returns verts[V,3] f32, uvs[T,2] f32, faces[F,3] i64 (0-based), tfaces[F,3] i64 (0-based).
    """
    v, vt, f, ft = [], [], [], []
    with open(path, "r") as fh:
        for line in fh:
            if line.startswith("v "):
                v.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("vt "):
                vt.append([float(x) for x in line.split()[1:3]])
            elif line.startswith("f "):
                toks = line.split()[1:]
                a, b = [], []
                for t in toks:
                    p = t.split("/")
                    a.append(int(p[0]) - 1)
                    b.append(int(p[1]) - 1 if len(p) > 1 and p[1] else 0)
                f.append(a[:3])
                ft.append(b[:3])
    return (np.asarray(v, np.float32), np.asarray(vt, np.float32),
            np.asarray(f, np.int64), np.asarray(ft, np.int64))


def build_bundle_from_reference(ref_root, out_npz=BUNDLE):
    """Pack the reference's public assets into one npz (run in the build container only)."""
    import torch  # landmark_embedding.npy pickles torch tensors
    a = os.path.join(ref_root, "assets")
    verts, uvs, faces, tfaces = parse_obj(os.path.join(a, "head_template.obj"))
    assert verts.shape == (V, 3) and faces.shape == (F, 3)
    emb = np.load(os.path.join(a, "landmark_embedding.npy"), allow_pickle=True, encoding="latin1")[()]
    masks = pickle.load(open(os.path.join(a, "FLAME_masks", "FLAME_masks.pkl"), "rb"), encoding="latin1")
    mp = np.load(os.path.join(a, "mediapipe_landmark_embedding", "mediapipe_landmark_embedding.npz"))
    out = dict(
        obj_verts=verts, obj_uvs=uvs, obj_faces=faces.astype(np.int32), obj_tfaces=tfaces.astype(np.int32),
        l_eyelid=np.load(os.path.join(a, "l_eyelid.npy")), r_eyelid=np.load(os.path.join(a, "r_eyelid.npy")),
        static_lmk_faces_idx=np.asarray(emb["static_lmk_faces_idx"]),
        static_lmk_bary_coords=np.asarray(emb["static_lmk_bary_coords"]),
        dynamic_lmk_faces_idx=emb["dynamic_lmk_faces_idx"].numpy(),
        dynamic_lmk_bary_coords=emb["dynamic_lmk_bary_coords"].numpy(),
        full_lmk_faces_idx=np.asarray(emb["full_lmk_faces_idx"]),
        full_lmk_bary_coords=np.asarray(emb["full_lmk_bary_coords"]),
        mp_lmk_face_idx=mp["lmk_face_idx"], mp_lmk_b_coords=mp["lmk_b_coords"],
        mp_landmark_indices=mp["landmark_indices"],
    )
    for k, val in masks.items():
        out["mask_" + k] = np.asarray(val)
    tri = np.load(os.path.join(a, "FLAME_masks", "FLAME_masks_triangles.npy"), allow_pickle=True).item()
    for k, val in tri.items():
        out["tri_" + k] = np.asarray(val).astype(np.int32)
    os.makedirs(os.path.dirname(out_npz), exist_ok=True)
    np.savez_compressed(out_npz, **out)
    return out_npz
