"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy float32 restatement of the reference FLAME layer:
  src/FLAME/FLAME.py:50-115 (constructor / buffers), :117-159 (dynamic landmark LUT), :232-315 (forward)
  src/FLAME/lbs.py:26-32 (rot_mat_to_euler), :101-137 (vertices2landmarks), :140-227 (lbs),
                   :230-271 (vertices2joints, blend_shapes), :274-305 (batch_rodrigues),
                   :308-378 (transform_mat, batch_rigid_transform)
Pinned against the real reference classes: oracle/make_golden.py runs them (build container) and commits their outputs as
tests/golden/flame_golden.npz; tests/test_cpu_suite.py::test_flame_oracle_vs_reference_golden compares this restatement with them everywhere.
"""
import os
import pickle

import numpy as np

f32 = np.float32


def _to_np(a, dtype=np.float32):
    if "scipy.sparse" in str(type(a)):
        a = a.todense()
    return np.array(a, dtype=dtype)


class FlameRef:
    """Mirror of FLAME.__init__ (FLAME.py:50-115): same files, same buffer names/shapes."""

    def __init__(self, assets_root, n_shape=300, n_exp=50):
        a = os.path.join(assets_root, "assets")
        with open(os.path.join(a, "FLAME2020", "generic_model.pkl"), "rb") as fh:
            m = pickle.load(fh, encoding="latin1")
        self.n_shape, self.n_exp = n_shape, n_exp
        self.faces = _to_np(m["f"], np.int64)
        self.v_template = _to_np(m["v_template"])
        sd = _to_np(m["shapedirs"])
        self.shapedirs = np.concatenate([sd[:, :, :n_shape], sd[:, :, 300:300 + n_exp]], 2)  # FLAME.py:67-69
        npb = m["posedirs"].shape[-1]
        self.posedirs = _to_np(np.reshape(m["posedirs"], [-1, npb]).T)                     # FLAME.py:71-73
        self.J_regressor = _to_np(m["J_regressor"])
        parents = _to_np(m["kintree_table"][0]).astype(np.int64)
        parents[0] = -1
        self.parents = parents
        self.lbs_weights = _to_np(m["weights"])
        self.l_eyelid = np.load(os.path.join(a, "l_eyelid.npy")).astype(f32)[None]
        self.r_eyelid = np.load(os.path.join(a, "r_eyelid.npy")).astype(f32)[None]
        emb = np.load(os.path.join(a, "landmark_embedding.npy"), allow_pickle=True, encoding="latin1")[()]
        g = lambda x: x.numpy() if hasattr(x, "numpy") else np.asarray(x)
        self.lmk_faces_idx = g(emb["static_lmk_faces_idx"]).astype(np.int64)
        self.lmk_bary_coords = g(emb["static_lmk_bary_coords"]).astype(f32)
        self.dynamic_lmk_faces_idx = g(emb["dynamic_lmk_faces_idx"]).astype(np.int64)
        self.dynamic_lmk_bary_coords = g(emb["dynamic_lmk_bary_coords"]).astype(f32)
        self.full_lmk_faces_idx = g(emb["full_lmk_faces_idx"]).astype(np.int64)
        self.full_lmk_bary_coords = g(emb["full_lmk_bary_coords"]).astype(f32)
        chain, cur = [], 1                                                                  # FLAME.py:104-109
        while cur != -1:
            chain.append(cur)
            cur = int(self.parents[cur])
        self.neck_kin_chain = np.asarray(chain, np.int64)
        mp = np.load(os.path.join(a, "mediapipe_landmark_embedding", "mediapipe_landmark_embedding.npz"))
        self.mp_lmk_faces_idx = mp["lmk_face_idx"].astype("int32").astype(np.int64)
        self.mp_lmk_bary_coords = mp["lmk_b_coords"].astype(f32)

    # ---- lbs.py:274-305 ---------------------------------------------------------------
    @staticmethod
    def batch_rodrigues(rv):
        rv = rv.astype(f32)
        angle = np.sqrt(((rv + f32(1e-8)) ** 2).sum(1, keepdims=True, dtype=f32)).astype(f32)
        d = (rv / angle).astype(f32)
        c = np.cos(angle)[:, None].astype(f32)
        s = np.sin(angle)[:, None].astype(f32)
        rx, ry, rz = d[:, 0:1], d[:, 1:2], d[:, 2:3]
        z = np.zeros_like(rx)
        K = np.concatenate([z, -rz, ry, rz, z, -rx, -ry, rx, z], 1).reshape(-1, 3, 3).astype(f32)
        I = np.eye(3, dtype=f32)[None]
        return (I + s * K + (f32(1) - c) * np.matmul(K, K)).astype(f32)

    # ---- lbs.py:321-378 ---------------------------------------------------------------
    def batch_rigid_transform(self, R, J):
        B = R.shape[0]
        Jc = J[..., None].astype(f32)
        rel = Jc.copy()
        rel[:, 1:] -= Jc[:, self.parents[1:]]
        T = np.zeros((B, 5, 4, 4), f32)
        T[:, :, :3, :3] = R
        T[:, :, :3, 3:4] = rel
        T[:, :, 3, 3] = 1
        chain = [T[:, 0]]
        for i in range(1, 5):
            chain.append(np.matmul(chain[self.parents[i]], T[:, i]).astype(f32))
        Tr = np.stack(chain, 1)
        posed = Tr[:, :, :3, 3]
        Jh = np.concatenate([Jc, np.zeros((B, 5, 1, 1), f32)], 2)
        corr = np.matmul(Tr, Jh).astype(f32)                         # [B,5,4,1]
        A = Tr.copy()
        A[:, :, :, 3:4] -= corr                                       # F.pad(.., [3,0,...]) : only last column
        return posed, A

    # ---- lbs.py:140-227 ---------------------------------------------------------------
    def lbs(self, betas, pose):
        B = betas.shape[0]
        v_shaped = self.v_template[None] + np.einsum("bl,mkl->bmk", betas, self.shapedirs).astype(f32)
        J = np.einsum("bik,ji->bjk", v_shaped, self.J_regressor).astype(f32)
        R = self.batch_rodrigues(pose.reshape(-1, 3)).reshape(B, -1, 3, 3)
        pf = (R[:, 1:] - np.eye(3, dtype=f32)).reshape(B, -1)
        v_posed = (np.matmul(pf, self.posedirs).reshape(B, -1, 3) + v_shaped).astype(f32)
        Jt, A = self.batch_rigid_transform(R, J)
        T = np.matmul(self.lbs_weights[None], A.reshape(B, 5, 16)).reshape(B, -1, 4, 4).astype(f32)
        vh = np.concatenate([v_posed, np.ones((B, v_posed.shape[1], 1), f32)], 2)
        out = np.matmul(T, vh[..., None])[:, :, :3, 0].astype(f32)
        return out, Jt

    # ---- lbs.py:101-137 ---------------------------------------------------------------
    def vertices2landmarks(self, verts, idx, bary):
        tri = self.faces[idx]                                         # [B,L,3]
        B = verts.shape[0]
        lv = verts[np.arange(B)[:, None, None], tri]                  # [B,L,3,3]
        return np.einsum("blfi,blf->bli", lv, bary.astype(f32)).astype(f32)

    # ---- FLAME.py:117-159 (uses the FLAME.py copy: no minus sign, lbs.py:26-32 euler) ---
    def dynamic_lmk(self, full_pose):
        B = full_pose.shape[0]
        aa = full_pose.reshape(B, -1, 3)[:, self.neck_kin_chain]
        R = self.batch_rodrigues(aa.reshape(-1, 3)).reshape(B, -1, 3, 3)
        rel = np.broadcast_to(np.eye(3, dtype=f32), (B, 3, 3)).copy()
        for i in range(len(self.neck_kin_chain)):
            rel = np.matmul(R[:, i], rel).astype(f32)
        sy = np.sqrt(rel[:, 0, 0] * rel[:, 0, 0] + rel[:, 1, 0] * rel[:, 1, 0]).astype(f32)
        ang = np.arctan2(-rel[:, 2, 0], sy).astype(f32)
        deg = ((ang * f32(180.0)).astype(f32) / f32(np.pi)).astype(f32)
        y = np.round(np.minimum(deg, f32(39))).astype(np.int64)       # np.round == half-to-even, as torch.round
        neg = (y < 0).astype(np.int64)
        mask = (y < -39).astype(np.int64)
        negv = mask * 78 + (1 - mask) * (39 - y)
        y = neg * negv + (1 - neg) * y
        return self.dynamic_lmk_faces_idx[y], self.dynamic_lmk_bary_coords[y], y

    # ---- FLAME.py:232-315 -------------------------------------------------------------
    def forward(self, p, zero_expression=False, zero_shape=False, zero_pose=False):
        shape = np.asarray(p["shape_params"], f32)
        exp = np.asarray(p["expression_params"], f32)
        pose = p.get("pose_params"); jaw = p.get("jaw_params")
        eye = p.get("eye_pose_params"); neck = p.get("neck_pose_params"); eyelid = p.get("eyelid_params")
        B = shape.shape[0]
        if exp.shape[1] < self.n_exp:
            exp = np.concatenate([exp, np.zeros((B, self.n_exp - exp.shape[1]), f32)], 1)
        if shape.shape[1] < self.n_shape:
            shape = np.concatenate([shape, np.zeros((B, self.n_shape - shape.shape[1]), f32)], 1)
        if zero_expression:
            exp = np.zeros_like(exp); jaw = np.zeros_like(jaw)
        if zero_shape:
            shape = np.zeros_like(shape)
        if zero_pose:
            pose = np.zeros_like(pose); pose[..., 0] = 0.2; pose[..., 1] = -0.7
        if eye is None:
            eye = np.zeros((B, 6), f32)
        if neck is None:
            neck = np.zeros((B, 3), f32)
        betas = np.concatenate([shape, exp], 1).astype(f32)
        full_pose = np.concatenate([pose, neck, jaw, eye], 1).astype(f32)
        verts, _ = self.lbs(betas, full_pose)
        if eyelid is not None:
            eyelid = np.asarray(eyelid, f32)
            verts = verts + self.r_eyelid * eyelid[:, 1:2, None]
            verts = verts + self.l_eyelid * eyelid[:, 0:1, None]
            verts = verts.astype(f32)
        dfi, dbc, lut = self.dynamic_lmk(full_pose)
        fi = np.concatenate([dfi, np.broadcast_to(self.lmk_faces_idx[None], (B, 51))], 1)
        bc = np.concatenate([dbc, np.broadcast_to(self.lmk_bary_coords[None], (B, 51, 3))], 1)
        l2d = self.vertices2landmarks(verts, fi, bc)
        l3d = self.vertices2landmarks(verts, np.repeat(self.full_lmk_faces_idx, B, 0),
                                      np.repeat(self.full_lmk_bary_coords, B, 0))
        lmp = self.vertices2landmarks(verts, np.broadcast_to(self.mp_lmk_faces_idx[None], (B, 105)),
                                      np.broadcast_to(self.mp_lmk_bary_coords[None], (B, 105, 3)))
        return dict(vertices=verts, landmarks_fan=l2d, landmarks_fan_3d=l3d, landmarks_mp=lmp, _lut_idx=lut)
