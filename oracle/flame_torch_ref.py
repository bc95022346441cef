"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Differentiable torch-CPU restatement of the reference FLAME layer (same citations as oracle/flame_ref.py:
src/FLAME/FLAME.py:232-315, src/FLAME/lbs.py:101-378), used as the oracle of the BACKWARD pass: autograd through
this module gives the reference gradients w.r.t. shape / expression / pose / jaw / eyelid.  Pinned against autograd through the real
reference class by tests/golden/flame_grad_golden.npz (oracle/make_golden.py)."""
import numpy as np
import torch

from .flame_ref import FlameRef


class FlameTorchRef(torch.nn.Module):
    def __init__(self, assets_root, dtype=torch.float32):
        super().__init__()
        r = FlameRef(assets_root)
        t = lambda a: torch.from_numpy(np.asarray(a)).to(dtype) if np.asarray(a).dtype.kind == "f" else torch.from_numpy(np.asarray(a))
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "l_eyelid", "r_eyelid", "lmk_bary_coords",
                  "dynamic_lmk_bary_coords", "full_lmk_bary_coords", "mp_lmk_bary_coords"):
            self.register_buffer(k, t(getattr(r, k)))
        for k in ("faces", "parents", "lmk_faces_idx", "dynamic_lmk_faces_idx", "full_lmk_faces_idx", "mp_lmk_faces_idx", "neck_kin_chain"):
            self.register_buffer(k, torch.from_numpy(np.asarray(getattr(r, k))).long())
        self.n_shape, self.n_exp, self.dt = r.n_shape, r.n_exp, dtype

    @staticmethod
    def rodrigues(rv):
        angle = torch.norm(rv + 1e-8, dim=1, keepdim=True)
        d = rv / angle
        c, s = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
        rx, ry, rz = d[:, 0:1], d[:, 1:2], d[:, 2:3]
        z = torch.zeros_like(rx)
        K = torch.cat([z, -rz, ry, rz, z, -rx, -ry, rx, z], 1).view(-1, 3, 3)
        return torch.eye(3, dtype=rv.dtype)[None] + s * K + (1 - c) * torch.bmm(K, K)

    def lmk(self, verts, idx, bary):
        B = verts.shape[0]
        tri = self.faces[idx]
        lv = verts[torch.arange(B)[:, None, None], tri]
        return torch.einsum("blfi,blf->bli", lv, bary)

    def forward(self, p):
        shape, exp, pose, jaw = p["shape_params"], p["expression_params"], p["pose_params"], p["jaw_params"]
        B = shape.shape[0]
        neck = p.get("neck_pose_params", torch.zeros(B, 3, dtype=self.dt))
        eye = p.get("eye_pose_params", torch.zeros(B, 6, dtype=self.dt))
        eyelid = p.get("eyelid_params", None)
        if exp.shape[1] < self.n_exp:
            exp = torch.cat([exp, torch.zeros(B, self.n_exp - exp.shape[1], dtype=self.dt)], 1)
        if shape.shape[1] < self.n_shape:
            shape = torch.cat([shape, torch.zeros(B, self.n_shape - shape.shape[1], dtype=self.dt)], 1)
        betas = torch.cat([shape, exp], 1)
        full_pose = torch.cat([pose, neck, jaw, eye], 1)
        v_shaped = self.v_template[None] + torch.einsum("bl,mkl->bmk", betas, self.shapedirs)
        J = torch.einsum("bik,ji->bjk", v_shaped, self.J_regressor)
        R = self.rodrigues(full_pose.view(-1, 3)).view(B, -1, 3, 3)
        pf = (R[:, 1:] - torch.eye(3, dtype=self.dt)).view(B, -1)
        v_posed = torch.matmul(pf, self.posedirs).view(B, -1, 3) + v_shaped
        Jc = J[..., None]
        rel = Jc.clone()
        rel[:, 1:] = rel[:, 1:] - Jc[:, self.parents[1:]]
        T = torch.cat([torch.cat([R, rel], -1), torch.tensor([0, 0, 0, 1.0], dtype=self.dt).expand(B, 5, 1, 4)], -2)
        chain = [T[:, 0]]
        for i in range(1, 5):
            chain.append(torch.matmul(chain[int(self.parents[i])], T[:, i]))
        Tr = torch.stack(chain, 1)
        Jh = torch.cat([Jc, torch.zeros(B, 5, 1, 1, dtype=self.dt)], 2)
        A = Tr - torch.nn.functional.pad(torch.matmul(Tr, Jh), [3, 0])
        Tv = torch.matmul(self.lbs_weights[None].expand(B, -1, -1), A.view(B, 5, 16)).view(B, -1, 4, 4)
        vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=self.dt)], 2)
        verts = torch.matmul(Tv, vh[..., None])[:, :, :3, 0]
        if eyelid is not None:
            verts = verts + self.r_eyelid * eyelid[:, 1:2, None]
            verts = verts + self.l_eyelid * eyelid[:, 0:1, None]
        # dynamic contour LUT (piecewise constant: no gradient, FLAME.py:137-158)
        with torch.no_grad():
            aa = full_pose.view(B, -1, 3)[:, self.neck_kin_chain]
            Rn = self.rodrigues(aa.reshape(-1, 3)).view(B, -1, 3, 3)
            relr = torch.eye(3, dtype=self.dt).expand(B, 3, 3)
            for i in range(len(self.neck_kin_chain)):
                relr = torch.bmm(Rn[:, i], relr)
            sy = torch.sqrt(relr[:, 0, 0] ** 2 + relr[:, 1, 0] ** 2)
            y = torch.round(torch.clamp(torch.atan2(-relr[:, 2, 0], sy) * 180.0 / np.pi, max=39)).long()
            neg, mask = y.lt(0).long(), y.lt(-39).long()
            y = neg * (mask * 78 + (1 - mask) * (39 - y)) + (1 - neg) * y
        fi = torch.cat([self.dynamic_lmk_faces_idx[y], self.lmk_faces_idx[None].expand(B, -1)], 1)
        bc = torch.cat([self.dynamic_lmk_bary_coords[y], self.lmk_bary_coords[None].expand(B, -1, -1)], 1)
        return dict(vertices=verts, landmarks_fan=self.lmk(verts, fi, bc),
                    landmarks_fan_3d=self.lmk(verts, self.full_lmk_faces_idx.expand(B, -1), self.full_lmk_bary_coords.expand(B, -1, -1)),
                    landmarks_mp=self.lmk(verts, self.mp_lmk_faces_idx[None].expand(B, -1), self.mp_lmk_bary_coords[None].expand(B, -1, -1)))


def scalar_loss(out, seed=0):
    """A fixed random linear functional of all four outputs (so every output's gradient path is exercised)."""
    g = torch.Generator().manual_seed(seed)
    loss = 0.0
    ws = {}
    for k in ("vertices", "landmarks_fan", "landmarks_fan_3d", "landmarks_mp"):
        w = torch.randn(out[k].shape, generator=g, dtype=torch.float32).to(out[k].dtype)
        ws[k] = w
        loss = loss + (out[k] * w).sum()
    return loss, ws
