"""MI355X drop-in for the reference FLAME layer (src/FLAME/FLAME.py:45-315, src/FLAME/lbs.py).

Same constructor signature, asset files, buffer names and forward() contract as the reference class; the arithmetic runs
in libsmirk_hip.so (smirk_flame_forward: prologue -> fp32-MFMA blendshape GEMM with fused skinning -> landmark gather).
Inference only this round: outputs carry no autograd graph (FLAME backward is SURVEY.md §8 f-2).
"""
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L

_KCHUNK = 32


def _to_np(a, dtype=np.float32):
    if "scipy.sparse" in str(type(a)):
        a = a.todense()
    return np.array(a, dtype=dtype)


class FLAME(nn.Module):
    """Given FLAME parameters produce mesh vertices and 2D/3D landmarks (reference: FLAME.py:45-49)."""

    def __init__(self, flame_model_path='assets/FLAME2020/generic_model.pkl',
                 flame_lmk_embedding_path='assets/landmark_embedding.npy', n_shape=300, n_exp=50):
        super().__init__()
        with open(flame_model_path, 'rb') as f:
            model = pickle.load(f, encoding='latin1')
        get = (lambda k: model[k]) if isinstance(model, dict) else (lambda k: getattr(model, k))
        self.n_shape, self.n_exp, self.dtype = n_shape, n_exp, torch.float32

        faces = _to_np(get('f'), np.int64)
        v_template = _to_np(get('v_template'))
        sd = _to_np(get('shapedirs'))
        shapedirs = np.concatenate([sd[:, :, :n_shape], sd[:, :, 300:300 + n_exp]], 2)       # FLAME.py:67-69
        pd_raw = np.asarray(get('posedirs'))
        posedirs = _to_np(np.reshape(pd_raw, [-1, pd_raw.shape[-1]]).T)                      # FLAME.py:71-73
        J_regressor = _to_np(get('J_regressor'))
        parents = _to_np(get('kintree_table')[0]).astype(np.int64)
        parents[0] = -1
        if list(parents) != [-1, 0, 1, 1, 1] or posedirs.shape[0] != 36:
            raise L.SmirkHipError("the gfx950 FLAME kernel is specialised for the 5-joint FLAME kinematic tree")
        lbs_weights = _to_np(get('weights'))

        # ---- buffers with the reference's names (state_dict compatible, FLAME.py:58-113) ----
        self.register_buffer('faces_tensor', torch.from_numpy(faces))
        self.register_buffer('v_template', torch.from_numpy(v_template))
        self.register_buffer('shapedirs', torch.from_numpy(shapedirs))
        self.register_buffer('posedirs', torch.from_numpy(posedirs))
        self.register_buffer('J_regressor', torch.from_numpy(J_regressor))
        self.register_buffer('parents', torch.from_numpy(parents))
        self.register_buffer('lbs_weights', torch.from_numpy(lbs_weights))
        l_eye = np.load('assets/l_eyelid.npy').astype(np.float32)
        r_eye = np.load('assets/r_eyelid.npy').astype(np.float32)
        self.register_buffer('l_eyelid', torch.from_numpy(l_eye)[None])
        self.register_buffer('r_eyelid', torch.from_numpy(r_eye)[None])
        self.register_parameter('eye_pose', nn.Parameter(torch.zeros(1, 6), requires_grad=False))
        self.register_parameter('neck_pose', nn.Parameter(torch.zeros(1, 3), requires_grad=False))

        emb = np.load(flame_lmk_embedding_path, allow_pickle=True, encoding='latin1')[()]
        t = lambda x: x if torch.is_tensor(x) else torch.from_numpy(np.asarray(x))
        self.register_buffer('lmk_faces_idx', t(emb['static_lmk_faces_idx']).long())
        self.register_buffer('lmk_bary_coords', t(emb['static_lmk_bary_coords']).float())
        self.register_buffer('dynamic_lmk_faces_idx', t(emb['dynamic_lmk_faces_idx']).long())
        self.register_buffer('dynamic_lmk_bary_coords', t(emb['dynamic_lmk_bary_coords']).float())
        self.register_buffer('full_lmk_faces_idx', t(emb['full_lmk_faces_idx']).long())
        self.register_buffer('full_lmk_bary_coords', t(emb['full_lmk_bary_coords']).float())
        self.register_buffer('neck_kin_chain', torch.tensor([1, 0], dtype=torch.long))       # FLAME.py:104-109
        mp = np.load("assets/mediapipe_landmark_embedding/mediapipe_landmark_embedding.npz")
        self.register_buffer('mp_lmk_faces_idx', torch.from_numpy(mp['lmk_face_idx'].astype('int32')).long())
        self.register_buffer('mp_lmk_bary_coords', torch.from_numpy(mp['lmk_b_coords']).float())

        # ---- kernel-side packed constants (non-persistent: not part of the state_dict) ----
        V = v_template.shape[0]
        nb = n_shape + n_exp
        VP = (V + 31) // 32 * 32
        KP = (nb + 36 + _KCHUNK - 1) // _KCHUNK * _KCHUNK
        dirs = np.zeros((3, VP, KP), np.float32)
        dirs[:, :V, :nb] = shapedirs.transpose(1, 0, 2)
        dirs[:, :V, nb:nb + 36] = posedirs.reshape(36, V, 3).transpose(2, 1, 0)
        jd = np.einsum('jv,vck->jck', J_regressor.astype(np.float64), shapedirs.astype(np.float64))
        jt = J_regressor.astype(np.float64) @ v_template.astype(np.float64)
        nbuf = lambda name, arr: self.register_buffer(name, torch.from_numpy(np.ascontiguousarray(arr)), persistent=False)
        nbuf('_k_dirs', dirs)
        nbuf('_k_jdirs', jd.reshape(15, nb).astype(np.float32))
        nbuf('_k_jtemplate', jt.reshape(15).astype(np.float32))
        nbuf('_k_faces', faces.astype(np.int32))
        nbuf('_k_l_eyelid', l_eye)
        nbuf('_k_r_eyelid', r_eye)
        nbuf('_k_static_faces', self.lmk_faces_idx.numpy().astype(np.int32))
        nbuf('_k_dyn_faces', self.dynamic_lmk_faces_idx.numpy().astype(np.int32))
        nbuf('_k_full_faces', self.full_lmk_faces_idx.numpy().reshape(-1).astype(np.int32))
        nbuf('_k_mp_faces', self.mp_lmk_faces_idx.numpy().astype(np.int32))
        nbuf('_k_full_bary', self.full_lmk_bary_coords.numpy().reshape(-1, 3).astype(np.float32))
        nbuf('_k_static_bary', self.lmk_bary_coords.numpy().astype(np.float32))
        nbuf('_k_dyn_bary', self.dynamic_lmk_bary_coords.numpy().astype(np.float32))
        nbuf('_k_mp_bary', self.mp_lmk_bary_coords.numpy().astype(np.float32))
        nbuf('_k_v_template', v_template)
        nbuf('_k_lbs_weights', lbs_weights)
        self._dims = dict(V=V, VP=VP, F=faces.shape[0], n_shape=n_shape, n_exp=n_exp, KP=KP,
                          n_static=self.lmk_faces_idx.shape[0], n_dyn=self.dynamic_lmk_faces_idx.shape[1],
                          n_lut=self.dynamic_lmk_faces_idx.shape[0], n_full=self._k_full_faces.shape[0],
                          n_mp=self.mp_lmk_faces_idx.shape[0])
        self._ws = L.Workspace()
        self._model_struct = None
        self._model_key = None
        self._k_dirs_t = None

    # ------------------------------------------------------------------------------------------------------------
    def _struct(self):
        key = (self._k_dirs.data_ptr(), self._k_v_template.data_ptr())
        if self._model_struct is None or self._model_key != key:
            if not self._k_dirs.is_cuda:
                raise L.SmirkHipError("FLAME module is on the CPU: move it to the HIP device (.to('cuda')); there is no CPU path")
            m = L.SmirkFlameModel()
            for k, v in self._dims.items():
                setattr(m, k, v)
            P = L.ptr
            I = torch.int32
            m.dirs, m.v_template, m.lbs_weights = P(self._k_dirs), P(self._k_v_template), P(self._k_lbs_weights)
            m.jdirs, m.jtemplate = P(self._k_jdirs), P(self._k_jtemplate)
            m.l_eyelid, m.r_eyelid, m.faces = P(self._k_l_eyelid), P(self._k_r_eyelid), P(self._k_faces, I)
            m.static_faces, m.static_bary = P(self._k_static_faces, I), P(self._k_static_bary)
            m.dyn_faces, m.dyn_bary = P(self._k_dyn_faces, I), P(self._k_dyn_bary)
            m.full_faces, m.full_bary = P(self._k_full_faces, I), P(self._k_full_bary)
            m.mp_faces, m.mp_bary = P(self._k_mp_faces, I), P(self._k_mp_bary)
            self._model_struct, self._model_key = m, key
        return self._model_struct

    def forward(self, param_dictionary, zero_expression=False, zero_shape=False, zero_pose=False, _return_lut=False):
        p = param_dictionary
        shape, exp = L.as_f32c(p['shape_params']), L.as_f32c(p['expression_params'])
        pose, jaw = p.get('pose_params', None), p.get('jaw_params', None)
        eye, neck, eyelid = p.get('eye_pose_params', None), p.get('neck_pose_params', None), p.get('eyelid_params', None)
        B, dev = shape.shape[0], shape.device
        if shape.shape[1] > self.n_shape or exp.shape[1] > self.n_exp:
            raise L.SmirkHipError("more shape/expression coefficients than the model holds")
        if pose is None:      # the reference would raise AttributeError here (FLAME.py:265 self.pose_params never defined)
            raise AttributeError("'FLAME' object has no attribute 'pose_params'")
        if zero_expression:                                                                  # FLAME.py:251-253
            exp, jaw = torch.zeros_like(exp), torch.zeros_like(jaw)
        if zero_shape:                                                                       # FLAME.py:255-256
            shape = torch.zeros_like(shape)
        if zero_pose:                                                                        # FLAME.py:259-262
            pose = torch.zeros_like(pose)
            pose[..., 0] = 0.2
            pose[..., 1] = -0.7
        pose, jaw = L.as_f32c(pose), L.as_f32c(jaw)
        # FLAME.py:267-271: absent eye / neck poses default to the (frozen, zero-initialised) module parameters
        eye = L.as_f32c(self.eye_pose.expand(B, -1) if eye is None else eye)
        neck = L.as_f32c(self.neck_pose.expand(B, -1) if neck is None else neck)
        eyelid = None if eyelid is None else L.as_f32c(eyelid)
        needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                     for t in (shape, exp, pose, neck, jaw, eye, eyelid))
        if needs_grad:                                     # training step 1 (smirk_trainer.py:94-104): differentiable w.r.t. the parameters
            has_eyelid = eyelid is not None
            verts, fan, fan3d, mp = _FlameFunction.apply(self, has_eyelid, shape, exp, pose, neck, jaw, eye,
                                                         eyelid if has_eyelid else shape.new_zeros(B, 2))
            lut = None
        else:
            verts, fan, fan3d, mp, lut, _ = self._launch(shape, exp, pose, neck, jaw, eye, eyelid, _return_lut, False)
        out = {'vertices': verts, 'landmarks_fan': fan, 'landmarks_fan_3d': fan3d, 'landmarks_mp': mp}
        if _return_lut:
            out['_lut_idx'] = lut
        return out

    def _launch(self, shape, exp, pose, neck, jaw, eye, eyelid, want_lut, want_vposed):
        d = self._dims
        m = self._struct()
        lib = L.lib()
        B, dev = shape.shape[0], shape.device
        verts = torch.empty(B, d['V'], 3, device=dev)
        fan = torch.empty(B, d['n_dyn'] + d['n_static'], 3, device=dev)
        fan3d = torch.empty(B, d['n_full'], 3, device=dev)
        mp = torch.empty(B, d['n_mp'], 3, device=dev)
        lut = torch.empty(B, dtype=torch.int32, device=dev) if want_lut else None
        vposed = torch.empty(B, d['V'], 3, device=dev) if want_vposed else None
        nws = lib.smirk_flame_workspace_bytes(m, B)
        ws = self._ws.get(nws, dev)
        P = L.ptr
        L.check(lib.smirk_flame_forward(m, B, P(shape), shape.shape[1], P(exp), exp.shape[1], P(pose), P(neck, allow_none=True),
                                        P(jaw), P(eye, allow_none=True), P(eyelid, allow_none=True), P(verts), P(fan),
                                        P(fan3d), P(mp), P(lut, torch.int32, allow_none=True), P(vposed, allow_none=True),
                                        P(ws, torch.uint8), nws, L.stream_ptr()))
        return verts, fan, fan3d, mp, lut, vposed

    def _dirs_t(self):
        """[KP][3*VP] coefficient-major copy of the blendshape basis for the backward GEMM; built on first use (25 MB)."""
        if self._k_dirs_t is None or self._k_dirs_t.device != self._k_dirs.device:
            d = self._dims
            self._k_dirs_t = self._k_dirs.reshape(3 * d['VP'], d['KP']).t().contiguous()
        return self._k_dirs_t

    def _launch_backward(self, shape, exp, pose, neck, jaw, eye, eyelid, vposed, lut, grads):
        d = self._dims
        m = self._struct()
        lib = L.lib()
        B, dev = shape.shape[0], shape.device
        z = lambda *s: torch.empty(*s, device=dev)
        d_shape, d_exp, d_pose, d_neck, d_jaw, d_eye = z(B, shape.shape[1]), z(B, exp.shape[1]), z(B, 3), z(B, 3), z(B, 3), z(B, 6)
        d_eyelid = z(B, 2) if eyelid is not None else None
        g = [None if t is None else L.as_f32c(t) for t in grads]
        nws = lib.smirk_flame_backward_workspace_bytes(m, B)
        ws = self._ws.get(nws, dev)
        P = L.ptr
        N = lambda t: P(t, allow_none=True)
        L.check(lib.smirk_flame_backward(m, P(self._dirs_t()), B, P(shape), shape.shape[1], P(exp), exp.shape[1], P(pose), N(neck), P(jaw),
                                         N(eye), N(eyelid), P(vposed), P(lut, torch.int32), N(g[0]), N(g[1]), N(g[2]), N(g[3]),
                                         P(d_shape), P(d_exp), P(d_pose), P(d_neck), P(d_jaw), P(d_eye), N(d_eyelid),
                                         P(ws, torch.uint8), nws, L.stream_ptr()))
        return d_shape, d_exp, d_pose, d_neck, d_jaw, d_eye, d_eyelid


class _FlameFunction(torch.autograd.Function):
    """autograd bridge: forward = smirk_flame_forward (saving v_posed + the LUT row), backward = smirk_flame_backward."""

    @staticmethod
    def forward(ctx, module, has_eyelid, shape, exp, pose, neck, jaw, eye, eyelid):
        eyl = eyelid if has_eyelid else None
        verts, fan, fan3d, mp, lut, vposed = module._launch(shape, exp, pose, neck, jaw, eye, eyl, True, True)
        ctx.module, ctx.has_eyelid = module, has_eyelid
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(shape, exp, pose, neck, jaw, eye, eyelid, vposed, lut)
        return verts, fan, fan3d, mp

    @staticmethod
    def backward(ctx, g_verts, g_fan, g_fan3d, g_mp):
        shape, exp, pose, neck, jaw, eye, eyelid, vposed, lut = ctx.saved_tensors
        eyl = eyelid if ctx.has_eyelid else None
        d = ctx.module._launch_backward(shape, exp, pose, neck, jaw, eye, eyl, vposed, lut, (g_verts, g_fan, g_fan3d, g_mp))
        d_shape, d_exp, d_pose, d_neck, d_jaw, d_eye, d_eyelid = d
        return None, None, d_shape, d_exp, d_pose, d_neck, d_jaw, d_eye, d_eyelid
