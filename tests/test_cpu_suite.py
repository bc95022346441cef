"""CPU-side checks (`-m "not gpu"`): oracle vs golden vectors, analytic rasteriser KATs, host logic, C-ABI exports."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import assets as A
from oracle import generator_ref as G
from oracle import mobilenet_ref as M
from oracle import render_ref as R
from oracle.flame_ref import FlameRef

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- oracle vs committed reference outputs
def test_flame_oracle_vs_reference_golden(sandbox, golden_dir):
    g = np.load(os.path.join(golden_dir, "flame_golden.npz"))
    p = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    o = FlameRef(sandbox).forward(p)
    assert np.sqrt(((o["vertices"] - g["vertices"]) ** 2).sum(-1)).max() < 1e-6
    for k in ("landmarks_fan", "landmarks_fan_3d", "landmarks_mp"):
        assert np.abs(o[k] - g[k]).max() < 1e-6


def test_flame_oracle_properties(sandbox):
    fr = FlameRef(sandbox)
    z = dict(shape_params=np.zeros((1, 300), np.float32), expression_params=np.zeros((1, 50), np.float32),
             pose_params=np.zeros((1, 3), np.float32), jaw_params=np.zeros((1, 3), np.float32))
    assert np.abs(fr.forward(z)["vertices"][0] - fr.v_template).max() < 1e-6      # zero params -> template
    # pure global rotation rotates rigidly about joint 0
    p = dict(z, pose_params=np.array([[0.1, 0.3, -0.2]], np.float32))
    v = fr.forward(p)["vertices"][0]
    Rm = fr.batch_rodrigues(p["pose_params"])[0]
    J0 = fr.J_regressor[0] @ fr.v_template
    assert np.abs(v - ((fr.v_template - J0) @ Rm.T + J0)).max() < 2e-6


def test_render_oracle_vs_reference_golden(sandbox, golden_dir):
    g = np.load(os.path.join(golden_dir, "render_golden.npz"))
    f = np.load(os.path.join(golden_dir, "flame_golden.npz"))
    o = R.RendererRef(sandbox).forward(f["vertices"][:2], g["cam"], landmarks_fan=f["landmarks_fan"][:2],
                                        landmarks_mp=f["landmarks_mp"][:2])
    assert np.array_equal(o["transformed_vertices"], g["transformed_vertices"])
    assert np.array_equal(o["landmarks_fan"], g["landmarks_fan"])
    assert np.abs(o["rendered_img"][:, 0] - g["rendered_ch0"]).max() < 2e-6
    assert np.array_equal(o["rendered_img"][:, 0] == 0, g["rendered_ch0"] == 0)


def test_generator_oracle_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "generator_golden.npz"))
    sd = G.synth_state_dict()
    y = G.forward(sd, A.synth_generator_input(1, seed=int(g["seed"]))).numpy()
    assert np.abs(y[:, :, ::4, ::4] - g["y_sub4"]).max() < 1e-6
    assert len(sd) == 178


def test_encoder_oracle_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "encoder_golden.npz"))
    sd = M.synth_encoder_state_dict()
    m = M.SmirkEncoderRef(); m.load_state_dict(sd); m.eval()
    with torch.no_grad():
        o = m(A.synth_images(2, seed=int(g["seed"])))
    for k in ("pose_params", "cam", "shape_params", "expression_params", "eyelid_params", "jaw_params"):
        assert np.abs(o[k].numpy() - g[k]).max() < 1e-5, k


def test_mobilenet_restatement_matches_published_counts():
    """timm publishes 3.92 M / 2.04 M params for the full models; minus conv_head+classifier that is 1,413,208 / 428,888
     — the only external anchor available for the un-vendored backbone."""
    lg, sm = M.create_model("tf_mobilenetv3_large_minimal_100"), M.create_model("tf_mobilenetv3_small_minimal_100")
    assert sum(p.numel() for p in lg.parameters()) == 1413208
    assert sum(p.numel() for p in sm.parameters()) == 428888
    assert sum(p.numel() for p in lg.parameters()) + 960 * 1280 + 1280 + 1280 * 1000 + 1000 == 3924288
    assert [f["num_chs"] for f in lg.feature_info] == [16, 24, 40, 112, 960]
    assert [f["num_chs"] for f in sm.feature_info] == [16, 16, 24, 48, 576]
    assert len(lg.state_dict()) == 276 and len(sm.state_dict()) == 204
    with torch.no_grad():
        shapes = [tuple(t.shape[1:]) for t in lg.eval()(torch.zeros(1, 3, 224, 224))]
    assert shapes == [(16, 112, 112), (24, 56, 56), (40, 28, 28), (112, 14, 14), (960, 7, 7)]


# ---------------------------------------------------------------- rasteriser known-answer tests (pytorch3d semantics, App. B)
def _tri(*pts):
    return np.asarray(pts, np.float32).reshape(1, -1, 3, 3)


def test_raster_kat_axis_aligned_triangle_pixel_count_and_orientation():
    S = 8
    # right triangle covering the +x/+y quadrant corner region: vertices in pytorch3d NDC (+X left, +Y up)
    fv = _tri([[0.0, 0.0, 1.0], [1.0, 0.0, 1.0], [0.0, 1.0, 1.0]])
    p2f, zb, bary = R.rasterize_naive(fv, S, S)
    p2n, _, _ = R.rasterize_numpy(fv, S, S)
    assert np.array_equal(p2f, p2n)
    # pixel centres are at -1+(2i+1)/8; inside iff x>0,y>0,x+y<1 strictly -> centres (.125,.125),(.375,.125),(.625,.125),(.125,.375),(.375,.375),(.125,.625)
    assert (p2f >= 0).sum() == 6
    # +X is LEFT and +Y is UP in the output image: covered pixels sit in the top-left quadrant
    ys, xs = np.nonzero(p2f[0] >= 0)
    assert ys.max() < S // 2 and xs.max() < S // 2
    assert np.allclose(zb[p2f >= 0], 1.0) and (zb[p2f < 0] == -1).all()
    assert np.allclose(bary[p2f >= 0].sum(-1), 1.0, atol=1e-6)


def test_raster_kat_z_order_tie_and_shared_edge():
    S = 8
    big_near = [[-1, -1, 1.0], [3, -1, 1.0], [-1, 3, 1.0]]
    big_far = [[-1, -1, 2.0], [3, -1, 2.0], [-1, 3, 2.0]]
    p2f, _, _ = R.rasterize_naive(_tri(big_far, big_near), S, S)
    assert (p2f[p2f >= 0] == 1).all()                                        # nearer face wins regardless of order
    p2f, _, _ = R.rasterize_naive(_tri(big_near, big_near), S, S)
    assert (p2f[p2f >= 0] == 0).all()                                        # exact z tie -> lower face index
    # two triangles sharing the diagonal x == y: pixel centres ON the shared edge belong to neither (strict w > 0)
    a = [[-1, -1, 1.0], [1, -1, 1.0], [1, 1, 1.0]]
    b = [[-1, -1, 1.0], [1, 1, 1.0], [-1, 1, 1.0]]
    p2f, _, _ = R.rasterize_naive(_tri(a, b), S, S)
    assert (np.diag(p2f[0]) == -1).all() and (p2f[0][~np.eye(S, dtype=bool)] >= 0).all()
    # back-facing (clockwise) triangles are kept (cull_backfaces=False)
    p2f, _, _ = R.rasterize_naive(_tri(a[::-1]), S, S)
    assert (p2f >= 0).sum() > 0
    # degenerate (zero-area) and behind-camera faces are skipped
    p2f, _, _ = R.rasterize_naive(_tri([[0, 0, 1.0], [1, 1, 1.0], [.5, .5, 1.0]], [[-1, -1, -2.0], [3, -1, -2.0], [-1, 3, -2.0]]), S, S)
    assert (p2f == -1).all()


def test_raster_kat_face_straddling_z0_is_dropped_whole():
    """z rule of oracle/raster_ref.c (pytorch3d 0.7.x CheckPointOutsideBoundingBox): z_invalid = zmin < kEpsilon — a face with ANY vertex at or behind
    the camera plane is skipped for every pixel (not clipped), even where its interpolated depth is positive; a face with zmin just above eps renders."""
    S = 8
    full = lambda z0, z1, z2: [[-1, -1, z0], [3, -1, z1], [-1, 3, z2]]                  # covers the whole image
    # (1) straddles z = 0: zmin = -1 < eps <= zmax = 5  -> dropped everywhere, although pz > 0 on most pixels
    for fn in (R.rasterize_naive, R.rasterize_numpy):
        p2f, zb, bary = fn(_tri(full(-1.0, 5.0, 5.0)), S, S)
        assert (p2f == -1).all() and (zb == -1).all() and (bary == -1).all()
    # (2) zmin < eps < zmax with zmin POSITIVE but below kEpsilon (5e-9): still dropped
    for fn in (R.rasterize_naive, R.rasterize_numpy):
        assert (fn(_tri(full(5e-9, 2.0, 2.0)), S, S)[0] == -1).all()
    # (3) zmin == 2e-8 > eps: rendered on every pixel, depth = the interpolated z
    for fn in (R.rasterize_naive, R.rasterize_numpy):
        p2f, zb, _ = fn(_tri(full(2e-8, 2.0, 2.0)), S, S)
        assert (p2f == 0).all() and (zb > 0).all()
    # (4) a dropped straddling face does not occlude the valid face behind it
    p2f, zb, _ = R.rasterize_naive(_tri(full(-1.0, 5.0, 5.0), full(3.0, 3.0, 3.0)), S, S)
    assert (p2f == 1).all() and np.allclose(zb, 3.0)
    # (5) entirely behind the camera: dropped (the rule the older zmax form also implemented)
    assert (R.rasterize_naive(_tri(full(-2.0, -2.0, -2.0)), S, S)[0] == -1).all()


def test_raster_c_matches_numpy_on_random_soup_with_faces_behind_the_camera():
    rng = np.random.default_rng(5)
    fv = rng.uniform(-1.2, 1.2, (2, 40, 3, 3)).astype(np.float32)
    fv[..., 2] = rng.uniform(-1.0, 3, (2, 40, 3))                                        # about a third of the faces have a vertex behind z = 0
    a, b = R.rasterize_naive(fv, 16, 16), R.rasterize_numpy(fv, 16, 16)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    zmin = fv[..., 2].min(-1)
    drawn = np.unique(a[0][0][a[0][0] >= 0])
    assert (zmin[0][drawn] >= 1e-8).all() and (zmin < 1e-8).any()


def test_raster_c_matches_numpy_on_random_soup():
    rng = np.random.default_rng(0)
    fv = rng.uniform(-1.2, 1.2, (2, 40, 3, 3)).astype(np.float32)
    fv[..., 2] = rng.uniform(0.5, 3, (2, 40, 3))
    a, b = R.rasterize_naive(fv, 16, 16), R.rasterize_numpy(fv, 16, 16)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


# ---------------------------------------------------------------- host logic / drop-in contract
def test_generator_state_dict_keys_match_reference_contract():
    from smirk_amd import SmirkGenerator
    m = SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
    sd = G.synth_state_dict()
    assert set(m.state_dict().keys()) == set(sd.keys()) and len(sd) == 178
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    assert sum(p.numel() for p in m.parameters()) == 31367171


def test_encoder_state_dict_keys_match_timm_layout():
    from smirk_amd import SmirkEncoder
    m = SmirkEncoder()
    ref = M.SmirkEncoderRef()
    a, b = m.state_dict(), ref.state_dict()
    assert set(a.keys()) == set(b.keys())
    for k in a:
        assert tuple(a[k].shape) == tuple(b[k].shape), k
    assert {"pose_encoder", "shape_encoder", "expression_encoder"} <= set(dict(m.named_children()))
    # head inits of smirk_encoder.py:26-31,61-63
    assert float(m.pose_encoder.pose_cam_layers[0].bias[3]) == 7.0 and float(m.shape_encoder.shape_layers[0].weight.abs().sum()) == 0.0


def test_flame_and_renderer_construct_with_reference_buffers(in_sandbox):
    from smirk_amd import FLAME, Renderer
    f, r = FLAME(), Renderer()
    exp = {"faces_tensor": (9976, 3), "v_template": (5023, 3), "shapedirs": (5023, 3, 350), "posedirs": (36, 15069),
           "J_regressor": (5, 5023), "parents": (5,), "lbs_weights": (5023, 5), "l_eyelid": (1, 5023, 3),
           "lmk_faces_idx": (51,), "dynamic_lmk_faces_idx": (79, 17), "full_lmk_bary_coords": (1, 68, 3),
           "mp_lmk_faces_idx": (105,), "eye_pose": (1, 6), "neck_pose": (1, 3), "neck_kin_chain": (2,)}
    sd = f.state_dict()
    for k, s in exp.items():
        assert tuple(sd[k].shape) == s, k
    assert not any(k.startswith("_k_") for k in sd)
    assert f.faces_tensor.dtype == torch.int64
    assert tuple(r.faces.shape) == (1, 3408, 3) and r.image_size == 224 and "face" in r.flame_masks
    assert tuple(r.face_colors.shape) == (1, 3408, 3, 3)
    ref = R.RendererRef(in_sandbox)
    assert np.array_equal(r.faces[0].numpy(), ref.faces)
    # CSR of the normal gather reproduces the index_add_ order: contributions of a vertex come corner-1 pass first
    ptr, nf, nc = r._k_nrm_ptr.numpy(), r._k_nrm_face.numpy(), r._k_nrm_corner.numpy()
    assert ptr[-1] == 3 * 3408
    for v in (0, 17, 1786):
        seg = list(zip(nc[ptr[v]:ptr[v + 1]], nf[ptr[v]:ptr[v + 1]]))
        order = {1: 0, 2: 1, 0: 2}
        assert seg == sorted(seg, key=lambda t: (order[t[0]], t[1]))
        assert all(ref.faces[f_, c_] == v for c_, f_ in seg)
    with pytest.raises(Exception):
        f.forward({k: torch.from_numpy(v) for k, v in A.synth_flame_params(1).items()})     # CPU tensors: no fallback


def test_c_abi_exports_every_declared_symbol():
    from smirk_amd import _lib
    hdr = open(os.path.join(REPO, "include", "smirk_hip.h")).read()
    declared = set(re.findall(r"\b(smirk_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert os.path.exists(_lib.LIB_PATH), "run `python -m smirk_amd.build`"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        getattr(dll, name)
    dll.smirk_strerror.restype = ctypes.c_char_p
    assert dll.smirk_strerror(0) == b"ok" and dll.smirk_abi_version() == _lib.ABI_VERSION


def test_masking_oracle_vs_reference_golden(sandbox, golden_dir):
    """oracle/masking_ref.py against outputs of the reference's src/utils/masking.py functions (deterministic parts)."""
    from oracle import masking_ref as MR
    g = np.load(os.path.join(golden_dir, "masking_golden.npz"))
    r = np.load(os.path.join(golden_dir, "render_golden.npz"))
    fr = FlameRef(sandbox)
    tv = r["transformed_vertices"]
    b = A.load_bundle()
    prob = np.zeros(9976, np.float32)
    wts = {'neck': 0.0, 'right_eyeball': 0.0, 'right_ear': 0.0, 'lips': 0.5, 'nose': 0.5, 'left_ear': 0.0, 'eye_region': 1.0, 'forehead': 1.0,
           'left_eye_region': 1.0, 'right_eye_region': 1.0, 'face_clean': 1.0, 'cleaner_lips': 1.0}
    for k, v in wts.items():
        prob[b["tri_" + k]] = v
    w = MR.face_weights(tv, fr.faces, prob)
    assert np.array_equal(w > 0, g["weights"] > 0)
    assert np.abs(w - g["weights"]).max() < 1e-7
    pts, _ = MR.points_from_coords(tv, fr.faces, g["idx"].astype(np.int64), g["bary"])
    assert (pts != g["npoints"]).mean() < 1e-3 and np.abs(pts - g["npoints"]).max() <= 1
    img = A.synth_images(2, seed=int(g["img_seed"])).numpy()
    hull = (A.synth_generator_input(2, seed=int(g["img_seed"]))[:, 3:4] == 0).float().numpy()
    rimg = R.RendererRef(sandbox).forward(np.load(os.path.join(golden_dir, "flame_golden.npz"))["vertices"][:2], r["cam"])["rendered_img"]
    rmask = 1 - (rimg == 0).all(1, keepdims=True).astype(np.float32)
    pmask = np.zeros_like(rmask)
    for bi in range(2):
        pmask[bi, :, g["npoints"][bi, :, 1], g["npoints"][bi, :, 0]] = 1
    masked = MR.masking(img, hull, img * pmask, 10, rendered_mask=rmask)
    assert np.abs(masked[:, :, ::2, ::2] - g["masked_sub2"]).max() < 1e-6
    tp = MR.transfer_pixels(img, g["npoints"], g["npoints"][:, ::-1])
    assert int((tp != 0).sum()) == int(g["transfer_nonzero"]) and abs(tp.astype(np.float64).sum() - float(g["transfer_sum"])) < 1e-3


def test_flame_gradient_oracle_vs_reference_golden(sandbox, golden_dir):
    """oracle/flame_torch_ref.py autograd == autograd through the real reference FLAME."""
    import torch
    from oracle.flame_torch_ref import FlameTorchRef, scalar_loss
    g = np.load(os.path.join(golden_dir, "flame_grad_golden.npz"))
    tp = {k[3:]: torch.from_numpy(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith("in_")}
    loss, _ = scalar_loss(FlameTorchRef(sandbox)(tp), seed=int(g["loss_seed"]))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-3 * max(1.0, abs(float(g["loss"])))
    for k, v in tp.items():
        ref = g["d_" + k]
        assert np.abs(v.grad.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k


def test_render_gradient_oracle_vs_reference_golden(sandbox, golden_dir):
    """oracle/render_torch_ref.py autograd == autograd through the real reference Renderer.forward (the barycentric
    backward on both sides is autograd of pytorch3d's formula — third-party part parity-unpinned)."""
    import torch
    from oracle.render_torch_ref import RendererTorchRef, scalar_loss
    g = np.load(os.path.join(golden_dir, "render_grad_golden.npz"))
    leaf = lambda k: torch.from_numpy(g["in_" + k]).clone().requires_grad_(True)
    v, c, lf, lm = leaf("vertices"), leaf("cam"), leaf("landmarks_fan"), leaf("landmarks_mp")
    out = RendererTorchRef(sandbox).forward(v, c, landmarks_fan=lf, landmarks_mp=lm)
    loss, _ = scalar_loss(out, seed=int(g["loss_seed"]))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-3
    for k, t in (("vertices", v), ("cam", c), ("landmarks_fan", lf), ("landmarks_mp", lm)):
        ref = g["d_" + k]
        assert np.abs(t.grad.numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), k
    # the face-region sub-mesh is the only part of the mesh the image sees: gradients elsewhere come from transformed_vertices only
    assert np.count_nonzero(g["d_vertices"]) > 0


# ---- video loop pre/post-processing: oracle/video_ref.py restates cv2 / skimage (not on disk: parity unpinned) -------
def test_video_crop_transform_closed_form_equals_umeyama():
    from oracle import video_ref as V
    from smirk_amd.video import crop_transform
    rng = np.random.default_rng(3)
    for _ in range(5):
        lm = rng.uniform(50, 900, (478, 3))
        a, b = crop_transform(lm[:, :2], 1.4, 224), V.crop_transform(lm[:, :2], 1.4, 224)
        assert np.abs(a - b).max() < 1e-9 * max(1.0, np.abs(b).max())


def test_video_oracle_known_answers():
    import torch
    import torch.nn.functional as F
    from oracle import video_ref as V
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    assert np.array_equal(V.warp_u8(img, np.eye(3), (40, 56)), img)                        # identity warp
    sh = np.array([[1, 0, 3.0], [0, 1, -2.0], [0, 0, 1]])                                   # integer shift: out(r,c) = in(r-2, c+3), 0 outside
    w = V.warp_u8(img, sh, (40, 56))
    assert np.array_equal(w[2:, :53], img[:38, 3:]) and not w[:2].any() and not w[:, 53:].any()
    half = V.warp_u8(img, np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1]]), (40, 55))          # half-pixel: mean of two neighbours, truncated
    assert np.array_equal(half, ((img[:, :55].astype(np.float64) + img[:, 1:56]) / 2).astype(np.uint8))
    assert np.array_equal(V.resize_linear_u8(img, (56, 40)), img)
    up = V.resize_linear_u8(np.full((8, 8, 3), 77, np.uint8), (21, 13))
    assert up.shape == (13, 21, 3) and (up == 77).all()                                     # constant images stay constant
    m = V.hull_mask(np.array([[10.9, 10.2], [30.1, 10.7], [30.5, 25.9], [10.0, 25.0], [20, 18]]), (40, 48))
    assert m.sum() == 40 * 48 - 21 * 16 and not m[10:26, 10:31].any()                       # the closed square [10,30]x[10,25]
    tri = V.hull_mask(np.array([[0, 0], [8, 0], [0, 8]]), (10, 10))
    assert [int((tri[y] == 0).sum()) for y in range(10)] == [9, 8, 7, 6, 5, 4, 3, 2, 1, 0]
    x = torch.rand(2, 3, 24, 24)
    for hw in ((37, 53), (24, 24), (12, 100)):
        assert np.abs(V.interp_bilinear(x.numpy(), hw) - F.interpolate(x, hw, mode="bilinear").numpy()).max() < 1e-5
    u = np.arange(256, dtype=np.uint8)
    rq = V.to_u8(V.from_u8(u))
    assert (np.abs(rq.astype(int) - u) <= 1).all() and (rq == u).mean() > 0.9               # float32 /255 *255 truncation: a few values drop by 1


def test_barycentric_backward_three_derivations_agree():
    """The renderer-backward oracle differentiates pytorch3d's barycentric formula with autograd; pytorch3d itself uses a hand-written
    backward.  Its restatement (oracle/render_torch_ref.py::barycentric_backward_p3d), autograd and central finite differences must agree."""
    import torch
    from oracle.render_torch_ref import K_EPS, barycentric_backward_p3d
    rng = np.random.default_rng(4)
    for _ in range(20):
        tri = rng.uniform(-1, 1, (3, 2)); p = tri.mean(0) + rng.uniform(-0.05, 0.05, 2); g = rng.standard_normal(3)
        def bary(pts):
            v0, v1, v2 = pts[0], pts[1], pts[2]
            e = lambda a, b, c: (a[0] - b[0]) * (c[1] - b[1]) - (a[1] - b[1]) * (c[0] - b[0])
            area = e(v2, v0, v1) + K_EPS
            return torch.stack([e(p_t, v1, v2), e(p_t, v2, v0), e(p_t, v0, v1)]) / area
        p_t = torch.tensor(p, dtype=torch.float64)
        pts = torch.tensor(tri, dtype=torch.float64, requires_grad=True)
        (bary(pts) * torch.tensor(g)).sum().backward()
        auto = pts.grad.numpy()
        _, g0, g1, g2 = barycentric_backward_p3d(p, tri[0], tri[1], tri[2], g)
        hand = np.stack([g0, g1, g2])
        fd = np.zeros((3, 2))
        for i in range(3):
            for j in range(2):
                d = np.zeros((3, 2)); d[i, j] = 1e-6
                with torch.no_grad():
                    fd[i, j] = (((bary(torch.tensor(tri + d)) - bary(torch.tensor(tri - d))) * torch.tensor(g)).sum() / 2e-6).item()
        scale = max(1.0, np.abs(auto).max())
        assert np.abs(auto - hand).max() < 1e-9 * scale
        assert np.abs(auto - fd).max() < 1e-4 * scale


def test_video_warp_restatement_agrees_with_scipy_bilinear():
    """oracle/video_ref.warp_u8 restates skimage.transform.warp(order=1, mode='constant', cval=0) (skimage is not on disk).  scipy IS on
    disk: ndimage.map_coordinates(order=1, mode='grid-constant') is the same bilinear-with-zero-neighbours rule — an independent cross-check
    of the interpolation semantics (floor/ceil neighbours, zero outside), up to the final truncation to uint8."""
    from scipy import ndimage
    from oracle import video_ref as V
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for _ in range(4):
        s, a = rng.uniform(0.5, 2.0), rng.uniform(-0.6, 0.6)
        M = np.array([[s * np.cos(a), -s * np.sin(a), rng.uniform(-10, 20)], [s * np.sin(a), s * np.cos(a), rng.uniform(-10, 20)], [0, 0, 1]])
        Ho, Wo = 41, 47
        got = V.warp_u8(img, M, (Ho, Wo))
        cc, rr = np.meshgrid(np.arange(Wo, dtype=np.float64), np.arange(Ho, dtype=np.float64))
        x = M[0, 0] * cc + M[0, 1] * rr + M[0, 2]
        y = M[1, 0] * cc + M[1, 1] * rr + M[1, 2]
        ref = np.stack([ndimage.map_coordinates(img[..., c].astype(np.float64), [y, x], order=1, mode="grid-constant", cval=0.0) for c in range(3)], -1)
        d = np.abs(got.astype(np.float64) - np.floor(ref + 1e-9))
        assert (d > 0).mean() < 2e-3 and d.max() <= 1        # identical up to truncation ties at exact integers


def test_video_convex_hull_agrees_with_scipy():
    """oracle/video_ref.convex_hull (the hull create_mask fills, cv2.convexHull in the reference) vs scipy.spatial.ConvexHull (qhull)."""
    from scipy.spatial import ConvexHull
    from oracle import video_ref as V
    rng = np.random.default_rng(11)
    for _ in range(5):
        pts = rng.integers(0, 224, (478, 2))
        mine = {tuple(p) for p in V.convex_hull(pts).tolist()}
        ref = {tuple(pts[i].tolist()) for i in ConvexHull(pts.astype(np.float64)).vertices}
        assert mine == ref


def test_cycle_loss_formula_reproduces_the_reference_run(golden_dir):
    """smirk_amd/cycle.py::cycle_loss (smirk_trainer.py:304-313) on the re-encoded parameters the REAL reference classes produced == the loss they produced"""
    from oracle import make_cycle_golden as MC
    from smirk_amd.cycle import cycle_loss
    g = np.load(os.path.join(golden_dir, "cycle_golden.npz"))
    _, _, feats = MC.inputs()
    out = {k[len("out64/"):]: torch.from_numpy(g[k]).double() for k in g.files if k.startswith("out64/")}
    loss = cycle_loss(out, {k: v.double() for k, v in feats.items()}, use_eyelids=True, generator_frozen=False)
    assert abs(loss.item() - float(g["loss64"])) < 2e-6 * float(g["loss64"])
    frozen = cycle_loss(out, {k: v.double() for k, v in feats.items()}, generator_frozen=True)       # :310-311 the shape term only without the freeze
    assert frozen.item() < loss.item()


def test_train_golden_records_the_reference_under_bf16_autocast(golden_dir):
    """BASELINE config 5 trains under bf16 autocast.  The train golden carries, per tensor, the distance of the REAL reference class to its own float64 run in
    fp32 (`ref32_vs_64/*`, what the fp32-class HIP path is held to in tests/test_generator_train_gpu.py) and under torch.autocast("cpu", torch.bfloat16)
    (`refbf16_vs_64/*`, the bound for a single-MFMA 16-bit mode).  Both sets are complete and the autocast run is orders of magnitude looser."""
    g = np.load(os.path.join(golden_dir, "generator_train_golden.npz"))
    k32 = {k.split("/", 1)[1] for k in g.files if k.startswith("ref32_vs_64/")}
    k16 = {k.split("/", 1)[1] for k in g.files if k.startswith("refbf16_vs_64/")}
    assert k32 == k16 and {"y", "dx"} <= k16 and len(k16) > 90
    assert float(g["ref32_vs_64/y"]) < 2e-5 and float(g["refbf16_vs_64/y"]) > 1e-2                  # forward: 9e-6 vs 0.17 of the output range
    params = sorted(k16 - {"y", "dx"})
    med32 = float(np.median([float(g["ref32_vs_64/" + k]) for k in params]))
    med16 = float(np.median([float(g["refbf16_vs_64/" + k]) for k in params]))
    assert med32 < 1e-2 and med16 > 20 * med32, (med32, med16)                                        # gradients: 0.4 % vs 80 % (B = 3 BatchNorm + ReLU switching)
