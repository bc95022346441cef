"""GPU parity: smirk_amd.Renderer (HIP) vs the oracle (oracle/render_ref.py + raster_ref.c) and the committed reference outputs.
Raster indices must be BIT-EXACT given identical vertex bits; pixels within the stated fp32 tolerance."""
import os

import numpy as np
import pytest
import torch

from oracle import assets as A
from oracle.flame_ref import FlameRef
from oracle.render_ref import RendererRef, rasterize_naive

pytestmark = pytest.mark.gpu

PIX_TOL = 2e-6       # rendered pixels are <= 0.794; fp32 shading arithmetic, summation order may differ by an ulp or two


@pytest.fixture(scope="module")
def rend(sandbox):
    from smirk_amd import Renderer
    cwd = os.getcwd(); os.chdir(sandbox)
    try:
        r = Renderer().cuda()
    finally:
        os.chdir(cwd)
    return r


def _gpu(rend, verts, cam, **lm):
    out = rend.forward(torch.from_numpy(verts).cuda(), torch.from_numpy(cam).cuda(), _aux=True,
                       **{k: torch.from_numpy(v).cuda() for k, v in lm.items()})
    torch.cuda.synchronize()
    aux = {k: v.cpu().numpy() for k, v in out.pop("_aux").items()}
    return {k: v.cpu().numpy() for k, v in out.items()}, aux


def test_render_matches_reference_golden(rend, golden_dir):
    g = np.load(os.path.join(golden_dir, "render_golden.npz"))
    f = np.load(os.path.join(golden_dir, "flame_golden.npz"))
    out, _ = _gpu(rend, f["vertices"][:2], g["cam"], landmarks_fan=f["landmarks_fan"][:2], landmarks_mp=f["landmarks_mp"][:2])
    assert np.array_equal(out["transformed_vertices"], g["transformed_vertices"])
    assert np.array_equal(out["landmarks_fan"], g["landmarks_fan"]) and np.array_equal(out["landmarks_mp"], g["landmarks_mp"])
    img = out["rendered_img"]
    assert np.array_equal(img[:, 0], img[:, 1]) and np.array_equal(img[:, 0], img[:, 2])
    assert np.array_equal(img[:, 0] == 0, g["rendered_ch0"] == 0)          # background exactly 0.0, same coverage
    assert np.abs(img[:, 0] - g["rendered_ch0"]).max() < PIX_TOL


@pytest.mark.parametrize("B,seed", [(1, 0), (3, 5), (8, 9)])
def test_render_matches_oracle_bit_exact_indices(rend, sandbox, B, seed):
    p = A.synth_flame_params(B, seed=seed)
    p["shape_params"] *= 0.4
    verts = FlameRef(sandbox).forward(p)["vertices"]
    cam = A.synth_cam(B, seed=seed)
    ref = RendererRef(sandbox).forward(verts, cam)
    out, aux = _gpu(rend, verts, cam)
    Ff = 3408
    p2f = ref["_aux"]["pix_to_face"].astype(np.int64)
    packed = np.where(p2f >= 0, p2f + (np.arange(B, dtype=np.int64) * Ff)[:, None, None], -1)
    assert np.array_equal(out["transformed_vertices"], ref["transformed_vertices"])
    assert np.array_equal(aux["pix_to_face"], packed)                        # bit-exact triangle indices
    assert np.array_equal(aux["bary"], ref["_aux"]["bary"])                  # same fp32 op sequence, no FMA contraction
    assert np.array_equal(aux["zbuf"], ref["_aux"]["zbuf"])
    assert np.abs(aux["normals"] - ref["_aux"]["normals"]).max() < 1e-6
    assert np.abs(out["rendered_img"] - ref["rendered_img"]).max() < PIX_TOL
    cov = (packed >= 0).mean()
    assert 0.1 < cov < 0.9


@pytest.mark.parametrize("kind", ["cloud", "flat", "layers", "slivers"])
def test_render_depth_culling_is_bit_exact_on_adversarial_vertex_sets(rend, sandbox, kind):
    """round 5: the rasteriser sorts faces front to back and culls by a per-face lower depth bound (csrc/render.hip).  The FLAME-shaped meshes of the
    other tests are kind to that; these vertex sets are not: a random point cloud (every triangle spans the image, hundreds of layers, wild depth ranges), an exactly
    FLAT mesh (every face at the same depth: every comparison is a tie on pz, the lower face index must win), two coincident layers a hair apart, and near-degenerate
    slivers (the bound must give up: rho large).  pix_to_face / barycentrics / z-buffer must equal the naive oracle bit for bit."""
    rng = np.random.default_rng({"cloud": 1, "flat": 2, "layers": 3, "slivers": 4}[kind])
    B, V = 3, 5023
    base = FlameRef(sandbox).forward(A.synth_flame_params(B, seed=17))["vertices"]
    if kind == "cloud":
        verts = rng.uniform(-0.12, 0.12, (B, V, 3)).astype(np.float32)
    elif kind == "flat":
        verts = base.copy(); verts[..., 2] = 0.0                                 # orthographic projection: xy untouched, all depths equal
    elif kind == "layers":
        verts = base.copy(); verts[:, ::2, 2] = 0.01; verts[:, 1::2, 2] = 0.01 + 1e-6
    else:
        verts = base.copy(); verts[..., 1] = verts[..., 0] * 0.5 + rng.normal(0, 2e-6, (B, V)).astype(np.float32)   # all vertices almost on one line
    cam = np.array([[8, 0.0, 0.0], [6, 0.02, -0.03], [10, -0.05, 0.04]], np.float32)
    ref = RendererRef(sandbox).forward(verts, cam)
    out, aux = _gpu(rend, verts, cam)
    p2f = ref["_aux"]["pix_to_face"].astype(np.int64)
    packed = np.where(p2f >= 0, p2f + (np.arange(B, dtype=np.int64) * 3408)[:, None, None], -1)
    assert np.array_equal(aux["pix_to_face"], packed), kind
    assert np.array_equal(aux["bary"], ref["_aux"]["bary"]) and np.array_equal(aux["zbuf"], ref["_aux"]["zbuf"]), kind
    if kind != "slivers":
        assert (packed >= 0).mean() > 0.05, kind                                # the case really draws something


def test_render_full_head_mesh_takes_the_unsorted_fallback_and_matches_the_oracle(sandbox):
    """advisor r05: a mesh beyond SORT_N = 4096 faces skips the in-LDS depth sort (raster_face_setup, mesh order, zlow = 0, four per-wave bin lists of Ff / 4 entries).
    Renderer(render_full_head=True) is exactly that mesh in the reference (renderer.py:50-74: all 9976 faces of head_template.obj) — every shipped test rendered
    the 3408-face sub-mesh, so the branch never ran against the oracle.  pix_to_face / barycentrics / z-buffer bit-exact, pixels within PIX_TOL."""
    from smirk_amd import Renderer

    class FullHeadRef(RendererRef):
        def __init__(self, root):
            super().__init__(root)
            _, _, faces, _ = A.parse_obj(os.path.join(root, "assets", "head_template.obj"))
            self.final_mask, self.faces = np.arange(int(faces.max()) + 1), faces

    cwd = os.getcwd(); os.chdir(sandbox)
    try:
        full = Renderer(render_full_head=True).cuda()
    finally:
        os.chdir(cwd)
    B = 2
    p = A.synth_flame_params(B, seed=23)
    p["shape_params"] *= 0.4
    verts, cam = FlameRef(sandbox).forward(p)["vertices"], A.synth_cam(B, seed=23)
    ref = FullHeadRef(sandbox).forward(verts, cam)
    out, aux = _gpu(full, verts, cam)
    Ff = 9976
    assert full.faces.shape[1] == Ff > 4096
    p2f = ref["_aux"]["pix_to_face"].astype(np.int64)
    packed = np.where(p2f >= 0, p2f + (np.arange(B, dtype=np.int64) * Ff)[:, None, None], -1)
    assert np.array_equal(aux["pix_to_face"], packed)
    assert np.array_equal(aux["bary"], ref["_aux"]["bary"]) and np.array_equal(aux["zbuf"], ref["_aux"]["zbuf"])
    assert np.abs(out["rendered_img"] - ref["rendered_img"]).max() < PIX_TOL
    assert np.array_equal(out["transformed_vertices"][..., :2], ref["transformed_vertices"][..., :2])
    assert np.array_equal(out["transformed_vertices"][..., 2], ref["transformed_vertices"][..., 2] + np.float32(10))     # renderer.py:141 quirk (in-place z += 10)
    assert (packed >= 0).mean() > 0.1


def test_render_edge_cases(rend, sandbox):
    """mesh partly / wholly off-screen, tiny scale (sub-pixel triangles), huge scale (few big triangles)."""
    fr = FlameRef(sandbox)
    p = A.synth_flame_params(4, seed=2)
    verts = fr.forward(p)["vertices"]
    cam = np.array([[8, 0.12, -0.1], [8, 3.0, 0.0], [0.5, 0, 0], [40, 0, 0.02]], np.float32)
    ref = RendererRef(sandbox).forward(verts, cam)
    out, aux = _gpu(rend, verts, cam)
    p2f = ref["_aux"]["pix_to_face"].astype(np.int64)
    packed = np.where(p2f >= 0, p2f + (np.arange(4, dtype=np.int64) * 3408)[:, None, None], -1)
    assert np.array_equal(aux["pix_to_face"], packed)
    assert (packed[1] == -1).all() and out["rendered_img"][1].max() == 0.0      # fully off-screen => exact zeros
    assert np.abs(out["rendered_img"] - ref["rendered_img"]).max() < PIX_TOL


def test_render_faces_straddling_the_camera_plane(rend, sandbox):
    """z rule (oracle/raster_ref.c header; pytorch3d 0.7.x): a face with ANY vertex at z < kEpsilon is dropped whole.  The mesh is pushed along z until the
    plane z = 0 (after renderer.py:144's +10) cuts through it: HIP and oracle must drop exactly the same faces — bit-exact indices, barycentrics, depth."""
    fr = FlameRef(sandbox)
    p = A.synth_flame_params(3, seed=11)
    p["shape_params"] *= 0.4
    verts = fr.forward(p)["vertices"]
    cam = A.synth_cam(3, seed=11)
    rr = RendererRef(sandbox)
    base = rr.forward(verts, cam)
    cov0 = (base["_aux"]["pix_to_face"] >= 0).mean()
    hit = False
    for sign in (1.0, -1.0):
        v = verts.copy()
        v[..., 2] += sign * 10.0 / cam[:, None, 0]                                  # z' = +-s z + 10 crosses 0 inside the head for one of the two signs
        ref = rr.forward(v, cam)
        out, aux = _gpu(rend, v, cam)
        p2f = ref["_aux"]["pix_to_face"].astype(np.int64)
        packed = np.where(p2f >= 0, p2f + (np.arange(3, dtype=np.int64) * 3408)[:, None, None], -1)
        assert np.array_equal(aux["pix_to_face"], packed)
        assert np.array_equal(aux["bary"], ref["_aux"]["bary"]) and np.array_equal(aux["zbuf"], ref["_aux"]["zbuf"])
        assert np.abs(out["rendered_img"] - ref["rendered_img"]).max() < PIX_TOL
        cov = (packed >= 0).mean()
        hit = hit or (0.0 < cov < 0.9 * cov0)                                       # part of the mesh dropped, part still drawn
    assert hit


def test_render_batch_permutation(rend, sandbox):
    p = A.synth_flame_params(5, seed=4)
    verts = FlameRef(sandbox).forward(p)["vertices"]
    cam = A.synth_cam(5, seed=4)
    perm = np.array([3, 0, 4, 1, 2])
    a, _ = _gpu(rend, verts, cam)
    b, _ = _gpu(rend, verts[perm], cam[perm])
    assert np.array_equal(a["rendered_img"][perm], b["rendered_img"])


# ---- backward pass: smirk_render_backward vs autograd through the reference / the torch oracle --------------------
GRAD_RTOL = 1e-5     # relative to max(1, largest |gradient|) of that tensor; measured 3e-7..1.5e-6.  Gradients of the barycentric weights scale with 1/area of
                     # sub-pixel triangles (|g| ~ 1e3-1e4 for unit pixel weights), so fp32 evaluation order shows at the 1e-6..1e-5 level.


def _hip_render_grads(rend, v, c, lms, loss_seed, keys=None):
    from oracle.render_torch_ref import scalar_loss
    leaf = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda().requires_grad_(True)
    tv, tc = leaf(v), leaf(c)
    tl = {k: leaf(a) for k, a in lms.items()}
    out = rend.forward(tv, tc, **tl)
    kw = {} if keys is None else {"keys": keys}
    _, ws = scalar_loss({k: o.detach().cpu() for k, o in out.items()}, seed=loss_seed, **kw)
    sum((out[k] * w.cuda()).sum() for k, w in ws.items()).backward()
    torch.cuda.synchronize()
    grads = {"vertices": tv.grad, "cam": tc.grad, **{k: t.grad for k, t in tl.items()}}
    return {k: g.cpu().numpy() for k, g in grads.items() if g is not None}, {k: o.detach().cpu().numpy() for k, o in out.items()}


def _oracle_render_grads(sandbox, v, c, lms, loss_seed, keys=None, dtype=torch.float32):
    from oracle.render_torch_ref import RendererTorchRef, scalar_loss
    leaf = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype).requires_grad_(True)
    tv, tc = leaf(v), leaf(c)
    tl = {k: leaf(a) for k, a in lms.items()}
    out = RendererTorchRef(sandbox, dtype=dtype).forward(tv, tc, **tl)
    kw = {} if keys is None else {"keys": keys}
    loss, _ = scalar_loss(out, seed=loss_seed, **kw)
    loss.backward()
    grads = {"vertices": tv.grad, "cam": tc.grad, **{k: t.grad for k, t in tl.items()}}
    return {k: g.double().numpy() for k, g in grads.items() if g is not None}


def _cmp_render_grads(got, ref):
    worst = {}
    for k in ref:
        scale = max(1.0, np.abs(ref[k]).max())
        worst[k] = np.abs(got[k] - ref[k]).max() / scale
        assert worst[k] < GRAD_RTOL, (k, worst[k])
    return worst


def test_render_backward_matches_reference_autograd_golden(rend, golden_dir):
    g = np.load(os.path.join(golden_dir, "render_grad_golden.npz"))
    lms = {"landmarks_fan": g["in_landmarks_fan"], "landmarks_mp": g["in_landmarks_mp"]}
    got, _ = _hip_render_grads(rend, g["in_vertices"], g["in_cam"], lms, int(g["loss_seed"]))
    _cmp_render_grads(got, {k: g["d_" + k] for k in ("vertices", "cam", "landmarks_fan", "landmarks_mp")})


@pytest.mark.parametrize("B,seed", [(1, 2), (4, 6)])
def test_render_backward_matches_oracle_autograd(rend, sandbox, B, seed):
    fo = FlameRef(sandbox).forward(A.synth_flame_params(B, seed=seed))
    cam = A.synth_cam(B, seed=seed)
    lms = {"landmarks_fan": fo["landmarks_fan"], "landmarks_mp": fo["landmarks_mp"]}
    got, _ = _hip_render_grads(rend, fo["vertices"], cam, lms, 3)
    _cmp_render_grads(got, _oracle_render_grads(sandbox, fo["vertices"], cam, lms, 3))
    # image loss only (the photometric path of smirk_trainer.py:362): vertices outside the face region get exactly zero gradient
    got, _ = _hip_render_grads(rend, fo["vertices"], cam, {}, 4, keys=("rendered_img",))
    ref = _oracle_render_grads(sandbox, fo["vertices"], cam, {}, 4, keys=("rendered_img",))
    _cmp_render_grads(got, ref)
    assert np.array_equal(got["vertices"] == 0, ref["vertices"] == 0) or np.count_nonzero(got["vertices"][ref["vertices"] == 0]) == 0


def test_render_backward_is_deterministic_and_forward_unchanged(rend, sandbox):
    fo = FlameRef(sandbox).forward(A.synth_flame_params(3, seed=8))
    cam = A.synth_cam(3, seed=8)
    a, out = _hip_render_grads(rend, fo["vertices"], cam, {}, 1)
    b, _ = _hip_render_grads(rend, fo["vertices"], cam, {}, 1)
    for k in a:
        assert np.array_equal(a[k], b[k]), k                      # gather-based backward: bit-reproducible
    base, _ = _gpu(rend, fo["vertices"], cam)
    assert np.array_equal(out["rendered_img"], base["rendered_img"])
    assert np.array_equal(out["transformed_vertices"], base["transformed_vertices"])


def test_flame_to_render_chain_backward(rend, sandbox):
    """Parameter gradients of an image + landmark loss through FLAME -> Renderer with both autograd bridges chained, vs the same loss
    split at the vertices: HIP renderer gradients (validated above) pushed through the torch FLAME oracle.  (Comparing against a fully
    independent oracle chain is ill-posed: its vertices differ in the last bit, a few boundary pixels change owner, and each pixel's
    barycentric gradient is ~1e3 — measured 5e-4 relative noise.)"""
    from smirk_amd import FLAME
    from oracle.flame_torch_ref import FlameTorchRef
    from oracle.render_torch_ref import scalar_loss
    cwd = os.getcwd(); os.chdir(sandbox)
    try:
        fl = FLAME().cuda()
    finally:
        os.chdir(cwd)
    B = 2
    p = A.synth_flame_params(B, seed=31)
    cam = A.synth_cam(B, seed=31)
    keys = ("rendered_img", "landmarks_fan", "landmarks_mp")
    lkeys = ("vertices", "landmarks_fan", "landmarks_mp")
    # (1) chained HIP autograd
    gp = {k: torch.from_numpy(v).cuda().requires_grad_(True) for k, v in p.items()}
    gc = torch.from_numpy(cam).cuda().requires_grad_(True)
    fo = fl.forward(gp)
    ro = rend.forward(fo["vertices"], gc, landmarks_fan=fo["landmarks_fan"], landmarks_mp=fo["landmarks_mp"])
    _, ws = scalar_loss({k: ro[k].detach().cpu() for k in keys}, seed=2, keys=keys)
    sum((ro[k] * ws[k].cuda()).sum() for k in keys).backward()
    # (2) split: renderer gradients at the same vertices, then the FLAME oracle's autograd
    leaves = {k: fo[k].detach().clone().requires_grad_(True) for k in lkeys}
    gc2 = torch.from_numpy(cam).cuda().requires_grad_(True)
    ro2 = rend.forward(leaves["vertices"], gc2, landmarks_fan=leaves["landmarks_fan"], landmarks_mp=leaves["landmarks_mp"])
    sum((ro2[k] * ws[k].cuda()).sum() for k in keys).backward()
    torch.cuda.synchronize()
    assert torch.equal(gc.grad, gc2.grad)
    tp = {k: torch.from_numpy(v).requires_grad_(True) for k, v in p.items()}
    ofo = FlameTorchRef(sandbox)(tp)
    torch.autograd.backward([ofo[k] for k in lkeys], [leaves[k].grad.cpu() for k in lkeys])
    for k in p:
        ref = tp[k].grad.numpy()
        err = np.abs(gp[k].grad.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
        assert err < 2e-5, (k, err)
