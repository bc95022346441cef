"""GPU parity of smirk_amd.masking (src/utils/masking.py drop-in) vs the oracle / reference golden, plus a
statistical check of the HIP multinomial sampler (the reference's own draws differ between CPU and CUDA, so only the law is pinned)."""
import os

import numpy as np
import pytest
import torch

from oracle import assets as A
from oracle import masking_ref as MR
from oracle.flame_ref import FlameRef

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(sandbox, golden_dir):
    from smirk_amd import masking as M
    cwd = os.getcwd(); os.chdir(sandbox)
    try:
        prob = M.load_probabilities_per_FLAME_triangle()
    finally:
        os.chdir(cwd)
    g = np.load(os.path.join(golden_dir, "masking_golden.npz"))
    r = np.load(os.path.join(golden_dir, "render_golden.npz"))
    faces = torch.from_numpy(FlameRef(sandbox).faces).cuda()
    return M, prob, g, r, faces


def test_points_from_coords_match_reference(ctx):
    M, prob, g, r, faces = ctx
    tv = torch.from_numpy(r["transformed_vertices"]).cuda()
    coords = {"sampled_faces_indices": torch.from_numpy(g["idx"]).long().cuda(), "barycentric_coords": torch.from_numpy(g["bary"]).cuda()}
    npts, out = M.mesh_based_mask_uniform_faces(tv, faces, prob, mask_ratio=0.01, coords=coords)
    npts = npts.cpu().numpy()
    assert npts.shape == g["npoints"].shape and npts.dtype == np.int64
    assert (npts != g["npoints"]).mean() < 1e-3 and np.abs(npts - g["npoints"]).max() <= 1      # float->long truncation, 1-ulp sum order
    assert torch.equal(out["sampled_faces_indices"].cpu(), torch.from_numpy(g["idx"]).long())


def test_face_weights_and_sampler_law(ctx):
    M, prob, g, r, faces = ctx
    from smirk_amd import _lib as L
    tv = torch.from_numpy(r["transformed_vertices"]).cuda()
    B, V, F = tv.shape[0], tv.shape[1], faces.shape[0]
    mesh, bufs, _, _ = M._full_mesh(faces)
    normals, w = torch.empty(B, V, 3, device="cuda"), torch.empty(B, F, device="cuda")
    L.check(L.lib().smirk_vertex_normals(mesh, B, L.ptr(tv), L.ptr(normals), L.stream_ptr()))
    L.check(L.lib().smirk_mask_face_weights(L.ptr(tv), L.ptr(normals), L.ptr(bufs["faces"], torch.int32), L.ptr(prob.cuda()), B, V, F,
                                            L.ptr(w), L.stream_ptr()))
    w = w.cpu().numpy()
    assert np.array_equal(w > 0, g["weights"] > 0) and np.abs(w - g["weights"]).max() < 5e-7       # shoelace cancellation of O(1) products
    # sampler: many draws, chi-square-like check of empirical frequencies against the weights + barycentric validity
    torch.manual_seed(123)
    npts, out = M.mesh_based_mask_uniform_faces(tv, faces, prob, mask_ratio=2.0)       # 100352 draws per image
    idx, bary = out["sampled_faces_indices"].cpu().numpy(), out["barycentric_coords"].cpu().numpy()
    assert idx.shape == (B, 100352) and (bary >= 0).all() and np.abs(bary.sum(-1) - 1).max() < 1e-6
    for b in range(B):
        p = g["weights"][b].astype(np.float64); p /= p.sum()
        cnt = np.bincount(idx[b], minlength=F)
        assert cnt[p == 0].sum() == 0                                                   # zero-weight triangles are never drawn
        big = p > 2e-4
        z = (cnt[big] - p[big] * idx.shape[1]) / np.sqrt(p[big] * idx.shape[1])
        assert np.abs(z).max() < 6 and abs(z.mean()) < 0.3
    # reproducible from torch's seed, different across calls
    torch.manual_seed(123)
    _, out2 = M.mesh_based_mask_uniform_faces(tv, faces, prob, mask_ratio=2.0)
    assert torch.equal(out2["sampled_faces_indices"], out["sampled_faces_indices"])
    _, out3 = M.mesh_based_mask_uniform_faces(tv, faces, prob, mask_ratio=2.0)
    assert not torch.equal(out3["sampled_faces_indices"], out["sampled_faces_indices"])


def test_masking_and_transfer_match_reference(ctx, sandbox, golden_dir):
    M, prob, g, r, faces = ctx
    seed = int(g["img_seed"])
    img = A.synth_images(2, seed=seed)
    hull = (A.synth_generator_input(2, seed=seed)[:, 3:4] == 0).float()
    from oracle.render_ref import RendererRef
    rimg = RendererRef(sandbox).forward(np.load(os.path.join(golden_dir, "flame_golden.npz"))["vertices"][:2], r["cam"])["rendered_img"]
    rmask = torch.from_numpy(1 - (rimg == 0).all(1, keepdims=True).astype(np.float32))
    npts = torch.from_numpy(g["npoints"])
    pmask = torch.zeros_like(rmask)
    for bi in range(2):
        pmask[bi, :, npts[bi, :, 1], npts[bi, :, 0]] = 1
    extra = img * pmask
    out = M.masking(img.cuda(), hull.cuda(), extra.cuda(), 10, rendered_mask=rmask.cuda(), extra_noise=False, random_mask=0).cpu().numpy()
    assert np.abs(out[:, :, ::2, ::2] - g["masked_sub2"]).max() < 1e-6 and abs(out.astype(np.float64).sum() - float(g["masked_sum"])) < 1e-2
    # with explicit random fields vs the oracle
    rng = np.random.default_rng(0)
    noise = (rng.standard_normal(img.shape) * 0.05 + 1).astype(np.float32)
    field = (rng.uniform(size=(2, 1, 224, 224)) < 0.01).astype(np.float32)
    o2 = M.masking(img.cuda(), hull.cuda(), extra.cuda(), 10, rendered_mask=rmask.cuda(), _noise_mult=torch.from_numpy(noise).cuda(),
                   _random_field=torch.from_numpy(field).cuda()).cpu().numpy()
    ref = MR.masking(img.numpy(), hull.numpy(), extra.numpy(), 10, rendered_mask=rmask.numpy(), noise_mult=noise, random_field=field)
    assert np.abs(o2 - ref).max() < 1e-6
    # generated noise / dropout: statistics only
    torch.manual_seed(7)
    o3 = M.masking(img.cuda(), hull.cuda(), extra.cuda(), 10, rendered_mask=rmask.cuda()).cpu().numpy()
    sel = (extra.numpy() > 0) & (o3 == o3)
    ratio = o3[sel] / extra.numpy()[sel]
    kept = ratio > 0.5
    assert 0.2 < kept.mean() < 0.5                         # P(pixel in no 11x11 patch of a 1 % Bernoulli field) = 0.99^121 ~ 0.30 (more near borders)
    assert abs(ratio[kept].mean() - 1) < 0.01 and 0.035 < ratio[kept].std() < 0.065
    tp = M.transfer_pixels(img.cuda(), npts.cuda(), torch.flip(npts, [1]).cuda()).cpu().numpy()
    assert int((tp != 0).sum()) == int(g["transfer_nonzero"]) and abs(tp.astype(np.float64).sum() - float(g["transfer_sum"])) < 1e-3
    rb = torch.tensor([100, 300])
    tpb = M.transfer_pixels(img.cuda(), npts.cuda(), torch.flip(npts, [1]).cuda(), rbound=rb.cuda()).cpu().numpy()
    assert np.array_equal(tpb, MR.transfer_pixels(img.numpy(), g["npoints"], g["npoints"][:, ::-1], rbound=rb.numpy()))


def test_pipeline_with_hull_mask(ctx, sandbox):
    """SmirkPipeline(hull_mask=...) runs demo.py:138-165 on the GPU between render and generate."""
    M, prob, g, r, faces = ctx
    from oracle import generator_ref as G, mobilenet_ref as MB
    from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator
    from smirk_amd.pipeline import OverlappedPipeline, SmirkPipeline
    cwd = os.getcwd(); os.chdir(sandbox)
    try:
        fl, rn = FLAME().cuda(), Renderer().cuda()
    finally:
        os.chdir(cwd)
    enc = SmirkEncoder(); enc.load_state_dict(MB.synth_encoder_state_dict()); enc = enc.cuda().eval()
    gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(G.synth_state_dict()); gen = gen.cuda().eval()
    pipe = SmirkPipeline(enc, fl, rn, gen, face_probabilities=prob.cuda())
    img = A.synth_images(3, seed=9).cuda()
    hull = (A.synth_generator_input(3, seed=9)[:, 3:4] != 0).float().cuda()
    torch.manual_seed(0)
    out = pipe(img, hull_mask=hull)
    mk, rimg = out["masked_img"], out["rendered_img"]
    assert mk.shape == img.shape and torch.isfinite(out["reconstructed_img"]).all()
    eroded = 1 - torch.nn.functional.max_pool2d(1 - hull, 21, 1, 10)
    bg = (eroded == 1) & (rimg[:, :1] == 0)
    a, b = mk[bg.expand_as(mk)], img[bg.expand_as(img)]                                   # kept photo pixels: untouched, except the few sampled
    diff = a != b                                                                         # mesh points that fall outside the rendered face region
    assert diff.float().mean().item() < 0.05 and ((a[diff] / b[diff]) - 1).abs().max().item() < 0.35
    inside = (rimg[:, :1] != 0).expand_as(mk)
    frac = (mk[inside] != 0).float().mean().item()
    assert 0.0 < frac < 0.2                                                               # only the sparse sampled points survive inside the face
    torch.manual_seed(0)
    run = OverlappedPipeline(pipe)
    assert run.submit(img, hull_mask=hull) is None
    o2 = run.flush()
    torch.cuda.synchronize()
    assert torch.equal(o2["masked_img"], mk) and torch.equal(o2["reconstructed_img"], out["reconstructed_img"])
