"""GPU parity of the video-loop kernels (csrc/video.hip) and of smirk_amd.VideoPipeline (demo_video.py:107-214) against
oracle/video_ref.py (numpy restatement of the cv2 / skimage calls; third-party sources not on disk: parity unpinned) and torch-CPU."""
import os

import numpy as np
import pytest
import torch

from oracle import assets as A
from oracle import generator_ref as G
from oracle import mobilenet_ref as M
from oracle import video_ref as V

pytestmark = pytest.mark.gpu


def _lib():
    from smirk_amd import _lib as L
    return L, L.lib()


def _mats(Ts):
    return torch.from_numpy(np.stack([T[:2].reshape(6) for T in Ts])).cuda()


def _frames(n, H, W, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    base = (np.sin(xx / 17.0)[..., None] * np.array([60, 80, 40]) + np.cos(yy / 11.0)[..., None] * np.array([50, 30, 70]) + 128)
    return [np.clip(base + rng.integers(-40, 40, (H, W, 3)), 0, 255).astype(np.uint8) for _ in range(n)]


def _landmarks(n, H, W, seed, L=478):
    rng = np.random.default_rng(seed + 1)
    out = []
    for _ in range(n):
        c = np.array([W * rng.uniform(0.35, 0.65), H * rng.uniform(0.35, 0.65)])
        r = min(H, W) * rng.uniform(0.15, 0.3)
        ang, rad = rng.uniform(0, 2 * np.pi, L), np.sqrt(rng.uniform(0, 1, L)) * r
        out.append(np.concatenate([c + np.stack([np.cos(ang) * rad, np.sin(ang) * rad * 1.2], 1), rng.uniform(-1, 1, (L, 1))], 1))
    return out


def test_warp_kernel_matches_oracle_exactly():
    L, lib = _lib()
    fr = _frames(3, 90, 130, 1)
    rng = np.random.default_rng(2)
    Ts = []
    for i in range(3):
        s, a = rng.uniform(0.4, 2.5), rng.uniform(-0.5, 0.5)
        Ts.append(np.array([[s * np.cos(a), -s * np.sin(a), rng.uniform(-20, 40)], [s * np.sin(a), s * np.cos(a), rng.uniform(-20, 40)], [0, 0, 1]]))
    src = torch.from_numpy(np.stack(fr)).cuda()
    of = torch.empty(3, 3, 64, 72, device="cuda")
    ou = torch.empty(3, 64, 72, 3, dtype=torch.uint8, device="cuda")
    L.check(lib.smirk_warp_affine_u8(L.ptr(src, torch.uint8), 3, 90, 130, L.ptr(_mats(Ts), torch.float64), 64, 72, 1, L.ptr(of),
                                     L.ptr(ou, torch.uint8), L.stream_ptr()))
    torch.cuda.synchronize()
    for i in range(3):
        ref = V.warp_u8(fr[i], Ts[i], (64, 72))
        assert (ref == 0).any() and (ref != 0).any()                      # the transforms reach outside the frame
        assert np.array_equal(ou[i].cpu().numpy()[..., ::-1], ref)       # swap_rb on the uint8 output
        assert np.array_equal(of[i].cpu().numpy(), V.from_u8(ref[..., ::-1]).transpose(2, 0, 1))


@pytest.mark.parametrize("src_hw,dst_hw", [((360, 480), (224, 224)), ((100, 90), (224, 224)), ((224, 224), (224, 224)), ((7, 5), (3, 11))])
def test_resize_kernel_matches_oracle_exactly(src_hw, dst_hw):
    L, lib = _lib()
    fr = _frames(2, *src_hw, seed=4)
    src = torch.from_numpy(np.stack(fr)).cuda()
    ou = torch.empty(2, *dst_hw, 3, dtype=torch.uint8, device="cuda")
    L.check(lib.smirk_resize_linear_u8(L.ptr(src, torch.uint8), 2, *src_hw, *dst_hw, 0, None, L.ptr(ou, torch.uint8), L.stream_ptr()))
    torch.cuda.synchronize()
    for i in range(2):
        assert np.array_equal(ou[i].cpu().numpy(), V.resize_linear_u8(fr[i], (dst_hw[1], dst_hw[0])))


def test_hull_mask_kernel_matches_oracle_exactly():
    L, lib = _lib()
    lms = _landmarks(5, 224, 224, 7)
    lms.append(np.array([[10.0, 10, 0], [10, 10, 0], [50, 50, 0], [30, 30, 0]] + [[20.0, 20, 0]] * 474))     # collinear / duplicate points
    lms.append(np.array([[-30.0, 100, 0], [300, 90, 0], [100, -20, 0], [110, 260, 0]] + [[100.0, 100, 0]] * 474))   # hull leaves the image
    x = torch.from_numpy(np.stack([l[:, :2].astype(np.int32) for l in lms]).astype(np.float32)).cuda()
    out = torch.empty(len(lms), 1, 224, 224, device="cuda")
    L.check(lib.smirk_hull_mask(L.ptr(x), len(lms), 478, 2, 224, 224, L.ptr(out), L.stream_ptr()))
    torch.cuda.synchronize()
    for i, l in enumerate(lms):
        assert np.array_equal(out[i, 0].cpu().numpy(), V.hull_mask(l, (224, 224)).astype(np.float32)), i


def test_interp_and_u8_conversions():
    import torch.nn.functional as F
    L, lib = _lib()
    x = torch.rand(2, 3, 224, 224)
    for hw in ((480, 640), (224, 224), (100, 333)):
        o = torch.empty(2, 3, *hw, device="cuda")
        L.check(lib.smirk_interp_bilinear_f32(L.ptr(x.cuda()), 6, 224, 224, *hw, L.ptr(o), L.stream_ptr()))
        assert np.abs(o.cpu().numpy() - V.interp_bilinear(x.numpy(), hw)).max() < 2e-7
        assert (o.cpu() - F.interpolate(x, hw, mode="bilinear")).abs().max().item() < 1e-5     # torch-CPU kernel rounds its weights differently
    grid = torch.zeros(2, 224, 448, 3, dtype=torch.uint8, device="cuda")
    L.check(lib.smirk_f32_nchw_to_u8_grid(L.ptr(x.cuda()), 2, 224, 224, 1, L.ptr(grid, torch.uint8), 448, 224, L.stream_ptr()))
    g = grid.cpu().numpy()
    assert not g[:, :, :224].any()
    assert np.array_equal(g[:, :, 224:], V.to_u8(x.numpy()).transpose(0, 2, 3, 1)[..., ::-1])
    u = torch.from_numpy(np.stack(_frames(2, 50, 60, 9))).cuda()
    f = torch.empty(2, 3, 50, 60, device="cuda")
    L.check(lib.smirk_u8_hwc_to_f32_nchw(L.ptr(u, torch.uint8), 2, 50, 60, 1, L.ptr(f), L.stream_ptr()))
    assert np.array_equal(f.cpu().numpy(), V.from_u8(u.cpu().numpy()[..., ::-1]).transpose(0, 3, 1, 2))


@pytest.fixture(scope="module")
def modules(sandbox):
    from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator, masking
    cwd = os.getcwd(); os.chdir(sandbox)
    try:
        fl, rn = FLAME().cuda(), Renderer().cuda()
        prob = masking.load_probabilities_per_FLAME_triangle().cuda()
    finally:
        os.chdir(cwd)
    enc = SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict()); enc = enc.cuda().eval()
    gen = SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5); gen.load_state_dict(G.synth_state_dict())
    return enc, fl, rn, gen.cuda().eval(), prob


def _reference_loop(modules, frames, lmks, crop, render_orig):
    """demo_video.py:107-214 frame by frame with the oracle's pre/post-processing around the (already parity-tested) GPU modules."""
    enc, fl, rn, _, _ = modules
    outs = []
    for f, kp in zip(frames, lmks):
        Hv, Wv = f.shape[:2]
        if crop:
            T = V.crop_transform(kp[:, :2], scale=1.4, image_size=224)
            cropped = V.warp_u8(f, np.linalg.inv(T), (224, 224))
        else:
            cropped = f
        cropped = V.resize_linear_u8(cropped[..., ::-1], (224, 224))
        img = torch.from_numpy(V.from_u8(cropped).transpose(2, 0, 1)[None].copy()).cuda()
        with torch.no_grad():
            o = enc(img)
            flo = fl.forward(o)
            rend = rn.forward(flo["vertices"], o["cam"])["rendered_img"].cpu().numpy()
        if render_orig:
            if crop:
                r8 = V.to_u8(rend[0].transpose(1, 2, 0))
                rend_o = V.from_u8(V.warp_u8(r8, T, (Hv, Wv))).transpose(2, 0, 1)[None]
            else:
                rend_o = V.interp_bilinear(rend, (Hv, Wv))
            full = V.from_u8(f[..., ::-1]).transpose(2, 0, 1)[None]
            grid = np.concatenate([full, rend_o], 3)
        else:
            grid = np.concatenate([img.cpu().numpy(), rend], 3)
        outs.append(V.to_u8(grid[0].transpose(1, 2, 0))[..., ::-1])
    return outs


@pytest.mark.parametrize("crop,render_orig,hw", [(True, False, (300, 420)), (True, True, (300, 420)), (False, False, (240, 320)),
                                                 (False, True, (240, 320)), (False, False, (224, 224))])
def test_video_pipeline_matches_frame_by_frame_reference_loop(modules, crop, render_orig, hw):
    from smirk_amd import VideoPipeline
    enc, fl, rn, gen, prob = modules
    n = 11                                                     # 11 frames, batch 4 -> ragged last batch, 3 batches over 2 slots
    frames, lmks = _frames(n, *hw, seed=5), _landmarks(n, *hw, seed=5)
    vp = VideoPipeline(enc, fl, rn, batch_size=4, crop=crop, render_orig=render_orig)
    got = list(vp.run(iter(frames), iter(lmks)))
    ref = _reference_loop(modules, frames, lmks, crop, render_orig)
    assert len(got) == n
    for i in range(n):
        assert got[i].shape == ref[i].shape and got[i].dtype == np.uint8
        d = np.abs(got[i].astype(int) - ref[i].astype(int))
        # identical pre/post arithmetic; the encoder at batch 4 vs batch 1 can move a rendered pixel by an ulp -> at most +-1 after x255
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, (i, d.max(), (d > 0).mean())
        assert np.array_equal(got[i][:, :ref[i].shape[1] // 2], ref[i][:, :ref[i].shape[1] // 2])      # the input panel is exact


def test_video_pipeline_with_generator_and_requirements(modules):
    from smirk_amd import VideoPipeline
    enc, fl, rn, gen, prob = modules
    frames, lmks = _frames(6, 256, 256, seed=8), _landmarks(6, 256, 256, seed=8)
    vp = VideoPipeline(enc, fl, rn, gen, prob, batch_size=4, crop=True, use_smirk_generator=True)
    torch.manual_seed(0)
    out = list(vp.run(frames, lmks))
    assert len(out) == 6 and out[0].shape == (224, 672, 3)
    two = list(VideoPipeline(enc, fl, rn, batch_size=4, crop=True).run(frames, lmks))
    for a, b in zip(out, two):
        assert np.array_equal(a[:, :448], b)                   # first two panels do not depend on the generator
        assert a[:, 448:].std() > 1.0                          # the reconstruction panel is a real image
    with pytest.raises(ValueError):
        list(vp.run(frames, [None] * 6))                       # demo_video.py:177-179: no landmarks -> cannot build the hull mask
    with pytest.raises(ValueError):
        VideoPipeline(enc, fl, rn, use_smirk_generator=True)


def test_video_pipeline_tolerates_missing_landmarks_without_crop_or_generator(modules):
    """demo_video.py:110-119: without --crop / --use_smirk_generator the landmarks are never consumed, so frames where mediapipe found nothing
    (None), in any mix with frames that have landmarks, must stream through unchanged."""
    from smirk_amd import VideoPipeline
    enc, fl, rn, _, _ = modules
    frames, lmks = _frames(6, 240, 320, seed=4), _landmarks(6, 240, 320, seed=4)
    ref = list(VideoPipeline(enc, fl, rn, batch_size=4).run(frames, lmks))
    mixed = [None, lmks[1], None, None, lmks[4], lmks[5]]
    got = list(VideoPipeline(enc, fl, rn, batch_size=4).run(frames, mixed))
    assert len(got) == 6
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
