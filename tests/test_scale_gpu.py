"""GPU parity AT THE BENCHMARKED SIZES (BASELINE.json configs 3 and 4): B = 128 per GPU for the full path (the 8-GPU shard) and B = 1024 (the whole
job in one pass on one GPU, bench.py's default), B = 256 for encoder + FLAME + renderer, FLAME at B = 512 in test_flame_gpu.py.

The CPU oracle cannot run 128-256 frames of the generator in a few seconds, so every stage is pinned twice:
  1. against the oracle on a strided SUB-SAMPLE of the big batch (every k-th frame, compared in full), and
  2. through a size-independent property on EVERY frame: a frame's result does not depend on which batch it travels in
     (`stage(x)[i:j] == stage(x[i:j])`, bitwise) — the K walk, tile shapes and reduction orders of the kernels are functions of the
     layer geometry only, so the large-batch launch (784-tile grids, XCD-aware tile order, many patches per workgroup, offsets close
     to 2^31 bytes) must reproduce the small-batch launch the oracle checked, bit for bit.
The whole pipeline (serial and 2-stream overlapped) is then repeated 8 times at B = 128 and every run must be bit-identical and finite.
"""
import os

import numpy as np
import pytest
import torch

from oracle import assets as A
from oracle import generator_ref as G
from oracle import mobilenet_ref as M
from oracle.flame_ref import FlameRef
from oracle.render_ref import RendererRef

pytestmark = pytest.mark.gpu

OUT_TOL = 2e-5          # generator sigmoid output, abs (same bound as tests/test_generator_gpu.py)
PIX_TOL = 2e-6          # rendered pixels (same bound as tests/test_render_gpu.py)
from enc_tolerances import VS_FP32 as ENC_TOL      # 4 x the measured max |HIP - fp32 oracle| over these very batches (tests/enc_tolerances.py)


@pytest.fixture(scope="module")
def mods(sandbox):
    from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator
    cwd = os.getcwd(); os.chdir(sandbox)
    try:
        fl, rn = FLAME().cuda(), Renderer().cuda()
    finally:
        os.chdir(cwd)
    esd, gsd = M.synth_encoder_state_dict(), G.synth_state_dict()
    enc = SmirkEncoder(); enc.load_state_dict(esd, strict=True); enc = enc.cuda().eval()
    gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(gsd, strict=True); gen = gen.cuda().eval()
    return dict(enc=enc, flame=fl, rend=rn, gen=gen, esd=esd, gsd=gsd)


def _slices(B, n):
    return [(i, min(i + n, B)) for i in range(0, B, n)]


@pytest.mark.parametrize("B", [128, 1024])
def test_generator_bench_batch_subsample_vs_oracle_and_batch_invariance(mods, B):
    """B = 128: one rank's shard of the 1024-frame job on 8 GPUs; B = 1024: the whole job in ONE pass on one GPU, as bench.py runs it (the 224x224
    activations are 6.6 GB each: byte offsets beyond 2^32, the buffer-addressed DMA paths fall back to 64-bit pointers)"""
    gen, gsd = mods["gen"], mods["gsd"]
    x = A.synth_generator_input(B, seed=4100)
    xg = x.cuda()
    with torch.no_grad():
        y = gen(xg)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    sub = list(range(5, B, B // 8))                   # 8 frames spread over the batch (different tiles / XCDs / patch lists)
    yr = G.forward(gsd, x[sub])
    assert (y[sub].cpu() - yr).abs().max().item() < OUT_TOL
    for lo, hi in _slices(B, 24 if B <= 128 else 168):  # neither divides B: the small launches are ragged in a different way
        with torch.no_grad():
            ys = gen(xg[lo:hi].contiguous())
        assert torch.equal(ys, y[lo:hi]), (lo, hi)


@pytest.mark.parametrize("B", [128, 256, 1024])
def test_encoder_bench_batch_vs_oracle_and_batch_invariance(mods, B):
    enc, esd = mods["enc"], mods["esd"]
    img = A.synth_images(B, seed=5200 + B)
    ig = img.cuda()
    with torch.no_grad():
        o = enc(ig)
    torch.cuda.synchronize()
    sub = list(range(1, B, B // 32))
    ref = M.SmirkEncoderRef(); ref.load_state_dict(esd); ref.eval()
    with torch.no_grad():
        r = ref(img[sub])
    for k, tol in ENC_TOL.items():
        assert torch.isfinite(o[k]).all(), k
        assert (o[k][sub].cpu() - r[k]).abs().max().item() < tol, k
    for lo, hi in _slices(B, 40 if B <= 256 else 200):
        with torch.no_grad():
            s = enc(ig[lo:hi].contiguous())
        for k in ENC_TOL:
            assert torch.equal(s[k], o[k][lo:hi]), (k, lo, hi)


@pytest.mark.parametrize("B", [128, 256, 1024])
def test_flame_renderer_bench_batch_vs_oracle_and_batch_invariance(mods, sandbox, B):
    """config 3's tail (FLAME -> Renderer) at the bench batch: vertices < 1e-5 L2, raster indices bit-exact, pixels < 2e-6 on the
    sub-sample; bitwise batch invariance on every frame."""
    fl, rn = mods["flame"], mods["rend"]
    p = A.synth_flame_params(B, seed=6300 + B)
    p["shape_params"] *= 0.4
    cam = A.synth_cam(B, seed=6300 + B)
    pg = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
    camg = torch.from_numpy(cam).cuda()
    with torch.no_grad():
        f = fl.forward(pg)
        r = rn.forward(f["vertices"], camg, _aux=True, landmarks_fan=f["landmarks_fan"], landmarks_mp=f["landmarks_mp"])
    torch.cuda.synchronize()
    sub = list(range(3, B, B // 16))
    ps = {k: v[sub] for k, v in p.items()}
    fr = FlameRef(sandbox).forward(ps)
    v = f["vertices"][sub].cpu().numpy()
    assert np.sqrt(((v - fr["vertices"]) ** 2).sum(-1)).max() < 1e-5
    rr = RendererRef(sandbox).forward(v, cam[sub])                     # oracle renderer on the GPU's vertex bits
    p2f = rr["_aux"]["pix_to_face"].astype(np.int64)
    mine = r["_aux"]["pix_to_face"][sub].cpu().numpy()
    base = (np.asarray(sub, dtype=np.int64) * 3408)[:, None, None]
    assert np.array_equal(mine, np.where(p2f >= 0, p2f + base, -1))     # packed index b*Ff + f exactly as pytorch3d packs it
    assert np.array_equal(r["_aux"]["bary"][sub].cpu().numpy(), rr["_aux"]["bary"])
    assert np.abs(r["rendered_img"][sub].cpu().numpy() - rr["rendered_img"]).max() < PIX_TOL
    assert np.array_equal(r["transformed_vertices"][sub].cpu().numpy(), rr["transformed_vertices"])
    for lo, hi in _slices(B, 40 if B <= 256 else 200):
        with torch.no_grad():
            fs = fl.forward({k: t[lo:hi].contiguous() for k, t in pg.items()})
            rs = rn.forward(fs["vertices"], camg[lo:hi].contiguous(), _aux=True)
        assert torch.equal(fs["vertices"], f["vertices"][lo:hi])
        assert torch.equal(fs["landmarks_fan"], f["landmarks_fan"][lo:hi]) and torch.equal(fs["landmarks_mp"], f["landmarks_mp"][lo:hi])
        assert torch.equal(rs["rendered_img"], r["rendered_img"][lo:hi])
        local = rs["_aux"]["pix_to_face"]
        want = r["_aux"]["pix_to_face"][lo:hi]
        assert torch.equal(torch.where(local >= 0, local + lo * 3408, local), want)


def test_full_pipeline_B128_repeated_serial_and_overlapped(mods):
    """BASELINE config 4 shard (128 frames): encode -> FLAME -> render -> generate, serial and software-pipelined over two streams with
    two different batches in flight, 8 repetitions: every run finite and bit-identical to the first serial run; generator sub-sample
    checked against the oracle on the GPU's own rendered images."""
    from smirk_amd.pipeline import OverlappedPipeline, SmirkPipeline
    pipe = SmirkPipeline(mods["enc"], mods["flame"], mods["rend"], mods["gen"])
    B = 128
    batches = []
    for s in (7001, 7002):
        batches.append((A.synth_images(B, seed=s).cuda(), A.synth_generator_input(B, seed=s)[:, 3:].contiguous().cuda()))
    keys = ("vertices", "rendered_img", "reconstructed_img", "cam", "landmarks_fan", "landmarks_mp", "expression_params")
    first = [pipe(i, k) for i, k in batches]
    torch.cuda.synchronize()
    for o in first:
        for k in keys:
            assert torch.isfinite(o[k]).all(), k
        cov = (o["rendered_img"][:, 0] != 0).float().mean().item()
        assert 0.02 < cov < 0.95, cov                 # the synthetic encoder puts a face on the screen
    sub = [9, 77, 120]
    o0 = first[0]
    x = torch.cat([o0["rendered_img"][sub].cpu(), batches[0][1][sub].cpu()], 1)
    assert (o0["reconstructed_img"][sub].cpu() - G.forward(mods["gsd"], x)).abs().max().item() < OUT_TOL
    for trial in range(8):
        if trial % 2 == 0:
            got = [pipe(i, k) for i, k in batches]
        else:
            run = OverlappedPipeline(pipe, generator_streams=1 + (trial // 2) % 2)       # 1 or 2 generator streams
            got = [run.submit(i, k) for i, k in batches + batches]
            while True:
                o = run.flush()
                if o is None:
                    break
                got.append(o)
            got = [o for o in got if o is not None]
            assert len(got) == 4
            for a, b in zip(got[:2], got[2:]):                                          # the same batch twice through the pipeline
                for k in keys:
                    assert torch.equal(a[k], b[k]), (trial, k, "repeat")
            got = got[:2]
        torch.cuda.synchronize()
        for a, b in zip(first, got):
            for k in keys:
                assert torch.equal(a[k], b[k]), (trial, k)


def test_rotating_pipeline_B128_identical_to_one_batch_at_a_time(mods):
    """BASELINE config 3's schedule in bench.py: consecutive generator-less batches rotate over 2 / 3 streams (the raster tail of batch i under the backbones
    of batch i+1).  Same kernels, same inputs: every output bit-identical to SmirkPipeline.__call__, in submission order, over repeated rounds."""
    from smirk_amd.pipeline import RotatingPipeline, SmirkPipeline
    pipe = SmirkPipeline(mods["enc"], mods["flame"], mods["rend"], None)
    batches = [A.synth_images(128, seed=s).cuda() for s in (7101, 7102, 7103)]
    keys = ("vertices", "rendered_img", "cam", "landmarks_fan", "landmarks_mp", "expression_params", "shape_params", "pose_params")
    first = [pipe(i) for i in batches]
    torch.cuda.synchronize()
    for lanes in (2, 3, 2):
        run = RotatingPipeline(pipe, lanes=lanes)
        got = [run.submit(i) for i in batches + batches]
        while (o := run.flush()) is not None:
            got.append(o)
        got = [o for o in got if o is not None]
        assert len(got) == 6
        torch.cuda.synchronize()
        for a, b in zip(first + first, got):
            for k in keys:
                assert torch.equal(a[k], b[k]), (lanes, k)
    with pytest.raises(ValueError):
        RotatingPipeline(SmirkPipeline(mods["enc"], mods["flame"], mods["rend"], mods["gen"]))


def test_pipeline_hull_mask_path_finite_B128(mods, sandbox):
    """the masking utilities inside the step (demo.py:138-165) at the bench batch: finite, masked image is a sub-set of the photo's pixels
    plus sampled points, output in (0,1)."""
    from smirk_amd import masking as MK
    from smirk_amd.pipeline import SmirkPipeline
    cwd = os.getcwd(); os.chdir(sandbox)
    try:
        fp = MK.load_probabilities_per_FLAME_triangle().cuda()
    finally:
        os.chdir(cwd)
    pipe = SmirkPipeline(mods["enc"], mods["flame"], mods["rend"], mods["gen"], face_probabilities=fp)
    B = 128
    img = A.synth_images(B, seed=8001).cuda()
    hull = (A.synth_generator_input(B, seed=8001)[:, 3:4] != 0).float().contiguous().cuda()
    torch.manual_seed(11)
    o = pipe(img, hull_mask=hull)
    torch.cuda.synchronize()
    y, m = o["reconstructed_img"], o["masked_img"]
    assert torch.isfinite(y).all() and torch.isfinite(m).all()
    assert y.min() >= 0 and y.max() <= 1
    nz = m != 0
    # masking keeps / removes photo pixels; the sampled in-face points carry a multiplicative N(1, 0.05) noise (masking.py:84-87)
    assert ((m[nz] - img[nz]).abs() <= 0.3 * img[nz] + 1e-6).all()
    assert (m[nz] == img[nz]).float().mean().item() > 0.5
    assert 0.05 < nz.float().mean().item() < 0.95


@pytest.mark.parametrize("B", [128, 1024])
def test_overlapped_hull_mask_schedule_is_bitwise_the_serial_pipeline(mods, sandbox, B):
    """The schedule bench.py times by default: OverlappedPipeline.submit(img, hull_mask=...) — the masking utilities (demo.py:138-165) run on
    the generator stream while the next batch's encoder / FLAME / renderer kernels are co-resident on the front stream.  With the Philox
    stream pinned (masking.PhiloxStream), 8 repetitions with 1 and 2 generator streams must equal the serial SmirkPipeline(..., hull_mask=...)
    bit for bit on EVERY key including masked_img.  B = 128 is a rank's shard on 8 GPUs, B = 1024 the one-GPU bench pass."""
    from smirk_amd import masking as MK
    from smirk_amd.pipeline import OverlappedPipeline, SmirkPipeline
    cwd = os.getcwd(); os.chdir(sandbox)
    try:
        fp = MK.load_probabilities_per_FLAME_triangle().cuda()
    finally:
        os.chdir(cwd)
    pipe = SmirkPipeline(mods["enc"], mods["flame"], mods["rend"], mods["gen"], face_probabilities=fp)
    batches = []
    for s in (9101, 9102):
        n = min(B, 128)                                     # 128 distinct synthetic frames, tiled to B (the CPU synthesis is the slow part)
        img = A.synth_images(n, seed=s)
        hull = (A.synth_generator_input(n, seed=s)[:, 3:4] != 0).float()
        rep = B // n
        batches.append((img.repeat(rep, 1, 1, 1).contiguous().cuda(), hull.repeat(rep, 1, 1, 1).contiguous().cuda()))
    SEED, OFF = 0x5EED5EED, (1 << 40)

    def rng(i):
        return MK.PhiloxStream(SEED + i, OFF)

    first = [pipe(im, hull_mask=hu, mask_rng=rng(i)) for i, (im, hu) in enumerate(batches)]
    torch.cuda.synchronize()
    keys = [k for k, v in first[0].items() if torch.is_tensor(v)]
    assert "masked_img" in keys and "reconstructed_img" in keys and "vertices" in keys
    for o in first:
        for k in keys:
            assert torch.isfinite(o[k].float()).all(), k
        assert (o["masked_img"] != 0).float().mean().item() > 0.05
    # the draws depend on the stream state only: a second serial run reproduces the first
    again = pipe(batches[0][0], hull_mask=batches[0][1], mask_rng=rng(0))
    for k in keys:
        assert torch.equal(again[k], first[0][k]), ("serial repeat", k)
    del again
    for trial in range(8):
        run = OverlappedPipeline(pipe, generator_streams=1 + trial % 2)
        got = [run.submit(im, hull_mask=hu, mask_rng=rng(i)) for i, (im, hu) in enumerate(batches)]
        while True:
            o = run.flush()
            if o is None:
                break
            got.append(o)
        got = [o for o in got if o is not None]
        assert len(got) == 2
        torch.cuda.synchronize()
        for a, b in zip(first, got):
            for k in keys:
                assert torch.equal(a[k], b[k]), (trial, k)
        del got, run


@pytest.mark.parametrize("radius,comp", [(10, 3), (5, 2), (10, 0), (5, 1), (3, 3), (15, 3)])
@pytest.mark.parametrize("shape", [(3, 224, 224), (2, 37, 53), (1, 64, 300)])
def test_maxpool_sq_fused_lds_kernel_equals_max_pool2d(radius, comp, shape):
    """smirk_maxpool_sq (masking.py:78,96): the one-launch LDS kernel (radius 5 / 10) and the generic two-pass fallback against
    F.max_pool2d with -inf padding, ragged heights / widths included; exact (max and 1 - x are exact)."""
    from smirk_amd import masking as MK
    B, H, W = shape
    g = torch.Generator().manual_seed(radius * 100 + comp + H)
    x = (torch.rand(B, 1, H, W, generator=g) < 0.02).float() * torch.rand(B, 1, H, W, generator=g)
    x = x.cuda()
    got = MK._maxpool_sq(x, radius, comp)
    src = 1 - x if comp & 1 else x
    want = torch.nn.functional.max_pool2d(src, 2 * radius + 1, 1, radius)
    want = 1 - want if comp & 2 else want
    assert torch.equal(got, want)
