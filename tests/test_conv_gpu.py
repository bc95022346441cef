"""GPU parity of the convolution kernels against each other and against torch fp64:
   exact-fp32 MFMA implicit GEMM  vs  split-fp16 (f16x3) implicit GEMM  vs  split-fp16 persistent halo-patch kernel."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 3e-5      # abs on O(1..10) outputs: both arithmetic modes are fp32-class


def _run(B, H, C0, C1, Cout, k=3, reflect=False, seed=0, residual=False, relu=True, only_split=False):
    from smirk_amd import _lib as L
    from smirk_amd.smirk_generator import _split16, split16_to_float
    lib, dev = L.lib(), torch.device("cuda")
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(B, H, H, C0, generator=g)
    x1 = torch.randn(B, H, H, C1, generator=g) if C1 else None
    w = torch.randn(Cout, k * k * (C0 + C1), generator=g) * 0.05
    sc, sh = torch.rand(Cout, generator=g) + .5, torch.randn(Cout, generator=g)
    # torch fp64 reference (NCHW)
    xin = torch.cat([x0, x1], -1) if C1 else x0
    wt = w.reshape(Cout, k, k, C0 + C1).permute(0, 3, 1, 2).double()
    xp = xin.permute(0, 3, 1, 2).double()
    if k == 3:
        xp = F.pad(xp, (1, 1, 1, 1), mode="reflect" if reflect else "constant")
    ref = (F.conv2d(xp, wt) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).permute(0, 2, 3, 1)
    res = torch.randn(B, H, H, Cout, generator=g) if residual else None
    if residual:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    d = L.SmirkConvDesc()
    d.B, d.H, d.W, d.C0, d.C1, d.Cout, d.KH, d.KW, d.stride = B, H, H, C0, C1, Cout, k, k, 1
    d.pad_t = d.pad_l = (k - 1) // 2
    d.Ho, d.Wo, d.pad_mode, d.out_mode = H, H, (L.PAD_REFLECT if reflect else L.PAD_ZERO), L.OUT_NHWC
    d.act = L.ACT_RELU if relu else L.ACT_NONE
    P = L.ptr
    x0d, x1d, wd, scd, shd = x0.to(dev), (x1.to(dev) if C1 else None), w.to(dev), sc.to(dev), sh.to(dev)
    o32 = torch.empty(B, H, H, Cout, device=dev)
    resd = res.to(dev) if residual else None
    if not only_split:
        L.check(lib.smirk_conv_igemm_f32(d, P(x0d), P(x1d, allow_none=True), P(wd), P(scd), P(shd), P(resd, allow_none=True), P(o32), L.stream_ptr()))

    def split(t):
        o = torch.empty_like(t)
        L.check(lib.smirk_f32_to_split16(P(t), P(o), t.numel(), L.stream_ptr()))
        return o
    s0, s1, ws = split(x0d), (split(x1d) if C1 else None), _split16(wd)
    os_ = torch.empty(B, H, H, Cout, device=dev)
    rs = split(resd) if residual else None
    L.check(lib.smirk_conv_igemm_f16x3(d, P(s0), P(s1, allow_none=True), P(ws), P(scd), P(shd), P(rs, allow_none=True), P(os_), L.stream_ptr()))
    torch.cuda.synchronize()
    if only_split:
        return ref, os_
    return ref, o32.cpu().double(), split16_to_float(os_).cpu().double()


# (B, H, C0, C1, Cout): patch-kernel shapes (H >= 64, Cout 32/64, weights resident) incl. >2 patches per workgroup, 2 sources,
# the 8-channel first layer; and implicit-GEMM shapes (small images, wide layers, reflect padding, ragged M)
CASES = [(2, 112, 128, 0, 64), (3, 112, 64, 64, 64), (2, 64, 128, 0, 64),       # streamed-weights patch kernel (Cout 64, weights > LDS)
         (2, 224, 32, 32, 32), (3, 224, 8, 0, 32), (2, 224, 32, 0, 32), (5, 112, 32, 0, 64), (2, 64, 32, 32, 32),
         (2, 32, 32, 0, 32), (3, 28, 64, 64, 128), (5, 14, 128, 0, 256), (1, 56, 64, 0, 64)]


@pytest.mark.parametrize("cfg", CASES)
def test_conv3x3_modes_agree(cfg):
    ref, o32, os_ = _run(*cfg)
    assert (o32 - ref).abs().max().item() < TOL
    assert (os_ - ref).abs().max().item() < TOL


def test_conv_reflect_and_1x1():
    ref, o32, os_ = _run(3, 14, 64, 0, 64, reflect=True)
    assert (o32 - ref).abs().max().item() < TOL and (os_ - ref).abs().max().item() < TOL
    ref, o32, os_ = _run(2, 28, 40, 0, 72, k=1)                     # encoder-style pointwise conv, K = 40 (partial chunk)
    assert (o32 - ref).abs().max().item() < TOL and (os_ - ref).abs().max().item() < TOL


# shapes served by the 8-wave ping-pong kernel (conv_pp.hip: 3x3, power-of-two channel counts >= 32, Cout % 128 == 0, M >= 1024): the generator's
# 56^2 / 28^2 / 14^2 layers incl. the decoder's two-source convs, the reflect-padded residual blocks (residual add, no ReLU) and ragged last tiles
PP_CASES = [dict(B=4, H=56, C0=64, C1=0, Cout=128), dict(B=2, H=56, C0=128, C1=128, Cout=128), dict(B=5, H=28, C0=256, C1=256, Cout=256),
            dict(B=9, H=14, C0=512, C1=0, Cout=512, reflect=True, residual=True, relu=False), dict(B=7, H=14, C0=256, C1=0, Cout=512),
            dict(B=3, H=28, C0=128, C1=0, Cout=256, reflect=True)]


@pytest.mark.parametrize("cfg", PP_CASES)
def test_conv_pingpong_kernel_matches_fp64_and_the_128x128_kernel_bitwise(cfg):
    """conv_pp_kernel walks K in the same order and issues the same MFMA sequence per accumulator as conv_igemm_kernel's channel-major walk:
    within tolerance of torch fp64, and BIT-IDENTICAL to the kernel it replaces (SMIRK_IGEMM_PP=0)."""
    import os
    from smirk_amd.smirk_generator import split16_to_float
    os.environ["SMIRK_IGEMM_HALO"] = "0"             # conv_halo.hip takes these shapes first by default
    os.environ["SMIRK_IGEMM_PP"] = "all"             # also the 28x28 / 56x56 shapes the dispatcher leaves on the 128x128 kernel by default
    try:
        ref, a = _run(**cfg, only_split=True)
    finally:
        del os.environ["SMIRK_IGEMM_PP"]
    os.environ["SMIRK_IGEMM_PP"] = "0"
    try:
        _, b = _run(**cfg, only_split=True)
    finally:
        del os.environ["SMIRK_IGEMM_PP"], os.environ["SMIRK_IGEMM_HALO"]
    assert (split16_to_float(a).cpu().double() - ref).abs().max().item() < TOL
    assert torch.equal(a, b)


# shapes served by the halo-staged ping-pong kernel (conv_halo.hip): every PP case plus images narrower / wider than the tile's row structure,
# tiles that straddle several images (14x14: 1.3 images per 256-row tile), ragged last tiles, zero and reflect padding, two sources with different widths
# (an even number of 32-channel chunks everywhere: that is when the 128x128 kernel walks K channel-major too, the order the bitwise comparison needs)
HALO_CASES = PP_CASES + [dict(B=11, H=14, C0=512, C1=0, Cout=512), dict(B=6, H=14, C0=128, C1=64, Cout=128, reflect=True),
                         dict(B=2, H=56, C0=128, C1=0, Cout=128, reflect=True), dict(B=3, H=31, C0=64, C1=0, Cout=128),
                         dict(B=2, H=32, C0=64, C1=0, Cout=128, reflect=True), dict(B=1, H=63, C0=64, C1=128, Cout=256),
                         dict(B=300, H=2, C0=64, C1=0, Cout=128, reflect=True), dict(B=40, H=6, C0=128, C1=0, Cout=128)]


@pytest.mark.parametrize("cfg", HALO_CASES)
def test_conv_halo_kernel_matches_fp64_and_the_128x128_kernel_bitwise(cfg):
    """conv_halo_kernel stages one pixel halo per channel chunk and reads all nine taps from it; K order and per-accumulator MFMA sequence are those of
    conv_igemm_kernel's channel-major walk: within tolerance of torch fp64 and BIT-IDENTICAL to the 128x128 kernel (SMIRK_IGEMM_HALO=0, SMIRK_IGEMM_PP=0)."""
    import os
    from smirk_amd.smirk_generator import split16_to_float
    os.environ["SMIRK_IGEMM_HALO"] = "all"
    try:
        ref, a = _run(**cfg, only_split=True)
    finally:
        del os.environ["SMIRK_IGEMM_HALO"]
    os.environ["SMIRK_IGEMM_HALO"] = "0"; os.environ["SMIRK_IGEMM_PP"] = "0"
    try:
        _, b = _run(**cfg, only_split=True)
    finally:
        del os.environ["SMIRK_IGEMM_HALO"], os.environ["SMIRK_IGEMM_PP"]
    assert (split16_to_float(a).cpu().double() - ref).abs().max().item() < TOL
    assert torch.equal(a, b)


def test_split16_roundtrip_is_fp32_class():
    from smirk_amd import _lib as L
    x = (torch.randn(1 << 16) * torch.exp(torch.randn(1 << 16) * 3)).cuda()
    s, back = torch.empty_like(x), torch.empty_like(x)
    L.check(L.lib().smirk_f32_to_split16(L.ptr(x), L.ptr(s), x.numel(), L.stream_ptr()))
    L.check(L.lib().smirk_split16_to_f32(L.ptr(s), L.ptr(back), x.numel(), L.stream_ptr()))
    err, mag = (back - x).abs(), x.abs()
    normal = (mag >= 2.0 ** -10) & (mag < 6e4)          # hi is a normal fp16 => 22 significand bits survive
    assert (err[normal] / mag[normal]).max().item() < 2.0 ** -21
    assert err[mag < 2.0 ** -10].max().item() < 2.0 ** -31   # below that the error is bounded in absolute terms (fp16 subnormal grid / 2^11)
