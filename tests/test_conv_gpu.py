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
    # the sample's tail (|x| up to ~1e5) is beyond what `hi` can carry: the conversion kernel says so through the library's always-on range flag
    torch.cuda.synchronize()
    assert bool((mag >= 65520).any()) == bool(L.lib().smirk_range_flag_peek())
    L.lib().smirk_range_flag_clear()


# ---- enc1_fused.hip: conv(8 -> 32) + BN + ReLU -> conv(32 -> 32) + BN + ReLU -> e1 + maxpool(e1) in one launch (smirk_generator.py:52-53) ----------------
def _enc1_case(B, H, W, seed):
    from smirk_amd import _lib as L
    from smirk_amd.smirk_generator import _split16
    lib, dev = L.lib(), torch.device("cuda")
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, 8, generator=g)
    x[..., 6:] = 0.0                                                       # the packed network input: 6 real channels, 2 zero
    w1 = torch.randn(32, 72, generator=g) * 0.15
    w2 = torch.randn(32, 288, generator=g) * 0.06
    sc1, sh1 = torch.rand(32, generator=g) + .5, torch.randn(32, generator=g) * 0.3
    sc2, sh2 = torch.rand(32, generator=g) + .5, torch.randn(32, generator=g) * 0.3
    P = L.ptr
    t = dict(x=x, w1=w1, w2=w2, sc1=sc1, sh1=sh1, sc2=sc2, sh2=sh2)
    d = {k: v.to(dev) for k, v in t.items()}
    xs = torch.empty_like(d["x"])
    L.check(lib.smirk_f32_to_split16(P(d["x"]), P(xs), xs.numel(), L.stream_ptr()))
    w1s, w2s = _split16(d["w1"]), _split16(d["w2"])
    return L, lib, t, d, xs, w1s, w2s


def _enc1_unfused(L, lib, d, xs, w1s, w2s, B, H, W):
    P = L.ptr
    dev = xs.device
    def desc(cin):
        c = L.SmirkConvDesc()
        c.B, c.H, c.W, c.C0, c.C1, c.Cout, c.KH, c.KW, c.stride, c.pad_t, c.pad_l = B, H, W, cin, 0, 32, 3, 3, 1, 1, 1
        c.Ho, c.Wo, c.pad_mode, c.act, c.out_mode = H, W, L.PAD_ZERO, L.ACT_RELU, L.OUT_NHWC
        return c
    t1, e1, pl = torch.empty(B, H, W, 32, device=dev), torch.empty(B, H, W, 32, device=dev), torch.empty(B, H // 2, W // 2, 32, device=dev)
    L.check(lib.smirk_conv_igemm_f16x3(desc(8), P(xs), None, P(w1s), P(d["sc1"]), P(d["sh1"]), None, P(t1), L.stream_ptr()))
    L.check(lib.smirk_conv_igemm_f16x3(desc(32), P(t1), None, P(w2s), P(d["sc2"]), P(d["sh2"]), None, P(e1), L.stream_ptr()))
    L.check(lib.smirk_maxpool2x2_split16(P(e1), P(pl), B, H, W, 32, L.stream_ptr()))
    return e1, pl


def _enc1_fused(L, lib, d, xs, w1s, w2s, B, H, W):
    P = L.ptr
    dev = xs.device
    e1 = torch.full((B, H, W, 32), float("nan"), device=dev)
    pl = torch.full((B, H // 2, W // 2, 32), float("nan"), device=dev)
    L.check(lib.smirk_enc1_fused_split16(P(xs), P(w1s), P(d["sc1"]), P(d["sh1"]), P(w2s), P(d["sc2"]), P(d["sh2"]), P(e1), P(pl), B, H, W, L.stream_ptr()))
    torch.cuda.synchronize()
    return e1, pl


@pytest.mark.parametrize("B,H,W,seed", [(1, 16, 16, 0), (2, 32, 48, 1), (3, 224, 224, 2), (1, 64, 16, 3), (5, 16, 32, 4), (7, 112, 224, 5)])
def test_enc1_fused_block_matches_fp64_and_the_unfused_kernels(B, H, W, seed):
    """one patch (every halo pixel outside the image), ragged patch counts (idle groups / workgroups), the benchmark geometry: the fused block is within the
    conv tolerance of torch fp64, within fp32 rounding of the three launches it replaces, and its pooled output is EXACTLY MaxPool2d(2,2) of its own e1"""
    from smirk_amd.smirk_generator import split16_to_float
    L, lib, t, d, xs, w1s, w2s = _enc1_case(B, H, W, seed)
    e1, pl = _enc1_fused(L, lib, d, xs, w1s, w2s, B, H, W)
    u1, upl = _enc1_unfused(L, lib, d, xs, w1s, w2s, B, H, W)
    torch.cuda.synchronize()
    xq = split16_to_float(xs).cpu().double().permute(0, 3, 1, 2)             # the split16 image of x is what both paths convolve
    w1q = split16_to_float(w1s.view(1, 1, 32, 72)).view(32, 72).cpu().double()
    w2q = split16_to_float(w2s.view(1, 1, 32, 288)).view(32, 288).cpu().double()
    bn = lambda y, sc, sh: F.relu(y * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    y1 = bn(F.conv2d(xq, w1q.reshape(32, 3, 3, 8).permute(0, 3, 1, 2), padding=1), t["sc1"], t["sh1"])
    y2 = bn(F.conv2d(y1, w2q.reshape(32, 3, 3, 32).permute(0, 3, 1, 2), padding=1), t["sc2"], t["sh2"])
    ref_e1, ref_pl = y2.permute(0, 2, 3, 1), F.max_pool2d(y2, 2, 2).permute(0, 2, 3, 1)
    f_e1, f_pl = split16_to_float(e1).cpu().double(), split16_to_float(pl).cpu().double()
    assert torch.isfinite(f_e1).all() and torch.isfinite(f_pl).all()         # every pixel of both outputs was written (they were NaN-filled)
    def where(err):                                                          # which (row, column, channel) sets are wrong: names the broken index map at a glance
        bad = (err >= TOL).nonzero()
        return {n: sorted(set(bad[:, k].tolist()))[:40] for k, n in enumerate(("b", "y", "x", "c"))}
    e_e1, e_pl = (f_e1 - ref_e1).abs(), (f_pl - ref_pl).abs()
    assert e_e1.max().item() < TOL, where(e_e1)
    assert e_pl.max().item() < TOL, where(e_pl)
    # against the kernels it replaces: same arithmetic class, different k packing of conv1 (two taps per MFMA step) => fp32 rounding only
    assert (f_e1 - split16_to_float(u1).cpu().double()).abs().max().item() < 2e-5
    # pooled == max-pool of its own e1, bit for bit (max commutes with the monotone split)
    own = torch.empty_like(pl)
    L.check(lib.smirk_maxpool2x2_split16(L.ptr(e1), L.ptr(own), B, H, W, 32, L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(own, pl)


def test_enc1_fused_is_batch_invariant_and_deterministic():
    """a frame's result does not depend on the batch it travels in (patch -> workgroup assignment changes with B) nor on the run"""
    L, lib, t, d, xs, w1s, w2s = _enc1_case(6, 64, 64, 11)
    e1, pl = _enc1_fused(L, lib, d, xs, w1s, w2s, 6, 64, 64)
    e1b, plb = _enc1_fused(L, lib, d, xs, w1s, w2s, 6, 64, 64)
    assert torch.equal(e1, e1b) and torch.equal(pl, plb)
    for b in (0, 5):
        s1, sp = _enc1_fused(L, lib, d, xs[b:b + 1].contiguous(), w1s, w2s, 1, 64, 64)
        assert torch.equal(s1[0], e1[b]) and torch.equal(sp[0], pl[b])


def test_generator_forward_with_and_without_the_fused_first_block(monkeypatch):
    """smirk_generator_forward takes enc1_fused.hip by default; SMIRK_DISABLE_ENC1_FUSED=1 restores the three launches — both within the generator tolerance of each other"""
    from oracle import generator_ref as G
    from smirk_amd import SmirkGenerator
    import synthdata as synth
    gsd = G.synth_state_dict()
    gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(gsd); gen = gen.cuda().eval()
    x = synth.synth_generator_input(2, seed=3).cuda()
    with torch.no_grad():
        a = gen(x)
        monkeypatch.setenv("SMIRK_DISABLE_ENC1_FUSED", "1")
        b = gen(x)
    torch.cuda.synchronize()
    assert (a - b).abs().max().item() < 2e-5
    y = G.forward(gsd, x.cpu())
    assert (a.cpu() - y).abs().max().item() < 2e-5


# ---- conv_ring.hip: Cout = 64 on large images, weights streamed through an LDS ring, optional fused 2 x 2 max-pool ------------------------------------------
RING_CASES = [dict(B=2, H=112, C0=32, C1=0, Cout=64), dict(B=3, H=112, C0=64, C1=0, Cout=64), dict(B=2, H=112, C0=64, C1=64, Cout=64),
              dict(B=5, H=64, C0=128, C1=0, Cout=64), dict(B=1, H=64, C0=32, C1=32, Cout=64), dict(B=7, H=80, C0=64, C1=0, Cout=64),
              dict(B=2, H=224, C0=32, C1=32, Cout=32), dict(B=3, H=64, C0=64, C1=0, Cout=32), dict(B=9, H=96, C0=32, C1=32, Cout=32)]   # Cout = 32, >= 2 chunks: three workgroups per CU


@pytest.mark.parametrize("cfg", RING_CASES)
def test_conv_ring64_matches_fp64_and_the_kernels_it_replaces(cfg, monkeypatch):
    """1-4 channel chunks, one and two sources, patch counts below / above the persistent grid, a non-power-of-two image size: within the conv tolerance of torch
    fp64 and within fp32 rounding of (bit-identical to, where that kernel walks K in the same order) the round-3 kernel for the shape ($SMIRK_CONV_RING=0)"""
    from smirk_amd.smirk_generator import split16_to_float
    ref, a = _run(**cfg, only_split=True)
    monkeypatch.setenv("SMIRK_CONV_RING", "0")
    _, b = _run(**cfg, only_split=True)
    fa, fb = split16_to_float(a).cpu().double(), split16_to_float(b).cpu().double()
    assert (fa - ref).abs().max().item() < TOL
    assert (fa - fb).abs().max().item() < 1e-5
    if cfg["Cout"] == 32 or cfg["C0"] + cfg["C1"] >= 128 or cfg["C0"] + cfg["C1"] == 32:     # round 3: streamed- / resident-weights patch kernels, same K order and MFMA sequence
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,H,C", [(2, 112, 64), (3, 64, 32), (9, 112, 64)])
def test_conv3x3_pool_entry_equals_conv_then_pool(B, H, C):
    """smirk_conv3x3_pool_f16x3: the conv output is bit-identical to smirk_conv_igemm_f16x3's and the pooled output to smirk_maxpool2x2_split16 of it"""
    from smirk_amd import _lib as L
    from smirk_amd.smirk_generator import _split16
    lib, dev = L.lib(), torch.device("cuda")
    g = torch.Generator().manual_seed(B * 7 + C)
    x = torch.randn(B, H, H, C, generator=g).to(dev)
    w = (torch.randn(64, 9 * C, generator=g) * 0.05).to(dev)
    sc, sh = (torch.rand(64, generator=g) + .5).to(dev), torch.randn(64, generator=g).to(dev)
    P = L.ptr
    xs = torch.empty_like(x)
    L.check(lib.smirk_f32_to_split16(P(x), P(xs), x.numel(), L.stream_ptr()))
    ws = _split16(w)
    d = L.SmirkConvDesc()
    d.B, d.H, d.W, d.C0, d.C1, d.Cout, d.KH, d.KW, d.stride, d.pad_t, d.pad_l = B, H, H, C, 0, 64, 3, 3, 1, 1, 1
    d.Ho, d.Wo, d.pad_mode, d.act, d.out_mode = H, H, L.PAD_ZERO, L.ACT_RELU, L.OUT_NHWC
    o0, o1 = torch.empty(B, H, H, 64, device=dev), torch.full((B, H, H, 64), float("nan"), device=dev)
    p0, p1 = torch.empty(B, H // 2, H // 2, 64, device=dev), torch.full((B, H // 2, H // 2, 64), float("nan"), device=dev)
    L.check(lib.smirk_conv_igemm_f16x3(d, P(xs), None, P(ws), P(sc), P(sh), None, P(o0), L.stream_ptr()))
    L.check(lib.smirk_maxpool2x2_split16(P(o0), P(p0), B, H, H, 64, L.stream_ptr()))
    L.check(lib.smirk_conv3x3_pool_f16x3(d, P(xs), None, P(ws), P(sc), P(sh), P(o1), P(p1), L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(o0, o1) and torch.equal(p0, p1)
    d.Cout = 32                                                             # no kernel with a fused pool for this shape: the caller is told, nothing is launched
    assert lib.smirk_conv3x3_pool_f16x3(d, P(xs), None, P(ws), P(sc), P(sh), P(o1), P(p1), L.stream_ptr()) == -4     # SMIRK_ERR_UNSUPPORTED
