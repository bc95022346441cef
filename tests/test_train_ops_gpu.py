"""GPU parity of the individual TRAINING kernels (csrc/train.hip + the data-gradient use of the forward conv kernels), one op at a time,
against float64 torch autograd of the op the reference's module graph contains (smirk_generator.py:88-119 `_block`, :121-178 ResnetBlock,
:40-49 up-convs and head).

Why per-op: a whole-network gradient is ill-conditioned at the ReLU / max-pool switching points (the reference's own fp32 run differs from its
fp64 run by 0.5-8 % for that reason — tests/test_generator_train_gpu.py measures it), so the tight bound lives HERE, where every input is built
to stay clear of the switching points: an op fed the same operands must agree with float64 to fp32 round-off.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 3e-6            # relative to max|reference|, split-fp16 x3 / fp32 data path


def _ops():
    from smirk_amd import generator_train as T
    return T, T._Ops(torch.device("cuda"))


def _act(t):
    """fp32 NHWC -> (split16 device tensor, the exactly-representable values as float64 NCHW on the CPU)"""
    from smirk_amd.smirk_generator import _split16, split16_to_float
    B, H, W, C = t.shape
    s = _split16(t.reshape(-1, C).cuda()).reshape(B, H, W, C)
    return s, split16_to_float(s).cpu().double().permute(0, 3, 1, 2).contiguous()


def _val(s):
    from smirk_amd.smirk_generator import split16_to_float
    return split16_to_float(s).cpu().double().permute(0, 3, 1, 2)


def _rel(a, b):
    return (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("relu,res", [(True, False), (False, True), (False, False)])
@pytest.mark.parametrize("shape", [(3, 6, 10, 32), (2, 4, 4, 512), (1, 16, 16, 64), (2, 40, 40, 72)])
def test_batchnorm_train_forward_backward(shape, relu, res):
    """train-mode BatchNorm forward (+ running statistics) and backward against float64 autograd"""
    T, ops = _ops()
    _batchnorm_case(T, ops, shape, relu, res)


def _batchnorm_case(T, ops, shape, relu, res):
    B, H, W, C = shape
    g = _gen(C + H)
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g)); bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    rm0, rv0 = bn.running_mean.cpu().double(), bn.running_var.cpu().double()
    z = torch.randn(B, H, W, C, generator=g) * 2 + 0.7
    ga, be = bn.weight.detach().cpu().double(), bn.bias.detach().cpu().double()
    for _ in range(8):                                        # keep the pre-activation clear of the ReLU switching point
        zs, z64 = _act(z)
        pre = F.batch_norm(z64, None, None, ga, be, True, 0.1, 1e-5)
        bad = pre.abs() < 2e-3
        if not relu or not bad.any():
            break
        z = z + 0.05 * bad.permute(0, 2, 3, 1).float()
    rs, r64 = _act(torch.randn(B, H, W, C, generator=g)) if res else (None, None)
    dys, dy64 = _act(torch.randn(B, H, W, C, generator=g))
    y, mean, inv = ops.bn_forward(zs, bn, relu, residual=rs)
    zr = z64.clone().requires_grad_(True)
    gr, br = ga.clone().requires_grad_(True), be.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    yr = F.batch_norm(zr, rm, rv, gr, br, True, 0.1, 1e-5)
    if res:
        yr = yr + r64
    if relu:
        yr = F.relu(yr)
    e = dict(y=_rel(_val(y), yr.detach()), rm=_rel(bn.running_mean.cpu(), rm), rv=_rel(bn.running_var.cpu(), rv))
    yr.backward(dy64)
    dz, dg, db = ops.bn_backward(zs, dys, bn, mean, inv, relu)
    e.update(dz=_rel(_val(dz), zr.grad), dg=_rel(dg.cpu(), gr.grad), db=_rel(db.cpu(), br.grad))
    assert int(bn.num_batches_tracked) == 1
    assert all(v < TOL for v in e.values()), e


@pytest.mark.parametrize("B,H,W,cin,cout,k,reflect", [(2, 8, 8, 32, 64, 3, False), (3, 4, 4, 512, 512, 3, True), (1, 3, 3, 512, 512, 3, True),
                                                       (2, 2, 2, 512, 512, 3, True), (2, 16, 12, 8, 32, 3, False), (1, 64, 64, 32, 32, 3, False),
                                                       (2, 8, 8, 32, 8, 1, False), (2, 6, 6, 256, 512, 1, False),
                                                       # few-channel 3x3 layers with W % 16 == 0: the all-taps halo kernel (32/64 x 32/64 channels), ragged K splits
                                                       (2, 16, 16, 32, 32, 3, False), (1, 48, 32, 64, 32, 3, False), (3, 16, 48, 32, 64, 3, False),
                                                       (1, 32, 16, 64, 64, 3, False), (2, 112, 112, 64, 64, 3, False),
                                                       # partial M / N tiles (MobileNetV3 widths), a K split that ends inside the last image
                                                       (3, 7, 7, 40, 120, 1, False), (2, 14, 14, 160, 72, 1, False), (5, 5, 3, 24, 200, 3, False)])
@pytest.mark.parametrize("mode", [2, 1, 0])
def test_conv_weight_gradient(B, H, W, cin, cout, k, reflect, mode):
    """smirk_conv_wgrad_f32 against autograd's weight gradient: split-fp16 x3 on the fp16 matrix pipe with LDS transpose reads (mode 2 = the default, 1 = one
    chunk per barrier) and the exact-fp32 MFMA kernels (mode 0); generic split-K tiles and the all-taps halo kernels, ragged K splits, partial M / N tiles"""
    T, ops = _ops()
    g = _gen(H * cin + cout)
    xs, x64 = _act(torch.randn(B, H, W, cin, generator=g))
    ds, d64 = _act(torch.randn(B, H, W, cout, generator=g))
    ops.lib.smirk_conv_wgrad_set_mode(mode)
    try:
        dw = ops.wgrad(ds, xs, B, H, W, cout, cin, k, reflect=reflect)
    finally:
        ops.lib.smirk_conv_wgrad_set_mode(-1)
    w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    xin = F.pad(x64, (1, 1, 1, 1), mode="reflect") if reflect else x64
    F.conv2d(xin, w, padding=0 if (reflect or k == 1) else 1).backward(d64)
    got = T._to_conv_weight_grad(dw, cout, cin, k)
    assert _rel(got.cpu(), w.grad) < TOL


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 8, 8, 32, 64), (1, 16, 16, 8, 32), (2, 4, 4, 512, 512), (1, 32, 32, 64, 32)])
def test_conv_data_gradient_zero_pad(B, H, W, cin, cout):
    """dL/dx of Conv2d(3x3, pad 1) = the forward kernel over dL/dz with the 180-degree rotated, Cin<->Cout swapped weights"""
    T, ops = _ops()
    g = _gen(cin + 3 * cout)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    ds, d64 = _act(torch.randn(B, H, W, cout, generator=g))
    got = ops.conv(ds, None, T._pack_dgrad(wt.cuda()), B, H, W, cin)
    from smirk_amd.smirk_generator import _split16, split16_to_float
    w64 = split16_to_float(_split16(wt.permute(0, 2, 3, 1).reshape(cout, -1).cuda()).reshape(1, 1, cout, -1)).reshape(cout, 3, 3, cin).permute(0, 3, 1, 2).cpu().double()
    x = torch.zeros(B, cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w64, padding=1).backward(d64)
    assert _rel(_val(got), x.grad) < TOL


@pytest.mark.parametrize("B,H,W,C", [(2, 4, 4, 512), (1, 3, 3, 512), (2, 2, 2, 512), (1, 14, 14, 512)])
def test_conv_data_gradient_reflect_pad(B, H, W, C):
    """ResnetBlock: ReflectionPad2d(1) + Conv2d(3x3, pad 0).  dL/dx = fold(full correlation of dL/dz), + the identity branch's gradient"""
    T, ops = _ops()
    from smirk_amd import _lib as L
    from smirk_amd.smirk_generator import _split16, split16_to_float
    g = _gen(H + C)
    wt = torch.randn(C, C, 3, 3, generator=g) * 0.05
    ds, d64 = _act(torch.randn(B, H, W, C, generator=g))
    as_, a64 = _act(torch.randn(B, H, W, C, generator=g))
    dpad = ops.conv(ds, None, T._pack_dgrad(wt.cuda()), B, H, W, C, pad=2, out_hw=(H + 2, W + 2))
    out = torch.empty(B, H, W, C, device="cuda")
    L.check(ops.lib.smirk_reflect_pad1_backward_split16(L.ptr(dpad), L.ptr(as_), L.ptr(out), B, H, W, C, ops.st))
    w64 = split16_to_float(_split16(wt.permute(0, 2, 3, 1).reshape(C, -1).cuda()).reshape(1, 1, C, -1)).reshape(C, 3, 3, C).permute(0, 3, 1, 2).cpu().double()
    x = torch.zeros(B, C, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w64).backward(d64)
    assert _rel(_val(out), x.grad + a64) < TOL


@pytest.mark.parametrize("B,H,W,C,add", [(2, 8, 8, 32, True), (1, 6, 10, 64, False), (3, 2, 2, 256, True)])
def test_maxpool_backward_with_skip_gradient(B, H, W, C, add):
    T, ops = _ops()
    from smirk_amd import _lib as L
    g = _gen(H * W + C)
    t = torch.randn(B, H, W, C, generator=g)
    t[0, :2, :2, :4] = 0.25                                   # an exact tie: the first element of the window takes the gradient (ATen's scan order)
    ts, t64 = _act(t)
    ds, d64 = _act(torch.randn(B, H // 2, W // 2, C, generator=g))
    as_, a64 = _act(torch.randn(B, H, W, C, generator=g)) if add else (None, None)
    out = torch.empty(B, H, W, C, device="cuda")
    L.check(ops.lib.smirk_maxpool2x2_backward_split16(L.ptr(ts), L.ptr(ds), L.ptr(as_, allow_none=True), L.ptr(out), B, H, W, C, ops.st))
    x = t64.clone().requires_grad_(True)
    F.max_pool2d(x, 2, 2).backward(d64)
    want = x.grad + (a64 if add else 0)
    if add:
        assert _rel(_val(out), want) < 1e-6                   # the sum of the two gradients is re-rounded to the split16 storage (22 bits)
    else:
        assert torch.equal(_val(out), want)                   # a routing op: exact


@pytest.mark.parametrize("B,h,w,cin,cout", [(2, 4, 4, 512, 256), (1, 8, 6, 64, 32), (2, 2, 2, 256, 128)])
def test_conv_transpose_backward(B, h, w, cin, cout):
    """ConvTranspose2d(k=2, s=2) backward = space-to-depth of dL/dy, then a 1x1 convolution (data) and a 1x1 weight gradient"""
    T, ops = _ops()
    from smirk_amd import _lib as L
    from smirk_amd.smirk_generator import _split16, split16_to_float
    g = _gen(cin + cout + h)
    wt = torch.randn(cin, cout, 2, 2, generator=g) * 0.1
    xs, x64 = _act(torch.randn(B, h, w, cin, generator=g))
    gs, g64 = _act(torch.randn(B, 2 * h, 2 * w, cout, generator=g))
    s2d = torch.empty(B, h, w, 4 * cout, device="cuda")
    L.check(ops.lib.smirk_space_to_depth2_split16(L.ptr(gs), L.ptr(s2d), B, h, w, cout, ops.st))
    gb = ops.colsum(gs)
    gw = ops.wgrad(xs, s2d, B, h, w, cin, 4 * cout, 1).reshape(cin, 2, 2, cout).permute(0, 3, 1, 2)
    wd = _split16(wt.cuda().permute(0, 2, 3, 1).reshape(cin, 4 * cout).contiguous())
    gx = ops.conv(s2d, None, wd, B, h, w, cin, k=1)
    w64 = split16_to_float(wd.reshape(1, 1, cin, 4 * cout)).reshape(cin, 2, 2, cout).permute(0, 3, 1, 2).cpu().double().contiguous().requires_grad_(True)
    x = x64.clone().requires_grad_(True)
    b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(x, w64, b, stride=2).backward(g64)
    assert _rel(_val(gx), x.grad) < TOL
    assert _rel(gw.cpu(), w64.grad) < TOL
    assert _rel(gb.cpu(), b.grad) < TOL


@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 32), (1, 8, 24, 64)])
def test_head_conv1x1_sigmoid_backward(B, H, W, C):
    T, ops = _ops()
    from smirk_amd import _lib as L
    g = _gen(H + W + C)
    wt = (torch.randn(3, C, generator=g) * 0.2)
    bias = torch.randn(3, generator=g) * 0.1
    ds, d64 = _act(torch.randn(B, H, W, C, generator=g))
    gy = torch.randn(B, 3, H, W, generator=g)
    y = torch.empty(B, 3, H, W, device="cuda")
    wg, bg = wt.cuda().contiguous(), bias.cuda().contiguous()
    L.check(ops.lib.smirk_conv1x1_sigmoid_nchw_split16(L.ptr(ds), L.ptr(wg), L.ptr(bg), L.ptr(y), B, H, W, C, 3, ops.st))
    dd, dl8 = torch.empty(B, H, W, C, device="cuda"), torch.empty(B, H, W, 8, device="cuda")
    L.check(ops.lib.smirk_conv1x1_sigmoid_backward_split16(L.ptr(gy.cuda().contiguous()), L.ptr(y), L.ptr(wg), L.ptr(dd), L.ptr(dl8), B, H, W, C, 3, ops.st))
    gw = ops.wgrad(dl8, ds, B, H, W, 8, C, 1)[:3]
    gb = ops.colsum(dl8)[:3]
    x = d64.clone().requires_grad_(True)
    w64 = wt.double().reshape(3, C, 1, 1).requires_grad_(True)
    b64 = bias.double().requires_grad_(True)
    yr = torch.sigmoid(F.conv2d(x, w64, b64))
    yr.backward(gy.double())
    assert (y.cpu().double() - yr.detach()).abs().max().item() < 2e-6
    assert _rel(_val(dd), x.grad) < 2e-5          # y (fp32, from the forward) enters as y*(1-y): its 1e-7 error is relative to 0.25, not to dd
    assert _rel(gw.cpu().reshape(3, C, 1, 1), w64.grad) < 2e-5 and _rel(gb.cpu(), b64.grad) < 2e-5


# ---- encoder-specific training kernels (csrc/train_encoder.hip) -----------------------------------------------------------------------
def _enc_ops():
    from smirk_amd import encoder_train as E
    return E, E._EncOps(torch.device("cuda"))


def _same_pad(x, k, s):
    """timm Conv2dSame padding (layers/padding.py): total = max((ceil(n/s)-1)*s + k - n, 0), leading = total // 2"""
    import math
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    return F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])


@pytest.mark.parametrize("C", [24, 40, 72, 200, 960])
def test_batchnorm_train_channel_counts_that_do_not_divide_256(C):
    """the MobileNet widths: 8-channel groups that do not tile a 256-thread block evenly"""
    T, ops = _ops()
    g = _gen(C)
    bn = torch.nn.BatchNorm2d(C, eps=1e-3).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.3 + 1.5)
    ga, be = bn.weight.detach().cpu().double(), bn.bias.detach().cpu().double()
    zs, z64 = _act(torch.randn(3, 5, 7, C, generator=g) + 0.3)
    dys, dy64 = _act(torch.randn(3, 5, 7, C, generator=g))
    y, mean, inv = ops.bn_forward(zs, bn, False)
    zr = z64.clone().requires_grad_(True)
    gr, br = ga.clone().requires_grad_(True), be.clone().requires_grad_(True)
    yr = F.batch_norm(zr, None, None, gr, br, True, 0.1, 1e-3)
    yr.backward(dy64)
    dz, dg, db = ops.bn_backward(zs, dys, bn, mean, inv, False)
    e = dict(y=_rel(_val(y), yr.detach()), dz=_rel(_val(dz), zr.grad), dg=_rel(dg.cpu(), gr.grad), db=_rel(db.cpu(), br.grad))
    assert all(v < TOL for v in e.values()), e


@pytest.mark.parametrize("B,H,W,C,stride,add", [(2, 8, 8, 16, 1, True), (2, 9, 7, 24, 2, False), (1, 14, 14, 72, 2, False), (3, 7, 7, 960, 1, False),
                                                 (2, 12, 12, 64, 2, False), (2, 6, 6, 40, 1, True)])
def test_depthwise_conv_backward(B, H, W, C, stride, add):
    """depthwise 3x3, stride 1 (pad 1) and stride 2 (TF 'SAME': asymmetric padding on even sizes) — forward, data gradient (+ skip gradient), weight gradient"""
    E, ops = _enc_ops()
    from smirk_amd import _lib as L
    g = _gen(H * C + stride)
    wt = torch.randn(C, 1, 3, 3, generator=g) * 0.3
    xs, x64 = _act(torch.randn(B, H, W, C, generator=g))
    w9c = wt.reshape(C, 9).t().contiguous().cuda()
    z = ops.depthwise(xs, w9c, stride)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    ds, d64 = _act(torch.randn(B, Ho, Wo, C, generator=g))
    as_, a64 = _act(torch.randn(B, H, W, C, generator=g)) if add else (None, None)
    x = x64.clone().requires_grad_(True)
    w64 = wt.double().requires_grad_(True)
    zr = F.conv2d(_same_pad(x, 3, stride), w64, stride=stride, groups=C) if stride == 2 else F.conv2d(x, w64, padding=1, groups=C)
    assert _rel(_val(z), zr.detach()) < TOL
    zr.backward(d64)
    dx = ops.depthwise_dgrad(ds, w9c, as_, B, H, W, C, stride)
    assert _rel(_val(dx), x.grad + (a64 if add else 0)) < TOL
    dw = ops.depthwise_wgrad(ds, xs, stride)
    assert _rel(dw.cpu(), w64.grad) < TOL


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 45, 37), (3, 64, 48)])
def test_stem_conv_backward(B, H, W):
    """Conv2d(3, 16, 3, stride 2, TF 'SAME') from an NCHW fp32 image: bare forward, weight gradient, image gradient"""
    E, ops = _enc_ops()
    from smirk_amd import _lib as L
    g = _gen(H + W)
    wt = torch.randn(16, 3, 3, 3, generator=g) * 0.2
    img = torch.rand(B, 3, H, W, generator=g)
    wp = wt.permute(0, 2, 3, 1).reshape(16, 27).contiguous().cuda()
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    z = torch.empty(B, Ho, Wo, 16, device="cuda")
    imgc = img.cuda()
    L.check(ops.lib.smirk_stem_conv_s2_raw_split16(L.ptr(imgc), L.ptr(wp), L.ptr(z), B, H, W, 16, ops.st))
    x = img.double().requires_grad_(True)
    w64 = wt.double().requires_grad_(True)
    zr = F.conv2d(_same_pad(x, 3, 2), w64, stride=2)
    assert _rel(_val(z), zr.detach()) < TOL
    ds, d64 = _act(torch.randn(B, Ho, Wo, 16, generator=g))
    zr.backward(d64)
    nws = ops.lib.smirk_stem_conv_s2_wgrad_workspace_bytes(16)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    dw = torch.empty(16, 27, device="cuda")
    L.check(ops.lib.smirk_stem_conv_s2_wgrad_split16(L.ptr(imgc), L.ptr(ds), L.ptr(dw), B, H, W, 16, L.ptr(ws, torch.uint8), nws, ops.st))
    assert _rel(dw.reshape(16, 3, 3, 3).permute(0, 3, 1, 2).cpu(), w64.grad) < TOL
    dimg = torch.empty(B, 3, H, W, device="cuda")
    L.check(ops.lib.smirk_stem_conv_s2_dgrad_split16(L.ptr(ds), L.ptr(wp), L.ptr(dimg), B, H, W, 16, ops.st))
    assert _rel(dimg.cpu(), x.grad) < TOL


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 7, 7, 72, 24), (3, 5, 5, 160, 960), (2, 14, 14, 16, 64), (1, 28, 28, 40, 120)])
def test_pointwise_conv_backward_mobilenet_widths(B, H, W, cin, cout):
    """1x1 convolution at the MobileNet widths (not multiples of the 32-channel K chunk / 128-wide tile): forward, data gradient through the
    transposed weight (+ skip gradient in the epilogue), weight gradient"""
    E, ops = _enc_ops()
    g = _gen(cin + cout)
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(cout, cin, 1, 1, generator=g) * 0.2)
    conv = conv.cuda()
    xs, x64 = _act(torch.randn(B, H, W, cin, generator=g))
    ds, d64 = _act(torch.randn(B, H, W, cout, generator=g))
    as_, a64 = _act(torch.randn(B, H, W, cin, generator=g))
    from smirk_amd.smirk_generator import split16_to_float
    w64 = split16_to_float(E._pw(conv).reshape(1, 1, cout, cin)).reshape(cout, cin, 1, 1).cpu().double().requires_grad_(True)
    x = x64.clone().requires_grad_(True)
    zr = F.conv2d(x, w64)
    z = ops.pointwise(xs, E._pw(conv), cout)
    assert _rel(_val(z), zr.detach()) < TOL
    zr.backward(d64)
    dx = ops.pointwise(ds, E._pw_t(conv), cin, residual=as_)
    assert _rel(_val(dx), x.grad + a64) < TOL
    dw = ops.wgrad(ds, xs, B, H, W, cout, cin, 1)
    assert _rel(dw.reshape(cout, cin, 1, 1).cpu(), w64.grad) < TOL


@pytest.mark.parametrize("B,hw,C,N", [(3, 9, 960, 55), (2, 49, 576, 6), (5, 4, 960, 300)])
def test_pool_linear_head_backward(B, hw, C, N):
    E, ops = _enc_ops()
    from smirk_amd import _lib as L
    g = _gen(C + N)
    h = int(hw ** 0.5)
    fs, f64 = _act(torch.randn(B, h, h, C, generator=g))
    wt, bias = torch.randn(N, C, generator=g) * 0.05, torch.randn(N, generator=g)
    wg, bg = wt.cuda().contiguous(), bias.cuda().contiguous()
    pooled, out = torch.empty(B, C, device="cuda"), torch.empty(B, N, device="cuda")
    L.check(ops.lib.smirk_gap_linear_split16(L.ptr(fs), L.ptr(wg), L.ptr(bg), L.ptr(out), L.ptr(pooled), B, h * h, C, N, ops.st))
    f = f64.clone().requires_grad_(True)
    w64, b64 = wt.double().requires_grad_(True), bias.double().requires_grad_(True)
    o = F.linear(F.adaptive_avg_pool2d(f, 1).flatten(1), w64, b64)
    assert _rel(out.cpu(), o.detach()) < TOL
    go = torch.randn(B, N, generator=g)
    o.backward(go.double())
    dw, db, df = torch.empty(N, C, device="cuda"), torch.empty(N, device="cuda"), torch.empty(B, h, h, C, device="cuda")
    L.check(ops.lib.smirk_gap_linear_backward_split16(L.ptr(go.cuda().contiguous()), L.ptr(wg), L.ptr(pooled), L.ptr(dw), L.ptr(db), L.ptr(df), B, h * h, C, N,
                                                      ops.st))
    assert _rel(dw.cpu(), w64.grad) < TOL and _rel(db.cpu(), b64.grad) < TOL and _rel(_val(df), f.grad) < TOL


@pytest.mark.parametrize("cout,cin,k,off,sub,pad", [(32, 6, 3, 0, None, 8), (64, 32, 3, 0, None, None), (32, 64, 3, 32, 32, None), (512, 512, 3, 0, None, None),
                                                   (24, 72, 1, 0, None, None), (960, 160, 1, 0, None, None)])
def test_weight_packing_kernel_is_bitwise_the_torch_packing(cout, cin, k, off, sub, pad):
    """smirk_pack_conv_weights_split16 (one launch: forward image + 180-degree rotated, Cin<->Cout swapped data-gradient image, optional input-channel
    slice and zero channel padding) against the torch formulation the per-op tests above are written with"""
    T, ops = _ops()
    g = _gen(cout + cin + k)
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.3).cuda()
    n = cin - off if sub is None else sub
    fwd, dgr = ops.pack(w, off, n, cin_pad=pad)
    ws = w[:, off:off + n]
    if k == 3:
        want_f = T._pack_fwd(ws, pad)
        want_d = T._pack_dgrad(ws, pad)
    else:
        from smirk_amd.smirk_generator import _split16
        want_f = _split16(ws.reshape(cout, n).contiguous())
        want_d = _split16(ws.reshape(cout, n).t().contiguous())
    assert fwd.shape == want_f.shape and torch.equal(fwd.view(torch.int32), want_f.view(torch.int32))
    assert dgr.shape == want_d.shape and torch.equal(dgr.view(torch.int32), want_d.view(torch.int32))


def test_batchnorm_train_is_deterministic():
    """fixed-order fp64 partial sums: the BatchNorm forward / backward reproduce themselves bit for bit"""
    T, ops = _ops()
    g = _gen(77)
    bn = torch.nn.BatchNorm2d(120).cuda().train()
    zs, _ = _act(torch.randn(4, 28, 28, 120, generator=g) * 1.5 + 0.3)
    dys, _ = _act(torch.randn(4, 28, 28, 120, generator=g))
    runs = []
    for _ in range(3):
        y, mean, inv = ops.bn_forward(zs, bn, True)
        dz, dg, db = ops.bn_backward(zs, dys, bn, mean, inv, True)
        runs.append([t.clone() for t in (y, mean, inv, dz, dg, db)])
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(runs[0], r))


# ---- F16X1: BASELINE config 5's 16-bit class (one MFMA per product block, the hi halves only) -----------------------------------------------------------------------
def _hi64(s, nchw=True):
    """the fp16 hi halves of a split16 NHWC tensor as float64 (what the F16X1 kernels multiply)"""
    B, H, W, C = s.shape
    hi = s.reshape(B, H, W, C // 8, 8).view(torch.float16).reshape(B, H, W, C // 8, 2, 8)[..., 0, :].reshape(B, H, W, C)
    hi = hi.cpu().double()
    return hi.permute(0, 3, 1, 2).contiguous() if nchw else hi


@pytest.mark.parametrize("B,H,W,cin,cout,k", [(2, 16, 16, 32, 64, 3), (1, 32, 32, 64, 32, 3), (2, 14, 14, 512, 512, 3), (3, 7, 7, 40, 120, 1), (1, 56, 56, 128, 128, 3),
                                                (2, 28, 28, 256, 24, 1)])
def test_conv_f16x1_is_the_fp16_rounded_product_with_f32_accumulation(B, H, W, cin, cout, k):
    """smirk_conv_igemm_f16x1 == conv2d(fp16(x), fp16(w)) accumulated in fp32: exact semantics, so the bound is fp32 round-off — and it differs from the
    x3 result by the 2^-11 the lo halves carry (the two entries are not accidentally the same kernel)"""
    T, _ = _ops()
    o1, o3 = T._Ops(torch.device("cuda"), arith="f16x1"), T._Ops(torch.device("cuda"))
    g = _gen(cin + cout + k)
    xs, _ = _act(torch.randn(B, H, W, cin, generator=g))
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.1
    from smirk_amd.smirk_generator import _split16
    wp = _split16(wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().cuda())
    got1, got3 = o1.conv(xs, None, wp, B, H, W, cout, k=k), o3.conv(xs, None, wp, B, H, W, cout, k=k)
    w_hi = wp.reshape(cout, -1).view(torch.float16).reshape(cout, -1, 2, 8)[:, :, 0, :].reshape(cout, k, k, cin).permute(0, 3, 1, 2).cpu().double()
    ref = F.conv2d(_hi64(xs), w_hi, padding=(k - 1) // 2)
    assert _rel(_val(got1), ref) < TOL
    d13 = _rel(_val(got1), _val(got3))
    assert 2e-5 < d13 < 3e-3, d13


@pytest.mark.parametrize("B,H,W,cin,cout,k,reflect", [(2, 8, 8, 32, 64, 3, False), (2, 4, 4, 512, 512, 3, True), (2, 16, 16, 32, 32, 3, False), (1, 32, 16, 64, 64, 3, False),
                                                        (3, 7, 7, 40, 120, 1, False), (2, 112, 112, 64, 64, 3, False)])
def test_conv_weight_gradient_f16x1(B, H, W, cin, cout, k, reflect):
    """smirk_conv_wgrad_f16x1 (generic split-K tiles and the all-taps halo kernels) == autograd's weight gradient of the fp16-rounded operands"""
    T, _ = _ops()
    o1 = T._Ops(torch.device("cuda"), arith="f16x1")
    g = _gen(H * cin + cout + 1)
    xs, _ = _act(torch.randn(B, H, W, cin, generator=g))
    ds, _ = _act(torch.randn(B, H, W, cout, generator=g))
    dw = o1.wgrad(ds, xs, B, H, W, cout, cin, k, reflect=reflect)
    w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    x64, d64 = _hi64(xs), _hi64(ds)
    xin = F.pad(x64, (1, 1, 1, 1), mode="reflect") if reflect else x64
    F.conv2d(xin, w, padding=0 if (reflect or k == 1) else 1).backward(d64)
    assert _rel(T._to_conv_weight_grad(dw, cout, cin, k).cpu(), w.grad) < TOL


@pytest.mark.parametrize("arith", ["f16x3", "f16x1"])
def test_weight_gradient_in_parameter_layout_equals_the_packed_one(arith):
    """smirk_conv_wgrad_param writes the split-K sum straight into nn.Conv2d / nn.ConvTranspose2d parameter layouts (two sources into one tensor, padded input
    channels dropped): bit-identical to the packed result permuted / concatenated on the host, which is what the backward did through round 4."""
    from smirk_amd import generator_train as T
    ops = T._Ops(torch.device("cuda"), arith=arith)
    g = _gen(123)
    B, H, W = 2, 16, 16
    # two-source decoder convolution: cat((up 32 ch, skip 64 ch), 1) -> 64
    dz, _ = _act(torch.randn(B, H, W, 64, generator=g))
    x0, _ = _act(torch.randn(B, H, W, 32, generator=g))
    x1, _ = _act(torch.randn(B, H, W, 64, generator=g))
    ref = torch.cat([T._to_conv_weight_grad(ops.wgrad(dz, x0, B, H, W, 64, 32, 3), 64, 32), T._to_conv_weight_grad(ops.wgrad(dz, x1, B, H, W, 64, 64, 3), 64, 64)], 1)
    out = torch.full((64, 96, 3, 3), float("nan"), device="cuda")
    ops.wgrad_param(dz, x0, B, H, W, 64, 32, 3, out, cin_off=0)
    ops.wgrad_param(dz, x1, B, H, W, 64, 64, 3, out, cin_off=32)
    assert torch.equal(out, ref)
    # first layer: 6 real of 8 padded input channels
    x8, _ = _act(torch.randn(B, H, W, 8, generator=g))
    d32, _ = _act(torch.randn(B, H, W, 32, generator=g))
    ref = T._to_conv_weight_grad(ops.wgrad(d32, x8, B, H, W, 32, 8, 3), 32, 8, cin_real=6)
    out = torch.full((32, 6, 3, 3), float("nan"), device="cuda")
    ops.wgrad_param(d32, x8, B, H, W, 32, 8, 3, out, cin_real=6)
    assert torch.equal(out, ref)
    # reflect-padded ResNet convolution
    a, _ = _act(torch.randn(B, 6, 6, 64, generator=g)); b_, _ = _act(torch.randn(B, 6, 6, 64, generator=g))
    ref = T._to_conv_weight_grad(ops.wgrad(a, b_, B, 6, 6, 64, 64, 3, reflect=True), 64, 64)
    out = torch.full((64, 64, 3, 3), float("nan"), device="cuda")
    ops.wgrad_param(a, b_, B, 6, 6, 64, 64, 3, out, reflect=True)
    assert torch.equal(out, ref)
    # ConvTranspose2d(64, 32, 2, 2): dz = the layer's input (64 ch), x = space-to-depth of the output gradient (4 x 32 ch)
    xin, _ = _act(torch.randn(B, H, W, 64, generator=g)); s2d, _ = _act(torch.randn(B, H, W, 128, generator=g))
    ref = ops.wgrad(xin, s2d, B, H, W, 64, 128, 1).reshape(64, 2, 2, 32).permute(0, 3, 1, 2).contiguous()
    out = torch.full((64, 32, 2, 2), float("nan"), device="cuda")
    ops.wgrad_param(xin, s2d, B, H, W, 64, 128, 1, out, layout=2)
    assert torch.equal(out, ref)


# ---- round 6: BatchNorm statistics from the producing convolution's accumulators -------------------------------------------------------
@pytest.mark.parametrize("arith", ["f16x3", "f16x1"])
@pytest.mark.parametrize("B,H,W,cin,cout,k", [(2, 14, 14, 16, 64, 1), (3, 7, 7, 72, 24, 1), (1, 28, 28, 40, 120, 1), (5, 7, 7, 160, 960, 1), (2, 9, 11, 112, 672, 1),
                                              (1, 56, 56, 128, 128, 3), (2, 28, 28, 256, 256, 3), (3, 10, 6, 64, 64, 3), (1, 17, 13, 32, 32, 3),
                                              (1, 112, 112, 64, 64, 3), (2, 64, 64, 64, 32, 3), (5, 14, 14, 512, 512, 3)])
def test_conv_epilogue_statistics_equal_the_sums_of_the_stored_output(B, H, W, cin, cout, k, arith):
    """smirk_conv_igemm_stats_split16: the raw convolution's output is bit-identical to the plain entry's, and the per-tile partial sums it leaves (sum z, sum z^2 from the
    fp32 accumulators, one row per M tile and wave row; ragged last tiles, Cout not a multiple of the 128-wide tile, K not a multiple of 32; the implicit-GEMM walks,
    the halo kernel's 256-row tiles and the persistent ring kernels' per-workgroup rows) add up to the column sums of the stored tensor.  Then the BatchNorm that consumes them (finalise from partials + apply) equals the BatchNorm that reduces the stored tensor."""
    from smirk_amd import _lib as L
    T, ops = _ops()
    ops = T._Ops(torch.device("cuda"), arith=arith)
    g = _gen(cin * 7 + cout + k)
    xs, _ = _act(torch.randn(B, H, W, cin, generator=g))
    w = torch.randn(cout, cin, k, k, generator=g) * (1.5 / (cin * k * k) ** 0.5)
    wf = T._pack_fwd(w.cuda()) if k == 3 else T._split16(w.reshape(cout, cin).cuda())
    z_plain = ops.conv(xs, None, wf, B, H, W, cout, k=k)
    z, st = ops.conv_stats(xs, None, wf, B, H, W, cout, k=k)
    assert torch.equal(z, z_plain)
    assert st is not None, "the implicit-GEMM kernels serve these shapes and carry the statistics epilogue"
    part, rows = st
    M = B * H * W
    assert 0 < rows <= L.lib().smirk_conv_stats_rows_max(_desc(B, H, W, cin, cout, k)) == part.shape[0]
    p = part[:rows].double().sum(0).cpu()                                         # [cout][2]
    zv = _val(z).permute(0, 2, 3, 1).reshape(M, cout)
    s1, s2 = zv.sum(0), (zv * zv).sum(0)
    # the accumulators hold z BEFORE it is rounded into the split format (2^-22 relative per element) and are summed in fp32 per tile (<= 256 rows)
    assert (p[:, 0] - s1).abs().max().item() <= 2e-6 * zv.abs().sum(0).max().item() + 1e-6
    assert (p[:, 1] - s2).abs().max().item() <= 4e-6 * s2.max().item() + 1e-9
    bn = torch.nn.BatchNorm2d(cout).cuda().train()
    bn2 = torch.nn.BatchNorm2d(cout).cuda().train()
    y_a, mu_a, iv_a = ops.bn_forward(z, bn, True, stats=st)
    y_b, mu_b, iv_b = ops.bn_forward(z, bn2, True)
    assert (mu_a - mu_b).abs().max().item() <= 1e-6 * max(1.0, mu_b.abs().max().item())
    assert ((iv_a - iv_b).abs() / iv_b).max().item() <= 5e-6
    assert _rel(_val(y_a), _val(y_b)) < 1e-5
    assert torch.allclose(bn.running_mean, bn2.running_mean, rtol=1e-5, atol=1e-6) and torch.allclose(bn.running_var, bn2.running_var, rtol=1e-5, atol=1e-7)
    assert int(bn.num_batches_tracked) == 1


def _desc(B, H, W, cin, cout, k):
    from smirk_amd import _lib as L
    d = L.SmirkConvDesc()
    d.B, d.H, d.W, d.C0, d.C1, d.Cout, d.KH, d.KW, d.stride = B, H, W, cin, 0, cout, k, k, 1
    d.pad_t = d.pad_l = (k - 1) // 2
    d.Ho, d.Wo, d.pad_mode, d.act, d.out_mode = H, W, L.PAD_ZERO, L.ACT_NONE, L.OUT_NHWC
    return d


def test_pack_plan_relayouts_depthwise_stem_and_transposed_conv_weights_in_its_one_launch():
    """round 6: the depthwise [C,1,3,3] -> [9][C], stem [Cout,3,3,3] -> [Cout][(ky,kx,c)] and ConvTranspose2d [Cin,Cout,2,2] -> (forward 1x1 image, data-gradient image)
    re-layouts that a training step needs ride in the PackPlan's single launch (SmirkPackJob kinds) instead of ~6 ATen launches each: the kernel's images are bitwise the
    torch formulation's, for weights that CHANGED since the plan was sealed (the optimiser step), mixed with ordinary convolution jobs"""
    from smirk_amd import _lib as L
    T, ops = _ops()
    g = _gen(99)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.3).cuda()
    w_dw, w_c, w_st, w_ct, w_dw2 = mk(72, 1, 3, 3), mk(64, 32, 3, 3), mk(16, 3, 3, 3), mk(64, 32, 2, 2), mk(960, 1, 3, 3)
    plan = T.PackPlan()
    ops.plan = plan

    def requests():
        return [plan.request_special(ops, w_dw, L.PACK_DEPTHWISE), plan.request(ops, w_c, 0, None, None, True, True), plan.request_special(ops, w_st, L.PACK_STEM),
                plan.request_special(ops, w_ct, L.PACK_CONVT2X2), plan.request_special(ops, w_dw2, L.PACK_DEPTHWISE)]

    first = requests()                                                          # recording pass: torch formulation
    plan.seal([w_dw, w_c, w_st, w_ct, w_dw2], torch.device("cuda"))
    with torch.no_grad():
        for w in (w_dw, w_c, w_st, w_ct, w_dw2):
            w.mul_(-1.7).add_(0.05)                                             # "the optimiser stepped": same storage, new values
    want = [(w_dw.reshape(72, 9).t().contiguous(), None), (T._pack_fwd(w_c), T._pack_dgrad(w_c)), (w_st.permute(0, 2, 3, 1).reshape(16, 27).contiguous(), None),
            (T._split16(w_ct.permute(2, 3, 1, 0).reshape(4 * 32, 64).contiguous()), T._split16(w_ct.permute(0, 2, 3, 1).reshape(64, 4 * 32).contiguous())),
            (w_dw2.reshape(960, 9).t().contiguous(), None)]
    plan.run(ops)                                                               # ONE launch
    got = requests()
    torch.cuda.synchronize()
    for (gf, gd), (wf, wd), (ff, fd) in zip(got, want, first):
        assert gf.data_ptr() == ff.data_ptr()                                   # the plan hands out its own buffers again
        assert gf.shape == wf.shape and torch.equal(gf.view(torch.int32), wf.view(torch.int32))
        assert (gd is None) == (wd is None)
        if wd is not None:
            assert gd.shape == wd.shape and torch.equal(gd.view(torch.int32), wd.view(torch.int32))


@pytest.mark.parametrize("B,H,W,C,stride", [(2, 8, 8, 16, 1), (2, 9, 7, 24, 2), (1, 14, 14, 72, 2), (3, 7, 7, 960, 1), (2, 12, 12, 64, 2), (4, 28, 28, 120, 1), (2, 57, 33, 200, 2)])
def test_depthwise_train_kernel_output_and_statistics(B, H, W, C, stride):
    """round 6: dwconv3x3_train_kernel (weights of a lane's channel group held in registers, stage-1 sums of the BatchNorm statistics added while storing): the output is
    bit-identical to the inference kernel's raw output, the fp64 partial rows add up to the column sums of the stored tensor, and the BatchNorm that consumes them equals
    the BatchNorm that reduces the stored tensor (channel counts that do not divide 256, odd sizes, both strides)."""
    E, ops = _enc_ops()
    g = _gen(H * C + stride + 5)
    wt = torch.randn(C, 1, 3, 3, generator=g) * 0.3
    xs, _ = _act(torch.randn(B, H, W, C, generator=g))
    w9c = wt.reshape(C, 9).t().contiguous().cuda()
    z_ref = ops.depthwise(xs, w9c, stride)
    z, st = ops.depthwise_stats(xs, w9c, stride)
    assert torch.equal(z, z_ref) and st is not None
    part, rows = st
    zv = _val(z).permute(0, 2, 3, 1).reshape(-1, C)
    p = part[:rows * C * 2].reshape(rows, C, 2).sum(0).cpu()
    s1, s2 = zv.sum(0), (zv * zv).sum(0)
    assert (p[:, 0] - s1).abs().max().item() <= 2e-6 * zv.abs().sum(0).max().item() + 1e-6      # (the sums take z before it is rounded into the split format)
    assert (p[:, 1] - s2).abs().max().item() <= 4e-6 * s2.max().item() + 1e-9
    bn, bn2 = torch.nn.BatchNorm2d(C).cuda().train(), torch.nn.BatchNorm2d(C).cuda().train()
    y_a, mu_a, iv_a = ops.bn_forward(z, bn, True, stats=st)
    y_b, mu_b, iv_b = ops.bn_forward(z, bn2, True)
    assert (mu_a - mu_b).abs().max().item() <= 1e-6 * max(1.0, mu_b.abs().max().item()) and ((iv_a - iv_b).abs() / iv_b).max().item() <= 5e-6
    assert _rel(_val(y_a), _val(y_b)) < 1e-5
    assert torch.allclose(bn.running_var, bn2.running_var, rtol=1e-5, atol=1e-7) and int(bn.num_batches_tracked) == 1
