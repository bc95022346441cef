"""Independent cross-checks of the two UNPINNED third-party restatements (timm MobileNetV3-minimal, pytorch3d rasterize_meshes) against
code that IS on disk in this image (verdict r03 item 6).

* `transformers` ships its own TF-"SAME" padding (`modeling_mobilenet_v2.apply_tf_padding`) and its own inverted-residual block
  (`MobileNetV2InvertedResidual`: expand 1x1 -> BN -> act -> depthwise 3x3 (SAME) -> BN -> act -> reduce 1x1 -> BN, residual iff stride 1 and
  Cin == Cout).  It was written by different authors from the TF-slim original, not from timm, so agreement with `oracle/mobilenet_ref.py`'s
  `Conv2dSame` / `IR` (the restatement of timm's `Conv2dSame` / `InvertedResidual` SMIRK's encoder uses, src/smirk_encoder.py:7-12) pins the
  padding rule and the block's operator order to a second source.
* The rasteriser KATs for faces that straddle z = 0 live in tests/test_cpu_suite.py next to the other analytic KATs.
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import mobilenet_ref as M

transformers = pytest.importorskip("transformers")
from transformers.models.mobilenet_v2 import modeling_mobilenet_v2 as HF  # noqa: E402
from transformers.models.mobilenet_v2.configuration_mobilenet_v2 import MobileNetV2Config  # noqa: E402


@pytest.mark.parametrize("h,w", [(224, 224), (112, 112), (14, 14), (7, 7), (15, 9), (8, 13), (1, 1), (2, 3)])
@pytest.mark.parametrize("k,s", [(3, 2), (3, 1), (5, 2)])
def test_conv2d_same_padding_equals_transformers_tf_padding(h, w, k, s):
    """the oracle's Conv2dSame pads exactly like transformers' apply_tf_padding: same amounts, the odd pixel on the bottom / right"""
    g = torch.Generator().manual_seed(h * 131 + w * 7 + k + s)
    x = torch.randn(2, 4, h, w, generator=g)
    conv = M.Conv2dSame(4, 6, k, s, 0, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g))
        hf = nn.Conv2d(4, 6, k, s, 0, bias=False)
        hf.weight.copy_(conv.weight)
        want = hf(HF.apply_tf_padding(x, hf))
        got = conv(x)
    assert got.shape == want.shape == (2, 6, -(-h // s), -(-w // s))
    assert torch.equal(got, want)


def test_even_input_stride2_pads_bottom_right_only():
    """for even H the stride-2 3x3 'same' conv pads (top/left 0, bottom/right 1) — checked on an impulse"""
    conv = M.Conv2dSame(1, 1, 3, 2, 0, bias=False)
    with torch.no_grad():
        conv.weight.zero_(); conv.weight[0, 0, 0, 0] = 1.0        # picks the TOP-LEFT tap: output(y, x) = padded(2y, 2x)
        x = torch.arange(64, dtype=torch.float32).reshape(1, 1, 8, 8)
        y = conv(x)
    assert torch.equal(y[0, 0], x[0, 0, 0::2, 0::2])              # no padding on the top / left => tap (0,0) reads x(2y, 2x) itself


def _hf_block(ci, co, stride, mid, eps):
    cfg = MobileNetV2Config(tf_padding=True, hidden_act="relu", layer_norm_eps=eps, depth_divisible_by=8, min_depth=8, expand_ratio=1.0)
    blk = HF.MobileNetV2InvertedResidual(cfg, in_channels=ci, out_channels=co, stride=stride)
    # transformers derives the expanded width from config.expand_ratio (an integer-ish ratio in MobileNetV2); rebuild the three layers at timm's width
    blk.expand_1x1 = HF.MobileNetV2ConvLayer(cfg, in_channels=ci, out_channels=mid, kernel_size=1)
    blk.conv_3x3 = HF.MobileNetV2ConvLayer(cfg, in_channels=mid, out_channels=mid, kernel_size=3, stride=stride, groups=mid)
    blk.reduce_1x1 = HF.MobileNetV2ConvLayer(cfg, in_channels=mid, out_channels=co, kernel_size=1, use_activation=False)
    return blk.eval()


@pytest.mark.parametrize("ci,co,s,e,hw", [(16, 24, 2, 4, 112), (24, 24, 1, 3, 56), (24, 40, 2, 3, 56), (40, 40, 1, 3, 28), (80, 80, 1, 2.5, 14),
                                          (112, 160, 2, 6, 14), (16, 24, 2, 4.5, 57), (24, 24, 1, 3.67, 28)])
def test_inverted_residual_block_equals_transformers_block(ci, co, s, e, hw):
    """oracle IR (the restatement of timm's InvertedResidual with tf_ 'same' padding, BN eps 1e-3, ReLU) == transformers' MobileNetV2InvertedResidual with
    tf_padding, the same weights and BatchNorm statistics: operator order, padding, BN placement and the residual rule agree"""
    g = torch.Generator().manual_seed(ci * 1000 + co * 10 + s)
    ours = M.IR(ci, co, s, e).eval()
    mid = ours.conv_pw.out_channels
    assert mid == M.make_divisible(ci * e)
    hf = _hf_block(ci, co, s, mid, M.BN_EPS)
    with torch.no_grad():
        for p in ours.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        for bn in (ours.bn1, ours.bn2, ours.bn3):
            bn.running_mean.copy_(torch.randn(bn.running_mean.shape, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(bn.running_var.shape, generator=g) + 0.5)
            bn.weight.copy_(torch.rand(bn.weight.shape, generator=g) + 0.5)
        for (conv, bn), layer in (((ours.conv_pw, ours.bn1), hf.expand_1x1), ((ours.conv_dw, ours.bn2), hf.conv_3x3), ((ours.conv_pwl, ours.bn3), hf.reduce_1x1)):
            layer.convolution.weight.copy_(conv.weight)
            assert layer.convolution.bias is None
            n = layer.normalization
            assert abs(n.eps - M.BN_EPS) < 1e-12
            n.weight.copy_(bn.weight); n.bias.copy_(bn.bias); n.running_mean.copy_(bn.running_mean); n.running_var.copy_(bn.running_var)
        x = torch.randn(2, ci, hw, hw, generator=g)
        got, want = ours(x), hf(x)
    assert ours.has_skip == hf.use_residual == (s == 1 and ci == co)
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=0, atol=1e-5), float((got - want).abs().max())


def test_make_divisible_matches_transformers():
    """timm's make_divisible (round_limit 0.9) against transformers' (TF-slim's rule: +divisor if below 0.9 v) on the widths the two backbones use"""
    for ci, e in [(16, 4), (24, 3), (24, 3), (40, 3), (40, 6), (80, 2.5), (80, 2.3), (80, 6), (112, 6), (160, 6), (16, 4.5), (24, 3.67), (24, 4), (40, 3), (48, 6), (96, 6)]:
        assert M.make_divisible(ci * e) == HF.make_divisible(int(ci * e) if float(ci * e).is_integer() else ci * e, 8, 8), (ci, e)
    assert [M.make_divisible(v) for v in (64, 72, 120, 240, 200, 184, 480, 672, 960, 88, 96, 144, 288, 576)] == \
           [64, 72, 120, 240, 200, 184, 480, 672, 960, 88, 96, 144, 288, 576]
