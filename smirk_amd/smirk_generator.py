"""MI355X drop-in for the reference SmirkGenerator U-Net (src/smirk_generator.py:6-178).

The module tree only HOLDS the parameters, under exactly the reference's state_dict keys (encoder{1-4}.enc{i}conv{1,2}.weight,
...norm{1,2}.*, bottleneck.*, resnet_blocks.{k}.conv_block.{1,2,5,6}.*, upconv{1-4}.*, decoder{1-4}.*, conv.* — 178 keys).
forward() runs on libsmirk_hip.so: NHWC activations, every 3x3 / transposed convolution on the MFMA implicit-GEMM kernel with
BatchNorm(eval)+ReLU(+residual) fused in the epilogue, reflection padding and the decoder's channel concat done as address
arithmetic, 2x2 max-pool and the final 1x1+sigmoid as vectorised streaming kernels.

Two arithmetic modes (`SmirkGenerator.precision`, default from $SMIRK_AMD_GENERATOR_PRECISION or "f16x3"):
  "f16x3"  split-fp16: activations/weights carried as fp16 (hi, lo) pairs, 3 fp16 MFMAs per product with fp32 accumulation —
           fp32-class accuracy (same error vs fp64 as an fp32 GEMM) at the 16-bit matrix rate; CDNA4 has no TF32.
  "f32"    exact fp32 FMA chain on v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak).
Eval mode only this round (the cycle-path training step is BASELINE config 5 / SURVEY.md §7 step 7).
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L


def _double_conv(cin, cout, tag):
    return nn.Sequential(OrderedDict([
        (tag + "conv1", nn.Conv2d(cin, cout, 3, padding=1, bias=False)), (tag + "norm1", nn.BatchNorm2d(cout)),
        (tag + "relu1", nn.ReLU(inplace=True)),
        (tag + "conv2", nn.Conv2d(cout, cout, 3, padding=1, bias=False)), (tag + "norm2", nn.BatchNorm2d(cout)),
        (tag + "relu2", nn.ReLU(inplace=True))]))


class ResnetBlock(nn.Module):
    """Parameter holder for one reflect-padded residual block (smirk_generator.py:121-178): conv_block.{1,2,5,6}."""

    def __init__(self, dim, padding_type='reflect', norm_layer=nn.BatchNorm2d, use_dropout=False, use_bias=False):
        super().__init__()
        if padding_type != 'reflect' or use_dropout:
            raise NotImplementedError("only the configuration SMIRK instantiates (reflect padding, no dropout)")
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, padding=0, bias=use_bias), norm_layer(dim), nn.ReLU(True),
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, padding=0, bias=use_bias), norm_layer(dim))


def _bn_affine(bn):
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    return scale.contiguous(), (bn.bias.detach().float() - bn.running_mean.detach().float() * scale).contiguous()


def _split16(w):
    """fp32 [N][K] (K % 8 == 0) -> split-fp16 [N][K/8][2][8] halves, returned as a float32-typed buffer of the same byte size."""
    n, k = w.shape
    g = w.reshape(n, k // 8, 8)
    hi = g.half()
    lo = ((g - hi.float()) * 2048.0).half()
    return torch.stack([hi, lo], 2).contiguous().view(torch.float32).reshape(n, k)


def split16_to_float(t):
    """split16 activation [B,H,W,C] (float32-typed storage) -> fp32 NHWC values (test / debugging helper)."""
    B, H, W, C = t.shape
    h = t.view(torch.float16).reshape(B, H, W, C // 8, 2, 8).float()
    return (h[..., 0, :] + h[..., 1, :] / 2048.0).reshape(B, H, W, C)


def _pack3x3(w, cin_pad=None):
    """[Cout,Cin,3,3] -> [Cout][(ky,kx,c)] with optional zero channel padding."""
    w = w.detach().float()
    if cin_pad is not None and cin_pad > w.shape[1]:
        w = torch.cat([w, w.new_zeros(w.shape[0], cin_pad - w.shape[1], 3, 3)], 1)
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


class SmirkGenerator(nn.Module):
    def __init__(self, in_channels=3, out_channels=1, init_features=16, res_blocks=3):
        super().__init__()
        f = init_features
        self.encoder1 = _double_conv(in_channels, f, "enc1"); self.pool1 = nn.MaxPool2d(2, 2)
        self.encoder2 = _double_conv(f, f * 2, "enc2"); self.pool2 = nn.MaxPool2d(2, 2)
        self.encoder3 = _double_conv(f * 2, f * 4, "enc3"); self.pool3 = nn.MaxPool2d(2, 2)
        self.encoder4 = _double_conv(f * 4, f * 8, "enc4"); self.pool4 = nn.MaxPool2d(2, 2)
        self.bottleneck = _double_conv(f * 8, f * 16, "bottleneck")
        self.resnet_blocks = nn.ModuleList([ResnetBlock(f * 16) for _ in range(res_blocks)])
        self.upconv4 = nn.ConvTranspose2d(f * 16, f * 8, 2, 2); self.decoder4 = _double_conv(f * 16, f * 8, "dec4")
        self.upconv3 = nn.ConvTranspose2d(f * 8, f * 4, 2, 2); self.decoder3 = _double_conv(f * 8, f * 4, "dec3")
        self.upconv2 = nn.ConvTranspose2d(f * 4, f * 2, 2, 2); self.decoder2 = _double_conv(f * 4, f * 2, "dec2")
        self.upconv1 = nn.ConvTranspose2d(f * 2, f, 2, 2); self.decoder1 = _double_conv(f * 2, f, "dec1")
        self.conv = nn.Conv2d(f, out_channels, 1)
        self.in_channels, self.out_channels, self.features = in_channels, out_channels, f
        if f % 4 or out_channels > 4:
            raise L.SmirkHipError("gfx950 generator kernels need init_features % 4 == 0 and out_channels <= 4")
        self._packed, self._packed_key = None, None
        self.precision = os.environ.get("SMIRK_AMD_GENERATOR_PRECISION", "f16x3")

    # ---- weight packing (device-side, cached until a parameter changes) ---------------------------------------------------
    def _key(self):
        return (self.precision,) + tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def _pack(self):
        key = self._key()
        if self._packed is not None and self._packed_key == key:
            return self._packed
        if self.precision not in ("f16x3", "f32"):
            raise L.SmirkHipError(f"unknown SmirkGenerator.precision {self.precision!r}")
        split = self.precision == "f16x3"
        if split and (self.features % 8 or self.in_channels > 8):
            raise L.SmirkHipError("f16x3 mode needs init_features % 8 == 0 and in_channels <= 8")
        P = {}
        # 6 -> 8 input channels: keeps every im2col access a 16-byte vector / one split-fp16 group
        self._cin_pad = 8 if split else (self.in_channels + 3) // 4 * 4

        def block(seq, tag, first_pad=None):
            m = dict(seq.named_children())
            P[tag + "1"] = (_pack3x3(m[tag + "conv1"].weight, first_pad),) + _bn_affine(m[tag + "norm1"])
            P[tag + "2"] = (_pack3x3(m[tag + "conv2"].weight),) + _bn_affine(m[tag + "norm2"])

        block(self.encoder1, "enc1", self._cin_pad); block(self.encoder2, "enc2"); block(self.encoder3, "enc3")
        block(self.encoder4, "enc4"); block(self.bottleneck, "bottleneck")
        for k, rb in enumerate(self.resnet_blocks):
            cb = rb.conv_block
            P[f"res{k}a"] = (_pack3x3(cb[1].weight),) + _bn_affine(cb[2])
            P[f"res{k}b"] = (_pack3x3(cb[5].weight),) + _bn_affine(cb[6])
        for lvl in (4, 3, 2, 1):
            up = getattr(self, f"upconv{lvl}")
            w = up.weight.detach().float()                                   # [Cin, Cout, 2, 2]
            P[f"up{lvl}"] = (w.permute(2, 3, 1, 0).reshape(4 * w.shape[1], w.shape[0]).contiguous(), None,
                             up.bias.detach().float().contiguous())
            block(getattr(self, f"decoder{lvl}"), f"dec{lvl}")
        if split:
            P = {k: (_split16(v[0]),) + tuple(v[1:]) for k, v in P.items()}
        P["final"] = (self.conv.weight.detach().float().reshape(self.out_channels, self.features).contiguous(), None,
                      self.conv.bias.detach().float().contiguous())
        self._split = split
        self._packed, self._packed_key = P, key
        return P

    # ---- kernel calls ---------------------------------------------------------------------------------------------
    def _conv(self, lib, st, x0, x1, pk, B, H, W, cout, k=3, reflect=False, relu=True, residual=None, convt=False):
        w, scale, shift = pk
        d = L.SmirkConvDesc()
        d.B, d.H, d.W = B, H, W
        d.C0, d.C1 = x0.shape[-1], (x1.shape[-1] if x1 is not None else 0)
        d.Cout, d.KH, d.KW, d.stride = cout, k, k, 1
        d.pad_t = d.pad_l = (k - 1) // 2
        d.Ho, d.Wo = H, W
        d.pad_mode = L.PAD_REFLECT if reflect else L.PAD_ZERO
        d.act = L.ACT_RELU if relu else L.ACT_NONE
        d.out_mode = L.OUT_CONVT2X2 if convt else L.OUT_NHWC
        out = torch.empty((B, 2 * H, 2 * W, cout) if convt else (B, H, W, cout), device=x0.device)
        P = L.ptr
        n_gemm = 4 * cout if convt else cout
        flops = 2.0 * B * H * W * n_gemm * (k * k * (d.C0 + d.C1))
        fn = lib.smirk_conv_igemm_f16x3 if self._split else lib.smirk_conv_igemm_f32
        L.timed(L.igemm_kernel_name(n_gemm, self._split, d.C0, d.C1, k), flops, lambda: L.check(fn(
            d, P(x0), P(x1, allow_none=True), P(w), P(scale, allow_none=True), P(shift, allow_none=True),
            P(residual, allow_none=True), P(out), st)))
        return out

    def forward_nhwc(self, x_nhwc, taps=None):
        """x_nhwc [B,H,W,Cpad] (fp32 NHWC in "f32" mode, split16 in "f16x3" mode; channels >= in_channels zero)
        -> [B,out_channels,H,W] fp32 sigmoid image.  `taps` collects intermediate activations in the mode's storage format."""
        if self.training:
            raise NotImplementedError("smirk_amd.SmirkGenerator implements the eval-mode forward only (call .eval())")
        lib, st, P = L.lib(), L.stream_ptr(), self._pack()
        B, H, W, _ = x_nhwc.shape
        if H % 16 or W % 16:
            raise L.SmirkHipError("input size must be a multiple of 16 (4 pooling levels)")
        f = self.features
        t = taps if taps is not None else {}

        def dconv(x0, x1, tag, h, w, c):
            y = self._conv(lib, st, x0, x1, P[tag + "1"], B, h, w, c)
            return self._conv(lib, st, y, None, P[tag + "2"], B, h, w, c)

        def pool(x, h, w, c):
            o = torch.empty(B, h // 2, w // 2, c, device=x.device)
            L.check((lib.smirk_maxpool2x2_split16 if self._split else lib.smirk_maxpool2x2_nhwc)(L.ptr(x), L.ptr(o), B, h, w, c, st))
            return o

        e1 = dconv(x_nhwc, None, "enc1", H, W, f); t["enc1"] = e1
        e2 = dconv(pool(e1, H, W, f), None, "enc2", H // 2, W // 2, 2 * f); t["enc2"] = e2
        e3 = dconv(pool(e2, H // 2, W // 2, 2 * f), None, "enc3", H // 4, W // 4, 4 * f); t["enc3"] = e3
        e4 = dconv(pool(e3, H // 4, W // 4, 4 * f), None, "enc4", H // 8, W // 8, 8 * f); t["enc4"] = e4
        h16, w16 = H // 16, W // 16
        b = dconv(pool(e4, H // 8, W // 8, 8 * f), None, "bottleneck", h16, w16, 16 * f); t["bottleneck"] = b
        for k in range(len(self.resnet_blocks)):
            y = self._conv(lib, st, b, None, P[f"res{k}a"], B, h16, w16, 16 * f, reflect=True, relu=True)
            b = self._conv(lib, st, y, None, P[f"res{k}b"], B, h16, w16, 16 * f, reflect=True, relu=False, residual=b)
        t["res"] = b
        d = b
        out = torch.empty(B, self.out_channels, H, W, device=b.device)
        wf, _, bf = P["final"]
        for lvl, skip, div, c in ((4, e4, 16, 8 * f), (3, e3, 8, 4 * f), (2, e2, 4, 2 * f), (1, e1, 2, f)):
            up = self._conv(lib, st, d, None, P[f"up{lvl}"], B, H // div, W // div, c, k=1, relu=False, convt=True)
            if lvl == 1 and self._split and taps is None and f == 32 and H % 16 == 0 and W % 16 == 0 and H >= 64:
                # network tail fused: dec1conv2 + BN + ReLU + final 1x1 conv + sigmoid in one launch, dec1 never written to HBM
                y = self._conv(lib, st, up, skip, P["dec11"], B, H, W, f)
                w2, sc2, sh2 = P["dec12"]
                dd = L.SmirkConvDesc()
                dd.B, dd.H, dd.W, dd.C0, dd.C1, dd.Cout, dd.KH, dd.KW, dd.stride = B, H, W, f, 0, f, 3, 3, 1
                dd.pad_t = dd.pad_l = 1
                dd.Ho, dd.Wo, dd.pad_mode, dd.act, dd.out_mode = H, W, L.PAD_ZERO, L.ACT_RELU, L.OUT_NHWC
                Pp = L.ptr
                flops = 2.0 * B * H * W * f * 9 * f
                L.timed("conv3x3_patch_kernel", flops, lambda: L.check(lib.smirk_conv3x3_tail_f16x3(
                    dd, Pp(y), None, Pp(w2), Pp(sc2), Pp(sh2), Pp(wf), Pp(bf), Pp(out), self.out_channels, st)))
                return out
            d = dconv(up, skip, f"dec{lvl}", 2 * H // div, 2 * W // div, c)
            t[f"dec{lvl}"] = d
        fin = lib.smirk_conv1x1_sigmoid_nchw_split16 if self._split else lib.smirk_conv1x1_sigmoid_nchw
        L.check(fin(L.ptr(d), L.ptr(wf), L.ptr(bf), L.ptr(out), B, H, W, f, self.out_channels, st))
        return out

    def pack_input(self, rendered, masked):
        """Fused torch.cat([rendered, masked], 1) + NCHW->NHWC (smirk_trainer.py:94, demo.py:167) for 3+3 channel inputs."""
        rendered, masked = L.as_f32c(rendered), L.as_f32c(masked)
        B, _, H, W = rendered.shape
        self._pack()
        if self._cin_pad != 8 or rendered.shape[1] != 3 or masked.shape[1] != 3:
            raise L.SmirkHipError("pack_input is the 3+3 channel case (SmirkGenerator(in_channels=6, ...))")
        x = torch.empty(B, H, W, 8, device=rendered.device)
        if self._split:
            L.check(L.lib().smirk_pack_generator_input_split16(L.ptr(rendered), 3, L.ptr(masked), 3, L.ptr(x), B, H, W, L.stream_ptr()))
        else:
            L.check(L.lib().smirk_pack_generator_input(L.ptr(rendered), L.ptr(masked), L.ptr(x), B, H, W, L.stream_ptr()))
        return x

    def forward(self, x, _taps=None):
        """x [B,in_channels,H,W] (NCHW, like the reference: cat[rendered_img, masked_img]) -> sigmoid image [B,out,H,W]."""
        x = L.as_f32c(x)
        B, C, H, W = x.shape
        if C != self.in_channels:
            raise L.SmirkHipError(f"expected {self.in_channels} input channels, got {C}")
        self._pack()
        xn = torch.empty(B, H, W, self._cin_pad, device=x.device)
        if self._split:
            L.check(L.lib().smirk_pack_generator_input_split16(L.ptr(x), C, None, 0, L.ptr(xn), B, H, W, L.stream_ptr()))
        else:
            L.check(L.lib().smirk_nchw_to_nhwc_pad(L.ptr(x), L.ptr(xn), B, C, H, W, self._cin_pad, L.stream_ptr()))
        return self.forward_nhwc(xn, _taps)
