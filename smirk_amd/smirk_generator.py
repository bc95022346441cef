"""MI355X drop-in for the reference SmirkGenerator U-Net (src/smirk_generator.py:6-178).

The module tree only HOLDS the parameters, under exactly the reference's state_dict keys (encoder{1-4}.enc{i}conv{1,2}.weight,
...norm{1,2}.*, bottleneck.*, resnet_blocks.{k}.conv_block.{1,2,5,6}.*, upconv{1-4}.*, decoder{1-4}.*, conv.* — 178 keys).
forward() is ONE call of smirk_generator_forward (csrc/network.hip), which enqueues the whole layer schedule from C: NHWC activations in a
per-stream workspace, every 3x3 / transposed convolution on the MFMA implicit-GEMM or halo-patch kernels with BatchNorm(eval)+ReLU(+residual)
fused in the epilogue, reflection padding and the decoder's channel concat done as address arithmetic, 2x2 max-pool as a vectorised streaming
kernel, and the network tail (dec1conv2 + BN + ReLU + 1x1 conv + sigmoid) fused into one launch.

Two arithmetic modes (`SmirkGenerator.precision`, default from $SMIRK_AMD_GENERATOR_PRECISION or "f16x3"):
  "f16x3"  split-fp16: activations/weights carried as fp16 (hi, lo) pairs, 3 fp16 MFMAs per product with fp32 accumulation —
           fp32-class accuracy (same error vs fp64 as an fp32 GEMM) at the 16-bit matrix rate; CDNA4 has no TF32.
  "f32"    exact fp32 FMA chain on v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak).
Train mode (`.train()`): batch-statistics BatchNorm + a full backward pass, see smirk_amd/generator_train.py (BASELINE config 5, generator slice).
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
    """Drop-in for src/smirk_generator.py:SmirkGenerator (same constructor, state_dict keys and forward).  Limits of the gfx950 kernels, all reported by
    exceptions: init_features % 8 == 0 and out_channels <= 4 (constructor); input H, W multiples of 16 (the reference's own skip `torch.cat` fails otherwise);
    in_channels > 8 is served for gradient-free inference only (exact-fp32 kernels) — train mode and eval-mode input gradients need in_channels <= 8
    (SMIRK uses 6)."""

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
        if f % 8 or out_channels > 4:
            # measured (tools/gen_shape_probe.py): every init_features % 8 == 0 (8 ... 64) matches the oracle in both arithmetic modes, 12 / 20 do NOT in the
            # exact-fp32 mode either (round 3 accepted them there and returned wrong images) -> refused loudly
            raise L.SmirkHipError("gfx950 generator kernels need init_features % 8 == 0 and out_channels <= 4")
        self._packed, self._packed_key = None, None
        self._wstruct, self._wstruct_key = None, None
        self._ws = L.Workspace()
        self.precision = os.environ.get("SMIRK_AMD_GENERATOR_PRECISION", "f16x3")

    def train(self, mode=True):
        """nn.Module.train / .eval.  The eval-mode weight image folds BatchNorm's RUNNING statistics and is cached on (data_ptr, _version) of every
        parameter and buffer; HIP-graph replays of the train-mode path (smirk_amd.cycle.graph_cycle_modules) update the running statistics from captured
        kernels without bumping torch's version counters, so every train <-> eval transition drops the cache and the next eval forward re-folds."""
        if bool(mode) != self.training:
            self._packed = self._packed_key = self._wstruct = self._wstruct_key = None
        return super().train(mode)

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
        if split and self.in_channels > 8:
            # the reference accepts any in_channels: more than one 8-channel group at the input takes the exact-fp32 MFMA kernels instead of being refused
            # (same tolerances, ~2x slower; init_features % 8 is the remaining requirement, checked in __init__)
            if not getattr(self, "_warned_f32", False):
                import warnings
                warnings.warn("smirk_amd.SmirkGenerator: in_channels > 8 -> running the exact-fp32 MFMA kernels instead of the split-fp16 ('f16x3') mode",
                              stacklevel=3)
                self._warned_f32 = True
            split = False
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
            # a WEIGHT the fp16 `hi` half cannot carry is refused when the image is built, with the layer's name (the folded BatchNorm scale / shift stay fp32:
            # what they do to the activations is the device-side audit's business)
            for k, v in P.items():
                big = float(v[0].abs().max()) if v[0].numel() else 0.0
                if not big < 65504.0:
                    raise L.SmirkHipError(f"SmirkGenerator layer '{k}': |weight| = {big:.4g} does not fit the split-fp16 format (|x| < 65504); "
                                          "use precision = 'f32' for this checkpoint")
            P = {k: (_split16(v[0]),) + tuple(v[1:]) for k, v in P.items()}
        P["final"] = (self.conv.weight.detach().float().reshape(self.out_channels, self.features).contiguous(), None,
                      self.conv.bias.detach().float().contiguous())
        self._split = split
        self._packed, self._packed_key = P, key
        return P

    def _weights(self):
        """SmirkGeneratorWeights (host struct of device pointers into the packed tensors) for smirk_generator_forward; rebuilt with the pack."""
        P = self._pack()
        if self._wstruct is not None and self._wstruct_key == self._packed_key:
            return self._wstruct
        if len(self.resnet_blocks) > L.GEN_MAX_RES:
            raise L.SmirkHipError(f"at most {L.GEN_MAX_RES} ResNet blocks")
        w = L.SmirkGeneratorWeights()
        w.in_channels, w.out_channels, w.features, w.res_blocks = self.in_channels, self.out_channels, self.features, len(self.resnet_blocks)
        w.precision = L.PRECISION_F16X3 if self._split else L.PRECISION_F32
        w.cin_pad = self._cin_pad
        for l, tag in enumerate(("enc1", "enc2", "enc3", "enc4", "bottleneck")):
            w.enc[l][0], w.enc[l][1] = L.conv_layer(*P[tag + "1"]), L.conv_layer(*P[tag + "2"])
        for k in range(len(self.resnet_blocks)):
            w.res[k][0], w.res[k][1] = L.conv_layer(*P[f"res{k}a"]), L.conv_layer(*P[f"res{k}b"])
        for i, lvl in enumerate((4, 3, 2, 1)):
            w.up[i] = L.conv_layer(*P[f"up{lvl}"])
            w.dec[i][0], w.dec[i][1] = L.conv_layer(*P[f"dec{lvl}1"]), L.conv_layer(*P[f"dec{lvl}2"])
        w.final_w, w.final_b = P["final"][0].data_ptr(), P["final"][2].data_ptr()
        self._wstruct, self._wstruct_key = w, self._packed_key
        return w

    TAPS = ("enc1", "enc2", "enc3", "enc4", "bottleneck", "res", "dec4", "dec3", "dec2", "dec1")

    def _run(self, a, b, taps=None):
        """One call of smirk_generator_forward: cat(a, b) NCHW -> sigmoid image.  Every layer is enqueued from C on the current stream; the
        activations live in a per-stream workspace owned by this module."""
        L.raise_if_range_tripped("smirk_amd.SmirkGenerator.forward")      # an overflow of the split-fp16 format in an EARLIER call is reported now (no sync)
        if self.training:
            # train mode (smirk_trainer.py:349-355 calls self.train() before every step): batch-statistics BatchNorm with running-stat updates and a
            # real backward pass — one autograd.Function over the whole network (smirk_amd/generator_train.py, csrc/train.hip)
            if self.precision != "f16x3":
                raise L.SmirkHipError("train mode runs in the split-fp16 ('f16x3') arithmetic mode")
            if self.in_channels > 8:
                raise L.SmirkHipError("smirk_amd.SmirkGenerator: train mode / backward need in_channels <= 8 (one split16 channel group at the network input); "
                                      "in_channels > 8 is served for gradient-free inference only (exact-fp32 kernels)")
            from .generator_train import GeneratorTrainFunction
            x = a if b is None else torch.cat([a, b], 1)
            return GeneratorTrainFunction.apply(self, x, *self.parameters())
        srcs = [a] + ([] if b is None else [b])
        wants_input_grad = taps is None and torch.is_grad_enabled() and any(t.requires_grad for t in srcs)
        if wants_input_grad and (self.precision != "f16x3" or self.in_channels > 8):
            # decided on the EFFECTIVE arithmetic (in_channels > 8 runs the exact-fp32 kernels even when precision == "f16x3"): say so now instead of failing
            # inside the autograd function or handing back a tensor whose backward raises later
            raise L.SmirkHipError("smirk_amd.SmirkGenerator: a gradient with respect to the input is available in the split-fp16 ('f16x3') mode with "
                                  "in_channels <= 8 only (this module runs the exact-fp32 inference kernels: precision=%r, in_channels=%d)"
                                  % (self.precision, self.in_channels))
        if wants_input_grad:
            # eval mode with an INPUT that requires grad — smirk_trainer.py:108-113 freezes the generator, calls .eval() and back-propagates the emotion
            # loss through it into rendered_img: the same autograd.Function as train mode with BatchNorm taken from the running statistics.  (Forward-only
            # callers whose parameters merely have requires_grad=True, like demo.py without no_grad, stay on the whole-network C entry below.)
            from .generator_train import GeneratorTrainFunction
            x = a if b is None else torch.cat([a, b], 1)
            return GeneratorTrainFunction.apply(self, x, *self.parameters())
        a = L.as_f32c(a.detach())
        b = None if b is None else L.as_f32c(b.detach())
        B, Ca, H, W = a.shape
        Cb = 0 if b is None else b.shape[1]
        if Ca + Cb != self.in_channels:
            raise L.SmirkHipError(f"expected {self.in_channels} input channels, got {Ca + Cb}")
        if H % 16 or W % 16:
            # not a narrowing of the drop-in: the reference fails on such sizes too (4 x MaxPool2d(2) floors, 4 x ConvTranspose2d doubles, and torch.cat of
            # the skip tensor with a different size raises, smirk_generator.py:65-75)
            raise L.SmirkHipError("input size must be a multiple of 16 (4 pooling levels; the reference's torch.cat of the skip tensors fails otherwise too)")
        lib, w = L.lib(), self._weights()
        if not self._split and b is not None and not (Ca == 3 and Cb == 3 and self._cin_pad == 8):
            a, b, Ca, Cb = torch.cat([a, b], 1), None, Ca + Cb, 0            # exact-fp32 mode packs two sources only for the 3 + 3 case
        dev = a.device
        out = torch.empty(B, self.out_channels, H, W, device=dev)
        tp = None
        audit = self._split and os.environ.get("SMIRK_F16X3_RANGE_CHECK")
        if audit and taps is None:
            taps = {}
        if taps is not None:
            f = self.features
            shapes = [(H >> l, W >> l, f << l) for l in range(5)] + [(H >> 4, W >> 4, f << 4)] + [(H >> l, W >> l, f << l) for l in (3, 2, 1, 0)]
            bufs = [torch.empty(B, h, wd, c, device=dev) for h, wd, c in shapes]
            tp = (L._p * 10)(*[t.data_ptr() for t in bufs])
            taps.update(dict(zip(self.TAPS, bufs)))
        nws = lib.smirk_generator_workspace_bytes(w, B, H, W)
        ws = self._ws.get(nws, dev)
        L.check(lib.smirk_generator_forward(w, L.ptr(a), Ca, L.ptr(b, allow_none=True), Cb, L.ptr(out), B, H, W, tp, L.ptr(ws, torch.uint8), nws,
                                            L.stream_ptr()))
        if audit:
            self._range_audit(taps, float(audit) if audit not in ("1", "true") else 6.0e4)
        # the reference back-propagates through the generator (smirk_trainer.py:94-104); this forward has no backward yet, so make any
        # attempt to do so fail loudly instead of handing the caller zero / missing gradients
        return L.loud_cut("smirk_amd.SmirkGenerator.forward", out, srcs + list(self.parameters()))

    def _range_audit(self, taps, limit):
        """$SMIRK_F16X3_RANGE_CHECK=1 (or a limit): the split-fp16 format saturates at |x| = 65504 — BatchNorm'd activations of a trained network are
        O(1..100), but a badly scaled checkpoint would turn into inf / NaN silently.  Audits the ten block outputs of this forward (taps disable the
        fused tail, so this is a debugging mode) and raises naming the first offending block."""
        lib, st = L.lib(), L.stream_ptr()
        counts = torch.zeros(len(self.TAPS), 2, dtype=torch.int32, device=next(iter(taps.values())).device)
        for i, k in enumerate(self.TAPS):
            t = taps[k]
            L.check(lib.smirk_split16_range_check(L.ptr(t), t.numel(), limit, L.ptr(counts[i], torch.int32), st))
        c = counts.cpu()
        for i, k in enumerate(self.TAPS):
            if int(c[i, 0]):
                mx = float(c[i, 1:2].view(torch.float32))
                raise L.SmirkHipError(f"f16x3 range check: {int(c[i, 0])} activations of block '{k}' are non-finite or >= {limit:g} in magnitude "
                                      f"(largest finite |x| = {mx:.4g}); the split-fp16 mode cannot carry them — use precision = 'f32'")

    def forward_pair(self, rendered, masked, _taps=None):
        """generator(torch.cat([rendered, masked], 1)) without materialising the concatenation (smirk_trainer.py:94, demo.py:167)."""
        return self._run(rendered, masked, _taps)

    def forward(self, x, _taps=None):
        """x [B,in_channels,H,W] (NCHW, like the reference: cat[rendered_img, masked_img]) -> sigmoid image [B,out,H,W]."""
        return self._run(x, None, _taps)
