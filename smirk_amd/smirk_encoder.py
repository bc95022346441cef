"""MI355X drop-in for the reference SmirkEncoder (src/smirk_encoder.py:14-133) and the timm backbones it instantiates
(timm==0.9.16 `tf_mobilenetv3_{small,large}_minimal_100`, features_only=True — smirk_encoder.py:7-12; timm is NOT a dependency
of this package: the architecture is decoded here from the published arch strings).

The nn.Module tree only HOLDS parameters under timm's state_dict names (`<enc>.encoder.conv_stem.weight`, `.bn1.*`,
`.blocks.{stage}.{i}.conv_pw|conv_dw|conv_pwl|conv.weight`, `.bn{1,2,3}.*`, heads `pose_cam_layers.0`, `shape_layers.0`,
`expression_layers.0`) so the reference's strict checkpoint load (demo.py:56-58) works.  forward() runs on libsmirk_hip.so:
each sub-encoder is ONE call of smirk_backbone_forward (csrc/network.hip) that enqueues stem / fused MBConv blocks / depthwise /
pointwise (MFMA implicit GEMM with BatchNorm(eval) + ReLU + residual fused) / pooled linear head from C.  Eval mode only this round.
"""
import os

import torch
import torch.nn as nn

from . import _lib as L

BN_EPS = 1e-3   # "tf_" models
# arithmetic of the pointwise convolutions: "f16x3" = split-fp16 MFMA (fp32-class accuracy at the 16-bit matrix rate, activations kept in
# the split16 format through the whole backbone), "f32" = exact fp32 MFMA.  See smirk_generator.py / conv.hip.
PRECISION = os.environ.get("SMIRK_AMD_ENCODER_PRECISION", "f16x3")
_STREAMS = {}   # per-device side streams of SmirkEncoder.forward (kept off the module so copy.deepcopy(encoder) keeps working)

_ARCH = {
    # (block type, repeats, stride, expansion, out channels); every kernel is 3x3 (1x1 for 'cn'), ReLU, no SE ("minimal")
    "tf_mobilenetv3_large_minimal_100": [
        [("ds", 1, 1, 1.0, 16)],
        [("ir", 1, 2, 4.0, 24), ("ir", 1, 1, 3.0, 24)],
        [("ir", 3, 2, 3.0, 40)],
        [("ir", 1, 2, 6.0, 80), ("ir", 1, 1, 2.5, 80), ("ir", 2, 1, 2.3, 80)],
        [("ir", 2, 1, 6.0, 112)],
        [("ir", 3, 2, 6.0, 160)],
        [("cn", 1, 1, 1.0, 960)],
    ],
    "tf_mobilenetv3_small_minimal_100": [
        [("ds", 1, 2, 1.0, 16)],
        [("ir", 1, 2, 4.5, 24), ("ir", 1, 1, 3.67, 24)],
        [("ir", 1, 2, 4.0, 40), ("ir", 2, 1, 6.0, 40)],
        [("ir", 2, 1, 3.0, 48)],
        [("ir", 3, 2, 6.0, 96)],
        [("cn", 1, 1, 1.0, 576)],
    ],
}


def _round_channels(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    return new_v + divisor if new_v < 0.9 * v else new_v


def _bn(c):
    return nn.BatchNorm2d(c, eps=BN_EPS)


class _DS(nn.Module):          # DepthwiseSeparableConv holder
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv_dw = nn.Conv2d(cin, cin, 3, stride, 1, groups=cin, bias=False); self.bn1 = _bn(cin)
        self.conv_pw = nn.Conv2d(cin, cout, 1, bias=False); self.bn2 = _bn(cout)
        self.kind, self.stride, self.skip = "ds", stride, (stride == 1 and cin == cout)


class _IR(nn.Module):          # InvertedResidual holder
    def __init__(self, cin, cout, stride, exp):
        super().__init__()
        mid = _round_channels(cin * exp)
        self.conv_pw = nn.Conv2d(cin, mid, 1, bias=False); self.bn1 = _bn(mid)
        self.conv_dw = nn.Conv2d(mid, mid, 3, stride, 1, groups=mid, bias=False); self.bn2 = _bn(mid)
        self.conv_pwl = nn.Conv2d(mid, cout, 1, bias=False); self.bn3 = _bn(cout)
        self.kind, self.stride, self.skip = "ir", stride, (stride == 1 and cin == cout)


class _CN(nn.Module):          # ConvBnAct holder
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, bias=False); self.bn1 = _bn(cout)
        self.kind = "cn"


class MobileNetV3Features(nn.Module):
    """Parameter holder + HIP forward for one backbone; returns the last feature map, NHWC [B,h,w,C]."""

    def __init__(self, name):
        super().__init__()
        self.conv_stem = nn.Conv2d(3, 16, 3, 2, bias=False)
        self.bn1 = _bn(16)
        stages, cin, chans = [], 16, []
        for st in _ARCH[name]:
            blocks = []
            for kind, rep, stride, exp, cout in st:
                for i in range(rep):
                    s = stride if i == 0 else 1
                    blocks.append(_DS(cin, cout, s) if kind == "ds" else _IR(cin, cout, s, exp) if kind == "ir" else _CN(cin, cout))
                    cin = cout
            stages.append(nn.Sequential(*blocks))
            chans.append(cin)
        self.blocks = nn.Sequential(*stages)
        self.feature_info = [dict(num_chs=cin)]      # only [-1]['num_chs'] is consulted (smirk_encoder.py:11)
        self.num_features = cin
        self._packed, self._packed_key = None, None
        self._split = False
        self._wcache = {}
        self._ws = L.Workspace()

    def train(self, mode=True):
        """see SmirkGenerator.train: train <-> eval transitions drop the folded-BatchNorm weight cache (graph replays update running statistics without
        bumping torch's version counters)"""
        if bool(mode) != self.training:
            self._packed, self._packed_key = None, None
            self._wcache = {}
        return super().train(mode)

    def _key(self):
        return (PRECISION,) + tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    @staticmethod
    def _affine(bn):
        s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        return s.contiguous(), (bn.bias.detach().float() - bn.running_mean.detach().float() * s).contiguous()

    def _pack(self):
        key = self._key()
        if self._packed is not None and self._packed_key == key:
            return self._packed
        from .smirk_generator import _split16
        split = PRECISION == "f16x3"
        self._split = split
        pw32 = lambda conv: conv.weight.detach().float().reshape(conv.out_channels, conv.in_channels).contiguous()

        def pw16(conv):                                          # a weight the fp16 `hi` half cannot carry is refused when the image is built
            w = pw32(conv)
            big = float(w.abs().max()) if w.numel() else 0.0
            if not big < 65504.0:
                raise L.SmirkHipError(f"MobileNetV3 pointwise weight with |w| = {big:.4g} does not fit the split-fp16 format (|x| < 65504)")
            return _split16(w)
        pw = pw16 if split else pw32
        dw = lambda conv: conv.weight.detach().float().reshape(conv.out_channels, 9).t().contiguous()      # [9][C]
        P = {"stem": (self.conv_stem.weight.detach().float().permute(0, 2, 3, 1).reshape(16, 27).contiguous(),) + self._affine(self.bn1)}
        for si, st in enumerate(self.blocks):
            for bi, blk in enumerate(st):
                k = (si, bi)
                if blk.kind == "ds":
                    P[k] = dict(dw=(dw(blk.conv_dw),) + self._affine(blk.bn1), pw=(pw(blk.conv_pw),) + self._affine(blk.bn2))
                elif blk.kind == "ir":
                    P[k] = dict(pw=(pw(blk.conv_pw),) + self._affine(blk.bn1), dw=(dw(blk.conv_dw),) + self._affine(blk.bn2),
                                pwl=(pw(blk.conv_pwl),) + self._affine(blk.bn3))
                else:
                    P[k] = dict(pw=(pw(blk.conv),) + self._affine(blk.bn1))
        self._packed, self._packed_key = P, key
        return P

    def _weights(self, head=None, clamp_n_exp=-1):
        """SmirkBackboneWeights for smirk_backbone_forward (host struct of device pointers; cached per parameter version and head)."""
        P = self._pack()
        hkey = None if head is None else (head.weight.data_ptr(), head.weight._version, head.bias.data_ptr(), head.bias._version)
        ck = (self._packed_key, hkey, clamp_n_exp)
        hit = self._wcache.get("w")
        if hit is not None and hit[0] == ck:
            return hit[1]
        w = L.SmirkBackboneWeights()
        w.precision = L.PRECISION_F16X3 if self._split else L.PRECISION_F32
        w.clamp_n_exp, w.stem_cout, w.feat_ch = clamp_n_exp, 16, self.num_features
        w.stem = L.conv_layer(*P["stem"])
        n = 0
        for si, st in enumerate(self.blocks):
            for bi, blk in enumerate(st):
                if n >= L.BACKBONE_MAX_BLOCKS:
                    raise L.SmirkHipError("backbone has more blocks than SMIRK_BACKBONE_MAX_BLOCKS")
                pk, b = P[(si, bi)], w.blocks[n]
                b.stride = getattr(blk, "stride", 1)
                b.skip = int(bool(getattr(blk, "skip", False)))
                if blk.kind == "ds":
                    b.kind, b.cin, b.mid, b.cout = 0, blk.conv_dw.in_channels, blk.conv_dw.in_channels, blk.conv_pw.out_channels
                    b.dw, b.pw = L.conv_layer(*pk["dw"]), L.conv_layer(*pk["pw"])
                elif blk.kind == "ir":
                    b.kind, b.cin, b.mid, b.cout = 1, blk.conv_pw.in_channels, blk.conv_pw.out_channels, blk.conv_pwl.out_channels
                    b.pw, b.dw, b.pwl = L.conv_layer(*pk["pw"]), L.conv_layer(*pk["dw"]), L.conv_layer(*pk["pwl"])
                else:
                    b.kind, b.cin, b.mid, b.cout = 2, blk.conv.in_channels, blk.conv.in_channels, blk.conv.out_channels
                    b.pw = L.conv_layer(*pk["pw"])
                n += 1
        w.n_blocks = n
        keep = None
        if head is not None:
            hw, hb = head.weight.detach().float().contiguous(), head.bias.detach().float().contiguous()
            w.n_out, w.head_w, w.head_b = hw.shape[0], hw.data_ptr(), hb.data_ptr()
            keep = (hw, hb)
        self._wcache["w"] = (ck, w, keep)
        return w

    def run(self, img, head=None, clamp_n_exp=-1, want_features=False):
        """ONE call of smirk_backbone_forward (csrc/network.hip): stem -> blocks -> (global average pool -> Linear `head` (+ clamps)).
        Returns (head output [B, n_out] or None, last feature map NHWC or None)."""
        if self.training:
            # batch-statistics BatchNorm + autograd (smirk_trainer.py:349-355 calls self.train() before every step): csrc/train*.hip
            if head is None or want_features:
                raise L.SmirkHipError("train-mode backbone runs together with its Linear head (the encoders' own forward); use .eval() for bare features")
            from .encoder_train import BackboneTrainFunction
            out = BackboneTrainFunction.apply(self, head, clamp_n_exp, img, *(list(self.parameters()) + list(head.parameters())))
            return out, None
        src = img
        img = L.as_f32c(img.detach())
        if not img.is_cuda:
            raise L.SmirkHipError("smirk_amd runs on the MI355X HIP device only: got a CPU tensor (no CPU fallback exists)")
        B, _, H, W = img.shape
        lib, w = L.lib(), self._weights(head, clamp_n_exp)
        dev = img.device
        out = torch.empty(B, w.n_out, device=dev) if head is not None else None
        feat = torch.empty(B, (H + 31) // 32, (W + 31) // 32, self.num_features, device=dev) if want_features else None
        nws = lib.smirk_backbone_workspace_bytes(w, B, H, W)
        if nws == 0:
            raise L.SmirkHipError("unsupported backbone configuration / image size (H, W >= 32)")
        ws = self._ws.get(nws, dev)
        L.check(lib.smirk_backbone_forward(w, L.ptr(img), B, H, W, L.ptr(out, allow_none=True), L.ptr(feat, allow_none=True),
                                           L.ptr(ws, torch.uint8), nws, L.stream_ptr()))
        deps = [src] + list(self.parameters()) + ([] if head is None else list(head.parameters()))
        if out is not None:
            out = L.loud_cut("smirk_amd encoder forward", out, deps)
        if feat is not None:
            feat = L.loud_cut("smirk_amd encoder forward", feat, deps)
        return out, feat

    def _pointwise(self, lib, st, x, pk, relu, residual=None):
        w, sc, sh = pk
        B, H, W, C = x.shape
        d = L.SmirkConvDesc()
        d.B, d.H, d.W, d.C0, d.C1, d.Cout = B, H, W, C, 0, w.shape[0]
        d.KH = d.KW = d.stride = 1
        d.pad_t = d.pad_l = 0
        d.Ho, d.Wo, d.pad_mode = H, W, L.PAD_ZERO
        d.act, d.out_mode = (L.ACT_RELU if relu else L.ACT_NONE), L.OUT_NHWC
        out = torch.empty(B, H, W, w.shape[0], device=x.device)
        P = L.ptr
        fn = lib.smirk_conv_igemm_f16x3 if self._split else lib.smirk_conv_igemm_f32
        L.check(fn(d, P(x), None, P(w), P(sc), P(sh), P(residual, allow_none=True), P(out), st))
        return out

    def _depthwise(self, lib, st, x, pk, stride):
        w, sc, sh = pk
        B, H, W, C = x.shape
        out = torch.empty(B, (H + stride - 1) // stride, (W + stride - 1) // stride, C, device=x.device)
        P = L.ptr
        L.check((lib.smirk_dwconv3x3_split16 if self._split else lib.smirk_dwconv3x3)(P(x), P(w), P(sc), P(sh), P(out), B, H, W, C, stride, 1, st))
        return out

    @staticmethod
    def _fusable(lib, cin, pk, blk):
        if blk.kind == "ds":
            return False        # measured: no expanded tensor to keep on chip, the two streaming kernels are as fast (s1) or faster (s2)
        return bool(lib.smirk_mbconv_supported(cin, pk["dw"][0].shape[1], pk["pw" if blk.kind == "ds" else "pwl"][0].shape[0], blk.stride))

    def _fused_block(self, lib, st, x, pk, blk):
        """whole DS / IR block in one launch (csrc/mbconv.hip): the expanded activations stay in LDS"""
        B, H, W, C = x.shape
        P, N = L.ptr, (lambda t: L.ptr(t, allow_none=True))
        if blk.kind == "ds":
            (wd, s2, b2), (wp, s3, b3) = pk["dw"], pk["pw"]
            we = s1 = b1 = None
        else:
            (we, s1, b1), (wd, s2, b2), (wp, s3, b3) = pk["pw"], pk["dw"], pk["pwl"]
        mid, cout, s = wd.shape[1], wp.shape[0], blk.stride
        out = torch.empty(B, (H + s - 1) // s, (W + s - 1) // s, cout, device=x.device)
        L.check(lib.smirk_mbconv_fused_split16(P(x), N(we), N(s1), N(b1), P(wd), P(s2), P(b2), P(wp), P(s3), P(b3), int(bool(blk.skip)),
                                               P(out), B, H, W, C, mid, cout, s, st))
        return out

    def forward(self, img, _taps=None):
        """img [B,3,H,W] NCHW in [0,1] -> last feature map NHWC [B,H/32,W/32,C] (fp32, or split16 storage when PRECISION == "f16x3":
        see `features_f32`).  `_taps` (list) collects the stem output and every block's output (debugging / parity tools)."""
        if _taps is None:
            return self.run(img, want_features=True)[1]
        # per-layer schedule driven from Python: the debugging twin of smirk_backbone_forward (same kernels, same order), kept for `_taps`
        if self.training:
            raise L.SmirkHipError("the per-layer debugging schedule is eval-mode only; train mode runs through the encoders' forward (encoder_train.py)")
        lib, st, P = L.lib(), L.stream_ptr(), self._pack()
        img = L.as_f32c(img)
        B, _, H, W = img.shape
        w, sc, sh = P["stem"]
        x = torch.empty(B, (H + 1) // 2, (W + 1) // 2, 16, device=img.device)
        L.check((lib.smirk_stem_conv_s2_split16 if self._split else lib.smirk_stem_conv_s2)(L.ptr(img), L.ptr(w), L.ptr(sc), L.ptr(sh), L.ptr(x), B, H, W, 16, st))
        fused = self._split and not os.environ.get("SMIRK_DISABLE_MBCONV_FUSED")
        if _taps is not None:
            _taps.append(("stem", x))
        for si, stg in enumerate(self.blocks):
            for bi, blk in enumerate(stg):
                pk = P[(si, bi)]
                if fused and blk.kind in ("ds", "ir") and self._fusable(lib, x.shape[3], pk, blk):
                    x = self._fused_block(lib, st, x, pk, blk)
                elif blk.kind == "ds":
                    y = self._depthwise(lib, st, x, pk["dw"], blk.stride)
                    x = self._pointwise(lib, st, y, pk["pw"], relu=False, residual=x if blk.skip else None)
                elif blk.kind == "ir":
                    y = self._pointwise(lib, st, x, pk["pw"], relu=True)
                    y = self._depthwise(lib, st, y, pk["dw"], blk.stride)
                    x = self._pointwise(lib, st, y, pk["pwl"], relu=False, residual=x if blk.skip else None)
                else:
                    x = self._pointwise(lib, st, x, pk["pw"], relu=True)
                if _taps is not None:
                    _taps.append((f"block{si}.{bi}:{blk.kind}", x))
        return x


def features_f32(backbone, feat):
    """fp32 NHWC values of a backbone's output feature map (decodes the split16 storage of the f16x3 mode; test helper)."""
    if getattr(backbone, "_split", False):
        from .smirk_generator import split16_to_float
        return split16_to_float(feat)
    return feat


def create_backbone(backbone_name, pretrained=True):
    """smirk_encoder.py:7-12.  `pretrained` is accepted for signature compatibility; weights always come from the checkpoint
    the caller loads next (demo.py:55-58) — there is no download path."""
    backbone = MobileNetV3Features(backbone_name)
    return backbone, backbone.feature_info[-1]['num_chs']


class PoseEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder, feature_dim = create_backbone('tf_mobilenetv3_small_minimal_100')
        self.pose_cam_layers = nn.Sequential(nn.Linear(feature_dim, 6))
        self.init_weights()

    def init_weights(self):                                      # smirk_encoder.py:26-31
        with torch.no_grad():
            self.pose_cam_layers[-1].weight.mul_(0.001)
            self.pose_cam_layers[-1].bias.mul_(0.001)
            self.pose_cam_layers[-1].weight[3] = 0
            self.pose_cam_layers[-1].bias[3] = 7

    def forward(self, img):
        pose_cam, _ = self.encoder.run(img, self.pose_cam_layers[0])
        return {'pose_params': pose_cam[..., :3], 'cam': pose_cam[..., 3:]}


class ShapeEncoder(nn.Module):
    def __init__(self, n_shape=300):
        super().__init__()
        self.encoder, feature_dim = create_backbone('tf_mobilenetv3_large_minimal_100')
        self.shape_layers = nn.Sequential(nn.Linear(feature_dim, n_shape))
        self.init_weights()

    def init_weights(self):                                      # smirk_encoder.py:61-63
        with torch.no_grad():
            self.shape_layers[-1].weight.mul_(0)
            self.shape_layers[-1].bias.mul_(0)

    def forward(self, img):
        return {'shape_params': self.encoder.run(img, self.shape_layers[0])[0]}


class ExpressionEncoder(nn.Module):
    def __init__(self, n_exp=50):
        super().__init__()
        self.encoder, feature_dim = create_backbone('tf_mobilenetv3_large_minimal_100')
        self.expression_layers = nn.Sequential(nn.Linear(feature_dim, n_exp + 2 + 3))
        self.n_exp = n_exp
        self.init_weights()

    def init_weights(self):                                      # smirk_encoder.py:90-92
        with torch.no_grad():
            self.expression_layers[-1].weight.mul_(0.1)
            self.expression_layers[-1].bias.mul_(0.1)

    def forward(self, img):
        params, _ = self.encoder.run(img, self.expression_layers[0], clamp_n_exp=self.n_exp)
        n = self.n_exp
        return {'expression_params': params[..., :n], 'eyelid_params': params[..., n:n + 2],
                'jaw_params': params[..., n + 2:n + 5]}


class SmirkEncoder(nn.Module):
    def __init__(self, n_exp=50, n_shape=300):
        super().__init__()
        self.pose_encoder = PoseEncoder()
        self.shape_encoder = ShapeEncoder(n_shape=n_shape)
        self.expression_encoder = ExpressionEncoder(n_exp=n_exp)

    def forward(self, img):
        """The three regressors are independent (smirk_encoder.py:123-133 runs them back to back): each gets its own HIP stream so
        their many small launches interleave on the 256 CUs; the caller's stream joins all three before the dict is returned."""
        outputs = {}
        if not img.is_cuda:
            raise L.SmirkHipError("smirk_amd runs on the MI355X HIP device only: got a CPU tensor (no CPU fallback exists)")
        L.raise_if_range_tripped("smirk_amd.SmirkEncoder.forward")          # an overflow of the split-fp16 format in an EARLIER call is reported now (no sync)
        # SMIRK_ENCODER_SERIAL: profiling aid (reference order on one stream).  Training uses the three streams as well unless SMIRK_ENCODER_TRAIN_SERIAL is
        # set: autograd runs every backward node on the stream its forward ran on and orders the streams itself, so the backbones' backward passes
        # interleave exactly like their forward passes (their ~1000 launches per step are small and latency-bound one after the other).
        if os.environ.get("SMIRK_ENCODER_SERIAL") or (self.training and (getattr(self, "_train_serial", False) or os.environ.get("SMIRK_ENCODER_TRAIN_SERIAL"))):
            for enc in (self.pose_encoder, self.shape_encoder, self.expression_encoder):
                outputs.update(enc(img))
            return outputs
        main = torch.cuda.current_stream()
        if img.device not in _STREAMS:
            _STREAMS[img.device] = [torch.cuda.Stream(device=img.device) for _ in range(3)]
        fork = torch.cuda.Event()
        fork.record(main)
        for st, enc in zip(_STREAMS[img.device], (self.pose_encoder, self.shape_encoder, self.expression_encoder)):
            st.wait_event(fork)
            with torch.cuda.stream(st):
                o = enc(img)
            for t in o.values():                                 # allocated on the side stream, consumed on the caller's
                t.record_stream(main)
            outputs.update(o)
            done = torch.cuda.Event()
            done.record(st)
            main.wait_event(done)
        return outputs
