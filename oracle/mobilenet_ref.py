"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

torch-CPU fp32 restatement of the encoder backbones SMIRK obtains from timm (NOT on disk):
    timm==0.9.16 (requirements.txt:10), timm.create_model('tf_mobilenetv3_{small,large}_minimal_100',
    pretrained=True, features_only=True)            <- call site src/smirk_encoder.py:7-12,18,52,80
following (from memory) timm/models/mobilenetv3.py::_gen_mobilenet_v3 + MobileNetV3Features,
_efficientnet_blocks.py (DepthwiseSeparableConv / InvertedResidual / ConvBnAct), _efficientnet_builder.py
(decode_arch_def, make_divisible), layers/conv2d_same.py + padding.py ("tf_" => bn_eps 1e-3, pad_type 'same').
"minimal" => ReLU everywhere, 3x3 kernels only, no squeeze-excite, all convs bias-free.
PARITY UNPINNED: neither timm nor a checkpoint is available; defended by the parameter counts / feature shapes
that timm publishes (tests/test_oracle_mobilenet.py) — 1,413,208 / 428,888 feature params, taps [16,24,40,112,960]
/ [16,16,24,48,576].

Module / parameter NAMES reproduce timm's state_dict keys (conv_stem, bn1, blocks.{s}.{i}.conv_dw ...), because the
reference loads its checkpoint with strict=True (demo.py:56-58).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1e-3

ARCH = {
    "tf_mobilenetv3_large_minimal_100": [
        ["ds_r1_k3_s1_e1_c16"],
        ["ir_r1_k3_s2_e4_c24", "ir_r1_k3_s1_e3_c24"],
        ["ir_r3_k3_s2_e3_c40"],
        ["ir_r1_k3_s2_e6_c80", "ir_r1_k3_s1_e2.5_c80", "ir_r2_k3_s1_e2.3_c80"],
        ["ir_r2_k3_s1_e6_c112"],
        ["ir_r3_k3_s2_e6_c160"],
        ["cn_r1_k1_s1_c960"],
    ],
    "tf_mobilenetv3_small_minimal_100": [
        ["ds_r1_k3_s2_e1_c16"],
        ["ir_r1_k3_s2_e4.5_c24", "ir_r1_k3_s1_e3.67_c24"],
        ["ir_r1_k3_s2_e4_c40", "ir_r2_k3_s1_e6_c40"],
        ["ir_r2_k3_s1_e3_c48"],
        ["ir_r3_k3_s2_e6_c96"],
        ["cn_r1_k1_s1_c576"],
    ],
}


def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


def decode(name):
    """-> list of stages, each a list of dict(type, k, s, e, c) with repeats expanded (only first repeat strided)."""
    stages = []
    for stage in ARCH[name]:
        blocks = []
        for s in stage:
            ops = s.split("_")
            d = dict(type=ops[0], e=1.0)
            for o in ops[1:]:
                if o[0] == "r": r = int(o[1:])
                elif o[0] == "k": d["k"] = int(o[1:])
                elif o[0] == "s": d["s"] = int(o[1:])
                elif o[0] == "e": d["e"] = float(o[1:])
                elif o[0] == "c": d["c"] = int(o[1:])
            for i in range(r):
                b = dict(d)
                if i > 0:
                    b["s"] = 1
                blocks.append(b)
        stages.append(blocks)
    return stages


class Conv2dSame(nn.Conv2d):
    """TF 'SAME' padding computed from the input size (layers/conv2d_same.py)."""

    def forward(self, x):
        ih, iw = x.shape[-2:]
        k, s = self.kernel_size[0], self.stride[0]
        ph = max((math.ceil(ih / s) - 1) * s + (k - 1) + 1 - ih, 0)
        pw = max((math.ceil(iw / s) - 1) * s + (k - 1) + 1 - iw, 0)
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
        return F.conv2d(x, self.weight, None, self.stride, 0, self.dilation, self.groups)


def _conv(ci, co, k, s=1, groups=1):
    if s > 1 and k > 1:
        return Conv2dSame(ci, co, k, s, 0, groups=groups, bias=False)
    return nn.Conv2d(ci, co, k, s, (k - 1) // 2, groups=groups, bias=False)


class _BnAct(nn.BatchNorm2d):
    def __init__(self, c, act=True):
        super().__init__(c, eps=BN_EPS)
        self._act = act

    def forward(self, x):
        x = super().forward(x)
        return F.relu(x) if self._act else x


class DS(nn.Module):
    def __init__(self, ci, co, s):
        super().__init__()
        self.conv_dw = _conv(ci, ci, 3, s, groups=ci); self.bn1 = _BnAct(ci)
        self.conv_pw = _conv(ci, co, 1); self.bn2 = _BnAct(co, act=False)
        self.has_skip = (s == 1 and ci == co)

    def forward(self, x):
        y = self.bn2(self.conv_pw(self.bn1(self.conv_dw(x))))
        return y + x if self.has_skip else y


class IR(nn.Module):
    def __init__(self, ci, co, s, e):
        super().__init__()
        mid = make_divisible(ci * e)
        self.conv_pw = _conv(ci, mid, 1); self.bn1 = _BnAct(mid)
        self.conv_dw = _conv(mid, mid, 3, s, groups=mid); self.bn2 = _BnAct(mid)
        self.conv_pwl = _conv(mid, co, 1); self.bn3 = _BnAct(co, act=False)
        self.has_skip = (s == 1 and ci == co)

    def forward(self, x):
        y = self.bn3(self.conv_pwl(self.bn2(self.conv_dw(self.bn1(self.conv_pw(x))))))
        return y + x if self.has_skip else y


class CN(nn.Module):
    def __init__(self, ci, co):
        super().__init__()
        self.conv = _conv(ci, co, 1); self.bn1 = _BnAct(co)

    def forward(self, x):
        return self.bn1(self.conv(x))


class MobileNetV3Features(nn.Module):
    def __init__(self, name):
        super().__init__()
        self.conv_stem = _conv(3, 16, 3, 2)
        self.bn1 = _BnAct(16)
        ci, stages, self.feature_info, taps = 16, [], [], []
        dec = decode(name)
        red = 2
        for si, st in enumerate(dec):
            blocks = []
            if st[0]["s"] == 2:                                  # feature tap = last block before a strided stage
                taps.append(dict(num_chs=ci, reduction=red, stage=si))
                red *= 2
            for b in st:
                if b["type"] == "ds": blocks.append(DS(ci, b["c"], b["s"]))
                elif b["type"] == "ir": blocks.append(IR(ci, b["c"], b["s"], b["e"]))
                else: blocks.append(CN(ci, b["c"]))
                ci = b["c"]
            stages.append(nn.Sequential(*blocks))
        taps.append(dict(num_chs=ci, reduction=red, stage=len(dec)))
        self.blocks = nn.Sequential(*stages)
        self.feature_info = taps
        self._tap_stages = [t["stage"] for t in taps]

    def forward(self, x):
        x = self.bn1(self.conv_stem(x))
        feats = [x] if 0 in self._tap_stages else []
        for i, st in enumerate(self.blocks):
            x = st(x)
            if i + 1 in self._tap_stages:
                feats.append(x)
        return feats


def create_model(name):
    return MobileNetV3Features(name)


def synth_init_(module, gen):
    """SURVEY.md §8(d) synthetic weights, in place, deterministic in module order."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            fan_in = (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]   # keeps activations O(1) in depth
            m.weight.data = torch.randn(m.weight.shape, generator=gen) * (2.0 / fan_in) ** 0.5
        elif isinstance(m, nn.BatchNorm2d):
            c = m.num_features
            m.weight.data = torch.rand(c, generator=gen) + 0.5
            m.bias.data = torch.randn(c, generator=gen) * 0.1
            m.running_mean.data = torch.randn(c, generator=gen) * 0.1
            m.running_var.data = torch.rand(c, generator=gen) + 0.5


class SmirkEncoderRef(nn.Module):
    """Restatement of src/smirk_encoder.py:14-133 on top of the restated backbones (same key names)."""

    def __init__(self, n_exp=50, n_shape=300):
        super().__init__()
        mk = lambda nm: nn.ModuleDict()
        self.n_exp = n_exp
        self.pose_encoder = nn.Module(); self.shape_encoder = nn.Module(); self.expression_encoder = nn.Module()
        self.pose_encoder.encoder = create_model("tf_mobilenetv3_small_minimal_100")
        self.pose_encoder.pose_cam_layers = nn.Sequential(nn.Linear(576, 6))
        self.shape_encoder.encoder = create_model("tf_mobilenetv3_large_minimal_100")
        self.shape_encoder.shape_layers = nn.Sequential(nn.Linear(960, n_shape))
        self.expression_encoder.encoder = create_model("tf_mobilenetv3_large_minimal_100")
        self.expression_encoder.expression_layers = nn.Sequential(nn.Linear(960, n_exp + 5))

    @staticmethod
    def _feat(enc, img):
        f = enc(img)[-1]
        return F.adaptive_avg_pool2d(f, (1, 1)).squeeze(-1).squeeze(-1)

    def forward(self, img):
        B = img.size(0)
        pc = self.pose_encoder.pose_cam_layers(self._feat(self.pose_encoder.encoder, img)).reshape(B, -1)
        sh = self.shape_encoder.shape_layers(self._feat(self.shape_encoder.encoder, img)).reshape(B, -1)
        ex = self.expression_encoder.expression_layers(self._feat(self.expression_encoder.encoder, img)).reshape(B, -1)
        n = self.n_exp
        return dict(pose_params=pc[..., :3], cam=pc[..., 3:], shape_params=sh, expression_params=ex[..., :n],
                    eyelid_params=torch.clamp(ex[..., n:n + 2], 0, 1),
                    jaw_params=torch.cat([F.relu(ex[..., n + 2].unsqueeze(-1)), torch.clamp(ex[..., n + 3:n + 5], -.2, .2)], -1))


def _calibrate_bn_(module, x, gen):
    """Set every BN's running stats to the batch statistics of a calibration batch (momentum 1 train pass), then
    perturb them, so the random network is as well-conditioned as a trained one (activations O(1) at every depth)."""
    bns = [m for m in module.modules() if isinstance(m, nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    module.train()
    with torch.no_grad():
        module(x)
    module.eval()
    for m in bns:
        m.momentum = 0.1
        sd = m.running_var.sqrt()
        m.running_mean.data = m.running_mean + torch.randn(m.num_features, generator=gen) * 0.1 * sd
        m.running_var.data = m.running_var * (torch.rand(m.num_features, generator=gen) + 0.5)
        m.num_batches_tracked.zero_()


def _calibrate_head_(lin, feats, gen, mean, std):
    """Linear head whose outputs on the calibration features have roughly the given per-output mean / std."""
    mu, sg = feats.mean(0), feats.std(0).clamp_min(1e-3 * feats.abs().mean())
    C = feats.shape[1]
    mean = torch.as_tensor(mean, dtype=torch.float32).expand(lin.out_features)
    std = torch.as_tensor(std, dtype=torch.float32).expand(lin.out_features)
    W = torch.randn(lin.out_features, C, generator=gen) / C ** 0.5 / sg[None] * std[:, None]
    lin.weight.data = W
    lin.bias.data = mean - W @ mu


def synth_encoder_state_dict(seed=1234, n_exp=50, n_shape=300):
    """Seeded synthetic SmirkEncoder weights (adapted): conv ~ N(0, sqrt(2/fan_in)), BN stats calibrated
    on 16 synthetic images then perturbed, heads scaled so the regressed FLAME parameters fall in the ranges the real
    network produces (pose +-0.4, cam scale ~8, |t|<0.1, shape/exp ~N(0,1), jaw[0] in [0,.5], eyelid in [0,1])."""
    from .assets import synth_images
    g = torch.Generator().manual_seed(seed)
    m = SmirkEncoderRef(n_exp, n_shape)
    synth_init_(m, g)
    x = synth_images(16, seed=4242)
    for enc in (m.pose_encoder.encoder, m.shape_encoder.encoder, m.expression_encoder.encoder):
        _calibrate_bn_(enc, x, g)
    m.eval()
    with torch.no_grad():
        fp = m._feat(m.pose_encoder.encoder, x)
        fs = m._feat(m.shape_encoder.encoder, x)
        fe = m._feat(m.expression_encoder.encoder, x)
    _calibrate_head_(m.pose_encoder.pose_cam_layers[0], fp, g, [0, 0, 0, 8, 0, 0], [.15, .15, .15, .7, .03, .03])
    _calibrate_head_(m.shape_encoder.shape_layers[0], fs, g, 0.0, 0.5)
    _calibrate_head_(m.expression_encoder.expression_layers[0], fe, g,
                     [0.0] * n_exp + [.5, .5, .2, 0, 0], [1.0] * n_exp + [.3, .3, .15, .1, .1])
    return {k: v.clone() for k, v in m.state_dict().items()}
