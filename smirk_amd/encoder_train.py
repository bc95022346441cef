"""TRAIN-mode forward and backward of one SmirkEncoder backbone (+ its Linear head) on the HIP path (BASELINE config 5, encoder slice).

What the reference computes after `self.train()` (smirk_trainer.py:349-355 -> base_trainer.py:108-111) when autograd differentiates
`self.smirk_encoder(reconstructed_img_2nd_path)` (smirk_trainer.py:297): every BatchNorm2d of the three timm `tf_mobilenetv3_*_minimal_100`
feature extractors (smirk_encoder.py:7-12) normalises with batch statistics and updates its running estimates — frozen sub-encoders included,
`freeze_module` only clears requires_grad — and the backward pass returns parameter gradients for the sub-encoders that are being optimised plus
the IMAGE gradient that the cycle loss sends back into the generator (config_train.yaml: freeze_generator_in_second_path False).

One torch.autograd.Function per backbone.  Forward: bare stem / depthwise / pointwise convolutions, each followed by
`smirk_bn_train_forward_split16`; global average pool + Linear; the ExpressionEncoder clamps (smirk_encoder.py:104-107).  Backward walks the tape:
BatchNorm backward, pointwise data gradients on the forward implicit-GEMM kernel with the transposed weight, pointwise weight gradients on the
split-fp16 x3 weight-gradient kernels (`smirk_conv_wgrad_f32`, default mode 2; the exact-fp32 MFMA kernels are selectable with $SMIRK_WGRAD_F16=0), depthwise / stem / head companions from csrc/train_encoder.hip.  Weight gradients are computed only for parameters that require
them, the image gradient only if the image requires it.
"""
import torch

from . import _lib as L
from .generator_train import _Ops
from .smirk_generator import _split16


class _EncOps(_Ops):
    def __init__(self, device, arith="f16x3"):
        super().__init__(device, arith=arith)
        self.dw_ws = None
        self.ones, self.zeros = {}, {}

    def pointwise(self, x, w_split, cout, residual=None):
        B, H, W, _ = x.shape
        return self.conv(x, None, w_split, B, H, W, cout, k=1, residual=residual)

    def pointwise_stats(self, x, w_split, cout):
        """1x1 convolution followed by a BatchNorm: -> (z, partial sums of z or None) (generator_train._Ops.conv_stats)"""
        B, H, W, _ = x.shape
        return self.conv_stats(x, None, w_split, B, H, W, cout, k=1)

    def depthwise(self, x, w9c, stride):
        B, H, W, C = x.shape
        if C not in self.ones:
            self.ones[C], self.zeros[C] = torch.ones(C, device=self.dev), torch.zeros(C, device=self.dev)
        out = torch.empty(B, (H + stride - 1) // stride, (W + stride - 1) // stride, C, device=self.dev)
        P = L.ptr
        L.check(self.lib.smirk_dwconv3x3_split16(P(x), P(w9c), P(self.ones[C]), P(self.zeros[C]), P(out), B, H, W, C, stride, 0, self.st))
        return out

    def depthwise_stats(self, x, w9c, stride):
        """train mode: the raw depthwise convolution + the stage-1 partial sums of its output's statistics in one launch -> (z, (fp64 partial rows, count) or None)"""
        if not self.fuse_stats:
            return self.depthwise(x, w9c, stride), None
        B, H, W, C = x.shape
        out = torch.empty(B, (H + stride - 1) // stride, (W + stride - 1) // stride, C, device=self.dev)
        part = torch.empty(1024 * C * 2, dtype=torch.float64, device=self.dev)
        rows = L.C.c_int(0)
        rc = self.lib.smirk_dwconv3x3_stats_split16(L.ptr(x), L.ptr(w9c), L.ptr(out), B, H, W, C, stride, L.ptr(part, torch.float64), L.C.byref(rows), self.st)
        if rc == L.SMIRK_ERR_UNSUPPORTED:                # inputs of 2 GiB and more: the plain kernel + the stand-alone statistics pass
            return self.depthwise(x, w9c, stride), None
        L.check(rc)
        return out, (part, rows.value)

    def depthwise_dgrad(self, dz, w9c, add, B, H, W, C, stride):
        dx = torch.empty(B, H, W, C, device=self.dev)
        L.check(self.lib.smirk_dwconv3x3_dgrad_split16(L.ptr(dz), L.ptr(w9c), L.ptr(add, allow_none=True), L.ptr(dx), B, H, W, C, stride, self.st))
        return dx

    def depthwise_wgrad(self, dz, x, stride):
        B, H, W, C = x.shape
        need = self.lib.smirk_dwconv3x3_wgrad_workspace_bytes(C)
        if self.dw_ws is None or self.dw_ws.numel() < need:
            self.dw_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        dw = torch.empty(C, 1, 3, 3, device=self.dev)                 # the parameter's own layout: no transpose-copy afterwards
        L.check(self.lib.smirk_dwconv3x3_wgrad_param_split16(L.ptr(dz), L.ptr(x), L.ptr(dw), B, H, W, C, stride, L.ptr(self.dw_ws, torch.uint8), self.dw_ws.numel(),
                                                             self.st))
        return dw


def _pw(conv):
    """nn.Conv2d(cin, cout, 1) weight -> split16 [cout][cin]   (test helper; the step itself uses _EncOps.pack: both images in one launch)"""
    return _split16(conv.weight.detach().float().reshape(conv.out_channels, conv.in_channels).contiguous())


def _pw_t(conv):
    """transposed: the weight of the data-gradient 1x1 convolution, split16 [cin][cout]"""
    return _split16(conv.weight.detach().float().reshape(conv.out_channels, conv.in_channels).t().contiguous())


def _dw(conv):
    return conv.weight.detach().float().reshape(conv.out_channels, 9).t().contiguous()                  # [9][C]


class BackboneTrainFunction(torch.autograd.Function):
    """(backbone, head Linear, clamp_n_exp, img, *params) -> head output [B, n_out] with the ExpressionEncoder clamps applied when clamp_n_exp >= 0.
    `params` = list(backbone.parameters()) + list(head.parameters()), passed so autograd routes their gradients."""

    @staticmethod
    def forward(ctx, backbone, head, clamp_n_exp, img, *params):
        img = L.as_f32c(img.detach())
        if not img.is_cuda:
            raise L.SmirkHipError("smirk_amd runs on the MI355X HIP device only: got a CPU tensor (no CPU fallback exists)")
        B, C3, H, W = img.shape
        if C3 != 3 or H < 32 or W < 32:
            raise L.SmirkHipError("SmirkEncoder: expected [B, 3, H >= 32, W >= 32] images")
        ctx.arith = getattr(backbone, "train_arith", "f16x3")            # generator_train.TRAIN_ARITH: "f16x1" = pointwise convs, their data and weight gradients on one MFMA
        ops = _EncOps(img.device, arith=ctx.arith)
        lib, st = ops.lib, ops.st
        from .generator_train import PackPlan
        plan = getattr(backbone, "_pack_plan", None)                      # every pointwise weight of the backbone: one packing launch per forward
        if plan is None or not plan.valid_for(params):
            plan = PackPlan()
            object.__setattr__(backbone, "_pack_plan", plan)
        else:
            plan.run(ops)
        ops.plan = plan
        tape = []
        c0 = backbone.conv_stem.out_channels
        wst = plan.request_special(ops, backbone.conv_stem.weight, L.PACK_STEM)[0]          # [c0][(ky,kx,c)] fp32, by the plan's one packing launch
        z = torch.empty(B, (H + 1) // 2, (W + 1) // 2, c0, device=img.device)
        L.check(lib.smirk_stem_conv_s2_raw_split16(L.ptr(img), L.ptr(wst), L.ptr(z), B, H, W, c0, st))
        x, mu, iv = ops.bn_forward(z, backbone.bn1, True)
        tape.append(("stem", (img, wst, z, mu, iv)))
        for stage in backbone.blocks:
            for blk in stage:
                if blk.kind == "ds":
                    wdw = plan.request_special(ops, blk.conv_dw.weight, L.PACK_DEPTHWISE)[0]           # [9][C] fp32 (the plan's one packing launch)
                    z1, sd = ops.depthwise_stats(x, wdw, blk.stride)
                    y1, m1, i1 = ops.bn_forward(z1, blk.bn1, True, stats=sd)
                    wf, wt = ops.pack(blk.conv_pw.weight)
                    z2, s2 = ops.pointwise_stats(y1, wf, blk.conv_pw.out_channels)
                    out, m2, i2 = ops.bn_forward(z2, blk.bn2, False, residual=x if blk.skip else None, stats=s2)
                    tape.append(("ds", blk, (x, z1, m1, i1, y1, z2, m2, i2, wdw, wt)))
                elif blk.kind == "ir":
                    wf1, wt1 = ops.pack(blk.conv_pw.weight)
                    z1, s1 = ops.pointwise_stats(x, wf1, blk.conv_pw.out_channels)
                    y1, m1, i1 = ops.bn_forward(z1, blk.bn1, True, stats=s1)
                    wdw = plan.request_special(ops, blk.conv_dw.weight, L.PACK_DEPTHWISE)[0]
                    z2, sd = ops.depthwise_stats(y1, wdw, blk.stride)
                    y2, m2, i2 = ops.bn_forward(z2, blk.bn2, True, stats=sd)
                    wf3, wt3 = ops.pack(blk.conv_pwl.weight)
                    z3, s3 = ops.pointwise_stats(y2, wf3, blk.conv_pwl.out_channels)
                    out, m3, i3 = ops.bn_forward(z3, blk.bn3, False, residual=x if blk.skip else None, stats=s3)
                    tape.append(("ir", blk, (x, z1, m1, i1, y1, z2, m2, i2, y2, z3, m3, i3, wt1, wdw, wt3)))
                else:
                    wf, wt = ops.pack(blk.conv.weight)
                    z1, s1 = ops.pointwise_stats(x, wf, blk.conv.out_channels)
                    out, m1, i1 = ops.bn_forward(z1, blk.bn1, True, stats=s1)
                    tape.append(("cn", blk, (x, z1, m1, i1, wt)))
                x = out
        Bf, hf, wf, Cf = x.shape
        hw_, hb = head.weight.detach().float().contiguous(), head.bias.detach().float().contiguous()
        N = hw_.shape[0]
        pooled = torch.empty(B, Cf, device=img.device)
        raw = torch.empty(B, N, device=img.device)
        L.check(lib.smirk_gap_linear_split16(L.ptr(x), L.ptr(hw_), L.ptr(hb), L.ptr(raw), L.ptr(pooled), B, hf * wf, Cf, N, st))
        out = raw
        if clamp_n_exp >= 0:
            out = raw.clone()
            L.check(lib.smirk_expression_clamps(L.ptr(out), B, clamp_n_exp, st))
        if not plan.sealed:
            plan.seal(params, img.device)
        ctx.plan, (ctx.plan_generation, ctx.plan_versions) = plan, plan.stamp(params)       # see PackPlan.check_tape
        ctx.backbone, ctx.head, ctx.tape = backbone, head, tape
        ctx.headrec = (hw_, pooled, raw, clamp_n_exp, (B, hf, wf, Cf, N))
        ctx.img_shape = (B, H, W)
        return out

    @staticmethod
    def backward(ctx, gout):
        if ctx.tape is None:
            raise RuntimeError("BackboneTrainFunction.backward called a second time: the tape of raw convolution outputs is released after the first "
                               "backward pass (retain_graph=True is not supported by the HIP training path; run the forward again)")
        backbone, head, tape = ctx.backbone, ctx.head, ctx.tape
        ctx.plan.check_tape(ctx.plan_generation, ctx.plan_versions)
        hw_, pooled, raw, n_exp, (B, hf, wf, Cf, N) = ctx.headrec
        ops = _EncOps(raw.device, arith=ctx.arith)
        lib, st = ops.lib, ops.st
        grads = {}
        need = lambda p: p.requires_grad
        gout = L.as_f32c(gout)
        if n_exp >= 0:                                      # torch.clamp / F.relu backward (smirk_encoder.py:104-107): gradient passes inside the range
            m = torch.ones_like(raw)
            e, j0, j = raw[:, n_exp:n_exp + 2], raw[:, n_exp + 2:n_exp + 3], raw[:, n_exp + 3:n_exp + 5]
            m[:, n_exp:n_exp + 2] = ((e >= 0) & (e <= 1)).float()
            m[:, n_exp + 2:n_exp + 3] = (j0 > 0).float()
            m[:, n_exp + 3:n_exp + 5] = ((j >= -0.2) & (j <= 0.2)).float()
            gout = (gout * m).contiguous()
        want_head = need(head.weight) or need(head.bias)
        dwh = torch.empty(N, Cf, device=raw.device) if want_head else None
        dbh = torch.empty(N, device=raw.device) if want_head else None
        g = torch.empty(B, hf, wf, Cf, device=raw.device)
        L.check(lib.smirk_gap_linear_backward_split16(L.ptr(gout), L.ptr(hw_), L.ptr(pooled), L.ptr(dwh, allow_none=True), L.ptr(dbh, allow_none=True), L.ptr(g),
                                                      B, hf * wf, Cf, N, st))
        if want_head:
            grads[id(head.weight)], grads[id(head.bias)] = dwh, dbh

        def bn_back(z, dy, bn, mu, iv, relu):
            dz, dg, db = ops.bn_backward(z, dy, bn, mu, iv, relu)
            if need(bn.weight):
                grads[id(bn.weight)] = dg
            if need(bn.bias):
                grads[id(bn.bias)] = db
            return dz

        def pw_wgrad(conv, dz, xin):
            if need(conv.weight):
                b, h, w, cin = xin.shape
                grads[id(conv.weight)] = ops.wgrad(dz, xin, b, h, w, conv.out_channels, cin, 1).reshape(conv.weight.shape)

        def dw_wgrad(conv, dz, xin, stride):
            if need(conv.weight):
                grads[id(conv.weight)] = ops.depthwise_wgrad(dz, xin, stride)

        dimg = None
        for rec in reversed(tape):
            kind = rec[0]
            if kind == "cn":
                _, blk, (x, z1, m1, i1, wt) = rec
                dz1 = bn_back(z1, g, blk.bn1, m1, i1, True)
                pw_wgrad(blk.conv, dz1, x)
                g = ops.pointwise(dz1, wt, blk.conv.in_channels)
            elif kind == "ir":
                _, blk, (x, z1, m1, i1, y1, z2, m2, i2, y2, z3, m3, i3, wt1, wdw, wt3) = rec
                dz3 = bn_back(z3, g, blk.bn3, m3, i3, False)
                pw_wgrad(blk.conv_pwl, dz3, y2)
                dy2 = ops.pointwise(dz3, wt3, blk.conv_pwl.in_channels)
                dz2 = bn_back(z2, dy2, blk.bn2, m2, i2, True)
                dw_wgrad(blk.conv_dw, dz2, y1, blk.stride)
                b, h, w, c = y1.shape
                dy1 = ops.depthwise_dgrad(dz2, wdw, None, b, h, w, c, blk.stride)
                dz1 = bn_back(z1, dy1, blk.bn1, m1, i1, True)
                pw_wgrad(blk.conv_pw, dz1, x)
                g = ops.pointwise(dz1, wt1, blk.conv_pw.in_channels, residual=g if blk.skip else None)
            elif kind == "ds":
                _, blk, (x, z1, m1, i1, y1, z2, m2, i2, wdw, wt) = rec
                dz2 = bn_back(z2, g, blk.bn2, m2, i2, False)
                pw_wgrad(blk.conv_pw, dz2, y1)
                dy1 = ops.pointwise(dz2, wt, blk.conv_pw.in_channels)
                dz1 = bn_back(z1, dy1, blk.bn1, m1, i1, True)
                dw_wgrad(blk.conv_dw, dz1, x, blk.stride)
                b, h, w, c = x.shape
                g = ops.depthwise_dgrad(dz1, wdw, g if blk.skip else None, b, h, w, c, blk.stride)
            else:
                _, (img, wst, z, mu, iv) = rec
                dz = bn_back(z, g, backbone.bn1, mu, iv, True)
                Bi, H, W = ctx.img_shape
                c0 = wst.shape[0]
                if need(backbone.conv_stem.weight):
                    nws = lib.smirk_stem_conv_s2_wgrad_workspace_bytes(c0)
                    ws = torch.empty(nws, dtype=torch.uint8, device=raw.device)
                    dws = torch.empty(c0, 27, device=raw.device)
                    L.check(lib.smirk_stem_conv_s2_wgrad_split16(L.ptr(img), L.ptr(dz), L.ptr(dws), Bi, H, W, c0, L.ptr(ws, torch.uint8), nws, st))
                    grads[id(backbone.conv_stem.weight)] = dws.reshape(c0, 3, 3, 3).permute(0, 3, 1, 2).contiguous()
                if ctx.needs_input_grad[3]:
                    dimg = torch.empty(Bi, 3, H, W, device=raw.device)
                    L.check(lib.smirk_stem_conv_s2_dgrad_split16(L.ptr(dz), L.ptr(wst), L.ptr(dimg), Bi, H, W, c0, st))
        ctx.tape = ctx.headrec = None
        out = [None, None, None, dimg]
        for p in list(backbone.parameters()) + list(head.parameters()):
            gp = grads.get(id(p))
            out.append(None if gp is None else gp.reshape(p.shape).to(p.dtype))
        return tuple(out)
