"""TRAIN-mode forward and backward of the SmirkGenerator U-Net on the HIP path (BASELINE config 5, generator slice).

What the reference computes when `smirk_trainer.py:349-355` has called `self.train()` and autograd differentiates
`self.smirk_generator(torch.cat([rendered_img, masked_img], 1))` (smirk_trainer.py:94, :293): every BatchNorm2d normalises with the statistics of
the batch and updates its running estimates, and the backward pass returns gradients for the input and for all 178-ish parameters.

Here the whole network is ONE torch.autograd.Function.  Forward: raw convolutions on the implicit-GEMM / ping-pong / halo-patch kernels, then
`smirk_bn_train_forward_split16` (statistics, normalise, affine, residual, ReLU, running-stat update).  Backward walks the recorded tape:
`smirk_bn_train_backward_split16`, data gradients on the SAME forward conv kernels with the weights rotated by 180 degrees and Cin <-> Cout
swapped, weight gradients on `smirk_conv_wgrad_f32` (split-fp16 x3 MFMA with LDS transpose reads; exact fp32 MFMA selectable), max-pool / reflection-pad / ConvTranspose2d / sigmoid companions
(csrc/train.hip).  Arithmetic: split-fp16 x3 MFMA for the convolutions and their data gradients (fp32-class, like inference), fp32 / fp64 for
statistics and weight gradients — tighter than the bf16 autocast BASELINE config 5 names.  The layer schedule lives in Python for now (a training
step is ~25 ms of GPU time at B = 64; the host is not yet the limiter).
"""
import torch

from . import _lib as L
from .smirk_generator import _split16, split16_to_float

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
# Arithmetic of the convolutions, their data gradients and their weight gradients in TRAIN mode, selected per module by the attribute `train_arith`:
#   "f16x3"  split-fp16 x3 MFMA (fp32-class; the parity mode against the fp32 reference) — the default
#   "f16x1"  the hi halves only, one MFMA per product block, fp32 accumulation: the 16-bit class BASELINE config 5 names (the reference trains under bf16
#            autocast, base_trainer.py / train.py; fp16 keeps 11 significand bits to bf16's 8).  Activations stay split16 in memory, so the two modes
#            share every other kernel; tested against the reference's own bf16-autocast distance to float64 (tests/golden/generator_train_golden.npz).
TRAIN_ARITH = ("f16x3", "f16x1")


def _pack_fwd(w, cin_pad=None):
    """[Cout,Cin,3,3] -> split16 [Cout][(ky,kx,c)]"""
    w = w.detach().float()
    if cin_pad is not None and cin_pad > w.shape[1]:
        w = torch.cat([w, w.new_zeros(w.shape[0], cin_pad - w.shape[1], 3, 3)], 1)
    return _split16(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous())


def _pack_dgrad(w, cout_pad=None):
    """weights of the data-gradient convolution: [Cin][(ky',kx',co)] = W[co][ci][2-ky'][2-kx']   (rotation by 180 degrees, Cin <-> Cout)"""
    w = w.detach().float().flip(2, 3).permute(1, 2, 3, 0)                   # [Cin][ky][kx][Cout]
    if cout_pad is not None and cout_pad > w.shape[0]:
        w = torch.cat([w, w.new_zeros(cout_pad - w.shape[0], 3, 3, w.shape[3])], 0)
    return _split16(w.reshape(w.shape[0], -1).contiguous())


_X1_FALLBACK_WARNED = False


def _warn_x1_fallback(B, H, W, cout, cin, k):
    """train_arith = "f16x1" asked for, but this layer's weight gradient ran in the f32-class arithmetic (the fp16 weight-gradient kernels address their operands
    with 32-bit offsets: tensors of 2 GiB and more, or $SMIRK_WGRAD_F16=0, take the f32-class kernels).  More accurate, slower — and said once, not silently."""
    global _X1_FALLBACK_WARNED
    if not _X1_FALLBACK_WARNED:
        _X1_FALLBACK_WARNED = True
        import warnings
        warnings.warn(f"smirk_amd: train_arith='f16x1' but the weight gradient of a {k}x{k} conv ({cin}->{cout} at B={B}, {H}x{W}) was computed by the f32-class "
                      "kernels (operand >= 2 GiB or SMIRK_WGRAD_F16=0); further occurrences are not reported", RuntimeWarning, stacklevel=3)


def _grad_like(weight):
    """gradient buffer in the parameter's SHAPE, contiguous fp32 whatever the parameter's memory format (the C entry writes the plain row-major layout)"""
    return torch.empty(weight.shape, dtype=torch.float32, device=weight.device)


class _Ops:
    """thin stateful wrapper over the C entries: one reduction workspace, the launch stream, and conv descriptors"""

    def __init__(self, device, eval_bn=False, arith="f16x3"):
        self.lib, self.st, self.dev = L.lib(), L.stream_ptr(), device
        self.eval_bn = bool(eval_bn)                     # BatchNorm from the running statistics (module in .eval() inside an autograd graph)
        if arith not in TRAIN_ARITH:
            raise L.SmirkHipError(f"train_arith must be one of {TRAIN_ARITH}, got {arith!r}")
        self.x1 = arith == "f16x1"                       # one MFMA per product block (hi halves only): BASELINE config 5's 16-bit class
        self.plan = None
        import os
        self.fuse_stats = not os.environ.get("SMIRK_BN_STATS_UNFUSED")       # A/B switch for tests: BatchNorm statistics by a pass over the stored tensor
        self.rm_saved = {}                                # eval-mode BatchNorm: id(bn) -> running_mean as the forward saw it
        self.red_ws = torch.empty(self.lib.smirk_train_reduce_workspace_bytes(1024), dtype=torch.uint8, device=device)
        self.wg_ws = None

    def conv_stats(self, x0, x1, w, B, H, W, cout, k=3, reflect=False):
        """the raw convolution whose BatchNorm follows: -> (z, (partials, rows) or None).  The kernel leaves per-tile partial sums (sum z, sum z^2) of its output
        from the MFMA accumulators where the kernel family serving the shape can (smirk_conv_igemm_stats_split16); `None` = reduce the stored tensor instead."""
        if self.eval_bn or not self.fuse_stats:
            return self.conv(x0, x1, w, B, H, W, cout, k=k, reflect=reflect), None
        d = L.SmirkConvDesc()
        d.B, d.H, d.W = B, H, W
        d.C0, d.C1 = x0.shape[-1], (x1.shape[-1] if x1 is not None else 0)
        d.Cout, d.KH, d.KW, d.stride = cout, k, k, 1
        d.pad_t = d.pad_l = (k - 1) // 2
        d.Ho, d.Wo = H, W
        d.pad_mode = L.PAD_REFLECT if reflect else L.PAD_ZERO
        d.act, d.out_mode = L.ACT_NONE, L.OUT_NHWC
        out = torch.empty((B, H, W, cout), device=self.dev)
        part = torch.empty((self.lib.smirk_conv_stats_rows_max(d), cout, 2), device=self.dev)
        rows = L.C.c_int(0)
        P = L.ptr
        L.check(self.lib.smirk_conv_igemm_stats_split16(d, P(x0), P(x1, allow_none=True), P(w), P(out), P(part), L.C.byref(rows), int(self.x1), self.st))
        return out, ((part, rows.value) if rows.value > 0 else None)

    def conv(self, x0, x1, w, B, H, W, cout, k=3, reflect=False, pad=None, out_hw=None, convt=False, shift=None, residual=None):
        d = L.SmirkConvDesc()
        d.B, d.H, d.W = B, H, W
        d.C0, d.C1 = x0.shape[-1], (x1.shape[-1] if x1 is not None else 0)
        d.Cout, d.KH, d.KW, d.stride = cout, k, k, 1
        d.pad_t = d.pad_l = (k - 1) // 2 if pad is None else pad
        d.Ho, d.Wo = (H, W) if out_hw is None else out_hw
        d.pad_mode = L.PAD_REFLECT if reflect else L.PAD_ZERO
        d.act = L.ACT_NONE
        d.out_mode = L.OUT_CONVT2X2 if convt else L.OUT_NHWC
        out = torch.empty((B, 2 * H, 2 * W, cout) if convt else (B, d.Ho, d.Wo, cout), device=self.dev)
        P = L.ptr
        entry = self.lib.smirk_conv_igemm_f16x1 if self.x1 else self.lib.smirk_conv_igemm_f16x3
        L.check(entry(d, P(x0), P(x1, allow_none=True), P(w), None, P(shift, allow_none=True), P(residual, allow_none=True), P(out), self.st))
        return out

    def pack(self, weight, cin_off=0, cin=None, cin_pad=None, fwd=True, dgrad=True):
        """nn.Conv2d weight (fp32 parameter) -> (forward weight split16 [Cout][T*cin_pad] or None, data-gradient weight split16 [cin_pad][T*Cout] or None).
        With a PackPlan attached (`self.plan`) the call only RECORDS the job on the plan's first pass and hands out the plan's buffers afterwards: all
        weights of the network are then packed by ONE launch at the top of the forward (PackPlan.run) instead of one launch per call."""
        if self.plan is not None:
            return self.plan.request(self, weight, cin_off, cin, cin_pad, fwd, dgrad)
        return self._pack_now(weight, cin_off, cin, cin_pad, fwd, dgrad)

    def _pack_now(self, weight, cin_off=0, cin=None, cin_pad=None, fwd=True, dgrad=True):
        w = weight.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.float().contiguous()
        cout, ctot, k = w.shape[0], w.shape[1], w.shape[2]
        cin = ctot - cin_off if cin is None else cin
        cp = cin if cin_pad is None else max(cin, cin_pad)
        f = torch.empty(cout, k * k * cp, device=self.dev) if fwd else None
        d = torch.empty(cp, k * k * cout, device=self.dev) if dgrad else None
        L.check(self.lib.smirk_pack_conv_weights_split16(L.ptr(w), cout, ctot, cin_off, cin, k, cp, L.ptr(f, allow_none=True), L.ptr(d, allow_none=True), self.st))
        return f, d

    def bn_forward(self, z, bn, relu, residual=None, stats=None):
        C = z.shape[-1]
        M = z.numel() // C
        mean, var, inv = (torch.empty(C, device=self.dev) for _ in range(3))
        y = torch.empty_like(z)
        P = L.ptr
        if self.eval_bn:                                 # `mean` carries the zero vector the backward needs, `inv` = rsqrt(running_var + eps)
            if bn.running_mean is None or bn.running_var is None:
                raise L.SmirkHipError("eval-mode BatchNorm needs running statistics (track_running_stats=True)")
            L.check(self.lib.smirk_bn_eval_forward_split16(P(z), M, C, P(bn.weight.detach()), P(bn.bias.detach()), P(bn.running_mean), P(bn.running_var),
                                                           P(residual, allow_none=True), int(relu), float(bn.eps), P(inv), P(mean), P(y), self.st))
            # the backward re-derives the ReLU mask from (z - running_mean) * inv: it must see the running_mean of THIS forward even if a train-mode forward of
            # the same module moves the buffer before the backward runs (C floats per layer)
            self.rm_saved[id(bn)] = bn.running_mean.detach().clone()
            return y, mean, inv
        track = bn.running_mean is not None and bn.running_var is not None
        nbt = bn.num_batches_tracked if track else None
        if nbt is not None and (bn.momentum is None or nbt.dtype != torch.int64 or not nbt.is_cuda):
            nbt += 1                                     # the cumulative average needs the count on the host; otherwise the kernel below increments it
            nbt = None
        # nn.BatchNorm2d: momentum=None means a cumulative moving average, factor 1 / num_batches_tracked (after the increment above)
        if bn.momentum is None and track and torch.cuda.is_current_stream_capturing():
            raise L.SmirkHipError("BatchNorm2d(momentum=None) (cumulative average) cannot be captured into a HIP graph: its factor 1 / num_batches_tracked is a host "
                                  "scalar read with a device synchronisation and would be frozen into the replay; use a numeric momentum (the reference does)")
        mom = float(bn.momentum) if bn.momentum is not None else (1.0 / max(int(bn.num_batches_tracked), 1) if track else 0.0)
        if stats is not None:                            # the producing convolution left the partial sums of z: finalise from them, no pass over z
            part, rows = stats
            fp64 = int(part.dtype == torch.float64)
            L.check(self.lib.smirk_bn_train_forward_partials_split16(P(z), M, C, P(bn.weight.detach()), P(bn.bias.detach()), P(residual, allow_none=True), int(relu),
                                                                     float(bn.eps), mom, P(bn.running_mean if track else None, allow_none=True),
                                                                     P(bn.running_var if track else None, allow_none=True), P(nbt, torch.int64, allow_none=True),
                                                                     P(mean), P(var), P(inv), P(y), P(part, part.dtype), int(rows), fp64, self.st))
            return y, mean, inv
        L.check(self.lib.smirk_bn_train_forward_split16(P(z), M, C, P(bn.weight.detach()), P(bn.bias.detach()), P(residual, allow_none=True), int(relu),
                                                        float(bn.eps), mom, P(bn.running_mean if track else None, allow_none=True),
                                                        P(bn.running_var if track else None, allow_none=True), P(nbt, torch.int64, allow_none=True), P(mean), P(var),
                                                        P(inv), P(y),
                                                        P(self.red_ws, torch.uint8), self.red_ws.numel(), self.st))
        return y, mean, inv

    def bn_backward(self, z, dy, bn, mean, inv, relu):
        C = z.shape[-1]
        M = z.numel() // C
        dz, dg, db = torch.empty_like(z), torch.empty(C, device=self.dev), torch.empty(C, device=self.dev)
        P = L.ptr
        if self.eval_bn:                                 # dz = gamma * invstd * dy * relu-mask; the affine parameters' gradients are not needed (frozen)
            L.check(self.lib.smirk_bn_eval_backward_split16(P(z), P(dy), M, C, P(bn.weight.detach()), P(bn.bias.detach()), P(self.rm_saved.get(id(bn), bn.running_mean)), P(inv),
                                                            P(mean), int(relu), P(dz), self.st))
            return dz, None, None
        L.check(self.lib.smirk_bn_train_backward_split16(P(z), P(dy), M, C, P(bn.weight.detach()), P(bn.bias.detach()), P(mean), P(inv), int(relu), P(dz),
                                                         P(dg), P(db), P(self.red_ws, torch.uint8), self.red_ws.numel(), self.st))
        return dz, dg, db

    def wgrad(self, dz, x, B, H, W, cout, cin, k, reflect=False):
        need = self.lib.smirk_conv_wgrad_workspace_bytes(B, H, W, cout, cin, k)
        if self.wg_ws is None or self.wg_ws.numel() < need:
            self.wg_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        dw = torch.empty(cout, k * k * cin, device=self.dev)
        P = L.ptr
        args = (P(dz), P(x), P(dw), B, H, W, cout, cin, k, int(reflect), P(self.wg_ws, torch.uint8), self.wg_ws.numel(), self.st)
        rc = self.lib.smirk_conv_wgrad_f16x1(*args) if self.x1 else L.SMIRK_ERR_UNSUPPORTED
        if rc == L.SMIRK_ERR_UNSUPPORTED:                # (f16x1 needs the fp16 weight-gradient kernels: operands >= 2 GiB take the exact-fp32 kernel)
            if self.x1:
                _warn_x1_fallback(B, H, W, cout, cin, k)
            rc = self.lib.smirk_conv_wgrad_f32(*args)
        L.check(rc)
        return dw

    def wgrad_param(self, dz, x, B, H, W, cout, cin, k, out, layout=1, cin_off=0, cin_real=None, reflect=False):
        """weight gradient summed straight into `out`, a tensor in the PARAMETER's layout (smirk_conv_wgrad_param): nn.Conv2d weight [cout][cin_total][k][k],
        channel range [cin_off, cin_off + cin_real) (layout 1) or nn.ConvTranspose2d(2, 2) weight [cout][cin / 4][2][2] (layout 2) — no permute-copy, no torch.cat"""
        need = self.lib.smirk_conv_wgrad_workspace_bytes(B, H, W, cout, cin, k)
        if self.wg_ws is None or self.wg_ws.numel() < need:
            self.wg_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        P = L.ptr
        cin_total = out.shape[1] if layout == 1 else 0
        before = self.lib.smirk_conv_wgrad_x1_fallbacks() if self.x1 else 0
        L.check(self.lib.smirk_conv_wgrad_param(P(dz), P(x), P(out), B, H, W, cout, cin, k, int(reflect), layout, cin_total, cin_off,
                                                cin if cin_real is None else cin_real, int(self.x1), P(self.wg_ws, torch.uint8), self.wg_ws.numel(), self.st))
        if self.x1 and self.lib.smirk_conv_wgrad_x1_fallbacks() != before:
            _warn_x1_fallback(B, H, W, cout, cin, k)
        return out

    def colsum(self, x):
        C = x.shape[-1]
        out = torch.empty(C, device=self.dev)
        L.check(self.lib.smirk_colsum_split16(L.ptr(x), x.numel() // C, C, L.ptr(out), L.ptr(self.red_ws, torch.uint8), self.red_ws.numel(), self.st))
        return out


class PackPlan:
    """All conv-weight operand images of one module, packed by one launch per forward.  First forward: `request` runs the per-weight kernel and records the job
    (weight pointer, slice, padding, output buffers); `seal` uploads the job table.  Later forwards: `run` launches smirk_pack_conv_weights_batch_split16 once and
    `request` hands the recorded buffers out in the same order (the forward's sequence of pack calls is a function of the architecture only).  Buffers are owned
    by the plan and overwritten by the next forward: the tape of an earlier forward sees the new images — identical values unless the weights changed in between,
    which is also when a stale tape would be wrong in the reference (autograd raises there; here backward-before-next-forward is the contract)."""

    def __init__(self):
        self.jobs, self.bufs, self.sealed, self.cursor, self.table, self.total = [], [], False, 0, None, 0
        self.generation, self.versions = 0, None                     # which forward the buffers hold the weights of, and the weights' autograd versions then

    def stamp(self, params):
        """called by every forward after its weights are packed: (generation, versions) identify the buffer contents for that forward's tape"""
        self.generation += 1
        self.versions = tuple(int(p._version) for p in params)
        return self.generation, self.versions

    def check_tape(self, generation, versions):
        """backward of the forward stamped (generation, versions): the plan-owned data-gradient weights must still hold THAT forward's weights.  A later forward
        re-packs in place; that is harmless while the parameters are unchanged (identical images) and silently wrong once they were modified in between —
        the situation torch autograd reports through its version counters, so it is reported the same way here."""
        if generation != self.generation and versions != self.versions:
            raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation: SmirkGenerator's weights changed "
                               "after this forward and another forward of the same module has re-packed the plan-owned weight images since "
                               "(run backward before the next forward of the module, or do not update its parameters in between)")

    def valid_for(self, params):
        return self.sealed and self.key == tuple((p.data_ptr(), tuple(p.shape)) for p in params)

    def request(self, ops, weight, cin_off, cin, cin_pad, fwd, dgrad):
        if self.sealed:
            f, d = self.bufs[self.cursor]
            self.cursor += 1
            return f, d
        w = weight.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            raise L.SmirkHipError("PackPlan needs contiguous fp32 conv weights")
        f, d = ops._pack_now(weight, cin_off, cin, cin_pad, fwd, dgrad)
        cout, ctot, k = w.shape[0], w.shape[1], w.shape[2]
        cin_ = ctot - cin_off if cin is None else cin
        cp = cin_ if cin_pad is None else max(cin_, cin_pad)
        n = (cout * k * k * cp // 8 if fwd else 0) + (cp * k * k * cout // 8 if dgrad else 0)
        self.jobs.append((w.data_ptr(), f, d, cout, ctot, cin_off, cin_, k, cp, n))
        self.bufs.append((f, d))
        return f, d

    def request_special(self, ops, weight, kind):
        """weights that are not plain nn.Conv2d kernels, re-laid-out by the same one launch (SmirkPackJob kinds, include/smirk_hip.h):
        PACK_DEPTHWISE [C,1,3,3] -> fp32 [9][C];  PACK_STEM [Cout,3,3,3] -> fp32 [Cout][(ky,kx,c)];  PACK_CONVT2X2 [Cin,Cout,2,2] -> (split16 [(dydx,co)][ci],
        split16 [ci][(dydx,co)]).  First pass: computed with torch ops and recorded; afterwards the plan's buffers, refreshed by PackPlan.run."""
        if self.sealed:
            f, d = self.bufs[self.cursor]
            self.cursor += 1
            return f, d
        w = weight.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            raise L.SmirkHipError("PackPlan needs contiguous fp32 conv weights")
        if kind == L.PACK_DEPTHWISE:
            C = w.shape[0]
            f, d, n = w.reshape(C, 9).t().contiguous(), None, 9 * C // 8
            job = (w.data_ptr(), f, d, C, 1, 0, 1, kind, 8, n)
        elif kind == L.PACK_STEM:
            c0 = w.shape[0]
            f, d, n = w.permute(0, 2, 3, 1).reshape(c0, 27).contiguous(), None, c0 * 27
            job = (w.data_ptr(), f, d, c0, 3, 0, 3, kind, 8, n)
        elif kind == L.PACK_CONVT2X2:
            ci, co = w.shape[0], w.shape[1]
            f = _split16(w.permute(2, 3, 1, 0).reshape(4 * co, ci).contiguous())
            d = _split16(w.permute(0, 2, 3, 1).reshape(ci, 4 * co).contiguous())
            job = (w.data_ptr(), f, d, ci, co, 0, co, kind, 8, 2 * (4 * co * ci // 8))
        else:
            raise L.SmirkHipError(f"unknown pack job kind {kind}")
        self.jobs.append(job)
        self.bufs.append((f, d))
        return f, d

    def seal(self, params, device):
        arr = (L.SmirkPackJob * len(self.jobs))()
        start = 0
        for j, (wp, f, d, cout, ctot, off, cin_, k, cp, n) in zip(arr, self.jobs):
            j.w, j.fwd, j.dgrad = wp, (f.data_ptr() if f is not None else None), (d.data_ptr() if d is not None else None)
            j.Cout, j.cin_total, j.cin_off, j.Cin, j.KH, j.cin_pad, j.start = cout, ctot, off, cin_, k, cp, start
            start += n
        raw = bytes(arr)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.total, self.sealed = start, True
        self.key = tuple((p.data_ptr(), tuple(p.shape)) for p in params)

    def run(self, ops):
        self.cursor = 0
        L.check(ops.lib.smirk_pack_conv_weights_batch_split16(L.ptr(self.table, torch.uint8), len(self.jobs), self.total, ops.st))


def _to_conv_weight_grad(dw, cout, cin, k=3, cin_real=None):
    """packed [Cout][(ky,kx,ci)] -> nn.Conv2d layout [Cout,Cin,k,k]"""
    g = dw.reshape(cout, k, k, cin).permute(0, 3, 1, 2)
    if cin_real is not None and cin_real < cin:
        g = g[:, :cin_real]
    return g.contiguous()


class GeneratorTrainFunction(torch.autograd.Function):
    """y = SmirkGenerator(x) in train mode; differentiable w.r.t. x and every parameter (passed flat, in `module.parameters()` order)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        x = L.as_f32c(x.detach())
        B, Cx, H, W = x.shape
        if Cx != module.in_channels or H % 16 or W % 16:
            raise L.SmirkHipError("SmirkGenerator: expected [B, in_channels, H, W] with H, W multiples of 16")
        if module.features % 8 or module.in_channels > 8:
            raise L.SmirkHipError("training runs in the split-fp16 mode: init_features % 8 == 0 and in_channels <= 8")
        ctx.arith = getattr(module, "train_arith", "f16x3")
        ops = _Ops(x.device, eval_bn=not module.training, arith=ctx.arith)
        lib, st, f = ops.lib, ops.st, module.features
        ctx.eval_bn = not module.training
        plan = getattr(module, "_pack_plan", None)
        if plan is None or not plan.valid_for(params):                    # first forward (or the parameters were re-allocated): record
            plan = PackPlan()
            object.__setattr__(module, "_pack_plan", plan)
        else:
            plan.run(ops)                                                 # every conv weight of the network: one launch
        ops.plan = plan
        tape = []                                                         # records consumed in reverse by backward()
        xin = torch.empty(B, H, W, 8, device=x.device)
        L.check(lib.smirk_pack_generator_input_split16(L.ptr(x), Cx, None, 0, L.ptr(xin), B, H, W, st))

        def block(seq, tag, x0, x1, h, w, c):
            m = dict(seq.named_children())
            c1, n1, c2, n2 = m[tag + "conv1"], m[tag + "norm1"], m[tag + "conv2"], m[tag + "norm2"]
            c0 = x0.shape[-1]
            if x1 is None:
                wf1, wd1a = ops.pack(c1.weight, cin_pad=c0)
                wd1b = None
            else:                                                          # decoder: forward weight over cat(up, skip), one data-gradient weight per source
                wf1, _ = ops.pack(c1.weight, dgrad=False)
                wd1a = ops.pack(c1.weight, 0, c0, fwd=False)[1]
                wd1b = ops.pack(c1.weight, c0, x1.shape[-1], fwd=False)[1]
            z1, st1 = ops.conv_stats(x0, x1, wf1, B, h, w, c)
            y1, mu1, iv1 = ops.bn_forward(z1, n1, True, stats=st1)
            wf2, wd2 = ops.pack(c2.weight)
            z2, st2 = ops.conv_stats(y1, None, wf2, B, h, w, c)
            y2, mu2, iv2 = ops.bn_forward(z2, n2, True, stats=st2)
            tape.append(("block", (c1, n1, c2, n2), (x0, x1, z1, mu1, iv1, y1, z2, mu2, iv2, wd1a, wd1b, wd2), (h, w, c)))
            return y2

        def pool(t, h, w, c):
            o = torch.empty(B, h // 2, w // 2, c, device=t.device)
            L.check(lib.smirk_maxpool2x2_split16(L.ptr(t), L.ptr(o), B, h, w, c, st))
            tape.append(("pool", None, (t,), (h, w, c)))
            return o

        e1 = block(module.encoder1, "enc1", xin, None, H, W, f)
        e2 = block(module.encoder2, "enc2", pool(e1, H, W, f), None, H // 2, W // 2, 2 * f)
        e3 = block(module.encoder3, "enc3", pool(e2, H // 2, W // 2, 2 * f), None, H // 4, W // 4, 4 * f)
        e4 = block(module.encoder4, "enc4", pool(e3, H // 4, W // 4, 4 * f), None, H // 8, W // 8, 8 * f)
        h16, w16, c16 = H // 16, W // 16, 16 * f
        b = block(module.bottleneck, "bottleneck", pool(e4, H // 8, W // 8, 8 * f), None, h16, w16, c16)
        for rb in module.resnet_blocks:
            cb = rb.conv_block
            wfa, wda = ops.pack(cb[1].weight)
            za, sta = ops.conv_stats(b, None, wfa, B, h16, w16, c16, reflect=True)
            ya, mua, iva = ops.bn_forward(za, cb[2], True, stats=sta)
            wfb, wdb = ops.pack(cb[5].weight)
            zb, stb = ops.conv_stats(ya, None, wfb, B, h16, w16, c16, reflect=True)
            nb, mub, ivb = ops.bn_forward(zb, cb[6], False, residual=b, stats=stb)
            tape.append(("res", (cb[1], cb[2], cb[5], cb[6]), (b, za, mua, iva, ya, zb, mub, ivb, wda, wdb), (h16, w16, c16)))
            b = nb
        d = b
        for lvl, skip, div, c in ((4, e4, 16, 8 * f), (3, e3, 8, 4 * f), (2, e2, 4, 2 * f), (1, e1, 2, f)):
            up = getattr(module, f"upconv{lvl}")
            wup, wupd = plan.request_special(ops, up.weight, L.PACK_CONVT2X2)   # [Cin, Cout, 2, 2] -> forward 1x1 image, data-gradient image (same launch as every other weight)
            u = ops.conv(d, None, wup, B, H // div, W // div, c, k=1, convt=True, shift=up.bias.detach().float().contiguous())
            tape.append(("up", (up,), (d, wupd), (H // div, W // div, 2 * c, c)))
            d = block(getattr(module, f"decoder{lvl}"), f"dec{lvl}", u, skip, 2 * H // div, 2 * W // div, c)
        wf = module.conv.weight.detach().float().reshape(module.out_channels, f).contiguous()
        bf = module.conv.bias.detach().float().contiguous()
        y = torch.empty(B, module.out_channels, H, W, device=x.device)
        L.check(lib.smirk_conv1x1_sigmoid_nchw_split16(L.ptr(d), L.ptr(wf), L.ptr(bf), L.ptr(y), B, H, W, f, module.out_channels, st))
        if not plan.sealed:
            plan.seal(params, x.device)
        ctx.plan, (ctx.plan_generation, ctx.plan_versions) = plan, plan.stamp(params)
        ctx.rm_saved = ops.rm_saved
        ctx.module, ctx.tape, ctx.final = module, tape, (d, wf, y)
        ctx.shape = (B, Cx, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.tape is None:
            raise RuntimeError("GeneratorTrainFunction.backward called a second time: the tape of raw convolution outputs is released after the first "
                               "backward pass (retain_graph=True is not supported by the HIP training path; run the forward again)")
        module, tape, (d1, wf, y) = ctx.module, ctx.tape, ctx.final
        B, Cx, H, W = ctx.shape
        f = module.features
        ctx.plan.check_tape(ctx.plan_generation, ctx.plan_versions)
        ops = _Ops(y.device, eval_bn=ctx.eval_bn, arith=ctx.arith)
        ops.rm_saved = ctx.rm_saved
        lib, st = ops.lib, ops.st
        grads = {}                                                        # id(parameter) -> gradient in the parameter's layout
        # which parameters wanted a gradient WHEN THE FORWARD RAN (smirk_trainer.py:108-113 flips requires_grad off for the eval-mode forward and back on
        # before backward): a frozen generator pays for data gradients only; in eval mode BatchNorm's affine gradients are not computed at all
        wanted = {id(p) for p, n in zip(module.parameters(), ctx.needs_input_grad[2:]) if n}
        if ctx.eval_bn and any(id(bn_p) in wanted for m_ in module.modules() if isinstance(m_, torch.nn.BatchNorm2d) for bn_p in m_.parameters()):
            raise L.SmirkHipError("eval-mode SmirkGenerator inside autograd: parameter gradients are not implemented (freeze the generator as "
                                  "smirk_trainer.py:108-113 does, or call .train())")
        need = lambda p: id(p) in wanted
        want_dx = bool(ctx.needs_input_grad[1])
        gy = L.as_f32c(gy)
        # ---- final 1x1 conv + sigmoid ------------------------------------------------------------------------------------------------------
        dd = torch.empty_like(d1)
        dl8 = torch.empty(B, H, W, 8, device=y.device)
        L.check(lib.smirk_conv1x1_sigmoid_backward_split16(L.ptr(gy), L.ptr(y), L.ptr(wf), L.ptr(dd), L.ptr(dl8), B, H, W, f, module.out_channels, st))
        if need(module.conv.weight):
            grads[id(module.conv.weight)] = ops.wgrad(dl8, d1, B, H, W, 8, f, 1)[:module.out_channels].reshape(module.out_channels, f, 1, 1)
        if need(module.conv.bias):
            grads[id(module.conv.bias)] = ops.colsum(dl8)[:module.out_channels].contiguous()

        def block_backward(rec, g):
            """g = dL/d(block output) -> (dL/d x0, dL/d x1 or None)"""
            _, (c1, n1, c2, n2), (x0, x1, z1, mu1, iv1, y1, z2, mu2, iv2, wd1a, wd1b, wd2), (h, w, c) = rec
            dz2, dg2, db2 = ops.bn_backward(z2, g, n2, mu2, iv2, True)
            if dg2 is not None:
                grads[id(n2.weight)], grads[id(n2.bias)] = dg2, db2
            if need(c2.weight):
                grads[id(c2.weight)] = ops.wgrad_param(dz2, y1, B, h, w, c, c, 3, _grad_like(c2.weight))
            dy1 = ops.conv(dz2, None, wd2, B, h, w, c)
            dz1, dg1, db1 = ops.bn_backward(z1, dy1, n1, mu1, iv1, True)
            if dg1 is not None:
                grads[id(n1.weight)], grads[id(n1.bias)] = dg1, db1
            c0 = x0.shape[-1]
            if x1 is None:
                if need(c1.weight):                                       # (the network input is padded 6 -> 8 channels: the two padded ones are dropped)
                    if c1.weight.shape[1] > c0:
                        raise L.SmirkHipError(f"conv expects {c1.weight.shape[1]} input channels, its source carries {c0}")
                    grads[id(c1.weight)] = ops.wgrad_param(dz1, x0, B, h, w, c, c0, 3, _grad_like(c1.weight), cin_real=c1.weight.shape[1])
                if rec is tape[0] and not want_dx:                        # the network input needs no gradient (cycle path: it is detached)
                    return None, None
                return ops.conv(dz1, None, wd1a, B, h, w, c0), None
            cc1 = x1.shape[-1]
            if need(c1.weight):                                           # torch.cat((up, skip), 1): channels of source 0 first, both into one tensor
                if c0 + cc1 != c1.weight.shape[1]:                        # every channel of the parameter must be written by one of the two calls
                    raise L.SmirkHipError(f"decoder conv expects {c1.weight.shape[1]} input channels, its two sources carry {c0} + {cc1}")
                gw = _grad_like(c1.weight)
                ops.wgrad_param(dz1, x0, B, h, w, c, c0, 3, gw, cin_off=0)
                ops.wgrad_param(dz1, x1, B, h, w, c, cc1, 3, gw, cin_off=c0)
                grads[id(c1.weight)] = gw
            return ops.conv(dz1, None, wd1a, B, h, w, c0), ops.conv(dz1, None, wd1b, B, h, w, cc1)

        g = dd
        skip_grad = {}
        i = len(tape) - 1
        lvl = 1
        while i >= 0:
            rec = tape[i]
            kind = rec[0]
            if kind == "block":
                g0, g1 = block_backward(rec, g)
                if g1 is not None:                                         # decoder block: x0 = up-sampled tensor, x1 = the encoder's skip tensor
                    skip_grad[lvl] = g1
                    lvl += 1
                g = g0
            elif kind == "up":
                (up,), (xin_, wd), (h, w, cin, cout) = rec[1], rec[2], rec[3]
                s2d = torch.empty(B, h, w, 4 * cout, device=y.device)
                L.check(lib.smirk_space_to_depth2_split16(L.ptr(g), L.ptr(s2d), B, h, w, cout, st))
                if need(up.bias):
                    grads[id(up.bias)] = ops.colsum(g)
                if need(up.weight):                                        # packed [Cin][(dy,dx,co)] -> the parameter's [Cin][Cout][2][2] in the reduction itself
                    grads[id(up.weight)] = ops.wgrad_param(xin_, s2d, B, h, w, cin, 4 * cout, 1, _grad_like(up.weight), layout=2)
                g = ops.conv(s2d, None, wd, B, h, w, cin, k=1)
            elif kind == "res":
                _, (ca, na, cbv, nbv), (bin_, za, mua, iva, ya, zb, mub, ivb, wda, wdb), (h, w, c) = rec
                dzb, dgb, dbb = ops.bn_backward(zb, g, nbv, mub, ivb, False)
                if dgb is not None:
                    grads[id(nbv.weight)], grads[id(nbv.bias)] = dgb, dbb
                if need(cbv.weight):
                    grads[id(cbv.weight)] = ops.wgrad_param(dzb, ya, B, h, w, c, c, 3, _grad_like(cbv.weight), reflect=True)
                dpad = ops.conv(dzb, None, wdb, B, h, w, c, pad=2, out_hw=(h + 2, w + 2))
                dya = torch.empty_like(ya)
                L.check(lib.smirk_reflect_pad1_backward_split16(L.ptr(dpad), None, L.ptr(dya), B, h, w, c, st))
                dza, dga, dba = ops.bn_backward(za, dya, na, mua, iva, True)
                if dga is not None:
                    grads[id(na.weight)], grads[id(na.bias)] = dga, dba
                if need(ca.weight):
                    grads[id(ca.weight)] = ops.wgrad_param(dza, bin_, B, h, w, c, c, 3, _grad_like(ca.weight), reflect=True)
                dpad = ops.conv(dza, None, wda, B, h, w, c, pad=2, out_hw=(h + 2, w + 2))
                gin = torch.empty_like(bin_)
                L.check(lib.smirk_reflect_pad1_backward_split16(L.ptr(dpad), L.ptr(g), L.ptr(gin), B, h, w, c, st))     # + the identity branch
                g = gin
            elif kind == "pool":
                (t,), (h, w, c) = rec[2], rec[3]
                level = {f: 1, 2 * f: 2, 4 * f: 3, 8 * f: 4}[c]
                gx = torch.empty_like(t)
                L.check(lib.smirk_maxpool2x2_backward_split16(L.ptr(t), L.ptr(g), L.ptr(skip_grad.get(level), allow_none=True), L.ptr(gx), B, h, w, c, st))
                g = gx
            i -= 1
        dx = split16_to_float(g)[..., :Cx].permute(0, 3, 1, 2).contiguous() if (want_dx and g is not None) else None
        ctx.tape = ctx.final = None
        out = [None, dx]
        for p in module.parameters():
            gp = grads.get(id(p))
            out.append(None if gp is None else gp.reshape(p.shape).to(p.dtype))
        return tuple(out)
