"""BASELINE config 5 AT ITS OWN SIZE: 64 frames of 224 x 224 per GPU (round-2 verdict: the training parity ran at B = 4 / 96 x 96 only).

At this size every BatchNorm reduction sums 3.2 M elements per channel, the weight-gradient GEMMs split K = 3.2 M pixels up to 512 ways, the 224 x 224
activations are 822 MB (32-bit buffer offsets close to their limit) — the regime where round 1's intermittent inference bugs lived.  Four checks:

  (i)   the whole cycle step (generator(train) -> encoder(train) -> cycle loss -> backward -> clip_grad_norm_) against the REAL reference classes run on
        the CPU in fp32 and float64 at exactly this size (tests/golden/cycle_golden_b64.npz from `python -m oracle.make_cycle_golden --b64`):
        loss, generator gradient norm, reconstruction sub-sample, re-encoded parameters and every gradient tensor's head + norm (+ 7 tensors in full);
  (ii)  weight / data gradients of single convolutions at (64,224,224,32->32), (64,112,112,64->64), (64,14,14,512->512) against float64;
  (iii) train-mode BatchNorm forward / backward on a (64,224,224,32) tensor against float64;
  (iv)  the three weight-gradient kernel families ($SMIRK_WGRAD_F16 = 0 exact fp32 MFMA, 1 / 2 split-fp16 x3) agree at that size.

float64 references of (ii)-(iv) are evaluated with torch on the GPU (plain matmul / reductions in double precision): test arithmetic, not product code.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import generator_ref as G
from oracle import make_cycle_golden as MC
from oracle import mobilenet_ref as M

pytestmark = pytest.mark.gpu

TOL = 3e-6


def _ops():
    from smirk_amd import generator_train as T
    return T, T._Ops(torch.device("cuda"))


def _act(t):
    """fp32 NHWC (cuda) -> (split16 tensor, its exactly-representable values as float64 NHWC on the GPU)"""
    from smirk_amd.smirk_generator import _split16, split16_to_float
    B, H, W, C = t.shape
    s = _split16(t.reshape(-1, C)).reshape(B, H, W, C)
    return s, split16_to_float(s).double()


def _val(s):
    from smirk_amd.smirk_generator import split16_to_float
    return split16_to_float(s).double()


def _rel(a, b):
    return (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _randn(shape, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, generator=g, device="cuda")


def _wgrad64(d64, x64, k):
    """dW[co][(ky,kx,ci)] in float64 from NHWC operands: one [Cout, BHW] x [BHW, Cin] product per tap (zero padding)"""
    B, H, W, cout = d64.shape
    cin = x64.shape[-1]
    r = (k - 1) // 2
    xp = F.pad(x64, (0, 0, r, r, r, r))
    dm = d64.reshape(-1, cout).t().contiguous()
    out = torch.empty(cout, k, k, cin, dtype=torch.float64, device=d64.device)
    for ky in range(k):
        for kx in range(k):
            out[:, ky, kx] = dm @ xp[:, ky:ky + H, kx:kx + W].reshape(-1, cin)
    return out.reshape(cout, k * k * cin)


def _dgrad64(d64, w64):
    """dL/dx of Conv2d(3x3, pad 1), float64, NHWC: dx[y, x] += d[y - ky + 1, x - kx + 1] . W[:, :, ky, kx]"""
    B, H, W, cout = d64.shape
    cin = w64.shape[1]
    dp = F.pad(d64, (0, 0, 1, 1, 1, 1))
    dx = torch.zeros(B, H, W, cin, dtype=torch.float64, device=d64.device)
    for ky in range(3):
        for kx in range(3):
            dx += (dp[:, 2 - ky:2 - ky + H, 2 - kx:2 - kx + W].reshape(-1, cout) @ w64[:, :, ky, kx]).reshape(B, H, W, cin)
    return dx


@pytest.mark.parametrize("B,H,W,cin,cout", [(64, 224, 224, 32, 32), (64, 112, 112, 64, 64), (64, 14, 14, 512, 512)])
def test_conv_gradients_at_config5_size_all_wgrad_modes(B, H, W, cin, cout):
    """(ii) + (iv): weight gradient in the three kernel families and the data gradient of one 3x3 layer at the benchmarked batch"""
    T, ops = _ops()
    xs, x64 = _act(_randn((B, H, W, cin), 11 + H))
    ds, d64 = _act(_randn((B, H, W, cout), 12 + H))
    want = _wgrad64(d64, x64, 3)
    got = {}
    for mode in (2, 1, 0):
        ops.lib.smirk_conv_wgrad_set_mode(mode)
        try:
            got[mode] = ops.wgrad(ds, xs, B, H, W, cout, cin, 3).double()
        finally:
            ops.lib.smirk_conv_wgrad_set_mode(-1)
        assert torch.isfinite(got[mode]).all()
        assert _rel(got[mode], want) < TOL, ("mode", mode, _rel(got[mode], want))
    assert _rel(got[2], got[0]) < TOL and _rel(got[1], got[0]) < TOL             # (iv) the families agree with each other
    del want, got
    wt = _randn((cout, cin, 3, 3), 13 + H) * (0.5 / (9 * cin) ** 0.5)
    wd = T._pack_dgrad(wt)
    from smirk_amd.smirk_generator import _split16, split16_to_float
    w64 = split16_to_float(_split16(wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()).reshape(1, 1, cout, -1)).reshape(cout, 3, 3, cin) \
        .permute(0, 3, 1, 2).double()
    dx = ops.conv(ds, None, wd, B, H, W, cin)
    assert _rel(_val(dx), _dgrad64(d64, w64)) < TOL


@pytest.mark.parametrize("relu", [True, False])
def test_batchnorm_train_at_config5_size(relu):
    """(iii): (64,224,224,32) — 3.2 M elements per channel through the two-stage fp64-partial reductions, forward and backward"""
    T, ops = _ops()
    B, H, W, C = 64, 224, 224, 32
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g)); bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    rm0, rv0 = bn.running_mean.double().clone(), bn.running_var.double().clone()
    ga, be = bn.weight.detach().double(), bn.bias.detach().double()
    z = _randn((B, H, W, C), 21) * 2 + 0.7
    zs, z64 = _act(z)
    if relu:                                                   # keep pre-activations clear of the ReLU switching point (the op itself is what is tested)
        n = z64.numel() // C
        mean, var = z64.reshape(-1, C).mean(0), z64.reshape(-1, C).var(0, unbiased=False)
        pre = (z64 - mean) * torch.rsqrt(var + 1e-5) * ga + be
        z = z + 0.05 * (pre.abs() < 2e-3).float()
        zs, z64 = _act(z)
        del pre
    dys, dy64 = _act(_randn((B, H, W, C), 22))
    y, mu, inv = ops.bn_forward(zs, bn, relu)
    n = z64.numel() // C
    zf = z64.reshape(-1, C)
    mean, var = zf.mean(0), zf.var(0, unbiased=False)
    istd = torch.rsqrt(var + 1e-5)
    xh = (zf - mean) * istd
    yr = xh * ga + be
    mask = (yr > 0) if relu else torch.ones_like(yr, dtype=torch.bool)
    yr = torch.where(mask, yr, torch.zeros_like(yr))
    e = dict(y=_rel(_val(y).reshape(-1, C), yr), mean=_rel(mu, mean), inv=_rel(inv, istd),
             rm=_rel(bn.running_mean, 0.9 * rm0 + 0.1 * mean), rv=_rel(bn.running_var, 0.9 * rv0 + 0.1 * var * n / (n - 1)))
    dyf = torch.where(mask, dy64.reshape(-1, C), torch.zeros_like(yr))
    db_r, dg_r = dyf.sum(0), (dyf * xh).sum(0)
    dz_r = ga * istd * (dyf - db_r / n - xh * dg_r / n)
    dz, dg, db = ops.bn_backward(zs, dys, bn, mu, inv, relu)
    e.update(dz=_rel(_val(dz).reshape(-1, C), dz_r), dg=_rel(dg, dg_r), db=_rel(db, db_r))
    assert all(v < TOL for v in e.values()), e


def test_cycle_step_at_config5_size_matches_reference_golden(golden_dir):
    """(i): B = 64, 224 x 224 — the step bench.py --workload train64 times, against the real reference classes (fp64 arbiter, fp32 spread as yardstick)"""
    from smirk_amd import SmirkEncoder, SmirkGenerator
    from smirk_amd.cycle import cycle_forward
    g = np.load(os.path.join(golden_dir, "cycle_golden_b64.npz"))
    gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(G.synth_state_dict()); gen = gen.cuda().train()
    enc = SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict()); enc = enc.cuda().train()
    for m in (enc.pose_encoder, enc.shape_encoder):
        for p in m.parameters():
            p.requires_grad_(False)
    rendered, masked, feats = MC.inputs(64, 224)
    loss, recon, out = cycle_forward(gen, enc, rendered.cuda(), masked.cuda(), {k: v.cuda() for k, v in feats.items()})
    rsub = recon.detach()[MC.RECON_SUB].cpu()
    assert (rsub - torch.from_numpy(g["recon"])).abs().max().item() < max(2e-5, 3 * float(g["recon_spread"]))
    for k in out:
        tol = max(2e-4, 3 * float(g["spread/out/" + k]))
        assert (out[k].detach().cpu() - torch.from_numpy(g["out64/" + k])).abs().max().item() < tol, k
    assert abs(loss.item() - float(g["loss64"])) < 3e-4 * abs(float(g["loss64"]))
    loss.backward()
    gnorm = float(torch.nn.utils.clip_grad_norm_(gen.parameters(), 0.1))
    ref_spread = abs(float(g["gen_norm32"]) - float(g["gen_norm64"])) / float(g["gen_norm64"])
    assert abs(gnorm - float(g["gen_norm64"])) / float(g["gen_norm64"]) < max(5e-3, 3 * ref_spread)
    named = [("smirk_generator." + k, p) for k, p in gen.named_parameters()] + [("smirk_encoder." + k, p) for k, p in enc.named_parameters()]
    errs, spreads = {}, {}
    for k, p in named:
        if "nograd/" + k in g.files:
            assert p.grad is None, k
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        sp = float(g["spread/" + k])
        if sp < 0:
            continue
        gmax = float(g["gmax64/" + k])
        e = (p.grad.flatten()[:32].cpu() - torch.from_numpy(g["ghead64/" + k])).abs().max().item() / gmax
        if "gfull64/" + k in g.files:
            e = max(e, (p.grad.cpu() - torch.from_numpy(g["gfull64/" + k])).abs().max().item() / gmax)
        e = max(e, abs(p.grad.double().norm().item() - float(g["gnorm64/" + k])) / max(float(g["gnorm64/" + k]), 1e-30))
        errs[k], spreads[k] = e, sp
    assert len(errs) >= 100
    med, smed = float(np.median(list(errs.values()))), float(np.median(list(spreads.values())))
    worst = max(errs, key=errs.get)
    print(f"cycle step B=64 224^2 vs float64: loss {loss.item():.6f} ({float(g['loss64']):.6f}), generator grad norm {gnorm:.2f} ({float(g['gen_norm64']):.2f}); "
          f"{len(errs)} gradient tensors: median {med:.2e} max {errs[worst]:.2e} [{worst}] (reference fp32: median {smed:.2e} max {max(spreads.values()):.2e})")
    assert med < max(1.5e-2, 3 * smed)
    assert errs[worst] < max(5e-2, 3 * max(spreads.values())), worst
