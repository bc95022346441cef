"""GPU parity of the TRAIN-mode SmirkGenerator (batch-statistics BatchNorm, running-stat update, full backward) — BASELINE config 5, generator slice.
Golden = the REAL reference class in .train() mode with autograd (tests/golden/generator_train_golden.npz, oracle/make_train_golden.py);
the functional restatement oracle/generator_ref.py::train_step (pinned to that class) covers other shapes.
Tolerances: forward like inference (2e-5 abs on the sigmoid image); gradients relative to max(1e-?, max|g|) of each tensor — the data path is
split-fp16 x3 (fp32-class) and the weight gradients are exact-fp32 MFMA accumulations, so 2e-4 relative holds with margin; bf16 autocast (what the
reference trainer would use on a GPU) is ~1e-2."""
import os

import numpy as np
import pytest
import torch

from oracle import generator_ref as G
from oracle import make_train_golden as MT

pytestmark = pytest.mark.gpu

OUT_TOL = 2e-5
GRAD_RTOL = 2e-4


def _module(sd):
    from smirk_amd import SmirkGenerator
    m = SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
    m.load_state_dict(sd, strict=True)
    return m.cuda().train()


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def test_train_step_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "generator_train_golden.npz"))
    sd = G.synth_state_dict()
    x, w = MT.inputs()
    m = _module(sd)
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    assert (y.detach().cpu() - torch.from_numpy(g["y"])).abs().max().item() < OUT_TOL
    loss = (y * w.cuda()).sum()
    assert abs(loss.item() - float(g["loss"])) < 1e-3 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    assert _rel(xg.grad.cpu(), torch.from_numpy(g["dx"])) < GRAD_RTOL
    worst = ("", 0.0)
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        gn = float(g["gnorm/" + k])
        e = abs(p.grad.double().norm().item() - gn) / max(gn, 1e-12)
        worst = max(worst, (k, e), key=lambda t: t[1])
        assert e < GRAD_RTOL, (k, "norm", e)
        head = torch.from_numpy(g["ghead/" + k])
        assert (p.grad.flatten()[:64].cpu() - head).abs().max().item() < GRAD_RTOL * max(p.grad.abs().max().item(), 1e-12) , (k, "head")
        if "gfull/" + k in g.files:
            assert _rel(p.grad.cpu(), torch.from_numpy(g["gfull/" + k])) < GRAD_RTOL, (k, "full")
    # running statistics after one step (momentum 0.1, unbiased variance), and the step counter
    for k, b in m.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            ref = torch.from_numpy(g["buf/" + k])
            assert (b.cpu() - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item()), k
        elif k.endswith("num_batches_tracked"):
            assert int(b) == 1, k
    print("worst per-parameter gradient-norm error:", worst)


@pytest.mark.parametrize("B,HW", [(2, 32), (1, 48)])
def test_train_step_matches_oracle_other_shapes(B, HW):
    """odd spatial sizes (bottleneck 2x2 / 3x3: reflection padding mirrors both borders into one row) and batch 1"""
    from oracle import assets as A
    sd = G.synth_state_dict()
    x = A.synth_generator_input(B, seed=61)[:, :, 90:90 + HW, 70:70 + HW].contiguous()
    w = torch.randn(B, 3, HW, HW, generator=torch.Generator().manual_seed(7))
    yr, lr, dxr, gr, br = G.train_step(sd, x, w)
    m = _module(sd)
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    (y * w.cuda()).sum().backward()
    assert (y.detach().cpu() - yr).abs().max().item() < OUT_TOL
    assert _rel(xg.grad.cpu(), dxr) < GRAD_RTOL
    for k, p in m.named_parameters():
        assert _rel(p.grad.cpu(), gr[k]) < GRAD_RTOL, k
    for k, b in m.named_buffers():
        if k in br:
            assert (b.cpu() - br[k]).abs().max().item() < 1e-5 * max(1.0, br[k].abs().max().item()), k


def test_train_mode_under_no_grad_and_eval_roundtrip():
    """train-mode forward without autograd still uses batch statistics and updates the running estimates; .eval() afterwards uses them"""
    from oracle import assets as A
    sd = G.synth_state_dict()
    m = _module(sd)
    x = A.synth_generator_input(2, seed=3)[:, :, 64:128, 64:128].contiguous().cuda()
    before = m.encoder1.enc1norm1.running_mean.clone()
    with torch.no_grad():
        y1 = m(x)
    assert not y1.requires_grad and not torch.equal(before, m.encoder1.enc1norm1.running_mean)
    m.eval()
    with torch.no_grad():
        y2 = m(x)
    sd2 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    assert (y2.cpu() - G.forward(sd2, x.cpu())).abs().max().item() < 5e-5
