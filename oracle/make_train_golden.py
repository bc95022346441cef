"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Train-mode golden of the REAL reference SmirkGenerator (src/smirk_generator.py, imported from /root/reference through oracle/sandbox.py):
one forward in .train() mode (batch-statistics BatchNorm, running-stat update) and one backward of a fixed linear loss, B = 3, 64 x 64.
Also asserts that the functional restatement oracle/generator_ref.py::train_step reproduces the reference class exactly, and stores the SAME class
run in float64 (`*64` keys) as the arbiter: a whole-network gradient is ill-conditioned at the ReLU / max-pool switching points, so the reference's
own fp32 run sits 0.5 % (dx) to 2 % (some parameters) away from its fp64 run; `ref32_vs_64/...` records that distance per tensor and the GPU test
requires the HIP path to be no further from fp64 than that.

    python -m oracle.make_train_golden      ->  tests/golden/generator_train_golden.npz
"""
import os
import tempfile

import numpy as np
import torch

from . import assets as A
from . import generator_ref as G
from . import sandbox as S

GOLD = os.path.join(A.REPO, "tests", "golden")
SEED_X, SEED_W = 51, 52


def inputs():
    x = A.synth_generator_input(3, seed=SEED_X)[:, :, 80:144, 72:136].contiguous()
    w = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(SEED_W))
    return x, w


def main():
    assert S.available(), "needs /root/reference"
    d = tempfile.mkdtemp(prefix="smirk_sandbox_")
    A.write_sandbox(d)
    sd = G.synth_state_dict()
    x, w = inputs()
    with S.reference(d) as ref:
        g = ref.SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
        g.load_state_dict(sd)
        g.train()
        xr = x.clone().requires_grad_(True)
        y = g(xr)
        loss = (y * w).sum()
        loss.backward()
        grads = {k: p.grad.clone() for k, p in g.named_parameters()}
        bufs = {k: v.clone() for k, v in g.named_buffers()}
    with S.reference(d) as ref:                                  # the arbiter: the same class, same weights, float64
        g64 = ref.SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
        g64.load_state_dict(sd)
        g64 = g64.double().train()
        x64 = x.double().requires_grad_(True)
        y64 = g64(x64)
        (y64 * w.double()).sum().backward()
        grads64 = {k: p.grad.clone() for k, p in g64.named_parameters()}
    # BASELINE config 5 trains under bf16 autocast (train.py / config_train.yaml); the SAME class, same weights and inputs, run under
    # torch.autocast("cpu", torch.bfloat16): its distance to the float64 arbiter is the tolerance a single-MFMA 16-bit mode of the HIP path has to meet
    # (`refbf16_vs_64/...`; the fp32-class path is held to `ref32_vs_64/...`).  fp16 keeps 11 significand bits to bf16's 8, so such a mode sits inside it.
    with S.reference(d) as ref:
        gb = ref.SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
        gb.load_state_dict(sd)
        gb.train()
        xb = x.clone().requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            yb = gb(xb)
        (yb.float() * w).sum().backward()
        gradsb = {k: p.grad.clone() for k, p in gb.named_parameters()}
    # the functional restatement must be the same computation
    y2, loss2, dx2, g2, b2 = G.train_step(sd, x, w)
    assert torch.equal(y2, y.detach()), (y2 - y.detach()).abs().max()
    assert (dx2 - xr.grad).abs().max() <= 1e-6 * xr.grad.abs().max()
    for k in grads:
        assert (g2[k] - grads[k]).abs().max() <= 2e-6 * max(1e-12, grads[k].abs().max()), k
    for k in b2:
        assert torch.equal(b2[k], bufs[k]), k
    out = dict(seed_x=SEED_X, seed_w=SEED_W, y=y.detach().numpy(), loss=np.float64(loss.item()), dx=xr.grad.numpy())
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    out["dx64"] = x64.grad.float().numpy()
    out["ref32_vs_64/dx"] = np.float64(rel(xr.grad, x64.grad))
    out["refbf16_vs_64/dx"] = np.float64(rel(xb.grad, x64.grad))
    out["refbf16_vs_64/y"] = np.float64(rel(yb.detach().float(), y64.detach()))
    out["ref32_vs_64/y"] = np.float64(rel(y.detach(), y64.detach()))
    for k, v in grads64.items():
        out["gnorm64/" + k] = np.float64(v.norm().item())
        out["ghead64/" + k] = v.flatten()[:64].float().numpy()
        out["gmax64/" + k] = np.float64(v.abs().max().item())
        out["ref32_vs_64/" + k] = np.float64(rel(grads[k], v))
        out["refbf16_vs_64/" + k] = np.float64(rel(gradsb[k], v))
        if v.numel() <= 4096:
            out["gfull64/" + k] = v.float().numpy()
    for k, v in grads.items():
        out["gnorm/" + k] = np.float64(v.double().norm().item())
        out["ghead/" + k] = v.flatten()[:64].numpy()
        if v.numel() <= 4096:
            out["gfull/" + k] = v.numpy()
    for k, v in bufs.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["buf/" + k] = v.numpy()
    np.savez_compressed(os.path.join(GOLD, "generator_train_golden.npz"), **out)
    print("generator_train_golden.npz", os.path.getsize(os.path.join(GOLD, "generator_train_golden.npz")) // 1024, "KiB; loss", loss.item(),
          "; reference fp32 vs fp64: dx", out["ref32_vs_64/dx"], "worst parameter", max(float(out["ref32_vs_64/" + k]) for k in grads),
          "; reference under bf16 autocast vs fp64: y", out["refbf16_vs_64/y"], "dx", out["refbf16_vs_64/dx"], "median parameter",
          float(np.median([float(out["refbf16_vs_64/" + k]) for k in grads])), "worst", max(float(out["refbf16_vs_64/" + k]) for k in grads))


if __name__ == "__main__":
    main()
