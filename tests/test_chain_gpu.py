"""Step-1 chain of the reference trainer (smirk_trainer.py:37-48,94-104) through the four HIP modules in ONE autograd graph:

    L1(reconstruction, img) -> SmirkGenerator (train) -> rendered_img -> Renderer backward -> FLAME backward -> SmirkEncoder (train)

Golden = the four REAL reference classes run that way in fp32 and float64 (tests/golden/chain_golden.npz, oracle/make_chain_golden.py).

What can and cannot be asserted.  The forward quantities (loss, reconstruction, re-encoded parameters) are tight.  The GRADIENTS of this chain are not a
well-conditioned function of the arithmetic: on this input the reference's own fp32 run differs from its own float64 run by 63 % (L2) already at the first
joint (dL/d rendered_img: 2 x 2 max-pool routing and ReLU switching inside the train-mode U-Net at batch 2, then 1/area barycentric gradients), 55 % median
over the 437 parameter tensors — the golden records it per tensor.  So the chain is pinned in two ways:
  1. WIRING, exactly: the gradients of the one-graph chain equal, bit for bit, the gradients obtained by cutting the chain at every joint (generator |
     renderer | FLAME | encoder) and feeding each stage the previous stage's gradient at identical forward values — every stage's backward is pinned on its
     own against float64 / reference goldens elsewhere (tests/test_generator_train_gpu.py, test_render_gpu.py, test_flame_gpu.py, test_encoder_train_gpu.py);
  2. SANITY against the float64 arbiter at the reference's own spread: every joint gradient within 3x the reference's fp32-vs-float64 L2 spread and
     positively correlated with it (a sign error, a dropped term or a factor-of-10 scale would fail), parameter gradients likewise in the median.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import generator_ref as G
from oracle import make_chain_golden as MC
from oracle import mobilenet_ref as M

pytestmark = pytest.mark.gpu


def _modules(sandbox):
    from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator
    cwd = os.getcwd(); os.chdir(sandbox)
    try:
        fl, rn = FLAME().cuda(), Renderer().cuda()
    finally:
        os.chdir(cwd)
    enc = SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict()); enc = enc.cuda().train()
    gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(G.synth_state_dict()); gen = gen.cuda().train()
    return enc, fl, rn, gen


def _l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))


def test_step1_chain_forward_and_gradients(sandbox, golden_dir):
    g = np.load(os.path.join(golden_dir, "chain_golden.npz"))
    enc, fl, rn, gen = _modules(sandbox)
    img, masked = MC.inputs()
    img, masked = img.cuda(), masked.cuda()
    enc0 = {k: v.clone() for k, v in enc.state_dict().items()}
    gen0 = {k: v.clone() for k, v in gen.state_dict().items()}

    # ---- (A) the chain in one autograd graph ---------------------------------------------------------------------------------------------------------
    out = enc(img)
    flo = fl.forward(out)
    ro = rn.forward(flo["vertices"], out["cam"])
    recon = gen(torch.cat([ro["rendered_img"], masked], 1))
    loss = F.l1_loss(recon, img)
    joints = {"rendered_img": ro["rendered_img"], "vertices": flo["vertices"], **{"enc/" + k: v for k, v in out.items()}}
    for t in joints.values():
        t.retain_grad()
    loss.backward()
    torch.cuda.synchronize()

    # forward quantities: tight
    assert abs(loss.item() - float(g["loss64"])) < 3e-4 * float(g["loss64"])
    assert (recon.detach()[:, :, ::4, ::4].cpu() - torch.from_numpy(g["recon"])).abs().max().item() < max(2e-5, 3 * float(g["recon_spread"]))
    for k, v in out.items():
        assert (v.detach().cpu() - torch.from_numpy(g["out64/" + k])).abs().max().item() < max(2e-4, 3 * float(g["spread/out/" + k])), k
    chain = {"joint/" + k: (t.grad.clone() if t.grad is not None else torch.zeros_like(t)) for k, t in joints.items()}
    pg = {"smirk_generator." + k: p.grad.clone() for k, p in gen.named_parameters()}
    pg.update({"smirk_encoder." + k: p.grad.clone() for k, p in enc.named_parameters()})

    # ---- (B) the same chain cut at every joint, at identical forward values ------------------------------------------------------------------------------
    enc.load_state_dict(enc0, strict=True); gen.load_state_dict(gen0, strict=True)      # BatchNorm running statistics back to their values before (A)
    for p in list(enc.parameters()) + list(gen.parameters()):
        p.grad = None
    out2 = enc(img)
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in out2.items()}
    flo2 = fl.forward(leaves)
    vleaf = flo2["vertices"].detach().clone().requires_grad_(True)
    cleaf = leaves["cam"].detach().clone().requires_grad_(True)
    ro2 = rn.forward(vleaf, cleaf)
    rleaf = ro2["rendered_img"].detach().clone().requires_grad_(True)
    recon2 = gen(torch.cat([rleaf, masked], 1))
    assert torch.equal(recon2, recon) and torch.equal(ro2["rendered_img"], ro["rendered_img"]) and torch.equal(flo2["vertices"], flo["vertices"])
    F.l1_loss(recon2, img).backward()                                     # generator: parameter gradients + dL/d rendered_img
    ro2["rendered_img"].backward(rleaf.grad)                              # renderer: dL/d vertices, dL/d cam (image path)
    flo2["vertices"].backward(vleaf.grad)                                 # FLAME: dL/d encoder outputs
    genc = {k: (leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])) for k in leaves}
    genc["cam"] = genc["cam"] + cleaf.grad
    torch.autograd.backward([out2[k] for k in out2], [genc[k] for k in out2])       # encoder: parameter gradients
    torch.cuda.synchronize()
    assert torch.equal(rleaf.grad, chain["joint/rendered_img"]) and torch.equal(vleaf.grad, chain["joint/vertices"])
    for k in out2:
        assert torch.equal(genc[k], chain["joint/enc/" + k]), k
    for k, p in [("smirk_generator." + k, p) for k, p in gen.named_parameters()] + [("smirk_encoder." + k, p) for k, p in enc.named_parameters()]:
        assert torch.equal(p.grad, pg[k]), k

    # ---- (C) sanity against the float64 arbiter at the reference's own spread -------------------------------------------------------------------------------
    report = {}
    for k, t in chain.items():
        key = "gfull64/" + k
        ref = torch.from_numpy(g[key])
        got = t.cpu() if k != "joint/rendered_img" else t.cpu()[:, :, ::4, ::4]
        e, c, sp = _l2(got, ref), _cos(got, ref), float(g["l2spread/" + k])
        report[k] = (e, c, sp)
        assert e < max(0.1, 3 * sp) and c > 0.3, (k, e, c, sp)
    errs, sps = [], []
    for k, v in pg.items():
        if "nograd/" + k in g.files or float(g["spread/" + k]) < 0:
            continue
        gn, rn64 = v.double().norm().item(), float(g["gnorm64/" + k])
        errs.append(abs(gn - rn64) / max(rn64, 1e-30)); sps.append(float(g["l2spread/" + k]))
    print("step-1 chain: loss", loss.item(), "(float64", float(g["loss64"]), "); joint gradients (L2 err, cosine, reference fp32 L2 spread):",
          {k: tuple(round(x, 3) for x in v) for k, v in report.items()}, f"; parameter gradient norms: median rel. err {np.median(errs):.3f} "
          f"(reference fp32 L2 spread median {np.median(sps):.3f}) over {len(errs)} tensors")
    assert np.median(errs) < max(0.1, 3 * float(np.median(sps)))
