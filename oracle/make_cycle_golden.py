"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Cycle-path golden (BASELINE config 5) from the REAL reference classes: src/smirk_generator.py::SmirkGenerator and src/smirk_encoder.py::SmirkEncoder
(its timm backbones come from oracle/mobilenet_ref.py through oracle/sandbox.py: unpinned against timm, see there), both in .train() mode, driven the way
smirk_trainer.py:293-313,365-370 drives them: generator(cat[rendered, masked]) -> encoder -> cycle loss -> backward -> clip_grad_norm_(generator, 0.1).
Pose and shape encoders frozen as config_train.yaml:41-43 has them (freeze_module: requires_grad False, BatchNorm still in train mode).
B = 4, 96 x 96, fp32 and float64 (the arbiter; `spread/...` = how far the reference's own fp32 run is from it), and the same step AT CONFIG 5's OWN SIZE
(B = 64, 224 x 224: 3.2 M pixels per channel in every BatchNorm reduction and weight-gradient K loop) with the reconstruction stored as a strided sub-sample.

    python -m oracle.make_cycle_golden          ->  tests/golden/cycle_golden.npz
    python -m oracle.make_cycle_golden --b64    ->  tests/golden/cycle_golden_b64.npz     (~25 GB of RAM, ~15 min on 8 cores)
"""
import os
import tempfile

import numpy as np
import torch
import torch.nn.functional as F

from . import assets as A
from . import generator_ref as G
from . import mobilenet_ref as M
from . import sandbox as S

GOLD = os.path.join(A.REPO, "tests", "golden")
B, HW = 4, 96
FULL = ("smirk_generator.conv.weight", "smirk_generator.conv.bias", "smirk_generator.encoder1.enc1norm1.weight", "smirk_generator.upconv1.bias",
        "smirk_encoder.expression_encoder.expression_layers.0.bias", "smirk_encoder.expression_encoder.encoder.bn1.weight",
        "smirk_encoder.expression_encoder.encoder.blocks.6.0.bn1.bias")


RECON_SUB = (slice(None, None, 16), slice(None), slice(3, None, 7), slice(5, None, 7))      # what of the B = 64 reconstruction is stored


def inputs(B=B, HW=HW):
    x = A.synth_generator_input(B, seed=71)
    x = (x if HW == 224 else x[:, :, 60:60 + HW, 64:64 + HW]).contiguous()
    g = torch.Generator().manual_seed(72)
    feats = dict(expression_params=torch.randn(B, 50, generator=g), jaw_params=torch.rand(B, 3, generator=g) * torch.tensor([0.5, 0.1, 0.1]),
                 eyelid_params=torch.rand(B, 2, generator=g), shape_params=torch.randn(B, 300, generator=g) * 0.5)
    return x[:, :3].contiguous(), x[:, 3:].contiguous(), feats


def loss_fn(r, t):                                               # smirk_trainer.py:304-313 with use_eyelids, generator not frozen
    return F.mse_loss(r['expression_params'], t['expression_params']) + 10.0 * F.mse_loss(r['jaw_params'], t['jaw_params']) + \
        10.0 * F.mse_loss(r['eyelid_params'], t['eyelid_params']) + F.mse_loss(r['shape_params'], t['shape_params'])


def run(ref, dtype, B=B, HW=HW):
    rendered, masked, feats = inputs(B, HW)
    gen = ref.SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5); gen.load_state_dict(G.synth_state_dict())
    enc = ref.SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict())
    gen, enc = gen.to(dtype).train(), enc.to(dtype).train()
    for m in (enc.pose_encoder, enc.shape_encoder):
        for p in m.parameters():
            p.requires_grad_(False)
    recon = gen(torch.cat([rendered, masked], 1).to(dtype))
    out = enc(recon)
    loss = loss_fn(out, {k: v.to(dtype) for k, v in feats.items()})
    loss.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(gen.parameters(), 0.1)
    grads = {"smirk_generator." + k: p.grad for k, p in gen.named_parameters()}
    grads.update({"smirk_encoder." + k: p.grad for k, p in enc.named_parameters()})
    return dict(recon=recon.detach(), out={k: v.detach() for k, v in out.items()}, loss=loss.item(), gen_norm=float(gnorm), grads=grads)


def main(B=B, HW=HW, name="cycle_golden.npz"):
    assert S.available(), "needs /root/reference"
    d = tempfile.mkdtemp(prefix="smirk_sandbox_")
    A.write_sandbox(d)
    with S.reference(d) as ref:
        r32 = run(ref, torch.float32, B, HW)
        r64 = run(ref, torch.float64, B, HW)
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    big = B * HW * HW > 1 << 20
    recon = r64["recon"][RECON_SUB].float().numpy() if big else r32["recon"].numpy()
    out = dict(recon=recon, loss32=np.float64(r32["loss"]), loss64=np.float64(r64["loss"]), gen_norm32=np.float64(r32["gen_norm"]),
               gen_norm64=np.float64(r64["gen_norm"]))
    for k, v in r64["out"].items():
        out["out64/" + k] = v.float().numpy()
        out["spread/out/" + k] = np.float64((r32["out"][k].double() - v).abs().max())
    gmax = max(float(v.abs().max()) for v in r64["grads"].values() if v is not None)
    for k, v in r64["grads"].items():
        if v is None:
            out["nograd/" + k] = np.int8(1)
            continue
        out["gmax64/" + k] = np.float64(v.abs().max())
        out["gnorm64/" + k] = np.float64(v.norm())               # AFTER clipping for the generator (what the optimiser sees)
        out["ghead64/" + k] = v.flatten()[:32].float().numpy()
        out["spread/" + k] = np.float64(rel(r32["grads"][k], v)) if float(v.abs().max()) > 1e-6 * gmax else np.float64(-1.0)
        if k in FULL:
            out["gfull64/" + k] = v.float().numpy()
    if big:
        out["recon_spread"] = np.float64((r32["recon"].double() - r64["recon"]).abs().max())
    p = os.path.join(GOLD, name)
    np.savez_compressed(p, **out)
    sp = [float(out[k]) for k in out if k.startswith("spread/smirk") and float(out[k]) >= 0]
    print(name, os.path.getsize(p) // 1024, "KiB; loss", r32["loss"], r64["loss"], "generator grad norm", r32["gen_norm"], r64["gen_norm"],
          "; fp32-vs-fp64 gradient spread: median", float(np.median(sp)), "max", max(sp))


if __name__ == "__main__":
    import sys
    if "--b64" in sys.argv:
        main(64, 224, "cycle_golden_b64.npz")
    else:
        main()
