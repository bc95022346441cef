"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Step-1 chain golden (smirk_trainer.py:37-48,94-104): the loss travels  L1(reconstruction, img) -> SmirkGenerator (train) -> rendered_img -> Renderer backward
-> FLAME backward -> SmirkEncoder (train), all four REAL reference classes from /root/reference through oracle/sandbox.py (timm / pytorch3d are the restated
shims: third-party part unpinned; the rasteriser's barycentrics are differentiable through oracle/render_torch_ref.py).  fp32 and float64 (the arbiter;
`spread/...` = how far the reference's own fp32 run is from it).  B = 2, 224 x 224.

    python -m oracle.make_chain_golden      ->  tests/golden/chain_golden.npz
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
B = int(os.environ.get("SMIRK_CHAIN_B", "2"))
LOSS = os.environ.get("SMIRK_CHAIN_LOSS", "l1")             # smirk_trainer.py:97 uses F.l1_loss; "mse" = a smooth probe of the same chain
FULL = ("joint/vertices", "joint/enc/cam", "joint/enc/pose_params", "joint/enc/expression_params", "joint/enc/shape_params", "joint/enc/jaw_params",
        "joint/enc/eyelid_params", "smirk_encoder.expression_encoder.expression_layers.0.bias", "smirk_encoder.pose_encoder.pose_cam_layers.0.bias",
        "smirk_encoder.shape_encoder.shape_layers.0.bias", "smirk_generator.conv.bias", "smirk_encoder.expression_encoder.encoder.bn1.weight")


def inputs():
    img = A.synth_images(B, seed=81)
    masked = A.synth_generator_input(B, seed=81)[:, 3:].contiguous()
    return img, masked


def run(ref, dtype):
    img, masked = inputs()                      # inputs and synthetic weights are drawn under the float32 default (their generators use the default dtype)
    esd, gsd = M.synth_encoder_state_dict(), G.synth_state_dict()
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)              # the reference FLAME creates torch.eye / zeros in the default dtype (FLAME.py:140-143)
    try:
        return _run(ref, dtype, img, masked, esd, gsd)
    finally:
        torch.set_default_dtype(old)


def _run(ref, dtype, img, masked, esd, gsd):
    enc = ref.SmirkEncoder(); enc.load_state_dict(esd)
    gen = ref.SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5); gen.load_state_dict(gsd)
    flame, rend = ref.FLAME(), ref.Renderer()
    enc, gen, flame, rend = enc.to(dtype).train(), gen.to(dtype).train(), flame.to(dtype), rend.to(dtype)
    flame.dtype = dtype                         # FLAME.py:53 pins self.dtype = torch.float32 and hands it to lbs / batch_rodrigues; the float64 arbiter needs float64
    out = enc(img.to(dtype))
    fl = flame.forward(out)
    r = rend.forward(fl["vertices"], out["cam"])
    recon = gen(torch.cat([r["rendered_img"], masked.to(dtype)], 1))
    loss = F.l1_loss(recon, img.to(dtype)) if LOSS == "l1" else F.mse_loss(recon, img.to(dtype))
    # the gradient at every joint of the chain: after the generator (rendered_img), after the renderer (vertices, cam), after FLAME (the encoder's outputs)
    joints = {"rendered_img": r["rendered_img"], "vertices": fl["vertices"], **{"enc/" + k: v for k, v in out.items()}}
    for t in joints.values():
        t.retain_grad()
    loss.backward()
    grads = {"smirk_generator." + k: p.grad for k, p in gen.named_parameters()}
    grads.update({"smirk_encoder." + k: p.grad for k, p in enc.named_parameters()})
    grads.update({"joint/" + k: (t.grad if t.grad is not None else torch.zeros_like(t)) for k, t in joints.items()})
    return dict(loss=loss.item(), recon=recon.detach(), rendered=r["rendered_img"].detach(), verts=fl["vertices"].detach(),
                out={k: v.detach() for k, v in out.items()}, grads=grads)


def main():
    assert S.available(), "needs /root/reference"
    d = tempfile.mkdtemp(prefix="smirk_sandbox_")
    A.write_sandbox(d)
    with S.reference(d) as ref:
        r32 = run(ref, torch.float32)
        r64 = run(ref, torch.float64)
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    out = dict(loss32=np.float64(r32["loss"]), loss64=np.float64(r64["loss"]), recon=r64["recon"][:, :, ::4, ::4].float().numpy(),
               recon_spread=np.float64((r32["recon"].double() - r64["recon"]).abs().max()),
               coverage=np.float64((r64["rendered"][:, 0] != 0).double().mean()),
               pixels_differ32=np.float64(((r32["rendered"] != 0) != (r64["rendered"] != 0)).double().mean()))
    for k, v in r64["out"].items():
        out["out64/" + k] = v.float().numpy()
        out["spread/out/" + k] = np.float64((r32["out"][k].double() - v).abs().max())
    gmax = max(float(v.abs().max()) for v in r64["grads"].values() if v is not None)
    for k, v in r64["grads"].items():
        if v is None:
            out["nograd/" + k] = np.int8(1)
            continue
        out["gmax64/" + k] = np.float64(v.abs().max())
        out["gnorm64/" + k] = np.float64(v.norm())
        out["ghead64/" + k] = v.flatten()[:32].float().numpy()
        out["spread/" + k] = np.float64(rel(r32["grads"][k], v)) if float(v.abs().max()) > 1e-6 * gmax else np.float64(-1.0)
        # the max-norm spread is set by single switching events (one ReLU / max-pool / pixel-owner flip moves one element by O(max)); the L2 spread says how
        # much of the tensor's ENERGY differs
        out["l2spread/" + k] = np.float64(float((r32["grads"][k].double() - v).norm() / v.norm().clamp_min(1e-300)))
        if k in FULL:
            out["gfull64/" + k] = v.float().numpy()
    p = os.path.join(GOLD, "chain_golden.npz" if (LOSS, B) == ("l1", 2) else f"chain_golden_{LOSS}_b{B}.npz")
    out["gfull64/joint/rendered_img"] = r64["grads"]["joint/rendered_img"][:, :, ::4, ::4].float().numpy()      # strided sub-sample of the image gradient
    np.savez_compressed(p, **out)
    for k in sorted(out):
        if k.startswith("spread/joint"):
            print("   ", k, float(out[k]), "L2:", float(out["l2" + k]))
    sp = [float(out[k]) for k in out if k.startswith("spread/smirk") and float(out[k]) >= 0]
    l2 = [float(out[k]) for k in out if k.startswith("l2spread/smirk")]
    print("L2 spread of the parameter gradients: median", float(np.median(l2)), "max", max(l2))
    print(os.path.basename(p), os.path.getsize(p) // 1024, "KiB; loss", r32["loss"], r64["loss"], "coverage", float(out["coverage"]),
          "fp32-vs-fp64 pixels re-assigned", float(out["pixels_differ32"]), "; gradient spread: median", float(np.median(sp)), "max", max(sp), "tensors", len(sp))


if __name__ == "__main__":
    main()
