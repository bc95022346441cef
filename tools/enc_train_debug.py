import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import assets as A
from oracle import mobilenet_ref as M
from smirk_amd import SmirkEncoder
esd = M.synth_encoder_state_dict()
enc = SmirkEncoder(); enc.load_state_dict(esd, strict=True); enc = enc.cuda().train()
for freeze in (False, True):
    for p in enc.pose_encoder.parameters():
        p.requires_grad_(not freeze)
    for mode in ("sum", "mse"):
        img = A.synth_images(2, seed=1)[:, :, :96, :96].contiguous().cuda().requires_grad_(True)
        out = enc(img)
        keys = ("expression_params", "jaw_params", "eyelid_params", "shape_params")
        if mode == "sum":
            loss = sum(out[k].sum() for k in keys)
        else:
            loss = sum(F.mse_loss(out[k], torch.zeros_like(out[k])) for k in keys)
        loss.backward()
        print(freeze, mode, "img.grad", None if img.grad is None else img.grad.abs().max().item(), {k: out[k].requires_grad for k in out})
