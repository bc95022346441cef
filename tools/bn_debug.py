import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import generator_train as T
from smirk_amd.smirk_generator import _split16, split16_to_float
ops = T._Ops(torch.device("cuda"))
for (B, H, W, C) in ((3, 6, 10, 32), (2, 4, 4, 512), (1, 16, 16, 64)):
    for relu in (0, 1):
        g = torch.Generator().manual_seed(C + H)
        bn = torch.nn.BatchNorm2d(C).cuda().train()
        z = torch.randn(B, H, W, C, generator=g) * 2 + 0.7
        dy = torch.randn(B, H, W, C, generator=g)
        zs = _split16(z.reshape(-1, C).cuda()).reshape(B, H, W, C); z64 = split16_to_float(zs).double()
        ds = _split16(dy.reshape(-1, C).cuda()).reshape(B, H, W, C); d64 = split16_to_float(ds).double()
        y, mean, inv = ops.bn_forward(zs, bn, relu)
        dz, dg, db = ops.bn_backward(zs, ds, bn, mean, inv, relu)
        zr = z64.clone().requires_grad_(True)
        pre = F.batch_norm(zr.permute(0, 3, 1, 2), None, None, bn.weight.detach().double(), bn.bias.detach().double(), True, 0.1, 1e-5)
        (F.relu(pre) if relu else pre).backward(d64.permute(0, 3, 1, 2))
        got = split16_to_float(dz).double()
        err = (got - zr.grad).abs()
        i = err.argmax().item()
        print(B, H, W, C, "relu", relu, "max err", err.max().item(), "at", i, "got", got.flatten()[i].item(), "want", zr.grad.flatten()[i].item(),
              "dy there", d64.flatten()[i].item(), "frac > 1e-5:", (err > 1e-5).float().mean().item(), "raw halves", dz.view(torch.float16).flatten()[2 * (i // 8 * 8): 2 * (i // 8 * 8) + 16].tolist())
