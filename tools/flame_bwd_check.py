"""Prints gradient errors of smirk_flame_backward vs the f64 oracle and forward/backward timings (run on the GPU box)."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import assets as A
from oracle.flame_torch_ref import FlameTorchRef, scalar_loss

d = tempfile.mkdtemp(); A.write_sandbox(d); os.chdir(d)
from smirk_amd import FLAME
fl = FLAME().cuda()
for B in (4, 128):
    p = A.synth_flame_params(B, seed=B)
    tp64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in p.items()}
    o = FlameTorchRef(d, dtype=torch.float64)(tp64)
    _, ws = scalar_loss({k: v.detach().float() for k, v in o.items()}, seed=1)
    sum((o[k] * ws[k].double()).sum() for k in ws).backward()
    tp32 = {k: torch.from_numpy(v).requires_grad_(True) for k, v in p.items()}
    o32 = FlameTorchRef(d)(tp32); sum((o32[k] * ws[k]).sum() for k in ws).backward()
    tg = {k: torch.from_numpy(v).cuda().requires_grad_(True) for k, v in p.items()}
    og = fl.forward(tg); sum((og[k] * ws[k].cuda()).sum() for k in ws).backward()
    for k in p:
        r = tp64[k].grad.numpy(); s = max(1.0, np.abs(r).max())
        print(f"B={B} {k:18s} |g|max={np.abs(r).max():9.3f} hip_err={np.abs(tg[k].grad.cpu().numpy()-r).max()/s:.2e} torch_cpu_f32_err={np.abs(tp32[k].grad.numpy()-r).max()/s:.2e}")
for B in (128, 512):
    p = A.synth_flame_params(B, seed=B)
    tg = {k: torch.from_numpy(v).cuda().requires_grad_(True) for k, v in p.items()}
    gw = None
    def step():
        global gw
        og = fl.forward(tg)
        if gw is None: gw = {k: torch.randn_like(v) for k, v in og.items()}
        torch.autograd.backward(list(og.values()), [gw[k] for k in og])
    with torch.no_grad():
        for _ in range(3): fl.forward(tg)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): fl.forward(tg)
        torch.cuda.synchronize(); tf = (time.perf_counter() - t) / 20
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); tb = (time.perf_counter() - t) / 20
    print(f"B={B}: forward {tf*1e3:.3f} ms, forward+backward {tb*1e3:.3f} ms")
