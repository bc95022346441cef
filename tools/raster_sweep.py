"""Renderer forward (and backward) time vs camera scale / parameter amplitude (run on the GPU box)."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synthdata as synth
B = 128
d = tempfile.mkdtemp(); synth.write_sandbox(d); os.chdir(d)
from smirk_amd import FLAME, Renderer
fl, rn = FLAME().cuda(), Renderer().cuda()
gi = torch.randn(B, 3, 224, 224, device="cuda")
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for amp in (0.0, 0.3, 1.0):
    p = synth.synth_flame_params(B, seed=1)
    tp = {k: torch.from_numpy(v * amp).cuda() for k, v in p.items()}
    with torch.no_grad():
        verts = fl.forward(tp)["vertices"]
    for s in (3.0, 6.0, 9.0):
        cam = torch.tensor([[s, 0.0, 0.02]], device="cuda").repeat(B, 1)
        with torch.no_grad():
            cov = (rn.forward(verts, cam)["rendered_img"][:, 0] > 0).float().mean().item()
            tf = t(lambda: rn.forward(verts, cam))
        v2, c2 = verts.clone().requires_grad_(True), cam.clone().requires_grad_(True)
        def fb():
            v2.grad = None; c2.grad = None
            rn.forward(v2, c2)["rendered_img"].backward(gi)
        tb = t(fb)
        print(f"amp={amp} scale={s}: coverage {cov:.3f}  forward {tf:.3f} ms  forward+backward {tb:.3f} ms")
