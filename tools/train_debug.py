"""Per-tensor gradient error of the train-mode generator against the float64 oracle, next to the fp32 oracle's own distance (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import assets as A
from oracle import generator_ref as G
from oracle import make_train_golden as MT
from smirk_amd import SmirkGenerator

sd = G.synth_state_dict()
rel = lambda a, b: (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
cases = [("golden", *MT.inputs())]
for B, HW in ((2, 32), (1, 48), (4, 64)):
    cases.append((f"B{B} {HW}", A.synth_generator_input(B, seed=61)[:, :, 90:90 + HW, 70:70 + HW].contiguous(),
                  torch.randn(B, 3, HW, HW, generator=torch.Generator().manual_seed(7))))
for name, x, w in cases:
    y32, l32, dx32, g32, b32 = G.train_step(sd, x, w)
    y64, l64, dx64, g64, b64 = G.train_step(sd, x, w, dtype=torch.float64)
    m = SmirkGenerator(6, 3, 32, 5); m.load_state_dict(sd); m = m.cuda().train()
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    (y * w.cuda()).sum().backward()
    print(f"== {name}: y mine-f64 {(y.detach().cpu().double() - y64).abs().max().item():.2e} ref32-f64 {(y32.double() - y64).abs().max().item():.2e}; "
          f"dx mine-f64 {rel(xg.grad.cpu(), dx64):.2e} ref32-f64 {rel(dx32, dx64):.2e}")
    mine = [rel(p.grad.cpu(), g64[k]) for k, p in m.named_parameters()]
    ref = [rel(g32[k], g64[k]) for k, p in m.named_parameters()]
    t = lambda v: torch.tensor(v)
    print(f"   params: mine-f64 median {t(mine).median():.2e} max {t(mine).max():.2e} | ref32-f64 median {t(ref).median():.2e} max {t(ref).max():.2e}")
    if "-v" in sys.argv:
        for (k, p), a, b in zip(m.named_parameters(), mine, ref):
            print(f"   {k:45s} mine {a:.2e} ref32 {b:.2e}")
