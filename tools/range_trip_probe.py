"""Which phase of a tiny train-mode cycle step (B = 2, 64 x 64: tests/test_cycle_gpu.py::test_graphed_cycle_modules_replay_equals_eager) trips the split-fp16 range flag?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import generator_ref as G, mobilenet_ref as M
from smirk_amd import SmirkEncoder, SmirkGenerator, _lib as L
from smirk_amd.cycle import cycle_forward
from smirk_amd.smirk_generator import split16_to_float
import smirk_amd.generator_train as GT

def flag(tag):
    torch.cuda.synchronize()
    f = L.lib().smirk_range_flag_peek(); L.lib().smirk_range_flag_clear()
    print(f"{tag:40s} flag={f}", flush=True)
    return f

torch.manual_seed(0)
gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(G.synth_state_dict()); gen = gen.cuda().train()
enc = SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict()); enc = enc.cuda().train()
for B, H in ((2, 64), (2, 224), (8, 224)):
    g = torch.Generator().manual_seed(11)
    feats = {"expression_params": torch.randn(B, 50, generator=g) * 0.5, "jaw_params": torch.rand(B, 3, generator=g) * 0.1,
             "eyelid_params": torch.rand(B, 2, generator=g), "shape_params": torch.randn(B, 300, generator=g) * 0.5}
    feats = {k: v.cuda() for k, v in feats.items()}
    r, m = torch.rand(B, 3, H, H, generator=g).cuda(), torch.rand(B, 3, H, H, generator=g).cuda()
    flag(f"B={B} H={H} start")
    # instrument: wrap _Ops methods to check the flag after each call
    names = ("conv", "bn_forward", "bn_backward", "wgrad", "wgrad_param", "colsum")
    orig = {n: getattr(GT._Ops, n) for n in names}
    seen = []
    def wrap(n):
        def f(self, *a, **k):
            out = orig[n](self, *a, **k)
            torch.cuda.synchronize()
            if L.lib().smirk_range_flag_peek():
                L.lib().smirk_range_flag_clear()
                shp = [tuple(t.shape) for t in a if torch.is_tensor(t)][:2]
                seen.append((n, shp))
            return out
        return f
    for n in names:
        setattr(GT._Ops, n, wrap(n))
    try:
        loss, recon, _ = cycle_forward(gen, enc, r, m, feats)
        flag("after forward")
        loss.backward()
        flag("after backward")
    finally:
        for n in names:
            setattr(GT._Ops, n, orig[n])
    print("  tripping ops:", seen[:12], "..." if len(seen) > 12 else "", len(seen), flush=True)
