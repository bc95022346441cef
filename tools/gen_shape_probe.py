"""which (in_channels, init_features) does SmirkGenerator serve correctly in each arithmetic mode?  (vs the torch-CPU oracle, 2e-5 on the output)"""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import generator_ref as G
from smirk_amd import SmirkGenerator
warnings.simplefilter("ignore")
for prec in ("f16x3", "f32"):
    for cin, feat in ((6, 8), (6, 12), (6, 16), (6, 20), (6, 24), (6, 40), (6, 48), (6, 64), (3, 32), (8, 32), (10, 32), (16, 32), (10, 16)):
        sd = G.synth_state_dict(in_channels=cin, out_channels=3, features=feat, res_blocks=1, seed=5)
        m = SmirkGenerator(in_channels=cin, out_channels=3, init_features=feat, res_blocks=1)
        m.load_state_dict(sd); m.precision = prec; m = m.cuda().eval()
        x = torch.rand(2, cin, 32, 48, generator=torch.Generator().manual_seed(1))
        y = G.forward(sd, x, res_blocks=1)
        try:
            with torch.no_grad():
                o = m(x.cuda()).cpu()
            print(f"{prec:6s} cin {cin:3d} feat {feat:3d}: max err {float((o - y).abs().max()):.2e}  (ran as {'f16x3' if m._split else 'f32'})", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"{prec:6s} cin {cin:3d} feat {feat:3d}: {type(e).__name__}: {str(e)[:90]}", flush=True)
