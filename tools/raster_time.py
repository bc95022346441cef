"""raster_tile timing under ablation ($SMIRK_RASTER_ABLATE is read once per process: run this script once per setting).  GPU box only."""
import os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import FLAME, Renderer
import synthdata as synth
from smirk_amd import _lib as L
sb = tempfile.mkdtemp(); synth.write_sandbox(sb)
cwd = os.getcwd(); os.chdir(sb)
fl, rn = FLAME().cuda(), Renderer().cuda()
os.chdir(cwd)
B = 167
p = synth.synth_flame_params(B, seed=3); p["shape_params"] *= 0.4
pg = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
cam = torch.from_numpy(synth.synth_cam(B, seed=3)).cuda()
with torch.no_grad():
    v = fl.forward(pg)["vertices"]
    rn.forward(v, cam)
    torch.cuda.synchronize()
    L.profile_start()
    for _ in range(3):
        o = rn.forward(v, cam)
    torch.cuda.synchronize()
    recs = L.profile_stop()
per = {}
for name, fl_, by, ms in recs:
    per.setdefault(name, []).append(ms)
print("ablate", os.environ.get("SMIRK_RASTER_ABLATE", "0"), {k: round(sum(v) / len(v), 4) for k, v in per.items()}, "coverage", float((o["rendered_img"][:, 0] != 0).float().mean()))
