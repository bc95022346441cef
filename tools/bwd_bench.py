"""FLAME + Renderer forward/backward timings at batch B (run on the GPU box; optionally under rocprofv3 --kernel-trace --stats)."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synthdata as synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
d = tempfile.mkdtemp(); synth.write_sandbox(d); os.chdir(d)
from smirk_amd import FLAME, Renderer
fl, rn = FLAME().cuda(), Renderer().cuda()
p = synth.synth_flame_params(B, seed=1)
cam = torch.from_numpy(synth.synth_cam(B, seed=1)).cuda()
tp = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
gi = torch.randn(B, 3, 224, 224, device="cuda")

def run(grad):
    for t in tp.values(): t.requires_grad_(grad); t.grad = None
    cam.requires_grad_(grad); cam.grad = None
    fo = fl.forward(tp)
    ro = rn.forward(fo["vertices"], cam, landmarks_fan=fo["landmarks_fan"], landmarks_mp=fo["landmarks_mp"])
    if grad:
        torch.autograd.backward([ro["rendered_img"], ro["landmarks_fan"], ro["landmarks_mp"]],
                                [gi, torch.ones_like(ro["landmarks_fan"]), torch.ones_like(ro["landmarks_mp"])])

for grad in (False, True):
    for _ in range(3): run(grad)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): run(grad)
    torch.cuda.synchronize()
    print(f"B={B} FLAME+Renderer {'forward+backward' if grad else 'forward only'}: {(time.perf_counter()-t)/20*1e3:.3f} ms")
