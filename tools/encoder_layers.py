"""Per-layer GPU time of one MobileNetV3 backbone (events around every pointwise / depthwise launch) + the bytes each launch must move."""
import os, sys
import torch
os.environ["SMIRK_ENCODER_SERIAL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import SmirkEncoder
import synthdata as synth
from smirk_amd.smirk_encoder import MobileNetV3Features as MB
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
enc = SmirkEncoder().cuda().eval()
img = synth.synth_images(B, seed=1).cuda()
rec = []
pw0, dw0 = MB._pointwise, MB._depthwise
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
def pw(self, lib, st, x, pk, relu, residual=None):
    a = ev(); y = pw0(self, lib, st, x, pk, relu, residual); b = ev()
    rec.append(("pw", tuple(x.shape), y.shape[-1], a, b, (x.numel() + y.numel() + (y.numel() if residual is not None else 0)) * 4)); return y
def dw(self, lib, st, x, pk, stride):
    a = ev(); y = dw0(self, lib, st, x, pk, stride); b = ev()
    rec.append((f"dw s{stride}", tuple(x.shape), y.shape[-1], a, b, (x.numel() + y.numel()) * 4)); return y
fb0 = MB._fused_block
def fb(self, lib, st, x, pk, blk):
    a = ev(); y = fb0(self, lib, st, x, pk, blk); b = ev()
    mid = pk["dw"][0].shape[1]
    rec.append((f"{blk.kind} s{blk.stride} mid{mid}", tuple(x.shape), y.shape[-1], a, b, (x.numel() + y.numel()) * 4)); return y
for name in ("shape_encoder", "pose_encoder"):
    bb = getattr(enc, name).encoder
    for _ in range(2): bb(img)
    MB._pointwise, MB._depthwise, MB._fused_block = pw, dw, fb
    rec.clear(); bb(img); torch.cuda.synchronize()
    MB._pointwise, MB._depthwise, MB._fused_block = pw0, dw0, fb0
    tot = 0.0
    print("=====", name)
    for kind, shp, co, a, b, byt in rec:
        t = a.elapsed_time(b); tot += t
        print(f"{kind:14s} {str(shp):24s} -> {co:4d}  {t*1e3:8.1f} us  {byt/1e6:8.1f} MB  {byt/t/1e9:7.2f} TB/s")
    print(f"total {tot:.3f} ms")
