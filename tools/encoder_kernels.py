"""Per-kernel GPU time of SmirkEncoder.forward with the three backbones on ONE stream (library launch profiler: HIP events around every launch).
    python tools/encoder_kernels.py [B]"""
import os, sys
os.environ["SMIRK_ENCODER_SERIAL"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import SmirkEncoder, _lib as L
import synthdata as synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
enc = SmirkEncoder(); synth.he_init_(enc, seed=1234); enc = enc.cuda().eval()
img = torch.cat([synth.synth_images(min(256, B - i), seed=1 + i).cuda() for i in range(0, B, 256)])
with torch.no_grad():
    for _ in range(2):
        enc(img)
    torch.cuda.synchronize()
    L.profile_start(); enc(img); torch.cuda.synchronize(); recs = L.profile_stop()
per = {}
for name, fl, by, ms in recs:
    a = per.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += ms
tot = sum(v[1] for v in per.values())
for k, v in sorted(per.items(), key=lambda t: -t[1][1]):
    print(f"{k:48s} x{v[0]:<3d} {v[1]:8.3f} ms")
print(f"B={B}: encoder kernel time (serial) {tot:.3f} ms")
