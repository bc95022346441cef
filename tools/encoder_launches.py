"""Every launch of one SmirkEncoder forward (B frames), in order, with its duration (library launch profiler).  GPU box only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import SmirkEncoder
import synthdata as synth
from smirk_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 167
os.environ["SMIRK_ENCODER_SERIAL"] = "1"
enc = SmirkEncoder(); synth.he_init_(enc, seed=1); enc = enc.cuda().eval()
img = synth.synth_images(min(B, 64), seed=1).cuda()
img = img.repeat((B + img.shape[0] - 1) // img.shape[0], 1, 1, 1)[:B].contiguous()
with torch.no_grad():
    enc(img); torch.cuda.synchronize()
    L.profile_start(); enc(img); torch.cuda.synchronize(); recs = L.profile_stop()
tot = 0.0
for i, (name, fl, by, ms) in enumerate(recs):
    tot += ms
    print(f"{i:3d} {name[:44]:44s} {ms*1e3:8.1f} us  {fl/ms/1e9 if ms else 0:7.1f} TF  {by/ms/1e6 if ms else 0:8.1f} GB/s  {by/1e6:8.1f} MB")
print("total ms", tot)
