"""The launch sequence of ONE backbone (in order) with each launch's GPU time, the three backbones on one stream: which kernels make up the critical chain of
SmirkEncoder.forward at a given batch.    python tools/encoder_chain.py [B] [pose|shape|expression]"""
import os, sys
os.environ["SMIRK_ENCODER_SERIAL"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import SmirkEncoder, _lib as L
import synthdata as synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
which = sys.argv[2] if len(sys.argv) > 2 else "expression"
enc = SmirkEncoder(); synth.he_init_(enc, seed=1234); enc = enc.cuda().eval()
img = synth.synth_images(B, seed=1).cuda()
bb = getattr(enc, which + "_encoder").encoder
with torch.no_grad():
    for _ in range(3):
        bb(img)
    torch.cuda.synchronize()
    L.profile_start(); bb(img); torch.cuda.synchronize(); recs = L.profile_stop()
tot = 0.0
for name, fl, by, ms in recs:
    tot += ms
    print(f"{name:52s} {ms * 1e3:8.1f} us   {fl / max(ms, 1e-9) / 1e9:8.1f} TFLOP/s  {by / max(ms, 1e-9) / 1e6:8.1f} GB/s")
print(f"B={B} {which}: {len(recs)} launches, {tot:.3f} ms of kernel time")
