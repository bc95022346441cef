"""Run only the SmirkEncoder (serial, one stream) so rocprofv3 --kernel-trace shows per-layer kernel durations."""
import os
import sys

import torch

os.environ["SMIRK_ENCODER_SERIAL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import SmirkEncoder  # noqa: E402
import synthdata as synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
enc = SmirkEncoder().cuda().eval()
img = synth.synth_images(B, seed=1).cuda()
for _ in range(3):
    out = enc(img)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    out = enc(img)
e1.record(); torch.cuda.synchronize()
print(f"encoder serial: {e0.elapsed_time(e1) / 5:.3f} ms per batch of {B}")
