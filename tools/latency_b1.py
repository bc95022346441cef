"""Per-frame latency of the whole path at small batch (the demo_video.py regime), eager launches vs a captured hipGraph."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator  # noqa: E402
import synthdata as synth  # noqa: E402
from smirk_amd.pipeline import SmirkPipeline  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sb = tempfile.mkdtemp(); synth.write_sandbox(sb)
cwd = os.getcwd(); os.chdir(sb)
fl, rn = FLAME().cuda(), Renderer().cuda()
os.chdir(cwd)
enc, gen = SmirkEncoder().cuda().eval(), SmirkGenerator(6, 3, 32, 5).cuda().eval()
pipe = SmirkPipeline(enc, fl, rn, gen)
img = synth.synth_images(B, seed=1).cuda()
masked = synth.synth_generator_input(B, seed=1)[:, 3:].contiguous().cuda()


def run(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        out = pipe(img, masked)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


run(5)
print(f"B={B}: eager {run(50):.3f} ms per batch")
try:
    os.environ["SMIRK_ENCODER_SERIAL"] = "1"
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            out = pipe(img, masked)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = pipe(img, masked)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    print(f"B={B}: hipGraph replay {(time.perf_counter() - t) / 50 * 1e3:.3f} ms per batch")
except Exception as e:   # noqa: BLE001
    print("graph capture failed:", repr(e)[:300])
