"""Throughput of smirk_amd.VideoPipeline from host uint8 frames to host uint8 grids (PCIe-inclusive), synthetic 720p frames.
    python tools/video_bench.py [n_frames] [batch]"""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synthdata as synth

n, B = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), (int(sys.argv[2]) if len(sys.argv) > 2 else 64)
d = tempfile.mkdtemp(); synth.write_sandbox(d); os.chdir(d)
from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator, VideoPipeline, masking
from oracle import generator_ref as G, mobilenet_ref as M       # synthetic weights only (tools/, not the product path)
fl, rn = FLAME().cuda(), Renderer().cuda()
prob = masking.load_probabilities_per_FLAME_triangle().cuda()
enc = SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict()); enc = enc.cuda().eval()
gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(G.synth_state_dict()); gen = gen.cuda().eval()
H, W = 720, 1280
rng = np.random.default_rng(0)
pool = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(8)]
lm = np.concatenate([rng.uniform(400, 880, (478, 1)), rng.uniform(150, 570, (478, 1)), np.zeros((478, 1))], 1)
frames = lambda: (pool[i % 8] for i in range(n))
lms = lambda: (lm for _ in range(n))
for name, kw in (("crop, 2 panels", dict(crop=True)), ("crop + generator, 3 panels", dict(crop=True, use_smirk_generator=True)),
                 ("crop + generator + render_orig", dict(crop=True, use_smirk_generator=True, render_orig=True))):
    vp = VideoPipeline(enc, fl, rn, gen, prob, batch_size=B, **kw)
    for _ in vp.run((pool[i % 8] for i in range(2 * B)), (lm for _ in range(2 * B))): pass
    torch.cuda.synchronize(); t = time.perf_counter(); cnt = 0
    for g in vp.run(frames(), lms()): cnt += 1
    dt = time.perf_counter() - t
    print(f"{name:34s} batch {B}: {cnt / dt:8.1f} frames/s  ({dt / cnt * 1e3:.3f} ms/frame, {H}x{W} in, {g.shape[0]}x{g.shape[1]} out)")
