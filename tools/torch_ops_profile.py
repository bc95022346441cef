"""Which ATen ops (torch-side kernels / copies) does one hot-path step launch?  (run on the GPU box)"""
import os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
sandbox = tempfile.mkdtemp()
mods = bench.build_modules(sandbox, dev)
import synthdata as synth
from smirk_amd.pipeline import SmirkPipeline
enc, flame, rend, gen = mods[:4]
from smirk_amd import masking
cwd = os.getcwd(); os.chdir(sandbox); prob = masking.load_probabilities_per_FLAME_triangle().to(dev); os.chdir(cwd)
pipe = SmirkPipeline(enc, flame, rend, gen, prob)
img = synth.synth_images(128, seed=0).to(dev)
hull = (synth.synth_generator_input(128, seed=1)[:, 3:4] == 0).float().to(dev)
for _ in range(2): pipe(img, hull_mask=hull)
torch.cuda.synchronize()
for name, fn in (("encoder", lambda: enc(img)), ("full step", lambda: pipe(img, hull_mask=hull))):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    print("=====", name)
    print(prof.key_averages().table(sort_by="count", row_limit=18, max_name_column_width=60))
