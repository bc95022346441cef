"""BatchNorm statistics / backward-sum kernels (colsum_stage1<0|1>): rows in flight per thread x stage-1 blocks, on the generator's and the backbones' tensor shapes at 64 frames.

(The two sweep switches $SMIRK_COLSUM_U / $SMIRK_COLSUM_BLOCKS existed for this measurement only — commit "colsum rows-in-flight / blocks as sweep switches" — and were removed
after it: profiles/r06_bn_sweep.txt.  Without them this tool times the shipped configuration, U = 4 and <= 512 blocks, in every row.)

    python tools/bn_sweep.py > gpurun_out/r06_bn_sweep.txt
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smirk_amd import generator_train as T, _lib as L

dev = torch.device("cuda")
shapes = [(64, 224, 224, 32), (64, 112, 112, 64), (64, 56, 56, 128), (64, 28, 28, 256), (64, 14, 14, 512), (64, 14, 14, 672), (64, 7, 7, 960)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


print("# us per call (HIP events, 20 calls): BatchNorm forward = stats + finalize + apply; backward = sums + stage 2 + apply.  Columns: U (rows in flight) / blocks cap")
for shp in shapes:
    B, H, W, C = shp
    z = torch.randn(B, H, W, C, device=dev)         # bit patterns are irrelevant for timing (any float32 words are valid split16 storage)
    dy = torch.randn(B, H, W, C, device=dev)
    bn = torch.nn.BatchNorm2d(C).to(dev).train()
    ops = T._Ops(dev)
    y, mu, iv = ops.bn_forward(z, bn, True)
    nbytes = z.numel() * 4
    print(f"shape {shp}  ({nbytes / 2**20:.0f} MB per tensor)")
    for U in (4, 2, 1):
        for blocks in (512, 1024, 2048, 4096):
            os.environ["SMIRK_COLSUM_U"], os.environ["SMIRK_COLSUM_BLOCKS"] = str(U), str(blocks)
            tf = timeit(lambda: ops.bn_forward(z, bn, True))
            tb = timeit(lambda: ops.bn_backward(z, dy, bn, mu, iv, True))
            print(f"   U={U} blocks={blocks:5d}   forward {tf:8.1f} us ({3 * nbytes / tf / 1e6:6.2f} TB/s over 3 passes)   backward {tb:8.1f} us ({5 * nbytes / tb / 1e6:6.2f} TB/s over 5 passes)")
    del z, dy, y
    torch.cuda.empty_cache()
