"""enc1_fused.hip against the three launches it replaces (conv 8->32, conv 32->32, 2x2 max-pool) and the whole generator with / without it.
    python tools/enc1_bench.py [--batches 128,1024] [--iters 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from smirk_amd import _lib as L  # noqa: E402


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="128,1024")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--generator", action="store_true")
    a = ap.parse_args()
    import test_conv_gpu as T
    for B in [int(b) for b in a.batches.split(",")]:
        Lm, lib, t, d, xs, w1s, w2s = T._enc1_case(B, 224, 224, 0)
        fused = timeit(lambda: T._enc1_fused(Lm, lib, d, xs, w1s, w2s, B, 224, 224), a.iters)
        unf = timeit(lambda: T._enc1_unfused(Lm, lib, d, xs, w1s, w2s, B, 224, 224), a.iters)
        fl = 2.0 * B * 224 * 224 * 32 * 360
        by = B * 224 * 224 * (32 + 128 + 32)
        print(f"B={B:5d}  enc1 block: fused {fused:7.3f} ms ({fl / fused / 1e9:6.1f} TFLOP/s, {by / fused / 1e6:6.0f} GB/s algorithmic)   unfused (3 launches) {unf:7.3f} ms", flush=True)
        del d, xs
        if a.generator:
            from smirk_amd import SmirkGenerator
            import synthdata as synth
            gen = SmirkGenerator(6, 3, 32, 5); synth.he_init_(gen, seed=4321); gen = gen.cuda().eval()
            x = torch.rand(B, 6, 224, 224, device="cuda")
            with torch.no_grad():
                on = timeit(lambda: gen(x), 3)
                os.environ["SMIRK_DISABLE_ENC1_FUSED"] = "1"
                off = timeit(lambda: gen(x), 3)
                del os.environ["SMIRK_DISABLE_ENC1_FUSED"]
            print(f"B={B:5d}  generator forward: {on:8.3f} ms with the fused block, {off:8.3f} ms without", flush=True)
            del gen, x
        torch.cuda.empty_cache()
