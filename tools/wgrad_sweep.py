"""Per-layer timing of the convolution weight gradient (smirk_conv_wgrad_f32) at the train64 sizes.  The kernel family is chosen by $SMIRK_WGRAD_F16
(read once per process): 0 = exact-fp32 MFMA, 1 / 2 = split-fp16 x3 with LDS transpose reads (1 / 2 chunks per barrier).  Usage: python tools/wgrad_sweep.py [B [H,Cout,Cin,k]]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ONLY = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else None          # "H,Cout,Cin,k": one layer (PMC runs)
lib = L.lib()
st = L.stream_ptr()
# (H, Cout, Cin, KH, reflect): SmirkGenerator layers (U-Net levels + bottleneck/res blocks + ConvT as 1x1 over space-to-depth) and two encoder pointwise shapes
layers = [(224, 32, 8, 3, 0), (224, 32, 32, 3, 0), (112, 64, 32, 3, 0), (112, 64, 64, 3, 0), (56, 128, 64, 3, 0), (56, 128, 128, 3, 0), (28, 256, 128, 3, 0),
          (28, 256, 256, 3, 0), (14, 512, 256, 3, 0), (14, 512, 512, 3, 0), (14, 512, 512, 3, 1), (28, 256, 512, 3, 0), (56, 128, 256, 3, 0), (112, 64, 128, 3, 0),
          (224, 32, 64, 3, 0), (14, 512, 1024, 1, 0), (28, 256, 512, 1, 0), (56, 128, 256, 1, 0), (112, 64, 128, 1, 0), (14, 960, 160, 1, 0), (28, 120, 40, 1, 0)]
g = torch.Generator(device="cuda").manual_seed(0)
tot = 0.0
print(f"SMIRK_WGRAD_F16={os.environ.get('SMIRK_WGRAD_F16', '(default)')} SMIRK_WGRAD_HALO={os.environ.get('SMIRK_WGRAD_HALO', '(default)')} B={B}")
for H, co, ci, k, refl in layers:
    if ONLY and (H, co, ci, k) != ONLY:
        continue
    dz = torch.randn(B, H, H, co, device="cuda", generator=g).view(torch.float32)       # any bit pattern is a valid split16 tensor for timing
    x = torch.randn(B, H, H, ci, device="cuda", generator=g)
    dw = torch.empty(co, k * k * ci, device="cuda")
    nws = lib.smirk_conv_wgrad_workspace_bytes(B, H, H, co, ci, k)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    run = lambda: L.check(lib.smirk_conv_wgrad_f32(L.ptr(dz), L.ptr(x), L.ptr(dw), B, H, H, co, ci, k, refl, L.ptr(ws, torch.uint8), nws, st))
    for _ in range(2): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * B * H * H * co * k * k * ci
    tot += ms
    print(f"H={H:3d} Cout={co:4d} Cin={ci:4d} k={k} reflect={refl}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
print(f"total {tot:.2f} ms")
