"""Numerical A/B of one conv layer: f16x3 (patch or igemm kernel) vs the exact-fp32 igemm kernel vs torch fp64 (run through gpurun)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import _lib as L  # noqa: E402
from smirk_amd.smirk_generator import _split16, split16_to_float  # noqa: E402


def to_split(x):
    o = torch.empty_like(x)
    L.check(L.lib().smirk_f32_to_split16(L.ptr(x), L.ptr(o), x.numel(), L.stream_ptr()))
    return o


def run(B, H, C0, C1, Cout):
    lib, dev = L.lib(), torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(0)
    x0 = torch.randn(B, H, H, C0, generator=g).to(dev)
    x1 = torch.randn(B, H, H, C1, generator=g).to(dev) if C1 else None
    K = 9 * (C0 + C1)
    w = (torch.randn(Cout, K, generator=g) * 0.05).to(dev)
    sc, sh = (torch.rand(Cout, generator=g) + .5).to(dev), torch.randn(Cout, generator=g).to(dev)
    d = L.SmirkConvDesc()
    d.B, d.H, d.W, d.C0, d.C1, d.Cout, d.KH, d.KW, d.stride = B, H, H, C0, C1, Cout, 3, 3, 1
    d.pad_t = d.pad_l = 1
    d.Ho, d.Wo, d.pad_mode, d.act, d.out_mode = H, H, L.PAD_ZERO, L.ACT_RELU, L.OUT_NHWC
    P = L.ptr
    o32 = torch.empty(B, H, H, Cout, device=dev)
    L.check(lib.smirk_conv_igemm_f32(d, P(x0), P(x1, allow_none=True), P(w), P(sc), P(sh), None, P(o32), L.stream_ptr()))
    os_ = torch.empty(B, H, H, Cout, device=dev)
    s0, s1, ws = to_split(x0), (to_split(x1) if C1 else None), _split16(w)      # keep references alive across the async launch
    L.check(lib.smirk_conv_igemm_f16x3(d, P(s0), P(s1, allow_none=True), P(ws), P(sc), P(sh), None, P(os_), L.stream_ptr()))
    torch.cuda.synchronize()
    a, b = o32, split16_to_float(os_)
    err = (a - b).abs()
    print(f"B={B} H={H} C0={C0} C1={C1} Cout={Cout}: max|f32-f16x3| = {err.max().item():.3e} (ref max {a.abs().max().item():.2f}); "
          f"bad pixels {(err.amax(-1) > 1e-3).sum().item()} of {B * H * H}")
    if err.max() > 1e-3:
        bad = (err.amax(-1) > 1e-3).nonzero()
        print("  first bad (b,y,x):", bad[:5].tolist(), " last:", bad[-3:].tolist())
        ch = (err > 1e-3).any(0).any(0).any(0).nonzero().flatten().tolist()
        print("  bad channels:", ch[:40])


def debug():
    print("---- repeat case 1 three times, then B=3, then with patch disabled order swapped")
    for cfg in [(2, 32, 32, 0, 32), (2, 32, 32, 0, 32), (3, 32, 32, 0, 32), (2, 32, 32, 0, 32), (3, 64, 8, 0, 32), (3, 64, 8, 0, 32)]:
        run(*cfg)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        debug(); sys.exit(0)
    for cfg in [(2, 32, 32, 0, 32), (2, 64, 32, 0, 32), (2, 64, 32, 32, 32), (3, 64, 8, 0, 32), (2, 64, 32, 0, 64), (2, 224, 32, 32, 32), (2, 224, 32, 0, 32), (3, 224, 8, 0, 32), (5, 112, 32, 0, 64), (2, 224, 32, 32, 32)]:
        run(*cfg)
