"""Per-layer timing of the generator's conv shapes on the MI355X (tuning aid; run through gpurun).

    python tools/conv_sweep.py [--batch 128] [--iters 5]
Prints one line per distinct (H, Cin, Cout, kind) with the achieved fp32 TFLOP/s of smirk_conv_igemm_f32.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import _lib as L  # noqa: E402


def run(B, H, C0, C1, Cout, k, convt, reflect, iters, split=False):
    lib = L.lib()
    dev = torch.device("cuda")
    x0 = torch.randn(B, H, H, C0, device=dev)
    x1 = torch.randn(B, H, H, C1, device=dev) if C1 else None
    n = 4 * Cout if convt else Cout
    K = k * k * (C0 + C1)
    w = torch.randn(n, K, device=dev) * 0.05
    sc, sh = torch.rand(Cout, device=dev) + .5, torch.randn(Cout, device=dev)
    out = torch.empty((B, 2 * H, 2 * H, Cout) if convt else (B, H, H, Cout), device=dev)
    d = L.SmirkConvDesc()
    d.B, d.H, d.W, d.C0, d.C1, d.Cout, d.KH, d.KW, d.stride = B, H, H, C0, C1, Cout, k, k, 1
    d.pad_t = d.pad_l = (k - 1) // 2
    d.Ho, d.Wo = H, H
    d.pad_mode = L.PAD_REFLECT if reflect else L.PAD_ZERO
    d.act = L.ACT_RELU
    d.out_mode = L.OUT_CONVT2X2 if convt else L.OUT_NHWC
    P = L.ptr
    fn = lib.smirk_conv_igemm_f16x3 if split else lib.smirk_conv_igemm_f32
    call = lambda: L.check(fn(d, P(x0), P(x1, allow_none=True), P(w), P(sc), P(sh), None, P(out), L.stream_ptr()))
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * H * H * n * K
    return ms, fl / ms / 1e9          # TFLOP/s (algorithmic)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--mode", default="f16x3", choices=["f32", "f16x3"])
    ap.add_argument("--only", default="", help="substring filter on the layer name")
    ap.add_argument("--ab-pp", action="store_true", help="time every layer twice: SMIRK_IGEMM_PP=0 (128x128 kernel) and default (ping-pong kernel where eligible)")
    ap.add_argument("--ab-env", default="", help="NAME=VALUE: time every layer twice, with this environment variable set and without")
    a = ap.parse_args()
    B = a.batch
    layers = [  # name, H, C0, C1, Cout, k, convt, reflect, count in the generator
        ("enc1a", 224, 8, 0, 32, 3, 0, 0, 1), ("enc1b/dec1b", 224, 32, 0, 32, 3, 0, 0, 2), ("dec1a", 224, 32, 32, 32, 3, 0, 0, 1),
        ("enc2a", 112, 32, 0, 64, 3, 0, 0, 1), ("enc2b/dec2b", 112, 64, 0, 64, 3, 0, 0, 2), ("dec2a", 112, 64, 64, 64, 3, 0, 0, 1),
        ("enc3a", 56, 64, 0, 128, 3, 0, 0, 1), ("enc3b/dec3b", 56, 128, 0, 128, 3, 0, 0, 2), ("dec3a", 56, 128, 128, 128, 3, 0, 0, 1),
        ("enc4a", 28, 128, 0, 256, 3, 0, 0, 1), ("enc4b/dec4b", 28, 256, 0, 256, 3, 0, 0, 2), ("dec4a", 28, 256, 256, 256, 3, 0, 0, 1),
        ("bott_a", 14, 256, 0, 512, 3, 0, 0, 1), ("bott_b", 14, 512, 0, 512, 3, 0, 0, 1), ("res(reflect)", 14, 512, 0, 512, 3, 0, 1, 10),
        ("up4", 14, 512, 0, 256, 1, 1, 0, 1), ("up3", 28, 256, 0, 128, 1, 1, 0, 1), ("up2", 56, 128, 0, 64, 1, 1, 0, 1),
        ("up1", 112, 64, 0, 32, 1, 1, 0, 1),
    ]
    tot_ms = tot_fl = 0.0
    for name, H, C0, C1, Cout, k, convt, refl, cnt in layers:
        if a.only and a.only not in name:
            continue
        extra = ""
        if a.ab_pp:
            os.environ["SMIRK_IGEMM_PP"] = "0"
            ms0, tf0 = run(B, H, C0, C1, Cout, k, convt, refl, a.iters, a.mode == "f16x3")
            del os.environ["SMIRK_IGEMM_PP"]
            extra = f"   [PP off: {ms0:8.3f} ms {tf0:7.1f} TFLOP/s]"
        if a.ab_env:
            k_, v_ = a.ab_env.split("=", 1)
            old_ = os.environ.get(k_)
            os.environ[k_] = v_
            ms0, tf0 = run(B, H, C0, C1, Cout, k, convt, refl, a.iters, a.mode == "f16x3")
            if old_ is None:
                del os.environ[k_]
            else:
                os.environ[k_] = old_
            extra = f"   [{a.ab_env}: {ms0:8.3f} ms {tf0:7.1f} TFLOP/s]"
        ms, tf = run(B, H, C0, C1, Cout, k, convt, refl, a.iters, a.mode == "f16x3")
        tot_ms += ms * cnt; tot_fl += tf * ms * cnt
        print(f"{name:14s} H={H:3d} Cin={C0 + C1:4d} Cout={Cout:4d} k={k} convT={convt} x{cnt:2d}: {ms:8.3f} ms  {tf:7.1f} TFLOP/s{extra}", flush=True)
    print(f"generator igemm total: {tot_ms:.2f} ms for B={B}  ->  {tot_fl / tot_ms:.1f} TFLOP/s average, {B / tot_ms * 1e3:.0f} faces/s bound")
