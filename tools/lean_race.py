"""Per-layer determinism of the conv kernels under a concurrently running stream (debugging aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import _lib as L
from smirk_amd.smirk_generator import _split16
lib = L.lib()
dev = torch.device("cuda")
def conv(B, H, C0, C1, Cout, reflect, x0, x1, w, sc, sh, out):
    d = L.SmirkConvDesc()
    convt = reflect == 2
    k = 1 if convt else 3
    d.B, d.H, d.W, d.C0, d.C1, d.Cout, d.KH, d.KW, d.stride = B, H, H, C0, C1, Cout, k, k, 1
    d.pad_t = d.pad_l = (k - 1) // 2; d.Ho, d.Wo = H, H
    d.pad_mode = L.PAD_REFLECT if reflect == 1 else L.PAD_ZERO
    d.act, d.out_mode = (L.ACT_NONE if convt else L.ACT_RELU), (L.OUT_CONVT2X2 if convt else L.OUT_NHWC)
    P = L.ptr
    L.check(lib.smirk_conv_igemm_f16x3(d, P(x0), P(x1, allow_none=True), P(w), P(sc), P(sh), None, P(out), L.stream_ptr()))
from smirk_amd import SmirkGenerator
g2 = SmirkGenerator(6, 3, 32, 5).cuda().eval(); big = torch.rand(64, 6, 224, 224, device=dev)
noise_s = torch.cuda.Stream()
for name, (B, H, C0, C1, Cout, reflect) in {"res 14 reflect": (3, 14, 512, 0, 512, 1), "bott 14 zero": (3, 14, 256, 0, 512, 0), "enc4b 28": (3, 28, 256, 0, 256, 0),
        "dec4a 28 two-source": (3, 28, 256, 256, 256, 0), "enc3b 56": (3, 56, 128, 0, 128, 0), "dec3a 56 two-source": (3, 56, 128, 128, 128, 0),
        "enc2b 112 (128x64 tile)": (3, 112, 64, 0, 64, 0), "dec2a 112 two-source 64+64": (3, 112, 64, 64, 64, 0),
        "up2 convT 56->112": (3, 56, 128, 0, 64, 2), "up3 convT": (3, 28, 256, 0, 128, 2), "up1 convT 112->224": (3, 112, 64, 0, 32, 2)}.items():
    x0 = _split16(torch.randn(B * H * H, C0, device=dev)).reshape(B, H, H, C0)
    x1 = _split16(torch.randn(B * H * H, C1, device=dev)).reshape(B, H, H, C1) if C1 else None
    w = _split16(torch.randn((4 * Cout) if reflect == 2 else Cout, (1 if reflect == 2 else 9) * (C0 + C1), device=dev) * 0.05)
    sc, sh = torch.rand(Cout, device=dev) + .5, torch.randn(Cout, device=dev)
    oshape = (B, 2 * H, 2 * H, Cout) if reflect == 2 else (B, H, H, Cout)
    ref = torch.empty(oshape, device=dev); conv(B, H, C0, C1, Cout, reflect, x0, x1, w, sc, sh, ref); torch.cuda.synchronize()
    bad = 0
    for it in range(10):
        out = torch.empty_like(ref)
        with torch.cuda.stream(noise_s), torch.no_grad():
            g2(big)
        conv(B, H, C0, C1, Cout, reflect, x0, x1, w, sc, sh, out)
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, ref))
    print(f"{name:28s} mismatching runs under load: {bad}/10")
