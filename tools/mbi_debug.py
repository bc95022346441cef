"""diagnostics for csrc/mbconv_image.hip against float64 (GPU box only)"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import _lib as L
from smirk_amd.smirk_generator import _split16, split16_to_float


def run(B, H, W, cin, mid, cout, res, mode):
    lib = L.lib()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, cin, generator=g)
    we = torch.randn(mid, cin, generator=g) * (1.5 / cin ** 0.5)
    wd = torch.randn(mid, 3, 3, generator=g) * 0.4
    wp = torch.randn(cout, mid, generator=g) * (1.5 / mid ** 0.5)
    aff = [(torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g) * 0.2) for n in (mid, mid, cout)]
    if mode == "wp0zero":
        wp[0] = 0
    if mode == "wezero":
        we[:] = 0
    if mode == "identity":      # E = relu(x[:, :mid]) when mid <= cin, dw = centre tap 1, project = identity
        we = torch.zeros(mid, cin); we[torch.arange(mid), torch.arange(mid) % cin] = 1
        wd = torch.zeros(mid, 3, 3); wd[:, 1, 1] = 1
        wp = torch.zeros(cout, mid); wp[torch.arange(cout), torch.arange(cout) % mid] = 1
        aff = [(torch.ones(n), torch.zeros(n)) for n in (mid, mid, cout)]
    xs = _split16(x.reshape(-1, cin).cuda()).reshape(B, H, W, cin)
    wes, wps = _split16(we.cuda().contiguous()), _split16(wp.cuda().contiguous())
    x64 = split16_to_float(xs).double().cpu().permute(0, 3, 1, 2)
    we64 = split16_to_float(wes.reshape(1, 1, mid, cin)).reshape(mid, cin).double().cpu()
    wp64 = split16_to_float(wps.reshape(1, 1, cout, mid)).reshape(cout, mid).double().cpu()
    bc = lambda t: t.double()[None, :, None, None]
    e = F.relu(F.conv2d(x64, we64[:, :, None, None]) * bc(aff[0][0]) + bc(aff[0][1]))
    d = F.relu(F.conv2d(e, wd.double()[:, None], padding=1, groups=mid) * bc(aff[1][0]) + bc(aff[1][1]))
    ref = F.conv2d(d, wp64[:, :, None, None]) * bc(aff[2][0]) + bc(aff[2][1])
    if res:
        ref = ref + x64
    out = torch.full((B, H, W, cout), 7777.0, device="cuda")
    P = L.ptr
    dev = lambda t: t.float().contiguous().cuda()
    t = [xs, wes, dev(aff[0][0]), dev(aff[0][1]), dev(wd.reshape(mid, 9).t()), dev(aff[1][0]), dev(aff[1][1]), wps, dev(aff[2][0]), dev(aff[2][1])]
    rc = lib.smirk_mbconv_image_split16(*[P(v) for v in t], int(res), P(out), B, H, W, cin, mid, cout, L.stream_ptr())
    torch.cuda.synchronize()
    raw = out.view(torch.int32)
    untouched = (out == 7777.0).float().mean().item()
    got = split16_to_float(out).permute(0, 3, 1, 2).cpu().double()
    err = (got - ref).abs()
    nan = torch.isnan(got)
    print(f"[{mode}] B={B} {H}x{W} {cin}->{mid}->{cout} res={res} rc={rc}: untouched dwords {untouched:.3f}, nan {nan.float().mean().item():.3f}, "
          f"max err (non-nan) {err[~nan].max().item() if (~nan).any() else float('nan'):.3e}, ref max {ref.abs().max().item():.2f}")
    e2 = torch.where(nan, torch.full_like(err, 1e9), err)
    per_c = e2.amax(dim=(0, 2, 3)); per_px = e2.amax(dim=(0, 1)).reshape(-1)
    print("   bad channels:", [i for i, v in enumerate(per_c.tolist()) if v > 1e-4][:40])
    print("   bad pixels  :", [i for i, v in enumerate(per_px.tolist()) if v > 1e-4][:40], "of", H * W)
    if B > 1:
        print("   per image   :", [f"{v:.1e}" for v in e2.amax(dim=(1, 2, 3)).tolist()])


run(1, 14, 14, 64, 32, 48, False, "random")        # one chunk
run(1, 14, 14, 64, 64, 48, False, "random")        # two chunks
run(1, 14, 14, 64, 96, 48, False, "random")        # three chunks
run(1, 14, 14, 64, 64, 48, False, "wp0zero")       # project row 0 = 0: channel 0 must be b3[0]
run(1, 14, 14, 64, 64, 48, False, "wezero")        # expand = 0: E = relu(b1)
