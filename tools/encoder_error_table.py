"""Where the encoder tolerances come from: per-head error of the HIP encoder at the benchmarked batch sizes against a FLOAT64 evaluation of
the same network on a strided sub-sample, beside torch-CPU fp32's own error and the HIP-vs-fp32-oracle distance the parity tests actually measure; then, for the
loosest head, the error of the pooled FEATURE vector and the head's amplification of it (which layer the 1e-3 comes from).

    python tools/encoder_error_table.py [out.txt]         (GPU box; ~1 min, the float64 oracle runs on the host cores)
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import assets as A                      # noqa: E402
from oracle import mobilenet_ref as M               # noqa: E402

KEYS = ("pose_params", "cam", "shape_params", "expression_params", "eyelid_params", "jaw_params")


def main(out_path=None):
    from smirk_amd import SmirkEncoder
    from smirk_amd.smirk_encoder import features_f32
    torch.set_num_threads(min(os.cpu_count(), 64))
    sd = M.synth_encoder_state_dict()
    enc = SmirkEncoder(); enc.load_state_dict(sd); enc = enc.cuda().eval()
    r32 = M.SmirkEncoderRef(); r32.load_state_dict(sd); r32.eval()
    r64 = M.SmirkEncoderRef(); r64.load_state_dict(sd); r64 = r64.double().eval()
    lines = []
    worst = {k: [0.0, 0.0, 0.0, 0.0] for k in KEYS}
    for B in (128, 256, 1024):
        img = A.synth_images(B, seed=5200 + B)              # the images of tests/test_scale_gpu.py::test_encoder_bench_batch_*
        with torch.no_grad():
            o = enc(img.cuda())
        torch.cuda.synchronize()
        sub = list(range(1, B, max(1, B // 64)))
        with torch.no_grad():
            a32, a64 = r32(img[sub]), r64(img[sub].double())
        lines.append(f"B = {B}: {len(sub)} sub-sampled frames; max |error| per head")
        lines.append(f"  {'head':20s} {'|value| max':>12s} {'HIP vs f64':>12s} {'CPU32 vs f64':>13s} {'HIP vs CPU32':>13s}")
        for k in KEYS:
            h = o[k][sub].cpu().double()
            e_h, e_c, e_hc = (h - a64[k]).abs().max().item(), (a32[k].double() - a64[k]).abs().max().item(), (h - a32[k].double()).abs().max().item()
            lines.append(f"  {k:20s} {a64[k].abs().max().item():12.3f} {e_h:12.3e} {e_c:13.3e} {e_hc:13.3e}")
            w = worst[k]
            w[0], w[1], w[2], w[3] = max(w[0], e_h), max(w[1], e_c), max(w[2], e_hc), max(w[3], a64[k].abs().max().item())
        if B == 1024:
            # attribution for the expression head: pooled feature error (relative to the feature scale) x the head's L1 row norms
            bb = enc.expression_encoder.encoder
            with torch.no_grad():
                fg = features_f32(bb, bb(img[sub].cuda())).permute(0, 3, 1, 2).double().cpu().mean((2, 3))
                f64 = M.SmirkEncoderRef._feat(r64.expression_encoder.encoder, img[sub].double())
                f32 = M.SmirkEncoderRef._feat(r32.expression_encoder.encoder, img[sub])
            W = r64.expression_encoder.expression_layers[0].weight
            lines.append("  expression head attribution (B = 1024 sub-sample):")
            lines.append(f"    pooled 960-feature vector: |f| max {f64.abs().max().item():.3f}, HIP vs f64 max {(fg - f64).abs().max().item():.3e}, "
                         f"CPU32 vs f64 max {(f32.double() - f64).abs().max().item():.3e}")
            lines.append(f"    head Linear(960 -> 55): max row L1 norm {W.abs().sum(1).max().item():.2f}, max row L2 norm {W.norm(dim=1).max().item():.3f} "
                         f"-> worst-case amplification of a uniform feature error = the L1 norm, of independent errors ~ the L2 norm")
            d = (fg - f64)
            lines.append(f"    feature error rms {d.pow(2).mean().sqrt().item():.3e} x L2 norm = {d.pow(2).mean().sqrt().item() * W.norm(dim=1).max().item():.3e} "
                         f"(expected head error if the 960 feature errors are independent)")
    lines.append("worst over the three batch sizes -> tolerance = 2 x max(HIP vs CPU32) rounded up to one significant digit")
    for k in KEYS:
        lines.append(f"  {k:20s} HIP vs f64 {worst[k][0]:.3e}   CPU32 vs f64 {worst[k][1]:.3e}   HIP vs CPU32 {worst[k][2]:.3e}   |value| max {worst[k][3]:.3f}")
    txt = "\n".join(lines)
    print(txt)
    if out_path:
        open(out_path, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
