"""Which generator kernel, co-running on another hardware queue, makes FLAME / the renderer produce wrong values?   (GPU box, via gpurun)
Victims: FLAME.forward and Renderer.forward on fixed inputs, on stream F.  Aggressors, one at a time on stream G: single conv layers of the
generator (every kernel variant), the whole generator, a torch GEMM.  Streams are drawn from torch's pool in the same order as the failing
OverlappedPipeline trials (second pair after the encoder's three side streams)."""
import os
import sys
import tempfile

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import assets as A                      # noqa: E402
from oracle import generator_ref as G               # noqa: E402
from oracle import mobilenet_ref as M               # noqa: E402


def main():
    from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator, synth, _lib as L
    B = 128
    skip_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    sb = tempfile.mkdtemp()
    synth.write_sandbox(sb)
    cwd = os.getcwd(); os.chdir(sb)
    try:
        fl, rn = FLAME().cuda(), Renderer().cuda()
    finally:
        os.chdir(cwd)
    enc = SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict()); enc = enc.cuda().eval()
    gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(G.synth_state_dict()); gen = gen.cuda().eval()
    img = A.synth_images(B, seed=7002).cuda()
    gin = A.synth_generator_input(B, seed=7001).cuda()
    lib = L.lib()
    with torch.no_grad():
        e0 = enc(img)                                   # creates the three side streams first, like the pipeline does
        f0 = fl.forward(e0)
        r0 = rn.forward(f0["vertices"], e0["cam"])
        gen(gin)
    torch.cuda.synchronize()
    for _ in range(skip_pairs):
        torch.cuda.Stream(); torch.cuda.Stream()
    sF, sG = torch.cuda.Stream(), torch.cuda.Stream()

    def conv_layer(H, cin0, cin1, cout, k=3, reflect=False, convt=False):
        x0 = torch.randn(B, H, H, cin0, device="cuda")
        x1 = torch.randn(B, H, H, cin1, device="cuda") if cin1 else None
        K = k * k * (cin0 + cin1)
        n = 4 * cout if convt else cout
        w = torch.randn(n, K, device="cuda") * 0.05
        xs0, ws = torch.empty_like(x0), torch.empty_like(w)
        lib.smirk_f32_to_split16(L.ptr(x0), L.ptr(xs0), x0.numel(), None)
        lib.smirk_f32_to_split16(L.ptr(w), L.ptr(ws), w.numel(), None)
        xs1 = None
        if x1 is not None:
            xs1 = torch.empty_like(x1)
            lib.smirk_f32_to_split16(L.ptr(x1), L.ptr(xs1), x1.numel(), None)
        sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        out = torch.empty((B, 2 * H, 2 * H, cout) if convt else (B, H, H, cout), device="cuda")
        d = L.SmirkConvDesc()
        d.B, d.H, d.W, d.C0, d.C1, d.Cout, d.KH, d.KW, d.stride = B, H, H, cin0, cin1, cout, k, k, 1
        d.pad_t = d.pad_l = (k - 1) // 2
        d.Ho, d.Wo = H, H
        d.pad_mode = L.PAD_REFLECT if reflect else L.PAD_ZERO
        d.act, d.out_mode = L.ACT_RELU, (L.OUT_CONVT2X2 if convt else L.OUT_NHWC)
        torch.cuda.synchronize()

        def run(reps):
            st = L.stream_ptr()
            for _ in range(reps):
                L.check(lib.smirk_conv_igemm_f16x3(d, L.ptr(xs0), L.ptr(xs1, allow_none=True), L.ptr(ws), L.ptr(sc), L.ptr(sh), None, L.ptr(out), st))
        return run

    a_mat = torch.randn(8192, 8192, device="cuda")
    aggressors = [
        ("none", lambda: None),
        ("whole generator", lambda: gen(gin)),
        ("torch.mm 8192^3 x8", lambda: [torch.mm(a_mat, a_mat) for _ in range(8)]),
        ("patch<1,1> 224^2 8->32 (enc1conv1)", lambda f=conv_layer(224, 8, 0, 32): f(8)),
        ("patch<1,2> 224^2 32+32->32 (dec1conv1)", lambda f=conv_layer(224, 32, 32, 32): f(5)),
        ("patch<2,2> 112^2 32->64 (enc2conv1)", lambda f=conv_layer(112, 32, 0, 64): f(16)),
        ("patch_stream 112^2 64+64->64 (dec2conv1)", lambda f=conv_layer(112, 64, 64, 64): f(5)),
        ("igemm<128,64,4> 112^2 64->64 (enc2conv2)", lambda f=conv_layer(112, 64, 0, 64): f(10)),
        ("igemm<128,128,5> 56^2 128->128", lambda f=conv_layer(56, 128, 0, 128): f(14)),
        ("igemm<128,128,5> 28^2 256->256", lambda f=conv_layer(28, 256, 0, 256): f(14)),
        ("igemm<128,128,5> 14^2 512->512 reflect", lambda f=conv_layer(14, 512, 0, 512, reflect=True): f(14)),
        ("igemm<128,128,4> convT 14^2 512->256", lambda f=conv_layer(14, 512, 0, 256, k=1, convt=True): f(40)),
        ("igemm<128,128,4> convT 112^2 64->32", lambda f=conv_layer(112, 64, 0, 32, k=1, convt=True): f(8)),
    ]
    if os.environ.get("DIFF5_SHORT"):
        aggressors = [a for a in aggressors if a[0].startswith(("none", "whole", "igemm<128,128,5> 28", "igemm<128,128,4> convT 14"))]
    with torch.no_grad():
        for name, load in aggressors:
            bad_f = bad_r = 0
            for trial in range(10):
                ev = torch.cuda.Event(); ev.record()
                with torch.cuda.stream(sG):
                    sG.wait_event(ev)
                    load()
                with torch.cuda.stream(sF):
                    sF.wait_event(ev)
                    outs = []
                    for _ in range(6):                   # keep the victims running for the length of the aggressor
                        f = fl.forward(e0)
                        r = rn.forward(f0["vertices"], e0["cam"])
                        outs.append((f["vertices"], r["rendered_img"]))
                torch.cuda.synchronize()
                bad_f += any(not torch.equal(v, f0["vertices"]) for v, _ in outs)
                bad_r += any(not torch.equal(i, r0["rendered_img"]) for _, i in outs)
            print(f"aggressor {name:48s}: FLAME wrong in {bad_f}/10 trials, renderer wrong in {bad_r}/10 trials", flush=True)


if __name__ == "__main__":
    main()
