"""Which stage stops being bit-reproducible when another stage runs concurrently on a second HIP stream?  (GPU box, via gpurun)

    python tools/overlap_diff.py [B]

Serial references first; then every component X (each backbone with per-block taps, whole encoder, FLAME, renderer, generator) is run TRIALS
times on its own stream while a LOAD (generator passes, or encoder passes when X is the generator) is in flight on another stream, and the
outputs are compared bitwise with the serial reference.  Prints, per component, the number of mismatching trials, the first differing
tap (backbones) and the size of the difference.
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import assets as A                      # noqa: E402  (inputs / synthetic calibrated weights only)
from oracle import generator_ref as G               # noqa: E402
from oracle import mobilenet_ref as M               # noqa: E402

TRIALS = 6


def main():
    import tempfile
    from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator, synth
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    sb = tempfile.mkdtemp()
    synth.write_sandbox(sb)
    cwd = os.getcwd(); os.chdir(sb)
    try:
        fl, rn = FLAME().cuda(), Renderer().cuda()
    finally:
        os.chdir(cwd)
    enc = SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict()); enc = enc.cuda().eval()
    gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(G.synth_state_dict()); gen = gen.cuda().eval()
    img = A.synth_images(B, seed=7001).cuda()
    gin = A.synth_generator_input(B, seed=7001).cuda()
    img2 = A.synth_images(B, seed=7002).cuda()
    gin2 = A.synth_generator_input(B, seed=7002).cuda()
    sA, sL = torch.cuda.Stream(), torch.cuda.Stream()

    def backbone(name):
        def f():
            taps = []
            getattr(enc, name).encoder(img, _taps=taps)
            return {k: v for k, v in taps}
        return f

    def encoder():
        return dict(enc(img))

    with torch.no_grad():
        e0 = enc(img)
        f0 = fl.forward(e0)
    torch.cuda.synchronize()

    def flame():
        return dict(fl.forward(e0))

    def render():
        return dict(rn.forward(f0["vertices"], e0["cam"]))

    def generator():
        return {"y": gen(gin)}

    def load_gen():
        for _ in range(2):
            gen(gin2)

    def load_enc():
        for _ in range(4):
            enc(img2)

    comps = [("pose_backbone", backbone("pose_encoder"), load_gen), ("shape_backbone", backbone("shape_encoder"), load_gen),
             ("encoder(3 streams)", encoder, load_gen), ("flame", flame, load_gen), ("renderer", render, load_gen),
             ("generator", generator, load_enc), ("generator|gen", generator, load_gen), ("shape_backbone|enc", backbone("shape_encoder"), load_enc)]
    for name, fn, load in comps:
        with torch.no_grad():
            ref = fn()
            torch.cuda.synchronize()
            ref2 = fn()                                  # serial repeatability first
            torch.cuda.synchronize()
            ser_bad = [k for k in ref if torch.is_tensor(ref[k]) and not torch.equal(ref[k], ref2[k])]
            bad_trials, first, worst = 0, None, 0.0
            for t in range(TRIALS):
                ev = torch.cuda.Event(); ev.record()
                with torch.cuda.stream(sL):
                    sL.wait_event(ev)
                    load()
                with torch.cuda.stream(sA):
                    sA.wait_event(ev)
                    got = fn()
                torch.cuda.synchronize()
                diffs = [(k, (got[k].float() - ref[k].float()).abs().max().item(), int((got[k] != ref[k]).sum().item()))
                         for k in ref if torch.is_tensor(ref[k]) and not torch.equal(got[k], ref[k])]
                if diffs:
                    bad_trials += 1
                    if first is None:
                        first = diffs[0]
                    worst = max(worst, max(d[1] for d in diffs))
        print(f"{name:24s} serial-repeat-mismatch={ser_bad[:2]}  concurrent: {bad_trials}/{TRIALS} trials differ  first={first}  worst_abs={worst:.3e}",
              flush=True)


if __name__ == "__main__":
    main()
