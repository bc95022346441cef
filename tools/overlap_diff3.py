"""Narrowing the overlapped-pipeline corruption: the front stages (encode -> FLAME -> render) of batch 1 run on one stream with a snapshot
(clone on the same stream) taken after every stage, while a LOAD runs on another stream.  For every trial prints which stage output was
produced wrong (snapshot != serial reference) and which was overwritten later (final tensor != its own snapshot).   GPU box, via gpurun."""
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
    from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator, synth
    B = 128
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
    a_mat = torch.randn(8192, 8192, device="cuda")
    sF, sL = torch.cuda.Stream(), torch.cuda.Stream()

    def front(snap):
        e = enc(img)
        if snap is not None:
            snap.update({"enc." + k: v.clone() for k, v in e.items()})
        f = fl.forward(e)
        if snap is not None:
            snap.update({"flame." + k: v.clone() for k, v in f.items()})
        r = rn.forward(f["vertices"], e["cam"], landmarks_fan=f["landmarks_fan"], landmarks_mp=f["landmarks_mp"])
        if snap is not None:
            snap.update({"render." + k: v.clone() for k, v in r.items()})
        out = {"enc." + k: v for k, v in e.items()}
        out.update({"flame." + k: v for k, v in f.items()})
        out.update({"render." + k: v for k, v in r.items()})
        return out

    with torch.no_grad():
        ref = front(None)
        torch.cuda.synchronize()

        def load_gen():
            gen(gin)

        def load_mm():
            for _ in range(12):
                torch.mm(a_mat, a_mat)

        def load_none():
            pass

        def load_gen_f32():
            gen.precision = "f32"
            try:
                gen(gin[:32])
            finally:
                gen.precision = "f16x3"

        for lname, load in (("generator", load_gen), ("torch.mm", load_mm), ("generator f32 mode (B=32)", load_gen_f32), ("none", load_none)):
            bad = 0
            for trial in range(8):
                ev = torch.cuda.Event(); ev.record()
                with torch.cuda.stream(sL):
                    sL.wait_event(ev)
                    load()
                snap = {}
                with torch.cuda.stream(sF):
                    sF.wait_event(ev)
                    out = front(snap)
                torch.cuda.synchronize()
                wrong = [k for k in ref if not torch.equal(snap[k], ref[k])]
                later = [k for k in ref if not torch.equal(out[k], snap[k])]
                if wrong or later:
                    bad += 1
                    det = []
                    for k in wrong[:4]:
                        d = (snap[k].float() - ref[k].float()).abs()
                        det.append(f"{k}: n={int((d > 0).sum())} max={d.max().item():.2e}")
                    print(f"  load={lname} trial {trial}: produced-wrong={wrong} overwritten-later={later} :: {'; '.join(det)}", flush=True)
            print(f"load={lname}: {bad}/8 trials corrupted", flush=True)


if __name__ == "__main__":
    main()
