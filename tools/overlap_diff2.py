"""Replays tests/test_scale_gpu.py::test_full_pipeline_B128_repeated_serial_and_overlapped and prints, per trial and per output key, how the
overlapped run differs from the serial one (max abs diff, number of differing elements, which frames).  GPU box, via gpurun."""
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
    from smirk_amd.pipeline import OverlappedPipeline, SmirkPipeline
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
    pipe = SmirkPipeline(enc, fl, rn, gen)
    batches = [(A.synth_images(B, seed=s).cuda(), A.synth_generator_input(B, seed=s)[:, 3:].contiguous().cuda()) for s in (7001, 7002)]
    first = [pipe(i, k) for i, k in batches]
    torch.cuda.synchronize()
    keys = [k for k, v in first[0].items() if torch.is_tensor(v)]

    def report(tag, got):
        torch.cuda.synchronize()
        for bi, (a, b) in enumerate(zip(first, got)):
            bad = []
            for k in keys:
                if not torch.equal(a[k], b[k]):
                    d = (a[k].float() - b[k].float()).abs()
                    frames = (d.reshape(d.shape[0], -1).max(1).values > 0).nonzero().flatten().tolist()
                    bad.append(f"{k}: max {d.max().item():.3e} n={int((d > 0).sum())} frames={frames[:8]}{'...' if len(frames) > 8 else ''}")
            print(f"{tag} batch{bi}: {'OK' if not bad else ' | '.join(bad)}", flush=True)

    for trial in range(3):
        run = OverlappedPipeline(pipe)
        got = [run.submit(i, k) for i, k in batches + batches[:1]][1:] + [run.flush()]
        report(f"overlapped[{trial}]", got[:2])
    for trial in range(2):
        run = OverlappedPipeline(pipe)
        got = []
        for i, k in batches + batches[:1]:
            got.append(run.submit(i, k)); torch.cuda.synchronize()
        got = got[1:] + [run.flush()]
        report(f"overlapped+sync-between-submits[{trial}]", got[:2])
    os.environ["SMIRK_ENCODER_SERIAL"] = "1"
    for trial in range(2):
        run = OverlappedPipeline(pipe)
        got = [run.submit(i, k) for i, k in batches + batches[:1]][1:] + [run.flush()]
        report(f"overlapped, encoder on one stream[{trial}]", got[:2])
    del os.environ["SMIRK_ENCODER_SERIAL"]
    # front stages only on a side stream, nothing concurrent
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        s.wait_stream(torch.cuda.current_stream())
        got = [pipe(i, k) for i, k in batches]
    report("serial on a side stream", got)


if __name__ == "__main__":
    main()
