"""The REAL OverlappedPipeline flow (3 submits + flush, fresh pipeline object per trial) with a snapshot (clone on the producing stream) of every
front-stage output: tells, per trial, whether a stage PRODUCED wrong data (snapshot != serial reference) or its output was OVERWRITTEN later
(final tensor != snapshot), together with the stream handles in play.   GPU box, via gpurun."""
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
    from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator
    import synthdata as synth
    from smirk_amd import smirk_encoder as SE
    from smirk_amd.pipeline import OverlappedPipeline, SmirkPipeline
    B = 128
    reuse = len(sys.argv) > 1 and sys.argv[1] == "reuse"
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
    snaps = []

    enc_fwd, fl_fwd, rn_fwd = enc.forward, fl.forward, rn.forward

    def w_enc(img):
        o = enc_fwd(img)
        snaps.append(("enc", {k: v.clone() for k, v in o.items()}))
        return o

    def w_fl(p, *a, **k):
        o = fl_fwd(p, *a, **k)
        snaps.append(("flame", {kk: v.clone() for kk, v in o.items()}))
        return o

    def w_rn(*a, **k):
        o = rn_fwd(*a, **k)
        snaps.append(("render", {kk: v.clone() for kk, v in o.items()}))
        return o

    enc.forward, fl.forward, rn.forward = w_enc, w_fl, w_rn
    run = None
    nbad = 0
    for trial in range(14):
        if run is None or not reuse:
            run = OverlappedPipeline(pipe)
        snaps.clear()
        got = [run.submit(i, k) for i, k in batches + batches[:1]][1:] + [run.flush()]
        torch.cuda.synchronize()
        ids = dict(front=hex(run.front_stream.cuda_stream), gen=hex(run.gen_stream.cuda_stream),
                   side=[hex(s.cuda_stream) for s in SE._STREAMS[batches[0][0].device]])
        msgs = []
        for bi in range(3):                        # submits: b0, b1, b0
            ref = first[bi % 2]
            final = got[bi]
            for stage, snap in snaps[3 * bi:3 * bi + 3]:
                for k, v in snap.items():
                    if k not in ref or ref[k].shape != v.shape:
                        continue
                    pw = not torch.equal(v, ref[k])
                    ol = not torch.equal(final[k], v)
                    if pw or ol:
                        d = (v.float() - ref[k].float()).abs()
                        msgs.append(f"submit{bi}:{stage}.{k} produced_wrong={pw}(n={int((d > 0).sum())}) overwritten_later={ol}")
                        if k in ("vertices", "rendered_img") and pw and len(msgs) < 3:
                            idx = (v != ref[k]).flatten().nonzero().flatten()
                            runs, start, prev = [], int(idx[0]), int(idx[0])
                            for x in idx[1:].tolist():
                                if x != prev + 1:
                                    runs.append((start, prev - start + 1)); start = x
                                prev = x
                            runs.append((start, prev - start + 1))
                            vf, rf = v.flatten(), ref[k].flatten()
                            addr = v.data_ptr()
                            if k == "vertices":
                                V = v.shape[1]
                                for fi in idx[:48:16].tolist():
                                    b_, rem = divmod(fi, V * 3)
                                    vv, cc = divmod(rem, 3)
                                    gotv = vf[fi]
                                    same_b1 = (ref[k][:, vv, cc] == gotv).nonzero().flatten().tolist()
                                    same_b0 = (first[0][k][:, vv, cc] == gotv).nonzero().flatten().tolist()
                                    anyc = [(int(c2), (ref[k][:, vv, c2] == gotv).nonzero().flatten().tolist()) for c2 in range(3)]
                                    print(f"    wrong (b={b_}, v={vv}, c={cc}) got={float(gotv)!r}: equals this batch's value of faces {same_b1}, "
                                          f"batch0's value of faces {same_b0}, any-coordinate matches {anyc}", flush=True)
                            print(f"    {k}: base=0x{addr:x} runs(first 8)={[(hex(addr + 4 * a), n) for a, n in runs[:8]]} "
                                  f"got={vf[idx[:6]].tolist()} want={rf[idx[:6]].tolist()}", flush=True)
        nbad += bool(msgs)
        print(f"trial {trial} {ids}: {'OK' if not msgs else ' | '.join(msgs[:6])}", flush=True)
    print(f"{nbad}/14 trials corrupted (reuse={reuse})")


if __name__ == "__main__":
    main()
