"""Where one config-5 step spends its time, WITHOUT a profiler attached: HIP events on the main stream between the phases of bench.TrainWorkload.step (front end,
generator forward, encoder forward, backward of the encoder part / of the generator part — separated by a hook on the generator's output — clip + Adam).

    python tools/train_phase_times.py [f16x3|f16x1] [steps]
"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench

arith = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
args = bench.parse_args(["--workload", "train64", "--train-arith", arith])
args.micro_batch = 1024
dev = torch.device("cuda", 0)
wl = bench.TrainWorkload(args, dev, 0, 1, tempfile.mkdtemp())
from smirk_amd.cycle import cycle_loss, render_second_path
names = ["front (FLAME x2, render x2, masking)", "generator forward", "encoder forward + loss", "re-encoded FLAME + render", "backward: encoders", "backward: generator",
         "clip + Adam"]
acc = [0.0] * len(names)
for it in range(steps + 2):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
    mid = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev[0].record()
    rendered, masked = render_second_path(wl.flame, wl.rend, wl.enc_out, wl.feats, wl.img, wl.mask, wl.face_prob, wl.MK)
    ev[1].record()
    recon = wl.gen(torch.cat([rendered, masked], 1).detach())
    recon.register_hook(lambda g: (mid.record(), g)[1])          # fires when the encoders' backward has produced d loss / d recon
    ev[2].record()
    feats = wl.enc(recon)
    loss = cycle_loss(feats, wl.feats, True, False)
    ev[3].record()
    fo = wl.flame.forward(feats)
    wl.rend.forward(fo['vertices'], feats['cam'])
    ev[4].record()
    wl.opt_e.zero_grad(set_to_none=True); wl.opt_g.zero_grad(set_to_none=True)
    loss.backward()
    ev[6].record()
    torch.nn.utils.clip_grad_norm_(wl.gen_params, 0.1)
    wl.opt_e.step(); wl.opt_g.step()
    ev[7].record()
    torch.cuda.synchronize()
    if it >= 2:
        t = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)] + [ev[4].elapsed_time(mid), mid.elapsed_time(ev[6]), ev[6].elapsed_time(ev[7])]
        acc = [a + x for a, x in zip(acc, t)]
print(f"# train64 {arith}: per-phase milliseconds (HIP events on the main stream, mean of {steps} steps; each step synchronised at both ends, so the sum is a step WITHOUT "
      "the overlap of consecutive steps)")
for n, a in zip(names, acc):
    print(f"{n:42s} {a / steps:8.3f} ms")
print(f"{'sum':42s} {sum(acc) / steps:8.3f} ms")
