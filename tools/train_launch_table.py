"""Every library launch of ONE config-5 training step (launch profiler: HIP events around each launch, so concurrency between the three backbone streams is serialised
away) with its algorithmic bytes / flop: which launches are long AND far from the HBM rate.

    python tools/train_launch_table.py [f16x3|f16x1] [top N] > gpurun_out/r06_train_launch_table.txt
"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench
from smirk_amd import _lib as L

arith = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 70
args = bench.parse_args(["--workload", "train64", "--train-arith", arith])
args.micro_batch = 1024
wl = bench.TrainWorkload(args, torch.device("cuda", 0), 0, 1, tempfile.mkdtemp())
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
L.profile_start()
wl.instrumented()
torch.cuda.synchronize()
recs = L.profile_stop()
tot = sum(r[3] for r in recs)
print(f"# train64 {arith}: {len(recs)} library launches, {tot:.2f} ms of event-bracketed kernel time; longest {top} launches (position in the step, kernel, ms, algorithmic GB/s, TFLOP/s)")
order = sorted(range(len(recs)), key=lambda i: -recs[i][3])[:top]
for i in order:
    n, fl, by, ms = recs[i]
    print(f"{i:5d}  {n[:64]:64s} {ms * 1e3:8.1f} us  {by / ms / 1e6 if by else 0:8.1f} GB/s  {fl / ms / 1e9 if fl else 0:7.1f} TF")
agg = {}
for n, fl, by, ms in recs:
    a = agg.setdefault(n, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += by
print("# per kernel: launches, total ms, mean GB/s over its launches")
for n, (c, ms, by) in sorted(agg.items(), key=lambda t: -t[1][1])[:40]:
    print(f"{n[:64]:64s} x{c:4d} {ms:8.3f} ms  {by / ms / 1e6 if by else 0:8.1f} GB/s")
