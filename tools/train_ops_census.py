"""Which ATen ops does one config-5 training step launch, and from which Python line?  torch.profiler with stacks over ONE step (GPU box).

    python tools/train_ops_census.py > gpurun_out/train_ops_census.txt
"""
import collections
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                        # noqa: E402
from torch.profiler import ProfilerActivity, profile              # noqa: E402

args = bench.parse_args(["--workload", "train64"])
os.environ["SMIRK_BENCH_FLAME_BASIS"] = args.flame_basis
args.micro_batch = bench.MICRO_BATCH
dev = torch.device("cuda:0")
sandbox = tempfile.mkdtemp()
wl = bench.TrainWorkload(args, dev, 0, 1, sandbox)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    wl.step()
    torch.cuda.synchronize()
# events that launched device work, keyed by (op, innermost repo frame)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cnt, tim = collections.Counter(), collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
        continue
    if not e.kernels:                                               # no device launch directly under this op
        continue
    site = next((s for s in e.stack if REPO in s and "tools/" not in s), e.stack[0] if e.stack else "?")
    site = site.replace(REPO + "/", "")
    k = (e.name, site[:110])
    cnt[k] += len(e.kernels)
    tim[k] += sum(kk.duration for kk in e.kernels)
print(f"# device launches under ATen ops in one train64 step: {sum(cnt.values())} launches, {sum(tim.values()) / 1e3:.2f} ms of device time")
for k, c in cnt.most_common(60):
    print(f"{c:5d} {tim[k] / 1e3:8.3f} ms  {k[0]:32s} {k[1]}")
