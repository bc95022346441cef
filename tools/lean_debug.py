"""Which generator layer goes wrong when a second generator runs concurrently on another stream? (debugging aid)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import SmirkGenerator
import synthdata as synth
from oracle import generator_ref as G
gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(G.synth_state_dict()); gen = gen.cuda().eval()
g2 = SmirkGenerator(6, 3, 32, 5).cuda().eval()
x = synth.synth_generator_input(3, seed=1).cuda()
big = torch.rand(64, 6, 224, 224, device="cuda")
n = torch.cuda.Stream()
def run(load):
    taps = {}
    with torch.no_grad():
        if load:
            with torch.cuda.stream(n):
                for _ in range(2): g2(big)
        out = gen.forward_pair(x[:, :3].contiguous(), x[:, 3:].contiguous(), _taps=taps)
        torch.cuda.synchronize()
    taps["out"] = out
    return {k: v.clone() for k, v in taps.items()}
q = run(False); q2 = run(False)
print("quiet twice:", {k: bool(torch.equal(q[k], q2[k])) for k in q})
bad = {}
N = 12
for i in range(N):
    l = run(True)
    first = next((k for k in q if not torch.equal(q[k], l[k])), None)
    if first: bad[first] = bad.get(first, 0) + 1
print(f"under load, {N} trials: first diverging layer counts: {bad or 'none'}")
