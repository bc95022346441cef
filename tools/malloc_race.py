"""Platform probe (no smirk code involved): do device-memory allocations made while kernels are in flight damage what those kernels write?
Writer: a chain of torch fill / add kernels over preallocated buffers on stream A.  Disturber: fresh allocations (new torch stream => the caching
allocator must hipMalloc) + first-touch writes on other streams, on the host thread while A's kernels are still queued / running.
Afterwards every writer buffer must hold exactly the last value written.      python tools/malloc_race.py        (GPU box)"""
import sys

import torch


def run(disturb, n_bufs=24, n_elems=8 << 20, rounds=30):
    dev = torch.device("cuda", 0)
    bufs = [torch.zeros(n_elems, device=dev) for _ in range(n_bufs)]
    src = torch.arange(n_elems, device=dev, dtype=torch.float32)
    torch.cuda.synchronize()
    sA = torch.cuda.Stream()
    keep = []
    bad_total = 0
    for r in range(rounds):
        with torch.cuda.stream(sA):
            for j, b in enumerate(bufs):
                torch.add(src, float(r * 100 + j), out=b)            # every element rewritten: b[i] = i + r*100 + j
        if disturb == "malloc":
            s = torch.cuda.Stream()                                  # a new stream's pool is empty: the allocator must hipMalloc
            with torch.cuda.stream(s):
                keep.append(torch.empty(24 << 20, device=dev))       # 96 MB, untouched
        elif disturb == "malloc+touch":
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                t = torch.empty(24 << 20, device=dev)
                t.fill_(1.0)
                keep.append(t)
        elif disturb == "free":
            if keep:
                keep.pop()
            torch.cuda.empty_cache() if r % 4 == 3 else None
            keep.append(torch.empty(24 << 20, device=dev))
        torch.cuda.synchronize()
        for j, b in enumerate(bufs):
            want = src + float(r * 100 + j)
            nbad = int((b != want).sum())
            if nbad:
                bad_total += nbad
                idx = (b != want).nonzero().flatten()
                print(f"  disturb={disturb} round {r} buf {j}: {nbad} wrong elements, first at {int(idx[0])}, value {float(b[idx[0]])} want {float(want[idx[0]])}",
                      flush=True)
    print(f"disturb={disturb}: {bad_total} wrong elements in {rounds} rounds x {n_bufs} buffers", flush=True)
    del keep
    torch.cuda.empty_cache()


if __name__ == "__main__":
    for d in ("none", "malloc", "malloc+touch", "free"):
        run(d)
