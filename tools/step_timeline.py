"""One step of a kernel trace (tools/trace_extract.py csv) as a text timeline: which stream shares which hardware queue, and for every stream when its
first kernel of the step started relative to the kernels of the OTHER streams on the same queue.

    python tools/step_timeline.py gpurun_out/r04o_shard128.csv.gz [marker-kernel=raster_tile] > profiles/r04o_timeline_shard128.txt
"""
import collections
import csv
import gzip
import sys


def main(path, marker="raster_tile"):
    rows = list(csv.DictReader(gzip.open(path, "rt")))
    for r in rows:
        r["start"], r["end"] = int(r["start"]), int(r["end"])
    marks = [r for r in rows if r["name"].startswith(marker)]
    a, b = marks[-2]["start"], marks[-1]["start"]
    win = [r for r in rows if a <= r["start"] < b]
    print(f"# {path}: last full step = [{marker} .. next {marker}) = {(b - a) / 1e6:.3f} ms, {len(win)} dispatches")
    q_of = collections.defaultdict(collections.Counter)
    for r in rows:
        q_of[r["stream"]][r["queue"]] += 1
    by_q = collections.defaultdict(list)
    for s, c in q_of.items():
        by_q[c.most_common(1)[0][0]].append(s)
    print("# hardware queue -> HIP streams on it (whole trace):")
    for q in sorted(by_q, key=int):
        desc = []
        for s in sorted(by_q[q], key=int):
            top = collections.Counter(r["name"].split("<")[0] for r in rows if r["stream"] == s).most_common(1)[0]
            desc.append(f"stream {s} ({sum(q_of[s].values())} launches, mostly {top[0][:28]})")
        print(f"#   queue {q}: " + "; ".join(desc))
    ev = sorted((r["start"], r["end"]) for r in win)
    busy, cs, ce = 0, *ev[0]
    for s, e in ev[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print(f"# GPU busy (union of all kernels) {busy / 1e6:.3f} ms of the step")
    print("# per stream in this step: first start, last end (ms from the step's start), launches, sum of kernel time")
    streams = sorted({r["stream"] for r in win}, key=int)
    for s in streams:
        rs = [r for r in win if r["stream"] == s]
        q = q_of[s].most_common(1)[0][0]
        first = min(rs, key=lambda r: r["start"])
        # what ended on the same queue (other stream) right before this stream's first kernel started?
        prev = [r for r in win if r["stream"] != s and q_of[r["stream"]].most_common(1)[0][0] == q and r["end"] <= first["start"] + 20000]
        gate = max(prev, key=lambda r: r["end"]) if prev else None
        g = f"   first kernel starts {(first['start'] - gate['end']) / 1e3:+.1f} us after {gate['name'][:30]} (stream {gate['stream']}, same queue) ended" if gate else ""
        print(f"  stream {s:>3s} queue {q}: {(first['start'] - a) / 1e6:7.3f} .. {(max(r['end'] for r in rs) - a) / 1e6:7.3f} ms  {len(rs):3d} launches  {sum(r['end'] - r['start'] for r in rs) / 1e6:7.3f} ms{g}")
    print("# timeline (start us, duration us, stream, kernel) of launches >= 100 us")
    for r in win:
        if r["end"] - r["start"] >= 100000:
            print(f"{(r['start'] - a) / 1e3:9.1f} {(r['end'] - r['start']) / 1e3:8.1f}  s{r['stream']:>3s}  {r['name'][:64]}")


if __name__ == "__main__":
    main(*sys.argv[1:])
