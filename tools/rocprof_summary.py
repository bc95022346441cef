"""Turn a rocprofv3 `--kernel-trace --stats` result database (rocpd sqlite) into a small text summary for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/r01_results.db profiles/r01_kernel_stats.txt "command line that was profiled" [avg.json]

With a fourth argument the per-kernel call counts and average durations are also written as JSON together with the sha of smirk_amd/csrc/* they were
measured on (bench.py quotes rocprofv3's per-launch figure of its dominant kernel from profiles/rocprofv3_kernel_avg_<workload>.json when the sha matches).
"""
import json
import os
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    if "at::native" in name:
        return "torch::" + name.split("at::native::")[1].split("<")[0].split("(")[0]
    return name.split("(")[0]


def main(db, out, cmd, avg_json=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n# command: {cmd}\n# source db: {db}\n")
        f.write(f"# total kernel time {tot / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches (durations in microseconds)\n")
        f.write(f"{'kernel':70s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}\n")
        for n, calls, total, avg, pct in rows:
            f.write(f"{short(n)[:70]:70s} {calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}\n")
    print(open(out).read())
    if avg_json:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        json.dump({"kernel_sources_sha": bench.kernel_sources_sha(), "command": cmd, "summary_file": os.path.basename(out),
                   "kernels": {short(n).replace(", ", ","): {"calls": calls, "avg_us": round(avg, 2)} for n, calls, total, avg, pct in rows}},
                  open(avg_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "", sys.argv[4] if len(sys.argv) > 4 else None)
