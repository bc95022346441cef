"""Per-kernel average of one PMC counter from a rocprofv3 `--pmc X --kernel-trace` rocpd database.
    python tools/pmc_summary.py db counter_name  ->  prints kernel, dispatches, mean counter value"""
import sqlite3
import sys


def main(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {name_col}, counter_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    agg = {}
    for k, _, v in rows:
        a = agg.setdefault(k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], [0, 0.0])
        a[0] += 1; a[1] += float(v)
    for k, (n, s) in sorted(agg.items(), key=lambda t: -t[1][1]):
        print(f"{k[:80]:80s} {n:6d} {s / n:16.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
