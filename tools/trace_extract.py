"""rocpd sqlite (rocprofv3 --kernel-trace) -> a compact gzip csv of (kernel, stream, queue, start_ns, end_ns), small enough to travel back through gpurun_out/.

    python tools/trace_extract.py /tmp/rp/x_results.db gpurun_out/x_trace.csv.gz
"""
import gzip
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in names else None
    if view is None:
        print("no 'kernels' view; objects:", names)
        return 1
    cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
    print("columns:", cols)
    pick = lambda *cands: next((x for x in cands if x in cols), None)
    name, start, end = pick("name", "kernel_name"), pick("start", "start_timestamp"), pick("end", "end_timestamp")
    stream, queue = pick("stream_id", "stream"), pick("queue_id", "queue")
    gx, wx = pick("grid_x", "grid_size_x", "grid_size"), pick("workgroup_x", "workgroup_size_x", "workgroup_size")
    sel = [name, stream or "0", queue or "0", start, end, gx or "0", wx or "0"]
    rows = list(c.execute(f"select {', '.join(sel)} from {view} order by {start}"))
    with gzip.open(out, "wt") as f:
        f.write("name,stream,queue,start,end,grid,wg\n")
        for r in rows:
            n = str(r[0]).replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].replace(",", ";")
            f.write(f"{n},{r[1]},{r[2]},{r[3]},{r[4]},{r[5]},{r[6]}\n")
    print(len(rows), "dispatches ->", out)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
