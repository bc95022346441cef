"""Effective shader clock per kernel: one rocprofv3 pass with GRBM_GUI_ACTIVE (summed over the 8 XCDs) next to the dispatch durations of the SAME pass.
    python tools/pmc_clock.py <workload> <out.txt>     f_eff = GRBM_GUI_ACTIVE / 8 / duration   (MI355X_MICROARCH.md: power-capped MFMA kernels run well below 2.4 GHz)"""
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main(workload, out):
    d = tempfile.mkdtemp(prefix="smirk_clock_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", "GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "--kernel-trace", "--stats", "-d", d, "-o", "p", "--", sys.executable,
           os.path.join(REPO, "bench.py"), "--pmc-inner", "--workload", workload, "--steps", "2", "--warmup", "1", "--no-overlap"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=200)
    dbs = [os.path.join(p, f) for p, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
    if not dbs:
        print("no db", r.returncode, r.stdout.decode(errors="replace")[-600:]); return
    c = sqlite3.connect(dbs[0])
    cols = [x[1] for x in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    agg = {}
    for k, cn, v in c.execute(f"select {name_col}, counter_name, value from counters_collection"):
        a = agg.setdefault(short(k), {}).setdefault(cn, [0, 0.0]); a[0] += 1; a[1] += float(v)
    dur = {}
    try:
        for n, calls, total, avg in c.execute("select name, total_calls, total_duration, average from top_kernels"):
            dur[short(n)] = (calls, float(avg))
        unit = "top_kernels.average"
    except sqlite3.Error as e:
        unit = f"no top_kernels view ({e}); tables: " + ", ".join(x[0] for x in c.execute("select name from sqlite_master where type in ('table','view')"))[:1500]
    with open(out, "w") as fh:
        fh.write(f"# one rocprofv3 pass (GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES + kernel trace) of bench.py --pmc-inner --workload {workload} --no-overlap; durations: {unit}\n")
        fh.write(f"{'kernel':46s} {'calls':>5s} {'avg_dur(raw)':>13s} {'GUI_ACTIVE':>13s} {'GUI/8/dur':>10s} {'MFMA_BUSY':>14s} {'mfma busy / (GUI/8*1024)':>26s}\n")
        for k, v in sorted(agg.items(), key=lambda t: -t[1].get("GRBM_GUI_ACTIVE", [1, 0])[1]):
            if k.startswith("at::") or k.startswith("__amd"):
                continue
            g = v.get("GRBM_GUI_ACTIVE", [1, 0.0]); g = g[1] / g[0]
            m = v.get("SQ_VALU_MFMA_BUSY_CYCLES", [1, 0.0]); m = m[1] / m[0]
            calls, avg = dur.get(k, (0, 0.0))
            fh.write(f"{k[:46]:46s} {calls:5d} {avg:13.2f} {g:13.0f} {(g / 8 / avg if avg else 0):10.4f} {m:14.0f} {(m / (g / 8 * 1024) if g else 0):26.3f}\n")
    shutil.rmtree(d, ignore_errors=True)
    print(open(out).read()[:2500])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
