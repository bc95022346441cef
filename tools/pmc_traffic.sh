#!/bin/bash
# HBM-side traffic per kernel launch from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over one bench step.
#   usage (GPU box, via gpurun): bash tools/pmc_traffic.sh <tag>      -> gpurun_out/<tag>_pmc_hbm.txt, gpurun_out/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r01}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p -- python /root/repo/bench.py --steps 1 --warmup 0 --cpu-faces 0 --no-overlap > /tmp/pmc_$c.log 2>&1
done
python - <<PY
import sqlite3, json, glob
def load(counter):
    db = glob.glob(f"/tmp/pmc_{counter}/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    agg = {}
    for k, v in c.execute(f"select {name_col}, value from counters_collection where counter_name = ?", (counter,)):
        k = k.split("(")[0].replace("void ", "")
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(v)
    return agg
f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
rows = []
for k in f:
    n = f[k][0]; fk = f[k][1] / n; wk = w.get(k, [1, 0.0])[1] / max(w.get(k, [1, 0.0])[0], 1)
    rows.append((k, n, fk, wk, (2 * fk + wk) * 1024))
rows.sort(key=lambda r: -r[4] * r[1])
with open("/root/repo/gpurun_out/${TAG}_pmc_hbm.txt", "w") as o:
    o.write("# rocprofv3 PMC passes (separate runs): FETCH_SIZE and WRITE_SIZE, per-dispatch mean, KB units\n")
    o.write("# command: rocprofv3 --pmc <X> --kernel-trace -- python bench.py --steps 1 --warmup 0 --cpu-faces 0 --no-overlap  (2 steps incl. instrumented)\n")
    o.write("# HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE under-reports wide reads by 2x, MI355X_MICROARCH.md; Infinity-Cache hits are included)\n")
    o.write(f"{'kernel':60s} {'launches':>8s} {'FETCH_KB':>12s} {'WRITE_KB':>12s} {'bytes/launch':>14s}\n")
    for k, n, fk, wk, b in rows:
        o.write(f"{k[:60]:60s} {n:8d} {fk:12.1f} {wk:12.1f} {b:14.0f}\n")
json.dump({k.replace(", ", ","): b for k, n, fk, wk, b in rows}, open("/root/repo/gpurun_out/pmc_traffic.json", "w"), indent=1)
print(open("/root/repo/gpurun_out/${TAG}_pmc_hbm.txt").read()[:3000])
PY
