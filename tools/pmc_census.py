"""Counter census of a whole bench workload: a few rocprofv3 --pmc passes (kernel-trace only, one counter group per pass) over one serial step, aggregated per kernel.
    python tools/pmc_census.py <workload> <out.txt>        (on the GPU box, via gpurun; ~20 s per pass)
Every column is the MEAN PER DISPATCH of that kernel.  SQ_* cycle counters are in quad-cycles summed over waves (MI355X_MICROARCH.md, PMC slots)."""
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

GROUPS = [
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_SMEM"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INST_LEVEL_LDS", "TCC_HIT_sum", "TCC_MISS_sum"],
]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def one_pass(workload, group, table, notes):
    d = tempfile.mkdtemp(prefix="smirk_census_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc"] + group + ["--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.join(REPO, "bench.py"), "--pmc-inner", "--workload", workload,
           "--steps", "1", "--warmup", "1", "--no-overlap"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    dbs = [os.path.join(p, f) for p, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
    if r.returncode != 0 or not dbs:
        notes.append(f"pass {group} failed rc={r.returncode}: {r.stdout.decode(errors='replace')[-300:]}")
        shutil.rmtree(d, ignore_errors=True)
        return
    c = sqlite3.connect(dbs[0])
    cols = [x[1] for x in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    for k, cn, v in c.execute(f"select {name_col}, counter_name, value from counters_collection"):
        a = table.setdefault(short(k), {}).setdefault(cn, [0, 0.0])
        a[0] += 1; a[1] += float(v)
    shutil.rmtree(d, ignore_errors=True)


def main(workload, out):
    table, notes = {}, []
    for g in GROUPS:
        one_pass(workload, g, table, notes)
    names = [c for g in GROUPS for c in g]
    with open(out, "w") as fh:
        fh.write(f"# rocprofv3 --pmc census of `bench.py --pmc-inner --workload {workload} --steps 1 --warmup 1 --no-overlap` (serial schedule), mean per dispatch\n")
        for n in notes:
            fh.write("# NOTE " + n.replace("\n", " ") + "\n")
        order = sorted(table, key=lambda k: -(table[k].get("SQ_BUSY_CYCLES", [1, 0.0])[1]))
        for k in order:
            if k.startswith("at::") or k.startswith("__amd"):
                continue
            n = max(v[0] for v in table[k].values())
            fh.write(f"\n{k}   dispatches {n}\n")
            for cn in names:
                if cn in table[k]:
                    m, s = table[k][cn]
                    fh.write(f"    {cn:28s} {s / m:18.1f}\n")
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
