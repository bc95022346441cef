"""L2 (TCC) counter passes over one serial step of a bench workload, per kernel: hit / miss and the memory-side (EA) read-request mix of the dominant kernel
(what part of conv_halo_kernel's 2.03 GB of FETCH_SIZE traffic per launch is L2 misses of which size).

    python tools/pmc_tcc.py <workload> <out.txt> [kernel substring]        (GPU box; ~25 s per pass)

Counter names differ between ROCm releases: the candidates below are intersected with `rocprofv3 -L`; 4 TCC slots per pass (MI355X_MICROARCH.md).  Neither the
TCC nor any other block rocprofv3 exposes on gfx950 counts Infinity-Cache (MALL) hits: the cache sits behind the fabric, on the memory side of the EA interface, so
EA read requests = L2 misses = MALL hits + HBM reads.  What CAN be separated from the counters is how much of the request stream the 4 MB L2 absorbs; the MALL / HBM
split is bounded from the other side by the tensor sizes (a 9.4 MB weight tensor re-read 98 times cannot be HBM traffic when 256 MB of MALL hold it)."""
import os
import re
import shutil
import sqlite3
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANDIDATES = ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_sum",
              "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_RDREQ_GMI_sum", "TCC_EA0_RDREQ_IO_sum", "TCC_BUBBLE_sum", "TCC_EA0_RD_UNCACHED_32B_sum", "TCC_TAG_STALL_sum",
              "TCC_NORMAL_WRITEBACK_sum", "TCC_ALL_TC_OP_WB_WRITEBACK_sum", "TCC_WRITE_sum", "TCC_EA0_RDREQ_LEVEL_sum", "TCC_STREAMING_REQ_sum", "TCC_NC_REQ_sum"]


def available():
    try:
        r = subprocess.run(["rocprofv3", "-L"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120, cwd="/tmp")
        txt = r.stdout.decode(errors="replace")
    except Exception as e:                  # noqa: BLE001
        return None, str(e)
    names = set(re.findall(r"\b((?:TCC|TCP|MALL|UMC|DF)_[A-Za-z0-9_]+)\b", txt))
    return names, txt


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def one_pass(workload, group, table, notes):
    d = tempfile.mkdtemp(prefix="smirk_tcc_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc"] + group + ["--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.join(REPO, "bench.py"), "--pmc-inner", "--workload", workload,
           "--steps", "1", "--warmup", "1", "--no-overlap"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", SMIRK_GEN_SPLIT_CHAINS="0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
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


def main(workload, out, filt="conv_halo"):
    names, raw = available()
    notes = []
    if names is None:
        notes.append("rocprofv3 -L failed: " + raw)
        use = CANDIDATES[:8]
    else:
        use = [c for c in CANDIDATES if c in names or c.replace("_sum", "") in names]
        notes.append("memory-side / MALL-looking counters rocprofv3 -L lists: " + ", ".join(sorted(n for n in names if re.search(r"MALL|UMC|DF_|DRAM|GMI", n))) or "none")
    groups = [use[i:i + 4] for i in range(0, min(len(use), 16), 4)]
    table = {}
    for g in groups:
        one_pass(workload, g, table, notes)
    with open(out, "w") as fh:
        fh.write(f"# rocprofv3 --pmc TCC passes of `bench.py --pmc-inner --workload {workload} --steps 1 --warmup 1 --no-overlap` (serial schedule, one chain), MEAN PER DISPATCH\n")
        fh.write("# groups: " + " | ".join(" ".join(g) for g in groups) + "\n")
        for n in notes:
            fh.write("# NOTE " + n.replace("\n", " ")[:1500] + "\n")
        for k in sorted(table, key=lambda k: -sum(v[1] for v in table[k].values())):
            if filt and filt not in k:
                continue
            n = max(v[0] for v in table[k].values())
            fh.write(f"\n{k}   dispatches {n}\n")
            t = {cn: s / m for cn, (m, s) in table[k].items()}
            for cn in use:
                if cn in t:
                    fh.write(f"    {cn:34s} {t[cn]:18.1f}\n")
            if "TCC_HIT_sum" in t and "TCC_MISS_sum" in t:
                fh.write(f"    -> L2 hit rate {t['TCC_HIT_sum'] / max(t['TCC_HIT_sum'] + t['TCC_MISS_sum'], 1):.4f}; misses x 128 B = {t['TCC_MISS_sum'] * 128 / 1e9:.3f} GB per launch\n")
            if "TCC_EA0_RDREQ_sum" in t:
                r32 = t.get("TCC_EA0_RDREQ_32B_sum", 0.0)
                fh.write(f"    -> EA read requests {t['TCC_EA0_RDREQ_sum']:.0f} of which 32-byte {r32:.0f}: bytes = 32 B x {r32:.0f} + 64 B x {t['TCC_EA0_RDREQ_sum'] - r32:.0f} (rocprofv3's FETCH_SIZE "
                         f"convention) = {(32 * r32 + 64 * (t['TCC_EA0_RDREQ_sum'] - r32)) / 1e9:.3f} GB; x 2 for the gfx950 128-byte requests = {(32 * r32 + 128 * (t['TCC_EA0_RDREQ_sum'] - r32)) / 1e9:.3f} GB\n")
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "conv_halo")
