"""Static instruction census of one kernel in a hipcc -S listing, per basic block: VALU / SALU / LDS / VMEM / MFMA counts and the branch that ends the block.
    python tools/isa_blocks.py file.s <substring of the mangled kernel name>"""
import re
import sys


def main(path, pat):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    print(lines[start])
    blocks, cur = [], {"label": "entry", "n": {}, "br": []}
    tot = {}
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\S+):", s)
            if m:
                blocks.append(cur); cur = {"label": m.group(1), "n": {}, "br": []}
            continue
        op = s.split()[0]
        cls = ("mfma" if "mfma" in op else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
               "valu" if op.startswith("v_") else "smem" if op.startswith("s_load") or op.startswith("s_buffer") else
               "wait" if op.startswith(("s_waitcnt", "s_barrier", "s_nop")) else "branch" if op.startswith(("s_cbranch", "s_branch")) else "salu")
        cur["n"][cls] = cur["n"].get(cls, 0) + 1
        tot[cls] = tot.get(cls, 0) + 1
        if cls == "branch":
            cur["br"].append(s.split()[-1])
    blocks.append(cur)
    for b in blocks:
        n = b["n"]
        if sum(n.values()) >= 8:
            print(f"{b['label']:14s} valu {n.get('valu', 0):5d} salu {n.get('salu', 0):4d} lds {n.get('lds', 0):4d} vmem {n.get('vmem', 0):4d} mfma {n.get('mfma', 0):3d} "
                  f"wait {n.get('wait', 0):3d}  -> {' '.join(b['br'])}")
    print("total", tot)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
