"""VGPRs / AGPRs / SGPRs / scratch / static LDS of the gfx950 kernels in an object file or shared library (no GPU needed):

    python tools/kernel_resources.py smirk_amd/lib/train.o [name filter]
"""
import os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
tmp = tempfile.mkdtemp(prefix="kres_")
shutil.copy(src, os.path.join(tmp, "in.o"))
subprocess.run([LLVM + "llvm-objdump", "--offloading", "in.o"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
for f in sorted(os.listdir(tmp)):
    if "amdgcn" not in f:
        continue
    txt = subprocess.run([LLVM + "llvm-readelf", "--notes", f], cwd=tmp, capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
        if flt in dem:
            print(f"{dem[:90]:90s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
shutil.rmtree(tmp, ignore_errors=True)
