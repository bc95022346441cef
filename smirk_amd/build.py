"""Builds smirk_amd/lib/libsmirk_hip.so from smirk_amd/csrc/*.hip with hipcc for gfx950 (cross-compiles without a GPU).

    python -m smirk_amd.build [--force]

Every translation unit is compiled on its own (render.hip with -ffp-contract=off: the rasteriser must reproduce the
reference's un-fused fp32 arithmetic bit for bit) and linked into one shared object with a C ABI (include/smirk_hip.h).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsmirk_hip.so")
ARCH = "gfx950"
# -fno-slp-vectorize: keeps hipcc from fusing pairs of fp32 operations into packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 /
# v_pk_add_f32).  Measured on the MI355X boxes of this pool (DESIGN.md §6 "packed-FP32 under co-residency"; tools/overlap_diff.py and its siblings in the git history,
# round 3): a wave executing v_pk_*_f32 returns WRONG results in some 16-lane groups while a wave of ANOTHER kernel that issues
# fp16 MFMAs + ds_read_b128 (the generator's implicit-GEMM kernels, launched from a second stream) is resident on the same CU — FLAME
# vertices and rendered pixels were corrupted in 10/10 concurrent trials with the packed instructions and in 0/10 without them; waits,
# barriers, LDS contents and out-of-bounds writes were all ruled out first.  The two-stream pipeline needs that co-residency, so the packed
# forms are not generated at all (cost: < 1 % on the conv epilogues, nothing measurable end to end).  -fno-vectorize: the LOOP vectoriser
# produces the same packed forms (round 2 shipped two v_pk_add_f32 in maxfilter1d_kernel that way); tests/test_isa_cpu.py disassembles
# the built library and fails on any v_pk_{add,mul,fma}_f32.
COMMON = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize", "-fno-vectorize"]
PER_FILE = {"render.hip": ["-ffp-contract=off"], "video.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "smirk_hip.h"))
    objs, jobs = [], []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc] + COMMON + PER_FILE.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
