"""ISA-level contracts of the built library, checked WITHOUT a GPU by disassembling smirk_amd/lib/libsmirk_hip.so.

* No packed-FP32 VALU instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) anywhere: DESIGN.md §8.1 — on the MI355X boxes of this pool
  that instruction class returns wrong 16-lane groups while an fp16-MFMA + ds_read_b128 kernel of another stream is co-resident, and the
  two-stream pipeline (bench.py's default schedule) needs that co-residency.  Round 2 shipped two of them in maxfilter1d_kernel (loop
  vectoriser); build.py now passes -fno-slp-vectorize AND -fno-vectorize.
* The instructions the design claims are really there (fp16 / fp32 MFMA, direct-to-LDS DMA, LDS transpose reads) — a silently de-optimised
  build would otherwise pass every numerical test.
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "smirk_amd", "lib", "libsmirk_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(lib=LIB):
    """-> {kernel symbol: [instruction lines]} over every gfx950 code object bundled in the shared library."""
    tmp = tempfile.mkdtemp(prefix="smirk_isa_")
    try:
        dst = os.path.join(tmp, "lib.so")
        shutil.copy(lib, dst)
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        kernels = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", f], cwd=tmp, check=True, capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1)
                    kernels.setdefault(cur, [])
                elif cur is not None and line.startswith("\t"):
                    kernels[cur].append(line.strip())
        return kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.fixture(scope="module")
def isa():
    if not os.path.exists(LIB):
        from smirk_amd import build
        build.build(verbose=False)
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not in this image")
    k = disassemble()
    assert len(k) > 100, "expected the library's kernels in the offload bundles"
    return k


def test_no_packed_fp32_valu_instructions(isa):
    bad = {}
    for name, ins in isa.items():
        n = sum(1 for i in ins if re.match(r"v_pk_(add|mul|fma)_f32\b", i))
        if n:
            bad[name] = n
    assert not bad, f"packed-FP32 VALU instructions are banned from this library (DESIGN.md §8.1): {bad}"


def test_design_instructions_present(isa):
    allins = [i for ins in isa.values() for i in ins]

    def count(pat):
        return sum(1 for i in allins if re.match(pat, i))
    assert count(r"v_mfma_f32_32x32x16_f16\b") > 1000          # split-fp16 x3 convolutions / weight gradients
    assert count(r"v_mfma_f32_32x32x2_f32\b") > 100            # exact-fp32 mode, FLAME contraction
    assert count(r"buffer_load_dwordx4 .* lds\b") > 300        # direct-to-LDS operand DMA
    assert count(r"ds_read_b64_tr_b16\b") > 50                 # LDS transpose reads of the weight-gradient kernels


def test_fused_maxpool_kernel_is_in_the_library(isa):
    names = [k for k in isa if "maxpool_sq_lds_kernel" in k]
    assert len(names) == 2, names                              # R = 10 (masking.py:78) and R = 5 (masking.py:96)


def test_no_scratch_memory_in_any_kernel(isa):
    """No kernel of the library touches scratch (private) memory: a register spill inside a chunk loop is a global-memory round trip per iteration, and the
    register-bound fused encoder kernels (mbconv_image_kernel<7,4> / <6,4> / <5,4> / <7,3>, mbconv_fused_kernel<1,true,3>) shipped with 3-34 scratch instructions
    through round 4.  Their loop-invariant address arithmetic is now re-derived per chunk behind `asm volatile` fences and the epilogues re-derive the
    lane / wave ids (mbcnt, an SGPR copy) instead of keeping the thread id alive across the loop."""
    bad = {}
    for name, ins in isa.items():
        n = sum(1 for i in ins if re.match(r"scratch_(load|store)", i))
        if n:
            bad[name] = n
    assert not bad, f"kernels with scratch traffic: {bad}"
