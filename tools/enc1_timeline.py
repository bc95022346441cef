"""Phase timeline of enc1_fused_kernel (wave 0 of both groups of workgroups 0 and 97), from a -DSMIRK_DEBUG_HOOKS variant build:
    bash tools/build_variant.sh "-DSMIRK_DEBUG_HOOKS" enc1_fused.hip && SMIRK_HIP_LIBRARY=smirk_amd/lib_fz/libsmirk_hip_variant.so python tools/enc1_timeline.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_conv_gpu as T

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
IT = 24
Lm, lib, t, d, xs, w1s, w2s = T._enc1_case(B, 224, 224, 0)
T._enc1_fused(Lm, lib, d, xs, w1s, w2s, B, 224, 224)
dbg = torch.zeros(2 * 2 * IT * 16, dtype=torch.int64, device="cuda")
os.environ["SMIRK_ENC1_DBG"] = hex(dbg.data_ptr())
T._enc1_fused(Lm, lib, d, xs, w1s, w2s, B, 224, 224)
del os.environ["SMIRK_ENC1_DBG"]
r = dbg.cpu().reshape(2, 2, IT, 16).double()
names = ["conv1", "bar", "epi1(+dma issue)", "bar", "conv2", "vmcnt wait", "bar", "epi2", "bar"]
for blk in range(2):
    for g in range(2):
        x = r[blk, g]
        ok = (x[:, 0] > 0) & (x[:, 9] > 0)
        x = x[ok][2:-1]
        if len(x) < 3:
            continue
        seg = (x[:, 1:10] - x[:, :9]).mean(0)
        sub = [(x[:, 10] - x[:, 2]).mean(), (x[:, 11] - x[:, 10]).mean(), (x[:, 12] - x[:, 11]).mean(), (x[:, 13] - x[:, 12]).mean(), (x[:, 14] - x[:, 7]).mean(), (x[:, 15] - x[:, 14]).mean(), (x[:, 8] - x[:, 15]).mean()]
        period = (x[1:, 0] - x[:-1, 0]).mean()
        print(f"workgroup {(0, 97)[blk]} group {g}: iterations {len(x)}  period {period:8.0f} cycles  | " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, seg)))
        print("      epi1: dma issue %5.0f  tile0 %5.0f  tile1 %5.0f  tile2 %5.0f   epi2: tile0 valu+lds writes %5.0f  tile0 readback+stores %5.0f  tile1 %5.0f" % tuple(sub))
