"""Phase timeline of the halo-patch conv kernel (wave 0 of every 64th workgroup): where a patch's time goes.  GPU box only.
    python tools/patch_timeline.py [Cin] [Cout] [H] [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smirk_amd import _lib as L
import tools.conv_sweep as CS

Cin, Cout, H, B = (int(v) for v in (sys.argv[1:5] + ["32", "32", "224", "128"][len(sys.argv) - 1:]))
dbg = torch.zeros(8 * 48 * 6, dtype=torch.int64, device="cuda")
CS.run(B, H, Cin, 0, Cout, 3, 0, 0, 2, True)                       # warm-up without stamps
os.environ["SMIRK_PATCH_DBG"] = hex(dbg.data_ptr())
ms, tf = CS.run(B, H, Cin, 0, Cout, 3, 0, 0, 1, True)
torch.cuda.synchronize()
del os.environ["SMIRK_PATCH_DBG"]
d = dbg.cpu().reshape(8, 48, 6).double()
print(f"layer {Cin}->{Cout} {H}x{H} B={B}: {ms:.3f} ms (last launch of 2 stamped; s_memtime ticks = 100 MHz * ? -> shown in ticks)")
for blk in range(8):
    r = d[blk]
    ok = (r[:, 0] > 0) & (r[:, 3] > 0)
    if ok.sum() < 3:
        continue
    r = r[ok][1:-1]
    w = (r[:, 1] - r[:, 0]).mean(); c = (r[:, 2] - r[:, 1]).mean(); e = (r[:, 3] - r[:, 2]).mean()
    tail = (r[:, 4] - r[:, 3]).mean() if (r[:, 4] > 0).all() else float("nan")
    iss = (r[:, 5] - r[:, 4]).mean() if (r[:, 5] > 0).all() else float("nan")
    period = (r[1:, 0] - r[:-1, 0]).mean()
    print(f"block {blk * 64:4d}: items {len(r):3d}  wait+barrier {w:8.0f}  mfma loop {c:8.0f}  epilogue {e:8.0f}  end barrier {tail:8.0f}  issue next DMA {iss:8.0f}  | period {period:8.0f} ticks")
