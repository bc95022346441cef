"""Phase timeline of conv_halo_kernel (waves 0 and 4 of every 97th workgroup): where a chunk's time goes.  GPU box only; needs the
-DSMIRK_DEBUG_HOOKS variant library:
    bash tools/build_variant.sh "-DSMIRK_DEBUG_HOOKS -fno-slp-vectorize -fno-vectorize" conv_halo.hip
    SMIRK_HIP_LIBRARY=smirk_amd/lib_fz/libsmirk_hip_variant.so SMIRK_IGEMM_HALO=all python tools/halo_timeline.py [H] [Cin] [Cout] [B] [reflect]
Per chunk and wave four stamps (s_memtime): load-phase start, load-phase end (DMA issued, vmcnt/lgkmcnt waited), matrix-phase start (after the
barrier), matrix-phase end (24 MFMAs issued).  Printed: mean cycles of the load phase, the wait at barrier 1, the matrix phase, the wait at barrier 2."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.conv_sweep as CS

H, Cin, Cout, B, refl = (int(v) for v in (sys.argv[1:6] + ["14", "512", "512", "1024", "1"][len(sys.argv) - 1:]))
N = 1024
dbg = torch.zeros(8 * 2 * N, dtype=torch.int64, device="cuda")
CS.run(B, H, Cin, 0, Cout, 3, 0, refl, 2, True)                     # warm-up without stamps
os.environ["SMIRK_HALO_DBG"] = hex(dbg.data_ptr())
ms, tf = CS.run(B, H, Cin, 0, Cout, 3, 0, refl, 1, True)
torch.cuda.synchronize()
del os.environ["SMIRK_HALO_DBG"]
d = dbg.cpu().reshape(8, 2, N).double()
print(f"layer {Cin}->{Cout} {H}x{H} B={B} EB={os.environ.get('SMIRK_HALO_EB', 'default')}: {ms:.3f} ms ({tf:.0f} TFLOP/s) with stamps (last launch of 2 stamped)")
for slot in range(8):
    for w in range(2):
        r = d[slot, w]
        n = int((r > 0).sum())
        if n < 9:
            continue
        nb = (int((r[:N - 4] > 0).sum()) - 1) // 4
        t = r[:nb * 4].reshape(nb, 4)
        end = r[nb * 4]
        entry, ep0, ep1 = r[N - 4], r[N - 3], r[N - 2]
        load = (t[:, 1] - t[:, 0])[2:-2].mean(); b1 = (t[:, 2] - t[:, 1])[2:-2].mean(); mm = (t[:, 3] - t[:, 2])[2:-2].mean()
        b2 = (t[1:, 0] - t[:-1, 3])[2:-2].mean()
        period = (t[1:, 0] - t[:-1, 0])[2:-2].mean()
        print(f"wg slot {slot} wave {4 * w}: chunks {nb:4d}  load {load:7.0f}  barrier1 {b1:6.0f}  matrix {mm:7.0f}  barrier2 {b2:6.0f} | period {period:7.0f} cycles; prologue {t[0, 0] - entry:7.0f}  main loop {t[-1, 3] - t[0, 0]:9.0f}  drain {ep0 - t[-1, 3]:7.0f}  epilogue {ep1 - ep0:7.0f}  total {ep1 - entry:9.0f}")
