"""Sustained MFMA rate of the chip (tools/micro/mfma_peak.hip): flop/s with 1 or 2 waves per SIMD and 2/4/8 independent accumulators."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "micro", "libmfma_peak.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "micro", "mfma_peak.hip"), "-o", so])
lib = ctypes.CDLL(so)
dev = torch.device("cuda")
clk = torch.zeros(2, dtype=torch.int64, device=dev)
for blocks in (256, 512, 1024):
    for nacc in (2, 4, 8):
        out = torch.empty(blocks * 256, device=dev)
        iters = 20000
        run = lambda: lib.mfma_peak_run(ctypes.c_void_p(out.data_ptr()), blocks, iters, nacc, ctypes.c_void_p(clk.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        flop = blocks * 4 * iters * nacc * 2.0 * 32 * 32 * 16
        c = clk.cpu().tolist()
        print(f"{blocks:5d} workgroups x 4 waves, {nacc} accumulators: {ms:8.3f} ms  {flop / ms / 1e9:8.1f} TFLOP/s   shader clocks/MFMA {c[0] / (iters * nacc):6.1f}   "
              f"shader clock {c[0] / (c[1] / 100e6) / 1e9:5.2f} GHz (vs 100 MHz wall clock)")
