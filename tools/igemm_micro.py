"""Ingredient-by-ingredient replica of the conv_igemm hot loop (tools/micro/igemm_micro.hip): cycles per 32-k chunk.
The operand source region (walked in 32 KB steps with wrap-around) is sized to be L2-, Infinity-Cache- or HBM-resident."""
import ctypes, os, subprocess
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "micro", "libigemm_micro.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "micro", "igemm_micro.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.igemm_micro_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda")
chunks = 144 * 4
names = {0: "MFMA only", 1: "+ LDS fragment reads", 3: "+ reads + barrier", 4: "+ DMA only", 5: "+ reads + DMA",
         14: "barrier + DMA + vmcnt(0)", 7: "reads + barrier + DMA (vmcnt free-running)", 15: "reads + barrier + DMA + vmcnt(0)  [= real loop]"}
out = torch.empty(1024 * 256, device=dev)
for blocks in (256, 512):
    for region_mb, label in ((2, "2 MB region (L2)"), (64, "64 MB region (Infinity Cache)"), (2048, "2 GB region (HBM)")):
        region_dw = region_mb * 1024 * 1024 // 4
        src = torch.zeros(region_dw + 16384, device=dev)
        stride_dw = (region_dw // blocks) // 8192 * 8192 or 8192
        print(f"--- {blocks} workgroups ({blocks // 256} per CU), operand source: {label}")
        for mode in (0, 1, 3, 4, 5, 14, 7, 15):
            if region_mb != 2 and not (mode & 4):
                continue
            run = lambda: lib.igemm_micro_run(mode, src.data_ptr(), out.data_ptr(), blocks, chunks, stride_dw, region_dw, torch.cuda.current_stream().cuda_stream)
            assert run() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            per_cu = blocks // 256
            print(f"  mode {mode:2d} {names[mode]:52s} {ms:7.3f} ms   {ms * 1e-3 * 2.4e9 / chunks / per_cu:7.0f} cycles@2.4GHz per chunk per WG   "
                  f"{blocks * chunks * 4 * 24 * 32768 / ms / 1e9:7.0f} TFLOP/s(MFMA)   {blocks * chunks * 32768 / ms / 1e9:6.2f} TB/s DMA")
        del src
