"""Does `buffer_load_dwordx4 ... offen lds` write zeros into LDS for out-of-range lanes?  (it does: the lean K walks rely on it for the zero padding)"""
import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "micro", "liboob_lds.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "micro", "oob_lds.hip"), "-o", so])
lib = ctypes.CDLL(so)
src = torch.arange(1, 257, dtype=torch.float32, device="cuda")
out = torch.zeros(256, device="cuda")
lib.oob_lds_run(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()), 1024, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
o = out.cpu().view(64, 4)
print("lane 0 (in range):", o[0].tolist(), " lane 1 (out of range):", o[1].tolist(), " lane 2:", o[2].tolist(), " lane 3 (oob):", o[3].tolist())
