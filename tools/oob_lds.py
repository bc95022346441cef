import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "liboob_lds.so"))
src = torch.arange(1, 257, dtype=torch.float32, device="cuda")
out = torch.zeros(256, device="cuda")
lib.oob_lds_run(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()), 1024, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
o = out.cpu().view(64, 4)
print("lane 0 (in range):", o[0].tolist(), " lane 1 (out of range):", o[1].tolist(), " lane 2:", o[2].tolist(), " lane 3 (oob):", o[3].tolist())
