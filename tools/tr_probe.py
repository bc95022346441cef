"""Lane geometry of gfx950's LDS transpose read `ds_read_b64_tr_b16` (the weight-gradient kernels of train.hip build their MFMA operands with it).
LDS holds halfs equal to their own index; lane l reads at byte address 8*l (a contiguous row-major [4][16] block per 16-lane group).  Expected (model "M1",
what wgrad_f16_kernel assumes with trmap = 0): lane c of a group receives elements (c, 16 + c, 32 + c, 48 + c) of its group's 64-half block."""
import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "micro", "libtr_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "micro", "tr_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
addr = (torch.arange(64, dtype=torch.int32) * 8).cuda()
out = torch.zeros(256, device="cuda")
lib.tr_probe_run(ctypes.c_void_p(addr.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
o = out.cpu().view(64, 4).long()
for l in (0, 1, 2, 3, 4, 5, 15, 16, 17, 32, 63):
    print(f"lane {l:2d}: {o[l].tolist()}")
exp = torch.tensor([[(l >> 4) * 64 + j * 16 + (l & 15) for j in range(4)] for l in range(64)])
alt = torch.tensor([[(l >> 4) * 64 + ((l & 15) & 3) * 16 + ((l & 15) >> 2) * 4 + j for j in range(4)] for l in range(64)])
print("matches model M1 (trmap 0: lane 4r+q supplies row r, columns 4q..4q+3; lane c receives column c):", bool((o == exp).all()))
print("matches plain b64 read of a different lane (no transpose):", bool((o == alt).all()))
