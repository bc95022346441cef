"""Do the conv kernels damage the LDS / registers of a kernel that is co-resident on the same CUs from another hardware queue?  (GPU box)
Victim = tools/micro/canary.hip (pattern in LDS or VGPRs, re-verified in a loop); aggressor = one generator conv layer on another stream."""
import ctypes
import os
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "micro", "libcanary.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "micro", "canary.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "micro", "canary.hip"), "-o", so])
can = ctypes.CDLL(so)
can.canary_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
can.canary_run2.argtypes = can.canary_run.argtypes


def main():
    from smirk_amd import _lib as L
    lib = L.lib()
    B = 128
    CAP = 64

    def conv_layer(H, cin0, cin1, cout, k=3, convt=False):
        x0 = torch.randn(B, H, H, cin0, device="cuda")
        K = k * k * (cin0 + cin1)
        n = 4 * cout if convt else cout
        w = torch.randn(n, K, device="cuda") * 0.05
        xs0, ws = torch.empty_like(x0), torch.empty_like(w)
        lib.smirk_f32_to_split16(L.ptr(x0), L.ptr(xs0), x0.numel(), None)
        lib.smirk_f32_to_split16(L.ptr(w), L.ptr(ws), w.numel(), None)
        xs1 = None
        if cin1:
            xs1 = torch.empty(B, H, H, cin1, device="cuda")
            lib.smirk_f32_to_split16(L.ptr(torch.randn(B, H, H, cin1, device="cuda")), L.ptr(xs1), xs1.numel(), None)
        sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        out = torch.empty((B, 2 * H, 2 * H, cout) if convt else (B, H, H, cout), device="cuda")
        d = L.SmirkConvDesc()
        d.B, d.H, d.W, d.C0, d.C1, d.Cout, d.KH, d.KW, d.stride = B, H, H, cin0, cin1, cout, k, k, 1
        d.pad_t = d.pad_l = (k - 1) // 2
        d.Ho, d.Wo, d.pad_mode, d.act = H, H, L.PAD_ZERO, L.ACT_RELU
        d.out_mode = L.OUT_CONVT2X2 if convt else L.OUT_NHWC
        torch.cuda.synchronize()

        def run(reps):
            st = L.stream_ptr()
            for _ in range(reps):
                L.check(lib.smirk_conv_igemm_f16x3(d, L.ptr(xs0), L.ptr(xs1, allow_none=True), L.ptr(ws), L.ptr(sc), L.ptr(sh), None, L.ptr(out), st))
        return run

    a_mat = torch.randn(8192, 8192, device="cuda")
    aggressors = [("none", lambda: None), ("torch.mm", lambda: [torch.mm(a_mat, a_mat) for _ in range(6)]),
                  ("igemm<128,128,5> 28^2 256->256", lambda f=conv_layer(28, 256, 0, 256): f(12)),
                  ("igemm<128,128,4> convT 14^2 512->256", lambda f=conv_layer(14, 512, 0, 256, k=1, convt=True): f(30)),
                  ("igemm<128,64,4> 112^2 64->64", lambda f=conv_layer(112, 64, 0, 64): f(8)),
                  ("patch<1,2> 224^2 32+32->32", lambda f=conv_layer(224, 32, 32, 32): f(4)),
                  ("patch<1,1> 224^2 8->32", lambda f=conv_layer(224, 8, 0, 32): f(6))]
    torch.cuda.synchronize()
    for _ in range(4):                                   # walk torch's stream pool like the failing pipeline trials do
        torch.cuda.Stream()
    for pair in range(2):
        sV, sA = torch.cuda.Stream(), torch.cuda.Stream()
        for kind, kname in ((1, "LDS"), (3, "BARRIER"), (4, "VALU")):
            for name, load in aggressors:
                log = torch.zeros(8 + CAP * 8, dtype=torch.int32, device="cuda")
                ev = torch.cuda.Event(); ev.record()
                with torch.cuda.stream(sA):
                    sA.wait_event(ev)
                    load()
                with torch.cuda.stream(sV):
                    sV.wait_event(ev)
                    st_ = torch.cuda.current_stream().cuda_stream
                    for _rep in range(4):                 # ~4 ms of victim work in total: spans the aggressor
                        if kind == 1:
                            assert can.canary_run(1, log.data_ptr(), CAP, 2048, 1500, st_) == 0
                        else:
                            assert can.canary_run2(kind, log.data_ptr(), CAP, 4096, 20000 if kind == 3 else 6000, st_) == 0
                torch.cuda.synchronize()
                h = log.cpu()
                n = int(h[0])
                msg = ""
                if n:
                    recs = h[8:8 + min(n, CAP) * 8].view(-1, 8)[:6]
                    msg = " e.g. " + "; ".join(f"blk {int(r[1])} idx {int(r[2])} got 0x{int(r[3]) & 0xffffffff:08x} want 0x{int(r[4]) & 0xffffffff:08x} round {int(r[5])}" for r in recs)
                print(f"streams pair {pair}: victim {kname:4s} canary, aggressor {name:40s}: {n} damaged words{msg}", flush=True)


if __name__ == "__main__":
    main()
