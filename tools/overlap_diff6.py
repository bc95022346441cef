"""Aggressor = the ingredient-by-ingredient replica of the igemm hot loop (tools/micro/igemm_micro.hip) or the real conv kernels under their
A/B switches; victims = FLAME.forward / Renderer.forward on another hardware queue.  Which ingredient breaks the victims?   (GPU box)"""
import ctypes
import os
import subprocess
import sys
import tempfile

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import assets as A                      # noqa: E402
from oracle import mobilenet_ref as M               # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "micro", "libigemm_micro.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "micro", "igemm_micro.hip"), "-o", so])
mic = ctypes.CDLL(so)
so2 = os.path.join(here, "micro", "libcanary.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "micro", "canary.hip"), "-o", so2])
can = ctypes.CDLL(so2)
can.lds_hog_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
mic.igemm_micro_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]


def main():
    from smirk_amd import FLAME, Renderer, SmirkEncoder, synth, _lib as L
    B = 128
    sb = tempfile.mkdtemp()
    synth.write_sandbox(sb)
    cwd = os.getcwd(); os.chdir(sb)
    try:
        fl, rn = FLAME().cuda(), Renderer().cuda()
    finally:
        os.chdir(cwd)
    enc = SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict()); enc = enc.cuda().eval()
    img = A.synth_images(B, seed=7002).cuda()
    lib = L.lib()
    with torch.no_grad():
        e0 = enc(img)
        f0 = fl.forward(e0)
        r0 = rn.forward(f0["vertices"], e0["cam"])
    torch.cuda.synchronize()
    for _ in range(2):
        torch.cuda.Stream()
    sF, sG = torch.cuda.Stream(), torch.cuda.Stream()
    region_dw = 2 * 1024 * 1024 // 4
    src = torch.zeros(region_dw + 16384, device="cuda")
    mout = torch.empty(1024 * 256, device="cuda")

    def micro(mode, blocks=512, chunks=144 * 12):
        def run():
            assert mic.igemm_micro_run(mode, src.data_ptr(), mout.data_ptr(), blocks, chunks, 8192, region_dw, torch.cuda.current_stream().cuda_stream) == 0
        return run

    def conv(H, cin, cout, k=3, convt=False, reps=12):
        x0 = torch.randn(B, H, H, cin, device="cuda")
        n = 4 * cout if convt else cout
        w = torch.randn(n, k * k * cin, device="cuda") * 0.05
        xs0, ws = torch.empty_like(x0), torch.empty_like(w)
        lib.smirk_f32_to_split16(L.ptr(x0), L.ptr(xs0), x0.numel(), None)
        lib.smirk_f32_to_split16(L.ptr(w), L.ptr(ws), w.numel(), None)
        sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        out = torch.empty((B, 2 * H, 2 * H, cout) if convt else (B, H, H, cout), device="cuda")
        d = L.SmirkConvDesc()
        d.B, d.H, d.W, d.C0, d.C1, d.Cout, d.KH, d.KW, d.stride = B, H, H, cin, 0, cout, k, k, 1
        d.pad_t = d.pad_l = (k - 1) // 2
        d.Ho, d.Wo, d.pad_mode, d.act = H, H, L.PAD_ZERO, L.ACT_RELU
        d.out_mode = L.OUT_CONVT2X2 if convt else L.OUT_NHWC
        torch.cuda.synchronize()

        def run():
            st = L.stream_ptr()
            for _ in range(reps):
                L.check(lib.smirk_conv_igemm_f16x3(d, L.ptr(xs0), None, L.ptr(ws), L.ptr(sc), L.ptr(sh), None, L.ptr(out), st))
        return run

    def hog(mode, iters, blocks=512):
        def run():
            assert can.lds_hog_run(mout.data_ptr(), blocks, iters, mode, torch.cuda.current_stream().cuda_stream) == 0
        return run

    tag = os.environ.get("DIFF6_TAG", "default switches")
    aggressors = [("LDS hog 2x64 KB per CU, sleeping (no LDS traffic)", hog(0, 4000)), ("LDS hog 2x64 KB per CU, streaming ds_read_b128", hog(1, 60000)),
                  ("LDS hog, 256 blocks (1 per CU), streaming ds_read_b128", hog(1, 60000, blocks=256)),
                  ("micro 0: MFMA only", micro(0)), ("micro 1: MFMA + LDS fragment reads", micro(1)), ("micro 3: + s_barrier", micro(3)),
                  ("micro 4: MFMA + operand DMA (global_load_lds)", micro(4)), ("micro 15: reads + barrier + DMA + vmcnt(0)", micro(15)),
                  ("micro 15, 256 blocks (1 WG/CU)", micro(15, blocks=256)),
                  (f"real igemm 28^2 256->256 [{tag}]", conv(28, 256, 256)), (f"real convT 14^2 512->256 [{tag}]", conv(14, 512, 256, k=1, convt=True, reps=30))]
    with torch.no_grad():
        for name, load in aggressors:
            bad_f = bad_r = 0
            for trial in range(8):
                ev = torch.cuda.Event(); ev.record()
                with torch.cuda.stream(sG):
                    sG.wait_event(ev)
                    load()
                with torch.cuda.stream(sF):
                    sF.wait_event(ev)
                    outs = []
                    for _ in range(6):
                        f = fl.forward(e0)
                        r = rn.forward(f0["vertices"], e0["cam"])
                        outs.append((f["vertices"], r["rendered_img"]))
                torch.cuda.synchronize()
                bad_f += any(not torch.equal(v, f0["vertices"]) for v, _ in outs)
                bad_r += any(not torch.equal(i, r0["rendered_img"]) for _, i in outs)
            print(f"aggressor {name:60s}: FLAME wrong {bad_f}/8, renderer wrong {bad_r}/8", flush=True)


if __name__ == "__main__":
    main()
