#!/usr/bin/env python
"""bench.py — faces/sec of SMIRK's per-frame hot path (encode -> FLAME -> render -> generate) at 224x224 on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_PER_GPU]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the whole path over `--batch` synthetic frames per GPU (default 128 = BASELINE config 4's 1024-frame
batch sharded over 8 GPUs; weak scaling), inputs resident in HBM, followed by the asynchronous RCCL all-gather of the outputs
(vertices + rendered + re-synthesised image) which overlaps the next step.  Rank 0 prints ONE JSON line (contract in the task
statement) carrying `roofline` for the dominant kernel (HIP-event timing of every conv_igemm launch in an extra instrumented step)
and `cpu_baseline` (the CPU oracle — a port of the reference path — timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_FACE = 28.77e9          # encoder 0.929 G + FLAME 12.7 M + render ~2 M + generator 27.826 G
PEAK_FP32_MFMA = 157.3e12        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA = 2500e12          # MI355X_MICROARCH.md: bf16/fp16 dense MFMA peak (the f16x3 kernel issues 3 MFMA-flop per algorithmic flop)


def build_modules(sandbox, device):
    from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator
    from smirk_amd import synth
    synth.write_sandbox(sandbox)
    cwd = os.getcwd()
    os.chdir(sandbox)
    try:
        flame, rend = FLAME(), Renderer()
    finally:
        os.chdir(cwd)
    torch.manual_seed(1234)
    enc = SmirkEncoder()
    # random-init weights of the reference architecture (no checkpoint is obtainable offline); the shape head is zero-initialised
    # by the reference (smirk_encoder.py:61-63) — give it a small std so FLAME sees non-trivial shape coefficients
    with torch.no_grad():
        enc.shape_encoder.shape_layers[0].weight.normal_(0, 1e-3)
        enc.expression_encoder.expression_layers[0].weight.mul_(0.3)
    gen = SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
    mods = [m.to(device).eval() for m in (enc, flame, rend, gen)]
    return mods


def cpu_baseline(sandbox, n_faces):
    """The oracle (CPU port of the reference path) on `n_faces` frames; returns faces/s.  Checker code, timed beside the GPU."""
    import numpy as np
    from oracle import generator_ref as G, mobilenet_ref as M
    from oracle.flame_ref import FlameRef
    from oracle.render_ref import RendererRef
    from smirk_amd import synth
    nthr = min(os.cpu_count(), 32)          # more threads than this only thrash on a 4-16 frame sample
    torch.set_num_threads(nthr)
    os.environ["OMP_NUM_THREADS"] = str(nthr)
    encr = M.SmirkEncoderRef().eval()
    with torch.no_grad():
        encr.shape_encoder.shape_layers[0].weight.normal_(0, 1e-3)
    gsd = G.synth_state_dict(calibrate=False)
    fr, rr = FlameRef(sandbox), RendererRef(sandbox)
    img = synth.synth_images(n_faces, seed=5)
    masked = synth.synth_generator_input(n_faces, seed=5)[:, 3:]

    stages = {}

    def run():
        t0 = time.perf_counter()
        with torch.no_grad():
            e = encr(img)
        t1 = time.perf_counter()
        p = {k: v.numpy() for k, v in e.items()}
        p["cam"] = np.clip(p["cam"], [6, -.1, -.1], [10, .1, .1]).astype(np.float32)
        fl = fr.forward(p)
        t2 = time.perf_counter()
        r = rr.forward(fl["vertices"], p["cam"])
        t3 = time.perf_counter()
        x = torch.cat([torch.from_numpy(r["rendered_img"]), masked], 1)
        y = G.forward(gsd, x)
        t4 = time.perf_counter()
        stages.update(encode=t1 - t0, flame=t2 - t1, render=t3 - t2, generate=t4 - t3)
        return y

    run()                                   # warm-up (also builds raster_ref.c if needed)
    t = time.perf_counter()
    run()
    dt = time.perf_counter() - t
    return n_faces / dt, dt, nthr, {k: round(v, 3) for k, v in stages.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128, help="frames per GPU per step")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="run each batch strictly stage after stage (default: generator of batch i overlaps the front of batch i+1)")
    ap.add_argument("--given-masked", action="store_true",
                    help="feed a precomputed masked image instead of running the masking utilities (mesh sampling + masking) in the step")
    ap.add_argument("--cpu-faces", type=int, default=96, help="sample size of the CPU baseline (0 = skip)")
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from smirk_amd import _lib as L, synth
    from smirk_amd.pipeline import OutputGatherer, OverlappedPipeline, SmirkPipeline
    sandbox = tempfile.mkdtemp(prefix=f"smirk_bench_r{rank}_")
    enc, flame, rend, gen = build_modules(sandbox, dev)
    from smirk_amd import masking as MK
    cwd = os.getcwd(); os.chdir(sandbox)
    try:
        face_prob = MK.load_probabilities_per_FLAME_triangle().to(dev)
    finally:
        os.chdir(cwd)
    pipe = SmirkPipeline(enc, flame, rend, gen, face_probabilities=face_prob)
    B = args.batch
    img = synth.synth_images(B, seed=1000 + rank).to(dev)                  # resident in HBM before the timed region
    masked = synth.synth_generator_input(B, seed=1000 + rank)[:, 3:].contiguous().to(dev)
    # hull mask (1 = keep the photo, 0 = face region): a synthetic disc stands in for demo.py's mediapipe/cv2 convex hull (CPU preprocessing,
    # out of scope); the masked image itself is then produced on the GPU by the masking utilities exactly as demo.py:138-165 does
    hull = (synth.synth_generator_input(B, seed=1000 + rank)[:, 3:4] != 0).float().contiguous().to(dev)
    kw = dict(masked_img=masked) if args.given_masked else dict(hull_mask=hull)
    gather = OutputGatherer()
    runner = OverlappedPipeline(pipe) if args.overlap else None

    def finish(out):
        gather.wait()                       # previous step's all-gather must have landed before its buffers are reused
        gather.start(out)

    def step():
        """one batch of B frames enters the path; with --overlap its generator stage runs under the next batch's front stages
        (independent batches, identical results) and completes in the next step() / in drain()."""
        if runner is None:
            finish(pipe(img, with_landmarks=True, **kw))
        else:
            done = runner.submit(img, **kw)
            if done is not None:
                finish(done)

    def drain():
        if runner is not None:
            done = runner.flush()
            if done is not None:
                finish(done)
        gather.wait()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_host = time.perf_counter() - t0       # host time to ENQUEUE the K steps (launches are asynchronous)
    drain()                                 # every one of the K batches is fully processed (and gathered) inside the timed region
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- roofline of the dominant kernel: one extra instrumented step, HIP events around every conv_igemm launch --------
    roof = None
    if rank == 0:
        L.TIMER = []
        finish(pipe(img, with_landmarks=True, **kw)); gather.wait(); torch.cuda.synchronize()
        per = {}
        for name, flops, e0, e1 in L.TIMER:
            a = per.setdefault(name, [0.0, 0.0, 0])
            a[0] += flops; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
        L.TIMER = None
        dom = max(per, key=lambda k: per[k][1])
        fl, tm, n = per[dom]
        split = ",true," in dom or dom.endswith("true>")
        traffic = None                          # HBM-side bytes per launch of this kernel from the committed rocprofv3 PMC passes
        try:
            traffic = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json"))).get(dom)
        except Exception:
            pass
        peak = PEAK_F16_MFMA if split else PEAK_FP32_MFMA
        roof = {"bound": "mfma", "kernel": dom, "achieved": fl / tm / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": fl / tm / peak, "traffic": traffic, "launches_per_step": n,
                "traffic_note": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, rocprofv3 --pmc in separate passes (profiles/r01d_pmc_hbm.txt); "
                                "includes Infinity-Cache hits; algorithmic operand bytes per launch = activations in + out + weights",
                "mfma_issue_frac": (3.0 if split else 1.0) * fl / tm / peak,
                "note": ("achieved = ALGORITHMIC 2*M*N*K flop per launch / HIP-event launch time; the split-fp16 kernel issues 3 fp16 MFMAs per "
                         "product (hi.hi, hi.lo, lo.hi) so its matrix-pipe occupancy is mfma_issue_frac; peak = dense fp16 MFMA") if split
                else "achieved = algorithmic 2*M*N*K flop per launch / HIP-event launch time; peak = f32-input MFMA",
                "avg_launch_ms": tm / n * 1e3, "flop_per_launch": fl / n,
                "all_igemm": {k: {"tflops": v[0] / v[1] / 1e12, "ms_per_step": v[1] * 1e3, "launches": v[2]} for k, v in per.items()},
                "igemm_share_of_step": sum(v[1] for v in per.values()) / (dt / args.steps)}

    if rank == 0:
        faces = B * world * args.steps
        value = faces / dt
        cpu = None
        if world == 1 and args.cpu_faces > 0:
            v, cdt, nthr, stages = cpu_baseline(sandbox, args.cpu_faces)
            cpu = {"value": v, "unit": "faces/sec", "cores": nthr, "kind": "port", "host_cores": os.cpu_count(),
                   "stage_seconds": stages,
                   "sample": f"{args.cpu_faces} synthetic 224x224 frames through the CPU oracle (torch-CPU fp32 encoder+generator, numpy "
                             f"FLAME, C rasteriser with OpenMP; {nthr} threads), 1 warm-up + 1 timed pass = {cdt:.1f} s"}
        print(json.dumps({
            "metric": "faces/sec (encode+FLAME+render+generate) @224x224", "value": value, "unit": "faces/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 results; generator convs as split-fp16 x3 MFMA with f32 accumulate (fp32-class error), rest f32" if gen.precision == "f16x3" else "f32",
            "data": "synthetic",
            "config": {"workload": "full inference incl. SmirkGenerator re-synthesis (BASELINE config 4: 1024 frames / 8 GPUs)",
                       "frames_per_gpu": B, "global_batch": B * world, "image": "224x224", "parallelism": f"dp{world}",
                       "collective": "async all_gather(vertices, rendered_img, reconstructed_img)" if world > 1 else "none (1 GPU)",
                       "weights": "random-init reference architecture (no checkpoint offline)",
                       "masking": "given masked image" if args.given_masked else "utils/masking.py stage on GPU (mesh-based point sampling + masking) from a synthetic hull mask",
                       "schedule": "2-stream software pipeline: generator(batch i) || encode+FLAME+render(batch i+1)" if args.overlap else "serial stages"},
            "host_enqueue_ms_per_step": t_host / args.steps * 1e3,
            "path_tflops_per_gpu": value / world * FLOP_PER_FACE / 1e12,
            "path_frac_of_fp32_mfma_peak": value / world * FLOP_PER_FACE / PEAK_FP32_MFMA,
            "path_frac_of_f16_mfma_peak": value / world * FLOP_PER_FACE / PEAK_F16_MFMA,
            "roofline": roof, "cpu_baseline": cpu}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
