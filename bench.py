#!/usr/bin/env python
"""bench.py — faces/sec of SMIRK's per-frame hot path (encode -> FLAME -> render -> generate) at 224x224 on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload full|infer256|flame512|train64] [--global-batch G | --batch B_PER_GPU]

`--gpus N` with N > 1 launches itself as N ranks (one process per GPU, torch.distributed.run, rendezvous on 127.0.0.1); when the driver
already started it under torch.distributed.run (RANK / WORLD_SIZE set) it just joins.

Workloads (BASELINE.json configs):
  full      config 4 — full inference incl. SmirkGenerator re-synthesis on a 1024-frame batch: the batch is sharded in contiguous slices
            over the N ranks (strong scaling: 1024 frames per step in total, whatever N is), every rank runs its shard in one pass (or in
            `--micro-batch` sized passes), and the outputs (vertices + rendered + re-synthesised image) are all-gathered with RCCL, asynchronously, so that the
            gather of one micro-batch overlaps the compute of the next.  `--batch B` instead fixes B frames PER GPU (weak scaling).
  infer256  config 3 — encoder + FLAME + renderer, 256 frames.
  flame512  config 2 — FLAME only, 512 random parameter vectors -> 5023 vertices.
One "step" = one pass of the workload over its batch, inputs resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

`roofline`: every kernel launch of one extra instrumented pass is bracketed by HIP events on its launch stream by libsmirk_hip.so's own
launch profiler, which also reports the kernel instantiation it launched and the launch's algorithmic flop / bytes; the dominant kernel's
ALGORITHMIC flop (2*M*N*K) per launch / its mean launch time is `achieved`.  `traffic` comes from
two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) that this script starts on itself (`--traffic measure`, the default
on one GPU), corrected as MI355X_MICROARCH.md prescribes for gfx950 (2*FETCH_SIZE + WRITE_SIZE, KiB units).
`cpu_baseline`: the CPU oracle (a port of the reference path) timed on this box's host cores on a bounded sample — checker code,
reported beside the GPU number.
"""
import argparse
import glob
import hashlib
import json
import os
import shutil
import socket
import statistics
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# One hardware queue per HIP stream (read by the HIP runtime when it initialises, i.e. before `import torch` touches the device).  The pipeline uses up to 11
# streams (front, 3 backbones, 2 generator streams with a chain side stream each, the gather / copy streams); the runtime's default of 4 hardware queues makes
# streams SHARE a queue, whose packets retire in order — the 128-frame step's trace (profiles/r04o_timeline_shard128.txt) shows each backbone stream starting
# only when a generator kernel of the queue it landed on had finished.  smirk_amd/__init__.py sets the same default for library users.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

FLOP_PER_FACE = 28.77e9          # encoder 0.929 G + FLAME 12.7 M + render ~2 M + generator 27.826 G
FLOP_PER_FACE_INFER = 0.929e9 + 12.7e6 + 2e6
FLOP_PER_FACE_FLAME = 12.7e6
# config 5 (cycle-path training step): generator forward + data gradient + weight gradient (3 x 27.826 G), the three encoders forward (0.929 G), the
# expression encoder's data + weight gradients and the shape encoder's data gradient (3 x ~0.41 G), FLAME + renderer three times
FLOP_PER_FACE_TRAIN = 3 * 27.826e9 + 0.929e9 + 3 * 0.41e9 + 3 * 14.7e6
PEAK_FP32_MFMA = 157.3e12        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA = 2500e12          # MI355X_MICROARCH.md: bf16/fp16 dense MFMA peak (the f16x3 kernel issues 3 MFMA-flop per algorithmic flop)
PEAK_HBM = 8.0e12
MICRO_BATCH = 1024               # frames per pass of the path = a rank's whole shard: 288 GB of HBM hold the 1024-frame job's ~64 GB of activations, and one big
                                 # pass beats several tile-quantisation-tuned ones (profiles/r02z_microbatch_sweep.txt: 167 -> 7494, 334 -> 7709, 512 -> 7741-7769,
                                 # 1024 -> 7816 faces/s; 167 = whole rounds of tiles at 14^2 / 28^2 / 56^2 but six passes + a 22-frame tail)
METRIC = {"full": "faces/sec (encode+FLAME+render+generate) @224x224",
          "infer256": "faces/sec (encode+FLAME+render) @224x224",
          "flame512": "faces/sec (FLAME-only: shape,exp,pose,jaw -> 5023 vertices)",
          "train64": "faces/sec (cycle-path training step: render + generate + re-encode + cycle loss + backward + optimiser) @224x224"}


# ------------------------------------------------------------------------------------------------------------------------------
# launch plumbing
# ------------------------------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=("full", "infer256", "flame512", "train64"), default="full")
    ap.add_argument("--global-batch", type=int, default=None, help="frames per step over ALL GPUs (strong scaling; default 1024 / 256 / 512 by workload)")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (weak scaling; overrides --global-batch)")
    ap.add_argument("--micro-batch", type=int, default=None,
                    help="frames per pass of the path (default: the rank's whole shard, up to 1024 frames; smaller values walk the shard in several passes with\n"
                         "the generator of pass i overlapping the front end of pass i+1)")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="run each micro-batch strictly stage after stage (default: generator of batch i overlaps the front of batch i+1)")
    ap.add_argument("--train-arith", choices=("f16x3", "f16x1"), default="f16x3",
                    help="train64: arithmetic of the two CNNs' convolutions and their gradients: f16x3 = split-fp16 x3 MFMA (fp32-class, the parity mode), "
                         "f16x1 = one fp16 MFMA per product block, the 16-bit class BASELINE config 5 names (the reference: bf16 autocast)")
    ap.add_argument("--train-graphs", dest="train_graphs", action="store_true",
                    help="train64: replay the two CNNs' forward / backward from HIP graphs instead of launching kernel by kernel (measured: host enqueue 43 -> 27 ms "
                         "per step, but the replay of ~1300 chained kernel nodes runs 3.8 ms slower on the GPU, which is the bound: 53.1 vs 49.4 ms)")
    ap.add_argument("--infer-lanes", type=int, default=1,
                    help="infer256: consecutive batches rotate over this many streams (pipeline.RotatingPipeline; 1 = one batch at a time).  Measured equal: 39.7 / 39.4 / "
                         "40.2 k faces/s with 1 / 2 / 3 lanes (profiles/r04r_infer_lanes.txt) - the step is bound by the GPU time of its kernels, not by its critical path")
    ap.add_argument("--generator-streams", type=int, default=None,
                    help="generator stages of consecutive passes alternate over this many streams (default 2: the partial last round of one pass's deep layers "
                         "overlaps the next pass's; with one hardware queue per stream +6 %% at 128 frames, +1.5 %% at 1024, profiles/r04q_streams_chains.txt)")
    ap.add_argument("--given-masked", action="store_true",
                    help="feed a precomputed masked image instead of running the masking utilities (mesh sampling + masking) in the step")
    ap.add_argument("--cpu-faces", type=int, default=24, help="sample size of the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-passes", type=int, default=5, help="timed passes of the CPU baseline (median reported; the `also` children use 3)")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the CPU baseline with os.cpu_count() threads (takes ~15 min on a 256-thread host: the sample is too small for that many threads)")
    ap.add_argument("--traffic", choices=("measure", "file", "off"), default=None,
                    help="roofline.traffic: run two rocprofv3 --pmc passes of this script (default at 1 GPU), read profiles/pmc_traffic.json, or skip")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="default `full` run on one GPU: skip the short timed regions of the other BASELINE configs that are reported under \"also\" "
                         "(flame512, infer256, train64 in both arithmetics, the 128-frame shard of the 8-GPU job with the RCCL gather enqueued)")
    ap.add_argument("--flame-basis", choices=("random", "smooth"), default="random",
                    help="synthetic FLAME blendshape basis: 'random' = SURVEY.md 8(d) (i.i.d. directions: ~21 x 22-pixel triangle boxes); 'smooth' = low-frequency "
                         "fields like a real shape model (~4-pixel triangles) - only the rasteriser's share of the step changes")
    ap.add_argument("--force-collective", action="store_true",
                    help="1 GPU only: create a world-size-1 RCCL group and really enqueue the output all-gather / gradient all-reduce of every step (what a rank of "
                         "the N-GPU job does besides its shard); use with --global-batch 128 to measure the 8-GPU per-rank regime on one MI355X")
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--plumbing-test", action="store_true", help=argparse.SUPPRESS)   # CPU/gloo stub of the path: tests/test_distributed_cpu.py
    ap.add_argument("--pmc-inner", action="store_true", help=argparse.SUPPRESS)       # the run that rocprofv3 wraps (no JSON line, no baseline)
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec as N ranks under torch.distributed.run (one process per GPU)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


# ------------------------------------------------------------------------------------------------------------------------------
# modules and inputs
# ------------------------------------------------------------------------------------------------------------------------------
def build_modules(sandbox, device, want=("enc", "flame", "rend", "gen")):
    """Random-init weights of the reference architecture (no checkpoint is obtainable offline), He-initialised so that activations stay
    O(1) through all 30+ layers (nn.Conv2d's default init shrinks them by ~2.4x per layer); every output is checked finite after warm-up."""
    import torch
    from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator
    import synthdata as synth
    synth.write_sandbox(sandbox, basis=os.environ.get("SMIRK_BENCH_FLAME_BASIS", "random"))
    cwd = os.getcwd()
    os.chdir(sandbox)
    try:
        flame = FLAME() if "flame" in want else None
        rend = Renderer() if "rend" in want else None
    finally:
        os.chdir(cwd)
    enc = gen = None
    if "enc" in want:
        enc = SmirkEncoder()
        synth.he_init_(enc, seed=1234)
    if "gen" in want:
        gen = SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
        synth.he_init_(gen, seed=4321)
    mods = [m.to(device).eval() if m is not None else None for m in (enc, flame, rend, gen)]
    if enc is not None:
        # the heads are rescaled on the features the random backbones actually produce, so that pose / camera / shape / expression land in
        # the ranges of the trained network (the reference's own head init — zero shape head, 1e-3 pose head — assumes trained-scale features)
        synth.calibrate_encoder_heads_(mods[0], device)
    return mods


def assert_finite(out, keys, where):
    import torch
    for k in keys:
        if k in out and torch.is_tensor(out[k]) and out[k].is_floating_point():
            if not bool(torch.isfinite(out[k]).all()):
                raise SystemExit(f"bench.py: non-finite values in '{k}' ({where}) — the timed run would be measuring inf/NaN propagation")


def output_stats(out):
    s = {}
    if "reconstructed_img" in out:
        y = out["reconstructed_img"]
        s["reconstructed_mean"], s["reconstructed_std"] = float(y.mean()), float(y.std())
        s["reconstructed_saturated_frac"] = float(((y < 1e-4) | (y > 1 - 1e-4)).float().mean())
    if "rendered_img" in out:
        s["rendered_coverage"] = float((out["rendered_img"][:, 0] != 0).float().mean())
    if "vertices" in out:
        s["vertices_absmax"] = float(out["vertices"].abs().max())
        s["mesh_sane"] = bool(s["vertices_absmax"] < 1.0)          # the head template spans about +-0.15; a random encoder could blow it up
    return s


# ------------------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = a port of the reference path; checker code, timed beside the GPU on a bounded sample)
# ------------------------------------------------------------------------------------------------------------------------------
def cpu_baseline(sandbox, workload, n_faces, threads=None, all_cores=False, passes=5):
    """threads=None: the 32-thread measurement.  all_cores=True (--cpu-all-cores) adds, under "all_cores", the same sample with
    torch.set_num_threads(os.cpu_count()) as BASELINE.md section 3 asks - opt-in, because on the 256-thread GPU host that leg oversubscribes
    the torch-CPU convolutions of a 24-frame sample so badly (0.19 faces/s against 7.8 at 32 threads: profiles/r03z_bench_full.json) that it
    alone takes 15 minutes.  threads=N: that thread count only."""
    if threads is None:
        base = cpu_baseline(sandbox, workload, n_faces, threads=min(os.cpu_count(), 32), passes=passes)
        if all_cores and os.cpu_count() > base["cores"]:
            allc = cpu_baseline(sandbox, workload, n_faces, threads=os.cpu_count(), passes=passes)
            base["all_cores"] = {k: allc[k] for k in ("value", "unit", "cores", "stage_seconds", "sample")}
        elif os.cpu_count() > base["cores"]:
            base["all_cores"] = ("not run by default (opt in with --cpu-all-cores): the sample is a few dozen frames, and with one torch thread per hardware thread of "
                                 "the 256-thread host the per-layer fork/join of its convolutions dominates; `cores` is the thread count actually used, min(os.cpu_count(), 32)")
        return base
    import numpy as np
    import torch
    from oracle import generator_ref as G, mobilenet_ref as M
    from oracle.flame_ref import FlameRef
    from oracle.render_ref import RendererRef
    import synthdata as synth
    nthr = int(threads)
    torch.set_num_threads(nthr)
    os.environ["OMP_NUM_THREADS"] = str(nthr)
    fr = FlameRef(sandbox)
    stages = {}
    if workload == "train64":
        # the cycle path on the CPU oracle: numpy FLAME + C rasteriser (x3), torch-CPU autograd through the restated generator and encoders in train mode
        import torch.nn.functional as F
        n_faces = min(n_faces, 8)
        rr = RendererRef(sandbox)
        encr = M.SmirkEncoderRef(); encr.load_state_dict(M.synth_encoder_state_dict()); encr.train()
        for m in (encr.pose_encoder, encr.shape_encoder):
            for p in m.parameters():
                p.requires_grad_(False)
        gsd = G.synth_state_dict()
        gp, gb = G.split_state_dict(gsd)
        fp = synth.synth_flame_params(n_faces, seed=5)
        cam = synth.synth_cam(n_faces, seed=5)
        masked = synth.synth_generator_input(n_faces, seed=5)[:, 3:]
        tgt = dict(expression_params=torch.randn(n_faces, 50), jaw_params=torch.rand(n_faces, 3) * .2, eyelid_params=torch.rand(n_faces, 2),
                   shape_params=torch.randn(n_faces, 300) * .5)

        def run():
            t0 = time.perf_counter()
            for _ in range(2):                                     # the encoder estimate and the augmented parameters
                r = rr.forward(fr.forward(fp)["vertices"], cam)
            t1 = time.perf_counter()
            y = G.train_forward(gp, gb, torch.cat([torch.from_numpy(r["rendered_img"]), masked], 1))
            o = encr(y)
            loss = F.mse_loss(o["expression_params"], tgt["expression_params"]) + 10 * F.mse_loss(o["jaw_params"], tgt["jaw_params"]) + \
                10 * F.mse_loss(o["eyelid_params"], tgt["eyelid_params"]) + F.mse_loss(o["shape_params"], tgt["shape_params"])
            t2 = time.perf_counter()
            for p in list(gp.values()) + list(encr.parameters()):
                p.grad = None
            loss.backward()
            t3 = time.perf_counter()
            pn = {k: v.detach().numpy() for k, v in o.items()}
            pn["cam"] = np.clip(pn["cam"], [6, -.1, -.1], [10, .1, .1]).astype(np.float32)
            rr.forward(fr.forward(pn)["vertices"], pn["cam"])
            stages.update(flame_render_x2=t1 - t0, forward=t2 - t1, backward=t3 - t2, flame_render_reencoded=time.perf_counter() - t3)

        run()
        times, keep = [], {}
        for _ in range(3):
            t = time.perf_counter()
            run()
            times.append(time.perf_counter() - t)
            keep = dict(stages) if times[-1] <= min(times) else keep
        med = statistics.median(times)
        return {"value": n_faces / med, "unit": "faces/sec", "cores": nthr, "kind": "port", "host_cores": os.cpu_count(),
                "stage_seconds": {k: round(v, 4) for k, v in keep.items()},
                "sample": f"{n_faces} synthetic frames through the CPU oracle's cycle step (numpy FLAME + C rasteriser x3, torch-CPU fp32 autograd through the "
                          f"restated generator and encoders in train mode, no optimiser step; {nthr} threads), 1 warm-up + 3 timed passes, median {med:.2f} s"}
    if workload == "flame512":
        # the reference's FLAME.forward is torch (einsum / bmm on the host's BLAS): oracle/flame_torch_ref.py restates exactly that op sequence, pinned against
        # the real class by tests/golden/flame_golden.npz — SURVEY 6 measured the reference class itself at ~5.1 k faces/s on 8 cores; the numpy port
        # (oracle/flame_ref.py, the parity checker) is timed once beside it
        from oracle.flame_torch_ref import FlameTorchRef
        p = synth.synth_flame_params(n_faces, seed=5)
        ft = FlameTorchRef(sandbox).eval()
        pt = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()}
        t0 = time.perf_counter()
        fr.forward(p)
        numpy_port_s = time.perf_counter() - t0

        def run():
            t0 = time.perf_counter()
            with torch.no_grad():
                ft(pt)
            stages["flame_torch"] = time.perf_counter() - t0
            stages["flame_numpy_port_once"] = numpy_port_s
    else:
        encr = M.SmirkEncoderRef().eval()
        with torch.no_grad():
            encr.shape_encoder.shape_layers[0].weight.normal_(0, 1e-3)
        gsd = G.synth_state_dict(calibrate=False) if workload == "full" else None
        rr = RendererRef(sandbox)
        img = synth.synth_images(n_faces, seed=5)
        masked = synth.synth_generator_input(n_faces, seed=5)[:, 3:]

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
            stages.update(encode=t1 - t0, flame=t2 - t1, render=t3 - t2)
            if gsd is not None:
                G.forward(gsd, torch.cat([torch.from_numpy(r["rendered_img"]), masked], 1))
                stages["generate"] = time.perf_counter() - t3

    run()                                   # warm-ups (the first also builds raster_ref.c if needed)
    if passes >= 5:
        run()
    times, keep = [], {}
    for _ in range(max(1, passes)):
        t = time.perf_counter()
        run()
        times.append(time.perf_counter() - t)
        keep = dict(stages) if times[-1] <= min(times) else keep
    med = statistics.median(times)
    what = {"full": "torch-CPU fp32 encoder + generator, numpy FLAME, C rasteriser with OpenMP", "infer256": "torch-CPU fp32 encoder, numpy FLAME, C rasteriser with OpenMP",
            "flame512": "torch-CPU fp32 restatement of the reference FLAME.forward, oracle/flame_torch_ref.py"}[workload]
    return {"value": n_faces / med, "unit": "faces/sec", "cores": nthr, "kind": "port", "host_cores": os.cpu_count(),
            "stage_seconds": {k: round(v, 4) for k, v in keep.items()},
            "sample": f"{n_faces} synthetic frames through the CPU oracle ({what}; {nthr} threads), {2 if passes >= 5 else 1} warm-up(s) + {max(1, passes)} timed passes, median {med:.2f} s "
                      f"(min {min(times):.2f}, max {max(times):.2f})"}


# ------------------------------------------------------------------------------------------------------------------------------
# roofline helpers
# ------------------------------------------------------------------------------------------------------------------------------
def kernel_sources_sha():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "smirk_amd", "csrc", "*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_parse(db_dir, counter):
    import sqlite3
    dbs = glob.glob(os.path.join(db_dir, "**", "*.db"), recursive=True)
    if not dbs:
        raise RuntimeError(f"no rocpd database under {db_dir}")
    c = sqlite3.connect(dbs[0])
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    agg = {}
    for k, v in c.execute(f"select {name_col}, value from counters_collection where counter_name = ?", (counter,)):
        k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace(", ", ",")
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(v)
    return agg


def measure_traffic(args):
    """Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE — separate runs, kernel-trace only) over one step of this same workload.
    Returns {kernel: bytes per launch} with the gfx950 correction (FETCH_SIZE counts 64 B per 128-B request: x2), or raises."""
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 not on PATH")
    inner = [sys.executable, os.path.abspath(__file__), "--pmc-inner", "--workload", args.workload, "--steps", "1", "--warmup", "1",
             "--micro-batch", str(args.micro_batch), "--no-overlap"]
    inner += ["--batch", str(min(args.micro_batch, per_rank_batch(args, 1)))] if args.workload == "full" else []
    if args.given_masked:
        inner.append("--given-masked")
    out = {}
    # the counter passes launch the generator's deep section as ONE chain, like the instrumented (launch-profiler) pass whose per-launch flop / bytes the
    # traffic is compared with (the timed steps may run it as two half-batch chains: half-sized launches)
    env = dict(os.environ, TMPDIR="/tmp", SMIRK_GEN_SPLIT_CHAINS="0")
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"smirk_pmc_{c}_", dir="/tmp")
        r = subprocess.run(["rocprofv3", "--pmc", c, "--kernel-trace", "-d", d, "-o", "p", "--"] + inner, cwd="/tmp", env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        if r.returncode != 0:
            raise RuntimeError(f"rocprofv3 --pmc {c} failed rc={r.returncode}: {r.stdout.decode(errors='replace')[-400:]}")
        out[c] = pmc_parse(d, c)
        shutil.rmtree(d, ignore_errors=True)
    f, w = out["FETCH_SIZE"], out["WRITE_SIZE"]
    table = {}
    for k, (n, s) in f.items():
        wn, ws = w.get(k, (1, 0.0))
        table[k] = {"launches": n, "fetch_kb": s / n, "write_kb": ws / max(wn, 1), "bytes_per_launch": (2 * s / n + ws / max(wn, 1)) * 1024}
    return table


def rocprofv3_avg(workload, kernel, variant=""):
    """rocprofv3 --kernel-trace --stats' own per-launch average of `kernel` for this workload's command line, from the committed summary of the SAME kernel
    sources (tools/rocprof_summary.py writes the JSON); None-with-reason otherwise.  variant="_serial": the summary of the SERIAL schedule (`--no-overlap`,
    the generator's deep section as one whole-batch chain: the state roofline.achieved / frac describe), so that `frac` can be recomputed from profiles/ alone."""
    workload = workload + variant
    f = os.path.join(REPO, "profiles", f"rocprofv3_kernel_avg_{workload}.json")
    try:
        j = json.load(open(f))
    except Exception:                       # noqa: BLE001
        return {"avg_launch_us": None, "why": f"no profiles/rocprofv3_kernel_avg_{workload}.json"}
    if j.get("kernel_sources_sha") != kernel_sources_sha():
        return {"avg_launch_us": None, "why": "profiles/rocprofv3_kernel_avg_%s.json was taken on other kernel sources (sha %s)" % (workload, j.get("kernel_sources_sha"))}
    key = kernel.split("[")[0].replace(", ", ",")
    k = j["kernels"].get(key)
    if k is None:
        return {"avg_launch_us": None, "why": f"{key} not in {os.path.basename(f)}"}
    return {"avg_launch_us": k["avg_us"], "calls": k["calls"], "command": j.get("command"), "file": "profiles/" + str(j.get("summary_file")),
            "kernel_sources_sha": j["kernel_sources_sha"],
            "note": ("rocprofv3's average over the serial schedule (--no-overlap, SMIRK_GEN_SPLIT_CHAINS=0: one whole-batch chain, nothing else on the device) - the state "
                     "`achieved` / `frac` describe: flop_per_launch / this average reproduces `frac` from profiles/ alone" if variant == "_serial" else
                     "rocprofv3's average over the timed-region schedule (two half-batch chains in the deep section, generator overlapping the next pass's front end)")}


def per_rank_batch(args, world):
    if args.batch is not None:
        return args.batch
    if args.workload == "train64" and args.global_batch is None:
        return 64                                                  # config 5 is 64 frames PER GPU (x8 data-parallel): weak scaling by definition
    g = args.global_batch or {"full": 1024, "infer256": 256, "flame512": 512, "train64": 64}[args.workload]
    return max(1, g // world)


# ------------------------------------------------------------------------------------------------------------------------------
# the workloads
# ------------------------------------------------------------------------------------------------------------------------------
class Workload:
    """step()/drain() + bookkeeping shared by the three workloads; `last` holds the most recent outputs of a micro-batch."""
    keys = ()
    last = None

    def step(self):
        raise NotImplementedError

    def drain(self):
        pass


class FullWorkload(Workload):
    keys = ("vertices", "rendered_img", "reconstructed_img", "cam", "expression_params")

    def __init__(self, args, dev, rank, world, sandbox):
        import torch
        from smirk_amd import masking as MK
        import synthdata as synth
        from smirk_amd.pipeline import OutputGatherer, OverlappedPipeline, SmirkPipeline
        enc, flame, rend, gen = build_modules(sandbox, dev)
        self.gen = gen
        cwd = os.getcwd(); os.chdir(sandbox)
        try:
            face_prob = MK.load_probabilities_per_FLAME_triangle().to(dev)
        finally:
            os.chdir(cwd)
        self.pipe = SmirkPipeline(enc, flame, rend, gen, face_probabilities=face_prob)
        self.B = per_rank_batch(args, world)
        mb = min(args.micro_batch, self.B)
        self.slices = [(i, min(i + mb, self.B)) for i in range(0, self.B, mb)]
        B = self.B
        # a distinct seeded shard per rank, resident in HBM before the timed region; generated in chunks to bound host memory
        self.img = torch.cat([synth.synth_images(hi - lo, seed=1000 + 97 * rank + lo).to(dev) for lo, hi in self.slices])
        gi = [synth.synth_generator_input(hi - lo, seed=1000 + 97 * rank + lo) for lo, hi in self.slices]
        self.masked = torch.cat([g[:, 3:].contiguous().to(dev) for g in gi])
        # hull mask (1 = keep the photo, 0 = face region): a synthetic disc stands in for demo.py's mediapipe/cv2 convex hull (CPU
        # preprocessing, out of scope); the masked image itself is produced on the GPU by the masking utilities as demo.py:138-165 does
        self.hull = torch.cat([(g[:, 3:4] != 0).float().contiguous().to(dev) for g in gi])
        assert self.img.shape[0] == B
        self.given = args.given_masked
        self.gather = OutputGatherer(force_collective=bool(getattr(args, "force_collective", False)))
        self.runner = OverlappedPipeline(self.pipe, generator_streams=args.generator_streams) if args.overlap else None

    def _kw(self, lo, hi):
        return dict(masked_img=self.masked[lo:hi]) if self.given else dict(hull_mask=self.hull[lo:hi])

    def _finish(self, out):
        self.gather.wait()                  # the previous micro-batch's all-gather must have landed before its buffers are reused
        self.gather.start(out)
        self.last = out

    def step(self):
        """the rank's whole shard enters the path, one micro-batch after the other; with overlap the generator stage of a micro-batch runs
        under the front stages of the next one (independent frames, identical results) and completes in the next submit / in drain()."""
        for lo, hi in self.slices:
            if self.runner is None:
                self._finish(self.pipe(self.img[lo:hi], with_landmarks=True, **self._kw(lo, hi)))
            else:
                done = self.runner.submit(self.img[lo:hi], **self._kw(lo, hi))
                if done is not None:
                    self._finish(done)

    def drain(self):
        if self.runner is not None:
            while True:
                done = self.runner.flush()
                if done is None:
                    break
                self._finish(done)
        self.gather.wait()

    def instrumented(self):
        lo, hi = self.slices[0]
        self._finish(self.pipe(self.img[lo:hi], with_landmarks=True, **self._kw(lo, hi)))
        self.gather.wait()

    def instrumented_pipeline(self):
        """three passes through the two-stage software pipeline, as the timed steps run them: the generator of a pass shares the GPU with the front end of
        the next one, which stretches every launch (rocprofv3's per-kernel averages of this command are taken in that state)"""
        if self.runner is None:
            return False
        lo, hi = self.slices[0]
        for _ in range(3):
            done = self.runner.submit(self.img[lo:hi], **self._kw(lo, hi))
            if done is not None:
                self._finish(done)
        self.drain()
        return True


class InferWorkload(Workload):
    keys = ("vertices", "rendered_img", "cam", "expression_params", "landmarks_fan")

    def __init__(self, args, dev, rank, world, sandbox):
        import torch
        import synthdata as synth
        from smirk_amd.pipeline import RotatingPipeline, SmirkPipeline
        enc, flame, rend, _ = build_modules(sandbox, dev, want=("enc", "flame", "rend"))
        self.pipe = SmirkPipeline(enc, flame, rend, None)
        self.runner = RotatingPipeline(self.pipe, lanes=args.infer_lanes) if args.infer_lanes > 1 else None
        self.B = per_rank_batch(args, world)
        mb = min(256, self.B)
        self.slices = [(i, min(i + mb, self.B)) for i in range(0, self.B, mb)]
        self.img = torch.cat([synth.synth_images(hi - lo, seed=2000 + 97 * rank + lo).to(dev) for lo, hi in self.slices])

    def step(self):
        for lo, hi in self.slices:
            if self.runner is None:
                self.last = self.pipe(self.img[lo:hi], with_landmarks=True)
            else:                                                 # batches are independent: batch i's raster tail overlaps batch i+1's backbones
                done = self.runner.submit(self.img[lo:hi], with_landmarks=True)
                self.last = done if done is not None else self.last

    def drain(self):
        while self.runner is not None and (done := self.runner.flush()) is not None:
            self.last = done

    def instrumented(self):
        for lo, hi in self.slices:
            self.last = self.pipe(self.img[lo:hi], with_landmarks=True)


class FlameWorkload(Workload):
    keys = ("vertices", "landmarks_fan", "landmarks_mp")

    def __init__(self, args, dev, rank, world, sandbox):
        import torch
        import synthdata as synth
        _, flame, _, _ = build_modules(sandbox, dev, want=("flame",))
        self.flame = flame
        self.B = per_rank_batch(args, world)
        p = synth.synth_flame_params(self.B, seed=3000 + rank)
        self.params = {k: torch.from_numpy(v).to(dev) for k, v in p.items()}

    def step(self):
        import torch
        with torch.no_grad():
            self.last = self.flame.forward(self.params)

    instrumented = step


class TrainWorkload(Workload):
    """BASELINE config 5: the cycle path of smirk_trainer.py:184-332 + the backward / clip / optimiser part of `step` (:365-376), 64 frames per GPU.
    The parameter augmentation (random draws on [B, 50] tensors, trainer code) is done once at set-up; everything the four modules compute per step is
    inside step(): FLAME + renderer of the encoder estimate and of the augmented parameters, point sampling / pixel transfer / masking, generator and
    encoders in TRAIN mode, FLAME + renderer of the re-encoded parameters, cycle loss, backward, gradient all-reduce (N > 1), clip, two Adam steps."""
    keys = ("reconstructed_img", "loss")

    def __init__(self, args, dev, rank, world, sandbox):
        import torch
        from smirk_amd import masking as MK
        import synthdata as synth
        enc, flame, rend, gen = build_modules(sandbox, dev)
        self.enc, self.flame, self.rend, self.gen, self.MK = enc, flame, rend, gen, MK
        cwd = os.getcwd(); os.chdir(sandbox)
        try:
            self.face_prob = MK.load_probabilities_per_FLAME_triangle().to(dev)
        finally:
            os.chdir(cwd)
        self.B = B = per_rank_batch(args, world)
        self.slices = [(0, B)]
        chunks = [(i, min(i + 32, B)) for i in range(0, B, 32)]
        self.img = torch.cat([synth.synth_images(hi - lo, seed=5000 + 97 * rank + lo).to(dev) for lo, hi in chunks])
        self.mask = torch.cat([(synth.synth_generator_input(hi - lo, seed=5000 + 97 * rank + lo)[:, 3:4] != 0).float().contiguous().to(dev) for lo, hi in chunks])
        with torch.no_grad():                                        # step1's encoder estimate (smirk_trainer.py:95-96), detached as step2 receives it
            self.enc_out = {k: v.detach().clone() for k, v in enc(self.img).items()}
        g = torch.Generator(device="cpu").manual_seed(6000 + rank)
        f = {k: v.clone() for k, v in self.enc_out.items()}
        f["expression_params"] = (f["expression_params"] + 0.5 * torch.randn(B, f["expression_params"].shape[1], generator=g).to(dev)).clamp(-4, 4)
        f["jaw_params"] = f["jaw_params"] + 0.05 * torch.randn(B, 3, generator=g).to(dev) * torch.tensor([1., .1, .1], device=dev)
        f["jaw_params"][:, 0].clamp_(0.0, 0.5)
        f["eyelid_params"] = (f["eyelid_params"] + 0.25 * (2 * torch.rand(B, 2, generator=g).to(dev) - 1)).clamp(0, 1)
        self.feats = f
        enc.train(); gen.train()
        for m in (enc.pose_encoder, enc.shape_encoder):              # config_train.yaml:41-43: only the expression encoder is optimised
            for p in m.parameters():
                p.requires_grad_(False)
        self.enc_params = [p for p in enc.parameters() if p.requires_grad]
        self.gen_params = list(gen.parameters())
        self.opt_e = torch.optim.Adam(self.enc_params, lr=0.25e-3)   # base_trainer.py:36-55
        self.opt_g = torch.optim.Adam(self.gen_params, lr=1e-3)
        self.world = world
        self.buckets = 0
        self.gen_step, self.enc_step = gen, enc
        from smirk_amd.cycle import set_train_arith
        set_train_arith(gen, enc, getattr(args, "train_arith", "f16x3"))
        self.graphs = bool(getattr(args, "train_graphs", False))
        self.force_collective = bool(getattr(args, "force_collective", False))
        if self.graphs:                                              # forward + backward of both CNNs as four HIP graphs (smirk_amd/cycle.py)
            from smirk_amd.cycle import graph_cycle_modules
            self.gen_step, self.enc_step = graph_cycle_modules(gen, enc, torch.zeros(B, 6, 224, 224, device=dev), torch.zeros(B, 3, 224, 224, device=dev))

    def step(self):
        import torch
        from smirk_amd.cycle import allreduce_gradients, cycle_forward, render_second_path
        rendered, masked = render_second_path(self.flame, self.rend, self.enc_out, self.feats, self.img, self.mask, self.face_prob, self.MK)
        loss, recon, rf = cycle_forward(self.gen_step, self.enc_step, rendered, masked, self.feats)
        fo = self.flame.forward(rf)                                  # smirk_trainer.py:299-300 (feeds the visualisation grid only)
        self.rend.forward(fo['vertices'], rf['cam'])
        self.opt_e.zero_grad(set_to_none=True); self.opt_g.zero_grad(set_to_none=True)
        loss.backward()
        self.buckets = allreduce_gradients(self.enc_params + self.gen_params, force_collective=self.force_collective)
        torch.nn.utils.clip_grad_norm_(self.gen_params, 0.1)
        self.opt_e.step(); self.opt_g.step()
        self.last = {"reconstructed_img": recon.detach(), "loss": loss.detach().reshape(1)}

    def instrumented(self):
        """the launch profiler sees launches, not graph replays: the instrumented pass runs the same step kernel by kernel"""
        g, e = self.gen_step, self.enc_step
        self.gen_step, self.enc_step = self.gen, self.enc
        try:
            self.step()
        finally:
            self.gen_step, self.enc_step = g, e


class PlumbingWorkload(Workload):
    """CPU/gloo stand-in for the path used ONLY by tests/test_distributed_cpu.py to drive this script's rank / shard / micro-batch /
    gather / timing bookkeeping end to end without a GPU.  It computes nothing of SMIRK and its JSON line is labelled as such."""
    keys = ("vertices", "rendered_img", "reconstructed_img")

    def __init__(self, args, dev, rank, world, sandbox):
        import torch
        from smirk_amd.pipeline import OutputGatherer, shard_bounds
        self.B = per_rank_batch(args, world)
        self.lo, _ = shard_bounds(self.B * world, rank, world)
        mb = min(args.micro_batch, self.B)
        self.slices = [(i, min(i + mb, self.B)) for i in range(0, self.B, mb)]
        self.ids = torch.arange(self.lo, self.lo + self.B, dtype=torch.float32)
        self.gather = OutputGatherer()
        self.seen = []

    def step(self):
        for lo, hi in self.slices:
            v = self.ids[lo:hi]
            out = {"vertices": v[:, None, None].expand(-1, 2, 3).contiguous(), "rendered_img": v[:, None, None, None].expand(-1, 1, 2, 2).contiguous(),
                   "reconstructed_img": (v * 2)[:, None, None, None].expand(-1, 1, 2, 2).contiguous()}
            self.gather.wait()
            self.gather.start(out)
            self.last = out
            bufs = self.gather.wait()
            self.seen.append(sorted(set(bufs["vertices"][:, 0, 0].tolist())))

    def drain(self):
        self.gather.wait()


class PlumbingTrainWorkload(Workload):
    """CPU/gloo stand-in for config 5's data-parallel step, used ONLY by tests/test_distributed_cpu.py: drives the part of TrainWorkload.step that is not
    kernel work — the bucketed gradient all-reduce of smirk_amd.cycle.allreduce_gradients over the rank group (frozen parameters left out, a missing
    gradient counted as zero), the optimiser step on the averaged gradients — on a stub network.  Computes nothing of SMIRK; labelled as such."""
    keys = ()

    def __init__(self, args, dev, rank, world, sandbox):
        import torch
        self.B = args.batch if args.batch is not None else (args.global_batch // world if args.global_batch else 64)
        self.rank, self.world = rank, world
        torch.manual_seed(0)                                         # identical replicas on every rank, like DDP after its broadcast
        self.net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Linear(32, 32), torch.nn.Linear(32, 8))
        for p in self.net[1].parameters():
            p.requires_grad_(False)                                  # the frozen pose / shape encoders
        self.params = list(self.net.parameters())
        self.opt = torch.optim.SGD([p for p in self.params if p.requires_grad], lr=0.1)
        self.x = torch.arange(self.B * 16, dtype=torch.float32).reshape(self.B, 16) / (self.B * 16) * (rank + 1)
        self.buckets, self.checked, self.last = 0, [], None

    def step(self):
        import torch
        from smirk_amd.cycle import allreduce_gradients
        self.opt.zero_grad(set_to_none=True)
        self.net(self.x).square().mean().backward()
        local = [None if p.grad is None else p.grad.clone() for p in self.params]
        self.buckets = allreduce_gradients(self.params, bucket_bytes=1024)          # tiny buckets: several all-reduces per step
        # every rank's replica sees the same inputs scaled by (rank + 1): the averaged gradient can be checked against a local recomputation
        ref = []
        for r in range(self.world):
            net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Linear(32, 32), torch.nn.Linear(32, 8))
            net.load_state_dict(self.net.state_dict())
            net(self.x / (self.rank + 1) * (r + 1)).square().mean().backward()
            ref.append([p.grad for p in net.parameters()])
        ok = all(torch.allclose(p.grad, sum(g[i] for g in ref) / self.world, rtol=1e-5, atol=1e-7)
                 for i, p in enumerate(self.params) if p.requires_grad)
        self.checked.append(bool(ok and all(l is not None for l, p in zip(local, self.params) if p.requires_grad)))
        self.opt.step()


# ------------------------------------------------------------------------------------------------------------------------------
def roofline_from_records(recs, workload, traffic_table, traffic_source, dt_pass):
    """recs: [(kernel, flop, bytes, ms)] from the library's launch profiler over ONE instrumented pass of a micro-batch."""
    per = {}
    for name, flops, nbytes, ms in recs:
        a = per.setdefault(name, [0.0, 0.0, 0.0, 0])
        a[0] += flops; a[1] += nbytes; a[2] += ms * 1e-3; a[3] += 1
    if not per:
        return None
    dom = max(per, key=lambda k: per[k][2])
    fl, by, tm, n = per[dom]
    split = ",true," in dom or dom.endswith("true>") or dom.startswith(("conv_pp_kernel", "conv_halo_kernel")) or "_f16_kernel" in dom      # split-fp16 (f16x3) kernels
    # wgrad_kernel / wgrad3x3_halo_kernel: exact-fp32 MFMA; wgrad_f16_kernel / wgrad3x3_halo_f16_kernel: split-fp16 x3 (LDS transpose reads)
    gemm = dom.startswith(("conv_igemm_kernel", "conv_pp_kernel", "conv_halo_kernel", "flame_blend_skin", "flame_bwd", "wgrad_kernel", "wgrad_f16_kernel", "wgrad3x3_halo"))
    traffic = None
    if traffic_table:
        key = dom.split("[")[0]                                   # the profiler appends a "[tile,waves,stages]" tag to some names
        k2 = key.replace(", ", ",")
        t = traffic_table.get(dom) or traffic_table.get(key) or traffic_table.get(k2)
        if t is None and k2.endswith(">"):                        # rocprofv3 spells defaulted template arguments out (wgrad_f16_kernel<128,2> is <128,2,false> there)
            for extra in (",false>", ",false,false>", ",0>"):
                t = t or traffic_table.get(k2[:-1] + extra)
        traffic = (t["bytes_per_launch"] if isinstance(t, dict) else t) if t is not None else None
    # a GEMM-shaped launch whose arithmetic intensity sits below the ridge point (MFMA peak of its arithmetic mode / HBM peak) is priced against HBM:
    # the encoder's 1x1 convolutions over 16-72 channels move 4 bytes per 8-36 flop
    if gemm and fl > 0 and by > 0 and fl / by < ((PEAK_F16_MFMA / 3.0) if split else PEAK_FP32_MFMA) / PEAK_HBM:
        gemm = False
    if gemm and fl > 0:
        peak = PEAK_F16_MFMA if split else PEAK_FP32_MFMA
        roof = {"bound": "mfma", "kernel": dom, "achieved": fl / tm / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": fl / tm / peak,
                "traffic": traffic, "flop_per_launch": fl / n, "algorithmic_bytes_per_launch": by / n if by else None,
                "mfma_issue_frac": (3.0 if split else 1.0) * fl / tm / peak,
                "note": ("achieved = ALGORITHMIC 2*M*N*K flop per launch / HIP-event launch time; the split-fp16 kernel issues 3 fp16 MFMAs per product "
                         "(hi.hi, hi.lo, lo.hi), so its matrix-pipe occupancy is mfma_issue_frac; peak = dense fp16 MFMA at the nominal 2.4 GHz - a separate "
                         "GRBM_GUI_ACTIVE pass (tools/pmc_clock.py, profiles/r03u_pmc_clock_full.txt, round 3; not re-measured by this run) found the deep-layer kernel "
                         "running at 1.62 GHz under the power cap with its matrix pipe 0.81 busy at that clock.  launches_per_pass / avg_launch_ms / flop_per_launch "
                         "describe the INSTRUMENTED pass, which runs the generator's deep section (H/8 and H/16 layers) as ONE whole-batch chain because the launch "
                         "profiler times launches on one stream; the TIMED region runs that section as two half-batch chains on two streams (csrc/network.hip), "
                         "i.e. twice as many launches of half the flop each, overlapping each other and the next pass's front end - `rocprofv3` below is the "
                         "per-launch average of that state") if split else
                        "achieved = algorithmic flop per launch / HIP-event launch time; peak = f32-input MFMA (v_mfma_f32_32x32x2_f32)"}
    else:
        roof = {"bound": "hbm", "kernel": dom, "achieved": by / tm / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": by / tm / PEAK_HBM,
                "traffic": traffic, "bytes_per_launch": by / n,
                "note": "achieved = algorithmic bytes (unique operands in + out) per launch / HIP-event launch time" if by else
                        "the dispatcher states no algorithmic byte count for this kernel"}
    if roof["frac"] < 0.05:
        # a dominant kernel below 5 % of the roof it is priced against is bound by neither: its launches are short dependent steps (one 7 x 7 image per workgroup,
        # a few hundred workgroups per launch) whose cost is launch + fill + drain latency.  Say so, and report what such a step is made of: how many launches the
        # pass chains and how long the chain takes end to end (the three backbones run on three streams, so the pass's wall time is its critical path).
        roof["priced_against"] = {"bound": roof["bound"], "achieved": roof["achieved"], "peak": roof["peak"], "unit": roof["unit"], "frac": roof["frac"]}
        roof["bound"] = "latency"
        roof["launches_on_critical_path"] = sum(v[3] for v in per.values())
        roof["critical_path_us"] = dt_pass * 1e6
        roof["note"] = ("dominant kernel at < 0.05 of the %s roof: latency-bound (short dependent launches); `achieved` / `peak` / `frac` keep the figure against that roof "
                        "for the record, the meaningful quantities are launches_on_critical_path (library launches per pass) and critical_path_us (wall time of one pass in "
                        "the timed region)" % roof["priced_against"]["bound"])
    roof["rocprofv3"] = rocprofv3_avg(workload, dom)
    if workload == "full":
        roof["rocprofv3_serial"] = rocprofv3_avg(workload, dom, "_serial")
        us = roof["rocprofv3_serial"].get("avg_launch_us")
        if us and roof.get("flop_per_launch"):
            roof["rocprofv3_serial"]["frac_from_this_average"] = roof["flop_per_launch"] / (us * 1e-6) / (roof["peak"] * 1e12)
    roof.update(launches_per_pass=n, launches_total_per_pass=sum(v[3] for v in per.values()), avg_launch_ms=tm / n * 1e3, traffic_source=traffic_source,
                kernel_name_source="libsmirk_hip.so launch profiler (smirk_profile_start/stop): the instantiation that was launched, HIP events on its launch stream",
                kernels={k: {"ms_per_pass": round(v[2] * 1e3, 4), "launches": v[3], **({"tflops": round(v[0] / v[2] / 1e12, 2)} if v[0] > 0 else {}),
                             **({"gbps": round(v[1] / v[2] / 1e9, 1)} if v[1] > 0 else {})} for k, v in sorted(per.items(), key=lambda t: -t[1][2])[:28]},
                profiled_kernel_ms_per_pass=sum(v[2] for v in per.values()) * 1e3, wall_ms_per_pass=dt_pass * 1e3)
    return roof

# ------------------------------------------------------------------------------------------------------------------------------
# "also": the other BASELINE configs, timed in the same process AFTER the headline's timed region (the driver only ever runs `bench.py --gpus 1`)
# ------------------------------------------------------------------------------------------------------------------------------
ALSO_SPECS = (   # name, argv of the child run, steps, warmup, frames of the child's CPU baseline (0: the entry names the line that holds the same CPU path)
    ("flame512", ["--workload", "flame512"], 50, 10, 512),
    ("infer256", ["--workload", "infer256"], 20, 5, 24),
    ("train64_f16x3", ["--workload", "train64", "--train-arith", "f16x3"], 10, 3, 8),
    ("train64_f16x1", ["--workload", "train64", "--train-arith", "f16x1"], 10, 3, 0),
    ("full_shard128_collective", ["--workload", "full", "--global-batch", "128", "--force-collective"], 20, 5, 0),
)
ALSO_CPU_SAME_AS = {"train64_f16x1": "also.train64_f16x3.cpu_baseline (the CPU path has one arithmetic: torch-CPU fp32 autograd)",
                    "full_shard128_collective": "cpu_baseline of this line (same workload, config 4)"}


def run_also(args):
    """BASELINE configs 2, 3, 5 (both training arithmetics) and the per-rank shard of the 8-GPU job (128 frames, the asynchronous all-gather really enqueued through
    a world-size-1 RCCL group), each as its OWN `python bench.py --workload ...` process started after the headline's timed region: exactly the command a builder
    would run by hand, so the numbers are the stand-alone numbers.  (Round 5 first ran them inside this process: after the 1024-frame headline had been through the
    allocator every streaming kernel of the later workloads ran at ~0.65 of its stand-alone rate — train64 67.5 ms in-process against 43.6 ms alone on the same box,
    infer256 38.8 k against 43.5 k, profiles/r05b_* — so the in-process numbers described the bench process, not the library.)  Each child computes its own roofline
    from the launch profiler (no counter passes) and times its own bounded CPU baseline after its timed region (flame512: 512 parameter vectors through the torch
    restatement of the reference FLAME.forward; infer256: 24 frames; train64: 8 frames, shared by both arithmetics — about 30 s in all); an entry that fails reports the error instead of taking the headline down."""
    out, t_all = {}, time.perf_counter()
    for name, argv, steps, warmup, cpu_faces in ALSO_SPECS:
        t_entry = time.perf_counter()
        # counter traffic of the child's dominant kernel: from the committed table of the same kernel sources (profiles/pmc_traffic_<workload>.json, written by a
        # `--traffic measure` run of the same command line; withheld when its source sha differs) — two rocprofv3 passes per child would triple the run
        # (the 128-frame shard's launches are not the 1024-frame table's launches: no traffic for that entry)
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--steps", str(steps), "--warmup", str(warmup), "--traffic", "off" if "--global-batch" in argv else "file",
                                                                    "--cpu-faces", str(cpu_faces if args.cpu_faces > 0 else 0), "--cpu-passes", "3", "--no-also",
                                                                    "--flame-basis", args.flame_basis]
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, cwd=REPO)
            lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError(f"rc={r.returncode}: {r.stderr.decode(errors='replace')[-300:]}")
            j = json.loads(lines[-1])
            roof = j.get("roofline") or {}
            e = {"metric": j["metric"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"], "warmup": j["warmup"],
                 "frames_per_step": j["config"]["frames_per_gpu_per_step"], "dtype": j["dtype"], "command": "python bench.py " + " ".join(cmd[2:]),
                 "host_enqueue_ms_one_step_idle_queue": j.get("host_enqueue_ms_one_step_idle_queue"),
                 "roofline": {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "launches_per_pass",
                                                       "avg_launch_ms", "profiled_kernel_ms_per_pass")},
                 "launches_per_step": roof.get("launches_total_per_pass"),
                 "cpu_baseline": j.get("cpu_baseline") or ALSO_CPU_SAME_AS.get(name)}
            for k in ("critical_path_us", "launches_on_critical_path"):
                if roof.get(k) is not None:
                    e["roofline"][k] = roof[k]
            out[name] = e
        except Exception as ex:                     # noqa: BLE001 — the headline line must still be produced
            out[name] = {"error": f"{type(ex).__name__}: {str(ex)[:400]}"}
        out[name]["wall_s"] = round(time.perf_counter() - t_entry, 2)
    out["_note"] = ("each entry is a separate `python bench.py --workload ...` process run after the headline's timed region (inputs resident, same step / drain / "
                    "synchronize bracketing, roofline from the launch profiler); full-length builder-side lines of the same commands are under profiles/; total "
                    "wall %.1f s" % (time.perf_counter() - t_all))
    return out


def main():
    args = parse_args()
    os.environ["SMIRK_BENCH_FLAME_BASIS"] = args.flame_basis     # read by build_modules (also in the ranks / rocprofv3 children this process launches)
    if args.micro_batch is None:
        args.micro_batch = MICRO_BATCH
    if args.generator_streams is None:
        args.generator_streams = 2      # with one hardware queue per stream two generator streams win at every shard size (profiles/r04q_streams_chains.txt)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    import torch
    import torch.distributed as dist
    rank, world, local = dist_env()
    if args.plumbing_test:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(args.backend, **({"device_id": dev} if dev.type == "cuda" else {}))
    elif args.force_collective and not args.plumbing_test:
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port_ = s_.getsockname()[1]; s_.close()
        dist.init_process_group(args.backend, init_method=f"tcp://127.0.0.1:{port_}", rank=0, world_size=1, device_id=dev)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or let bench.py launch itself)")
    # every rank present and reachable over the collective backend (RCCL on GPUs): an actual all-gather of the rank ids
    ranks_seen = 1
    if world > 1:
        ids = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int64, device=dev))
        ranks_seen = len(set(ids.tolist()))

    from smirk_amd import _lib as L
    sandbox = tempfile.mkdtemp(prefix=f"smirk_bench_r{rank}_")
    cls = (PlumbingTrainWorkload if args.workload == "train64" else PlumbingWorkload) if args.plumbing_test else {"full": FullWorkload, "infer256": InferWorkload, "flame512": FlameWorkload, "train64": TrainWorkload}[args.workload]
    wl = cls(args, dev, rank, world, sandbox)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        wl.step()
    wl.drain()
    sync()
    if wl.last is not None and not args.plumbing_test:
        assert_finite(wl.last, wl.keys, "after warm-up")
    if args.pmc_inner:                      # the pass rocprofv3 wraps: one more plain step, nothing else
        for _ in range(args.steps):
            wl.instrumented()
        sync()
        return
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    t_host = time.perf_counter() - t0       # host time to ENQUEUE the K steps (launches are asynchronous)
    wl.drain()                              # every frame of the K steps is fully processed (and gathered) inside the timed region
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stats = {}
    if wl.last is not None and not args.plumbing_test:
        assert_finite(wl.last, wl.keys, "after the timed steps")
        stats = output_stats(wl.last)
    # host time to enqueue ONE step into an EMPTY launch queue (after the synchronisation above, outside the timed region).  Over the K back-to-back steps of
    # the timed region the launch queue fills and the host blocks on it, so t_host / K measures back-pressure, not enqueue cost (round 4: 33.9 ms in the
    # driver's 20-step run against 2.2 ms in a 5-step run of the same code); both are reported, under names that say which is which.
    t_enq1 = None
    if not args.plumbing_test and not args.pmc_inner:
        te = time.perf_counter()
        wl.step()
        t_enq1 = time.perf_counter() - te
        wl.drain(); sync()

    roof = None
    if rank == 0 and not args.no_roofline and not args.plumbing_test:
        # ---- one extra instrumented pass of a micro-batch: HIP events around every library launch, on the launch stream -------------
        L.profile_start()
        wl.instrumented(); wl.drain(); torch.cuda.synchronize()
        recs = L.profile_stop()
        mode = args.traffic or ("measure" if world == 1 else "file")
        table, src = None, "not collected"
        if mode == "measure":
            try:
                table = measure_traffic(args)
                src = ("measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two separate passes, --kernel-trace only) over one step; "
                       "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md; Infinity-Cache hits included)")
                try:
                    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
                    json.dump({"kernel_sources_sha": kernel_sources_sha(), "workload": args.workload, "kernels": table},
                              open(os.path.join(REPO, "gpurun_out", f"pmc_traffic_{args.workload}.json"), "w"), indent=1)
                except OSError:
                    pass
            except Exception as e:          # noqa: BLE001 — the bench line must still be produced
                src = f"rocprofv3 pass failed ({type(e).__name__}: {str(e)[:200]})"
                mode = "file"
        if table is None and mode == "file":
            try:
                j = json.load(open(os.path.join(REPO, "profiles", f"pmc_traffic_{args.workload}.json")))
                if j.get("kernel_sources_sha") == kernel_sources_sha() and j.get("workload", "full") == args.workload:
                    table = j["kernels"]
                    src = ((src + "; ") if src != "not collected" else "") + (f"profiles/pmc_traffic_{args.workload}.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a "
                           f"`bench.py --workload {args.workload} --traffic measure` run on the same kernel sources (sha " + j["kernel_sources_sha"] +
                           "); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch")
                else:
                    src += f"; profiles/pmc_traffic_{args.workload}.json was measured on different kernel sources / workload -> traffic withheld (null)"
            except Exception:               # noqa: BLE001
                src += f"; no profiles/pmc_traffic_{args.workload}.json"
        roof = roofline_from_records(recs, args.workload, table, src, dt / args.steps / max(1, len(getattr(wl, "slices", [0]))))
        if roof is not None and hasattr(wl, "instrumented_pipeline"):
            # the same kernel inside the overlapped schedule of the timed region (rocprofv3 --kernel-trace --stats of this command averages THIS state)
            L.profile_start()
            ran = wl.instrumented_pipeline(); torch.cuda.synchronize()
            recs2 = [r for r in L.profile_stop() if r[0] == roof["kernel"]]
            if ran and recs2:
                tm, fl = sum(r[3] for r in recs2) * 1e-3, sum(r[1] for r in recs2)
                roof["in_pipeline"] = {"avg_launch_ms": tm / len(recs2) * 1e3, "achieved": fl / tm / 1e12 if fl > 0 else None,
                                       "frac": fl / tm / (roof["peak"] * 1e12) if fl > 0 and roof["unit"] == "TFLOP/s" else None, "launches": len(recs2),
                                       "note": "the dominant kernel while the generator of one pass overlaps encode + FLAME + render of the next (the timed region's steady "
                                               "state; rocprofv3's per-kernel average of this command is taken in this state); `achieved` / `frac` above are the kernel alone"}

    also = None
    if (rank == 0 and world == 1 and args.workload == "full" and not args.no_also and not args.plumbing_test and args.global_batch is None
            and args.batch is None and not args.force_collective):
        B_headline, gen_prec_headline = wl.B, getattr(getattr(wl, "gen", None), "precision", None)
        wl_keep = type("Done", (), {"B": B_headline, "gen": type("G", (), {"precision": gen_prec_headline})(), "buckets": 0, "graphs": False})()
        wl = wl_keep                          # release the headline's modules and its ~64 GB of activations before the other configs run
        import gc
        gc.collect(); torch.cuda.empty_cache()
        also = run_also(args)
    if rank == 0:
        B = wl.B
        faces = B * world * args.steps
        value = faces / dt
        cpu = None
        if world == 1 and args.cpu_faces > 0 and not args.plumbing_test:
            cpu = cpu_baseline(sandbox, args.workload, args.cpu_faces if args.workload != "flame512" else 512, all_cores=args.cpu_all_cores, passes=args.cpu_passes)
        weak = args.batch is not None or (args.workload == "train64" and args.global_batch is None)
        flop_face = {"full": FLOP_PER_FACE, "infer256": FLOP_PER_FACE_INFER, "flame512": FLOP_PER_FACE_FLAME, "train64": FLOP_PER_FACE_TRAIN}[args.workload]
        gen_prec = getattr(getattr(wl, "gen", None), "precision", None)
        line = {
            "metric": METRIC[args.workload] if not args.plumbing_test else "PLUMBING TEST (CPU stub of the path; not a measurement)",
            "value": value, "unit": "faces/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": ("f16 products with f32 accumulate (--train-arith f16x1: one fp16 MFMA per product block on the hi halves of the split16 operands; the reference "
                      "config trains under bf16 autocast): convolutions, their data gradients and their weight gradients; depthwise / stem weight gradients f32 VALU, "
                      "BatchNorm statistics f64, FLAME / raster f32, Adam f32" if args.workload == "train64" and args.train_arith == "f16x1" else
                      "f32-class throughout (reference config: bf16 autocast): convolutions, their data gradients and their weight gradients as split-fp16 x3 MFMA "
                      "with f32 accumulate (depthwise / stem weight gradients f32 VALU), BatchNorm statistics f64, FLAME / raster f32, Adam f32" if args.workload == "train64" else
                      "f32 results; encoder + generator convs as split-fp16 x3 MFMA with f32 accumulate (fp32-class error), FLAME / raster f32"
                      if gen_prec == "f16x3" or args.workload == "infer256" else "f32"),
            "data": "synthetic",
            "config": {"workload": {"full": "BASELINE config 4: full inference incl. SmirkGenerator re-synthesis, 1024-frame batch sharded over the GPUs",
                                    "infer256": "BASELINE config 3: full inference (encoder + FLAME + renderer), batch 256",
                                    "flame512": "BASELINE config 2: FLAME-only, batch 512 random (shape, exp, pose, jaw, eyelid) -> 5023 vertices",
                                    "train64": "BASELINE config 5: smirk_trainer.py cycle-path training step (render + generate + re-encode + losses + backward + "
                                               "clip + Adam), 64 frames per GPU, data-parallel"}[args.workload],
                       "frames_per_gpu_per_step": B, "global_batch": B * world, "micro_batch": min(args.micro_batch, B), "image": "224x224",
                       "parallelism": f"dp{world}", "rccl_ranks_seen": ranks_seen,
                       "collective": ("async all_gather_into_tensor(vertices, rendered_img, reconstructed_img) per micro-batch" if world > 1 else
                                      "async all_gather_into_tensor(...) per micro-batch through a world-size-1 RCCL group (--force-collective)"
                                      if getattr(args, "force_collective", False) else "none (1 GPU)")
                       if args.workload == "full" or (args.plumbing_test and args.workload != "train64") else
                       (f"bucketed all_reduce of the gradients after backward ({getattr(wl, 'buckets', 0)} buckets of <= 64 MiB)" if world > 1 else "none (1 GPU)")
                       if args.workload == "train64" else "none (outputs stay on the rank)",
                       "weights": "random-init (He) reference architecture, encoder heads rescaled to the trained network's parameter ranges; no checkpoint offline; outputs asserted finite",
                       **({"masking": "given masked image" if args.given_masked else
                           "utils/masking.py stage on GPU (mesh-based point sampling + masking) from a synthetic hull mask",
                           "schedule": (f"software pipeline: generator(micro-batch i) on {args.generator_streams} stream(s) || encode+FLAME+render(micro-batch i+1)" if args.overlap else "serial stages")}
                          if args.workload == "full" else
                          {"schedule": ("forward + backward of the two CNNs replayed from four HIP graphs (torch.cuda.make_graphed_callables over the HIP autograd "
                                        "functions); FLAME / renderer / masking / loss / clip / Adam launched eagerly; roofline pass kernel by kernel"
                                        if getattr(wl, "graphs", False) else "every kernel launched from Python (ctypes)")}
                          if args.workload == "train64" else
                          {"schedule": (f"consecutive 256-frame batches rotate over {args.infer_lanes} streams (independent batches: the raster tail of batch i runs under the "
                                        "backbones of batch i+1)" if args.infer_lanes > 1 else "one batch at a time")}
                          if args.workload == "infer256" else {}),
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")},
            "host_enqueue_ms_one_step_idle_queue": (t_enq1 * 1e3 if t_enq1 is not None else None),
            "host_submit_ms_per_step_incl_queue_backpressure": t_host / args.steps * 1e3,
            "path_tflops_per_gpu": value / world * flop_face / 1e12,
            "path_frac_of_f16_mfma_peak": value / world * flop_face / PEAK_F16_MFMA,
            "output_stats": stats, "roofline": roof, "cpu_baseline": cpu}
        if also is not None:
            line["also"] = also
        if args.plumbing_test and args.workload == "train64":
            sd = wl.net.state_dict()
            line["plumbing"] = {"buckets": wl.buckets, "averaged_gradients_ok": wl.checked, "steps_checked": len(wl.checked),
                                "weights_checksum": float(sum(v.double().sum() for v in sd.values()))}
        elif args.plumbing_test:
            line["plumbing"] = {"gathered_ids_last": wl.seen[-1], "micro_batches_per_step": len(wl.slices), "gathers": len(wl.seen)}
        try:                                 # RCCL prints its version banner through C stdio: flush it out BEFORE the JSON line, which stays the last line on stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:                    # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
