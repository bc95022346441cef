"""N>1 bookkeeping on CPU with gloo (world_size 2): shard bounds and the output all-gather used by bench.py / the pipeline."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smirk_amd.pipeline import OutputGatherer, shard_bounds


def test_shard_bounds_cover_and_balance():
    for total in (0, 1, 7, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _worker(rank, world, port, q, done=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total, n = 6, 3
        lo, hi = shard_bounds(total, rank, world)
        full_v = torch.arange(total * 4 * 3, dtype=torch.float32).reshape(total, 4, 3)
        full_i = torch.arange(total * 3 * 2 * 2, dtype=torch.float32).reshape(total, 3, 2, 2) * 0.5
        g = OutputGatherer(keys=("vertices", "rendered_img", "reconstructed_img"))
        for it in range(2):                                      # two rounds: buffers are reused, wait() before start()
            out = {"vertices": full_v[lo:hi] + it, "rendered_img": full_i[lo:hi] - it, "unrelated": torch.zeros(1)}
            g.wait()
            g.start(out)
        bufs = g.wait()
        ok = torch.equal(bufs["vertices"], full_v + 1) and torch.equal(bufs["rendered_img"], full_i - 1) \
            and "reconstructed_img" not in bufs and "unrelated" not in bufs and (hi - lo) == n
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()
    if done is not None:
        done.wait(120)


def test_output_gather_gloo_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(60) for p in procs]
    assert res == [(0, True), (1, True)]


def test_bench_self_launch_world2_gloo_plumbing():
    """`python bench.py --gpus 2` launches itself as two ranks (torch.distributed.run on 127.0.0.1) and drives its rank / shard / micro-batch /
    all-gather / max-over-ranks bookkeeping end to end — here on CPU with gloo and a stub of the path (`--plumbing-test`), the same code
    the GPU run executes around the real pipeline."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--backend", "gloo", "--plumbing-test", "--steps", "3",
                        "--warmup", "1", "--global-batch", "24", "--micro-batch", "4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300, cwd=repo)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["rccl_ranks_seen"] == 2 and j["config"]["global_batch"] == 24
    assert j["config"]["frames_per_gpu_per_step"] == 12 and j["scaling"] == "strong" and j["steps"] == 3
    p = j["plumbing"]
    assert p["micro_batches_per_step"] == 3 and p["gathers"] == 3 * 4
    assert p["gathered_ids_last"] == [8.0, 9.0, 10.0, 11.0, 20.0, 21.0, 22.0, 23.0]      # last micro-batch of rank 0 and of rank 1
    assert abs(j["value"] - 24 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]
    # a launcher-provided WORLD_SIZE that contradicts --gpus is refused instead of silently mis-reporting
    r2 = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--plumbing-test"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, cwd=repo)
    assert r2.returncode != 0 and b"WORLD_SIZE=1" in r2.stderr


def _run_bench(argv, timeout=600, **env_extra):
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", **env_extra)
    return subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=repo)


def test_bench_self_launch_world8_gloo_plumbing_of_the_1024_frame_job():
    """The driver's 8-GPU run (BASELINE config 4: 1024 frames sharded over 8 ranks, 128 per rank) must not be the first time this code path sees 8 ranks:
    `python bench.py --gpus 8 --global-batch 1024 --micro-batch 128` on CPU / gloo with the stub path — shard bounds, gather order, max-over-ranks timing,
    exactly one JSON line."""
    import json
    r = _run_bench(["--gpus", "8", "--backend", "gloo", "--plumbing-test", "--steps", "2", "--warmup", "1", "--global-batch", "1024", "--micro-batch", "128"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["rccl_ranks_seen"] == 8 and j["config"]["global_batch"] == 1024 and j["config"]["parallelism"] == "dp8"
    assert j["config"]["frames_per_gpu_per_step"] == 128 and j["config"]["micro_batch"] == 128 and j["scaling"] == "strong" and j["steps"] == 2
    p = j["plumbing"]
    assert p["micro_batches_per_step"] == 1 and p["gathers"] == 3            # (warm-up + 2 steps) x 1 micro-batch
    assert p["gathered_ids_last"] == [float(i) for i in range(1024)]          # every rank's 128-frame shard, in rank order = frame order
    assert abs(j["value"] - 1024 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]      # whole-job faces / max-over-ranks time
    # two micro-batches per rank: the gather of pass 0 is waited before pass 1 reuses the buffers; last gather = second halves of all shards
    r = _run_bench(["--gpus", "8", "--backend", "gloo", "--plumbing-test", "--steps", "1", "--warmup", "0", "--global-batch", "1024", "--micro-batch", "64"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    j = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
    assert j["plumbing"]["micro_batches_per_step"] == 2
    assert j["plumbing"]["gathered_ids_last"] == [float(128 * rk + 64 + i) for rk in range(8) for i in range(64)]


def test_bench_train64_world2_gloo_plumbing_drives_the_bucketed_allreduce():
    """`bench.py --workload train64` at world 2 (gloo, stub network): the step's bucketed gradient all-reduce (smirk_amd.cycle.allreduce_gradients) runs
    between backward and the optimiser step, frozen parameters stay out, the averaged gradients equal a local recomputation of both ranks' gradients,
    weak scaling (64 frames per rank) is what the line reports."""
    import json
    r = _run_bench(["--gpus", "2", "--backend", "gloo", "--plumbing-test", "--workload", "train64", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["frames_per_gpu_per_step"] == 64 and j["config"]["global_batch"] == 128
    assert "bucketed all_reduce" in j["config"]["collective"]
    p = j["plumbing"]
    assert p["buckets"] >= 2 and p["steps_checked"] == 3 and all(p["averaged_gradients_ok"])
    assert abs(j["value"] - 128 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]


def _grad_worker(rank, world, port, q, done):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from smirk_amd.cycle import allreduce_gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    for p in net[1].parameters():
        p.requires_grad_(False)                                    # a frozen module in the middle (pose / shape encoders)
    x = torch.arange(14, dtype=torch.float32).reshape(2, 7) * (rank + 1)
    net(x).sum().backward()
    net[2].bias.grad = None                                        # a trainable parameter that happened to get no gradient on this rank
    # plain numpy arrays on the queue (pickled by value): torch tensors travel as shared-memory file descriptors that the parent must fetch from a LIVE
    # worker — a worker that exits first resets the connection (`ConnectionResetError` in multiprocessing.reduction.recvfds, 1 run in 7 in round 5)
    local = [None if p.grad is None else p.grad.numpy().copy() for p in net.parameters()]
    nb = allreduce_gradients(list(net.parameters()), bucket_bytes=64)     # tiny buckets: several all-reduces
    q.put((rank, nb, local, [None if p.grad is None else p.grad.numpy().copy() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()
    done.wait(120)                                                 # stay alive until the parent has drained the queue


def test_gradient_allreduce_world2_buckets_and_frozen_parameters():
    """C2: bucketed gradient averaging; frozen parameters are left out, missing gradients count as zero"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, done = ctx.Queue(), ctx.Event()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = [ctx.Process(target=_grad_worker, args=(r, 2, port, q, done)) for r in range(2)]
    for p in ps:
        p.start()
    try:
        res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    finally:
        done.set()
    for p in ps:
        p.join(60)
    assert all(p.exitcode == 0 for p in ps)
    import numpy as np
    (_, nb0, l0, a0), (_, nb1, l1, a1) = res
    assert nb0 == nb1 and nb0 >= 2
    for i, (x0, x1, y0, y1) in enumerate(zip(l0, l1, a0, a1)):
        if i in (2, 3):                                            # the frozen layer
            assert y0 is None and y1 is None
            continue
        z0 = x0 if x0 is not None else np.zeros_like(y0)
        z1 = x1 if x1 is not None else np.zeros_like(y1)
        assert np.allclose(y0, (z0 + z1) / 2) and np.array_equal(y0, y1)
