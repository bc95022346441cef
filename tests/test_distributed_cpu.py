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


def _worker(rank, world, port, q):
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


def test_output_gather_gloo_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(60) for p in procs]
    assert res == [(0, True), (1, True)]
