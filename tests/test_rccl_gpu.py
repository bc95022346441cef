"""RCCL executed on the hardware that is there: a world-size-1 `nccl` process group on one MI355X drives the two exchange steps of the path —
the asynchronous output all-gather (smirk_amd.pipeline.OutputGatherer -> all_gather_into_tensor) and the bucketed gradient
all-reduce of the cycle step (smirk_amd.cycle.allreduce_gradients -> all_reduce) — through real RCCL kernels with async_op=True, and checks
results and stream ordering.  (N > 1 is covered under gloo in tests/test_distributed_cpu.py; the driver runs the 8-GPU scaling bench.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    yield
    dist.destroy_process_group()


def test_output_gatherer_runs_a_real_rccl_all_gather_async(nccl_world1):
    from smirk_amd.pipeline import OutputGatherer
    g = OutputGatherer(force_collective=True)
    assert g.world == 1 and g.force
    side = torch.cuda.Stream()
    for it in range(3):
        with torch.cuda.stream(side):                                   # outputs produced on a NON-default stream, as the overlapped pipeline does
            a = torch.full((64, 5023, 3), 0.0, device="cuda")
            for _ in range(20):                                         # a chain of kernels the collective must wait for
                a = a + 0.05 * (it + 1)
            out = dict(vertices=a, rendered_img=torch.rand(64, 3, 224, 224, device="cuda"), reconstructed_img=torch.rand(64, 3, 224, 224, device="cuda"))
            g.wait()
            g.start(out)                                                # c10d orders the communication stream after `side`
            assert len(g.pending) == 3 and all(hasattr(w, "wait") for w in g.pending)
            bufs = g.wait()
            for k, t in out.items():
                assert bufs[k].data_ptr() != t.data_ptr()               # a real gather into the pooled [world * n, ...] buffer, not the short-circuit
                assert bufs[k].shape == t.shape and torch.equal(bufs[k], t), k
            assert torch.allclose(bufs["vertices"], torch.full_like(a, 20 * 0.05 * (it + 1)), atol=1e-5)
        side.synchronize()


def test_gradient_allreduce_runs_real_rccl_all_reduce(nccl_world1):
    from smirk_amd.cycle import allreduce_gradients
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(n, device="cuda")) for n in (5_000_000, 12_000_000, 300, 7)]     # 20 + 48 MB: two 64 MiB buckets
    frozen = torch.nn.Parameter(torch.randn(10, device="cuda"), requires_grad=False)
    for p in params[:-1]:
        p.grad = torch.randn_like(p)
    want = [None if p.grad is None else p.grad.clone() for p in params]
    assert allreduce_gradients(params + [frozen]) == 0                                           # default: a world of one short-circuits
    n = allreduce_gradients(params + [frozen], force_collective=True)
    assert n == 2
    torch.cuda.synchronize()
    for p, w in zip(params[:-1], want[:-1]):
        assert torch.equal(p.grad, w)                                                           # sum over one rank / 1
    assert params[-1].grad is not None and float(params[-1].grad.abs().sum()) == 0.0            # missing gradient -> zeros on every rank
    assert frozen.grad is None
