"""BASELINE config 5 — the cycle path of smirk_trainer.py:293-313,365-370 through the HIP modules: SmirkGenerator (train) -> SmirkEncoder (train) -> cycle loss ->
backward -> clip_grad_norm_(generator, 0.1), pose / shape encoders frozen as config_train.yaml:41-43 has them.
Golden = the REAL reference classes run that way in fp32 and float64 (tests/golden/cycle_golden.npz, oracle/make_cycle_golden.py).  Bounds: loss and the
re-encoded parameters tight; gradients against the float64 run with the spread the reference's own fp32 run shows (median 1.7e-2, max 0.17 on this input —
31 + 3x60 BatchNorm/ReLU layers back to back; see tests/test_generator_train_gpu.py's header), per-op tight bounds in tests/test_train_ops_gpu.py."""
import os

import numpy as np
import pytest
import torch

from oracle import generator_ref as G
from oracle import make_cycle_golden as MC
from oracle import mobilenet_ref as M

pytestmark = pytest.mark.gpu


def test_cycle_path_step_matches_reference_golden(golden_dir):
    from smirk_amd import SmirkEncoder, SmirkGenerator
    from smirk_amd.cycle import cycle_forward
    g = np.load(os.path.join(golden_dir, "cycle_golden.npz"))
    gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(G.synth_state_dict()); gen = gen.cuda().train()
    enc = SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict()); enc = enc.cuda().train()
    for m in (enc.pose_encoder, enc.shape_encoder):
        for p in m.parameters():
            p.requires_grad_(False)
    rendered, masked, feats = MC.inputs()
    loss, recon, out = cycle_forward(gen, enc, rendered.cuda(), masked.cuda(), {k: v.cuda() for k, v in feats.items()})
    assert (recon.detach().cpu() - torch.from_numpy(g["recon"])).abs().max().item() < 2e-5
    for k in out:
        tol = max(2e-4, 3 * float(g["spread/out/" + k]))
        assert (out[k].detach().cpu() - torch.from_numpy(g["out64/" + k])).abs().max().item() < tol, k
    assert abs(loss.item() - float(g["loss64"])) < 3e-4 * abs(float(g["loss64"]))
    loss.backward()
    gnorm = float(torch.nn.utils.clip_grad_norm_(gen.parameters(), 0.1))
    ref_spread = abs(float(g["gen_norm32"]) - float(g["gen_norm64"])) / float(g["gen_norm64"])
    assert abs(gnorm - float(g["gen_norm64"])) / float(g["gen_norm64"]) < max(5e-3, 3 * ref_spread)
    named = [("smirk_generator." + k, p) for k, p in gen.named_parameters()] + [("smirk_encoder." + k, p) for k, p in enc.named_parameters()]
    errs, spreads = {}, {}
    for k, p in named:
        if "nograd/" + k in g.files:
            assert p.grad is None, k
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        sp = float(g["spread/" + k])
        if sp < 0:                                               # mathematically zero gradient: only smallness can be asserted
            continue
        gmax = float(g["gmax64/" + k])
        e = (p.grad.flatten()[:32].cpu() - torch.from_numpy(g["ghead64/" + k])).abs().max().item() / gmax
        if "gfull64/" + k in g.files:
            e = max(e, (p.grad.cpu() - torch.from_numpy(g["gfull64/" + k])).abs().max().item() / gmax)
        e = max(e, abs(p.grad.double().norm().item() - float(g["gnorm64/" + k])) / max(float(g["gnorm64/" + k]), 1e-30))
        errs[k], spreads[k] = e, sp
    med, smed = float(np.median(list(errs.values()))), float(np.median(list(spreads.values())))
    worst = max(errs, key=errs.get)
    print(f"cycle step vs float64: loss {loss.item():.6f} ({float(g['loss64']):.6f}), generator grad norm {gnorm:.1f} ({float(g['gen_norm64']):.1f}); "
          f"{len(errs)} gradient tensors: median {med:.2e} max {errs[worst]:.2e} [{worst}] (reference fp32: median {smed:.2e} max {max(spreads.values()):.2e})")
    assert med < max(1.5e-2, 3 * smed)
    assert errs[worst] < max(5e-2, 3 * max(spreads.values())), worst


def test_graphed_cycle_modules_replay_equals_eager():
    """smirk_amd.cycle.graph_cycle_modules: forward + backward of both CNNs replayed from HIP graphs give the loss, the reconstruction and the parameter
    gradients of the eager launches bit for bit (all kernels are deterministic; BatchNorm in train mode does not read its running statistics), on inputs
    the graphs were NOT captured with, twice in a row (static buffers are reused), with an optimiser-like in-place parameter change in between."""
    from smirk_amd import SmirkEncoder, SmirkGenerator
    from smirk_amd.cycle import cycle_forward, graph_cycle_modules
    torch.manual_seed(0)
    gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(G.synth_state_dict()); gen = gen.cuda().train()
    enc = SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict()); enc = enc.cuda().train()
    for m in (enc.pose_encoder, enc.shape_encoder):
        for p in m.parameters():
            p.requires_grad_(False)
    B, H = 2, 64
    params = [p for p in list(gen.parameters()) + list(enc.parameters()) if p.requires_grad]

    def inputs(seed):
        g = torch.Generator().manual_seed(seed)
        feats = {"expression_params": torch.randn(B, 50, generator=g) * 0.5, "jaw_params": torch.rand(B, 3, generator=g) * 0.1,
                 "eyelid_params": torch.rand(B, 2, generator=g), "shape_params": torch.randn(B, 300, generator=g) * 0.5}
        return torch.rand(B, 3, H, H, generator=g).cuda(), torch.rand(B, 3, H, H, generator=g).cuda(), {k: v.cuda() for k, v in feats.items()}

    def run(g_mod, e_mod, seed):
        for p in params:
            p.grad = None
        rendered, masked, feats = inputs(seed)
        loss, recon, _ = cycle_forward(g_mod, e_mod, rendered, masked, feats)
        loss.backward()
        return loss.detach().clone(), recon.detach().clone(), [torch.zeros_like(p) if p.grad is None else p.grad.detach().clone() for p in params]

    saved = [p.detach().clone() for p in params]

    def set_params(scale):
        with torch.no_grad():
            for p, s0 in zip(params, saved):
                p.copy_(s0 * scale)

    eager = [run(gen, enc, 11), run(gen, enc, 12)]
    set_params(1.001)
    eager.append(run(gen, enc, 13))
    set_params(1.0)
    g_mod, e_mod = graph_cycle_modules(gen, enc, torch.zeros(B, 6, H, H, device="cuda"), torch.zeros(B, 3, H, H, device="cuda"))
    graphed = [run(g_mod, e_mod, 11), run(g_mod, e_mod, 12)]
    set_params(1.001)
    graphed.append(run(g_mod, e_mod, 13))
    for (l0, r0, g0), (l1, r1, g1) in zip(eager, graphed):
        assert torch.equal(l0, l1) and torch.equal(r0, r1)
        assert all(torch.equal(a, b) for a, b in zip(g0, g1))
    assert not torch.equal(eager[0][0], eager[1][0]) and not torch.equal(eager[1][1], eager[2][1])      # the three runs really differ
