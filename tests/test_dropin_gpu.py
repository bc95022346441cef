"""The reference callers' patterns, reproduced against the drop-in classes ON THE GPU (the reference scripts themselves cannot travel to the GPU box;
tests/test_dropin_cpu.py runs the real demo.py up to its first device call in the build container):
  demo.py:54-72      strict load of the module-prefixed checkpoint, .eval(), FLAME() / Renderer() from cwd-relative assets, .to(device)
  demo.py:107-169    encoder -> flame.forward(outputs) -> renderer.forward(vertices, cam, landmarks_fan=, landmarks_mp=) -> rendered mask ->
                     mesh_based_mask_uniform_faces -> per-image point budgets -> masking(image, hull_mask[1,H,W], extra_points, ...) -> torch.cat ->
                     smirk_generator(...) -> F.interpolate
  base_trainer.py:236-254   copy.deepcopy(encoder).eval() as the frozen base encoder; load_state_dict(strict=False) of the prefixed dict on a parent module
  smirk_trainer.py:349-355  trainer.train() / .eval() switching before each step
"""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import assets as A
from oracle import generator_ref as G
from oracle import mobilenet_ref as M

pytestmark = pytest.mark.gpu


def _checkpoint():
    ckpt = {"smirk_encoder." + k: v for k, v in M.synth_encoder_state_dict().items()}
    ckpt.update({"smirk_generator." + k: v for k, v in G.synth_state_dict().items()})
    return ckpt


def test_demo_call_pattern(in_sandbox):
    from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator
    from smirk_amd import masking as masking_utils
    device = "cuda"
    checkpoint = _checkpoint()
    smirk_encoder = SmirkEncoder().to(device)
    checkpoint_encoder = {k.replace('smirk_encoder.', ''): v for k, v in checkpoint.items() if 'smirk_encoder' in k}
    smirk_encoder.load_state_dict(checkpoint_encoder)                      # strict
    smirk_encoder.eval()
    smirk_generator = SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5).to(device)
    smirk_generator.load_state_dict({k.replace('smirk_generator.', ''): v for k, v in checkpoint.items() if 'smirk_generator' in k})
    smirk_generator.eval()
    flame = FLAME().to(device)
    renderer = Renderer().to(device)

    cropped_image = A.synth_images(1, seed=3).to(device)                   # demo.py:102-105 hands over [1,3,224,224] in [0,1]
    outputs = smirk_encoder(cropped_image)                                 # NOT under no_grad, exactly like demo.py
    flame_output = flame.forward(outputs)
    renderer_output = renderer.forward(flame_output['vertices'], outputs['cam'],
                                       landmarks_fan=flame_output['landmarks_fan'], landmarks_mp=flame_output['landmarks_mp'])
    rendered_img = renderer_output['rendered_img']
    assert rendered_img.shape == (1, 3, 224, 224) and renderer_output['landmarks_fan'].shape == (1, 68, 2)
    # demo.py:138-167
    hull_mask = np.ones((224, 224), np.float32); hull_mask[60:170, 50:180] = 0
    face_probabilities = masking_utils.load_probabilities_per_FLAME_triangle()
    rendered_mask = 1 - (rendered_img == 0).all(dim=1, keepdim=True).float()
    npoints, _ = masking_utils.mesh_based_mask_uniform_faces(renderer_output['transformed_vertices'], flame_faces=flame.faces_tensor,
                                                             face_probabilities=face_probabilities, mask_ratio=0.05)
    pmask = torch.zeros_like(rendered_mask)
    rsing = torch.randint(0, 2, (npoints.size(0),)).to(npoints.device) * 2 - 1
    rscale = torch.rand((npoints.size(0),)).to(npoints.device) * 4 + 1
    rbound = (npoints.size(1) * (1 / 5) * (rscale ** rsing)).long()
    for bi in range(npoints.size(0)):
        pmask[bi, :, npoints[bi, :rbound[bi], 1], npoints[bi, :rbound[bi], 0]] = 1
    hull = torch.from_numpy(hull_mask).type(dtype=torch.float32).unsqueeze(0).to(device)            # [1,H,W]: 3-D, as the reference passes it
    extra_points = cropped_image * pmask
    masked_img = masking_utils.masking(cropped_image, hull, extra_points, 10, rendered_mask=rendered_mask)
    assert masked_img.shape == cropped_image.shape and not masked_img.requires_grad
    inside = (hull[0] == 0) & (pmask[0, 0] == 0)
    assert (masked_img[0][:, inside] == 0).all()                           # the face region is removed except at the sampled points
    smirk_generator_input = torch.cat([rendered_img, masked_img], dim=1)
    reconstructed_img = smirk_generator(smirk_generator_input)
    assert reconstructed_img.shape == (1, 3, 224, 224) and torch.isfinite(reconstructed_img).all()
    up = F.interpolate(reconstructed_img, (300, 260), mode='bilinear').cpu()
    assert up.shape == (1, 3, 300, 260)
    grid = torch.cat([cropped_image.cpu(), rendered_img.detach().cpu(), reconstructed_img.detach().cpu()], dim=3)
    assert (grid.permute(0, 2, 3, 1).numpy() * 255.0).astype(np.uint8).shape == (1, 224, 672, 3)
    # the oracle agrees with what the demo pattern produced (generator on the drop-in's own inputs)
    y = G.forward({k.replace('smirk_generator.', ''): v for k, v in checkpoint.items() if 'smirk_generator' in k}, smirk_generator_input.detach().cpu())
    assert (reconstructed_img.detach().cpu() - y).abs().max().item() < 5e-5


def test_trainer_call_patterns(in_sandbox):
    from smirk_amd import SmirkEncoder, SmirkGenerator

    class Trainer(nn.Module):                                              # the attribute layout of src/base_trainer.py / src/smirk_trainer.py:15-25
        def __init__(self):
            super().__init__()
            self.smirk_encoder = SmirkEncoder()
            self.smirk_generator = SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
            self.mica = nn.Linear(2, 2)                                    # stands in for the modules the checkpoint does not hold

    t = Trainer().cuda()
    missing, unexpected = t.load_state_dict(_checkpoint(), strict=False)   # base_trainer.py:254
    assert not unexpected and all(k.startswith("mica.") for k in missing)
    t.eval()
    base_encoder = copy.deepcopy(t.smirk_encoder)                          # base_trainer.py:236-238
    base_encoder.eval()
    img = A.synth_images(3, seed=11).cuda()
    with torch.no_grad():                                                  # smirk_trainer.py:40 / :65
        a, b = t.smirk_encoder(img), base_encoder(img)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # parameter groups the optimisers are built from (base_trainer.py:45-49, smirk_trainer.py:335-347)
    n = sum(p.numel() for p in t.smirk_encoder.expression_encoder.parameters())
    assert n == sum(p.numel() for p in M.SmirkEncoderRef().expression_encoder.parameters())
    for p in t.smirk_generator.parameters():                               # smirk_trainer.py:110-112: frozen generator, eval mode
        p.requires_grad_(False)
    t.smirk_generator.eval()
    with torch.no_grad():
        y = t.smirk_generator(A.synth_generator_input(1, seed=2).cuda())
    assert y.shape == (1, 3, 224, 224)
    # smirk_trainer.py:108-113: the frozen, eval-mode generator INSIDE the autograd graph — the loss reaches its input, no parameter gets a gradient
    x = A.synth_generator_input(1, seed=2).cuda().requires_grad_(True)
    out = t.smirk_generator(x)
    assert out.requires_grad and (out.detach() - y).abs().max().item() < 2e-5
    out.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
    assert all(p.grad is None for p in t.smirk_generator.parameters())
    # differentiating through a forward-only CNN must fail loudly, never silently drop the gradient: the eval-mode encoder has no backward
    t.smirk_encoder.eval()
    for p in t.smirk_encoder.parameters():
        p.requires_grad_(True)
    eo = t.smirk_encoder(A.synth_images(1, seed=3).cuda().requires_grad_(True))
    assert eo["expression_params"].requires_grad
    with pytest.raises(NotImplementedError):
        eo["expression_params"].sum().backward()


def test_renderer_full_head_returns_the_shifted_z(in_sandbox):
    """reference quirk (renderer.py:141): with render_full_head=True the in-place `z += 10` lands in the returned transformed_vertices"""
    from smirk_amd import Renderer
    from oracle.flame_ref import FlameRef
    v = torch.from_numpy(FlameRef(in_sandbox).forward(A.synth_flame_params(2, seed=1))["vertices"]).cuda()
    cam = torch.from_numpy(A.synth_cam(2, seed=1)).cuda()
    face, full = Renderer().cuda(), Renderer(render_full_head=True).cuda()
    a, b = face.forward(v, cam), full.forward(v, cam)
    assert torch.equal(a["transformed_vertices"][..., :2], b["transformed_vertices"][..., :2])
    assert torch.allclose(b["transformed_vertices"][..., 2], a["transformed_vertices"][..., 2] + 10)
    assert (b["rendered_img"] != 0).float().mean() >= (a["rendered_img"] != 0).float().mean()      # the whole head covers at least the face


_LATE_IMPORT = r"""
import os, sys, warnings, tempfile
sys.path.insert(0, {repo!r})
os.environ.pop("GPU_MAX_HW_QUEUES", None)
late = {late}
import torch
if late:
    torch.cuda.init(); torch.zeros(1, device="cuda")           # the HIP runtime has read its default of 4 hardware queues
import smirk_amd
from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator
from smirk_amd.pipeline import OverlappedPipeline, SmirkPipeline
from oracle import assets as A, generator_ref as G, mobilenet_ref as M
sb = tempfile.mkdtemp(); A.write_sandbox(sb); os.chdir(sb)
dev = "cuda"
enc = SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict()); enc = enc.to(dev).eval()
gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(G.synth_state_dict()); gen = gen.to(dev).eval()
pipe = SmirkPipeline(enc, FLAME().to(dev), Renderer().to(dev), gen)
img = A.synth_images(8, seed=11).to(dev)
masked = A.synth_generator_input(8, seed=11)[:, 3:].contiguous().to(dev)
serial = pipe(img, masked)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    run = OverlappedPipeline(pipe, generator_streams=2)
warned = any("hardware queues" in str(x.message) for x in w)
outs = []
for _ in range(3):
    d = run.submit(img, masked)
    if d is not None: outs.append(d)
while (d := run.flush()) is not None: outs.append(d)
torch.cuda.synchronize()
same = all(torch.equal(o[k], serial[k]) for o in outs for k in ("vertices", "rendered_img", "reconstructed_img", "expression_params"))
import hashlib
h = hashlib.sha256(b"".join(serial[k].cpu().numpy().tobytes() for k in ("vertices", "rendered_img", "reconstructed_img"))).hexdigest()[:16]
print("RESULT", smirk_amd.HW_QUEUES_STATE, warned, len(outs), same, h)
"""


def test_late_import_warns_and_results_do_not_depend_on_the_hardware_queue_count():
    """`import torch; torch.cuda.init(); import smirk_amd`: the runtime keeps its 4 hardware queues (streams share queues, overlap is lost), the package says so
    through HW_QUEUES_STATE and OverlappedPipeline's warning, and every output is bit-identical to the in-time import (16 queues) and to the serial pipeline."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for late in (True, False):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        r = subprocess.run([sys.executable, "-c", _LATE_IMPORT.format(repo=repo, late=late)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[late] = next(ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")).split()[1:]
    state, warned, n, same, h_late = res[True]
    assert state == "too-late" and warned == "True" and n == "3" and same == "True"
    state, warned, n, same, h_time = res[False]
    assert state == "default-after-torch" and warned == "False" and n == "3" and same == "True"
    assert h_late == h_time
