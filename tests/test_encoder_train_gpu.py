"""GPU parity of the TRAIN-mode SmirkEncoder (batch-statistics BatchNorm in all three backbones, running-stat update, backward to the parameters and to
the image) — BASELINE config 5, encoder slice (smirk_trainer.py:297-306: the cycle loss on expression / jaw / eyelid / shape parameters).
Arbiter = float64 autograd through oracle/mobilenet_ref.py::SmirkEncoderRef (the restatement of smirk_encoder.py:14-133 on the restated timm
backbones; parity unpinned against timm itself, see that file's header).  Whole-network gradients are judged with the measured fp32-vs-fp64 spread of
the oracle (see tests/test_generator_train_gpu.py's header); the tight per-op bounds are in tests/test_train_ops_gpu.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import assets as A
from oracle import mobilenet_ref as M

pytestmark = pytest.mark.gpu

GRAD_RTOL, GRAD_MEDIAN_RTOL = 5e-2, 1.5e-2
LOSS_W = dict(expression_params=1.0, jaw_params=10.0, eyelid_params=10.0, shape_params=1.0)     # the cycle loss weights, smirk_trainer.py:299-306


def _rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.abs().max().item(), 1e-12)


def _loss(out, tgt):
    return sum(w * F.mse_loss(out[k], tgt[k].to(out[k])) for k, w in LOSS_W.items())


def _oracle(esd, img, tgt, dtype, frozen=()):
    ref = M.SmirkEncoderRef(); ref.load_state_dict(esd); ref = ref.to(dtype).train()
    for name in frozen:
        for p in getattr(ref, name).parameters():
            p.requires_grad_(False)
    x = img.detach().clone().to(dtype).requires_grad_(True)      # clone: .to(float32) would alias `img` and mark IT as requiring grad
    out = ref(x)
    loss = _loss(out, tgt)
    loss.backward()
    return out, loss.item(), x.grad, {k: p.grad for k, p in ref.named_parameters()}, dict(ref.named_buffers())


@pytest.mark.parametrize("B,HW", [(4, 96), (2, 128)])
def test_encoder_train_step_vs_float64_oracle(B, HW):
    from smirk_amd import SmirkEncoder
    esd = M.synth_encoder_state_dict()
    img = A.synth_images(B, seed=900 + B)[:, :, 40:40 + HW, 50:50 + HW].contiguous()
    g = torch.Generator().manual_seed(5)
    tgt = dict(expression_params=torch.randn(B, 50, generator=g), jaw_params=torch.rand(B, 3, generator=g) * 0.2, eyelid_params=torch.rand(B, 2, generator=g),
               shape_params=torch.randn(B, 300, generator=g) * 0.5)
    frozen = ("pose_encoder",)                                   # config_train.yaml: optimize_pose False -> freeze_module (requires_grad False, BN still in train mode)
    o64, l64, dx64, g64, b64 = _oracle(esd, img, tgt, torch.float64, frozen)
    o32, l32, dx32, g32, _ = _oracle(esd, img, tgt, torch.float32, frozen)
    enc = SmirkEncoder(); enc.load_state_dict(esd, strict=True); enc = enc.cuda().train()
    for p in enc.pose_encoder.parameters():
        p.requires_grad_(False)
    x = img.detach().cuda().requires_grad_(True)
    out = enc(x)
    loss = _loss(out, {k: v.cuda() for k, v in tgt.items()})
    loss.backward()
    for k in o64:
        ref_e = (o32[k].double() - o64[k]).abs().max().item()
        assert (out[k].detach().cpu().double() - o64[k].detach()).abs().max().item() < max(2e-4, 3 * ref_e), k
    assert abs(loss.item() - l64) < 1e-3 * max(1.0, abs(l64))
    e_dx, r_dx = _rel(x.grad.cpu(), dx64), _rel(dx32, dx64)
    errs, refs = {}, {}
    gmax = max(v.abs().max().item() for v in g64.values() if v is not None)
    for k, p in enc.named_parameters():
        if g64[k] is None:
            assert p.grad is None, k                              # frozen pose encoder: no gradient is produced
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        if g64[k].abs().max().item() < 1e-6 * gmax:               # mathematically zero gradients (a conv feeding BatchNorm is scale-free along its own
            assert p.grad.abs().max().item() < 1e-4 * gmax, k     # weight direction ...): only smallness can be asserted, a ratio is noise / noise
            continue
        errs[k], refs[k] = _rel(p.grad.cpu(), g64[k]), _rel(g32[k], g64[k])
    med, rmed = float(np.median(list(errs.values()))), float(np.median(list(refs.values())))
    print(f"vs float64: dimg {e_dx:.2e} (fp32 oracle {r_dx:.2e}); parameters median {med:.2e} max {max(errs.values()):.2e} "
          f"(fp32 oracle: median {rmed:.2e} max {max(refs.values()):.2e})")
    assert e_dx < max(GRAD_RTOL, 3 * r_dx)
    assert max(errs.values()) < max(GRAD_RTOL, 3 * max(refs.values())), max(errs, key=errs.get)
    assert med < max(GRAD_MEDIAN_RTOL, 3 * rmed)
    for k, b in enc.named_buffers():                              # running statistics of ALL three backbones moved (frozen ones included)
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert (b.cpu().double() - b64[k]).abs().max().item() < 1e-4 * max(1.0, b64[k].abs().max().item()), k
        elif k.endswith("num_batches_tracked"):
            assert int(b) == 1, k


def test_encoder_train_then_eval_uses_the_updated_running_statistics():
    from smirk_amd import SmirkEncoder
    esd = M.synth_encoder_state_dict()
    enc = SmirkEncoder(); enc.load_state_dict(esd, strict=True); enc = enc.cuda().train()
    img = A.synth_images(3, seed=77)[:, :, :96, :96].contiguous().cuda()
    with torch.no_grad():
        o1 = enc(img)
    assert not o1["expression_params"].requires_grad
    enc.eval()
    with torch.no_grad():
        o2 = enc(img)
    ref = M.SmirkEncoderRef(); ref.load_state_dict({k: v.detach().cpu() for k, v in enc.state_dict().items()}); ref.eval()
    with torch.no_grad():
        r = ref(img.cpu())
    for k in r:
        assert (o2[k].cpu() - r[k]).abs().max().item() < 2e-3, k


def test_encoder_train_three_streams_equal_serial(monkeypatch):
    """TRAIN-mode SmirkEncoder.forward forks the three backbones onto three HIP streams (smirk_amd/smirk_encoder.py) and autograd runs each backward on
    its forward's stream: outputs, image gradient and parameter gradients must be those of the one-stream order, bit for bit, repeatedly."""
    from smirk_amd import SmirkEncoder
    esd = M.synth_encoder_state_dict()
    g = torch.Generator().manual_seed(5)
    img = torch.rand(3, 3, 96, 96, generator=g).cuda()
    tgt = {"expression_params": torch.randn(3, 50, generator=g), "jaw_params": torch.rand(3, 3, generator=g) * 0.2,
           "eyelid_params": torch.rand(3, 2, generator=g), "shape_params": torch.randn(3, 300, generator=g)}

    def run():
        enc = SmirkEncoder(); enc.load_state_dict(esd); enc = enc.cuda().train()
        for p in enc.pose_encoder.parameters():
            p.requires_grad_(False)
        x = img.clone().requires_grad_(True)
        out = enc(x)
        _loss(out, tgt).backward()
        torch.cuda.synchronize()
        return ({k: v.detach().clone() for k, v in out.items()}, x.grad.clone(),
                {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}, {k: b.clone() for k, b in enc.named_buffers()})

    monkeypatch.setenv("SMIRK_ENCODER_TRAIN_SERIAL", "1")
    ref = run()
    monkeypatch.delenv("SMIRK_ENCODER_TRAIN_SERIAL")
    for _ in range(3):
        got = run()
        assert all(torch.equal(ref[0][k], got[0][k]) for k in ref[0])
        assert torch.equal(ref[1], got[1])
        assert ref[2].keys() == got[2].keys() and all(torch.equal(ref[2][k], got[2][k]) for k in ref[2])
        assert all(torch.equal(ref[3][k], got[3][k]) for k in ref[3])
